class LinearMapping:
    pass


class AsinhMapping:
    pass
