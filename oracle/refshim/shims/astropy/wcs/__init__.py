class WCS:
    pass
