class ArrayBox:
    """Placeholder: nothing is ever boxed by the shim."""

    @classmethod
    def register(cls, other):
        return other
