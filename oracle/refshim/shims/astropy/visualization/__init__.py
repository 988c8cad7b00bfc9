from . import lupton_rgb  # noqa: F401
