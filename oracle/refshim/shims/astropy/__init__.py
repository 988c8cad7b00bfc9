"""Container-only stand-in so that `import astropy` in scarlet/frame.py works."""
from . import wcs, visualization  # noqa: F401
