from oracle.pgm import l2sq  # noqa: F401
