from scipy.special import *  # noqa: F401,F403
