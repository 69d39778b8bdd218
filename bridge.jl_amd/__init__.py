"""bridgehip -- MI355X-native guided-proposal diffusion-bridge sampler (the hot path of
mschauer/Bridge.jl behind its own sample/solve/llikelihood interface).

The directory is called `bridge.jl_amd`; import it as `bridgehip` (repo-root shim `bridgehip.py`).
"""
from .api import *  # noqa: F401,F403
from .api import _cm, _uncm  # noqa: F401
from . import _lib  # noqa: F401
from . import dist  # noqa: F401
from . import coeffs  # noqa: F401
from .coeffs import b, sigma, a, Gamma, constdiff, B, beta, r, H, guided_b  # noqa: F401
