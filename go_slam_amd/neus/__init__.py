"""Mapping hot path: hash-grid NeuS renderer on the HIP library (replaces tiny-cuda-nn use)."""
from .tcnn_compat import Encoding, Network  # noqa: F401
from .instant_neus import InstantNeuS  # noqa: F401
from .render import Renderer  # noqa: F401
