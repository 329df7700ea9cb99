"""sgdnet_amd: MI355X (gfx950) SAGA elastic-net backend behind the sgdnet() API.

The compute lives in sgdnet_amd/lib/libsgdnet_hip.so (hand-written HIP, C ABI in
include/sgdnet_hip.h).  This package is the host-side mirror of the reference's
R front-end for the fit path plus benchmark/multi-GPU plumbing.
"""
from ._lib import LIB_PATH, SgdnetError, load  # noqa: F401
from .api import SgdnetFit, sgdnet  # noqa: F401
from .cv import CvSgdnet, cv_sgdnet  # noqa: F401
from .predict import coef, predict  # noqa: F401
from .score import score  # noqa: F401
from .solver import RRng, SagaSolver, auto_batch, get_option, link_peers, option, set_option, shard_window  # noqa: F401

__all__ = ["sgdnet", "SgdnetFit", "SagaSolver", "RRng", "auto_batch", "SgdnetError", "load", "LIB_PATH",
           "cv_sgdnet", "CvSgdnet", "predict", "coef", "score", "set_option", "get_option", "option"]
