"""MI355X-native NeRF volume-rendering hot path behind the reference's plugin surface.

Public surface mirrors rkin100g/Nerf-Replication (paths relative to the reference root):
    Renderer   <- src/models/nerf/renderer/volume_renderer.py  (Renderer(net).render(batch))
    Network    <- src/models/nerf/network.py                    (Network().forward(...), state_dict keys)
The arithmetic runs in hand-written HIP kernels (csrc/) reached through the C ABI declared in
include/nerf_mi355x.h; there is no CPU or eager-PyTorch fallback: without the built library or a
GPU the product path raises.
"""
from .network import NeRF, Network          # noqa: F401
from .volume_renderer import Renderer       # noqa: F401
from .rays import generate_rays             # noqa: F401
from .evaluator import Evaluator            # noqa: F401
from . import _lib                          # noqa: F401
from .checkpoint import load_network, load_model, save_model   # noqa: F401

__all__ = ["NeRF", "Network", "Renderer", "Evaluator", "generate_rays", "load_network", "load_model", "save_model"]
