# Overlay for a checkout of the reference: src/models/nerf/network.py -> the HIP-backed Network
# (same class names, forward signature and state_dict keys).
from nerf_replication_amd.network import NeRF, Network  # noqa: F401
