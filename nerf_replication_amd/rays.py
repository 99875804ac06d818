"""On-device pinhole ray generation: the formula of
src/datasets/nerf/blender.py:102-127 executed by nerf_generate_rays, so a frame (or a rank's tile of
it) needs only the 3x4 pose and the intrinsics instead of a 15 MB host->device ray copy."""
import ctypes
import math

import torch

from . import _lib


def generate_rays(c2w, H, W, camera_angle_x, device, pixel_begin=0, n_pixels=None, pixel_ids=None):
    """-> rays_o, rays_d [n,3] float32 on `device` for pixels [pixel_begin, pixel_begin+n_pixels) in
    row-major order, or for `pixel_ids` (int64 tensor) when given."""
    lib = _lib.load()
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.NerfLibraryError("generate_rays runs on the GPU only")
    focal = W / (2.0 * math.tan(camera_angle_x / 2.0))
    m = torch.as_tensor(c2w, dtype=torch.float64)[:3, :4].reshape(-1).tolist()    # row-major 3x4
    arr = (ctypes.c_double * 12)(*m)
    ids_ptr = None
    if pixel_ids is not None:
        pixel_ids = pixel_ids.to(device=device, dtype=torch.int64).contiguous()
        n_pixels = pixel_ids.numel()
        ids_ptr = pixel_ids.data_ptr()
    elif n_pixels is None:
        n_pixels = H * W - pixel_begin
    o = torch.empty((n_pixels, 3), dtype=torch.float32, device=device)
    d = torch.empty((n_pixels, 3), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.nerf_generate_rays(arr, H, W, focal, pixel_begin, n_pixels, ids_ptr, o.data_ptr(), d.data_ptr(),
                                          _lib.stream_of(device)), "nerf_generate_rays")
    return o, d
