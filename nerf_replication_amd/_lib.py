"""ctypes binding of libnerf_mi355x.so (include/nerf_mi355x.h).  Plumbing only: device memory,
streams and error translation.  Fails loudly when the library is missing -- no fallback."""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (imported first so that the HIP runtime torch loaded is the one we bind to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnerf_mi355x.so")
CSRC = os.path.join(_HERE, "csrc")

N_SAMPLES = 64
N_IMPORTANCE = 128
PREC_F32 = 0
PREC_F16 = 1
PREC_F32X = 2
PREC_F16S = 3          # fp16 arithmetic on 16x16x32 MFMA tiles (A/B partner of PREC_F16's 32x32x16 tiles)
# "f16" is the fp16-activation path of BASELINE config 5.  Both MFMA shapes are built and tested; the 16x16x32 kernel measured
# 0.3-0.9 % (fine launch) / 3 % (coarse launch) faster in interleaved A/Bs (profiles/r03_f16_variants.csv) and is what "f16"
# selects; "f16m32" selects the 32x32x16 kernel explicitly.
PRECISIONS = {"f32": PREC_F32, "fp32": PREC_F32, "f16": PREC_F16S, "fp16": PREC_F16S, "f16s": PREC_F16S, "f16m32": PREC_F16, "f32x": PREC_F32X}

_c = ctypes
_F = _c.c_void_p   # device pointers travel as integers (tensor.data_ptr())
_PROTOS = {
    "nerf_abi_version": (_c.c_int32, []),
    "nerf_build_flags": (_c.c_int32, []),
    "nerf_last_error": (_c.c_char_p, []),
    "nerf_packed_model_bytes": (_c.c_int64, [_c.c_int32]),
    "nerf_pack_model": (_c.c_int32, [_c.POINTER(_c.c_void_p), _F, _c.c_int32, _c.c_void_p]),
    "nerf_positional_encoding": (_c.c_int32, [_F, _c.c_int64, _c.c_int32, _F, _c.c_void_p]),
    "nerf_mlp_forward": (_c.c_int32, [_F, _F, _c.c_int64, _c.c_int32, _F, _F, _c.c_int32, _c.c_void_p]),
    "nerf_mlp_forward_rays": (_c.c_int32, [_F, _F, _F, _c.c_int64, _c.c_int64, _c.c_int32, _F, _F,
                                           _c.c_int32, _c.c_void_p]),
    "nerf_mlp_forward_rays_for_compositing": (_c.c_int32, [_F, _F, _F, _c.c_int64, _c.c_int64, _c.c_int32, _F, _F,
                                                           _c.c_int32, _c.c_void_p]),
    "nerf_mlp_forward_rays_density": (_c.c_int32, [_F, _F, _F, _c.c_int64, _c.c_int64, _c.c_int32, _F, _F,
                                                   _c.c_int32, _c.c_void_p]),
    "nerf_composite_backward": (_c.c_int32, [_F, _F, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _F, _F, _F, _F,
                                             _c.c_void_p]),
    "nerf_sample_fine_backward": (_c.c_int32, [_F, _F, _F, _c.c_int64, _F, _F, _F, _c.c_void_p]),
    "nerf_adam_step": (_c.c_int32, [_c.c_int32, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_void_p),
                                    _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_int64), _c.c_float, _c.c_float, _c.c_float,
                                    _c.c_float, _c.c_float, _c.c_float, _c.c_int64, _c.c_void_p]),
    "nerf_train_grad_floats": (_c.c_int64, [_c.c_int64]),
    "nerf_train_live_count_offset": (_c.c_int64, [_c.c_int64]),
    "nerf_packed_bwd_bytes": (_c.c_int64, [_c.c_int32]),
    "nerf_pack_model_bwd": (_c.c_int32, [_c.POINTER(_c.c_void_p), _F, _c.c_int32, _c.c_void_p]),
    "nerf_mlp_backward": (_c.c_int32, [_F, _F, _F, _c.c_int64, _c.c_int64, _c.c_int32, _F, _F, _F, _F, _F,
                                       _c.POINTER(_c.c_void_p), _c.c_int32, _c.c_void_p]),
    "nerf_mlp_forward_rays_save_density": (_c.c_int32, [_F, _F, _F, _c.c_int64, _c.c_int64, _c.c_int32, _F, _F, _F, _c.c_int32,
                                                        _c.c_void_p]),
    "nerf_mlp_backward_density": (_c.c_int32, [_F, _F, _F, _c.c_int64, _c.c_int64, _c.c_int32, _F, _F, _F, _F, _F,
                                               _c.POINTER(_c.c_void_p), _c.c_int32, _c.c_void_p]),
    "nerf_mlp_forward_points_save": (_c.c_int32, [_F, _F, _c.c_int64, _c.c_int32, _F, _F, _F, _c.c_int32, _c.c_void_p]),
    "nerf_mlp_backward_points": (_c.c_int32, [_F, _c.c_int64, _c.c_int32, _F, _F, _F, _F, _F,
                                              _c.POINTER(_c.c_void_p), _c.c_int32, _c.c_void_p]),
    "nerf_viewdirs_backward": (_c.c_int32, [_F, _c.c_int64, _c.c_int32, _F, _F, _F, _c.c_void_p]),
    "nerf_wgrad": (_c.c_int32, [_F, _c.c_int64, _c.c_int32, _c.c_int32, _F, _c.c_int64, _c.c_int32, _c.c_int32, _F,
                                _c.c_int64, _c.c_int32, _F, _c.c_int64, _c.c_void_p]),
    "nerf_train_save_floats": (_c.c_int64, [_c.c_int64]),
    "nerf_mlp_forward_rays_save": (_c.c_int32, [_F, _F, _F, _c.c_int64, _c.c_int64, _c.c_int32, _F, _F, _F, _c.c_int32,
                                                _c.c_void_p]),
    "nerf_mlp_forward_rays_save_for_compositing": (_c.c_int32, [_F, _F, _F, _c.c_int64, _c.c_int64, _c.c_int32, _F, _F, _F, _c.c_int32,
                                                _c.c_void_p]),
    "nerf_sample_fine": (_c.c_int32, [_F, _F, _F, _c.c_int64, _F, _F, _F, _c.c_float, _c.c_float, _c.c_void_p]),
    "nerf_composite": (_c.c_int32, [_F, _F, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _F, _F, _F,
                                    _c.c_void_p]),
    "nerf_generate_rays": (_c.c_int32, [_c.POINTER(_c.c_double), _c.c_int32, _c.c_int32, _c.c_double, _c.c_int64,
                                        _c.c_int64, _F, _F, _F, _c.c_void_p]),
    "nerf_image_metrics": (_c.c_int32, [_F, _F, _c.c_int64, _F, _c.c_void_p]),
    "nerf_image_ssim": (_c.c_int32, [_F, _F, _c.c_int32, _c.c_int32, _F, _c.c_void_p]),
    "nerf_render_workspace_bytes": (_c.c_int64, [_c.c_int64, _c.c_int32, _c.c_int32]),
    "nerf_render_forward": (_c.c_int32, [_F, _F, _c.c_int64, _F, _F, _F, _F, _c.c_int32, _c.c_int32,
                                         _c.c_int32, _c.c_int32, _c.c_float, _F, _c.c_int64, _F, _F, _c.c_void_p]),
}
EXPORTS = tuple(_PROTOS)

_lib = None


class NerfLibraryError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-C", CSRC], capture_output=True, text=True)
    if res.returncode != 0:
        raise NerfLibraryError("building libnerf_mi355x.so failed:\n" + res.stdout + res.stderr)
    if verbose:
        print(res.stdout)
    return LIB_PATH


def load():
    """dlopen the library and declare prototypes; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NerfLibraryError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()); "
                "there is no CPU fallback for the render path")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.nerf_abi_version() != 2:
            raise NerfLibraryError("libnerf_mi355x.so ABI version mismatch")
        if lib.nerf_build_flags() != 0:
            raise NerfLibraryError(f"{LIB_PATH} is a timing build (nerf_build_flags() = {lib.nerf_build_flags()}): kernels "
                                   "compiled with NERF_*_HACK_* switches compute wrong results; rebuild with `make -C csrc`")
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().nerf_last_error().decode("utf-8", "replace")
        raise NerfLibraryError(f"{what or 'nerf call'} failed (status {rc}): {msg}")


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous CUDA tensor of `dtype` (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NerfLibraryError("the HIP render path needs tensors on a GPU (cuda) device; got a CPU tensor")
    if t.dtype != dtype or not t.is_contiguous():
        raise NerfLibraryError(f"expected a contiguous {dtype} tensor")
    return t.data_ptr()


def stream_of(device):
    return torch.cuda.current_stream(device).cuda_stream


def packed_model_bytes(precision: int) -> int:
    n = int(load().nerf_packed_model_bytes(precision))
    if n <= 0:
        raise NerfLibraryError(f"unknown precision {precision}")
    return n
