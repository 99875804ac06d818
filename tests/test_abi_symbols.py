"""CPU checks of the C-ABI library: it builds for gfx950, loads, and exports every symbol that
include/nerf_mi355x.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "nerf_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nerf_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_binding_surface():
    import nerf_replication_amd._lib as L
    assert sorted(L.EXPORTS) == _declared_symbols()


def test_library_builds_loads_and_exports():
    import nerf_replication_amd._lib as L
    L.build()
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    lib.nerf_abi_version.restype = ctypes.c_int32
    assert lib.nerf_abi_version() == 2
    lib.nerf_packed_model_bytes.restype = ctypes.c_int64
    lib.nerf_packed_model_bytes.argtypes = [ctypes.c_int32]
    # f32: 2 x K=64 layers + 8 x K=256 layers + views (K=288 -> 128) + biases + heads
    assert lib.nerf_packed_model_bytes(0) == 4 * (2 * 64 * 256 + 8 * 256 * 256 + 288 * 128 + 9 * 256 + 128 + 256 + 384 + 4)
    # f16 (both MFMA shapes): 16 KiB const region + 1184 A fragments of 1 KiB (37 chunks of 32), + the split-fp16 stream of the same
    # model behind it (16 KiB + 2368 fragments: the far-plane guard of nerf_render_forward); f32x: that stream alone
    assert lib.nerf_packed_model_bytes(1) == lib.nerf_packed_model_bytes(3) == (16384 + 1184 * 1024) + (16384 + 2368 * 1024)
    assert lib.nerf_packed_model_bytes(2) == 16384 + 2368 * 1024
    assert lib.nerf_packed_model_bytes(7) == -1
    lib.nerf_render_workspace_bytes.restype = ctypes.c_int64
    lib.nerf_render_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    # 4864 B per ray (raw_coarse, t_sorted, raw_fine) + the fp16 far-plane guard's last-sample ids and count; bounded by one ray
    # block of 2^20 rays whatever the frame
    assert lib.nerf_render_workspace_bytes(640000, 128, 0) == 640000 * 4864 + 640000 * 4 + 256
    assert lib.nerf_render_workspace_bytes(10 ** 8, 128, 0) == lib.nerf_render_workspace_bytes(1 << 20, 128, 0)
    assert lib.nerf_render_workspace_bytes(640000, 0, 0) == 640000 * 1024 + 640000 * 4 + 256
    assert lib.nerf_render_workspace_bytes(640000, 128, 1) == 640000 * (4864 + 192 + 768) + 256


def test_product_path_has_no_cpu_fallback(synthetic_sd):
    import torch
    import nerf_replication_amd as pkg
    net = pkg.Network()
    net.load_state_dict(synthetic_sd, strict=True)
    assert list(net.state_dict().keys()) == list(synthetic_sd.keys())
    with pytest.raises(pkg._lib.NerfLibraryError):
        pkg.Renderer(net).render({"rays_o": torch.zeros(1, 4, 3), "rays_d": torch.zeros(1, 4, 3)})
    if not torch.cuda.is_available():
        with pytest.raises(pkg._lib.NerfLibraryError):
            net.packed("")


def test_product_never_imports_oracle():
    pkg_dir = os.path.join(REPO, "nerf_replication_amd")
    for root, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "nerf_oracle" not in src and "oracle/" not in src, f


def test_inline_asm_weight_stream_has_no_register_hazard():
    """The fp32 inference kernel fills its weight ring with inline-asm loads that complete behind the compiler's
    back; tools/check_asm_stream.py compiles the kernel to ISA and proves that no instruction touches a ring
    register between its load and the s_waitcnt that covers it (a spill, copy or reuse there would move stale data)."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    tool = [sys.executable, os.path.join(REPO, "tools", "check_asm_stream.py")]
    # csrc/Makefile runs this very check on every build and records the hash of the sources it proved hazard-free next to
    # the library; when that stamp matches the sources in the tree, the (55 s) positive run is not repeated here
    import glob
    import hashlib
    csrc = os.path.join(REPO, "nerf_replication_amd", "csrc")
    deps = sorted(set(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.inc")) +
                      [os.path.join(REPO, "include", "nerf_mi355x.h"), os.path.join(REPO, "tools", "check_asm_stream.py"),
                       os.path.join(csrc, "Makefile")]), key=lambda p: os.path.relpath(p, csrc))
    h = hashlib.sha256()
    for p in [os.path.join(csrc, "nerf_kernels.hip"), os.path.join(csrc, "nerf_kernels_x.hip")] + deps:      # (csrc/Makefile: SRCS, DEPS)
        h.update(open(p, "rb").read())
    stamp = os.path.join(REPO, "nerf_replication_amd", "libnerf_mi355x.so.checked")
    proven = os.path.exists(stamp) and open(stamp).read().strip() == h.hexdigest() and \
        os.path.exists(os.path.join(REPO, "nerf_replication_amd", "libnerf_mi355x.so"))
    if not proven:
        r = subprocess.run(tool, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.count(" 0 hazards") == 28 and "no asm loads found" not in r.stdout    # 12 fp32 + 8 small-layer + 8 split-fp16 instances
    # the checker itself: re-creating the prefetch past the stream end (whose registers the compiler reuses) must be caught
    bad = subprocess.run(tool, capture_output=True, text=True, env=dict(os.environ, NERF_CHECK_EXTRA_FLAGS="-DNERF_F32_ASM_OVERRUN=1 -DNERF_TIMING_BUILD"))
    assert bad.returncode == 1 and "touched before its wait" in bad.stdout


def test_stray_timing_switch_does_not_build():
    """A numerics-breaking timing switch (-DNERF_*_HACK_*) must not yield a product library: without -DNERF_TIMING_BUILD
    the translation unit #errors (round-1 VERDICT "Weak 9"); with it, nerf_build_flags() reports the build and the Python
    loader refuses it (nerf_replication_amd/_lib.py)."""
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if hipcc is None:
        pytest.skip("hipcc not available")
    src = os.path.join(REPO, "nerf_replication_amd", "csrc", "nerf_kernels.hip")
    base = [hipcc, "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-fsyntax-only", src]
    for flag in ("-DNERF_F32_HACK_NOBIAS=1", "-DNERF_F16_HACK_NOADV=1", "-DNERF_SAVE_TAPS=0"):
        r = subprocess.run(base + [flag], capture_output=True, text=True)
        assert r.returncode != 0 and "timing switch is set" in r.stderr, (flag, r.stderr[-400:])
    ok = subprocess.run(base, capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr[-400:]
