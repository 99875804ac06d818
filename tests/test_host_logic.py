"""CPU-only checks of host-side logic that needs no GPU: the bench's camera pose is the oracle's, the bench's FLOP accounting
follows SURVEY 8(d), the package's lazy attributes are complete, and the C header declares exactly what the ctypes binding binds."""
import importlib.util
import os
import re

import torch

from conftest import REPO


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_pose_and_flop_constants(oracle):
    b = _load_bench()
    c2w = b.camera_pose_40()
    ref = oracle.camera_pose(40.0)[:3, :4].double()
    assert torch.allclose(c2w, ref, atol=1e-6)                       # the frame the bench renders is the frame the tests check
    # 2 x 593 408 MAC per point, 64 + 192 points per ray
    macs = 63 * 256 + 4 * 256 * 256 + 319 * 256 + 2 * 256 * 256 + 256 * 256 + 256 + 283 * 128 + 128 * 3
    assert macs == 593408 and b.FLOP_PER_POINT == 2 * macs and b.POINTS_PER_RAY == 256
    assert b.FLOP_PER_POINT * b.POINTS_PER_RAY == 303824896
    # the colour branch the density-only coarse pass skips: feature + views + rgb
    assert 2 * (256 * 256 + 283 * 128 + 128 * 3) == 204288
    assert b.PEAK_F32_MFMA == 157.3e12 and b.PEAK_F16_MFMA == 2.5e15


def test_header_and_binding_declare_the_same_entry_points():
    import nerf_replication_amd as pkg
    hdr = open(os.path.join(REPO, "include", "nerf_mi355x.h")).read()
    declared = set(re.findall(r"^(?:int32_t|int64_t|const char\*)\s+(nerf_\w+)\(", hdr, flags=re.M))
    assert declared == set(pkg._lib.EXPORTS), (declared ^ set(pkg._lib.EXPORTS))


def test_package_lazy_attributes_are_complete():
    import nerf_replication_amd as pkg
    for name in pkg.__all__:
        assert getattr(pkg, name) is not None
    assert set(pkg.__all__) <= set(dir(pkg))
    assert pkg.training.FusedAdam is not None and pkg.dist.shard_bounds(10, 1, 4) == (3, 6, 3)
