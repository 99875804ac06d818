"""The reference's OWN rounding floor for the final image, measured -- not prose (round-1 VERDICT "Weak 2d").

SURVEY section 8c proposed `max|d rgb| <= 1e-4, max|d depth| <= 1e-3` per ray.  The reference algorithm cannot meet
that against itself: its inverse-CDF sampler is discontinuous in its own rounding (`denom < 1e-5 -> 1`,
volume_renderer.py:259-260; a searchsorted index flipping next to a nearly empty CDF bin), so a perturbation of
the coarse densities at fp32-rounding level moves single fine samples by up to a bin and the ray's colour with
them.  Probe: the oracle (bit-exact to the reference on these very fixtures, test_oracle_golden.py) evaluates
encoding + MLP in float64 on the same fp32 inputs/weights, everything else unchanged in fp32.  The measured
deviation of the reference from this better-rounded self IS the floor; the GPU parity tests
(tests/test_gpu_parity.py::test_family_parity_attributed) therefore bound non-flipped rays at 1e-4 / 1e-3 and
attribute every larger deviation to moved samples, instead of a blanket per-ray maximum.

Numbers asserted below were measured in the build container (torch 2.10 CPU); ranges leave room for another
CPU's GEMM blocking.  The record goes to profiles/parity_r03.json under "noise_floor" when PARITY_RECORD is set.
"""
import json
import os

import pytest
import torch

from conftest import parity_record

EPS_RGB, EPS_DEP, EPS_T = 1e-4, 1e-3, 1e-4      # SURVEY 8c image tolerances; a sample counts as moved beyond 1e-4


def floor_stats(oracle, sd, o, d):
    with torch.no_grad():
        r32, z32, p32 = oracle.render(sd, o[None], d[None], return_parts=True)
        r64, z64, p64 = oracle.render(sd, o[None], d[None], return_parts=True, mlp_dtype=torch.float64)
    e_rgb = (r32 - r64).abs().max(-1).values
    e_dep = (z32 - z64).abs()
    dt = (p32["t_sorted"] - p64["t_sorted"]).abs()
    flipped = dt.max(1).values > EPS_T
    moved_any = dt.max(1).values > 1e-5
    over = (e_rgb > EPS_RGB) | (e_dep > EPS_DEP)
    return dict(n_rays=int(o.shape[0]), psnr_db=round(oracle.psnr(r32, r64), 2),
                rgb_max=e_rgb.max().item(), rgb_q99=torch.quantile(e_rgb, 0.99).item(),
                depth_max=e_dep.max().item(), depth_q99=torch.quantile(e_dep, 0.99).item(),
                rays_over_tolerance=int(over.sum()), rays_with_moved_samples=int(flipped.sum()),
                moved_samples=int((dt > EPS_T).sum()), max_sample_move=dt.max().item(),
                over_tolerance_but_no_moved_sample=int((over & ~moved_any).sum()))


@pytest.mark.parametrize("family", ["base", "sharp", "white", "trained"])
def test_reference_fp32_vs_fp64_floor(oracle, golden, family_sd, family):
    g = golden(f"render_family_{family}.npz")
    sd = family_sd(family)
    o = torch.cat([g["seed_rays_o"], g["pin_rays_o"]])
    d = torch.cat([g["seed_rays_d"], g["pin_rays_d"]])
    o2, d2 = oracle.seeded_rays(2048, 5)       # more rays for a stable figure (over-tolerance rays are ~1 in 10^3)
    o, d = torch.cat([o, o2]), torch.cat([d, d2])
    st = floor_stats(oracle, sd, o, d)
    print(f"noise floor [{family}]: {json.dumps(st)}")
    if os.environ.get("PARITY_RECORD"):
        parity_record("noise_floor_reference_fp32_vs_fp64_mlp", family, st)
    # (1) samples DO move under fp32-level perturbation, on every synthetic family.  (Not on "trained": its unsupervised coarse
    #     network is a smooth fog, so the coarse pdf has no empty bins and the sampler is well conditioned: floor 8e-6 / 5e-5.)
    if family != "trained":
        assert st["rays_with_moved_samples"] >= 1 and st["max_sample_move"] > 10 * EPS_T
    else:
        assert st["rgb_max"] <= EPS_RGB and st["depth_max"] <= EPS_DEP
    # (2) and whenever a ray leaves the SURVEY tolerance, a moved sample is the cause: rounding alone stays inside
    #     (not asserted on the white-noise field: there the fine network varies on the scale of the 2^9 octave's
    #     wavelength 0.012, so even a 1e-5 move of a sample is visible -- measured: 7e-5 / 9e-4 for unmoved rays)
    if family != "white":
        assert st["over_tolerance_but_no_moved_sample"] == 0
    # (3) a flat per-ray max of 1e-4 / 1e-3 is not meetable by the reference against itself
    #     (asserted on the white-noise field, where ~0.5 % of the rays do it; on the other two it is ~1 ray in 2000 and
    #     which ray depends on the CPU's GEMM blocking -- measured here: base 1.4e-4 / 5.8e-4, sharp 3.5e-5 / 5.9e-4)
    if family == "white":
        assert st["rgb_max"] > 10 * EPS_RGB and st["depth_max"] > 10 * EPS_DEP and st["rays_over_tolerance"] >= 3, st
    # (4) while the bulk is orders of magnitude tighter
    assert st["rgb_q99"] <= EPS_RGB and st["depth_q99"] <= EPS_DEP
