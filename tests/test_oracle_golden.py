"""The CPU oracle against the golden vectors produced by the real reference (oracle/gen_golden.py).

Everything built from the same torch ops in the same order is required to be BIT-EXACT; this is
what pins the oracle (the reference ships no tests of its own for this path).
"""
import os

import pytest
import torch

from conftest import GOLDEN


def test_state_dict_matches_checkpoint_fixture(oracle, synthetic_sd):
    ck = torch.load(os.path.join(GOLDEN, "synthetic_ckpt.pth"), weights_only=True)
    assert set(ck.keys()) >= {"net"}
    assert list(ck["net"].keys()) == oracle.state_dict_keys() and len(ck["net"]) == 48
    regen = oracle.synthetic_state_dict(seed=0)     # same values up to CPU-dependent rounding of its calibration
    for k, v in ck["net"].items():
        assert torch.equal(v, synthetic_sd[k]), k
        assert torch.allclose(v, regen[k], rtol=1e-4, atol=1e-6), k
    n_params = sum(v.numel() for k, v in ck["net"].items() if k.startswith("model."))
    assert n_params == 595844


def test_positional_encoding(oracle, golden):
    g = golden("pe.npz")
    assert torch.equal(oracle.freq_encode(g["x"], 10), g["pe_xyz"])
    assert torch.equal(oracle.freq_encode(g["dirs"], 4), g["pe_dir"])
    assert g["pe_xyz"].shape == (64, 63) and g["pe_dir"].shape == (64, 27)


@pytest.mark.parametrize("tag,prefix", [("coarse", "model"), ("fine", "model_fine")])
def test_mlp_per_layer(oracle, golden, synthetic_sd, tag, prefix):
    g = golden("mlp_layers.npz")
    with torch.no_grad():
        out, acts = oracle.nerf_mlp(synthetic_sd, prefix, g["emb"], return_activations=True)
    assert torch.equal(out, g[f"{tag}_out"])
    for k, v in acts.items():
        assert torch.equal(v, g[f"{tag}_{k}"]), k


def test_network_forward(oracle, golden, synthetic_sd):
    g = golden("network_forward.npz")
    with torch.no_grad():
        assert torch.equal(oracle.network_forward(synthetic_sd, g["pts"], g["viewdirs"], ""), g["raw_coarse"])
        assert torch.equal(oracle.network_forward(synthetic_sd, g["pts"], g["viewdirs"], "fine"), g["raw_fine"])


def test_weights_and_fine_sampling(oracle, golden):
    g = golden("sampling.npz")
    sigma_c = torch.relu(g["raw_coarse"][..., 3])
    T, w = oracle.transmittance_weights(sigma_c, g["t_coarse"])
    assert torch.equal(T, g["T64"]) and torch.equal(w, g["w64"])
    t_f, parts = oracle.fine_sample(sigma_c, g["t_coarse"], return_parts=True)
    assert torch.equal(parts["cdf"], g["cdf"]) and torch.equal(parts["inds"], g["inds"])
    assert torch.equal(t_f, g["t_fine"])
    assert torch.equal(oracle.points_on_rays(g["rays_o"], g["rays_d"], t_f), g["pts_fine"])
    # every u in the last CDF interval collapses onto bins[61]
    bins61 = 0.5 * (g["t_coarse"][:, 62] + g["t_coarse"][:, 61])
    assert torch.equal(t_f[:, -1], bins61) and int(g["inds"].max()) == 63
    t_sorted, _ = torch.sort(torch.cat([g["t_coarse"], t_f], 1), dim=-1)
    assert torch.equal(t_sorted, g["t_sorted"])
    T, w = oracle.transmittance_weights(torch.relu(g["raw_fine"][..., 3]), g["t_sorted"])
    assert torch.equal(T, g["T192"]) and torch.equal(w, g["w192"])
    assert torch.all(t_f[:, 1:] >= t_f[:, :-1])          # fine depths come out sorted


def test_render_full_and_coarse_only(oracle, golden, synthetic_sd):
    g = golden("render.npz")
    with torch.no_grad():
        rgb, dep = oracle.render(synthetic_sd, g["rays_o"][None], g["rays_d"][None])
        rgb0, dep0 = oracle.render(synthetic_sd, g["rays_o"][None], g["rays_d"][None], n_importance=0)
        prgb, pdep = oracle.render(synthetic_sd, g["pin_rays_o"][None], g["pin_rays_d"][None])
    assert torch.equal(rgb, g["rgb_128"]) and torch.equal(dep, g["depth_128"])
    assert torch.equal(rgb0, g["rgb_0"]) and torch.equal(dep0, g["depth_0"])
    assert torch.equal(prgb, g["pin_rgb"]) and torch.equal(pdep, g["pin_depth"])
    # the fixture is a non-degenerate scene (varied colour / depth), not a constant image
    assert g["pin_rgb"].std() > 0.1 and g["pin_depth"].std() > 0.3


def test_render_batched_layout(oracle, golden, synthetic_sd):
    g = golden("render_batched.npz")
    with torch.no_grad():
        rgb, dep = oracle.render(synthetic_sd, g["rays_o"], g["rays_d"])
    assert rgb.shape == (192, 3) and dep.shape == (192,)
    assert torch.equal(rgb, g["rgb"]) and torch.equal(dep, g["depth"])


def test_pinhole_rays_match_fixture(oracle, golden):
    g = golden("render.npz")
    o, d = oracle.pinhole_rays(800, 800, g["pin_c2w"], pixel_ids=g["pin_ids"])
    assert torch.equal(o, g["pin_rays_o"]) and torch.equal(d, g["pin_rays_d"])
    assert torch.allclose(d.norm(dim=-1), torch.ones(256), atol=1e-6)


def test_autograd_fixture(oracle, golden, synthetic_sd):
    """loss is MSE on the fine RGB only, and gradients reach the coarse model through
    the sample positions (no detach).  The oracle under autograd reproduces loss and all 48 grads."""
    g = golden("autograd.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in synthetic_sd.items()}
    rgb, dep = oracle.render(sd, g["rays_o"][None], g["rays_d"][None])
    loss = torch.nn.functional.mse_loss(rgb, g["gt"])
    loss.backward()
    assert torch.equal(loss.detach(), g["loss"])
    for k, v in sd.items():
        assert v.grad is not None, k
        assert torch.allclose(v.grad, g["grad/" + k], rtol=1e-4, atol=1e-6), k
    coarse_l1 = sum(v.grad.abs().sum() for k, v in sd.items() if k.startswith("model."))
    assert coarse_l1 > 0


def test_evaluator_uint8_wraparound(oracle):
    """psnr_metric subtracts and squares uint8 images in uint8.  pred=0, gt=20/255:
    (0 - 20) wraps to 236, 236**2 = 55696 wraps to 144 -> the printed PSNR is 10 log10(255^2/144),
    not the true 10 log10(255^2/400)."""
    import math
    pred = torch.zeros(10, 3)
    gt = torch.full((10, 3), 20.4 / 255.0)
    mse, psnr_printed = oracle.evaluator_metrics(pred, gt)
    assert abs(mse - (20.4 / 255.0) ** 2) < 1e-7
    assert abs(psnr_printed - 10 * math.log10(255 ** 2 / 144.0)) < 1e-9


def test_ess_ert_masked_path(oracle, golden, synthetic_sd):
    """fast_sampling (off by default): the oracle's validity mask equals the
    reference's bit for bit; the masked render agrees to rounding (the reference compacts the valid
    points before its 512-point MLP chunks, so its GEMM row grouping differs from the oracle's)."""
    g = golden("render_masked.npz")
    sigma_c = torch.relu(g["raw_coarse"][..., 3])
    t_c = oracle.stratified_t().expand(256, 64)
    assert torch.equal(oracle.fine_valid_mask(sigma_c, t_c), g["valid_fine"].bool())
    assert torch.equal(oracle.fine_valid_mask(sigma_c, t_c, weights_threshold=0.02), g["valid_fine_thr002"].bool())
    assert 0.3 < g["valid_fine_thr002"].float().mean() < 0.9 and g["valid_fine"].float().mean() < 0.05
    with torch.no_grad():
        rgb, dep = oracle.render(synthetic_sd, g["rays_o"][None], g["rays_d"][None], fast_sampling=True)
        rgb2, dep2 = oracle.render(synthetic_sd, g["rays_o"][None], g["rays_d"][None], fast_sampling=True,
                                   weights_threshold=0.02)
    assert (rgb - g["rgb"]).abs().max() <= 1e-6 and (dep - g["depth"]).abs().max() <= 1e-5
    assert (rgb2 - g["rgb_thr002"]).abs().max() <= 1e-6 and (dep2 - g["depth_thr002"]).abs().max() <= 1e-5


@pytest.mark.parametrize("family", ["base", "sharp", "white", "trained"])
def test_weight_family_renders(oracle, golden, synthetic_sd, family_sd, family):
    """Round 2: parity scenes beyond the benign band-limited field.  The family state_dicts are exact elementwise
    transforms of the committed checkpoint (bit-reproducible anywhere); the REAL reference rendered them in
    oracle/gen_golden.py::family_fixtures; "trained" is a network trained by this build (tests/golden/trained_ckpt.pth).
    Oracle image and merged sample depths: bit-exact."""
    g = golden(f"render_family_{family}.npz")
    sd = family_sd(family)
    for tag in ("seed", "pin"):
        o, d = g[f"{tag}_rays_o"], g[f"{tag}_rays_d"]
        with torch.no_grad():
            rgb, dep, parts = oracle.render(sd, o[None], d[None], return_parts=True)
        assert torch.equal(parts["raw_coarse"][..., 3], g[f"{tag}_sigma_coarse_raw"])
        assert torch.equal(parts["t_sorted"], g[f"{tag}_t_sorted"])
        assert torch.equal(rgb, g[f"{tag}_rgb"]) and torch.equal(dep, g[f"{tag}_depth"])
    o, d = oracle.seeded_rays(512, 31)
    assert torch.equal(o, g["seed_rays_o"]) and torch.equal(d, g["seed_rays_d"])
    # what makes each family what it says: occupancy / sharpness of the coarse density along the fixture rays
    sig = torch.cat([g["seed_sigma_coarse_raw"], g["pin_sigma_coarse_raw"]])
    occ, std = (sig > 0).float().mean().item(), sig.std().item()
    # ("trained": a network trained by the build on the sharp scene, tools/make_trained_fixture.py: it learned hard surfaces)
    # ("trained": the reference's step never supervises the coarse network photometrically, and after 3000 steps
    #  it is a thin, nearly uniform fog -- sigma 0.41 +- 0.16 on 99.9 % of the samples -- while the fine network carries the scene)
    want = {"base": (0.15, 0.45, 4, 16), "sharp": (0.01, 0.10, 40, 90), "white": (0.15, 0.45, 4, 16), "trained": (0.9, 1.0, 0.05, 1.0)}[family]
    assert g["pin_rgb"].std() > 0.1 and g["pin_depth"].std() > 0.2          # a scene, not a constant image
    assert want[0] <= occ <= want[1] and want[2] <= std <= want[3], (family, occ, std)
    assert torch.equal(oracle.weight_family(synthetic_sd, "base")["model.alpha_linear.bias"],
                       synthetic_sd["model.alpha_linear.bias"])
