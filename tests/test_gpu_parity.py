"""-m gpu: the HIP path, called through the C ABI (ctypes), against the CPU oracle and the golden
vectors the real reference produced.  Tolerances (fp32 path):
    raw MLP outputs     |d| <= 2e-5 * max|ref| per channel  (different fp32 summation order)
    final image         PSNR >= 70 dB, 99 % of rays |d rgb| <= 1e-4 and |d depth| <= 1e-3 (SURVEY 8c's figures), and
                        ATTRIBUTION of everything above that (test_family_parity_attributed): with the reference's
                        own sample depths fed to the HIP fine pass + compositing EVERY ray is inside 1e-4 / 1e-3
The image bound cannot be a flat per-ray maximum because the reference algorithm is discontinuous / ill-conditioned
in its own rounding: an inverse-CDF sample jumps by up to one bin when `denom < 1e-5` flips
(volume_renderer.py:259-260) or a searchsorted index flips next to an empty bin, and inside a nearly empty bin
the division by `denom` ~ 1e-5 amplifies a 1e-7 cdf rounding to ~1e-3 in depth.  tests/test_noise_floor.py measures
what that does to the reference against itself (fp64 MLP probe, 3072 rays): max 1.4e-4 / 5.8e-4 on the band-limited
scene, 3.5e-5 / 5.9e-4 on the sharp-density scene, 1.8e-2 / 4.9e-2 on the white-noise scene (19 rays over).  So: stage tests on identical
inputs are tight, the end-to-end image is bounded in the bulk + by attribution, and the hard per-ray maximum is
pinned per scene family at ~3x the value measured on the GPU (kept in profiles/parity_r03.json, not just printed).
Bit-exact where the arithmetic is order-free (point construction, merge of sorted depths,
ray-permutation / chunk invariance).
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, parity_record
import pack_reference

pytestmark = pytest.mark.gpu

RAW_RTOL = 2e-5
EPS_RGB, EPS_DEP = 1e-4, 1e-3            # SURVEY 8c per-ray image tolerance
# hard per-ray maxima: ~3x the largest value measured on the MI355X for that scene family (profiles/parity_r03.json)
HARD_MAX = {"base": (8e-4, 3e-3), "sharp": (1e-4, 1e-3), "white": (6e-2, 2.5e-1), "trained": (1e-4, 1e-3)}
# rays allowed outside 1e-4 / 1e-3: base / sharp 1 % (measured <= 1 of 512), white-noise field 3 % (measured 7 of 512;
# the reference against its own fp64-MLP self: 19 of 3072, tests/test_noise_floor.py)
# "trained" (a network trained by the build, tests/golden/trained_ckpt.pth): SURVEY 8c's flat per-ray tolerance holds on EVERY ray
# (measured max 8.2e-5 / 4.5e-4 over 4096 + 1024 rays) -- its smooth coarse pdf has no empty bins, so no sample flips
# round 3: with the sampler summing its 62 weights in torch.sum's own order (nerf_kernels.hip torch_sum62) the white-noise family went
# from 53 to 33 rays of 4096 outside the tolerance (the reference against its fp64-MLP self: 19 of 3072 = 25 per 4096) -> 2 %
MAX_OVER_FRAC = {"base": 0.01, "sharp": 0.01, "white": 0.02, "trained": 0.0}


def image_stats(oracle, rgb, dep, ref_rgb, ref_dep):
    rgb, dep = rgb.detach().cpu(), dep.detach().cpu()
    e_rgb = (rgb - ref_rgb).abs().max(-1).values
    e_dep = (dep - ref_dep).abs()
    over = (e_rgb > EPS_RGB) | (e_dep > EPS_DEP)
    st = dict(n_rays=int(e_rgb.numel()), psnr_db=round(oracle.psnr(rgb, ref_rgb), 2),
              rgb_max=e_rgb.max().item(), rgb_q99=torch.quantile(e_rgb, 0.99).item(),
              depth_max=e_dep.max().item(), depth_q99=torch.quantile(e_dep, 0.99).item(),
              rays_over_tolerance=int(over.sum()))
    return st, over


def assert_image_close(oracle, rgb, dep, ref_rgb, ref_dep, max_rgb=None, max_dep=None, family="base", name=None):
    st, _ = image_stats(oracle, rgb, dep, ref_rgb, ref_dep)
    if name is not None:
        parity_record("image_vs_reference", name, st)
    print(name or "image", st)
    max_rgb = HARD_MAX[family][0] if max_rgb is None else max_rgb
    max_dep = HARD_MAX[family][1] if max_dep is None else max_dep
    assert st["psnr_db"] >= 70.0, st
    assert st["rgb_q99"] <= EPS_RGB and st["depth_q99"] <= EPS_DEP, st
    assert st["rgb_max"] <= max_rgb and st["depth_max"] <= max_dep, st
    return st


@pytest.fixture(scope="module")
def amd():
    import nerf_replication_amd as pkg
    pkg._lib.load()            # fail loudly if the HIP extension is not there
    return pkg


@pytest.fixture(scope="module")
def net(amd, synthetic_sd):
    n = amd.Network()
    n.load_state_dict(synthetic_sd, strict=True)
    return n.cuda().eval()


def _chan_err(got, ref):
    ref = ref.double()
    scale = ref.reshape(-1, ref.shape[-1]).abs().max(0).values.clamp_min(1e-6)
    return ((got.double().cpu() - ref).abs().reshape(-1, ref.shape[-1]).max(0).values / scale).max().item()


def test_library_loaded_from_tree(amd):
    assert os.path.dirname(amd._lib.LIB_PATH) == os.path.dirname(amd.__file__)
    assert amd._lib.load().nerf_abi_version() == 2


def test_checkpoint_layout_loads(amd, synthetic_sd):
    ck = torch.load(os.path.join(GOLDEN, "synthetic_ckpt.pth"), weights_only=True)
    n = amd.Network()
    n.load_state_dict(ck["net"], strict=True)       # net_utils.py:374-375 layout
    for k, v in n.state_dict().items():
        assert torch.equal(v, synthetic_sd[k])


@pytest.mark.parametrize("model,prefix", [("", "model"), ("fine", "model_fine")])
def test_pack_matches_layout_reference(net, synthetic_sd, model, prefix):
    got = net.packed(model).view(torch.float32).cpu().numpy()
    want = pack_reference.pack_model(synthetic_sd, prefix)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_repack_after_parameter_update(net):
    before = net.packed("").view(torch.float32).clone()
    saved = net.model.rgb_linear.bias.detach().clone()
    with torch.no_grad():
        net.model.rgb_linear.bias.add_(1.0)
    after = net.packed("").view(torch.float32)
    assert not torch.equal(before, after)
    assert torch.equal(after[-4:-1], saved + 1.0)       # head biases sit at the end of the stream
    with torch.no_grad():
        net.model.rgb_linear.bias.copy_(saved)
    assert torch.equal(net.packed("").view(torch.float32), before)


def test_positional_encoding(net, golden):
    g = golden("pe.npz")
    xyz = net.embed_fn(g["x"].cuda()).cpu()
    dirs = net.embeddirs_fn(g["dirs"].cuda()).cpu()
    assert xyz.shape == (64, 63) and dirs.shape == (64, 27)
    assert torch.equal(xyz[:, :3], g["x"])
    assert (xyz - g["pe_xyz"]).abs().max() <= 5e-7      # sin/cos: <= 2 ulp apart (Sleef vs ocml)
    assert (dirs - g["pe_dir"]).abs().max() <= 5e-7


def test_mlp_forward_golden_points(net, golden):
    g = golden("mlp_layers.npz")
    pts, vd = g["pts"].cuda(), g["viewdirs"].cuda()
    # every point has its own direction here: n=128 "rays" of one sample
    for model, tag in (("", "coarse"), ("fine", "fine")):
        raw = net.forward(pts[:, None, :], vd, None, model=model)
        assert raw.shape == (128, 1, 4)
        assert _chan_err(raw[:, 0], g[f"{tag}_out"]) <= RAW_RTOL


def test_network_forward_golden(net, golden):
    g = golden("network_forward.npz")
    for model, key in (("", "raw_coarse"), ("fine", "raw_fine")):
        raw = net.forward(g["pts"].cuda(), g["viewdirs"].cuda(), None, model=model)
        assert raw.shape == (8, 64, 4)
        assert _chan_err(raw, g[key]) <= RAW_RTOL


def test_network_forward_ragged_and_masked(net, oracle, synthetic_sd):
    gen = torch.Generator().manual_seed(3)
    pts = (torch.rand(5, 7, 3, generator=gen) * 2 - 1) * 3      # 35 points: not a multiple of the 32-point tile
    vd = torch.randn(5, 3, generator=gen)
    vd = vd / vd.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        ref = oracle.network_forward(synthetic_sd, pts, vd, "fine")
    raw = net.forward(pts.cuda(), vd.cuda(), None, model="fine")
    assert _chan_err(raw, ref) <= RAW_RTOL
    mask = torch.rand(5, 7, generator=gen) > 0.4
    rawm = net.forward(pts.cuda(), vd.cuda(), mask.cuda(), model="fine").cpu()
    assert torch.all(rawm[~mask] == 0)
    assert _chan_err(rawm[mask], ref[mask]) <= RAW_RTOL
    none = net.forward(pts.cuda(), vd.cuda(), torch.zeros(5, 7, dtype=torch.bool).cuda(), model="")
    assert torch.all(none == 0)


def test_mlp_rays_mode_points_bit_exact(amd, net, golden):
    """Ray mode builds o + d*t and d/||d|| in-kernel.  The points must be bit-identical to torch's
    (sigma depends on xyz only -> bit-equal); d/||d|| may differ from torch.norm by an ulp (reduction
    order inside norm), which only reaches the colour channels through the 4-octave direction PE."""
    g = golden("sampling.npz")
    lib, L = amd._lib.load(), amd._lib
    o, d, t = g["rays_o"].cuda(), g["rays_d"].cuda(), g["t_sorted"].cuda().contiguous()
    raw_rays = torch.empty(256, 192, 4, device="cuda")
    L.check(lib.nerf_mlp_forward_rays(L.ptr(o), L.ptr(d), L.ptr(t), 192, 256, 192, net.packed("fine").data_ptr(),
                                      L.ptr(raw_rays), 0, L.stream_of(o.device)))
    pts = (g["rays_o"][:, None, :] + g["rays_d"][:, None, :] * g["t_sorted"][:, :, None]).cuda()
    vd = (g["rays_d"] / torch.norm(g["rays_d"], dim=-1, keepdim=True)).cuda()
    raw_pts = net.forward(pts, vd, None, model="fine")
    assert torch.equal(raw_rays[..., 3], raw_pts[..., 3])
    assert (raw_rays[..., :3] - raw_pts[..., :3]).abs().max() <= 1e-5
    assert _chan_err(raw_rays, g["raw_fine"]) <= RAW_RTOL


@pytest.mark.parametrize("precision", ["f32", "f16", "f32x", "f16m32"])
@pytest.mark.parametrize("n_rays,S,stride", [(256, 192, 192), (1000, 64, 0), (3, 64, 0)])     # per-ray depths; the shared coarse table; a ragged tile
def test_density_only_forward_is_the_full_forwards_sigma(amd, synthetic_sd, golden, precision, n_rays, S, stride):
    """nerf_mlp_forward_rays_density (the coarse pass of nerf_render_forward when N_importance > 0: the reference reads only
    outputs[..., 3] of the coarse network, volume_renderer.py:335) must write BIT FOR BIT the sigma nerf_mlp_forward_rays
    writes, in every precision; the colour columns are zero (the network stops after the sigma head)."""
    lib, L = amd._lib.load(), amd._lib
    net = amd.Network(); net.load_state_dict(synthetic_sd); net = net.cuda().eval(); net.precision = precision
    prec = L.PRECISIONS[precision]
    gen = torch.Generator().manual_seed(n_rays)
    d = torch.randn(n_rays, 3, generator=gen).cuda()
    o = (torch.randn(n_rays, 3, generator=gen) * 0.1 + torch.tensor([0.0, 0.0, 4.0])).cuda()
    if stride:
        t = torch.sort(torch.rand(n_rays, S, generator=gen) * 4 + 2, dim=-1).values.cuda().contiguous()
    else:
        t = torch.linspace(2.0, 6.0, S).cuda()
    full = torch.full((n_rays, S, 4), float("nan"), device="cuda")
    dens = torch.full((n_rays, S, 4), float("nan"), device="cuda")
    st = L.stream_of(o.device)
    pk = net.packed("").data_ptr()
    L.check(lib.nerf_mlp_forward_rays(L.ptr(o), L.ptr(d), L.ptr(t), stride, n_rays, S, pk, L.ptr(full), prec, st))
    L.check(lib.nerf_mlp_forward_rays_density(L.ptr(o), L.ptr(d), L.ptr(t), stride, n_rays, S, pk, L.ptr(dens), prec, st))
    torch.cuda.synchronize()
    assert torch.isfinite(full).all() and torch.isfinite(dens).all()
    assert torch.equal(dens[..., 3], full[..., 3])
    assert torch.all(dens[..., :3] == 0)


@pytest.mark.parametrize("precision", ["f32", "f16", "f32x", "f16m32"])
@pytest.mark.parametrize("family", ["base", "sharp", "white", "trained"])
def test_compositing_forward_drops_only_colours_that_are_multiplied_by_zero(amd, golden, family_sd, family, precision):
    """nerf_mlp_forward_rays_for_compositing (the fine pass of nerf_render_forward) vs nerf_mlp_forward_rays on the reference's
    own merged depths of every scene family: sigma bit-identical everywhere; rgb bit-identical wherever it is not zeroed; a
    zeroed colour only at points with sigma <= 0 (weight exactly 0), exactly in the 32-sample tiles without a single sigma > 0; nerf_composite of the two raw buffers bit-identical.  The families cover 0 % ... 86 % dead tiles."""
    lib, L = amd._lib.load(), amd._lib
    g = golden(f"render_family_{family}.npz")
    net = amd.Network(); net.load_state_dict(family_sd(family)); net = net.cuda().eval(); net.precision = precision
    prec = L.PRECISIONS[precision]
    n = 509                                                        # ragged last tile pair
    o, d = g["pin_rays_o"][:n].cuda().contiguous(), g["pin_rays_d"][:n].cuda().contiguous()
    t = g["pin_t_sorted"][:n].cuda().contiguous()
    st = L.stream_of(o.device)
    pk = net.packed("fine").data_ptr()
    full = torch.full((n, 192, 4), float("nan"), device="cuda")
    comp = torch.full((n, 192, 4), float("nan"), device="cuda")
    L.check(lib.nerf_mlp_forward_rays(L.ptr(o), L.ptr(d), L.ptr(t), 192, n, 192, pk, L.ptr(full), prec, st))
    L.check(lib.nerf_mlp_forward_rays_for_compositing(L.ptr(o), L.ptr(d), L.ptr(t), 192, n, 192, pk, L.ptr(comp), prec, st))
    torch.cuda.synchronize()
    assert torch.isfinite(comp).all()
    assert torch.equal(comp[..., 3], full[..., 3])
    dropped = (comp[..., :3] != full[..., :3]).any(-1)              # points whose colour differs at all
    assert torch.all(comp[..., :3][dropped] == 0)
    assert torch.all(full[..., 3][dropped] <= 0)
    dead_tiles = (full[..., 3] <= 0).reshape(n, 6, 32).all(-1)
    # every precision decides per 32 consecutive samples (fp32: a one-wave workgroup tile; fp16 / split-fp16: one wave of the tile)
    live_pts = ~dead_tiles[:, :, None].expand(n, 6, 32).reshape(n, 192)
    assert not dropped[live_pts].any()                              # every tile with a live point ran in full
    # (a dead tile whose full colours happen to be exactly 0 would not show up in `dropped`: only an upper bound here)
    assert dropped.reshape(n, 6, 32).any(-1).sum() <= dead_tiles.sum()
    if family in ("sharp", "trained", "base"):
        assert dropped.any(), "fixture lost its dead tiles"
    outs = []
    for raw in (full, comp):
        rgb, dep = torch.empty(n, 3, device="cuda"), torch.empty(n, device="cuda")
        L.check(lib.nerf_composite(L.ptr(raw), L.ptr(t), 192, n, 192, 1, L.ptr(rgb), L.ptr(dep), None, st))
        outs.append((rgb, dep))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_fine_sampling_stage(amd, golden):
    g = golden("sampling.npz")
    lib, L = amd._lib.load(), amd._lib
    raw_c = g["raw_coarse"].cuda().contiguous()
    t_c = torch.linspace(2.0, 6.0, 64).cuda()
    u = torch.linspace(0.0, 1.0, 128).cuda()
    assert torch.equal(t_c.cpu(), g["t_coarse"][0])
    t_sorted = torch.empty(256, 192, device="cuda")
    t_fine = torch.empty(256, 128, device="cuda")
    L.check(lib.nerf_sample_fine(L.ptr(raw_c), L.ptr(t_c), L.ptr(u), 256, L.ptr(t_sorted), L.ptr(t_fine), None, 0.0, 0.0,
                                 L.stream_of(raw_c.device)))
    tf, ts = t_fine.cpu(), t_sorted.cpu()
    d = (tf - g["t_fine"]).abs()
    # continuous in the inputs; only the sum order of the 62 weights differs from torch (1-ulp pdf changes,
    # amplified where a u lands in an almost-empty CDF bin)
    # (a flip of `denom < 1e-5` or of a searchsorted index next to an empty bin moves one sample by
    # up to a bin width 4/63 -- the reference's own discontinuity, see module docstring)
    parity_record("stage_fine_sampling_identical_inputs", "sampling.npz/256x128", {
        "max_move": d.max().item(), "frac_within_2e-5": (d <= 2e-5).float().mean().item(),
        "samples_moved_gt_1e-4": int((d > 1e-4).sum()), "samples_moved_gt_1e-3": int((d > 1e-3).sum()),
        "n_samples": int(d.numel())})
    # measured on the MI355X: 99.97 % within 2e-5, 6 of 32 768 samples moved by more than 1e-3 (flips)
    assert d.max() <= 4.0 / 63 and (d <= 2e-5).float().mean() >= 0.999 and int((d > 1e-3).sum()) <= 33, \
        (d.max().item(), (d <= 2e-5).float().mean().item(), int((d > 1e-3).sum()))
    assert torch.equal(tf[:, -1], g["t_fine"][:, -1])                 # F7 tail collapse onto bins[61]
    # merged output is exactly the sort of (coarse U own fine depths)
    want, _ = torch.sort(torch.cat([g["t_coarse"], tf], 1), dim=-1)
    assert torch.equal(ts, want)
    assert (ts - g["t_sorted"]).abs().max() <= 4.0 / 63


def test_composite_stage(amd, oracle, golden):
    g = golden("sampling.npz")
    lib, L = amd._lib.load(), amd._lib
    raw, t = g["raw_fine"].cuda().contiguous(), g["t_sorted"].cuda().contiguous()
    rgb, dep, w = torch.empty(256, 3, device="cuda"), torch.empty(256, device="cuda"), torch.empty(256, 192, device="cuda")
    L.check(lib.nerf_composite(L.ptr(raw), L.ptr(t), 192, 256, 192, 1, L.ptr(rgb), L.ptr(dep), L.ptr(w),
                               L.stream_of(raw.device)))
    ref_rgb, ref_dep = oracle.composite(g["raw_fine"], g["t_sorted"], True)
    assert (w.cpu() - g["w192"]).abs().max() <= 2e-6
    assert (rgb.cpu() - ref_rgb).abs().max() <= 5e-6
    assert (dep.cpu() - ref_dep).abs().max() <= 2e-5
    # the staged (coalesced) kernel and the simple fallback (taken for a depth pointer that is not 16-B
    # aligned) run the same arithmetic in the same order: bit-identical
    t_off = torch.empty(256 * 192 + 1, device="cuda")[1:].view(256, 192)
    t_off.copy_(t)
    assert t_off.data_ptr() % 16 == 4
    rgb_b, dep_b, w_b = torch.empty_like(rgb), torch.empty_like(dep), torch.empty_like(w)
    L.check(lib.nerf_composite(L.ptr(raw), t_off.data_ptr(), 192, 256, 192, 1, L.ptr(rgb_b), L.ptr(dep_b), L.ptr(w_b),
                               L.stream_of(raw.device)))
    assert torch.equal(rgb, rgb_b) and torch.equal(dep, dep_b) and torch.equal(w, w_b)
    # ragged ray count (not a multiple of the 64-ray block)
    rgb_c, dep_c = torch.empty(101, 3, device="cuda"), torch.empty(101, device="cuda")
    L.check(lib.nerf_composite(L.ptr(raw), L.ptr(t), 192, 101, 192, 1, L.ptr(rgb_c), L.ptr(dep_c), None,
                               L.stream_of(raw.device)))
    assert torch.equal(rgb_c, rgb[:101]) and torch.equal(dep_c, dep[:101])
    # no white background, coarse table shared by all rays (stride 0)
    t64 = torch.linspace(2.0, 6.0, 64).cuda()
    rawc = g["raw_coarse"].cuda().contiguous()
    rgb0, dep0 = torch.empty(256, 3, device="cuda"), torch.empty(256, device="cuda")
    L.check(lib.nerf_composite(L.ptr(rawc), L.ptr(t64), 0, 256, 64, 0, L.ptr(rgb0), L.ptr(dep0), None,
                               L.stream_of(rawc.device)))
    r0, d0 = oracle.composite(g["raw_coarse"], g["t_coarse"], False)
    assert (rgb0.cpu() - r0).abs().max() <= 5e-6 and (dep0.cpu() - d0).abs().max() <= 2e-5


def _render(amd, net, o, d, n_importance=128):
    r = amd.Renderer(net)
    r.N_importance = n_importance
    with torch.no_grad():
        return r.render({"rays_o": o.cuda(), "rays_d": d.cuda()})


def test_render_golden(amd, net, golden, oracle):
    g = golden("render.npz")
    rgb, dep = _render(amd, net, g["rays_o"][None], g["rays_d"][None])
    assert rgb.shape == (256, 3) and dep.shape == (256,) and rgb.is_cuda
    assert_image_close(oracle, rgb, dep, g["rgb_128"], g["depth_128"], name="render.npz/seeded256/f32")
    rgb0, dep0 = _render(amd, net, g["rays_o"][None], g["rays_d"][None], n_importance=0)
    e0, z0 = (rgb0.cpu() - g["rgb_0"]).abs().max().item(), (dep0.cpu() - g["depth_0"]).abs().max().item()
    parity_record("image_vs_reference", "render.npz/coarse_only/f32", dict(rgb_max=e0, depth_max=z0, n_rays=256))
    assert e0 <= 2e-5 and z0 <= 1e-4          # no resampling: continuous, tight
    prgb, pdep = _render(amd, net, g["pin_rays_o"][None], g["pin_rays_d"][None])
    assert_image_close(oracle, prgb, pdep, g["pin_rgb"], g["pin_depth"], name="render.npz/pinhole256/f32")


def _hip_stages(amd, net, o, d, t_sorted_override=None):
    """The four launches of a frame, one by one through the C ABI, returning every intermediate.  With
    `t_sorted_override` the fine pass + compositing run on THOSE merged depths (the reference's) instead of the
    sampler's: the attribution leg."""
    lib, L = amd._lib.load(), amd._lib
    n = o.shape[0]
    st = L.stream_of(o.device)
    t_c, u = torch.linspace(2.0, 6.0, 64).cuda(), torch.linspace(0.0, 1.0, 128).cuda()
    raw_c = torch.empty(n, 64, 4, device="cuda")
    L.check(lib.nerf_mlp_forward_rays(L.ptr(o), L.ptr(d), L.ptr(t_c), 0, n, 64, net.packed("").data_ptr(), L.ptr(raw_c), 0, st))
    t_sorted = torch.empty(n, 192, device="cuda")
    L.check(lib.nerf_sample_fine(L.ptr(raw_c), L.ptr(t_c), L.ptr(u), n, L.ptr(t_sorted), None, None, 0.0, 0.0, st))
    t_use = t_sorted if t_sorted_override is None else t_sorted_override.cuda().contiguous()
    raw_f = torch.empty(n, 192, 4, device="cuda")
    L.check(lib.nerf_mlp_forward_rays(L.ptr(o), L.ptr(d), L.ptr(t_use), 192, n, 192, net.packed("fine").data_ptr(), L.ptr(raw_f), 0, st))
    rgb, dep = torch.empty(n, 3, device="cuda"), torch.empty(n, device="cuda")
    L.check(lib.nerf_composite(L.ptr(raw_f), L.ptr(t_use), 192, n, 192, 1, L.ptr(rgb), L.ptr(dep), None, st))
    return dict(raw_coarse=raw_c, t_sorted=t_sorted, raw_fine=raw_f, rgb=rgb, depth=dep)


@pytest.mark.parametrize("rays", ["seed", "pin"])
@pytest.mark.parametrize("family", ["base", "sharp", "white", "trained"])
def test_family_parity_attributed(amd, oracle, golden, family_sd, family, rays):
    """End-to-end parity on three scene families against renders of the REAL reference (render_family_*.npz), with
    every deviation above SURVEY 8c's per-ray tolerance attributed (round-1 VERDICT "Weak 2b/2c"):
      1. bulk: 99 % of the rays inside 1e-4 / 1e-3, PSNR bar, hard maximum pinned per family;
      2. attribution: the reference's own merged sample depths through the HIP fine MLP + compositing put EVERY ray
         inside 1e-4 / 1e-3 -- so whatever exceeds it end-to-end comes from sample positions, i.e. from
      3. the sampler chain, which is checked on identical inputs: the HIP coarse densities are within 2e-5 of the
         reference's, and the HIP sampler on the HIP densities equals the oracle's sampler on the SAME densities except
         for a bounded number of moved samples (summation-order flips of the reference's own discontinuities).
    All counts go to profiles/parity_r03.json."""
    g = golden(f"render_family_{family}.npz")
    sd = family_sd(family)
    net = amd.Network()
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    o, d = g[f"{rays}_rays_o"].cuda(), g[f"{rays}_rays_d"].cuda()
    ref_rgb, ref_dep, ref_t = g[f"{rays}_rgb"], g[f"{rays}_depth"], g[f"{rays}_t_sorted"]
    n = o.shape[0]
    with torch.no_grad():
        rgb, dep = amd.Renderer(net).render({"rays_o": o[None], "rays_d": d[None]})
    hip = _hip_stages(amd, net, o, d)
    assert torch.equal(hip["rgb"], rgb) and torch.equal(hip["depth"], dep)      # the staged calls ARE the render path
    st, over = image_stats(oracle, rgb, dep, ref_rgb, ref_dep)

    # (3) sampler chain on identical inputs
    sig_ref = g[f"{rays}_sigma_coarse_raw"]
    sig_hip = hip["raw_coarse"][..., 3].cpu()
    st["coarse_sigma_err_rel_to_range"] = ((sig_hip - sig_ref).abs().max() / sig_ref.abs().max()).item()
    t_c = oracle.stratified_t().expand(n, 64)
    t_f_same = oracle.fine_sample(torch.relu(sig_hip), t_c)                     # oracle sampler on the HIP densities
    t_same, _ = torch.sort(torch.cat([t_c, t_f_same], 1), dim=-1)
    dt_same = (hip["t_sorted"].cpu() - t_same).abs()
    dt_ref = (hip["t_sorted"].cpu() - ref_t).abs()
    for key, dt in (("sampler_same_inputs", dt_same), ("vs_reference_depths", dt_ref)):
        st[key] = {"max_move": dt.max().item(), "samples_moved_gt_1e-5": int((dt > 1e-5).sum()),
                   "samples_moved_gt_1e-4": int((dt > 1e-4).sum()), "samples_moved_gt_1e-3": int((dt > 1e-3).sum()),
                   "rays_with_move_gt_1e-4": int((dt.max(1).values > 1e-4).sum())}
    moved = dt_ref.max(1).values > 1e-5
    st["rays_over_tolerance_without_moved_sample"] = int((over & ~moved).sum())

    # (2) attribution: reference depths -> HIP fine pass + compositing
    att = _hip_stages(amd, net, o, d, t_sorted_override=ref_t)
    a_st, a_over = image_stats(oracle, att["rgb"], att["depth"], ref_rgb, ref_dep)
    st["attributed_on_reference_depths"] = dict(rgb_max=a_st["rgb_max"], depth_max=a_st["depth_max"],
                                                rays_over_tolerance=a_st["rays_over_tolerance"])
    parity_record("family_parity_attributed", f"{family}/{rays}512/f32", st)
    print(family, rays, st)

    assert st["coarse_sigma_err_rel_to_range"] <= RAW_RTOL, st
    # attribution, every ray, hard -- measured <= 7e-6 / 5.2e-5 on all six scenes, i.e. far inside SURVEY 8c's figures
    assert a_st["rgb_max"] <= 2e-5 and a_st["depth_max"] <= 2e-4, st
    assert st["rays_over_tolerance"] <= MAX_OVER_FRAC[family] * n, st
    # the bulk (restored in round 3; round 2 had dropped it after the white-noise family measured 1.23e-4 -- that excess was the
    # sampler's left-to-right weight sum, now 5.2e-5 .. 7.0e-5 on that family)
    assert st["rgb_q99"] <= EPS_RGB and st["depth_q99"] <= EPS_DEP, st
    assert st["psnr_db"] >= {"base": 95.0, "sharp": 110.0, "white": 60.0, "trained": 110.0}[family], st    # measured 107.8 / 115.2 / 67.2 / 123.7
    assert st["rgb_max"] <= HARD_MAX[family][0] and st["depth_max"] <= HARD_MAX[family][1], st
    # the sampler itself (same inputs): moves beyond a bin's rounding amplification are flips; a handful per 65 536
    assert st["sampler_same_inputs"]["samples_moved_gt_1e-3"] <= 0.001 * n * 128, st     # measured <= 22 of 65 536
    if family != "white":       # on the white-noise field a 1e-5 move is already visible (test_noise_floor.py)
        assert st["rays_over_tolerance_without_moved_sample"] == 0, st


def test_render_batched_layout_and_empty(amd, net, golden, oracle):
    g = golden("render_batched.npz")
    rgb, dep = _render(amd, net, g["rays_o"], g["rays_d"])
    assert rgb.shape == (192, 3) and dep.shape == (192,)
    assert_image_close(oracle, rgb, dep, g["rgb"], g["depth"], name="render_batched.npz/f32")
    e_rgb, e_dep = _render(amd, net, torch.zeros(1, 0, 3), torch.zeros(1, 0, 3))
    assert e_rgb.shape == (0, 3) and e_dep.shape == (0,)


def test_render_rejects_cpu_and_bad_args(amd, net):
    r = amd.Renderer(net)
    with pytest.raises(amd._lib.NerfLibraryError):
        r.render({"rays_o": torch.zeros(1, 4, 3), "rays_d": torch.zeros(1, 4, 3)})
    lib = amd._lib.load()
    assert lib.nerf_render_forward(None, None, 4, None, None, None, None, 128, 1, 0, 0, 0.25, None, 0, None, None, None) == -1
    assert b"null" in lib.nerf_last_error()
    assert lib.nerf_render_forward(None, None, 4, None, None, None, None, 64, 1, 0, 0, 0.25, None, 0, None, None, None) == -1
    assert lib.nerf_render_forward(None, None, 0, None, None, None, None, 128, 1, 0, 0, 0.25, None, 0, None, None, None) == 0
    # raw_coarse + t_sorted + raw_fine, + the last-sample ids and count of the fp16 far-plane guard (4 B per ray, 256-aligned, + 256)
    assert lib.nerf_render_workspace_bytes(1000, 128, 0) == 1000 * (1024 + 768 + 3072) + 4096 + 256
    # masked fine pass: (ray, sample) ids are int32 -> more than 2^31 / 192 rays per call is refused, not wrapped
    # (the check precedes every launch; the dummy non-null pointers are never dereferenced)
    big = (2 ** 31) // 192 + 1
    need = lib.nerf_render_workspace_bytes(big, 128, 1)
    assert lib.nerf_render_forward(8, 8, big, 8, 8, 8, 8, 128, 1, 0, 1, 0.25, 8, need, 8, 8, None) == -1
    assert b"int32" in lib.nerf_last_error()
    assert lib.nerf_build_flags() == 0            # product build: no timing switch compiled in


@pytest.mark.parametrize("precision", ["f32", "f16", "f32x"])
def test_ray_blocks_change_nothing(amd, synthetic_sd, oracle, monkeypatch, precision):
    """nerf_render_forward walks the frame in ray blocks (bounded workspace).  With 1024-ray blocks (five blocks, the last one
    ragged) the image is bit-identical to the one-block render: 64+128, coarse only, and the masked (ESS/ERT) fine pass; the fp16
    far-plane guard runs per block."""
    n = 4500
    ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(12))[:n]
    o, d = oracle.pinhole_rays(800, 800, oracle.camera_pose(63.0), pixel_ids=ids)
    net = amd.Network(); net.load_state_dict(synthetic_sd); net = net.cuda().eval(); net.precision = precision
    out = {}
    for tag, env in (("one", "100000000"), ("blocks", "1024")):
        monkeypatch.setenv("NERF_RENDER_BLOCK_RAYS", env)
        ren = amd.Renderer(net)
        res = []
        for n_imp, fast in ((128, False), (0, False), (128, True)):
            ren.N_importance, ren.fast_sampling, ren.weights_threshold = n_imp, fast, 0.02
            with torch.no_grad():
                rgb, dep = ren.render({"rays_o": o[None].cuda(), "rays_d": d[None].cuda()})
            res.append((rgb.clone(), dep.clone()))
        out[tag] = res
    monkeypatch.delenv("NERF_RENDER_BLOCK_RAYS")
    for (ra, da), (rb, db) in zip(out["one"], out["blocks"]):
        assert torch.isfinite(rb).all() and torch.equal(ra, rb) and torch.equal(da, db)
    lib = amd._lib.load()
    monkeypatch.setenv("NERF_RENDER_BLOCK_RAYS", "1024")
    assert lib.nerf_render_workspace_bytes(10 ** 7, 128, 0) == lib.nerf_render_workspace_bytes(1024, 128, 0)      # bounded by the block
    monkeypatch.delenv("NERF_RENDER_BLOCK_RAYS")


def test_full_frame_properties(amd, net, oracle, synthetic_sd):
    """BASELINE config 2 size (800x800 = 640 000 rays, 64+128): size-independent properties, plus the
    oracle on a random 512-ray subset."""
    c2w = oracle.camera_pose(40.0)
    o, d = oracle.pinhole_rays(800, 800, c2w)
    rgb, dep = _render(amd, net, o[None], d[None])
    rgb, dep = rgb.cpu(), dep.cpu()
    assert rgb.shape == (640000, 3) and torch.isfinite(rgb).all() and torch.isfinite(dep).all()
    assert rgb.min() >= -1e-6 and rgb.max() <= 1.0 + 1e-5 and dep.min() >= 0 and dep.max() <= 6.0 + 1e-4
    # rays are independent: a permuted subset, rendered alone, is bit-identical (chunk / tile invariance)
    gen = torch.Generator().manual_seed(0)
    idx = torch.randperm(640000, generator=gen)[:4097]
    rgb_s, dep_s = _render(amd, net, o[idx][None], d[idx][None])
    assert torch.equal(rgb_s.cpu(), rgb[idx]) and torch.equal(dep_s.cpu(), dep[idx])
    # determinism
    rgb2, dep2 = _render(amd, net, o[idx][None], d[idx][None])
    assert torch.equal(rgb2, rgb_s) and torch.equal(dep2, dep_s)
    sub = idx[:512]
    with torch.no_grad():
        ref_rgb, ref_dep = oracle.render(synthetic_sd, o[sub][None], d[sub][None])
    assert_image_close(oracle, rgb[sub], dep[sub], ref_rgb, ref_dep, name="full_frame_800x800/512_of_640000/f32")


# =============================================================================== fp16 activation path
# BASELINE config 5: fp16 activations/weights, fp32 accumulate.  Not the parity path: the bar is
# PSNR vs the reference render (north_star: >= 30 dB); measured ~45-60 dB, asserted >= 40 dB.
# Both MFMA shapes of the fp16 arithmetic run every test of this section: "f16" (the shipped one: v_mfma_f32_16x16x32_f16, round-2
# VERDICT item 3) and "f16m32" (v_mfma_f32_32x32x16_f16).
@pytest.fixture(scope="module", params=["f16m32", "f16"])
def net16(amd, synthetic_sd, request):
    n = amd.Network()
    n.load_state_dict(synthetic_sd, strict=True)
    n = n.cuda().eval()
    n.precision = request.param
    return n


@pytest.mark.parametrize("model,prefix", [("", "model"), ("fine", "model_fine")])
def test_f16_pack_matches_layout_reference(net16, synthetic_sd, model, prefix):
    got = net16.packed(model).cpu()
    const, stream = (pack_reference.pack_model_f16 if net16.precision == "f16m32" else pack_reference.pack_model_f16s)(synthetic_sd, prefix)
    n16 = 16384 + 1184 * 1024
    assert got.numel() == n16 + 16384 + 2368 * 1024           # + the split-fp16 stream of the far-plane guard behind it
    assert np.array_equal(got[:16384].view(torch.float32).numpy(), const)
    assert np.array_equal(got[16384:n16].view(torch.float16).numpy(), stream)
    xconst, xstream = pack_reference.pack_model_f16(synthetic_sd, prefix, split=True)
    assert np.array_equal(got[n16:n16 + 16384].view(torch.float32).numpy(), xconst)
    assert np.array_equal(got[n16 + 16384:].view(torch.float16).numpy(), xstream)


def test_f16_network_forward(net16, golden):
    g = golden("network_forward.npz")
    for model, key in (("", "raw_coarse"), ("fine", "raw_fine")):
        raw = net16.forward(g["pts"].cuda(), g["viewdirs"].cuda(), None, model=model)
        err = _chan_err(raw, g[key])
        print(f"f16 network_forward[{model or 'coarse'}]: max rel-to-range err {err:.3e}")
        assert err <= 2e-2
    g = golden("mlp_layers.npz")            # ragged: 128 one-sample rays, every point its own direction
    raw = net16.forward(g["pts"][:77, None, :].cuda().contiguous(), g["viewdirs"][:77].cuda().contiguous(), None, model="fine")
    assert _chan_err(raw[:, 0], g["fine_out"][:77]) <= 2e-2


def test_f16_many_tiles_match_f32_path(amd, net, net16, golden):
    """More points than one persistent pass (stream wrap-around across tiles, ragged last tile)."""
    g = golden("sampling.npz")
    pts = (g["rays_o"][:, None, :] + g["rays_d"][:, None, :] * g["t_sorted"][:, :, None])
    pts = torch.cat([pts] * 7, 0)[:1531].cuda().contiguous()            # 1531*192 = 293952 points = 1148.25 tiles
    vd = torch.cat([g["rays_d"]] * 7, 0)[:1531].cuda().contiguous()
    a = net.forward(pts, vd, None, model="fine")
    b = net16.forward(pts, vd, None, model="fine")
    err = _chan_err(b, a.cpu())
    print(f"f16 vs f32 on 293952 points: {err:.3e}")
    assert err <= 2e-2
    b2 = net16.forward(pts, vd, None, model="fine")
    assert torch.equal(b, b2)


def test_f16_render_psnr(amd, net16, golden, oracle):
    g = golden("render.npz")
    for o, d, ref, refd in ((g["rays_o"], g["rays_d"], g["rgb_128"], g["depth_128"]),
                            (g["pin_rays_o"], g["pin_rays_d"], g["pin_rgb"], g["pin_depth"])):
        rgb, dep = _render(amd, net16, o[None], d[None])
        psnr = oracle.psnr(rgb.cpu(), ref)
        print(f"f16 render PSNR {psnr:.1f} dB, max|d rgb| {(rgb.cpu() - ref).abs().max():.2e}, "
              f"max|d depth| {(dep.cpu() - refd).abs().max():.2e}")
        assert psnr >= 40.0


# =============================================================================== section 8(f) rows
def test_generate_rays_matches_dataset_formula(amd, oracle, golden):
    g = golden("render.npz")
    c2w = g["pin_c2w"]
    o, d = amd.generate_rays(c2w, 800, 800, oracle.LEGO_CAMERA_ANGLE_X, "cuda", pixel_ids=g["pin_ids"])
    # the fixture rays were produced with the reference formula in float64 (blender.py:102-127)
    assert torch.equal(o.cpu(), g["pin_rays_o"])
    assert (d.cpu() - g["pin_rays_d"]).abs().max() <= 6e-8           # <= 1 ulp: matmul association may differ
    # a contiguous tile (rank shard) of the full frame equals the same rows of the whole frame
    full_o, full_d = oracle.pinhole_rays(800, 800, c2w)
    o2, d2 = amd.generate_rays(c2w, 800, 800, oracle.LEGO_CAMERA_ANGLE_X, "cuda", pixel_begin=800 * 100, n_pixels=800 * 7)
    assert (d2.cpu() - full_d[800 * 100:800 * 107]).abs().max() <= 6e-8 and torch.equal(o2.cpu(), full_o[:5600])
    assert torch.allclose(d2.norm(dim=-1), torch.ones(5600, device="cuda"), atol=1e-6)
    lib = amd._lib.load()
    import ctypes
    bad = (ctypes.c_double * 12)(*([0.0] * 12))
    assert lib.nerf_generate_rays(bad, 800, 800, 1000.0, 639999, 2, None, None, None, None) == -1   # runs off the image


def test_evaluator_metrics(amd, oracle, tmp_path):
    gen = torch.Generator().manual_seed(5)
    gt = torch.rand(4000, 3, generator=gen)
    pred = (gt + 0.05 * torch.randn(4000, 3, generator=gen))          # some values leave [0,1] -> clip path
    ev = amd.Evaluator()
    ev.evaluate((pred.cuda(), None), {"colors": gt[None].cuda()})
    ev.evaluate((gt.cuda(), None), {"colors": gt[None].cuda()})         # identical images -> the 100 dB branch
    mse, psnr_printed = oracle.evaluator_metrics(pred, gt)
    assert abs(ev.mse[0] - mse) <= 1e-9
    assert abs(ev.psnr[0] - psnr_printed) <= 1e-6                      # incl. the uint8 wrap-around
    assert ev.psnr[1] == 100.0 and ev.mse[1] == 0.0
    assert abs(ev.psnr_float[0] - oracle.psnr(pred.clamp(0, 1), gt)) <= 1e-4
    s = ev.summarize()
    assert abs(s["psnr"] - (psnr_printed + 100.0) / 2) <= 1e-6 and s["ssim"] is None
    # whole-image batch: SSIM of ssim_metric (7x7 uniform window on the uint8 images)
    H, W = 37, 53
    img = torch.rand(H, W, 3, generator=gen)
    smooth = torch.nn.functional.avg_pool2d(img.permute(2, 0, 1)[None], 5, 1, 2)[0].permute(1, 2, 0)
    noisy = (smooth + 0.03 * torch.randn(H, W, 3, generator=gen)).clamp(0, 1)
    ev2 = amd.Evaluator(result_dir=str(tmp_path))          # also dumps images/view007_{pred,gt}.png (:50-61)
    ev2.evaluate((noisy.reshape(-1, 3).cuda(), None), {"colors": smooth.reshape(1, -1, 3).cuda(),
                                                      "H": torch.tensor(H), "W": torch.tensor(W), "id": torch.tensor(7)})
    import os
    assert sorted(os.listdir(tmp_path / "images")) == ["view007_gt.png", "view007_pred.png"]
    ref = oracle.evaluator_ssim(noisy, smooth)
    assert 0.3 < ref < 0.999 and abs(ev2.ssim[0] - ref) <= 1e-9
    assert abs(amd.evaluator.image_ssim(smooth.cuda(), smooth.cuda()) - 1.0) <= 1e-12


def test_ess_ert_mask_stage(amd, golden):
    """fast_sampling branch of fine_sample_points: validity of the 128 fine samples, recovered from the
    merged [n,192] mask (coarse samples are always valid), for the default threshold and a useful one."""
    g = golden("render_masked.npz")
    lib, L = amd._lib.load(), amd._lib
    raw_c = g["raw_coarse"].cuda().contiguous()
    t_c, u = torch.linspace(2.0, 6.0, 64).cuda(), torch.linspace(0.0, 1.0, 128).cuda()
    for thr, key in ((0.25, "valid_fine"), (0.02, "valid_fine_thr002")):
        t_sorted = torch.empty(256, 192, device="cuda")
        t_fine = torch.empty(256, 128, device="cuda")
        valid = torch.empty(256, 192, dtype=torch.uint8, device="cuda")
        L.check(lib.nerf_sample_fine(L.ptr(raw_c), L.ptr(t_c), L.ptr(u), 256, L.ptr(t_sorted), L.ptr(t_fine),
                                     valid.data_ptr(), thr, 0.45, L.stream_of(raw_c.device)))
        valid, ts, tf = valid.cpu().bool(), t_sorted.cpu(), t_fine.cpu()
        want = g[key].bool()
        # every ray keeps its 64 coarse samples; the number of surviving fine samples matches the reference
        n_valid_fine = valid.sum(1) - 64
        mism = (n_valid_fine != want.sum(1)).float().mean().item()
        print(f"thr {thr}: reference keeps {want.float().mean():.3f} of the fine samples, rays with a different count: {mism:.4f}")
        assert mism <= 0.02          # a weight sitting exactly on the threshold may flip with the sum order
        # and the surviving depths are the reference's surviving depths
        for r in range(0, 256, 17):
            if n_valid_fine[r] == want[r].sum():
                kept = torch.sort(torch.cat([t_c.cpu(), g["t_fine"][r][want[r]]])).values
                assert (ts[r][valid[r]] - kept).abs().max() <= 4.0 / 63


def test_render_masked_golden(amd, net, net16, golden, oracle):
    g = golden("render_masked.npz")
    for thr, k_rgb, k_dep in ((0.25, "rgb", "depth"), (0.02, "rgb_thr002", "depth_thr002")):
        r = amd.Renderer(net)
        r.fast_sampling, r.weights_threshold = True, thr
        with torch.no_grad():
            rgb, dep = r.render({"rays_o": g["rays_o"][None].cuda(), "rays_d": g["rays_d"][None].cuda()})
        # one more discontinuity than the plain path: a coarse weight sitting on the ESS threshold toggles
        # a fine sample in or out, so single rays may move by a whole sample's contribution
        assert_image_close(oracle, rgb, dep, g[k_rgb], g[k_dep], max_rgb=5e-4, max_dep=2e-3, name=f"render_masked.npz/thr{thr}/f32")       # measured 1.4e-4 / 5.9e-4
    r = amd.Renderer(net)
    r.fast_sampling = True
    with torch.no_grad():
        prgb, pdep = r.render({"rays_o": g["pin_rays_o"][None].cuda(), "rays_d": g["pin_rays_d"][None].cuda()})
    assert_image_close(oracle, prgb, pdep, g["pin_rgb"], g["pin_depth"], max_rgb=1.6e-2, max_dep=7e-2, name="render_masked.npz/pinhole/f32")   # measured 5.2e-3 / 2.4e-2 (one ray toggles a fine sample)
    r16 = amd.Renderer(net16)                          # fp16 path through the same compaction
    r16.fast_sampling, r16.weights_threshold = True, 0.02
    with torch.no_grad():
        rgb16, _ = r16.render({"rays_o": g["rays_o"][None].cuda(), "rays_d": g["rays_d"][None].cuda()})
    assert oracle.psnr(rgb16.cpu(), g["rgb_thr002"]) >= 40.0


def test_config5_size_1600x1600_f16(amd, net16, oracle, synthetic_sd):
    """BASELINE config 5 size on one GPU: 1600x1600 = 2 560 000 rays (491 M fine points, 12.5 GB of
    workspace), fp16 activations.  Size-independent checks + the CPU oracle on a random subset."""
    c2w = oracle.camera_pose(115.0)
    o, d = amd.generate_rays(c2w, 1600, 1600, oracle.LEGO_CAMERA_ANGLE_X, "cuda")
    assert o.shape == (2560000, 3)
    rgb, dep = _render(amd, net16, o[None], d[None])
    assert torch.isfinite(rgb).all() and torch.isfinite(dep).all()
    assert rgb.min() >= -1e-3 and rgb.max() <= 1.0 + 1e-3 and dep.min() >= 0 and dep.max() <= 6.01
    idx = torch.randperm(2560000, generator=torch.Generator().manual_seed(3))[:384]
    with torch.no_grad():
        ref_rgb, _ = oracle.render(synthetic_sd, o[idx].cpu()[None], d[idx].cpu()[None])
    psnr = oracle.psnr(rgb[idx.cuda()].cpu(), ref_rgb)
    print(f"1600x1600 f16: PSNR vs oracle on 384 rays {psnr:.1f} dB")
    assert psnr >= 40.0
    # the last rays of the frame (tail tiles of every kernel) equal the same rays rendered alone
    tail_rgb, tail_dep = _render(amd, net16, o[-1000:][None], d[-1000:][None])
    assert torch.equal(tail_rgb, rgb[-1000:]) and torch.equal(tail_dep, dep[-1000:])


# =============================================================================== "f32x": fp32-accurate on fp16 MFMA
# Operands split into hi + 2^-11 lo fp16 parts, 3 MFMAs per product, fp32 accumulate (nerf_mlp_f32x.hip.inc).
# Held to the SAME tolerances as the exact-fp32 path.
@pytest.fixture(scope="module")
def netx(amd, synthetic_sd):
    n = amd.Network()
    n.load_state_dict(synthetic_sd, strict=True)
    n = n.cuda().eval()
    n.precision = "f32x"
    return n


def test_f32x_pack_matches_layout_reference(netx, synthetic_sd):
    got = netx.packed("fine").cpu()
    const, stream = pack_reference.pack_model_f16(synthetic_sd, "model_fine", split=True)
    assert got.numel() == 16384 + 2368 * 1024
    assert np.array_equal(got[:16384].view(torch.float32).numpy(), const)
    assert np.array_equal(got[16384:].view(torch.float16).numpy(), stream)


def test_f32x_network_forward_fp32_tolerance(netx, net, golden):
    g = golden("network_forward.npz")
    for model, key in (("", "raw_coarse"), ("fine", "raw_fine")):
        raw = netx.forward(g["pts"].cuda(), g["viewdirs"].cuda(), None, model=model)
        err = _chan_err(raw, g[key])
        print(f"f32x network_forward[{model or 'coarse'}]: max rel-to-range err {err:.3e}")
        assert err <= RAW_RTOL
    g = golden("sampling.npz")                 # many tiles, ragged tail, stream wrap-around
    pts = (g["rays_o"][:, None, :] + g["rays_d"][:, None, :] * g["t_sorted"][:, :, None])
    pts = torch.cat([pts] * 5, 0)[:1111].cuda().contiguous()
    vd = torch.cat([g["rays_d"]] * 5, 0)[:1111].cuda().contiguous()
    a = net.forward(pts, vd, None, model="fine")
    b = netx.forward(pts, vd, None, model="fine")
    assert _chan_err(b, a.cpu()) <= RAW_RTOL
    assert torch.equal(b, netx.forward(pts, vd, None, model="fine"))


def test_f32x_render_golden(amd, netx, golden, oracle):
    g = golden("render.npz")
    rgb, dep = _render(amd, netx, g["rays_o"][None], g["rays_d"][None])
    assert_image_close(oracle, rgb, dep, g["rgb_128"], g["depth_128"], name="render.npz/seeded256/f32x")
    prgb, pdep = _render(amd, netx, g["pin_rays_o"][None], g["pin_rays_d"][None])
    assert_image_close(oracle, prgb, pdep, g["pin_rgb"], g["pin_depth"], name="render.npz/pinhole256/f32x")
    rgb0, dep0 = _render(amd, netx, g["rays_o"][None], g["rays_d"][None], n_importance=0)
    assert (rgb0.cpu() - g["rgb_0"]).abs().max() <= 2e-5 and (dep0.cpu() - g["depth_0"]).abs().max() <= 1e-4


def test_f32x_is_as_accurate_as_exact_fp32(net, netx, oracle, synthetic_sd, golden):
    """Against a float64 evaluation of the network (the true value), the split-operand path must be in
    the same error class as the exact-fp32 MFMA path and as PyTorch's own fp32 CPU result."""
    g = golden("network_forward.npz")
    sd64 = {k: v.double() for k, v in synthetic_sd.items()}
    pts, vd = g["pts"], g["viewdirs"]
    flat = pts.reshape(-1, 3)
    emb = torch.cat([oracle.freq_encode(flat.double(), 10),
                     oracle.freq_encode(vd[:, None].expand(8, 64, 3).reshape(-1, 3).double(), 4)], -1)
    with torch.no_grad():
        truth = oracle.nerf_mlp(sd64, "model_fine", emb).reshape(8, 64, 4)
    e32 = _chan_err(net.forward(pts.cuda(), vd.cuda(), None, model="fine"), truth)
    ex = _chan_err(netx.forward(pts.cuda(), vd.cuda(), None, model="fine"), truth)
    ecpu = _chan_err(g["raw_fine"], truth)
    print(f"error vs float64: torch-CPU fp32 {ecpu:.2e}, exact-fp32 MFMA {e32:.2e}, f32x {ex:.2e}")
    assert ex <= 4 * max(e32, ecpu) and ex <= 1e-5


def test_config4_second_scene_and_pose(amd, oracle):
    """BASELINE config 4 shape on one GPU: a second scene (different seeded weights, generated live so the
    oracle and the HIP path see the same tensors) and pose, rendered as two contiguous ray tiles the way the
    8-GPU shard does, against the CPU oracle."""
    from nerf_replication_amd.dist import shard_bounds
    sd = oracle.synthetic_state_dict(seed=1, occupied=0.25)
    net2 = amd.Network()
    net2.load_state_dict(sd, strict=True)
    net2 = net2.cuda().eval()
    ren = amd.Renderer(net2)
    ids = torch.sort(torch.randperm(800 * 800, generator=torch.Generator().manual_seed(44))[:600]).values
    o, d = oracle.pinhole_rays(800, 800, oracle.camera_pose(200.0, -20.0), pixel_ids=ids)
    parts = []
    with torch.no_grad():
        for r in range(2):
            lo, hi, _ = shard_bounds(600, r, 2)
            parts.append(ren.render({"rays_o": o[lo:hi][None].cuda(), "rays_d": d[lo:hi][None].cuda()}))
        whole = ren.render({"rays_o": o[None].cuda(), "rays_d": d[None].cuda()})
        ref_rgb, ref_dep = oracle.render(sd, o[None], d[None])
    rgb = torch.cat([p[0] for p in parts]); dep = torch.cat([p[1] for p in parts])
    assert torch.equal(rgb, whole[0]) and torch.equal(dep, whole[1])       # tiles == whole frame, bit for bit
    assert_image_close(oracle, rgb, dep, ref_rgb, ref_dep, name="config4_second_scene/600/f32")


def test_integration_md_ctypes_stub_runs_verbatim(amd, net):
    """INTEGRATION.md section 3 shows the ctypes binding a maintainer would write against include/nerf_mi355x.h.
    The code block is executed as printed and must reproduce the package's Renderer bit for bit."""
    import os
    from conftest import REPO
    src = open(os.path.join(REPO, "INTEGRATION.md")).read()
    start = src.index("```python\nimport ctypes, torch") + len("```python\n")
    code = src[start:src.index("```\n", start)].replace('"nerf_replication_amd/libnerf_mi355x.so"',
                                                        repr(os.path.join(REPO, "nerf_replication_amd", "libnerf_mi355x.so")))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)

    class Holder:
        pass
    r = Holder()
    r.pk_coarse, r.pk_fine = ns["pack"](net.model), ns["pack"](net.model_fine)
    gen = torch.Generator().manual_seed(3)
    o = torch.tensor([0.0, 0.0, 4.0]).expand(300, 3).contiguous().cuda()
    d = torch.nn.functional.normalize(torch.randn(300, 3, generator=gen) * 0.2 + torch.tensor([0.0, 0.0, -1.0]), dim=-1).cuda()
    rgb, dep = ns["render"](r, {"rays_o": o[None], "rays_d": d[None]})
    net.precision = "f32"
    with torch.no_grad():
        rgb2, dep2 = amd.Renderer(net).render({"rays_o": o[None], "rays_d": d[None]})
    assert torch.equal(rgb, rgb2) and torch.equal(dep, dep2)


def test_bench_two_rank_path_on_one_gpu():
    """bench.py's N>1 branch, rehearsed on the one GPU of this box (round-1 VERDICT "Weak 7c"): two ranks launched exactly
    as the driver does (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both on cuda:0
    (NERF_BENCH_SHARE_GPU=1) with gloo as the collective backend (RCCL refuses two ranks on one device).  Every rank
    generates and renders only its own tile; the all_gather, the max-over-ranks timing, the per-rank compute times and
    the training / config5 blocks with their collectives all run.  The JSON line must be self-consistent."""
    import json
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, NERF_BENCH_SHARE_GPU="1", NERF_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--res", "800"]
    res = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                    # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["dist_backend"] == "gloo" and out["steps"] == 1
    assert out["scaling"] == "strong" and out["config"]["rays_per_step"] == 640000 and out["dtype"] == "f32"
    assert abs(out["value"] - 640000 / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]
    assert len(out["per_rank_compute_ms"]) == 2 and all(0 < t <= out["ms_per_step"] * 1.05 for t in out["per_rank_compute_ms"])
    assert "cpu_baseline" not in out                               # N=1 only
    # (two ranks SHARE one GPU here, so nothing is said about speed; the blocks must exist and be error-free)
    assert "error" not in out["training"] and out["training"]["n_gpus"] == 2
    assert out["training"]["f32"]["rays_per_iter_per_gpu"] == 4096 and out["training"]["f32x"]["ms_per_step"] > 0
    assert "error" not in out["config5"] and out["config5"]["finite"] and out["config5"]["n_gpus"] == 2


def test_rccl_single_rank_collectives():
    """RCCL itself (round-2 VERDICT "Weak 9": every multi-rank rehearsal ran on gloo).  A FRESH child process with WORLD_SIZE=1 and
    backend "nccl" drives render_shard's all_gather_into_tensor and allreduce_gradients' flat in-place all-reduce through
    librccl.so (NERF_DIST_FORCE_COLLECTIVE=1 takes the collective path even for one rank): results bit-identical to the
    no-collective path, librccl in the process maps.  Then bench.py itself with NERF_BENCH_FORCE_DIST=1: `dist_backend` on its
    JSON line is "nccl".  (No N>1 curve exists: a 1-GPU box cannot run two RCCL ranks.)"""
    import json
    import socket
    import subprocess
    import sys
    from conftest import REPO

    def free_port():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            return str(s.getsockname()[1])

    base = dict(os.environ, MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    res = subprocess.run([sys.executable, os.path.join(REPO, "tests", "rccl_child.py")], cwd=REPO,
                         env=dict(base, MASTER_PORT=free_port()), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["world_size"] == 1
    assert out["render_bit_equal"] and out["grads_bit_equal"] and out["grads_copy_path_bit_equal"]
    assert out["librccl_mapped"] and out["libnerf_mapped"]
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--res", "200",
                          "--no-extras", "--cpu-sample", "0"], cwd=REPO, env=dict(base, MASTER_PORT=free_port(), NERF_BENCH_FORCE_DIST="1"),
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["dist_backend"] == "nccl" and line["world_size"] == 1 and line["n_gpus"] == 1 and line["value"] > 0


@pytest.mark.parametrize("family", ["base", "sharp", "white", "trained"])
def test_family_parity_large_sample(amd, oracle, family_sd, family):
    """The attributed-parity criteria of test_family_parity_attributed on a larger sample: 4096 random pixels of an 800x800 frame
    per scene family, against the CPU oracle (bit-exact to the real reference on the family fixtures, test_oracle_golden.py).
    The point is the STATISTICS -- rays outside the SURVEY tolerance, moved samples, attribution maxima -- which go to
    profiles/parity_r03.json; the assertions are the same family bounds."""
    sd = family_sd(family)
    net = amd.Network()
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    n = 4096
    ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(77))[:n]
    o, d = oracle.pinhole_rays(800, 800, oracle.camera_pose(140.0, -25.0), pixel_ids=ids)
    with torch.no_grad():
        ref_rgb, ref_dep, parts = oracle.render(sd, o[None], d[None], return_parts=True)
    oc, dc = o.cuda(), d.cuda()
    hip = _hip_stages(amd, net, oc, dc)
    st, over = image_stats(oracle, hip["rgb"], hip["depth"], ref_rgb, ref_dep)
    dt = (hip["t_sorted"].cpu() - parts["t_sorted"]).abs()
    st["samples_moved_gt_1e-4"] = int((dt > 1e-4).sum())
    st["rays_with_move_gt_1e-4"] = int((dt.max(1).values > 1e-4).sum())
    st["max_sample_move"] = dt.max().item()
    st["rays_over_tolerance_without_moved_sample"] = int((over & ~(dt.max(1).values > 1e-5)).sum())
    att = _hip_stages(amd, net, oc, dc, t_sorted_override=parts["t_sorted"])
    a_st, _ = image_stats(oracle, att["rgb"], att["depth"], ref_rgb, ref_dep)
    st["attributed_on_reference_depths"] = dict(rgb_max=a_st["rgb_max"], depth_max=a_st["depth_max"],
                                                rays_over_tolerance=a_st["rays_over_tolerance"])
    parity_record("family_parity_attributed", f"{family}/frame4096/f32", st)
    print(family, st)
    # attribution, every one of the 4096 rays: measured <= 1.7e-5 / 9.9e-5 (sharp), i.e. 6x / 10x inside SURVEY 8c's figures
    assert a_st["rgb_max"] <= 5e-5 and a_st["depth_max"] <= 5e-4, st
    assert st["rays_over_tolerance"] <= MAX_OVER_FRAC[family] * n, st                  # measured 4 / 2 / 33 / 0 of 4096 (round 2: 53 on white)
    assert st["rgb_q99"] <= EPS_RGB and st["depth_q99"] <= EPS_DEP, st
    assert st["psnr_db"] >= {"base": 95.0, "sharp": 95.0, "white": 70.0, "trained": 105.0}[family], st    # measured 102.4 / 110.5 / 80.2 / 117.4
    if family != "white":
        assert st["rays_over_tolerance_without_moved_sample"] == 0, st
    else:                       # (white: 4 of 4096, round 2: 9)
        assert st["rays_over_tolerance_without_moved_sample"] <= 0.002 * n, st


def test_render_is_hipgraph_capturable(amd, net, oracle):
    """The C ABI only enqueues on the caller's stream and never synchronises, so a whole
    Renderer.render call -- four kernel launches -- can be captured in a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed
    on new ray data: bit-identical to the eager call.  This is the launch-bound regime (BASELINE configs[0]-sized batches:
    1024 rays), where a replay removes the per-launch and Python overhead."""
    import time
    ren = amd.Renderer(net)
    ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(5))[:2048]
    o_all, d_all = oracle.pinhole_rays(800, 800, oracle.camera_pose(300.0), pixel_ids=ids)
    o_all, d_all = o_all.cuda(), d_all.cuda()
    static_o, static_d = o_all[:1024].clone(), d_all[:1024].clone()
    with torch.no_grad():
        ren.render({"rays_o": static_o[None], "rays_d": static_d[None]})          # warm-up: packs the weights, builds the tables
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ren.render({"rays_o": static_o[None], "rays_d": static_d[None]})      # (allocator warm-up on the capture stream)
            with torch.cuda.graph(graph, stream=side):
                g_rgb, g_dep = ren.render({"rays_o": static_o[None], "rays_d": static_d[None]})
        torch.cuda.current_stream().wait_stream(side)
        for lo in (0, 1024):                                                       # replay on two different ray sets
            static_o.copy_(o_all[lo:lo + 1024]); static_d.copy_(d_all[lo:lo + 1024])
            graph.replay()
            torch.cuda.synchronize()
            e_rgb, e_dep = ren.render({"rays_o": o_all[lo:lo + 1024][None], "rays_d": d_all[lo:lo + 1024][None]})
            assert torch.equal(g_rgb, e_rgb) and torch.equal(g_dep, e_dep)
        n = 50
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
        torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            ren.render({"rays_o": static_o[None], "rays_d": static_d[None]})
        torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / n
    print(f"1024-ray 64+128 render: eager {t_eager * 1e3:.3f} ms, hipGraph replay {t_graph * 1e3:.3f} ms")
    parity_record("hipgraph", "render_1024_rays", {"eager_ms": t_eager * 1e3, "replay_ms": t_graph * 1e3})
    assert t_graph <= 1.2 * t_eager


@pytest.mark.parametrize("family", ["base", "sharp", "white", "trained"])
def test_f16_and_f32x_psnr_on_every_family(amd, oracle, golden, family_sd, family):
    """BASELINE.json's bar for the fp16-activation path is PSNR >= 30 dB against the reference render; the f32x path is held to
    the fp32 bulk criterion.  Measured on the reference-rendered fixtures of every scene family and recorded."""
    g = golden(f"render_family_{family}.npz")
    sd = family_sd(family)
    out = {}
    for prec in ("f16", "f32x", "f16m32"):
        net = amd.Network()
        net.load_state_dict(sd, strict=True)
        net = net.cuda().eval()
        net.precision = prec
        worst = 1e9
        for rays in ("seed", "pin"):
            rgb, dep = _render(amd, net, g[f"{rays}_rays_o"][None], g[f"{rays}_rays_d"][None])
            st, _ = image_stats(oracle, rgb, dep, g[f"{rays}_rgb"], g[f"{rays}_depth"])
            out[f"{prec}/{rays}"] = st
            worst = min(worst, st["psnr_db"])
        out[f"{prec}/worst_psnr_db"] = worst
    parity_record("other_precisions_vs_reference", family, out)
    print(family, {k: (v if not isinstance(v, dict) else (v["psnr_db"], v["rays_over_tolerance"])) for k, v in out.items()})
    # BASELINE's bar is 30 dB; with the far-plane guard (nerf_render_forward re-evaluates the last sample of every ray with the
    # split-fp16 stream) measured 54 / 56 / 48 / 82 dB (round 2, without it: 34.5 / 34.0 / 47.9 / 78.2)
    floor16 = {"base": 45.0, "sharp": 45.0, "white": 40.0, "trained": 70.0}[family]
    assert out["f16/worst_psnr_db"] >= floor16 and out["f16m32/worst_psnr_db"] >= floor16, out
    for prec in ("f16", "f16m32"):
        for rays in ("seed", "pin"):
            assert out[f"{prec}/{rays}"]["depth_max"] <= 1.0, out       # no background ray becomes a far-plane hit (round 2: 6.0)
    assert out["f32x/worst_psnr_db"] >= {"base": 95.0, "sharp": 95.0, "white": 55.0, "trained": 105.0}[family], out
    for rays in ("seed", "pin"):
        assert out[f"f32x/{rays}"]["rays_over_tolerance"] <= max(1, MAX_OVER_FRAC[family] * 512), out
