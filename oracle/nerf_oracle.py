"""CPU oracle for the NeRF volume-rendering hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU (torch fp32) restatement of the arithmetic of the
reference renderer.  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
The product path (``nerf_replication_amd``) never imports anything from ``oracle/``.

Parity status: **pinned**.  ``oracle/gen_golden.py`` imports the real reference
(``/root/reference``) in the build container, runs it on seeded inputs and stores
inputs + every intermediate in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this restatement against those vectors (bit-exact for everything that uses the
same torch ops in the same order).  The reference ships no tests/golden vectors of
its own for this path, so the reference-generated fixtures are
the pin.

Reference lines each function follows (paths relative to /root/reference):

  freq_encode            src/models/encoding/freq.py:7-32, src/models/encoding/__init__.py:8-15
  nerf_mlp               src/models/nerf/network.py:49-74  (layer shapes :22-47)
  network_forward        src/models/nerf/network.py:199-258 (batchify :163-171, chunk 512 :131)
  stratified_t / points  src/models/nerf/renderer/volume_renderer.py:27-65
  transmittance_weights  src/models/nerf/renderer/volume_renderer.py:67-96
  fine_sample            src/models/nerf/renderer/volume_renderer.py:98-153, :247-272
  render                 src/models/nerf/renderer/volume_renderer.py:290-432

Quirks reproduced on purpose: ReLU (not softplus) density, always
deterministic sampling, near/far 2/6, MLP evaluated 512 points at a time, fine pass
evaluates all 192 sorted samples with the fine model, inverse-CDF index clamp to
N_samples-3 (tail collapse), 1e10 last interval, clamp(1-alpha, 1e-10, 1), +1e-5 on
the inner 62 weights, denom<1e-5 -> 1, white background.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

# Renderer constants the reference effectively hard-codes.
N_SAMPLES = 64
N_IMPORTANCE = 128
T_NEAR = 2.0
T_FAR = 6.0
MLP_CHUNK = 512          # network.py:131 / lego.yaml:16
RAYS_BLOCK = 160000      # volume_renderer.py:20
SAMPLE_BLOCK = 64        # volume_renderer.py:19
XYZ_FREQS = 10
DIR_FREQS = 4
HIDDEN = 256

# state_dict key order of one NeRF sub-model (network.py:22-47)
SUBMODEL_KEYS = tuple(
    [f"pts_linears.{i}.{p}" for i in range(8) for p in ("weight", "bias")]
    + [f"{n}.{p}" for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear")
       for p in ("weight", "bias")]
)

SUBMODEL_SHAPES = {
    "pts_linears.0.weight": (256, 63), "pts_linears.0.bias": (256,),
    **{f"pts_linears.{i}.weight": (256, 256) for i in (1, 2, 3, 4, 6, 7)},
    **{f"pts_linears.{i}.bias": (256,) for i in range(1, 8)},
    "pts_linears.5.weight": (256, 319),
    "views_linears.0.weight": (128, 283), "views_linears.0.bias": (128,),
    "feature_linear.weight": (256, 256), "feature_linear.bias": (256,),
    "alpha_linear.weight": (1, 256), "alpha_linear.bias": (1,),
    "rgb_linear.weight": (3, 128), "rgb_linear.bias": (3,),
}


def state_dict_keys():
    """The 48 tensor names of the reference Network.state_dict()."""
    return [f"{m}.{k}" for m in ("model", "model_fine") for k in SUBMODEL_KEYS]


def synthetic_state_dict(seed: int = 0, occupied: float = 0.3, sigma_std: float = 8.0,
                         rgb_std: float = 1.5, fine_jitter: float = 0.02, octave_decay: float = 1.0):
    """Seeded stand-in for latest.pth (unavailable offline): a random but scene-like field.

    nn.Linear-style uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) init for the coarse model; the
    fine model is the coarse one with `fine_jitter` relative noise (as in a trained NeRF both
    describe the same scene).  Input columns of positional-encoding octave k are damped by
    2^(-octave_decay*k) (the spectral bias of a trained NeRF: a band-limited, scene-like field
    instead of white noise).  The density and colour heads are then rescaled against a seeded
    probe set so that about `occupied` of space has sigma > 0 (std `sigma_std`) and the
    pre-sigmoid colours have std `rgb_std` -- rays get varied opacity, depth and colour.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k in SUBMODEL_KEYS:
        shape = SUBMODEL_SHAPES[k]
        wshape = SUBMODEL_SHAPES[k.replace("bias", "weight")]
        bound = 1.0 / math.sqrt(wshape[1])
        sd[f"model.{k}"] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    for k in range(XYZ_FREQS):                      # xyz PE columns feed pts_linears.0 and the skip into .5
        sd["model.pts_linears.0.weight"][:, 3 + 6 * k:9 + 6 * k] *= 2.0 ** (-octave_decay * k)
        sd["model.pts_linears.5.weight"][:, 3 + 6 * k:9 + 6 * k] *= 2.0 ** (-octave_decay * k)
    # calibrate heads on probe points inside the near/far shell around the origin
    probe = (torch.rand(4096, 3, generator=g) * 2 - 1) * 2.0
    pdir = torch.randn(4096, 3, generator=g)
    pdir = pdir / pdir.norm(dim=-1, keepdim=True)
    emb = torch.cat([freq_encode(probe, XYZ_FREQS), freq_encode(pdir, DIR_FREQS)], -1)
    with torch.no_grad():
        raw = nerf_mlp(sd, "model", emb)
    s = raw[:, 3]
    gain = sigma_std / s.std().item()
    sd["model.alpha_linear.weight"] = sd["model.alpha_linear.weight"] * gain
    q = torch.quantile((s - sd["model.alpha_linear.bias"]) * gain, 1.0 - occupied).item()
    sd["model.alpha_linear.bias"] = torch.full((1,), -q)
    cgain = rgb_std / raw[:, :3].std().item()
    sd["model.rgb_linear.weight"] = sd["model.rgb_linear.weight"] * cgain
    sd["model.rgb_linear.bias"] = sd["model.rgb_linear.bias"] * cgain
    for k in SUBMODEL_KEYS:
        w = sd[f"model.{k}"]
        sd[f"model_fine.{k}"] = w * (1.0 + fine_jitter * (torch.rand(w.shape, generator=g) * 2 - 1))
    return {k: sd[k].contiguous() for k in state_dict_keys()}


# Weight families for parity scenes.  Each is an EXACT elementwise transform of a base state_dict (the committed
# tests/golden/synthetic_ckpt.pth), so every machine derives bit-identical tensors from the one checkpoint file
# (synthetic_state_dict itself calibrates its heads with CPU GEMMs and is not bit-reproducible across CPU models).
#   "base"   the band-limited field as stored: sigma ~ N(-4.5, 8), ~30 % of space occupied
#   "sharp"  trained-NeRF-like density: sigma_raw' = 7.5*sigma_raw - 30  ->  std 60 (sigma up to several hundred), ~5 %
#            of the samples along the fixture rays occupied, ~40 % of the rays end on a hard surface: most CDF bins
#            empty (pdf = 1e-5/sum, i.e. the `denom < 1e-5` regime of volume_renderer.py:259-260 whenever a ray's
#            inner weights sum to ~1)
#   "white"  octave_decay = 0: the 2^-k damping of the positional-encoding input columns is undone exactly (x 2^k),
#            heads re-scaled by fixed constants to sigma ~ N(-4.2, 8), ~30 % occupied: a white-noise field, every
#            sample position matters (round 1's first smoke scene)
WEIGHT_FAMILIES = {
    "base": dict(alpha_gain=1.0, alpha_shift=0.0, undo_octave_decay=False),
    "sharp": dict(alpha_gain=7.5, alpha_shift=30.0, undo_octave_decay=False),
    "white": dict(alpha_gain=0.345, alpha_shift=3.0, undo_octave_decay=True),
}


def weight_family(base_sd: Dict[str, torch.Tensor], name: str) -> Dict[str, torch.Tensor]:
    spec = WEIGHT_FAMILIES[name]
    out = {k: v.clone() for k, v in base_sd.items()}
    for m in ("model", "model_fine"):
        out[f"{m}.alpha_linear.weight"] = out[f"{m}.alpha_linear.weight"] * spec["alpha_gain"]
        out[f"{m}.alpha_linear.bias"] = out[f"{m}.alpha_linear.bias"] * spec["alpha_gain"] - spec["alpha_shift"]
        if spec["undo_octave_decay"]:
            for k in range(XYZ_FREQS):
                out[f"{m}.pts_linears.0.weight"][:, 3 + 6 * k:9 + 6 * k] *= 2.0 ** k
                out[f"{m}.pts_linears.5.weight"][:, 3 + 6 * k:9 + 6 * k] *= 2.0 ** k
    return {k: out[k].contiguous() for k in state_dict_keys()}


def seeded_rays(n: int, seed: int):
    """Parity ray set: origin (0,0,4), directions normalize(randn*0.2 + (0,0,-1))."""
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3).contiguous()
    d = torch.randn(n, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, -1.0])
    d = d / d.norm(dim=-1, keepdim=True)
    return o, d.contiguous()


# ----------------------------------------------------------------------------- encoding
def freq_encode(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """[P,3] -> [P, 3+6*n_freqs]: x, then per octave sin(2^k x) (3 wide), cos(2^k x) (3 wide)."""
    bands = 2.0 ** torch.linspace(0.0, n_freqs - 1, steps=n_freqs)
    cols = [x]
    for f in bands:
        xf = x * f
        cols.append(torch.sin(xf))
        cols.append(torch.cos(xf))
    return torch.cat(cols, -1)


# ----------------------------------------------------------------------------- MLP
def nerf_mlp(sd: Dict[str, torch.Tensor], prefix: str, emb: torch.Tensor,
             return_activations: bool = False):
    """One NeRF MLP on an embedded chunk [p,90] -> [p,4] = (r,g,b,sigma) pre-activation."""
    W = lambda n: sd[f"{prefix}.{n}.weight"]
    B = lambda n: sd[f"{prefix}.{n}.bias"]
    lin = torch.nn.functional.linear
    pts, dirs = emb[:, :63], emb[:, 63:]
    acts = {}
    h = pts
    for i in range(8):
        h = torch.relu(lin(h, W(f"pts_linears.{i}"), B(f"pts_linears.{i}")))
        acts[f"h{i}"] = h
        if i == 4:
            h = torch.cat([pts, h], -1)
    sigma = lin(h, W("alpha_linear"), B("alpha_linear"))
    feat = lin(h, W("feature_linear"), B("feature_linear"))
    acts["feature"] = feat
    hv = torch.relu(lin(torch.cat([feat, dirs], -1), W("views_linears.0"), B("views_linears.0")))
    acts["views"] = hv
    rgb = lin(hv, W("rgb_linear"), B("rgb_linear"))
    out = torch.cat([rgb, sigma], -1)
    return (out, acts) if return_activations else out


def network_forward(sd, pts: torch.Tensor, viewdirs: torch.Tensor, model: str = "",
                    chunk: int = MLP_CHUNK, mlp_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """pts [n,s,3], viewdirs [n,3] -> raw [n,s,4]; MLP run `chunk` points at a time.
    `mlp_dtype=torch.float64` is NOT the reference's arithmetic: it evaluates encoding + MLP in double on
    the same fp32 inputs/weights and rounds raw back to fp32 -- the probe the noise-floor test uses to
    measure how far the reference's own fp32 rounding moves its image (tests/test_noise_floor.py)."""
    prefix = "model_fine" if model == "fine" else "model"
    n, s, _ = pts.shape
    flat = pts.reshape(-1, 3)
    dflat = viewdirs[:, None].expand(n, s, 3).reshape(-1, 3)
    if mlp_dtype != torch.float32:
        sd = {k: v.to(mlp_dtype) for k, v in sd.items() if k.startswith(prefix + ".")}
        flat, dflat = flat.to(mlp_dtype), dflat.to(mlp_dtype)
    emb = torch.cat([freq_encode(flat, XYZ_FREQS), freq_encode(dflat, DIR_FREQS)], -1)
    emb = emb.to(mlp_dtype)
    outs = [nerf_mlp(sd, prefix, emb[i:i + chunk]) for i in range(0, emb.shape[0], chunk)]
    return torch.cat(outs, 0).reshape(n, s, 4).to(torch.float32)


# ----------------------------------------------------------------------------- sampling
def stratified_t(n_samples: int = N_SAMPLES) -> torch.Tensor:
    return torch.linspace(T_NEAR, T_FAR, n_samples)


def fine_u(n_importance: int = N_IMPORTANCE) -> torch.Tensor:
    return torch.linspace(0.0, 1.0, steps=n_importance)


def points_on_rays(rays_o, rays_d, t):
    """o + d*t, separately rounded multiply and add: [N,3],[N,3],[N,S] -> [N,S,3]."""
    return rays_o[:, None, :] + rays_d[:, None, :] * t[:, :, None]


def transmittance_weights(sigma: torch.Tensor, t: torch.Tensor):
    """sigma,t [N,S] -> (T, w).  delta_last = 1e10; T = exclusive cumprod of clamp(1-a,1e-10,1)."""
    delta = t[:, 1:] - t[:, :-1]
    delta = torch.cat([delta, 1e10 * torch.ones_like(delta[:, :1])], -1)
    alpha = 1.0 - torch.exp(-sigma * delta)
    keep = torch.clamp(1.0 - alpha, min=1e-10, max=1.0)
    T = torch.cumprod(torch.cat([torch.ones(alpha.shape[0], 1), keep], -1), -1)[:, :-1]
    return T, T * alpha


def fine_sample(sigma_c: torch.Tensor, t_c: torch.Tensor, n_importance: int = N_IMPORTANCE,
                eps: float = 1e-5, return_parts: bool = False):
    """Deterministic inverse-CDF sampling. sigma_c (already ReLU'd), t_c [N,64] -> t_f [N,128]."""
    n_s = t_c.shape[1]
    _, w = transmittance_weights(sigma_c, t_c)
    w = w[:, 1:-1] + eps
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)            # [N, n_s-1]
    u = fine_u(n_importance).expand(cdf.shape[0], n_importance).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, 0, n_s - 3)
    above = torch.clamp(inds, 0, n_s - 3)
    bins = 0.5 * (t_c[:, 1:] + t_c[:, :-1])                             # [N, n_s-1]
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    frac = (u - cdf_b) / denom
    t_f = bin_b + frac * (bin_a - bin_b)
    if return_parts:
        return t_f, dict(cdf=cdf, inds=inds, below=below, above=above)
    return t_f


def fine_valid_mask(sigma_c: torch.Tensor, t_c: torch.Tensor, n_importance: int = N_IMPORTANCE,
                    weights_threshold: float = 0.25, ert_threshold: float = 0.45) -> torch.Tensor:
    """ESS/ERT validity of the fine samples, the `fast_sampling` branch of fine_sample_points
    (volume_renderer.py:116-123, :132-133, :158-193): a fine sample is dropped when its CDF bin
    neighbours carry coarse weight < weights_threshold (both neighbours for "object" rays with
    max sigma > 0.5, either one otherwise), when the coarse transmittance at its lower bin is already
    < ert_threshold, or when the whole ray is empty (sum sigma < 1e-3).  -> bool [N, n_importance]."""
    n_s = t_c.shape[1]
    empty_ray = sigma_c.sum(dim=-1) < 1e-3
    object_ray = sigma_c.max(dim=-1).values > 0.5
    T, w = transmittance_weights(sigma_c, t_c)
    w, T = w[:, 1:-1], T[:, 1:-1]
    empty_bins = w < weights_threshold
    _, parts = fine_sample(sigma_c, t_c, n_importance, return_parts=True)
    below, above = parts["below"], parts["above"]
    ert_base = torch.cat([torch.zeros_like(T[:, :1], dtype=torch.bool), T < ert_threshold], dim=1)
    ert_empty = torch.cummax(ert_base, dim=1)[0][:, 1:]
    ert_non_valid = torch.gather(ert_empty, 1, below)
    below_empty, above_empty = torch.gather(empty_bins, 1, below), torch.gather(empty_bins, 1, above)
    ess_non_valid = torch.where(object_ray[:, None], below_empty & above_empty, below_empty | above_empty)
    valid = ~(ess_non_valid | ert_non_valid)
    valid[empty_ray] = False
    return valid


def composite(raw: torch.Tensor, t: torch.Tensor, white_bkgd: bool = True):
    """raw [N,S,4] pre-activation, t [N,S] -> rgb [N,3], depth [N]."""
    rgb = torch.sigmoid(raw[..., :3])
    sigma = torch.relu(raw[..., 3])
    _, w = transmittance_weights(sigma, t)
    rgb_out = torch.sum(w[..., None] * rgb, dim=1)
    depth_out = torch.sum(w * t, dim=1)
    if white_bkgd:
        rgb_out = rgb_out + (1.0 - w.sum(dim=-1, keepdim=True))
    return rgb_out, depth_out


# ----------------------------------------------------------------------------- render
def render(sd, rays_o: torch.Tensor, rays_d: torch.Tensor, n_importance: int = N_IMPORTANCE,
           white_bkgd: bool = True, return_parts: bool = False, chunk: int = MLP_CHUNK,
           fast_sampling: bool = False, weights_threshold: float = 0.25,
           mlp_dtype: torch.dtype = torch.float32):
    """rays_o, rays_d [B,N,3] -> (rgb [B*N,3], depth [B*N]) exactly as Renderer.render does,
    including its 160000-ray x 64-sample blocking of the MLP calls."""
    rays_o = rays_o.reshape(-1, 3)
    rays_d = rays_d.reshape(-1, 3)
    n = rays_o.shape[0]
    t_c = stratified_t().unsqueeze(0).expand(n, N_SAMPLES).clone()
    pts_c = points_on_rays(rays_o, rays_d, t_c)
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)

    raw = torch.cat([network_forward(sd, pts_c[i:i + RAYS_BLOCK], viewdirs[i:i + RAYS_BLOCK], "", chunk, mlp_dtype)
                     for i in range(0, n, RAYS_BLOCK)], 0)
    depth = t_c
    parts = {"raw_coarse": raw, "t_coarse": t_c, "viewdirs": viewdirs}
    if n_importance > 0:
        sigma_c = torch.relu(raw[..., 3])
        t_f = fine_sample(sigma_c, t_c, n_importance)
        pts_f = points_on_rays(rays_o, rays_d, t_f)
        depth, order = torch.sort(torch.cat([t_c, t_f], 1), dim=-1)
        pts = torch.gather(torch.cat([pts_c, pts_f], 1), 1, order[..., None].expand(-1, -1, 3))
        rows = []
        for i in range(0, n, RAYS_BLOCK):
            cols = [network_forward(sd, pts[i:i + RAYS_BLOCK, j:j + SAMPLE_BLOCK],
                                    viewdirs[i:i + RAYS_BLOCK], "fine", chunk, mlp_dtype)
                    for j in range(0, pts.shape[1], SAMPLE_BLOCK)]
            rows.append(torch.cat(cols, 1))
        raw = torch.cat(rows, 0)
        if fast_sampling:      # volume_renderer.py:359-369 + network.py:238-253: masked-out samples get raw = 0
            valid = torch.cat([torch.ones(n, N_SAMPLES, dtype=torch.bool), fine_valid_mask(sigma_c, t_c, n_importance, weights_threshold)], 1)
            valid = torch.gather(valid, 1, order)
            raw = raw * valid[..., None]
            parts.update(valid_sorted=valid)
        parts.update(t_fine=t_f, t_sorted=depth, raw_fine=raw)
    rgb, dep = composite(raw, depth, white_bkgd)
    return (rgb, dep, parts) if return_parts else (rgb, dep)


# ----------------------------------------------------------------------------- inputs
LEGO_CAMERA_ANGLE_X = 0.6911112070083618   # synthetic parameter (not in the reference repo)
CAMERA_RADIUS = 4.031128874


def camera_pose(theta_deg: float, phi_deg: float = -30.0, radius: float = CAMERA_RADIUS):
    """Blender-style camera-to-world matrix on the upper hemisphere looking at the origin."""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    trans = torch.eye(4); trans[2, 3] = radius
    rot_phi = torch.tensor([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0],
                            [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1]], dtype=torch.float32)
    rot_th = torch.tensor([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0],
                           [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1]], dtype=torch.float32)
    flip = torch.tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=torch.float32)
    return flip @ rot_th @ rot_phi @ trans


def pinhole_rays(H: int, W: int, c2w: torch.Tensor, camera_angle_x: float = LEGO_CAMERA_ANGLE_X,
                 pixel_ids: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Ray generation formula of src/datasets/nerf/blender.py:102-127 (float64 math like numpy,
    result cast to float32).  Returns rays_o, rays_d [n,3] for all pixels row-major or `pixel_ids`."""
    f = W / (2.0 * math.tan(camera_angle_x / 2.0))
    cx, cy = W / 2.0, H / 2.0
    ids = torch.arange(H * W) if pixel_ids is None else pixel_ids
    u = (ids % W).to(torch.float64)
    v = (ids // W).to(torch.float64)
    dirs = torch.stack([(u - cx) / f, -(v - cy) / f, -torch.ones_like(u)], -1)
    R = c2w[:3, :3].to(torch.float64)
    d = (R @ dirs.T).T
    d = d / torch.linalg.norm(d, dim=-1, keepdim=True)
    o = c2w[:3, 3].to(torch.float64).expand_as(d)
    return o.to(torch.float32).contiguous(), d.to(torch.float32).contiguous()


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """Float PSNR, data_range 1 (not the evaluator's uint8-wrapping variant)."""
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return 100.0 if mse < 1e-20 else 10.0 * math.log10(1.0 / mse)


def evaluator_metrics(pred: torch.Tensor, gt: torch.Tensor):
    """src/evaluators/nerf.py:96-100 and psnr_metric :23-30 restated with numpy semantics: float MSE of
    the clipped images, and the PSNR the evaluator really prints -- its uint8 images are subtracted and
    squared IN uint8, so both wrap modulo 256.  The evaluator module itself cannot be
    imported here (cv2 / skimage absent): this row is restated from the source, not run-pinned."""
    import numpy as np
    p = np.clip(pred.detach().cpu().numpy().astype(np.float32), 0, 1)
    g = np.clip(gt.detach().cpu().numpy().astype(np.float32), 0, 1)
    mse = float(np.mean((p - g) ** 2))
    pu = (p * 255).astype(np.uint8)
    gu = (g * 255).astype(np.uint8)
    mse_u8 = float(np.mean((pu - gu) ** 2))
    psnr_printed = 100.0 if mse_u8 < 1e-10 else 10 * math.log10((255 ** 2) / mse_u8)
    return mse, psnr_printed


def evaluator_ssim(pred_hw3: torch.Tensor, gt_hw3: torch.Tensor) -> float:
    """ssim_metric of src/evaluators/nerf.py:49-77 = skimage.metrics.structural_similarity(pred_u8, gt_u8,
    win_size=7, channel_axis=2): uniform 7x7 filter, sample covariance, K1=.01, K2=.03, data_range 255, mean over
    the interior cropped by 3 pixels, averaged over channels.  skimage is absent here (and unpinned by the
    reference's requirements.txt): restated from its published algorithm with scipy's uniform_filter."""
    import numpy as np
    from scipy.ndimage import uniform_filter
    p = (np.clip(pred_hw3.detach().cpu().numpy().astype(np.float32), 0, 1) * 255).astype(np.uint8).astype(np.float64)
    g = (np.clip(gt_hw3.detach().cpu().numpy().astype(np.float32), 0, 1) * 255).astype(np.uint8).astype(np.float64)
    vals = []
    for c in range(3):
        X, Y = p[..., c], g[..., c]
        ux, uy = uniform_filter(X, size=7), uniform_filter(Y, size=7)
        uxx, uyy, uxy = uniform_filter(X * X, size=7), uniform_filter(Y * Y, size=7), uniform_filter(X * Y, size=7)
        cn = 49.0 / 48.0
        vx, vy, vxy = cn * (uxx - ux * ux), cn * (uyy - uy * uy), cn * (uxy - ux * uy)
        C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        vals.append(S[3:-3, 3:-3].mean())
    return float(np.mean(vals))
