"""Shared helpers of the multi-step training parity tests (test infrastructure; CPU oracle side).

The reference trains by iterating trainer.py:53-60 (`loss.backward(); clip_grad_value_(.., 40); optimizer.step()`) with the
Adam of optimizer.py:8-28 on the loss of trainers/nerf.py:27-33 (MSE on the fine RGB).  tests/golden/train_steps_*.npz hold K = 5
such iterations run by the REAL reference (oracle/gen_golden.py::training_fixtures).  The helpers below restate that loop
on the CPU oracle, with two extras the tests need:
  * the coarse raw output is a LEAF of the graph, so d loss / d sigma_coarse of every ray (the "sampler adjoint", the only way
    the coarse network learns) can be read off and compared ray by ray;
  * per-ray conditioning of the inverse-CDF sampler (volume_renderer.py:247-264): the smallest live `denom`, how close any
    `denom` is to the `< 1e-5 -> 1` switch, how close any searchsorted comparison is to flipping.
"""
import torch


def subsample(t):
    """The fixture's view of a parameter-shaped tensor (oracle/gen_golden.py::_subsample): small tensors in full, weight
    matrices as their flat stride-7 subsample."""
    f = t.detach().reshape(-1)
    return f.clone() if f.numel() <= 4096 else f[::7].clone()


def sampler_conditioning(oracle, sigma_c_raw, eps=1e-5):
    """Per-ray conditioning figures of fine_sample on the coarse densities `sigma_c_raw` [n,64] (pre-ReLU)."""
    n = sigma_c_raw.shape[0]
    t_c = oracle.stratified_t().expand(n, 64)
    with torch.no_grad():
        sig = torch.relu(sigma_c_raw)
        _, parts = oracle.fine_sample(sig, t_c, return_parts=True)
        cdf, below, above = parts["cdf"], parts["below"], parts["above"]
        u = oracle.fine_u().expand(n, 128)
        denom = torch.gather(cdf, 1, above) - torch.gather(cdf, 1, below)
        live = denom >= eps
        big = torch.full_like(denom, 1e9)
        min_live_denom = torch.where(live, denom, big).min(1).values               # smallest divisor actually used
        switch_gap = ((denom - eps).abs() / eps).min(1).values                      # relative distance of any denom to the switch
        flip_gap = (cdf[:, None, :] - u[:, :, None]).abs().min(2).values.min(1).values   # distance of any (cdf_k, u_j) comparison to a tie
    return dict(min_live_denom=min_live_denom, switch_gap=switch_gap, flip_gap=flip_gap, n_dead_denoms=(~live).sum(1))


def staged_step(oracle, sd, o, d, target, mlp_dtype=torch.float32, chunk=None, raw_c_given=None):
    """One forward + backward of the reference's loss with the coarse raw output as a leaf.  `sd`: dict of leaf tensors
    (requires_grad).  Returns loss, rgb, raw_c (detached), g_raw_c [n,64,4] = d loss / d raw_coarse, and leaves the parameter
    gradients in sd[k].grad (the coarse ones through raw_c's own backward)."""
    chunk = chunk or oracle.MLP_CHUNK
    n = o.shape[0]
    t_c = oracle.stratified_t().unsqueeze(0).expand(n, 64).clone()
    pts_c = oracle.points_on_rays(o, d, t_c)
    vd = d / torch.norm(d, dim=-1, keepdim=True)
    raw_c = oracle.network_forward(sd, pts_c, vd, "", chunk, mlp_dtype)
    leaf = (raw_c if raw_c_given is None else raw_c_given).detach().clone().requires_grad_(True)
    sigma_c = torch.relu(leaf[..., 3])
    t_f = oracle.fine_sample(sigma_c, t_c)
    pts_f = oracle.points_on_rays(o, d, t_f)
    depth, order = torch.sort(torch.cat([t_c, t_f], 1), dim=-1)
    pts = torch.gather(torch.cat([pts_c, pts_f], 1), 1, order[..., None].expand(-1, -1, 3))
    raw_f = oracle.network_forward(sd, pts, vd, "fine", chunk, mlp_dtype)
    rgb, dep = oracle.composite(raw_f, depth, True)
    loss = torch.nn.functional.mse_loss(rgb, target)
    loss.backward()
    g_raw_c = leaf.grad.detach().clone()
    if raw_c_given is None:
        raw_c.backward(g_raw_c)                 # carries the sampler adjoint on into the coarse parameters
    return dict(loss=loss.detach(), rgb=rgb.detach(), depth=dep.detach(), raw_c=raw_c.detach(), g_raw_c=g_raw_c,
                t_sorted=depth.detach())


def staged_step_fp64(oracle, sd, o, d, target, raw_c_given=None):
    """The same loss and backward pass evaluated ENTIRELY in float64 on the same fp32 inputs and weights (not the reference's
    arithmetic: the ground truth both fp32 evaluations -- torch's on the CPU and the HIP kernels' -- are judged against: the
    adjoint of a nearly converged scene is a sum of large terms of both signs, so fp32 itself is only good to ~1e-2 per ray).
    `raw_c_given`: use these coarse outputs as the leaf (attribution on identical densities).  Returns g_raw_c [n,64,4] (float64),
    the bins (below, above) it sampled from, the float64 parameter gradients and the loss."""
    keys = oracle.state_dict_keys()
    sd64 = {k: sd[k].detach().double().requires_grad_(True) for k in keys}
    o, d, target = o.double(), d.double(), target.double()
    n = o.shape[0]
    t_c = oracle.stratified_t().double().unsqueeze(0).expand(n, 64).clone()
    pts_c = oracle.points_on_rays(o, d, t_c)
    vd = d / torch.norm(d, dim=-1, keepdim=True)

    def mlp(prefix, pts, s):
        flat = pts.reshape(-1, 3)
        dflat = vd[:, None].expand(n, s, 3).reshape(-1, 3)
        emb = torch.cat([oracle.freq_encode(flat, oracle.XYZ_FREQS), oracle.freq_encode(dflat, oracle.DIR_FREQS)], -1)
        return oracle.nerf_mlp(sd64, prefix, emb).reshape(n, s, 4)

    raw_c = mlp("model", pts_c, 64)
    leaf = (raw_c if raw_c_given is None else raw_c_given.double()).detach().clone().requires_grad_(True)
    sigma_c = torch.relu(leaf[..., 3])
    _, w = oracle.transmittance_weights(sigma_c, t_c)
    w = w[:, 1:-1] + 1e-5
    cdf = torch.cumsum(w / torch.sum(w, -1, keepdim=True), -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = oracle.fine_u().double().expand(n, 128).contiguous()
    inds = torch.searchsorted(cdf.detach(), u, right=True)
    below, above = torch.clamp(inds - 1, 0, 61), torch.clamp(inds, 0, 61)
    bins = 0.5 * (t_c[:, 1:] + t_c[:, :-1])
    cb, ca = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bb, ba = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = ca - cb
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t_f = bb + (u - cb) / denom * (ba - bb)
    pts_f = oracle.points_on_rays(o, d, t_f)
    depth, order = torch.sort(torch.cat([t_c, t_f], 1), dim=-1)
    pts = torch.gather(torch.cat([pts_c, pts_f], 1), 1, order[..., None].expand(-1, -1, 3))
    raw_f = mlp("model_fine", pts, 192)
    rgb, _ = oracle.composite(raw_f, depth, True)
    loss = torch.nn.functional.mse_loss(rgb, target)
    loss.backward()
    g_raw_c = leaf.grad.detach().clone()
    if raw_c_given is None:
        raw_c.backward(g_raw_c)
    grads = {k: (sd64[k].grad.detach().clone() if sd64[k].grad is not None else torch.zeros_like(sd64[k])) for k in keys}
    return dict(loss=loss.detach(), g_raw_c=g_raw_c, below=below, above=above, grads=grads)


def fine_pass_adjoint(oracle, sd, o, d, target, t_sorted, dtype=torch.float32):
    """Stage (i) of the attribution: d loss / d t_sorted of the fine pass alone -- points o + d t on GIVEN merged sample depths
    -> fine network -> compositing -> MSE -- under torch autograd in `dtype` (fp32: the reference's arithmetic; float64: truth)."""
    keys = [k for k in oracle.state_dict_keys() if k.startswith("model_fine.")]
    sdx = {k: sd[k].detach().to(dtype) for k in keys}
    o, d, target = o.to(dtype), d.to(dtype), target.to(dtype)
    n = o.shape[0]
    leaf = t_sorted.detach().to(dtype).clone().requires_grad_(True)
    vd = d / torch.norm(d, dim=-1, keepdim=True)
    pts = oracle.points_on_rays(o, d, leaf)
    flat = pts.reshape(-1, 3)
    dflat = vd[:, None].expand(n, leaf.shape[1], 3).reshape(-1, 3)
    emb = torch.cat([oracle.freq_encode(flat, oracle.XYZ_FREQS), oracle.freq_encode(dflat, oracle.DIR_FREQS)], -1)
    raw_f = oracle.nerf_mlp(sdx, "model_fine", emb).reshape(n, leaf.shape[1], 4)
    rgb, _ = oracle.composite(raw_f, leaf, True)
    torch.nn.functional.mse_loss(rgb, target).backward()
    return leaf.grad.detach()


def sampler_adjoint(oracle, raw_c, g_t_sorted, dtype=torch.float32):
    """Stage (ii): d / d sigma_coarse of sum(t_sorted * G) through fine_sample + merge (volume_renderer.py:126-154, :247-264, :349-353)
    under torch autograd in `dtype`, for given coarse outputs and a given upstream gradient.  Returns (g_sigma [n,64], below, above)."""
    n = raw_c.shape[0]
    leaf = raw_c.detach().to(dtype).clone().requires_grad_(True)
    t_c = oracle.stratified_t().to(dtype).unsqueeze(0).expand(n, 64)
    sigma = torch.relu(leaf[..., 3])
    _, w = oracle.transmittance_weights(sigma, t_c)
    w = w[:, 1:-1] + 1e-5
    cdf = torch.cumsum(w / torch.sum(w, -1, keepdim=True), -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = oracle.fine_u().to(dtype).expand(n, 128).contiguous()
    inds = torch.searchsorted(cdf.detach(), u, right=True)
    below, above = torch.clamp(inds - 1, 0, 61), torch.clamp(inds, 0, 61)
    bins = 0.5 * (t_c[:, 1:] + t_c[:, :-1])
    cb, ca = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bb, ba = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = ca - cb
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t_f = bb + (u - cb) / denom * (ba - bb)
    t_sorted, _ = torch.sort(torch.cat([t_c, t_f], 1), dim=-1)
    (t_sorted * g_t_sorted.to(dtype)).sum().backward()
    return leaf.grad.detach()[..., 3], below, above


def fp32_bins(oracle, raw_c):
    """(below, above) of the fp32 sampler on these coarse outputs."""
    with torch.no_grad():
        n = raw_c.shape[0]
        _, parts = oracle.fine_sample(torch.relu(raw_c[..., 3]), oracle.stratified_t().expand(n, 64), return_parts=True)
    return parts["below"], parts["above"]


def adam_trajectory(oracle, sd0, o, d, target, K, mlp_dtype=torch.float32, chunk=None, keep_grads=(1,)):
    """K iterations of the reference's step on the CPU oracle (torch.optim.Adam as optimizer.py:8-28 builds it, clip 40)."""
    keys = oracle.state_dict_keys()
    params = {k: sd0[k].detach().clone().requires_grad_(True) for k in keys}
    opt = torch.optim.Adam([{"params": [params[k]], "lr": 5e-4, "weight_decay": 0.0, "eps": 1e-8} for k in keys],
                           5e-4, weight_decay=0.0, eps=1e-8)
    out = dict(loss=[], coarse_live_fraction=[], sigma_coarse_raw=[], params={}, grads={}, g_raw_c={})
    for step in range(1, K + 1):
        opt.zero_grad()
        r = staged_step(oracle, params, o, d, target, mlp_dtype, chunk)
        if step in keep_grads:
            out["grads"][step] = {k: params[k].grad.detach().clone() for k in keys}
            out["g_raw_c"][step] = r["g_raw_c"]
        torch.nn.utils.clip_grad_value_(list(params.values()), 40)
        opt.step()
        out["loss"].append(r["loss"])
        out["coarse_live_fraction"].append((r["raw_c"][..., 3] > 0).float().mean())
        out["sigma_coarse_raw"].append(r["raw_c"][..., 3].clone())
        if step in (1, K):
            out["params"][step] = {k: params[k].detach().clone() for k in keys}
    out["loss"] = torch.stack(out["loss"])
    out["coarse_live_fraction"] = torch.stack(out["coarse_live_fraction"])
    return out


def grad_agreement(a, b, keys):
    """Per-tensor comparison of two gradient dicts: max-norm relative difference, cosine, share of entries whose SIGN differs
    (what Adam's first step, lr * sign(g), turns into a parameter difference of 2 lr), and that share among the entries
    above 1e-3 of the tensor's maximum."""
    rows = {}
    for k in keys:
        x, y = a[k].double().reshape(-1), b[k].double().reshape(-1)
        scale = y.abs().max().clamp_min(1e-30)
        nz = (x != 0) | (y != 0)
        flips = ((torch.sign(x) != torch.sign(y)) & nz)
        big = y.abs() > 1e-3 * scale
        rows[k] = dict(rel_max=((x - y).abs().max() / scale).item(),
                       rel_l2=((x - y).norm() / y.norm().clamp_min(1e-30)).item(),
                       cos=(torch.dot(x, y) / (x.norm() * y.norm()).clamp_min(1e-30)).item(),
                       sign_flips=(flips.float().sum() / nz.float().sum().clamp_min(1)).item(),
                       sign_flips_big=((flips & big).float().sum() / big.float().sum().clamp_min(1)).item())
    return rows


def summarize(rows, prefix):
    sel = {k: v for k, v in rows.items() if k.startswith(prefix)}
    nonzero = {k: v for k, v in sel.items() if v["rel_l2"] == v["rel_l2"] and v["cos"] != 0.0}
    if not nonzero:
        return dict(n=0)
    return dict(n=len(nonzero), rel_max=max(v["rel_max"] for v in nonzero.values()), rel_l2=max(v["rel_l2"] for v in nonzero.values()),
                min_cos=min(v["cos"] for v in nonzero.values()), max_sign_flips=max(v["sign_flips"] for v in nonzero.values()),
                max_sign_flips_big=max(v["sign_flips_big"] for v in nonzero.values()))
