"""-m gpu: K-step training trajectories of the HIP path against the REAL reference's (round-2 VERDICT item 1).

Fixtures: tests/golden/train_steps_{synthetic,trained}.npz -- K = 5 iterations of trainer.py:53-60 (`loss.backward();
clip_grad_value_(40); Adam(5e-4, eps 1e-8).step()`, optimizer.py:8-28) on the loss of trainers/nerf.py:27-33, run by the reference
itself on one fixed 256-ray pinhole batch (oracle/gen_golden.py::training_fixtures).  Compared, for precision f32 and f32x:

  step 1   loss (<= 1e-5 relative: it is the forward), the image, all 48 gradients: the FINE network's tightly, the COARSE
           network's -- which exist only through the inverse-CDF sampler (volume_renderer.py:255-267) -- ATTRIBUTED:
           (a) all 48 gradients against a float64 evaluation of the same step: the HIP gradients must be as close to that ground
           truth as the REFERENCE's own fp32 gradients are (the reference is 5.7 % / 0.44 % off it on the trained batch: near a
           converged scene the adjoint is a sum of large terms of both signs); (b) with the HIP path's own coarse densities fed to
           the CPU oracle's sampler + fine pass, the per-ray sampler adjoint d loss / d sigma_coarse of the HIP kernels and of torch's
           fp32 autograd, both measured from the float64 adjoint, on every ray whose float64 bins equal the fp32 ones (rays with a
           searchsorted / `denom < 1e-5` flip are set aside; the share of the adjoint's energy they carry is recorded);
  step 1   the parameters after the first Adam step (lr * g / (|g| + eps): the sign pattern of the gradient);
  steps 2..K  loss and share of live coarse samples per step, parameters after step K -- judged against the reference's OWN floor,
           measured in the same test on the CPU oracle (fp32 vs float64 MLP; tests/test_train_noise_floor.py explains why a flat
           1e-5 on the loss of step 5 is not meetable by the reference against itself: 5 % between 512- and 65 536-point chunks).
Everything measured goes to profiles/parity_r03.json, section "training_steps".
"""
import json
import os

import pytest
import torch

import train_steps_common as T
from conftest import GOLDEN, parity_record

pytestmark = pytest.mark.gpu
K = 5


@pytest.fixture(scope="module")
def amd():
    import nerf_replication_amd as pkg
    pkg._lib.load()
    return pkg


def _ckpt(oracle, tag):
    ck = torch.load(os.path.join(GOLDEN, f"{tag}_ckpt.pth"), weights_only=True)["net"]
    return {k: ck[k] for k in oracle.state_dict_keys()}


_FLOOR = {}


def _floor(oracle, g, tag):
    """fp32 vs float64-MLP trajectories of the CPU oracle on the fixture batch (the reference's own rounding floor)."""
    if tag not in _FLOOR:
        sd0 = _ckpt(oracle, tag)
        a32 = T.adam_trajectory(oracle, sd0, g["rays_o"], g["rays_d"], g["target"], K, chunk=1 << 16)
        a64 = T.adam_trajectory(oracle, sd0, g["rays_o"], g["rays_d"], g["target"], K, mlp_dtype=torch.float64, chunk=1 << 16)
        keys = oracle.state_dict_keys()
        rows = T.grad_agreement(a32["grads"][1], a64["grads"][1], keys)
        _FLOOR[tag] = dict(
            coarse=T.summarize(rows, "model."), fine=T.summarize(rows, "model_fine."),
            loss_rel=[max(a, b) for a, b in zip(((a32["loss"] - a64["loss"]).abs() / a64["loss"]).tolist(),
                                                ((a32["loss"] - g["loss"]).abs() / g["loss"]).tolist())],
            pdist=max((a32["params"][K][k] - a64["params"][K][k]).abs().max().item() for k in keys),
            live=(a32["coarse_live_fraction"] - a64["coarse_live_fraction"]).abs().max().item())
    return _FLOOR[tag]


@pytest.mark.parametrize("precision", ["f32", "f32x"])
@pytest.mark.parametrize("tag", ["trained", "synthetic"])
def test_training_trajectory_matches_reference(amd, oracle, golden, tag, precision):
    from nerf_replication_amd.training import FusedAdam, train_step
    g = golden(f"train_steps_{tag}.npz")
    keys = oracle.state_dict_keys()
    sd0 = _ckpt(oracle, tag)
    net = amd.Network()
    net.load_state_dict(sd0, strict=True)
    net = net.cuda().train()
    net.precision = precision
    ren = amd.Renderer(net)
    o, d, target = g["rays_o"].cuda(), g["rays_d"].cuda(), g["target"].cuda()
    opt = FusedAdam(net.parameters(), lr=5e-4, eps=1e-8, clip_value=40.0)
    named = dict(net.named_parameters())
    assert list(named) == keys
    losses, live, cap = [], [], {}
    grads1 = params1 = None
    for step in range(1, K + 1):
        ren.capture_adjoints = cap if step == 1 else None
        losses.append(train_step(ren, opt, o, d, target).item())
        if step == 1:
            grads1 = {k: named[k].grad.detach().cpu().clone() for k in keys}
            params1 = {k: named[k].detach().cpu().clone() for k in keys}
            live.append((cap["raw_coarse"][..., 3] > 0).float().mean().item())
        else:
            live.append(float("nan"))
    ren.capture_adjoints = None
    paramsK = {k: named[k].detach().cpu().clone() for k in keys}
    with torch.no_grad():       # coarse densities after K steps (the share of live coarse samples the next step would see)
        t_c = oracle.stratified_t().unsqueeze(0).expand(256, 64).contiguous()
        pts_c = oracle.points_on_rays(g["rays_o"], g["rays_d"], t_c).cuda()
        vd = (g["rays_d"] / g["rays_d"].norm(dim=-1, keepdim=True)).cuda()
        net.eval()
        live_after = (net.forward(pts_c, vd, None, model="")[..., 3] > 0).float().mean().item()
        net.train()
    floor = _floor(oracle, g, tag)

    # ---- step 1: loss, gradients
    loss_rel = [abs(a - b) / b for a, b in zip(losses, g["loss"].tolist())]
    rows = {}
    for k in keys:
        ref, got = g["grad1/" + k].double(), T.subsample(grads1[k]).double()
        scale = ref.abs().max().clamp_min(1e-30)
        rows[k] = ((got - ref).abs().max() / scale).item() if ref.abs().max() > 0 else got.abs().max().item()
    fine = max(v for k, v in rows.items() if k.startswith("model_fine."))
    coarse = max(v for k, v in rows.items() if k.startswith("model."))
    p1 = max((T.subsample(params1[k]) - g["param1/" + k]).abs().max().item() for k in keys if k.startswith("model_fine."))
    # sign pattern after one Adam step: share of (subsampled) entries that moved the other way than the reference's
    def moved_other_way(prefix):
        bad = tot = 0
        for k in keys:
            if not k.startswith(prefix):
                continue
            p0 = T.subsample(sd0[k])
            a, b = T.subsample(params1[k]) - p0, g["param1/" + k] - p0
            sel = b.abs() > 2.5e-4                    # entries the reference moved by at least half a step
            bad += int(((a * b) < 0)[sel].sum()); tot += int(sel.sum())
        return bad / max(1, tot)
    flips_f, flips_c = moved_other_way("model_fine."), moved_other_way("model.")

    # ---- step 1 against the float64 GROUND TRUTH of the same step (train_steps_common.staged_step_fp64): the reference's own
    # fp32 gradients are only good to ~5 % (coarse) / 0.4 % (fine) on the trained batch, so "equal to the reference" is judged as
    # "as close to the truth as the reference is"
    t64 = T.staged_step_fp64(oracle, sd0, g["rays_o"], g["rays_d"], g["target"])
    def vs_truth(get):
        worst = {"model.": 0.0, "model_fine.": 0.0}
        for k in keys:
            tr = T.subsample(t64["grads"][k])
            if tr.abs().max() == 0:
                continue
            pre = "model_fine." if k.startswith("model_fine.") else "model."
            worst[pre] = max(worst[pre], ((get(k).double() - tr).abs().max() / tr.abs().max()).item())
        return worst
    hip_vs_truth = vs_truth(lambda k: T.subsample(grads1[k]))
    ref_vs_truth = vs_truth(lambda k: g["grad1/" + k])

    # ---- step 1, attributed stage by stage on the HIP path's OWN intermediates (captured in RenderFunction.backward), each stage
    # against torch autograd of the CPU oracle in fp32 (the reference's arithmetic) and in float64 (the truth):
    #  (i)  fine pass: d loss / d t_sorted at the HIP sampler's merged depths (compositing adjoint + MLP chain + d point / d t);
    #  (ii) sampler: d / d sigma_coarse of that upstream gradient on the HIP path's coarse densities -- a LINEAR map of the
    #       upstream gradient, so nothing but the adjoint arithmetic is compared; rays on which the float64 sampler picks other
    #       bins than the fp32 one (a searchsorted / `denom < 1e-5` flip) are set aside and their share of the energy recorded.
    # (Feeding only the densities and letting the oracle re-place the samples does not separate anything: any two implementations
    #  place the fine samples of an ill-conditioned ray 1e-5 apart -- different exp() -- and the synthetic field changes by 1 % over
    #  that distance: tests/test_noise_floor.py.)
    raw_c_hip, ts_hip, gts_hip = cap["raw_coarse"].cpu(), cap["t_sorted"].cpu(), cap["g_t_sorted"].cpu()
    gt32 = T.fine_pass_adjoint(oracle, sd0, g["rays_o"], g["rays_d"], g["target"], ts_hip)
    gt64 = T.fine_pass_adjoint(oracle, sd0, g["rays_o"], g["rays_d"], g["target"], ts_hip, torch.float64)
    sc_t = gt64.abs().amax(1).clamp_min(1e-30)
    live_t = gt64.abs().amax(1) > 0
    gt_hip_err = ((gts_hip.double() - gt64).abs().amax(1) / sc_t)[live_t]
    gt_cpu_err = ((gt32.double() - gt64).abs().amax(1) / sc_t)[live_t]
    q4 = lambda e: [torch.quantile(e, q).item() for q in (0.5, 0.9, 0.99, 1.0)] if e.numel() else [0.0] * 4
    qt_hip, qt_cpu = q4(gt_hip_err), q4(gt_cpu_err)
    g32, b32, a32 = T.sampler_adjoint(oracle, raw_c_hip, gts_hip)
    g64, b64, a64 = T.sampler_adjoint(oracle, raw_c_hip, gts_hip, torch.float64)
    same_bins = ((b32 == b64) & (a32 == a64)).all(1)
    ga = cap["g_raw_coarse"].cpu()[..., 3].double()
    assert torch.all(cap["g_raw_coarse"][..., :3] == 0)
    n = ga.shape[0]
    scale = g64.abs().amax(1).clamp_min(1e-30)
    err_hip, err_cpu = (ga - g64).abs().amax(1) / scale, (g32.double() - g64).abs().amax(1) / scale
    live_rays = same_bins & (g64.abs().amax(1) > 0)
    q_hip, q_cpu = q4(err_hip[live_rays]), q4(err_cpu[live_rays])
    e2 = g64.norm(dim=1) ** 2
    share_excluded = (e2[~same_bins].sum() / e2.sum().clamp_min(1e-300)).item()
    # the coarse-density agreement itself (forward): HIP vs the reference's stored sigma of step 1
    sig_err = (raw_c_hip[..., 3] - g["sigma_coarse_raw"][0]).abs().max().item() / g["sigma_coarse_raw"][0].abs().max().item()

    pK_f = max((T.subsample(paramsK[k]) - g[f"param{K}/" + k]).abs().max().item() for k in keys if k.startswith("model_fine."))
    pK_c = max((T.subsample(paramsK[k]) - g[f"param{K}/" + k]).abs().max().item() for k in keys if k.startswith("model."))
    st = dict(loss=losses, loss_reference=g["loss"].tolist(), loss_rel_err=loss_rel, floor_loss_rel=floor["loss_rel"],
              grad1_fine_worst_rel_err=fine, grad1_coarse_worst_rel_err=coarse,
              floor_grad1_fine=floor["fine"]["rel_max"], floor_grad1_coarse=floor["coarse"]["rel_max"],
              param1_fine_max_abs_diff=p1, moved_other_way_fine=flips_f, moved_other_way_coarse=flips_c,
              sigma_coarse_rel_err=sig_err,
              grad1_vs_fp64_truth={"hip_coarse": hip_vs_truth["model."], "reference_coarse": ref_vs_truth["model."],
                                   "hip_fine": hip_vs_truth["model_fine."], "reference_fine": ref_vs_truth["model_fine."]},
              fine_pass_adjoint_vs_fp64_truth={"hip_q50_q90_q99_max": qt_hip, "torch_cpu_fp32_q50_q90_q99_max": qt_cpu},
              sampler_adjoint_vs_fp64_truth={"rays": n, "same_bins_in_fp64": int(same_bins.sum()), "share_of_energy_in_excluded_rays": share_excluded,
                                             "hip_q50_q90_q99_max": q_hip, "torch_cpu_fp32_q50_q90_q99_max": q_cpu},
              paramK_fine_max_abs_diff=pK_f, paramK_coarse_max_abs_diff=pK_c, floor_paramK=floor["pdist"],
              coarse_live_fraction_step1=live[0], coarse_live_fraction_reference=g["coarse_live_fraction"].tolist(),
              coarse_live_fraction_after_K=live_after)
    print(f"training trajectory [{tag}/{precision}]: {json.dumps(st)}")
    parity_record("training_steps", f"{tag}/{precision}", st)

    assert loss_rel[0] <= 1e-5                                           # the forward
    assert (cap["raw_coarse"][..., 3].cpu() > 0).float().mean().item() == pytest.approx(g["coarse_live_fraction"][0].item(), abs=2e-4)
    # the fine gradient against the reference's: inside the class of the reference's own distance from the float64 truth (trained
    # batch: 4.4e-3 -- a converged fine network is sharp, sample positions that differ by 1e-6 move it by 3e-3; synthetic: 4.5e-4)
    assert fine <= 3.0 * ref_vs_truth["model_fine."] + 2e-4, (fine, ref_vs_truth)
    # the HIP gradients are as close to the float64 truth as the reference's own fp32 gradients are (measured, trained batch:
    # reference 5.7e-2 coarse / 4.4e-3 fine)
    assert hip_vs_truth["model."] <= 2.0 * ref_vs_truth["model."] + 5e-3, (hip_vs_truth, ref_vs_truth)
    assert hip_vs_truth["model_fine."] <= 2.0 * ref_vs_truth["model_fine."] + 2e-4, (hip_vs_truth, ref_vs_truth)
    assert coarse <= 3.0 * ref_vs_truth["model."] + 5e-3                 # and the direct difference is inside that error class
    # attributed, ray by ray: (i) the fine pass's adjoint and (ii) the sampler's adjoint, HIP kernels and torch's fp32 autograd both
    # measured from the float64 truth on identical inputs
    # (f32x on the trained batch: q99 1.3e-2 / max 4.3e-2 against torch's 5.6e-3 / 1.4e-2 -- the split-fp16 chain renormalises every
    #  layer's gradient by a power of two and is a little further from float64 than the exact-fp32 chain's 2.0e-4 / 4.3e-4)
    assert qt_hip[2] <= 4.0 * qt_cpu[2] + 1e-4 and qt_hip[3] <= 4.0 * qt_cpu[3] + 2e-3, (qt_hip, qt_cpu)
    assert q_hip[2] <= 3.0 * q_cpu[2] + 1e-3 and q_hip[3] <= 3.0 * q_cpu[3] + 5e-3, (q_hip, q_cpu)
    assert q_hip[0] <= 3.0 * q_cpu[0] + 1e-4, (q_hip, q_cpu)
    assert flips_f <= 1e-3
    # steps 2..K against the floor (3x the reference's own fp32 / fp64 / chunking spread, plus rounding)
    for s in range(1, K):
        assert loss_rel[s] <= 3.0 * max(floor["loss_rel"][s], floor["loss_rel"][max(1, s - 1)]) + 1e-3, (s, loss_rel, floor["loss_rel"])
    assert pK_f <= 3.0 * floor["pdist"] + 1e-3 and pK_c <= 3.0 * floor["pdist"] + 1e-3
    assert abs(live_after - g["coarse_live_fraction"][K - 1].item()) <= 0.1      # the coarse field is alive (or dead) like the reference's
