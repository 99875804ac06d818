"""The reference's OWN rounding floor for a training step and for a K-step trajectory, measured on the CPU (round-2 VERDICT item 1c).

tests/golden/train_steps_{synthetic,trained}.npz hold K = 5 iterations of the reference's step (trainer.py:53-60, optimizer.py:8-28,
trainers/nerf.py:27-33) run by the REAL reference on one fixed 256-ray batch (oracle/gen_golden.py::training_fixtures).  Here:

 1. the CPU oracle reproduces them (loss of step 1 bit for bit, the 48 gradients to 2e-5, the parameters after the first Adam step
    to 1e-7), so the oracle is pinned on the training step, not only on the 64-ray autograd.npz;
 2. the floor: the same algorithm with encoding + MLP evaluated in float64 (everything else unchanged), and with the MLP evaluated
    in 65 536-point instead of 512-point chunks (identical arithmetic, another GEMM row grouping).  What moves:
      * the FINE network's gradient: ~1e-4 relative -- smooth in the rounding;
      * the COARSE network's gradient: 0.6 % (trained) to 3 % (synthetic) relative, single rays by 20-30 % -- and against a float64
        evaluation of the WHOLE step (sampler and adjoints too) the reference's stored fp32 gradients are off by 5.7 % / 2.2 %
        (coarse) and 0.44 % / 0.045 % (fine): near a converged scene the adjoint is a sum of large terms of both signs; it only exists through the
        inverse-CDF sampler, whose `1 / denom` terms (volume_renderer.py:259-264) amplify a 1e-7 rounding of the cdf
        by up to 1e5 and whose bin index flips;
      * the TRAJECTORY: Adam's first steps move every weight by lr * sign(g) (5e-4), so the sign of every gradient entry that is
        within rounding of zero decides a 1e-3 parameter difference: after 5 steps the fp32 and fp64 runs (and the 512- and
        65 536-chunk runs) are 3e-3 apart in parameter space and their losses differ by up to ~5 % -- although every single step agrees
        to 1e-4 in the fine gradients.  The GPU tests (tests/test_gpu_train_steps.py) judge the HIP path against THIS floor, not
        against a flat 1e-5 on the loss of step 5, which the reference cannot meet against itself.
The record goes to profiles/parity_r03.json under "training_noise_floor" when PARITY_RECORD is set.
"""
import json
import os

import pytest
import torch

import train_steps_common as T
from conftest import GOLDEN, parity_record

K = 5


def _ckpt(oracle, tag):
    ck = torch.load(os.path.join(GOLDEN, f"{tag}_ckpt.pth"), weights_only=True)["net"]
    return {k: ck[k] for k in oracle.state_dict_keys()}


@pytest.mark.parametrize("tag", ["trained", "synthetic"])
def test_oracle_reproduces_reference_training_steps(oracle, golden, tag):
    g = golden(f"train_steps_{tag}.npz")
    keys = oracle.state_dict_keys()
    tr = T.adam_trajectory(oracle, _ckpt(oracle, tag), g["rays_o"], g["rays_d"], g["target"], 2, chunk=oracle.MLP_CHUNK)
    assert torch.equal(tr["loss"][0], g["loss"][0])                                   # the forward: bit for bit
    assert torch.equal(tr["sigma_coarse_raw"][0], g["sigma_coarse_raw"][0])
    worst_g = max(((T.subsample(tr["grads"][1][k]) - g["grad1/" + k]).abs().max() / g["grad1/" + k].abs().max().clamp_min(1e-30)).item()
                  for k in keys)
    worst_p = max((T.subsample(tr["params"][1][k]) - g["param1/" + k]).abs().max().item() for k in keys)
    print(f"[{tag}] oracle vs reference: grad1 {worst_g:.2e}, param1 {worst_p:.2e}, loss2 {tr['loss'][1].item():.9f} vs {g['loss'][1].item():.9f}")
    # (torch's autograd accumulates the 512-point chunks' weight gradients in another order than the reference's module graph:
    #  measured 1.1e-6 / 1.5e-5; Adam's step is lr * g / (|g| + 1e-8): entries with |g| ~ 1e-8 move by a fraction of lr)
    assert worst_g <= 1e-4 and worst_p <= 5e-6          # (1 % of one Adam step)
    assert abs(tr["loss"][1].item() - g["loss"][1].item()) <= 1e-5 * g["loss"][1].item()
    if tag == "synthetic":
        # on this batch the REFERENCE's first Adam step kills the fine density field (every sigma <= 0: white image, zero
        # gradient); the loss is then the constant mse(white, target).  (Seen from the HIP path in round 2)
        assert torch.all(g["loss"][1:] == g["loss"][1])
        assert abs(g["loss"][1].item() - torch.nn.functional.mse_loss(torch.ones_like(g["target"]), g["target"]).item()) <= 1e-7


@pytest.mark.parametrize("tag", ["trained", "synthetic"])
def test_reference_training_noise_floor(oracle, golden, tag):
    g = golden(f"train_steps_{tag}.npz")
    keys = oracle.state_dict_keys()
    o, d, target = g["rays_o"], g["rays_d"], g["target"]
    sd0 = _ckpt(oracle, tag)
    big = 1 << 16
    a32 = T.adam_trajectory(oracle, sd0, o, d, target, K, chunk=big)
    a64 = T.adam_trajectory(oracle, sd0, o, d, target, K, mlp_dtype=torch.float64, chunk=big)
    rows = T.grad_agreement(a32["grads"][1], a64["grads"][1], keys)
    coarse, fine = T.summarize(rows, "model."), T.summarize(rows, "model_fine.")
    # per-ray sampler adjoint d loss / d sigma_coarse
    g32, g64 = a32["g_raw_c"][1][..., 3], a64["g_raw_c"][1][..., 3]
    ray_err = (g32 - g64).abs().amax(1) / g64.abs().amax(1).clamp_min(1e-30)
    cond = T.sampler_conditioning(oracle, a32["sigma_coarse_raw"][0])
    ill = cond["min_live_denom"] < 1e-3
    e2 = g64.norm(dim=1) ** 2
    loss_rel = ((a32["loss"] - a64["loss"]).abs() / a64["loss"]).tolist()
    ref_rel = ((a32["loss"] - g["loss"]).abs() / g["loss"]).tolist()             # 65 536-point chunks vs the reference's 512
    pdist = max((a32["params"][K][k] - a64["params"][K][k]).abs().max().item() for k in keys)
    # the REFERENCE's stored fp32 gradients of step 1 against a float64 evaluation of the whole step (the ground truth)
    t64 = T.staged_step_fp64(oracle, sd0, o, d, target)
    ref_vs_truth = {"model.": 0.0, "model_fine.": 0.0}
    for k in keys:
        tr = T.subsample(t64["grads"][k])
        if tr.abs().max() > 0:
            pre = "model_fine." if k.startswith("model_fine.") else "model."
            ref_vs_truth[pre] = max(ref_vs_truth[pre], ((g["grad1/" + k].double() - tr).abs().max() / tr.abs().max()).item())
    st = dict(reference_fp32_grad_vs_fp64_truth={"coarse": ref_vs_truth["model."], "fine": ref_vs_truth["model_fine."]},
              coarse_grad_fp32_vs_fp64=coarse, fine_grad_fp32_vs_fp64=fine,
              ray_adjoint_err_q50=torch.quantile(ray_err, 0.5).item(), ray_adjoint_err_q99=torch.quantile(ray_err, 0.99).item(),
              ray_adjoint_err_max=ray_err.max().item(), rays_min_denom_below_1e_3=int(ill.sum()),
              share_of_adjoint_energy_in_those_rays=(e2[ill].sum() / e2.sum()).item(),
              loss_rel_diff_per_step_fp32_vs_fp64=loss_rel, loss_rel_diff_per_step_chunk65536_vs_reference_chunk512=ref_rel,
              max_param_distance_after_K_steps=pdist,
              coarse_live_fraction_fp32=a32["coarse_live_fraction"].tolist(), coarse_live_fraction_fp64=a64["coarse_live_fraction"].tolist(),
              coarse_live_fraction_reference=g["coarse_live_fraction"].tolist())
    print(f"training noise floor [{tag}]: {json.dumps(st)}")
    if os.environ.get("PARITY_RECORD"):
        parity_record("training_noise_floor", tag, st)
    # the fine gradient is smooth in the rounding, the coarse one is not (measured: fine 1.5e-4 / 3.4e-4; coarse 3.1e-2 / 6.0e-3)
    assert fine["rel_max"] <= 2e-3 and coarse["rel_max"] >= 5 * fine["rel_max"]
    assert ray_err.max() >= 0.05 and torch.quantile(ray_err, 0.5) <= 5e-3        # single rays jump, the bulk does not
    # measured: the reference's own fp32 gradient is 5.7e-2 (coarse) / 4.4e-3 (fine) off the float64 truth on the trained batch,
    # 2.2e-2 / 4.5e-4 on the synthetic one
    assert 5e-3 <= ref_vs_truth["model."] <= 0.3 and ref_vs_truth["model_fine."] <= 2e-2
    # step 1 is the forward: identical to rounding; the trajectory then leaves it by far more than 1e-5 unless the field is dead
    assert loss_rel[0] <= 1e-5 and ref_rel[0] <= 1e-5
    if tag == "trained":
        assert max(loss_rel[2:]) >= 1e-3 and max(ref_rel[2:]) >= 1e-3 and pdist >= 1e-3
    else:
        assert max(loss_rel[1:]) <= 1e-6          # the dead field: constant loss in every arithmetic
