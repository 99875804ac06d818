"""Generate tests/golden/*.npz by running the REAL reference on CPU  --  test infrastructure.

Runs only in the build container (needs /root/reference; never on the GPU box).  It
imports the reference's Network / Renderer / Encoder, loads a seeded state_dict, runs
seeded inputs through them and records inputs, intermediates and outputs.  The fixtures
are plain data (float/int arrays); no reference source is copied.

    python oracle/gen_golden.py            # rewrites tests/golden/

Import recipe per SURVEY.md section 8(c): argv/cwd must be set before `src.config` is imported
(argparse-at-import, CWD-relative paths); torch.cuda.synchronize is neutralised because
volume_renderer.py:382,405 call it unconditionally (it does not touch arithmetic).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, HERE)
import nerf_oracle as orc  # noqa: E402  (inputs/seeded weights come from the oracle helpers)


def import_reference():
    sys.argv = ["gen_golden", "--cfg_file", os.path.join(REF, "configs/nerf/lego.yaml")]
    sys.path.insert(0, REF)
    os.chdir(REF)
    torch.cuda.synchronize = lambda *a, **k: None
    from src.models.nerf.network import Network
    from src.models.nerf.renderer.volume_renderer import Renderer
    from src.models.encoding.freq import Encoder  # noqa: F401
    return Network, Renderer


seeded_rays = orc.seeded_rays


def npz(name, **arrs):
    conv = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(OUT, name), **conv)
    print("wrote", name, {k: v.shape for k, v in conv.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    Network, Renderer = import_reference()
    sd = orc.synthetic_state_dict(seed=0)
    net = Network()
    assert list(net.state_dict().keys()) == orc.state_dict_keys(), "state_dict key drift"
    net.load_state_dict(sd, strict=True)
    net.eval()
    ren = Renderer(net)
    assert ren.perturb is False and ren.N_samples == 64 and ren.N_importance == 128
    ren.device = torch.device("cpu")

    # checkpoint in the reference's {"net": state_dict} layout (net_utils.py:374-375)
    torch.save({"net": {k: v.clone() for k, v in sd.items()}, "epoch": 0}, os.path.join(OUT, "synthetic_ckpt.pth"))

    with torch.no_grad():
        # (1) positional encoding
        g = torch.Generator().manual_seed(11)
        x = (torch.rand(64, 3, generator=g) * 2 - 1) * 4.0
        dirs = torch.randn(64, 3, generator=g); dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        npz("pe.npz", x=x, dirs=dirs, pe_xyz=net.embed_fn(x), pe_dir=net.embeddirs_fn(dirs))

        # (2) NeRF.forward per-layer activations, 128 points, both sub-models
        pts = (torch.rand(128, 3, generator=g) * 2 - 1) * 3.0
        vd = torch.randn(128, 3, generator=g); vd = vd / vd.norm(dim=-1, keepdim=True)
        emb = torch.cat([net.embed_fn(pts), net.embeddirs_fn(vd)], -1)
        rec = {"pts": pts, "viewdirs": vd, "emb": emb}
        for tag, mdl in (("coarse", net.model), ("fine", net.model_fine)):
            acts = {}
            hooks = []
            for i, l in enumerate(mdl.pts_linears):
                hooks.append(l.register_forward_hook(lambda m, a, o, i=i: acts.__setitem__(f"h{i}", torch.relu(o))))
            hooks.append(mdl.feature_linear.register_forward_hook(lambda m, a, o: acts.__setitem__("feature", o)))
            hooks.append(mdl.views_linears[0].register_forward_hook(lambda m, a, o: acts.__setitem__("views", torch.relu(o))))
            out = mdl(emb)
            for h in hooks:
                h.remove()
            rec[f"{tag}_out"] = out
            for k, v in acts.items():
                rec[f"{tag}_{k}"] = v
        npz("mlp_layers.npz", **rec)

        # (3) Network.forward [8,64,3] -> [8,64,4], both models
        o8, d8 = seeded_rays(8, 3)
        t_c, pts_c = ren.stratified_sample_points_from_rays(o8, d8, N_samples=64, perturb=False)
        vd8 = d8 / torch.norm(d8, dim=-1, keepdim=True)
        npz("network_forward.npz", pts=pts_c, viewdirs=vd8,
            raw_coarse=net.forward(pts_c, vd8, None, model=""),
            raw_fine=net.forward(pts_c, vd8, None, model="fine"))

        # (4)-(6) weights, fine sampling (incl. F7 tail), sorted depths
        o, d = seeded_rays(256, 5)
        t_c, pts_c = ren.stratified_sample_points_from_rays(o, d, N_samples=64, perturb=False)
        vd = d / torch.norm(d, dim=-1, keepdim=True)
        raw_c = net.forward(pts_c, vd, None, model="")
        sigma_c = torch.relu(raw_c[..., 3])
        T64, w64 = ren.weights_computation(sigma_c, t_c)
        pts_f, t_f, vm = ren.fine_sample_points(sigma_c, o, d, t_c, 128, 64, 0.25)
        assert vm is None
        # re-derive the internals the reference does not return, with the reference's own ops
        w = w64[:, 1:-1] + 1e-5
        cdf = torch.cumsum(w / torch.sum(w, -1, keepdim=True), -1)
        cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
        u = torch.linspace(0., 1., steps=128).expand(256, 128).contiguous()
        inds = torch.searchsorted(cdf, u, right=True)
        t_sorted, order = torch.sort(torch.cat([t_c, t_f], 1), dim=-1)
        raw_f = net.forward(torch.gather(torch.cat([pts_c, pts_f], 1), 1, order[..., None].expand(-1, -1, 3)),
                            vd, None, model="fine")
        T192, w192 = ren.weights_computation(torch.relu(raw_f[..., 3]), t_sorted)
        npz("sampling.npz", rays_o=o, rays_d=d, t_coarse=t_c, raw_coarse=raw_c, T64=T64, w64=w64,
            cdf=cdf, inds=inds, t_fine=t_f, pts_fine=pts_f, t_sorted=t_sorted, raw_fine=raw_f,
            T192=T192, w192=w192)

        # (7) full render, N=256, N_importance in {0,128}; seeded-direction rays and pinhole rays
        batch = {"rays_o": o[None], "rays_d": d[None]}
        rgb128, dep128 = ren.render(batch)
        ren.N_importance = 0
        rgb0, dep0 = ren.render(batch)
        ren.N_importance = 128
        c2w = orc.camera_pose(30.0)
        ids = torch.from_numpy(np.random.default_rng(0).choice(800 * 800, 256, replace=False))
        po, pd = orc.pinhole_rays(800, 800, c2w, pixel_ids=ids)
        prgb, pdep = ren.render({"rays_o": po[None], "rays_d": pd[None]})
        npz("render.npz", rays_o=o, rays_d=d, rgb_128=rgb128, depth_128=dep128, rgb_0=rgb0, depth_0=dep0,
            pin_rays_o=po, pin_rays_d=pd, pin_rgb=prgb, pin_depth=pdep, pin_c2w=c2w, pin_ids=ids)

        # (7b) B>1 batch layout [2,96,3] -> [192,3]
        o2, d2 = seeded_rays(192, 9)
        r2, z2 = ren.render({"rays_o": o2.reshape(2, 96, 3), "rays_d": d2.reshape(2, 96, 3)})
        npz("render_batched.npz", rays_o=o2.reshape(2, 96, 3), rays_d=d2.reshape(2, 96, 3), rgb=r2, depth=z2)

    # (9) ESS/ERT masked fine pass (fast_sampling=True on the instance; off by default)
    with torch.no_grad():
        import io, contextlib
        ren.fast_sampling = True
        o, d = seeded_rays(256, 5)
        t_c, pts_c = ren.stratified_sample_points_from_rays(o, d, N_samples=64, perturb=False)
        vd = d / torch.norm(d, dim=-1, keepdim=True)
        raw_c = net.forward(pts_c, vd, None, model="")
        with contextlib.redirect_stdout(io.StringIO()):
            pts_f, t_f, vmask = ren.fine_sample_points(torch.relu(raw_c[..., 3]), o, d, t_c, 128, 64, 0.25)
            rgb_m, dep_m = ren.render({"rays_o": o[None], "rays_d": d[None]})
            prgb_m, pdep_m = ren.render({"rays_o": po[None], "rays_d": pd[None]})
            # the default threshold 0.25 masks out (almost) every fine sample; a second setting keeps some
            ren.weights_threshold = 0.02
            _, _, vmask2 = ren.fine_sample_points(torch.relu(raw_c[..., 3]), o, d, t_c, 128, 64, ren.weights_threshold)
            rgb_m2, dep_m2 = ren.render({"rays_o": o[None], "rays_d": d[None]})
            ren.weights_threshold = 0.25
        ren.fast_sampling = False
        npz("render_masked.npz", rays_o=o, rays_d=d, raw_coarse=raw_c, t_fine=t_f, valid_fine=vmask,
            rgb=rgb_m, depth=dep_m, pin_rays_o=po, pin_rays_d=pd, pin_rgb=prgb_m, pin_depth=pdep_m,
            valid_fine_thr002=vmask2, rgb_thr002=rgb_m2, depth_thr002=dep_m2)

    family_fixtures(net, ren, sd)

    # (8) autograd fixture: MSE on fine RGB, grads of all 48 tensors for a 64-ray step
    net.train()
    o, d = seeded_rays(64, 21)
    gt = torch.rand(64, 3, generator=torch.Generator().manual_seed(22))
    rgb, dep = ren.render({"rays_o": o[None], "rays_d": d[None]})
    loss = torch.nn.functional.mse_loss(rgb, gt)
    net.zero_grad()
    loss.backward()
    grads = {"grad/" + k: p.grad for k, p in net.named_parameters()}
    npz("autograd.npz", rays_o=o, rays_d=d, gt=gt, rgb=rgb, depth=dep, loss=loss, **grads)

    training_fixtures(Network, Renderer)


def family_fixtures(net, ren, base_sd):
    """(10) Parity scenes beyond the benign band-limited field (round-1 VERDICT "Weak 2"): for each weight family of
    oracle.WEIGHT_FAMILIES (exact elementwise transforms of the base checkpoint) the REAL reference renders 512 seeded
    rays and 512 pinhole rays; the merged sample depths are re-derived with the reference's own methods so the GPU
    tests can attribute any per-ray deviation to moved samples (inverse-CDF discontinuities)."""
    ids = torch.from_numpy(np.random.default_rng(1).choice(800 * 800, 512, replace=False))
    sets = {"seed": orc.seeded_rays(512, 31), "pin": orc.pinhole_rays(800, 800, orc.camera_pose(55.0), pixel_ids=ids)}
    # "trained": not a transform but a network TRAINED by the build itself (tools/make_trained_fixture.py: 3000 steps of the
    # reference's training step on the sharp scene, held-out PSNR 27.4 dB) -- Adam-shaped weights, committed as
    # tests/golden/trained_ckpt.pth in the reference's {"net": ...} layout
    trained = torch.load(os.path.join(OUT, "trained_ckpt.pth"), weights_only=True)["net"]
    for fam in list(orc.WEIGHT_FAMILIES) + ["trained"]:
        net.load_state_dict(trained if fam == "trained" else orc.weight_family(base_sd, fam), strict=True)
        rec = {}
        with torch.no_grad():
            for tag, (o, d) in sets.items():
                t_c, pts_c = ren.stratified_sample_points_from_rays(o, d, N_samples=64, perturb=False)
                vd = d / torch.norm(d, dim=-1, keepdim=True)
                raw_c = net.forward(pts_c, vd, None, model="")
                pts_f, t_f, vm = ren.fine_sample_points(torch.relu(raw_c[..., 3]), o, d, t_c, 128, 64, 0.25)
                assert vm is None
                t_sorted, _ = torch.sort(torch.cat([t_c, t_f], 1), dim=-1)
                rgb, dep = ren.render({"rays_o": o[None], "rays_d": d[None]})
                rec.update({f"{tag}_rays_o": o, f"{tag}_rays_d": d, f"{tag}_sigma_coarse_raw": raw_c[..., 3],
                            f"{tag}_t_sorted": t_sorted, f"{tag}_rgb": rgb, f"{tag}_depth": dep})
        npz(f"render_family_{fam}.npz", **rec)
    net.load_state_dict(base_sd, strict=True)


def _subsample(t):
    """Fixture-size control for parameter-shaped tensors: small tensors (biases, heads) in full, the large weight matrices
    as the flat stride-7 subsample (7 is coprime to every row length, so the sample walks all rows and columns)."""
    f = t.detach().reshape(-1)
    return f.clone() if f.numel() <= 4096 else f[::7].clone()


def training_fixtures(Network, Renderer):
    """(11) Multi-step training trajectories from the REAL reference (round-2 VERDICT item 1): K = 5 iterations of the
    reference's step -- Renderer.render under autograd, nn.MSELoss on the fine RGB (trainers/nerf.py:27-33), loss.backward(),
    clip_grad_value_(40), Adam(lr 5e-4, eps 1e-8, weight_decay 0) built by the reference's own make_optimizer
    (trainer.py:53-60, optimizer.py:8-28) -- on ONE fixed batch of 256 pinhole rays, once from the synthetic checkpoint and once
    from the trained one; the targets are the reference's render of the "sharp" teacher scene on the same rays.  Recorded per
    step: loss, the share of coarse samples with sigma > 0, every ray's coarse sigma; after step 1 and step K: the parameters
    (small tensors in full, weight matrices as their flat stride-7 subsample); for step 1 also the raw gradients (same
    subsample) and the rendered image."""
    K = 5
    ids = torch.from_numpy(np.random.default_rng(3).choice(800 * 800, 256, replace=False))
    o, d = orc.pinhole_rays(800, 800, orc.camera_pose(40.0), pixel_ids=ids)
    base = torch.load(os.path.join(OUT, "synthetic_ckpt.pth"), weights_only=True)["net"]
    trained = torch.load(os.path.join(OUT, "trained_ckpt.pth"), weights_only=True)["net"]
    teacher = Network()
    teacher.load_state_dict(orc.weight_family({k: base[k] for k in orc.state_dict_keys()}, "sharp"), strict=True)
    teacher.eval()
    t_ren = Renderer(teacher)
    t_ren.device = torch.device("cpu")
    with torch.no_grad():
        target, _ = t_ren.render({"rays_o": o[None], "rays_d": d[None]})
    try:
        from src.config import cfg as ref_cfg
        from src.train.optimizer import make_optimizer
    except ImportError as exc:          # (an ordinary missing-module error: build the identical optimizer by hand)
        print("reference make_optimizer not importable (%s): building Adam per optimizer.py:8-28 by hand" % exc)
        make_optimizer = None
    for tag, sd0 in (("synthetic", base), ("trained", trained)):
        net = Network()
        net.load_state_dict({k: sd0[k].clone() for k in orc.state_dict_keys()}, strict=True)
        net.train()
        ren = Renderer(net)
        ren.device = torch.device("cpu")
        assert ren.perturb is False
        if make_optimizer is not None:
            opt = make_optimizer(ref_cfg, net)
        else:
            opt = torch.optim.Adam([{"params": [p], "lr": 5e-4, "weight_decay": 0.0, "eps": 1e-8} for p in net.parameters()],
                                   5e-4, weight_decay=0.0, eps=1e-8)
        g0 = opt.param_groups[0]
        assert type(opt) is torch.optim.Adam and g0["lr"] == 5e-4 and g0["eps"] == 1e-8 and g0["weight_decay"] == 0.0 \
            and tuple(g0["betas"]) == (0.9, 0.999), g0
        crit = torch.nn.MSELoss()
        rec = {"rays_o": o, "rays_d": d, "pixel_ids": ids, "target": target, "K": K}
        losses, live_c, sig_c = [], [], []
        for step in range(1, K + 1):
            # the coarse densities of this step (what places the fine samples): re-derived with the reference's own methods
            with torch.no_grad():
                t_c, pts_c = ren.stratified_sample_points_from_rays(o, d, N_samples=64, perturb=False)
                vd = d / torch.norm(d, dim=-1, keepdim=True)
                sraw = net.forward(pts_c, vd, None, model="")[..., 3]
            rgb, dep = ren.render({"rays_o": o[None], "rays_d": d[None]})
            loss = crit(rgb, target)
            opt.zero_grad()
            loss.backward()
            if step == 1:
                rec["rgb_step1"], rec["depth_step1"] = rgb.detach().clone(), dep.detach().clone()
                for k, p in net.named_parameters():
                    rec["grad1/" + k] = _subsample(p.grad)
            torch.nn.utils.clip_grad_value_(net.parameters(), 40)
            opt.step()
            losses.append(loss.detach().clone())
            live_c.append((sraw > 0).float().mean())
            sig_c.append(sraw.clone())
            if step in (1, K):
                for k, p in net.named_parameters():
                    rec[f"param{step}/" + k] = _subsample(p)
            print(f"  [{tag}] step {step}: loss {loss.item():.8f}, coarse sigma>0 {live_c[-1].item():.4f}")
        rec["loss"] = torch.stack(losses)
        rec["coarse_live_fraction"] = torch.stack(live_c)
        rec["sigma_coarse_raw"] = torch.stack(sig_c)          # [K, 256, 64], before each step's update
        npz(f"train_steps_{tag}.npz", **rec)


def training_only():
    """python oracle/gen_golden.py --training : only the multi-step training fixtures, from the COMMITTED checkpoints."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    Network, Renderer = import_reference()
    training_fixtures(Network, Renderer)


def families_only():
    """python oracle/gen_golden.py --families : only the family fixtures, from the COMMITTED base checkpoint."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    Network, Renderer = import_reference()
    ck = torch.load(os.path.join(OUT, "synthetic_ckpt.pth"), weights_only=True)
    sd = {k: ck["net"][k] for k in orc.state_dict_keys()}
    net = Network()
    net.load_state_dict(sd, strict=True)
    net.eval()
    ren = Renderer(net)
    ren.device = torch.device("cpu")
    family_fixtures(net, ren, sd)


if __name__ == "__main__":
    if "--families" in sys.argv:
        sys.argv.remove("--families")
        families_only()
    elif "--training" in sys.argv:
        sys.argv.remove("--training")
        training_only()
    else:
        main()
