"""Developer probe: coarse-network gradients of one bench-like training step, dead-tile skipping on vs off."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import nerf_replication_amd as pkg
import bench
from nerf_replication_amd.training import render_with_grad
dev = torch.device("cuda:0")
sd = bench.load_weights()
N = int(os.environ.get("PROBE_RAYS", "4096"))
res = {}
for env in ("0", "1"):
    os.environ["NERF_DEAD_TILE_SKIP"] = env
    net = pkg.Network(); net.load_state_dict(sd); net = net.to(dev).train(); net.precision = "f32"
    ren = pkg.Renderer(net)
    ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(0))[:N].to(dev)
    o, d = pkg.generate_rays(bench.camera_pose_40(), 800, 800, 0.6911112070083618, dev, pixel_ids=ids)
    with torch.no_grad():
        net.eval(); rgb0, _ = ren.render({"rays_o": o[None], "rays_d": d[None]}); net.train()
    noise = torch.rand(N, 3, generator=torch.Generator().manual_seed(1)).to(dev) - 0.5
    colors = (rgb0.reshape(N, 3) + 0.1 * noise).clamp_(0, 1)
    ren.live_tile_stats = []
    rgb, dep = render_with_grad(ren, o, d)
    loss = torch.nn.functional.mse_loss(rgb, colors)
    loss.backward()
    torch.cuda.synchronize()
    gc = [p.grad.abs().max().item() for p in net.model.ordered_params()]
    gf = [p.grad.abs().max().item() for p in net.model_fine.ordered_params()]
    st = ren.live_tile_stats[0]
    print("skip" if env == "1" else "dense", "loss %.6f" % loss.item(), "coarse max|g| %.3e" % max(gc), "fine max|g| %.3e" % max(gf),
          "live fine %d/%d coarse %d/%d" % (int(st[0].item()), st[1], int(st[2].item()), st[3]))
    res[env] = ([p.grad.clone() for p in net.model.ordered_params()], [p.grad.clone() for p in net.model_fine.ordered_params()])
for which, idx in (("coarse", 0), ("fine", 1)):
    worst = 0.0
    for a, b in zip(res["0"][idx], res["1"][idx]):
        den = a.abs().max().clamp_min(1e-20)
        worst = max(worst, ((a - b).abs().max() / den).item())
    print(which, "max rel diff dense vs skip: %.3e" % worst)
