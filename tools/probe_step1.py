"""Developer probe: gradients of the first bench-like training step, f32 vs f32x, dead-tile skipping on / off."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import nerf_replication_amd as pkg
import bench
from nerf_replication_amd.training import render_with_grad
dev = torch.device("cuda:0")
sd = bench.load_weights()
n = 4096
res = {}
names = None
for precision, env in (("f32", "0"), ("f32x", "0"), ("f32", "1"), ("f32x", "1")):
    os.environ["NERF_DEAD_TILE_SKIP"] = env
    net = pkg.Network(); net.load_state_dict(sd); net = net.to(dev).train(); net.precision = precision
    ren = pkg.Renderer(net)
    ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(0))[:n].to(dev)
    o, d = pkg.generate_rays(bench.camera_pose_40(), 800, 800, 0.6911112070083618, dev, pixel_ids=ids)
    if "colors" not in globals():
        with torch.no_grad():
            net.eval(); net.precision = "f32"; rgb0, _ = ren.render({"rays_o": o[None], "rays_d": d[None]}); net.train(); net.precision = precision
        noise = torch.rand(n, 3, generator=torch.Generator().manual_seed(1)).to(dev) - 0.5
        colors = (rgb0.reshape(n, 3).float() + 0.1 * noise).clamp_(0, 1).contiguous()
    rgb, dep = render_with_grad(ren, o, d)
    loss = torch.nn.functional.mse_loss(rgb, colors)
    loss.backward()
    torch.cuda.synchronize()
    names = [k for k, _ in net.named_parameters()]
    res[(precision, env)] = [p.grad.clone() for p in net.parameters()]
    print(precision, env, "loss %.8f" % loss.item())
ref = res[("f32", "0")]
for key in (("f32x", "0"), ("f32", "1"), ("f32x", "1")):
    print("vs f32 dense:", key)
    for nm, a, b in zip(names, ref, res[key]):
        den = a.abs().max().item()
        diff = (a - b).abs().max().item()
        sign_flips = ((a * b) < 0).float().mean().item()
        big = ((a.abs() > 1e-8) & ((a * b) <= 0)).float().mean().item()
        if den == 0 and diff == 0:
            continue
        print("   %-34s max|g| %.3e  max diff %.3e  rel %.2e  sign flips %.4f (with |g|>1e-8: %.4f)" % (nm, den, diff, diff / max(den, 1e-30), sign_flips, big))
