"""Developer probe: live-tile counts and loss per step of the bench's training run, per precision / skip mode."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import nerf_replication_amd as pkg
import bench
from nerf_replication_amd.training import train_step, FusedAdam
dev = torch.device("cuda:0")
sd = bench.load_weights()
for precision, env in (("f32", "1"), ("f32x", "1"), ("f32", "0")):
    os.environ["NERF_DEAD_TILE_SKIP"] = env
    net = pkg.Network(); net.load_state_dict(sd); net = net.to(dev).train(); net.precision = precision
    ren = pkg.Renderer(net)
    ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(0))[:4096].to(dev)
    o, d = pkg.generate_rays(bench.camera_pose_40(), 800, 800, 0.6911112070083618, dev, pixel_ids=ids)
    with torch.no_grad():
        net.eval(); rgb0, _ = ren.render({"rays_o": o[None], "rays_d": d[None]}); net.train()
    noise = torch.rand(4096, 3, generator=torch.Generator().manual_seed(1)).to(dev) - 0.5
    colors = (rgb0.reshape(4096, 3).float() + 0.1 * noise).clamp_(0, 1).contiguous()
    opt = FusedAdam(net.parameters(), lr=5e-4, eps=1e-8, clip_value=40.0)
    ren.live_tile_stats = []
    losses = []
    for step in range(23):
        losses.append(train_step(ren, opt, o, d, colors))
    torch.cuda.synchronize()
    cf = [int(a.item()) for a, _, _, _ in ren.live_tile_stats]; cc = [int(c.item()) for _, _, c, _ in ren.live_tile_stats]
    print(precision, "skip" if env == "1" else "dense")
    print("  loss  ", " ".join("%.5f" % l.item() for l in losses[::2]))
    print("  fine  ", cf[::2])
    print("  coarse", cc[::2])
    # sigma statistics of the final coarse / fine nets on these rays
    with torch.no_grad():
        net.eval()
        L = pkg._lib; lib = L.load(); st = L.stream_of(dev)
        t_c, u = ren._get_tables(dev)
        raw_c = torch.empty(4096, 64, 4, device=dev)
        L.check(lib.nerf_mlp_forward_rays(L.ptr(o), L.ptr(d), L.ptr(t_c), 0, 4096, 64, net.packed("").data_ptr(), L.ptr(raw_c), 0, st))
        torch.cuda.synchronize()
        print("  final coarse sigma>0 fraction %.3f" % (raw_c[..., 3] > 0).float().mean().item())
