"""Developer probe: after a few training steps, the live-tile sets the f32 and f32x pipelines see for the SAME network."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import nerf_replication_amd as pkg
import bench
from nerf_replication_amd.training import train_step, FusedAdam
dev = torch.device("cuda:0")
sd = bench.load_weights()
net = pkg.Network(); net.load_state_dict(sd); net = net.to(dev).train(); net.precision = os.environ.get("PROBE_TRAIN_PREC", "f32")
ren = pkg.Renderer(net)
n = 4096
ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(0))[:n].to(dev)
o, d = pkg.generate_rays(bench.camera_pose_40(), 800, 800, 0.6911112070083618, dev, pixel_ids=ids)
with torch.no_grad():
    net.eval(); rgb0, _ = ren.render({"rays_o": o[None], "rays_d": d[None]}); net.train()
noise = torch.rand(n, 3, generator=torch.Generator().manual_seed(1)).to(dev) - 0.5
colors = (rgb0.reshape(n, 3).float() + 0.1 * noise).clamp_(0, 1).contiguous()
opt = FusedAdam(net.parameters(), lr=5e-4, eps=1e-8, clip_value=40.0)
for step in range(4):
    train_step(ren, opt, o, d, colors)
torch.cuda.synchronize()
L = pkg._lib; lib = L.load(); st = L.stream_of(dev)
t_c, u = ren._get_tables(dev)
for precision in ("f32", "f32x"):
    net.precision = precision
    prec = L.PRECISIONS[precision]
    raw_c = torch.empty(n, 64, 4, device=dev); save_c = torch.empty(int(lib.nerf_train_save_floats(n * 64)), device=dev)
    L.check(lib.nerf_mlp_forward_rays_save_density(L.ptr(o), L.ptr(d), L.ptr(t_c), 0, n, 64, net.packed("").data_ptr(), L.ptr(raw_c), L.ptr(save_c), prec, st))
    t_sorted = torch.empty(n, 192, device=dev)
    L.check(lib.nerf_sample_fine(L.ptr(raw_c), L.ptr(t_c), L.ptr(u), n, L.ptr(t_sorted), None, None, 0.0, 0.0, st))
    raw_f = torch.empty(n, 192, 4, device=dev); save_f = torch.empty(int(lib.nerf_train_save_floats(n * 192)), device=dev)
    L.check(lib.nerf_mlp_forward_rays_save(L.ptr(o), L.ptr(d), L.ptr(t_sorted), 192, n, 192, net.packed("fine").data_ptr(), L.ptr(raw_f), L.ptr(save_f), prec, st))
    rgb = torch.empty(n, 3, device=dev); dep = torch.empty(n, device=dev)
    L.check(lib.nerf_composite(L.ptr(raw_f), L.ptr(t_sorted), 192, n, 192, 1, L.ptr(rgb), L.ptr(dep), None, st))
    g_rgb = (2.0 / (n * 3)) * (rgb - colors)
    g_raw_f = torch.empty(n, 192, 4, device=dev); g_t = torch.empty(n, 192, device=dev)
    L.check(lib.nerf_composite_backward(L.ptr(raw_f), L.ptr(t_sorted), 192, n, 192, 1, L.ptr(g_rgb), None, L.ptr(g_raw_f), L.ptr(g_t), st))
    torch.cuda.synchronize()
    live_f = (g_raw_f != 0).any(-1).reshape(-1, 32).any(-1)
    print(precision, "coarse sigma>0 frac %.4f" % (raw_c[..., 3] > 0).float().mean().item(),
          "fine sigma>0 frac %.4f" % (raw_f[..., 3] > 0).float().mean().item(), "live fine tiles", int(live_f.sum()),
          "NaN in raw_f:", bool(torch.isnan(raw_f).any()), "NaN in g_raw_f:", bool(torch.isnan(g_raw_f).any()),
          "NaN g_t", bool(torch.isnan(g_t).any()))
    sig_live = (raw_f[..., 3] > 0).reshape(-1, 32).any(-1)
    print("    tiles with some sigma>0:", int(sig_live.sum()), " live but no sigma>0:", int((live_f & ~sig_live).sum()))
    g_raw_c = torch.empty(n, 64, 4, device=dev)
    L.check(lib.nerf_sample_fine_backward(L.ptr(raw_c), L.ptr(t_c), L.ptr(u), n, L.ptr(t_sorted), L.ptr(g_t), L.ptr(g_raw_c), st))
    torch.cuda.synchronize()
    print("    coarse live tiles (w != 0):", int((g_raw_c[..., 3] != 0).reshape(-1, 32).any(-1).sum()), " any channel:",
          int((g_raw_c != 0).any(-1).reshape(-1, 32).any(-1).sum()), "NaN:", bool(torch.isnan(g_raw_c).any()))
