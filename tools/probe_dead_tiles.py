import os, sys, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nerf_replication_amd as pkg
L = pkg._lib; lib = L.load()
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
dev = torch.device("cuda:0")
for name in ("synthetic", "trained"):
    ck = torch.load(os.path.join(REPO, "tests", "golden", f"{name}_ckpt.pth"), weights_only=True)["net"]
    net = pkg.Network(); net.load_state_dict(ck); net = net.to(dev).eval()
    ren = pkg.Renderer(net)
    n = 4096
    g = torch.Generator().manual_seed(0)
    ids = torch.randperm(800 * 800, generator=g)[:n].to(dev)
    import bench
    o_all, d_all = pkg.generate_rays(bench.camera_pose_40(), 800, 800, 0.6911112070083618, dev, pixel_begin=0, n_pixels=640000)
    o, d = o_all[ids].contiguous(), d_all[ids].contiguous()
    st = L.stream_of(dev)
    t_c, u = ren._get_tables(dev)
    raw_c = torch.empty(n, 64, 4, device=dev); t_sorted = torch.empty(n, 192, device=dev); raw_f = torch.empty(n, 192, 4, device=dev)
    rgb = torch.empty(n, 3, device=dev); dep = torch.empty(n, device=dev)
    L.check(lib.nerf_mlp_forward_rays(L.ptr(o), L.ptr(d), L.ptr(t_c), 0, n, 64, net.packed("").data_ptr(), L.ptr(raw_c), 0, st))
    L.check(lib.nerf_sample_fine(L.ptr(raw_c), L.ptr(t_c), L.ptr(u), n, L.ptr(t_sorted), None, None, 0.0, 0.0, st))
    L.check(lib.nerf_mlp_forward_rays(L.ptr(o), L.ptr(d), L.ptr(t_sorted), 192, n, 192, net.packed("fine").data_ptr(), L.ptr(raw_f), 0, st))
    L.check(lib.nerf_composite(L.ptr(raw_f), L.ptr(t_sorted), 192, n, 192, 1, L.ptr(rgb), L.ptr(dep), None, st))
    target = torch.rand(n, 3, generator=g).to(dev)
    g_rgb = (2.0 / (n * 3)) * (rgb - target)
    g_raw_f = torch.empty(n, 192, 4, device=dev); g_t = torch.empty(n, 192, device=dev)
    L.check(lib.nerf_composite_backward(L.ptr(raw_f), L.ptr(t_sorted), 192, n, 192, 1, L.ptr(g_rgb), None, L.ptr(g_raw_f), L.ptr(g_t), st))
    torch.cuda.synchronize()
    zero_pt = (g_raw_f == 0).all(-1)
    print(name, "fine: points with g_raw == 0: %.3f, 32-tiles all zero: %.3f, 16-groups all zero: %.3f" % (
        zero_pt.float().mean().item(), zero_pt.reshape(-1, 32).all(-1).float().mean().item(), zero_pt.reshape(-1, 16).all(-1).float().mean().item()))
    sig_dead = (raw_f[..., 3] <= 0)
    print("   sigma<=0 points %.3f tiles %.3f ; w==0 exactly beyond opaque: %.3f" % (sig_dead.float().mean().item(), sig_dead.reshape(-1, 32).all(-1).float().mean().item(), ((g_raw_f[..., :3] == 0).all(-1) & ~sig_dead).float().mean().item()))
    # coarse: gradient reaches sigma_c only where sigma_c > 0
    cdead = raw_c[..., 3] <= 0
    print("   coarse sigma<=0 points %.3f, 32-tiles %.3f" % (cdead.float().mean().item(), cdead.reshape(-1, 32).all(-1).float().mean().item()))
