"""Developer probe: backward chain + weight gradients for a density-only incoming gradient, f32 vs f32x, vs torch autograd."""
import os, sys, ctypes, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import nerf_replication_amd as pkg
import nerf_oracle as orc
os.environ["NERF_DEAD_TILE_SKIP"] = "0"
L = pkg._lib; lib = L.load()
dev = torch.device("cuda:0")
sd = torch.load(os.path.join(REPO, "tests", "golden", "synthetic_ckpt.pth"), weights_only=True)["net"]
gen = torch.Generator().manual_seed(3)
n, S = 64, 64
o = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3).contiguous()
d = torch.randn(n, 3, generator=gen) * 0.2 + torch.tensor([0.0, 0.0, -1.0]); d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
t = torch.linspace(2.0, 6.0, S)
for tag in ("sigma only", "rgb only", "all four", "sigma only, positive", "sigma only, n=41"):
    if tag.endswith("n=41"):
        n = 41
        o = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3).contiguous()
        d = torch.randn(n, 3, generator=gen) * 0.2 + torch.tensor([0.0, 0.0, -1.0]); d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
    G = torch.randn(n, S, 4, generator=gen) * 1e-3
    if tag.startswith("sigma only"):
        G[..., :3] = 0.0
    if tag == "rgb only":
        G[..., 3] = 0.0
    if tag.endswith("positive"):
        G[..., 3] = G[..., 3].abs()
    # autograd reference (torch fp32 on the CPU) on the oracle's MLP
    sdr = {k: v.clone().requires_grad_(k.startswith("model.")) for k, v in sd.items()}
    pts = (o[:, None, :] + d[:, None, :] * t[None, :, None])
    raw = orc.network_forward(sdr, pts, d, model="")
    (raw * G).sum().backward()
    ref = {k: v.grad.double() for k, v in sdr.items() if k.startswith("model.")}
    for precision in ("f32", "f32x"):
        net = pkg.Network(); net.load_state_dict(sd); net = net.to(dev).eval(); net.precision = precision
        prec = L.PRECISIONS[precision]
        params = [p.detach().contiguous() for p in net.model.ordered_params()]
        arr = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in params])
        st = L.stream_of(dev)
        pk_b = torch.empty(int(lib.nerf_packed_bwd_bytes(prec)), dtype=torch.uint8, device=dev)
        L.check(lib.nerf_pack_model_bwd(arr, pk_b.data_ptr(), prec, st))
        P = n * S
        od, dd, td, Gd = o.to(dev), d.to(dev), t.to(dev), G.to(dev).contiguous()
        rawd = torch.empty(n, S, 4, device=dev); save = torch.empty(int(lib.nerf_train_save_floats(P)), device=dev)
        gsave = torch.empty(int(lib.nerf_train_grad_floats(P)), device=dev); g_t = torch.empty(n, S, device=dev)
        grads = [torch.zeros_like(p) for p in params]
        garr = (ctypes.c_void_p * 24)(*[g.data_ptr() for g in grads])
        L.check(lib.nerf_mlp_forward_rays_save(L.ptr(od), L.ptr(dd), L.ptr(td), 0, n, S, net.packed("").data_ptr(), L.ptr(rawd), L.ptr(save), prec, st))
        L.check(lib.nerf_mlp_backward(L.ptr(od), L.ptr(dd), L.ptr(td), 0, n, S, pk_b.data_ptr(), L.ptr(Gd), L.ptr(save), L.ptr(gsave), L.ptr(g_t), garr, prec, st))
        torch.cuda.synchronize()
        errs = []
        for name, g in zip(orc.SUBMODEL_KEYS, grads):
            r = ref["model." + name]
            if r.abs().max() == 0: continue
            errs.append((((g.double().cpu() - r).abs().max() / r.abs().max()).item(), name))
        errs.sort(reverse=True)
        print(f"[{tag}] {precision}: " + ", ".join(f"{nm} {e:.1e}" for e, nm in errs[:4]))
