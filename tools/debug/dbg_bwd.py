import sys, ctypes, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import nerf_replication_amd as amd, nerf_oracle as oracle
L = amd._lib; lib = L.load()
sd0 = torch.load("tests/golden/synthetic_ckpt.pth", weights_only=True)["net"]
net = amd.Network().cuda(); net.load_state_dict(sd0)
n, S = 37, 5; gen = torch.Generator().manual_seed(7)
o = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3).contiguous()
d = torch.randn(n, 3, generator=gen) * 0.2 + torch.tensor([0.0, 0.0, -1.0]); d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
t = (torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, dim=-1).values).contiguous()
G = torch.randn(n, S, 4, generator=gen)
sub = net.model_fine; params = [p.detach().contiguous() for p in sub.ordered_params()]
arr = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in params]); st = L.stream_of(params[0].device)
P = n * S
od, dd, td, Gd = o.cuda(), d.cuda(), t.cuda(), G.cuda().contiguous()
raw = torch.empty(n, S, 4, device="cuda"); save = torch.empty(int(lib.nerf_train_save_floats(P)), device="cuda")
L.check(lib.nerf_mlp_forward_rays_save(L.ptr(od), L.ptr(dd), L.ptr(td), S, n, S, net.packed("fine").data_ptr(), L.ptr(raw), L.ptr(save), 0, st))
res = {}
for prec in (0, 2):
    pk_b = torch.zeros(int(lib.nerf_packed_bwd_bytes(prec)), dtype=torch.uint8, device="cuda")
    L.check(lib.nerf_pack_model_bwd(arr, pk_b.data_ptr(), prec, st))
    gsave = torch.zeros(int(lib.nerf_train_grad_floats(P)), device="cuda"); g_t = torch.zeros(n, S, device="cuda")
    grads = [torch.zeros_like(p) for p in params]
    ga = (ctypes.c_void_p * 24)(*[g.data_ptr() for g in grads])
    L.check(lib.nerf_mlp_backward(L.ptr(od), L.ptr(dd), L.ptr(td), S, n, S, pk_b.data_ptr(), L.ptr(Gd), L.ptr(save), L.ptr(gsave), L.ptr(g_t), ga, prec, st))
    torch.cuda.synchronize(); res[prec] = (gsave.cpu(), g_t.cpu())
a, b = res[0][0], res[2][0]
def rel(x, y): return float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))
print("gzv", rel(b[:P*128], a[:P*128])); print("gf", rel(b[P*128:P*384], a[P*128:P*384]))
for l in range(7, -1, -1):
    off = P * (384 + 256 * (7 - l)); x, y = b[off:off+P*256].view(P, 256), a[off:off+P*256].view(P, 256)
    print("gz", l, rel(x, y), "per-tile", [round(rel(x[:, 32*m:32*m+32], y[:, 32*m:32*m+32]), 3) for m in range(8)])
print("g_t", rel(res[2][1], res[0][1]))
gf = b[P*128:P*384].view(P, 256); Wf = params[18].cpu(); wa = params[20].cpu().view(256); gs = G.view(P, 4)[:, 3:4]
h7 = save.cpu()[P*(96+256*7):P*(96+256*8)].view(P, 256)
got = b[P*384:P*640].view(P, 256)
full = gf @ Wf + gs * wa[None]
for name, cand in (("full*mask", full * (h7 > 0)), ("noextra*mask", (gf @ Wf) * (h7 > 0)), ("full nomask", full), ("gemm only", gf @ Wf), ("extra only*mask", gs * wa[None] * (h7 > 0))):
    print(name, rel(got, cand))
torch.set_printoptions(precision=4, linewidth=200)
c = full * (h7 > 0)
print("got ", got[0, :16]); print("cand", c[0, :16]); print("gemm", (gf @ Wf)[0, :16]); print("extra", (gs * wa[None])[0, :16]); print("h7", h7[0,:16])
print("got p1", got[1, :16]); print("cand p1", c[1, :16])
print("got tile1", got[0, 32:48]); print("cand", c[0, 32:48])
