#!/usr/bin/env python3
"""Developer probe (GPU box): accuracy of nerf_sample_fine_backward on the multi-step fixture batch, against torch autograd of the CPU
oracle in fp32 and float64, for a random upstream gradient, the real one of a training step, and the real one scaled by 2^20."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "tests"))
import nerf_oracle as orc
import train_steps_common as T
import nerf_replication_amd as pkg
from nerf_replication_amd.training import train_step, FusedAdam

def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
    L, lib = pkg._lib, pkg._lib.load()
    with np.load(os.path.join(REPO, "tests", "golden", f"train_steps_{tag}.npz")) as z:
        g = {k: torch.from_numpy(z[k]) for k in z.files}
    ck = torch.load(os.path.join(REPO, "tests", "golden", f"{tag}_ckpt.pth"), weights_only=True)["net"]
    net = pkg.Network(); net.load_state_dict(ck, strict=True); net = net.cuda().train()
    ren = pkg.Renderer(net); cap = {}; ren.capture_adjoints = cap
    opt = FusedAdam(net.parameters(), lr=0.0)
    train_step(ren, opt, g["rays_o"].cuda(), g["rays_d"].cuda(), g["target"].cuda())
    raw_c = cap["raw_coarse"]; ts = cap["t_sorted"]; G_real = cap["g_t_sorted"]
    n = raw_c.shape[0]
    tcd, ud = torch.linspace(2.0, 6.0, 64).cuda(), torch.linspace(0.0, 1.0, 128).cuda()
    gen = torch.Generator().manual_seed(1)
    cases = {"random": torch.randn(n, 192, generator=gen).cuda(), "real": G_real.clone(), "real_x2^20": G_real * 1048576.0,
             "real_fine_slots_only": G_real.clone()}
    st = L.stream_of(raw_c.device)
    for name, G in cases.items():
        G = G.contiguous()
        out = torch.zeros(n, 64, 4, device="cuda")
        L.check(lib.nerf_sample_fine_backward(L.ptr(raw_c), L.ptr(tcd), L.ptr(ud), n, L.ptr(ts), L.ptr(G), L.ptr(out), st))
        torch.cuda.synchronize()
        g32, b32, a32 = T.sampler_adjoint(orc, raw_c.cpu(), G.cpu())
        g64, b64, a64 = T.sampler_adjoint(orc, raw_c.cpu(), G.cpu(), torch.float64)
        same = ((b32 == b64) & (a32 == a64)).all(1) & (g64.abs().amax(1) > 0)
        sc = g64.abs().amax(1).clamp_min(1e-300)
        eh = ((out[..., 3].cpu().double() - g64).abs().amax(1) / sc)[same]
        ec = ((g32.double() - g64).abs().amax(1) / sc)[same]
        q = lambda e: [float(torch.quantile(e, x)) for x in (0.5, 0.9, 0.99, 1.0)]
        print(f"{name:22s} rays {int(same.sum())}: hip {['%.2e' % v for v in q(eh)]}  torch32 {['%.2e' % v for v in q(ec)]}", flush=True)
        if name == "real":
            worst = torch.argsort(eh, descending=True)[:5]
            idx = same.nonzero()[:, 0][worst]
            cond = T.sampler_conditioning(orc, raw_c.cpu()[..., 3])
            # t_fine of the HIP forward vs the oracle's on the same densities
            tf = torch.empty(n, 128, device="cuda"); ts2 = torch.empty(n, 192, device="cuda")
            L.check(lib.nerf_sample_fine(L.ptr(raw_c), L.ptr(tcd), L.ptr(ud), n, L.ptr(ts2), L.ptr(tf), None, 0.0, 0.0, st))
            tf_o = orc.fine_sample(torch.relu(raw_c.cpu()[..., 3]), orc.stratified_t().expand(n, 64))
            dt = (tf.cpu() - tf_o).abs().amax(1)
            for i in idx.tolist():
                print(f"   ray {i}: hip err {eh[(same.nonzero()[:,0]==i).nonzero()[0,0]]:.2e} min_live_denom {cond['min_live_denom'][i]:.2e} dead {int(cond['n_dead_denoms'][i])} "
                      f"max|t_fine hip - oracle| {dt[i]:.2e} max|G| {G[i].abs().max():.2e} max|g64| {sc[i]:.2e}")
            print("   median max|t_fine hip - oracle| over rays:", float(dt.median()), " max:", float(dt.max()))

if __name__ == "__main__":
    main()
