#!/usr/bin/env python3
"""Interleaved in-process A/B timing of build variants of the TRAINING kernels (companion of tools/ab_bench.py).

    python tools/ab_train.py f32 "base:" "nosave:-DNERF_F32_HACK_NOSAVE=1" "nomask:-DNERF_BWD_HACK_NOMASK=1" ...
    python tools/ab_train.py f32x "base:" "nostore:-DNERF_F32X_HACK_SAVE_NOSTORE=1 -DNERF_XB_HACK_NOSTORE=1" "direct:-DNERF_F32X_STAGE_TRUNK=0 -DNERF_F32X_STAGE_DENS=0"

Per variant and round: nerf_mlp_forward_rays_save (fine model, 4096 x 192 points) and nerf_mlp_backward (data-gradient
chain + all weight-gradient launches) on the same inputs, HIP events around each call.  Developer tool."""
import ctypes
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import ab_bench  # noqa: E402  (build(), VDIR)


def main():
    prec_name = sys.argv[1]
    specs = [s.split(":", 1) for s in sys.argv[2:]]
    libs = {}
    for name, extra in specs:
        libs[name] = ab_bench.ensure(name, extra, force=os.environ.get("AB_BUILD_ONLY") == "1")
    if os.environ.get("AB_BUILD_ONLY") == "1":
        print("built", list(libs))
        return
    import nerf_replication_amd as pkg
    L = pkg._lib
    prec = L.PRECISIONS[prec_name]
    n_rays, S = int(os.environ.get("AB_RAYS", "4096")), 192
    P = n_rays * S
    dev = torch.device("cuda:0")
    ck = torch.load(os.path.join(REPO, "tests", "golden", "synthetic_ckpt.pth"), weights_only=True)["net"]
    net = pkg.Network(); net.load_state_dict(ck); net = net.to(dev).eval(); net.precision = prec_name
    g = torch.Generator(device="cpu").manual_seed(0)
    d = torch.randn(n_rays, 3, generator=g); d = (d / d.norm(dim=-1, keepdim=True)).to(dev)
    o = (torch.randn(n_rays, 3, generator=g) * 0.1 + torch.tensor([0., 0., 4.])).to(dev)
    t = torch.sort(torch.rand(n_rays, S, generator=g) * 4 + 2, dim=-1).values.to(dev).contiguous()
    draw = (torch.randn(n_rays, S, 4, generator=g) * 1e-4).to(dev)
    raw = torch.empty(n_rays, S, 4, device=dev)
    g_t = torch.empty(n_rays, S, device=dev)
    st = L.stream_of(dev)
    params = [p.detach().contiguous() for p in net.model_fine.ordered_params()]
    arr = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in params])
    grads = [torch.zeros_like(p) for p in params]
    garr = (ctypes.c_void_p * 24)(*[x.data_ptr() for x in grads])
    handles = {}
    for name, path in libs.items():
        lib = ctypes.CDLL(path)
        for fn, (res, args) in L._PROTOS.items():
            f = getattr(lib, fn); f.restype, f.argtypes = res, args
        pk = torch.empty(lib.nerf_packed_model_bytes(prec), dtype=torch.uint8, device=dev)
        pkb = torch.empty(lib.nerf_packed_bwd_bytes(prec), dtype=torch.uint8, device=dev)
        assert lib.nerf_pack_model(arr, pk.data_ptr(), prec, st) == 0
        assert lib.nerf_pack_model_bwd(arr, pkb.data_ptr(), prec, st) == 0
        save = torch.zeros(int(lib.nerf_train_save_floats(P)), device=dev)
        gsave = torch.zeros(int(lib.nerf_train_grad_floats(P)), device=dev)
        handles[name] = (lib, pk, pkb, save, gsave)
    times = {k: ([], []) for k in handles}
    rounds = int(os.environ.get("AB_ROUNDS", "7"))
    for rnd in range(rounds + 1):
        for name, (lib, pk, pkb, save, gsave) in handles.items():
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            rc = lib.nerf_mlp_forward_rays_save(o.data_ptr(), d.data_ptr(), t.data_ptr(), S, n_rays, S, pk.data_ptr(),
                                                raw.data_ptr(), save.data_ptr(), prec, st)
            e[1].record()
            rc2 = lib.nerf_mlp_backward(o.data_ptr(), d.data_ptr(), t.data_ptr(), S, n_rays, S, pkb.data_ptr(), draw.data_ptr(),
                                        save.data_ptr(), gsave.data_ptr(), g_t.data_ptr(), garr, prec, st)
            e[2].record(); torch.cuda.synchronize()
            assert rc == 0 and rc2 == 0, lib.nerf_last_error()
            if rnd:
                times[name][0].append(e[0].elapsed_time(e[1]))
                times[name][1].append(e[1].elapsed_time(e[2]))
    flop = P * 1186816
    for name, (tf, tb) in times.items():
        mf, mb = statistics.median(tf), statistics.median(tb)
        print(f"{name:>16}: save-forward {mf:7.3f} ms ({flop / mf / 1e9:6.1f} TFLOP/s)   backward(chain+wgrad) {mb:7.3f} ms (min {min(tb):7.3f})")


if __name__ == "__main__":
    main()
