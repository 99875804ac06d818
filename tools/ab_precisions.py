#!/usr/bin/env python3
"""A/B of arithmetic paths / MFMA shapes INSIDE one library, interleaved rounds in one process (cdna_hip_programming.md rule 24).

    python tools/ab_precisions.py f16m32 f16            # fine launch (160 000 rays x 192 samples) of each, AB_ROUNDS rounds
    python tools/ab_precisions.py f32x f32xs

Reports per precision the median / min launch time (HIP events on the launch stream) and the algorithmic TFLOP/s; the data
is the bench frame's (random, scene-like), never zeros (rule 25).  AB_KIND=coarse times the density-only coarse launch instead."""
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import nerf_replication_amd as pkg  # noqa: E402


def main():
    names = sys.argv[1:] or ["f16m32", "f16"]
    rounds = int(os.environ.get("AB_ROUNDS", "7"))
    kind = os.environ.get("AB_KIND", "fine")
    n = int(os.environ.get("AB_RAYS", "160000"))
    L, lib = pkg._lib, pkg._lib.load()
    dev = torch.device("cuda", 0)
    sd = bench.load_weights()
    nets = {}
    for p in names:
        net = pkg.Network(); net.load_state_dict(sd, strict=True); net = net.to(dev).eval(); net.precision = p
        nets[p] = net
    ren = pkg.Renderer(nets[names[0]])
    o, d = pkg.generate_rays(bench.camera_pose_40(), 800, 800, 0.6911112070083618, dev, pixel_begin=240000, n_pixels=n)
    t_c, u = ren._get_tables(dev)
    st = L.stream_of(dev)
    # sample positions from the exact path, shared by every variant
    ref = pkg.Network(); ref.load_state_dict(sd, strict=True); ref = ref.to(dev).eval()
    raw_c = torch.empty(n, 64, 4, device=dev); t_sorted = torch.empty(n, 192, device=dev)
    L.check(lib.nerf_mlp_forward_rays_density(L.ptr(o), L.ptr(d), L.ptr(t_c), 0, n, 64, ref.packed("").data_ptr(), L.ptr(raw_c), 0, st))
    L.check(lib.nerf_sample_fine(L.ptr(raw_c), L.ptr(t_c), L.ptr(u), n, L.ptr(t_sorted), None, None, 0.0, 0.0, st))
    raw = torch.empty(n, 192, 4, device=dev)

    def launch(p):
        prec = L.PRECISIONS[p]
        if kind == "coarse":
            L.check(lib.nerf_mlp_forward_rays_density(L.ptr(o), L.ptr(d), L.ptr(t_c), 0, n, 64, nets[p].packed("").data_ptr(), L.ptr(raw_c), prec, st))
        elif kind == "full":
            L.check(lib.nerf_mlp_forward_rays(L.ptr(o), L.ptr(d), L.ptr(t_sorted), 192, n, 192, nets[p].packed("fine").data_ptr(), L.ptr(raw), prec, st))
        else:
            L.check(lib.nerf_mlp_forward_rays_for_compositing(L.ptr(o), L.ptr(d), L.ptr(t_sorted), 192, n, 192, nets[p].packed("fine").data_ptr(), L.ptr(raw), prec, st))

    for p in names:                       # warm-up (packing, code load)
        launch(p)
    torch.cuda.synchronize()
    times = {p: [] for p in names}
    for _ in range(rounds):
        for p in names:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); launch(p); b.record(); torch.cuda.synchronize()
            times[p].append(a.elapsed_time(b))
    pts = n * (64 if kind == "coarse" else 192)
    flop = pts * (bench.FLOP_PER_POINT - (bench.FLOP_DENSITY_SKIPPED if kind == "coarse" else 0))
    for p in names:
        med, mn = statistics.median(times[p]), min(times[p])
        print(f"{p:6s} {kind}: median {med:8.3f} ms  min {mn:8.3f} ms  -> {flop / (med * 1e-3) / 1e12:7.1f} TFLOP/s (all colours counted)   "
              f"rounds {[round(t, 2) for t in times[p]]}", flush=True)


if __name__ == "__main__":
    main()
