#!/usr/bin/env python3
"""bench.py -- rays/sec of the NeRF volume-rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1, one rank per GPU)

A "step" is one pass of the hot path over one synthetic 800x800 frame (BASELINE.json configs[1]:
640 000 rays, 64 coarse + 128 fine samples), rays already resident in HBM.  With N>1 the frame's
rays are sharded over the ranks (contiguous tiles) and one RCCL all_gather reassembles it: strong
scaling, as BASELINE.json's north_star asks.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel (nerf_mlp_f32_kernel: both launches of a frame), algorithmic FLOP
                (1 186 816 per MLP point) / HIP-event time, vs 157.3 TFLOP/s fp32 MFMA
  cpu_baseline  the CPU oracle (port of the reference, its 512-point chunking) timed on this host
                on a bounded sample of the same workload (rank 0, N=1 only); cpu_baseline_config1 is
                BASELINE.json configs[0] (1024 rays, 64 coarse samples only) on the same host
  training      BASELINE.json configs[2]: 4096 rays/iter per GPU, fused fwd+bwd HIP MLP + clip + Adam, f32 and f32x
  config5       BASELINE.json configs[4]: one 1600x1600 frame, fp16 activations / fp32 accumulate, sharded like the headline
  world_size / per_rank_compute_ms   what torch.distributed really saw, and each rank's own render time per step
The headline value/metric/dtype are those of configs[1] only; the extra blocks run AFTER its timed region.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_POINT = 1186816                 # 2 x 593 408 MAC, unpadded
FLOP_DENSITY_SKIPPED = 204288            # feature_linear + views_linears.0 + rgb_linear: not evaluated by a density-only pass
POINTS_PER_RAY = 64 + 192
PEAK_F32_MFMA = 157.3e12                 # MI355X_MICROARCH.md chip table
PEAK_F16_MFMA = 2.5e15                   # dense fp16/bf16 MFMA, same table
H = W = 800


def camera_pose_40():
    """Blender-style camera-to-world (radius 4.0311, azimuth 40 deg, elevation 30 deg) -- the same pose
    the parity tests render (oracle.camera_pose(40.0)); float64, host side."""
    import math
    th, ph, rad = math.radians(40.0), math.radians(-30.0), 4.031128874
    rot_phi = torch.tensor([[1, 0, 0], [0, math.cos(ph), -math.sin(ph)], [0, math.sin(ph), math.cos(ph)]], dtype=torch.float64)
    rot_th = torch.tensor([[math.cos(th), 0, -math.sin(th)], [0, 1, 0], [math.sin(th), 0, math.cos(th)]], dtype=torch.float64)
    flip = torch.tensor([[-1.0, 0, 0], [0, 0, 1], [0, 1, 0]], dtype=torch.float64)
    R = flip @ rot_th @ rot_phi
    c2w = torch.zeros(3, 4, dtype=torch.float64)
    c2w[:, :3] = R
    c2w[:, 3] = R @ torch.tensor([0.0, 0.0, rad], dtype=torch.float64)
    return c2w


def load_weights(name="synthetic_ckpt.pth"):
    ck = torch.load(os.path.join(REPO, "tests", "golden", name), weights_only=True)
    return ck["net"]


def sharp_teacher(sd):
    """The "sharp" scene family of the parity tests (oracle.WEIGHT_FAMILIES, restated here: the product side never imports the
    oracle): density head of the synthetic checkpoint x 7.5, bias x 7.5 - 30 -- hard surfaces, ~5 % of the samples occupied.
    tests/golden/trained_ckpt.pth is a network this build trained against renders of exactly this scene."""
    out = {k: v.clone() for k, v in sd.items()}
    for m in ("model", "model_fine"):
        out[f"{m}.alpha_linear.weight"] = out[f"{m}.alpha_linear.weight"] * 7.5
        out[f"{m}.alpha_linear.bias"] = out[f"{m}.alpha_linear.bias"] * 7.5 - 30.0
    return out


def time_stages(pkg, net, ren, o, d, steps, prec=0, full_coarse=False):
    """HIP-event time of each kernel of the render path, launched through the C ABI on torch's
    current stream (the stream the events are recorded on)."""
    L, lib = pkg._lib, pkg._lib.load()
    n, dev = o.shape[0], o.device
    t_c, u = ren._get_tables(dev)
    pk_c, pk_f = net.packed(""), net.packed("fine")
    raw_c = torch.empty(n, 64, 4, device=dev)
    t_sorted = torch.empty(n, 192, device=dev)
    raw_f = torch.empty(n, 192, 4, device=dev)
    rgb, dep = torch.empty(n, 3, device=dev), torch.empty(n, device=dev)
    raw_f_full = None
    st = L.stream_of(dev)
    # mlp_coarse is the launch nerf_render_forward makes (density-only: the reference reads nothing but sigma of the coarse
    # output when N_importance > 0, volume_renderer.py:335); with --compare-full-coarse, mlp_coarse_full_network times the full
    # coarse network beside it (off by default: its launches would mix into the fine launches' row of `rocprofv3 --stats`)
    names = ["mlp_coarse_full_network", "mlp_coarse", "sample_fine", "mlp_fine", "composite", "mlp_fine_full_network"]
    acc = dict.fromkeys(names, 0.0)
    for _ in range(steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        ev[0].record()
        if full_coarse:
            L.check(lib.nerf_mlp_forward_rays(L.ptr(o), L.ptr(d), L.ptr(t_c), 0, n, 64, pk_c.data_ptr(), L.ptr(raw_c), prec, st))
        ev[1].record()
        L.check(lib.nerf_mlp_forward_rays_density(L.ptr(o), L.ptr(d), L.ptr(t_c), 0, n, 64, pk_c.data_ptr(), L.ptr(raw_c), prec, st))
        ev[2].record()
        L.check(lib.nerf_sample_fine(L.ptr(raw_c), L.ptr(t_c), L.ptr(u), n, L.ptr(t_sorted), None, None, 0.0, 0.0, st))
        ev[3].record()
        L.check(lib.nerf_mlp_forward_rays_for_compositing(L.ptr(o), L.ptr(d), L.ptr(t_sorted), 192, n, 192, pk_f.data_ptr(), L.ptr(raw_f), prec, st))
        ev[4].record()
        L.check(lib.nerf_composite(L.ptr(raw_f), L.ptr(t_sorted), 192, n, 192, 1, L.ptr(rgb), L.ptr(dep), None, st))
        ev[5].record()
        if full_coarse:        # the fine network with every colour computed (nerf_mlp_forward_rays), into its own buffer
            if raw_f_full is None:
                raw_f_full = torch.empty_like(raw_f)
            L.check(lib.nerf_mlp_forward_rays(L.ptr(o), L.ptr(d), L.ptr(t_sorted), 192, n, 192, pk_f.data_ptr(), L.ptr(raw_f_full), prec, st))
        ev[6].record()
        torch.cuda.synchronize()
        for i, k in enumerate(names):
            acc[k] += ev[i].elapsed_time(ev[i + 1])
    if not full_coarse:
        del acc["mlp_coarse_full_network"], acc["mlp_fine_full_network"]
    out = {k: v / steps for k, v in acc.items()}
    # fine tiles (32 consecutive samples) without a single sigma > 0: the fp32 fine launch stops those after the sigma head
    # (their colours are multiplied by exactly zero in compositing); counted here for the executed-FLOP figure
    out["_dead_fine_tiles"] = int((raw_f[..., 3] <= 0).reshape(-1, 32).all(-1).sum().item())
    out["_fine_tiles"] = n * 192 // 32
    return out


def host_cores():
    """CPU threads this process may really use: affinity mask, cgroup quota, and the GPU box's
    per-GPU CPU share (16) -- os.cpu_count() alone reports the whole host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cap = int(os.environ.get("NERF_BENCH_CPU_THREADS", "16"))      # a 1-GPU box's CPU share is 16 threads (the host shows 256)
    return max(1, min(n, cap))


def cpu_baseline(sd, n_sample, budget_s=20.0):
    """The CPU oracle (port of the reference incl. its 512-point MLP chunking) on a bounded sample of
    rays of the same frame.  Checker code used as the reported baseline, never as product."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import nerf_oracle as orc
    cores = host_cores()
    torch.set_num_threads(cores)
    ids = torch.randperm(H * W, generator=torch.Generator().manual_seed(0))[:n_sample]
    o, d = orc.pinhole_rays(H, W, orc.camera_pose(40.0), pixel_ids=ids)
    with torch.no_grad():
        orc.render(sd, o[None, :128], d[None, :128])              # warm-up
        t0 = time.perf_counter()
        orc.render(sd, o[None, :512], d[None, :512])              # probe -> size the sample to the budget
        probe = time.perf_counter() - t0
        n_sample = int(max(512, min(n_sample, 512 * budget_s / max(probe, 1e-3)))) // 512 * 512
        ids, o, d = ids[:n_sample], o[:n_sample], d[:n_sample]
        print(f"[bench] cpu_baseline: probe 512 rays {probe:.2f} s on {cores} threads -> sample {n_sample} rays",
              file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        rgb, dep = orc.render(sd, o[None], d[None])
        dt = time.perf_counter() - t0
    return {"value": n_sample / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_sample} random rays of the same {H}x{W} frame, 64+128, oracle/nerf_oracle.py "
                      f"(torch CPU fp32, 512-point MLP chunks), {dt:.1f} s, {cores} threads "
                      f"(os.cpu_count()={os.cpu_count()})"}, (ids, rgb, dep)


def run_training(pkg, sd, dev, precision, steps, warmup, world, rank, compare_dense=True):
    """BASELINE config 3: 4096 rays per iteration per GPU, forward with activation save, backward through the adjoint kernels,
    [one gradient all-reduce], clip 40, Adam lr 5e-4 -- the reference's intended step.  One "step" = one iteration.
    A representative training state (round-2 VERDICT item 1d): the student is tests/golden/trained_ckpt.pth (3000 steps of this
    very step), the targets are renders of the scene it was trained on (the "sharp" teacher), and every step draws FRESH random
    pixels (a pool of batches rendered by the teacher before the timed region: in the reference the targets are dataset images).
    `sd` is the synthetic checkpoint (the teacher's base).  Returns the measured block (max over ranks)."""
    from nerf_replication_amd.training import train_step, FusedAdam
    n_rays = 4096
    net = pkg.Network()
    net.load_state_dict(load_weights("trained_ckpt.pth"), strict=True)
    net = net.to(dev).train()
    net.precision = precision
    ren = pkg.Renderer(net)
    n_dense = 5 if compare_dense and os.environ.get("NERF_DEAD_TILE_SKIP") != "0" else 0
    n_batches = warmup + steps + (n_dense + 1 if n_dense else 0)
    teacher = pkg.Network()
    teacher.load_state_dict(sharp_teacher(sd), strict=True)
    teacher = teacher.to(dev).eval()
    t_ren = pkg.Renderer(teacher)
    gen = torch.Generator().manual_seed(1000 + rank)
    ids = torch.stack([torch.randperm(H * W, generator=gen)[:n_rays] for _ in range(n_batches)]).to(dev)
    pool_o, pool_d = pkg.generate_rays(camera_pose_40(), H, W, 0.6911112070083618, dev, pixel_ids=ids.reshape(-1))
    with torch.no_grad():
        pool_c, _ = t_ren.render({"rays_o": pool_o[None], "rays_d": pool_d[None]})
    pool_o, pool_d = pool_o.reshape(n_batches, n_rays, 3), pool_d.reshape(n_batches, n_rays, 3)
    pool_c = pool_c.reshape(n_batches, n_rays, 3).float().contiguous()
    del teacher, t_ren
    batch = lambda i: (pool_o[i], pool_d[i], pool_c[i])
    opt = FusedAdam(net.parameters(), lr=5e-4, eps=1e-8, clip_value=40.0)      # one launch: clip 40 + Adam
    for i in range(warmup):
        first_loss = train_step(ren, opt, *batch(i))
    ren.live_tile_stats = []                 # per timed step: live / all 32-point tiles of the two backward passes (device ints)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = train_step(ren, opt, *batch(warmup + i))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    ms = elapsed / steps * 1e3
    # live-tile fractions of the timed steps (the backward drops tiles whose incoming gradient is zero throughout: exact)
    stats, ren.live_tile_stats = ren.live_tile_stats, None
    live_f = live_c = 1.0
    if stats:
        cf = [int(a.item()) for a, _, _, _ in stats]; cc = [int(c.item()) for _, _, c, _ in stats]
        if min(cf) >= 0 and min(cc) >= 0:
            live_f = sum(cf) / (len(cf) * stats[0][1]); live_c = sum(cc) / (len(cc) * stats[0][3])
    # the same step with the skipping switched off (every tile computed), a few steps right after the timed ones
    ms_dense = None
    if n_dense:
        os.environ["NERF_DEAD_TILE_SKIP"] = "0"
        try:
            train_step(ren, opt, *batch(warmup + steps))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n_dense):
                train_step(ren, opt, *batch(warmup + steps + 1 + i))
            torch.cuda.synchronize()
            ms_dense = (time.perf_counter() - t1) / n_dense * 1e3
        finally:
            del os.environ["NERF_DEAD_TILE_SKIP"]
    # FLOP per step and GPU.  Reference algorithm: forward + data gradient + weight gradient = 3 x 256 points x
    # 1 186 816 per ray.  Executed: the COARSE pass is density-only in all three thirds of the fp32 chain (its colour branch --
    # feature 256x256, views 283x128, rgb 128x3 = 204 288 FLOP per point -- is computed by the reference but never used, SURVEY
    # F6/F10); f32x still runs the coarse colour branch in its two chain kernels (on zeros) and skips only the weight-gradient jobs
    flop_ref = n_rays * POINTS_PER_RAY * FLOP_PER_POINT * 3.0
    flop_dense = flop_ref - n_rays * 64 * FLOP_DENSITY_SKIPPED * (3.0 if precision == "f32" else 1.0)
    fwd_f = n_rays * 192 * FLOP_PER_POINT
    if precision == "f32x":      # forward in full (coarse colour branch included); the two backward thirds only on live tiles
        fwd_c = n_rays * 64 * FLOP_PER_POINT
        flop = (fwd_f + fwd_c) + (fwd_f * live_f + fwd_c * live_c) + (fwd_f * live_f + n_rays * 64 * (FLOP_PER_POINT - FLOP_DENSITY_SKIPPED) * live_c)
    else:                        # forward in full; the two backward thirds only on live tiles
        fwd_c = n_rays * 64 * (FLOP_PER_POINT - FLOP_DENSITY_SKIPPED)
        # (the fine forward also drops the colour branch of its density-free tiles: counted as every backward-dead tile,
        # which can only under-count what was executed)
        flop = (fwd_f - (1.0 - live_f) * n_rays * 192 * FLOP_DENSITY_SKIPPED + fwd_c) + 2.0 * (fwd_f * live_f + fwd_c * live_c)
    # f32x: three fp16 (or six bf16) MFMAs per algorithmic MAC -> ceiling = a third of the fp16 peak
    peak = PEAK_F32_MFMA if precision == "f32" else PEAK_F16_MFMA / 3.0
    traffic = None
    try:      # HBM bytes per step from separate rocprofv3 --pmc passes of `bench.py --mode train` (profiles/collect.sh)
        traffic = json.load(open(os.path.join(REPO, "profiles", f"traffic_train_{precision}.json")))["traffic_bytes_per_step"]
    except (OSError, ValueError, KeyError):
        pass
    return {"rays_per_s": round(n_rays * world / (ms * 1e-3), 1), "ms_per_step": round(ms, 3), "steps": steps,
            "warmup": warmup, "rays_per_iter_per_gpu": n_rays,
            "state": "student tests/golden/trained_ckpt.pth, targets = renders of the 'sharp' teacher scene, fresh random 4096 pixels per step",
            "roofline": {"bound": "mfma", "achieved": round(flop / (ms * 1e-3) / 1e12, 2), "peak": round(peak / 1e12, 1),
                         "unit": "TFLOP/s", "frac": round(flop / (ms * 1e-3) / peak, 4), "traffic": traffic,
                         "flop_per_step_executed": flop, "flop_per_step_reference_algorithm": flop_ref,
                         # scene-independent: every tile computed (NERF_DEAD_TILE_SKIP=0; the coarse pass still density-only)
                         "ms_per_step_every_tile": None if ms_dense is None else round(ms_dense, 3),
                         "rays_per_s_every_tile": None if ms_dense is None else round(n_rays * world / (ms_dense * 1e-3), 1),
                         "frac_every_tile": None if ms_dense is None else round(flop_dense / (ms_dense * 1e-3) / peak, 4),
                         # SURVEY 8(d)'s definition: the REFERENCE algorithm's FLOP over the measured time; exceeds what the kernels
                         # do (and may exceed 1) by exactly the work skipped: dead tiles and the coarse colour branch
                         "frac_reference_algorithm": round(flop_ref / (ms * 1e-3) / peak, 4)},
            "live_tile_fraction": {"fine": round(live_f, 4), "coarse": round(live_c, 4),
                                   "note": "32-point tiles with a non-zero incoming gradient, mean over the timed steps; the "
                                           "backward skips the others (exact: d loss / d raw is zero wherever relu(sigma) = 0); "
                                           "scene- and step-dependent"},
            "ms_per_step_without_dead_tile_skip": None if ms_dense is None else round(ms_dense, 3),
            "first_loss": round(first_loss.item(), 6) if warmup else None, "final_loss": round(loss.item(), 6)}


TRAIN_DTYPE = {"f32": "f32", "f32x": "f32x (split-fp16 fwd/bwd chains, bf16x3 weight gradients for the 256x256 layers, "
                                     "fp32 MFMA for the small ones)"}


def train_bench(pkg, sd, dev, args, world, rank):
    """`--mode train`: the config-3 step as its own JSON line (what profiles/collect.sh profiles)."""
    precision = args.precision if args.precision in ("f32", "f32x") else "f32"
    r = run_training(pkg, sd, dev, precision, args.steps, args.warmup, world, rank, compare_dense=args.compare_dense)
    if rank == 0:
        print(json.dumps({"metric": "rays/sec (training, 4096 rays/iter, 64+128, fwd+bwd+Adam)",
                          "value": r["rays_per_s"], "unit": "rays/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": TRAIN_DTYPE[precision], "data": "synthetic",
                          "config": {"workload": "BASELINE.json configs[2]: 4096 rays/iter per GPU, MSE on fine RGB (student: the trained "
                                                 "checkpoint; targets: renders of its teacher scene; fresh random pixels per step), "
                                                 "clip 40, Adam 5e-4; data parallel: one 4.77 MB gradient all-reduce per step"},
                          "roofline": r["roofline"], "final_loss": r["final_loss"],
                          "live_tile_fraction": r["live_tile_fraction"],
                          "ms_per_step_without_dead_tile_skip": r["ms_per_step_without_dead_tile_skip"]}), flush=True)


def run_config5(pkg, sd, dev, world, rank, steps=2):
    """BASELINE config 5: one 1600x1600 frame (2 560 000 rays, 64+128), fp16 activations / fp32 accumulate, rays
    sharded over the ranks exactly like the headline frame.  Rays generated per rank for its own tile."""
    from nerf_replication_amd.dist import render_shard, shard_bounds
    res = 1600
    net = pkg.Network()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    net.precision = "f16"
    ren = pkg.Renderer(net)
    n = res * res
    lo, hi, _ = shard_bounds(n, rank, world)
    o, d = pkg.generate_rays(camera_pose_40(), res, res, 0.6911112070083618, dev, pixel_begin=lo, n_pixels=hi - lo)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        render_shard(ren, o, d, n)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            rgb, dep = render_shard(ren, o, d, n)
        fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    finite = bool(torch.isfinite(rgb).all().item() and torch.isfinite(dep).all().item())
    rays_per_s = n * steps / elapsed
    # executed FLOP per ray: the coarse launch is density-only (see the headline's config.coarse_pass) and waves without
    # density skip the colour branch in the fine launch: counted from the sigma output of one more pass over this shard
    st5 = time_stages(pkg, net, ren, o, d, 1, prec=pkg._lib.PRECISIONS["f16"], full_coarse=True)
    dead5, tiles5 = st5["_dead_fine_tiles"], st5["_fine_tiles"]
    flop_per_ray = 64 * (FLOP_PER_POINT - FLOP_DENSITY_SKIPPED) + 192 * FLOP_PER_POINT - dead5 * 32 * FLOP_DENSITY_SKIPPED / max(1, hi - lo)
    ms5 = elapsed / steps * 1e3
    full5 = n / ((ms5 + st5["mlp_coarse_full_network"] - st5["mlp_coarse"] + st5["mlp_fine_full_network"] - st5["mlp_fine"]) * 1e-3)
    return {"workload": "1600x1600 frame = 2560000 rays, 64+128 (coarse pass density-only), fp16 activations + fp32 accumulate (nerf_mlp_f16s_kernel: 16x16x32 MFMA tiles; the last sample of every ray re-evaluated with the split-fp16 stream: far-plane guard)",
            "rays_per_s": round(rays_per_s, 1), "ms_per_frame": round(elapsed / steps * 1e3, 2), "steps": steps, "warmup": 1,
            "n_gpus": world, "finite": finite,
            "roofline": {"bound": "mfma", "achieved": round(rays_per_s * flop_per_ray / 1e12, 1),
                         "peak": PEAK_F16_MFMA * world / 1e12, "unit": "TFLOP/s",
                         "frac": round(rays_per_s * flop_per_ray / (PEAK_F16_MFMA * world), 4),
                         "flop_per_ray_executed": flop_per_ray, "flop_per_ray_reference_algorithm": POINTS_PER_RAY * FLOP_PER_POINT,
                         "rays_per_s_full_network": round(full5, 1),
                         "frac_full_network": round(full5 * POINTS_PER_RAY * FLOP_PER_POINT / (PEAK_F16_MFMA * world), 4),
                         "frac_reference_algorithm": round(rays_per_s * POINTS_PER_RAY * FLOP_PER_POINT / (PEAK_F16_MFMA * world), 4),
                         "fine_tiles_without_density": {"tiles": dead5, "of": tiles5}}}


def cpu_baseline_config1(sd):
    """BASELINE.json configs[0]: 1024-ray batch, 64 coarse samples only (N_importance = 0), the reference's CPU path
    -- here the oracle port with its 512-point MLP chunks, best of 3, on this host's cores."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import nerf_oracle as orc
    cores = host_cores()
    torch.set_num_threads(cores)
    ids = torch.from_numpy(__import__("numpy").random.default_rng(0).choice(800 * 800, 1024, replace=False))
    o, d = orc.pinhole_rays(800, 800, orc.camera_pose(40.0), pixel_ids=ids)
    best = float("inf")
    with torch.no_grad():
        for _ in range(4):
            t0 = time.perf_counter()
            rgb, dep = orc.render(sd, o[None], d[None], n_importance=0)
            best = min(best, time.perf_counter() - t0)
    return {"value": round(1024 / best, 1), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"BASELINE configs[0]: 1024 rays, 64 coarse samples only, best of 4 ({best * 1e3:.0f} ms), "
                      f"{cores} threads (os.cpu_count()={os.cpu_count()})"}, (o, d, rgb, dep)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=8192, help="rays in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--mode", default="render", choices=["render", "train"],
                    help="render (default: the headline 800x800 frame) or train (BASELINE config 3: 4096 rays/iter, "
                         "fused fwd+bwd HIP MLP + Adam; prints its own JSON line)")
    ap.add_argument("--res", type=int, default=800,
                    help="frame is res x res pixels (default 800 = the headline workload; 1600 = BASELINE configs[4], "
                         "camera_angle_x unchanged)")
    ap.add_argument("--fast-sampling", action="store_true",
                    help="the reference's optional ESS/ERT masked fine pass (volume_renderer.py:132-244, off in lego.yaml): "
                         "fine samples the coarse pass marks empty or occluded skip the MLP; a different image, reported "
                         "as its own metric")
    ap.add_argument("--compare-full-coarse", "--compare-full-network", dest="compare_full_coarse", action="store_true", default=True,
                    help="(default) also time the FULL coarse and fine networks (every colour computed) beside the launches the render "
                         "makes (density-only coarse pass; fine tiles without density stop after the sigma head), and report the frame "
                         "rate the headline would have with them: roofline.rays_per_s_full_network / frac_full_network")
    ap.add_argument("--no-full-network-compare", dest="compare_full_coarse", action="store_false",
                    help="skip those extra launches (profiles/collect.sh: keeps the profiled launch count minimal)")
    ap.add_argument("--no-dense-compare", dest="compare_dense", action="store_false",
                    help="--mode train: do not append the six steps with NERF_DEAD_TILE_SKIP=0 that give "
                         "ms_per_step_without_dead_tile_skip (profiles/collect.sh: keeps the rocprofv3 rows of the timed steps clean)")
    ap.add_argument("--no-extras", dest="extras", action="store_false",
                    help="skip the training (configs[2]) and 1600x1600 f16 (configs[4]) blocks that follow the headline")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16", "f32x", "f16m32"],
                    help="f32 (default, the reference's dtype: exact fp32 MFMA), f16 (BASELINE config 5: fp16 "
                         "activations, fp32 accumulate) or f32x (fp32-accurate: hi/lo split operands, 3 fp16 MFMAs per product)")
    args = ap.parse_args()
    global H, W
    H = W = args.res

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 as: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world
    # rehearsal switches for a 1-GPU box (the real N>1 run is one rank per GPU over RCCL):
    #   NERF_BENCH_SHARE_GPU=1  every rank uses cuda:0;  NERF_DIST_BACKEND=gloo  (RCCL refuses two ranks on one GPU)
    if os.environ.get("NERF_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("NERF_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NERF_BENCH_FORCE_DIST=1: a one-rank run that still builds the process group and makes every collective an N-rank run makes
    # (RCCL on the one GPU of a test box; nerf_replication_amd.dist honours NERF_DIST_FORCE_COLLECTIVE)
    force_dist = world == 1 and os.environ.get("NERF_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ["NERF_DIST_FORCE_COLLECTIVE"] = "1"
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # communicator setup (RCCL builds its rings lazily on the first collective of each kind): done here, outside any
        # step, so that even `--warmup 0` times the frame and not the library's initialisation
        _w = torch.zeros(4, device=dev)
        dist.all_reduce(_w)
        _g = torch.empty(4 * world, device=dev)
        dist.all_gather_into_tensor(_g, _w)
        torch.cuda.synchronize()

    import nerf_replication_amd as pkg
    from nerf_replication_amd.dist import render_shard, shard_bounds
    pkg._lib.load()                                   # fail loudly without the HIP extension
    sd = load_weights()
    if args.mode == "train":
        train_bench(pkg, sd, dev, args, world, rank)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    net = pkg.Network()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    net.precision = args.precision
    prec = pkg._lib.PRECISIONS[args.precision]
    # f32x executes 3 fp16 MFMAs per algorithmic MAC: its ceiling in ALGORITHMIC flops is a third of the fp16 peak
    peak = {0: PEAK_F32_MFMA, 1: PEAK_F16_MFMA, 2: PEAK_F16_MFMA / 3.0, 3: PEAK_F16_MFMA}[prec]
    ren = pkg.Renderer(net)
    ren.fast_sampling = bool(args.fast_sampling)
    # every rank generates ONLY its own tile of the frame's rays, on the device (nerf_generate_rays = the dataset
    # formula, blender.py:102-127), resident in HBM before the timed region starts; no rank holds the whole frame's rays
    n = H * W
    lo, hi, _ = shard_bounds(n, rank, world)
    o, d = pkg.generate_rays(camera_pose_40(), H, W, 0.6911112070083618, dev, pixel_begin=lo, n_pixels=hi - lo)
    events = []

    def step():
        with torch.no_grad():
            return render_shard(ren, o, d, n, events=events)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    del events[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rgb, dep = step()
    fence()
    elapsed = time.perf_counter() - t0
    assert rgb.shape == (n, 3) and dep.shape == (n,)
    my_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, len(events))     # this rank's own render time per step
    per_rank_ms = [my_ms]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        mine = torch.tensor([my_ms], device=dev, dtype=torch.float64)
        allms = torch.empty(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allms, mine)
        per_rank_ms = allms.tolist()
    ms_per_step = elapsed / args.steps * 1e3
    value = n * args.steps / elapsed
    if rank == 0:
        print(f"[bench] {args.steps} steps: {ms_per_step:.1f} ms/step, {value:.0f} rays/s", file=sys.stderr, flush=True)

    # dominant-kernel roofline on this rank's shard (HIP events on the launch stream)
    # (the full networks -- every colour computed -- are timed beside the launches the render makes: separate template instances,
    #  so their launches land in rows of their own in `rocprofv3 --stats`; --no-full-network-compare drops them)
    stages = time_stages(pkg, net, ren, o, d, max(1, min(args.steps, 3)), prec, full_coarse=args.compare_full_coarse)
    mlp_ms_per_launch = (stages["mlp_coarse"] + stages["mlp_fine"]) / 2.0
    # FLOP actually executed: the coarse launch stops after the sigma head (feature_linear 256x256, views_linears.0
    # 283x128 and rgb_linear 128x3 = 204 288 FLOP per point are never read by the reference's hierarchical render and are
    # not evaluated)
    coarse_flop_per_point = FLOP_PER_POINT - FLOP_DENSITY_SKIPPED
    dead_tiles, fine_tiles = stages.pop("_dead_fine_tiles"), stages.pop("_fine_tiles")
    flop_frame_executed = (hi - lo) * (64 * coarse_flop_per_point + 192 * FLOP_PER_POINT)
    flop_frame_executed -= dead_tiles * 32 * FLOP_DENSITY_SKIPPED        # dead tiles skip the colour branch (every precision)
    flop_frame_reference = (hi - lo) * POINTS_PER_RAY * FLOP_PER_POINT
    flop_per_launch = flop_frame_executed / 2.0
    achieved = flop_per_launch / (mlp_ms_per_launch * 1e-3) / 1e12
    traffic, traffic_src = None, None
    try:      # HBM bytes per launch come from separate rocprofv3 --pmc passes (cannot be read in-process);
        tj = json.load(open(os.path.join(REPO, "profiles", f"traffic_{args.precision}.json")))   # only for the profiled workload
        if world == 1 and H == 800:
            traffic, traffic_src = tj["traffic_bytes_per_launch"], f"profiles/traffic_{args.precision}.json (rocprofv3 --pmc, not this run)"
    except (OSError, ValueError, KeyError):
        pass
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": round(achieved / (peak / 1e12), 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": {"f32": "nerf_mlp_f32_kernel", "f16": "nerf_mlp_f16s_kernel", "f16m32": "nerf_mlp_f16_kernel", "f32x": "nerf_mlp_f32x_kernel"}[args.precision], "avg_launch_ms": round(mlp_ms_per_launch, 3),
                "flop_per_frame_executed": flop_frame_executed, "flop_per_frame_reference_algorithm": flop_frame_reference,
                "fine_tiles_without_density": {"tiles": dead_tiles, "of": fine_tiles,
                                               "note": "32-sample tiles whose sigma is <= 0 throughout: weight exactly 0 in compositing; "
                                                       "the fp32 fine launch stops them after the sigma head (scene-dependent)"},
                "stage_ms": {k: round(v, 3) for k, v in stages.items()}}
    # SURVEY 8(d)'s own definition: the REFERENCE algorithm's FLOP (303 824 896 per ray) over the measured time.  It exceeds what the
    # kernels execute -- and may exceed 1 -- by exactly the skipped dead code (coarse colour branch, colours of density-free tiles)
    roofline["frac_reference_algorithm"] = round(value * POINTS_PER_RAY * FLOP_PER_POINT / (peak * world), 4)
    if args.compare_full_coarse:
        # the same frame with the coarse network evaluated in full (its colour computed and dropped, as the reference
        # does): derived from the measured step time and the two coarse launches timed side by side
        roofline["rays_per_s_if_coarse_colour_were_computed"] = round(
            n / ((ms_per_step + stages["mlp_coarse_full_network"] - stages["mlp_coarse"]) * 1e-3), 1)
        # ... and with, in addition, the colours of the zero-density fine tiles computed (every FLOP of the reference's algorithm):
        # the SCENE-INDEPENDENT figures of this line
        full = n / ((ms_per_step + stages["mlp_coarse_full_network"] - stages["mlp_coarse"]
                     + stages["mlp_fine_full_network"] - stages["mlp_fine"]) * 1e-3)
        roofline["rays_per_s_if_every_colour_were_computed"] = round(full, 1)
        roofline["rays_per_s_full_network"] = round(full, 1)
        roofline["frac_full_network"] = round(full * POINTS_PER_RAY * FLOP_PER_POINT / (peak * world), 4)
        # the two full-network launches by themselves (kernel time only, this rank's shard)
        roofline["frac_full_network_kernels"] = round((hi - lo) * POINTS_PER_RAY * FLOP_PER_POINT /
                                                      ((stages["mlp_coarse_full_network"] + stages["mlp_fine_full_network"]) * 1e-3) / peak, 4)

    out = None
    if rank == 0:
        out = {"metric": f"rays/sec ({H}x{W}, 64+128 samples)" + (", ESS/ERT masked fine pass" if args.fast_sampling else ""), "value": round(value, 1), "unit": "rays/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": {0: "f32", 1: "f16 (fp32 accumulate, 32x32x16 MFMA tiles)", 2: "f32 emulated (hi/lo fp16 split x3, fp32 accumulate)",
                         3: "f16 (fp32 accumulate)"}[prec],
               "data": "synthetic",
               "config": {"workload": f"lego-shaped {H}x{W} frame = {H * W} pinhole rays, 64 coarse + 128 fine "
                                      "hierarchical samples, 8+1-layer W=256 NeRF x2, seeded synthetic weights "
                                      "(latest.pth unavailable offline); BASELINE.json "
                                      + ("configs[1]" if H == 800 else "configs[4] frame size" if H == 1600 else "custom frame size"),
                          "coarse_pass": ("density-only launch: sigma is the only coarse output the reference reads when "
                                          "N_importance > 0 (volume_renderer.py:335), rgb/depth are bit-identical to running "
                                          "the full coarse network; fp32 fine launch: 32-sample tiles without a single sigma > 0 stop after the sigma head "
                                          "too (weight exactly 0 in compositing).  --compare-full-network times the full networks beside "
                                          "them on this line: roofline.rays_per_s_full_network / frac_full_network"
                                          ),
                          "rays_per_step": n, "parallelism": f"ray-tile shard x{world} (each rank generates and renders only "
                                                             "its tile) + 1 all_gather",
                          **({"fast_sampling": "ESS/ERT masks, weights_threshold %.2f; the roofline block times the UNMASKED "
                                               "MLP launches" % ren.weights_threshold} if args.fast_sampling else {})},
               "roofline": roofline,
               # self-check for a scaling record: what torch.distributed really ran, and every rank's own render time
               "world_size": dist.get_world_size() if dist.is_initialized() else 1,
               "dist_backend": (dist.get_backend() if dist.is_initialized() else None),
               "per_rank_compute_ms": [round(x, 3) for x in per_rank_ms]}
        if world == 1 and args.cpu_sample > 0:
            base, (ids, ref_rgb, ref_dep) = cpu_baseline(sd, args.cpu_sample)
            base["value"] = round(base["value"], 1)
            out["cpu_baseline"] = base
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            import nerf_oracle as orc
            # the oracle rays are the float64-built fixtures' formula; re-render exactly those rays for PSNR
            oo, dd = orc.pinhole_rays(H, W, orc.camera_pose(40.0), pixel_ids=ids)
            with torch.no_grad():
                g_rgb, g_dep = ren.render({"rays_o": oo[None].to(dev), "rays_d": dd[None].to(dev)})
            out["psnr_vs_cpu_oracle_db"] = round(orc.psnr(g_rgb.cpu(), ref_rgb), 1)
            out["speedup_vs_cpu_baseline"] = round(value / base["value"], 1)
            # BASELINE configs[0]: the reference's own CPU-runnable case, and the HIP path on the same 1024 rays
            c1, (o1, d1, c1_rgb, _) = cpu_baseline_config1(sd)
            ren.N_importance = 0
            with torch.no_grad():
                o1d, d1d = o1[None].to(dev), d1[None].to(dev)
                ren.render({"rays_o": o1d, "rays_d": d1d})
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    g1, _ = ren.render({"rays_o": o1d, "rays_d": d1d})
                torch.cuda.synchronize()
                c1["hip_same_workload"] = {"rays_per_s": round(1024 * 20 / (time.perf_counter() - t1), 1),
                                           "max_abs_rgb_diff_vs_cpu": float((g1.cpu() - c1_rgb).abs().max())}
            ren.N_importance = 128
            out["cpu_baseline_config1"] = c1
            # informational: the other arithmetic modes of the same path on the same frame (NOT the headline value)
            others = {}
            for pname in ("f32x", "f16"):
                if pname == args.precision:
                    continue
                net.precision = pname
                with torch.no_grad():
                    ren.render({"rays_o": o[None], "rays_d": d[None]})
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(3):
                        ren.render({"rays_o": o[None], "rays_d": d[None]})
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t1) / 3
                    p_rgb, _ = ren.render({"rays_o": oo[None].to(dev), "rays_d": dd[None].to(dev)})
                others[pname] = {"rays_per_s": round(n / dt, 1), "ms_per_frame": round(dt * 1e3, 2),
                                 "psnr_vs_cpu_oracle_db": round(orc.psnr(p_rgb.cpu(), ref_rgb), 1),
                                 "arithmetic": {"f32x": "hi/lo fp16 operand split, 3 MFMAs per product, fp32 accumulate "
                                                        "(meets the fp32 path's parity tolerances)",
                                                "f16": "fp16 activations and weights, fp32 accumulate (config 5)"}[pname]}
            net.precision = args.precision
            out["other_precisions"] = others

    # BASELINE configs[2] and configs[4] on the same line (every rank takes part when N>1), AFTER the headline's timed
    # region; a failure here is reported in its block and never costs the headline measurement
    if args.extras and H == 800 and not args.fast_sampling:
        del net, ren, o, d
        extra = {}
        try:
            extra["training"] = {"workload": "BASELINE.json configs[2]: 4096 rays/iter per GPU, render (64+128) with "
                                             "activation save -> MSE on fine RGB -> adjoint HIP kernels -> [gradient "
                                             "all-reduce] -> clip 40 + Adam 5e-4 (one launch)", "n_gpus": world,
                                 **{p: run_training(pkg, sd, dev, p, 20, 3, world, rank) for p in ("f32", "f32x")}}
        except Exception as exc:                                 # noqa: BLE001
            extra["training"] = {"error": f"{type(exc).__name__}: {exc}"}
        try:
            extra["config5"] = run_config5(pkg, sd, dev, world, rank)
        except Exception as exc:                                 # noqa: BLE001
            extra["config5"] = {"error": f"{type(exc).__name__}: {exc}"}
        if out is not None:
            out.update(extra)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
