#!/usr/bin/env python3
"""Interleaved in-process A/B timing of build variants of the SMALL-LAYER weight-gradient kernels (companion of
tools/ab_bench.py / ab_train.py).

    python tools/ab_wgrad.py "base:" "c++:-DNERF_WGVEC_ASM=0" "pf32:-DNERF_WGVEC_PF41=32" ...

Per variant and round: nerf_wgrad on the six small shapes of a training step's fine pass (AB_POINTS points, default
4096 x 192), each between its own pair of HIP events, in the layouts mlp_backward_impl uses.  Developer tool."""
import ctypes
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import ab_bench  # noqa: E402

# name, (ldz, zc0, n_out), (ldh, hc0, n_in), (ldw, wc0), bias
SHAPES = [("rgb_linear 3x128", (4, 0, 3), (128, 0, 128), (128, 0), True),
          ("alpha_linear 1x256", (4, 3, 1), (256, 0, 256), (256, 0), True),
          ("views feature 128x256", (128, 0, 128), (256, 0, 256), (283, 0), True),
          ("views dirs 128x27", (128, 0, 128), (32, 0, 27), (283, 256), False),
          ("PE->256 256x63", (256, 0, 256), (64, 0, 63), (319, 0), True)]


def main():
    specs = [s.split(":", 1) for s in sys.argv[1:]]
    libs = {name: ab_bench.ensure(name, extra, force=os.environ.get("AB_BUILD_ONLY") == "1") for name, extra in specs}
    if os.environ.get("AB_BUILD_ONLY") == "1":
        print("built", list(libs))
        return
    import nerf_replication_amd as pkg
    L = pkg._lib
    P = int(os.environ.get("AB_POINTS", str(4096 * 192)))
    dev = torch.device("cuda:0")
    st = L.stream_of(dev)
    bufs = {}
    for name, (ldz, zc0, n_out), (ldh, hc0, n_in), (ldw, wc0), bias in SHAPES:
        bufs[name] = (torch.randn(P, ldz, device=dev), torch.randn(P, ldh, device=dev),
                      torch.zeros(n_out, ldw, device=dev), torch.zeros(n_out, device=dev))
    handles = {}
    for name, path in libs.items():
        lib = ctypes.CDLL(path)
        for fn, (res, args) in L._PROTOS.items():
            f = getattr(lib, fn); f.restype, f.argtypes = res, args
        handles[name] = lib
    times = {k: {s[0]: [] for s in SHAPES} for k in handles}
    first = {}
    rounds = int(os.environ.get("AB_ROUNDS", "7"))
    for rnd in range(rounds + 1):
        for vname, lib in handles.items():
            for name, (ldz, zc0, n_out), (ldh, hc0, n_in), (ldw, wc0), bias in SHAPES:
                dz, hin, dw, db = bufs[name]
                dw.zero_(); db.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = lib.nerf_wgrad(dz.data_ptr(), ldz, zc0, n_out, hin.data_ptr(), ldh, hc0, n_in, dw.data_ptr(), ldw, wc0,
                                    db.data_ptr() if bias else None, P, st)
                e1.record(); torch.cuda.synchronize()
                assert rc == 0, lib.nerf_last_error()
                if rnd == 0:
                    if name not in first:
                        first[name] = dw.clone()
                    else:
                        err = ((dw - first[name]).abs().max() / first[name].abs().max()).item()
                        print(f"  {vname} / {name}: max rel diff vs first variant = {err:.2e}")
                else:
                    times[vname][name].append(e0.elapsed_time(e1) * 1e3)
    for vname in handles:
        tot = 0.0
        parts = []
        for s in SHAPES:
            med = statistics.median(times[vname][s[0]])
            tot += med * (2 if s[0].startswith("PE") else 1)            # two PE -> 256 layers per pass
            parts.append(f"{s[0].split()[0]} {med:6.1f}")
        print(f"{vname:>10}: " + "  ".join(parts) + f"   | fine-pass total {tot:7.1f} us")


if __name__ == "__main__":
    main()
