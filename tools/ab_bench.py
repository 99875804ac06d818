#!/usr/bin/env python3
"""Interleaved in-process A/B timing of kernel build variants (cdna_hip_programming.md rule 24).

    python tools/ab_bench.py f16 "base:" "ring8:-DNERF_F16_PF_RING=8 -DNERF_F16_PF_DIST=6" ...

Builds one libnerf variant per spec into nerf_replication_amd/csrc/variants/, then times
nerf_mlp_forward_rays (fine model, 192 samples/ray) for each, round-robin, HIP events, and prints
median / min ms and TFLOP/s.  Developer tool, not part of the product or the tests."""
import ctypes
import os
import statistics
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
CSRC = os.path.join(REPO, "nerf_replication_amd", "csrc")
VDIR = os.path.join(CSRC, "variants")
FLAGS = "-O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -shared -fPIC -DNERF_TIMING_BUILD"
XFLAGS = "-mllvm -amdgpu-mfma-vgpr-form=1"        # csrc/Makefile: what nerf_kernels_x.hip is compiled with on top


def build(name, extra):
    os.makedirs(VDIR, exist_ok=True)
    out = os.path.join(VDIR, f"lib_{name}.so")
    # two translation units, as csrc/Makefile builds them (the split-fp16 kernels with XFLAGS on top)
    objs = []
    procs = []
    for unit, x in (("nerf_kernels", ""), ("nerf_kernels_x", XFLAGS)):
        obj = os.path.join(VDIR, f"{unit}_{name}.o")
        objs.append(obj)
        procs.append(subprocess.Popen(f"/opt/rocm/bin/hipcc {FLAGS.replace('-shared ', '')} {x} {extra} -c -o {obj} {os.path.join(CSRC, unit + '.hip')}", shell=True))
    if any(pr.wait() != 0 for pr in procs):
        raise SystemExit(f"variant {name!r}: compile failed")
    subprocess.run(f"/opt/rocm/bin/hipcc {FLAGS} -o {out} {' '.join(objs)}", shell=True, check=True)
    for o in objs:
        os.remove(o)
    with open(out + ".flags", "w") as f:          # what this binary was built with: a stale or flag-less rebuild cannot pose as it
        f.write(extra.strip())
    if os.environ.get("AB_NO_CHECK") != "1":      # a variant whose asm rings the compiler broke measures (and computes) nonsense
        device_flags = FLAGS.replace("-shared -fPIC", "")
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_asm_stream.py"), "--flags", f"{device_flags} {extra}"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            print(f"!! variant {name!r}: tools/check_asm_stream.py reports register hazards -- its results and timings are not valid:")
            print("\n".join(l for l in r.stdout.splitlines() if " 0 hazards" not in l)[:2000])
    return out


def ensure(name, extra, force=False):
    """Path of variant `name`, (re)built when missing, when `extra` differs from what the existing binary was built with, or
    on `force`.  A spec without flags ("name:") reuses an existing binary but never BUILDS one unless the name is base/new."""
    out = os.path.join(VDIR, f"lib_{name}.so")
    have = open(out + ".flags").read() if os.path.exists(out) and os.path.exists(out + ".flags") else None
    extra = extra.strip()
    if have is None and not extra and name not in ("base", "new"):
        raise SystemExit(f"variant {name!r}: no binary and no flags given -- refusing to build a flag-less look-alike")
    if force or have is None or (extra and extra != have):
        build(name, extra)
    return out


def main():
    prec_name = sys.argv[1]
    specs = [s.split(":", 1) for s in sys.argv[2:]]
    build_only = os.environ.get("AB_BUILD_ONLY") == "1"
    libs = {}
    for name, extra in specs:
        libs[name] = ensure(name, extra, force=build_only)
    if build_only:
        print("built", list(libs))
        return
    import nerf_replication_amd as pkg
    L = pkg._lib
    prec = L.PRECISIONS[prec_name]
    n_rays, S = int(os.environ.get("AB_RAYS", "160000")), 192
    dev = torch.device("cuda:0")
    ck = torch.load(os.path.join(REPO, "tests", "golden", "synthetic_ckpt.pth"), weights_only=True)["net"]
    net = pkg.Network(); net.load_state_dict(ck); net = net.to(dev).eval(); net.precision = prec_name
    g = torch.Generator(device="cpu").manual_seed(0)
    d = torch.randn(n_rays, 3, generator=g); d = (d / d.norm(dim=-1, keepdim=True)).to(dev)
    o = (torch.randn(n_rays, 3, generator=g) * 0.1 + torch.tensor([0., 0., 4.])).to(dev)
    t = torch.sort(torch.rand(n_rays, S, generator=g) * 4 + 2, dim=-1).values.to(dev).contiguous()
    raw = torch.empty(n_rays, S, 4, device=dev)
    st = L.stream_of(dev)
    handles = {}
    for name, path in libs.items():
        lib = ctypes.CDLL(path)
        for fn, (res, args) in L._PROTOS.items():
            f = getattr(lib, fn); f.restype, f.argtypes = res, args
        nbytes = lib.nerf_packed_model_bytes(prec)
        pk = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        params = [p.detach().contiguous() for p in net.model_fine.ordered_params()]
        arr = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in params])
        assert lib.nerf_pack_model(arr, pk.data_ptr(), prec, st) == 0
        handles[name] = (lib, pk)
    ref = None
    times = {k: [] for k in handles}
    rounds = int(os.environ.get("AB_ROUNDS", "7"))
    for rnd in range(rounds + 1):
        for name, (lib, pk) in handles.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.nerf_mlp_forward_rays(o.data_ptr(), d.data_ptr(), t.data_ptr(), S, n_rays, S, pk.data_ptr(),
                                           raw.data_ptr(), prec, st)
            e1.record(); torch.cuda.synchronize()
            assert rc == 0, lib.nerf_last_error()
            if rnd == 0:
                if ref is None:
                    ref = raw.clone()
                else:
                    err = (raw - ref).abs().max().item()
                    print(f"  {name}: max|diff vs first variant| = {err:.3e}")
            else:
                times[name].append(e0.elapsed_time(e1))
    flop = n_rays * S * 1186816
    for name, ts in times.items():
        med = statistics.median(ts)
        print(f"{name:>14}: median {med:8.3f} ms  min {min(ts):8.3f} ms  {flop / med / 1e9:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
