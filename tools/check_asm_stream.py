#!/usr/bin/env python3
"""Static check of the inline-asm weight stream of nerf_mlp_f32_kernel<*, false> (the inference instances).

The asm `global_load_dwordx4` loads are asynchronous behind the compiler's back: between a load and the asm
`s_waitcnt vmcnt(N)` that covers it, NO instruction may read or write the destination registers (the compiler could
spill, copy or reuse them -- it believes the asm's outputs are ready immediately).  This script compiles the kernels to
ISA and verifies exactly that for every asm load; it is run by tests/test_abi_symbols.py.

    python tools/check_asm_stream.py        # exit status 0 = no hazard
"""
import bisect
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "nerf_replication_amd", "csrc", "nerf_kernels.hip")
FLAGS = "-O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -S --cuda-device-only"
KERNELS = ["_Z19nerf_mlp_f32_kernelILb1ELb0EEv7MlpArgs", "_Z19nerf_mlp_f32_kernelILb0ELb0EEv7MlpArgs"]


def vregs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def check(lines, name):
    a = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    b = next(i for i, l in enumerate(lines) if i > a and ".amdhsa_kernel " + name in l)
    K = lines[a:b]
    ins = [(i, l.strip()) for i, l in enumerate(K) if l.strip() and not l.strip().startswith((";", "."))]
    in_asm = lambda idx: "ASMSTART" in K[ins[idx][0] - 1]
    is_vmem = lambda l: l.startswith(("global_", "buffer_", "scratch_", "flat_"))
    asm_load = lambda idx: ins[idx][1].startswith("global_load_dwordx4") and "s[" in ins[idx][1] and in_asm(idx)
    # every s_waitcnt with a vmcnt field covers, whoever wrote it (the compiler's own waits count all outstanding
    # memory operations, the asm loads included)
    waits = [(idx, int(re.search(r"vmcnt\((\d+)\)", l).group(1))) for idx, (i, l) in enumerate(ins)
             if l.startswith("s_waitcnt") and "vmcnt(" in l]
    vm = [idx for idx, (i, l) in enumerate(ins) if is_vmem(l)]
    loads = [v for v in vm if asm_load(v)]
    hazards = []
    for ld in loads:
        dst = vregs(ins[ld][1].split(",")[0])
        cover = None
        for w, cnt in waits:        # the first asm wait with at least `cnt` younger memory operations in between
            if w > ld and bisect.bisect_left(vm, w) - bisect.bisect_right(vm, ld) >= cnt:
                cover = w
                break
        if cover is None:
            hazards.append(("never waited for", ins[ld][1], ""))
            continue
        for j in range(ld + 1, cover):
            l = ins[j][1]
            touched = vregs(l.split(",")[0]) if asm_load(j) else vregs(l)
            if touched & dst:
                hazards.append(("touched before its wait", ins[ld][1], l))
    return len(loads), hazards


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(f"/opt/rocm/bin/hipcc {FLAGS} -o {out} {SRC}", shell=True, check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    bad = 0
    for k in KERNELS:
        n, hz = check(lines, k)
        print(f"{k}: {n} asm loads, {len(hz)} hazards")
        for kind, ld, use in hz[:10]:
            print("   ", kind, "|", ld, "|", use)
        bad += len(hz)
        if n == 0:
            print("    (no asm loads found: NERF_F32_ASM_LOADS off?)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
