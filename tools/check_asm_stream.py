#!/usr/bin/env python3
"""Static check of the kernels that fill a register ring with inline-asm loads: nerf_mlp_f32_kernel<*, false> (the
inference instances), nerf_wgrad256_f32_asm_kernel and every nerf_wgrad_vec_f32_asm_kernel instance.

The asm `global_load_dwordx4` loads are asynchronous behind the compiler's back: between a load and the asm
`s_waitcnt vmcnt(N)` that covers it, NO instruction may read or write the destination registers (the compiler could
spill, copy or reuse them -- it believes the asm's outputs are ready immediately).  This script compiles the kernels to
ISA and verifies exactly that for every asm load; csrc/Makefile runs it on every build of the library (with the
build's own flags; a hazard fails the build) and tests/test_abi_symbols.py runs it with a negative self-test.

    python tools/check_asm_stream.py        # exit status 0 = no hazard
"""
import bisect
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# code-generation flags: the Makefile passes ITS OWN ($(CXXFLAGS)) through --flags, so the ISA checked here is the ISA
# of the library being built; the default below mirrors csrc/Makefile for stand-alone runs
FLAGS = "-O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950"
HIPCC = "/opt/rocm/bin/hipcc"
KERNELS = ["_Z19nerf_mlp_f32_kernelILb1ELb0ELb0ELb0EEv7MlpArgs", "_Z19nerf_mlp_f32_kernelILb0ELb0ELb0ELb0EEv7MlpArgs",
           "_Z19nerf_mlp_f32_kernelILb1ELb1ELb0ELb0EEv7MlpArgs", "_Z19nerf_mlp_f32_kernelILb0ELb1ELb0ELb0EEv7MlpArgs",
           "_Z19nerf_mlp_f32_kernelILb1ELb1ELb1ELb0EEv7MlpArgs", "_Z19nerf_mlp_f32_kernelILb1ELb0ELb1ELb0EEv7MlpArgs",
           "_Z19nerf_mlp_f32_kernelILb1ELb0ELb0ELb1EEv7MlpArgs", "_Z19nerf_mlp_f32_kernelILb1ELb1ELb0ELb1EEv7MlpArgs",
           "_Z23nerf_mlp_bwd_f32_kernelILb0EEv7BwdArgs", "_Z23nerf_mlp_bwd_f32_kernelILb1EEv7BwdArgs",
           "_Z28nerf_wgrad256_f32_asm_kernelILb0EEv10WgradBatch", "_Z28nerf_wgrad256_f32_asm_kernelILb1EEv10WgradBatch"]
# every instance of these templates found in the ISA is checked too (their ring depth is part of the mangled name)
KERNEL_PREFIXES = ["_Z29nerf_wgrad_vec_f32_asm_kernelI", "_Z20nerf_mlp_f32x_kernelI"]
# The library is two translation units (csrc/Makefile): the split-fp16 kernels are compiled with XFLAGS on top of the common
# flags (nerf_kernels_x.hip says why); each unit is checked on the flags IT is built with.
XFLAGS = "-mllvm -amdgpu-mfma-vgpr-form=1"
CSRC = os.path.join(REPO, "nerf_replication_amd", "csrc")
UNITS = [{"src": os.path.join(CSRC, "nerf_kernels.hip"), "x": False, "kernels": KERNELS, "prefixes": ["_Z29nerf_wgrad_vec_f32_asm_kernelI"]},
         {"src": os.path.join(CSRC, "nerf_kernels_x.hip"), "x": True, "kernels": [], "prefixes": ["_Z20nerf_mlp_f32x_kernelI"]}]


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
    ins, label_at = [], {}
    for i, l in enumerate(K):
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            label_at[t.split(":")[0]] = len(ins)                 # index of the first instruction after the label
        elif t and not t.startswith((";", ".")):
            ins.append((i, t))
    # source line -> inside an inline-asm block (";;#ASMSTART" ... ";;#ASMEND"; a block may hold several instructions)
    asm_line, inside = set(), False
    for i, l in enumerate(K):
        if "ASMSTART" in l:
            inside = True
        elif "ASMEND" in l:
            inside = False
        elif inside:
            asm_line.add(i)
    in_asm = lambda idx: ins[idx][0] in asm_line
    is_vmem = lambda l: l.startswith(("global_", "buffer_", "scratch_", "flat_"))
    is_lds = lambda l: l.startswith("ds_")
    asm_vm_load = lambda idx: re.match(r"global_load_dword(x2|x4)?\b", ins[idx][1]) and "s[" in ins[idx][1] and in_asm(idx)
    # asm LDS reads (nerf_mlp_f32x.hip.inc and friends): covered by `s_waitcnt lgkmcnt(N)`.  LDS operations return in order, so a
    # read has landed once at most N operations may be outstanding and >= N LDS operations were issued after it (scalar loads share
    # the counter but complete out of order: they are NOT counted as younger operations -- that is the conservative direction)
    asm_lds_load = lambda idx: re.match(r"ds_read_b(32|64|96|128)\b", ins[idx][1]) and in_asm(idx)
    asm_load = lambda idx: asm_vm_load(idx) or asm_lds_load(idx)
    loads = [idx for idx in range(len(ins)) if asm_load(idx)]
    hazards = []

    def walk(idx, dst, younger, depth, seen, lds):
        """Follow the control flow from instruction idx until a wait covers the load (at most `cnt` younger memory
        operations outstanding); report every instruction on the way that touches dst."""
        steps = 0
        pat = r"lgkmcnt\((\d+)\)" if lds else r"vmcnt\((\d+)\)"
        while idx < len(ins) and steps < 6000:
            steps += 1
            if (idx, younger) in seen:
                return
            seen.add((idx, younger))
            l = ins[idx][1]
            m = re.search(pat, l) if l.startswith("s_waitcnt") else None
            if m and younger >= int(m.group(1)):
                return                                            # covered on this path
            touched = vregs(l.split(",")[0]) if asm_load(idx) else vregs(l)
            if touched & dst:
                hazards.append(("touched before its wait", ld_text, l))
                return
            if (is_lds(l) if lds else is_vmem(l)):
                younger += 1
            if l.startswith("s_endpgm"):
                if lds:
                    hazards.append(("never waited for before the end of the program", ld_text, l))
                return
            if l.startswith("s_branch"):
                idx = label_at[l.split()[1]]
                continue
            if l.startswith("s_cbranch") and depth < 12:
                walk(label_at[l.split()[1]], dst, younger, depth + 1, seen, lds)
            idx += 1

    for ld in loads:
        ld_text = ins[ld][1]
        walk(ld + 1, vregs(ld_text.split(",")[0]), 0, 0, set(), bool(asm_lds_load(ld)))
    return len(loads), hazards


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", default=FLAGS, help="code-generation flags of the build being checked (csrc/Makefile: $(CXXFLAGS))")
    ap.add_argument("--xflags", default=XFLAGS, help="what the split-fp16 unit is compiled with on top of --flags (csrc/Makefile: $(XFLAGS))")
    ap.add_argument("--hipcc", default=HIPCC)
    args = ap.parse_args()
    extra = os.environ.get("NERF_CHECK_EXTRA_FLAGS", "")     # e.g. -DNERF_F32_ASM_OVERRUN=1 -DNERF_TIMING_BUILD: must report hazards
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for i, u in enumerate(UNITS):                          # both units compile side by side
            out = os.path.join(d, f"k{i}.s")
            cmd = f"{args.hipcc} {args.flags} {args.xflags if u['x'] else ''} -S --cuda-device-only {extra} -o {out} {u['src']}"
            procs.append((u, out, subprocess.Popen(cmd, shell=True, stderr=subprocess.DEVNULL)))
        for u, out, pr in procs:
            if pr.wait() != 0:
                raise SystemExit(f"compiling {u['src']} failed")
            lines = open(out).read().split("\n")
            found = sorted({m.group(1) for l in lines for m in [re.match(r"^(_Z\w+):", l)] if m and m.group(1).startswith(tuple(u["prefixes"]))})
            for pre in u["prefixes"]:
                if not any(k.startswith(pre) for k in found):
                    print("    (no instance of", pre, "found)")
                    bad += 1
            for k in u["kernels"] + found:
                n, hz = check(lines, k)
                print(f"{k}: {n} asm loads, {len(hz)} hazards")
                for kind, ld, use in hz[:10]:
                    print("   ", kind, "|", ld, "|", use)
                bad += len(hz)
                if n == 0:
                    # the training (SAVE) instances of nerf_mlp_f32x_kernel read LDS with compiler-scheduled loads on purpose
                    save_f32x = k.startswith("_Z20nerf_mlp_f32x_kernelI") and re.match(r"_Z20nerf_mlp_f32x_kernelILb[01]ELb1E", k)
                    print("    (compiler-scheduled instance: nothing to check)" if save_f32x else "    (no asm loads found: NERF_F32_ASM_LOADS off?)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
