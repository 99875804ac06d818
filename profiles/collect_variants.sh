#!/bin/bash
# What bounds the fp16 kernel: busy x clock of timing-only build variants (tools/ab_bench.py builds them with
# -DNERF_TIMING_BUILD; they compute wrong results and are never loaded by the product).  One rocprofv3 process per
# variant, so every nerf_mlp_f16_kernel dispatch in its output belongs to that variant.
#   bash profiles/collect_variants.sh   -> gpurun_out/var_<name>/{trace,pmc}
#   python profiles/summarize_variants.py gpurun_out r02_f16_variants
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
PREC=${PREC:-f16}                      # f16 (default) or f32x
declare -A FLAGS=( [base]="" [noadv]="-DNERF_F16_HACK_NOADV=1" [norelu]="-DNERF_F16_HACK_NORELU=1" [noepi]="-DNERF_F16_HACK_NOEPI=1"
                   [noboth]="-DNERF_F16_HACK_NOADV=1 -DNERF_F16_HACK_NOEPI=1" [nobar]="-DNERF_F16_HACK_NOBARRIER=1"
                   [xnoadv]="-DNERF_F32X_HACK_NOADV=1" [xnoepi]="-DNERF_F32X_HACK_NOEPI=1" [xnope]="-DNERF_F32X_HACK_NOPE=1" )
for V in ${VARIANTS:-base noadv norelu noepi noboth}; do
  D=gpurun_out/var_$V; rm -rf $D; mkdir -p $D
  rm -f nerf_replication_amd/csrc/variants/lib_$V.so          # always rebuilt here, with the flags this script states
  AB_BUILD_ONLY=1 python3 tools/ab_bench.py $PREC "$V:${FLAGS[$V]}" > $D/build.log 2>&1
  AB_ROUNDS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 tools/ab_bench.py $PREC $V: > $D/trace.log 2>&1
  AB_ROUNDS=3 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $D/pmc -- python3 tools/ab_bench.py $PREC $V: > $D/pmc.log 2>&1
  tail -1 $D/trace.log
done
