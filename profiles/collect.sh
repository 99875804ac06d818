#!/bin/bash
# Re-collect the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   bash profiles/collect.sh                 -> gpurun_out/prof_<prec>/{trace,pmc_sq,pmc_fetch,pmc_write}         (render, per precision)
#                                               gpurun_out/prof_train_<prec>/{trace,pmc_sq,pmc_fetch,pmc_write}    (BASELINE configs[2])
#   python profiles/summarize.py gpurun_out/prof_<prec> r03_<prec>
#   python profiles/summarize.py gpurun_out/prof_train_<prec> r03_train_<prec>
# Counters are collected in their own passes (never together with trace domains), FETCH_SIZE and
# WRITE_SIZE separately (TCC slot limit), as MI355X_MICROARCH.md prescribes.
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"
for P in ${PRECS-f32 f16 f32x}; do
  D=gpurun_out/prof_$P; mkdir -p $D
  A="--cpu-sample 0 --no-extras --no-full-network-compare --precision $P"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 bench.py --steps 3 --warmup 1 $A > $D/bench_trace.log 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d $D/pmc_sq -- python3 bench.py --steps 1 --warmup 0 $A > $D/bench_pmc_sq.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 $A > $D/bench_pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- python3 bench.py --steps 1 --warmup 0 $A > $D/bench_pmc_write.log 2>&1
  tail -1 $D/bench_trace.log | cut -c1-200
done
for P in ${TRAIN_PRECS-f32 f32x}; do
  D=gpurun_out/prof_train_$P${TRAIN_TAG-}; mkdir -p $D      # TRAIN_TAG=_dense with NERF_DEAD_TILE_SKIP=0 in the environment: every tile computed
  A="--mode train --precision $P --no-dense-compare"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 bench.py --steps 10 --warmup 2 $A > $D/bench_trace.log 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d $D/pmc_sq -- python3 bench.py --steps 2 --warmup 0 $A > $D/bench_pmc_sq.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- python3 bench.py --steps 2 --warmup 0 $A > $D/bench_pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- python3 bench.py --steps 2 --warmup 0 $A > $D/bench_pmc_write.log 2>&1
  tail -1 $D/bench_trace.log | cut -c1-200
done
