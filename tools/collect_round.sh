set -e
PRECS="${PRECS-f32 f16 f16m32 f32x}" TRAIN_PRECS="${TRAIN_PRECS-f32 f32x}" timeout -k 10 1000 bash profiles/collect.sh > gpurun_out/r03_collect_all.log 2>&1
for P in ${PRECS-f32 f16 f16m32 f32x}; do python profiles/summarize.py gpurun_out/prof_$P r03_$P >> gpurun_out/r03_summarize.log 2>&1; done
for P in ${TRAIN_PRECS-f32 f32x}; do python profiles/summarize.py gpurun_out/prof_train_$P r03_train_$P >> gpurun_out/r03_summarize.log 2>&1; done
# the same training step with every tile computed (kernel quality without the scene's sparsity)
if [ -n "${TRAIN_PRECS-f32 f32x}" ]; then
  NERF_DEAD_TILE_SKIP=0 PRECS="" TRAIN_TAG=_dense TRAIN_PRECS="${TRAIN_PRECS-f32 f32x}" timeout -k 10 600 bash profiles/collect.sh >> gpurun_out/r03_collect_all.log 2>&1
  for P in ${TRAIN_PRECS-f32 f32x}; do python profiles/summarize.py gpurun_out/prof_train_${P}_dense r03_train_${P}_dense >> gpurun_out/r03_summarize.log 2>&1; done
fi
mkdir -p gpurun_out/profiles_r03 && cp profiles/r03_* profiles/traffic_*.json gpurun_out/profiles_r03/
python bench.py --steps 10 --warmup 3 > gpurun_out/profiles_r03/r03_bench_default_n1.json 2> gpurun_out/r03_bench_default.err
tail -3 gpurun_out/r03_summarize.log
