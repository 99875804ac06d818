set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_train_steps.py -x -q -m gpu > gpurun_out/r03_t7.log 2>&1 || { tail -40 gpurun_out/r03_t7.log; exit 1; }
tail -2 gpurun_out/r03_t7.log
mkdir -p gpurun_out/prof_t7 gpurun_out/prof_t7d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t7 -- python3 bench.py --mode train --precision f32x --steps 10 --warmup 2 --no-dense-compare > gpurun_out/r03_t7_bench.log 2>&1
grep -a "^{" gpurun_out/r03_t7_bench.log | tail -1 | cut -c1-200
NERF_DEAD_TILE_SKIP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t7d -- python3 bench.py --mode train --precision f32x --steps 10 --warmup 2 --no-dense-compare > gpurun_out/r03_t7d_bench.log 2>&1
grep -a "^{" gpurun_out/r03_t7d_bench.log | tail -1 | cut -c1-200
