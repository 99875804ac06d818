set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_train_steps.py -x -q -m gpu > gpurun_out/r03_t2.log 2>&1 || { tail -40 gpurun_out/r03_t2.log; exit 1; }
tail -3 gpurun_out/r03_t2.log
mkdir -p gpurun_out/prof_t2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t2 -- python3 bench.py --mode train --precision f32x --steps 10 --warmup 2 --no-dense-compare > gpurun_out/r03_t2_bench.log 2>&1
timeout -k 10 300 python3 bench.py --mode train --precision f32x --steps 20 --warmup 3 > gpurun_out/r03_t2_bench2.log 2>&1
grep -a "^{" gpurun_out/r03_t2_bench2.log | tail -1 | cut -c1-900
