set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_train_steps.py -x -q -m gpu > gpurun_out/r03_t5.log 2>&1 || { tail -40 gpurun_out/r03_t5.log; exit 1; }
tail -2 gpurun_out/r03_t5.log
timeout -k 10 300 python3 bench.py --mode train --precision f32x --steps 20 --warmup 3 > gpurun_out/r03_t5_bench.log 2>&1
grep -a "^{" gpurun_out/r03_t5_bench.log | tail -1 | cut -c1-200
PRECS="" TRAIN_PRECS="f32x" timeout -k 10 500 bash profiles/collect.sh > gpurun_out/r03_t5_collect.log 2>&1
python profiles/summarize.py gpurun_out/prof_train_f32x t5_train_f32x > gpurun_out/r03_t5_sum.log 2>&1
mkdir -p gpurun_out/t5 && cp profiles/t5_* gpurun_out/t5/
head -12 profiles/t5_train_f32x_sq_summary.csv | cut -c1-140
