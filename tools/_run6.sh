set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof_t6
NERF_DEAD_TILE_SKIP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t6 -- python3 bench.py --mode train --precision f32x --steps 10 --warmup 2 --no-dense-compare > gpurun_out/r03_t6_bench.log 2>&1
grep -a "^{" gpurun_out/r03_t6_bench.log | tail -1 | cut -c1-200
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_t6.log 2>&1 || { tail -40 gpurun_out/r03_t6.log; exit 1; }
tail -2 gpurun_out/r03_t6.log
