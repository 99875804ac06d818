set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof_t3
export NERF_DEAD_TILE_SKIP=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t3 -- python3 bench.py --mode train --precision f32x --steps 10 --warmup 2 --no-dense-compare > gpurun_out/r03_t3_bench.log 2>&1
grep -a "^{" gpurun_out/r03_t3_bench.log | tail -1 | cut -c1-300
