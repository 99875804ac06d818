set -e
PRECS="" TRAIN_PRECS="f32x" timeout -k 10 500 bash profiles/collect.sh > gpurun_out/r03_t4_collect.log 2>&1
python profiles/summarize.py gpurun_out/prof_train_f32x t4_train_f32x > gpurun_out/r03_t4_sum.log 2>&1
mkdir -p gpurun_out/t4 && cp profiles/t4_* gpurun_out/t4/
cat profiles/t4_train_f32x_sq_summary.csv | cut -c1-140
