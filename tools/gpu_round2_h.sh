#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q --durations=5 2>&1 | tail -25 > gpurun_out/r2h.log
timeout 600 python -m pytest tests/test_gpu_nnls.py -x -q 2>&1 | tail -3 >> gpurun_out/r2h.log
python bench.py --workload c4s --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r2h_bench_c4s.json
bash tools/prof_workload.sh c4s 6 > /dev/null 2>&1
head -12 gpurun_out/prof_c4s/kernel_stats.md >> gpurun_out/r2h.log
cat gpurun_out/r2h.log
