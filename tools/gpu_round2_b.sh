#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "== W'A shape (len 262144, ncols 8192), fp32 k=64"
timeout 300 tools/mb/mb_bp_sweep 64 262144 8192 0 21 100 101 102 103 104 105 106 107 108 109 13 22
echo "== H*At shape (len 8192, ncols 262144), fp32 k=64"
timeout 300 tools/mb/mb_bp_sweep 64 8192 262144 0 21 100 101 102 103 104 105 106 107 108 109
echo "== k=32 fp32 (len 32768 ncols 8192 and transposed)"
timeout 300 tools/mb/mb_bp_sweep 32 32768 8192 0 7 100 101 102 103 105 106 20
timeout 300 tools/mb/mb_bp_sweep 16 8192 4096 0 7 100 101 102 105
for v in 100 102 105; do SMK_BP_VARIANT_K64=$v SMK_BP_VARIANT=$v timeout 300 python tools/quick_parity.py 2>&1 | tail -1; done
} > gpurun_out/r2b_sweep.log 2>&1
timeout 600 python -m pytest tests/test_gpu_c5.py -x -q --durations=4 > gpurun_out/r2b_c5.log 2>&1
bash tools/prof_workload.sh c4s 6 > /dev/null 2>&1
head -12 gpurun_out/prof_c4s/kernel_stats.md > gpurun_out/r2b_prof_c4s.md
cat gpurun_out/r2b_sweep.log
tail -15 gpurun_out/r2b_c5.log
