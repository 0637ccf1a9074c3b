#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for v in 21 100 102 108 106; do
  SMK_BP_VARIANT=$v timeout 120 tools/mb/mb_bigprod 64 262144 8192 3 0
  SMK_BP_VARIANT=$v timeout 120 tools/mb/mb_bigprod 64 8192 262144 3 0
done
} > gpurun_out/r2c_phase.log 2>&1
timeout 600 python -m pytest tests/test_gpu_c5.py -x -q --durations=4 > gpurun_out/r2c_c5.log 2>&1
cat gpurun_out/r2c_phase.log; tail -12 gpurun_out/r2c_c5.log
