#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "== 2-term (3 products) fp32, k=64"
MB_NSPLIT=2 tools/mb/mb_bp_sweep 64 8192 262144 0 108 108 100 111 117 115
MB_NSPLIT=2 tools/mb/mb_bp_sweep 64 262144 8192 0 108 108 100 111 117 115
echo "== 2-term fp32, k=32 / k=16"
MB_NSPLIT=2 tools/mb/mb_bp_sweep 32 32768 8192 0 115 115 110 113
MB_NSPLIT=2 tools/mb/mb_bp_sweep 16 8192 4096 0 115 115 110 113
echo "== parity, default (6 products) and fast (3 products)"
timeout 300 python tools/quick_parity.py 2>&1 | tail -1
SMK_NSPLIT=2 timeout 300 python tools/quick_parity.py 2>&1 | tail -1
} > gpurun_out/r2f.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5 >> gpurun_out/r2f.log
python bench.py --workload c4s --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2f_bench_c4s.log 2>&1
SMK_NSPLIT=2 python bench.py --workload c4s --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2f_bench_c4s_fast.log 2>&1
python bench.py --workload c2 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2f_bench_c2.log 2>&1
cat gpurun_out/r2f.log
