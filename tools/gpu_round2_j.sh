#!/bin/bash
# fp16 two-term streaming product: accuracy + rate against the bf16 forms, then solver parity
cd /root/repo
mkdir -p gpurun_out
{
for ns in 3 2 4; do
  MB_NSPLIT=$ns timeout 300 tools/mb/mb_bp_sweep 64 262144 8192 0 108 115 111 110
  MB_NSPLIT=$ns timeout 300 tools/mb/mb_bp_sweep 64 8192 262144 0 108 115 111 110
  MB_NSPLIT=$ns timeout 300 tools/mb/mb_bp_sweep 32 32768 8192 0 115 108 111
done
for ns in 3 4; do
  SMK_NSPLIT=$ns timeout 600 python tools/quick_parity.py
done
SMK_NSPLIT=4 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
SMK_NSPLIT=4 timeout 600 python bench.py --workload c4s --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -2
SMK_NSPLIT=3 timeout 600 python bench.py --workload c4s --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -2
} > gpurun_out/r2j.log 2>&1
grep -v "^\[" gpurun_out/r2j.log | tail -60
