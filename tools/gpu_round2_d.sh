#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 300 tools/mb/mb_bp_sweep 64 262144 8192 0 21 108 110 111 112 113 114 115 116 117
timeout 300 tools/mb/mb_bp_sweep 64 8192 262144 0 21 108 110 111 112 113 114 115 116 117
timeout 300 tools/mb/mb_bp_sweep 32 32768 8192 0 7 113 114 115 106
timeout 300 tools/mb/mb_bp_sweep 16 8192 4096 0 7 113 114 115
for v in 111 113; do SMK_BP_VARIANT=$v timeout 300 python tools/quick_parity.py 2>&1 | tail -1; done
} > gpurun_out/r2d_sweep.log 2>&1
cat gpurun_out/r2d_sweep.log
