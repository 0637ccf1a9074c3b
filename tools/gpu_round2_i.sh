#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_nnls.py -x -q 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_variants.py -x -q 2>&1 | tail -12
} > gpurun_out/r2i.log 2>&1
cat gpurun_out/r2i.log
