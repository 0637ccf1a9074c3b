#!/bin/bash
# round 3, call z11: trailing tiles two at a time (workgroup per column)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03z11; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_nnls.py tests/test_gpu_parity.py -m gpu -x -q -k "above or not_positive or hard" 2>&1 | grep -E "passed|failed|error" | tail -2 > $OUT/tests.txt
python3 tools/fuzz_wide_bpp.py 50 51 2>&1 | tail -1 > $OUT/fuzz.txt
for k in 192 256 384 512; do python3 tools/wide_run.py 16384 8192 $k BPP 12 1 2>/dev/null | tail -1 >> $OUT/times.txt; done
python3 tools/wide_run.py 16384 8192 1024 BPP 2 1 2>/dev/null | tail -1 >> $OUT/times.txt
