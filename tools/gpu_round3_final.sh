#!/bin/bash
# round 3, last call: the whole GPU suite, the smoke test, the poisoned suite, and the rank sweep of profiles/r03_wide_rank_times.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03final; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/full_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke >> $OUT/full_gpu_tests.txt
for k in 80 100 128 160 192 256 384 512; do for alg in BPP MU HALS; do python3 tools/wide_run.py 16384 8192 $k $alg 12 1 2>/dev/null | tail -1; done; done > $OUT/r03_wide_rank_times.txt
python3 tools/wide_run.py 16384 8192 1024 BPP 2 1 2>/dev/null | tail -1 >> $OUT/r03_wide_rank_times.txt
bash tools/gpu_poison.sh > $OUT/poison.txt 2>&1
