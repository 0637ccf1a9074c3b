#!/bin/bash
# round 3, call z2: tile kernels at k in (32, 64] (SMK_NNLS_TILE128=2) against nnls_bpp_inv_kernel<64> on a C4 shard and at k = 48
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03z2; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
for t in 1 2 1 2; do
  SMK_NNLS_TILE128=$t python3 bench.py --workload c4s --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('level=$t c4s', d['value'], 'it/s', d['ms_per_step'], 'ms')" >> $OUT/times.txt
  for k in 48 64; do SMK_NNLS_TILE128=$t python3 tools/wide_run.py 16384 8192 $k BPP 12 1 2>/dev/null | tail -1 | sed "s/^/level=$t /" >> $OUT/times.txt; done
done
