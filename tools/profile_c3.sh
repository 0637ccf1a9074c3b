#!/bin/bash
# Regenerates the judged profile artefacts for the bench workload (C3) on the GPU box:
#   gpurun_out/prof/c3_kernel_stats.md          rocprofv3 --kernel-trace --stats summary
#   gpurun_out/prof/c3_results.db               the rocpd database behind it
#   gpurun_out/prof/pmc_fetch.db, pmc_write.db  FETCH_SIZE / WRITE_SIZE, one counter per pass
#   gpurun_out/prof/hbm_traffic.json            bytes per bigprod launch from the two passes
# Every profiled run is wrapped in `timeout` (a hung PMC pass once cost 15 GPU-minutes), the program
# follows `--` directly, and --pmc is never combined with other trace domains.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o c3 -- $BENCH > $OUT/bench_kt.log 2>&1
DB=$(find $OUT/kt -name '*.db' | head -1)
[ -n "$DB" ] && cp "$DB" $OUT/c3_results.db && python3 $ROOT/tools/prof_summary.py $OUT/c3_results.db $OUT/c3_kernel_stats.md > /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o f -- $BENCH > $OUT/bench_pf.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o w -- $BENCH > $OUT/bench_pw.log 2>&1
F=$(find $OUT/pf -name '*.db' | head -1); W=$(find $OUT/pw -name '*.db' | head -1)
[ -n "$F" ] && cp "$F" $OUT/pmc_fetch.db
[ -n "$W" ] && cp "$W" $OUT/pmc_write.db
[ -n "$F" ] && [ -n "$W" ] && python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch.db $OUT/pmc_write.db bigprod_kernel c3_n1 $OUT/hbm_traffic.json
rm -rf $OUT/kt $OUT/pf $OUT/pw
tail -2 $OUT/bench_kt.log | cut -c1-400
head -14 $OUT/c3_kernel_stats.md
