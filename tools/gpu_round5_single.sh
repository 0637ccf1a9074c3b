#!/bin/bash
# Single copy of A: C3 with and without the stored transpose, kernel shapes of the transposed source, LDS counters
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd $ROOT
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print("%8.1f it/s  %.4f ms/step  W^T A pass %.1f us  H A^T pass %.1f us" % (d["value"], d["ms_per_step"], r["pass_WtA_ms"]*1e3, r["pass_HAt_ms"]*1e3))'
{
echo "# C3 (65536 x 16384, k = 32, HALS, bf16): stored transpose vs single copy, by kernel shape of the transposed source"
echo -n "stored transpose          : "; python3 bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "$pick"
for v in 6 15 16 17; do
  echo -n "single copy, TR variant $v : "; SMK_BP_TR_VARIANT=$v python3 bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline --single-copy 2>/dev/null | python3 -c "$pick"
done
} > $OUT/r05_single_copy_c3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for sc in; do
  flag=""; [ $sc = single ] && flag="--single-copy"
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $OUT/pmc_sc_$sc -o x -- python3 $ROOT/bench.py --no-cpu-baseline --workload c3 --steps 5 --warmup 2 $flag > $OUT/pmc_sc_$sc.log 2>&1
  DB=$(find $OUT/pmc_sc_$sc -name '*.db' | head -1)
  [ -n "$DB" ] && python3 - "$DB" >> $OUT/r05_single_copy_c3.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%bigprod_kernel%' group by kernel_name, counter_name").fetchall()
for k, c, v, n in rows:
    print(f"{k[:60]:60s} {c:26s} avg {v:16.1f} (n={n})")
PY
  rm -rf $OUT/pmc_sc_$sc
done
cd $ROOT
echo "# default shape, bench lines" >> $OUT/r05_single_copy_c3.txt
python3 bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline --single-copy 2>/dev/null | tail -1 > $OUT/r05_bench_c3_single_copy.json
python3 -c "$pick" < $OUT/r05_bench_c3_single_copy.json >> $OUT/r05_single_copy_c3.txt
