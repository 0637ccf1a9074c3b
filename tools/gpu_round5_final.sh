#!/bin/bash
# Last step of round 5: the counter files keyed to the final bigprod.hip (HBM traffic, MFMA utilisation) and the bench lines that
# changed with the last kernels -> gpurun_out/r05/ (copy into profiles/)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline"
pmc() {  # name, counters, command...
  local name=$1 ctr=$2; shift 2
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_$name -o x -- "$@" > $OUT/pmc_${name}.log 2>&1
  local DB=$(find $OUT/pmc_$name -name '*.db' | head -1)
  [ -n "$DB" ] && cp "$DB" $OUT/pmc_$name.db
  rm -rf $OUT/pmc_$name
}
cp $ROOT/profiles/hbm_traffic.json $OUT/hbm_traffic.json 2>/dev/null
pmc c4_fetch FETCH_SIZE $B --workload c4 --steps 3 --warmup 1
pmc c4_write WRITE_SIZE $B --workload c4 --steps 3 --warmup 1
pmc c3_fetch FETCH_SIZE $B --workload c3 --steps 10 --warmup 2
pmc c3_write WRITE_SIZE $B --workload c3 --steps 10 --warmup 2
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_c4_fetch.db $OUT/pmc_c4_write.db bigprod_f3 c4_n1 $OUT/hbm_traffic.json > /dev/null
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_c3_fetch.db $OUT/pmc_c3_write.db "bigprod_kernel<" c3_n1 $OUT/hbm_traffic.json > /dev/null
rm -f $OUT/pmc_c4_fetch.db $OUT/pmc_c4_write.db $OUT/pmc_c3_fetch.db $OUT/pmc_c3_write.db
bash $ROOT/tools/gpu_round5_mfma.sh > /dev/null 2>&1
cp $OUT/hbm_traffic.json $ROOT/profiles/hbm_traffic.json; cp $OUT/mfma_util.json $ROOT/profiles/mfma_util.json      # so that the lines below carry them
cd $ROOT
python3 bench.py 2> $OUT/bench_c4.err | tail -1 > $OUT/r05_bench_c4.json
python3 bench.py --workload c3 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/r05_bench_c3.json
python3 bench.py --workload c2 --steps 200 --warmup 20 2>/dev/null | tail -1 > $OUT/r05_bench_c2.json
$B --single-copy 2>/dev/null | tail -1 > $OUT/r05_bench_c4_single_copy.json
$B --single-copy --workload c4x2 --steps 10 --warmup 2 2>/dev/null | tail -1 > $OUT/r05_bench_c4x2_single_copy.json
$B --single-copy --workload c3 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/r05_bench_c3_single_copy.json
$B --single-copy --emulate-world 8 2>/dev/null | tail -1 > $OUT/r05_bench_c4_single_copy_emulate8.json
for f in c4 c3 c2 c4_single_copy c4x2_single_copy c3_single_copy c4_single_copy_emulate8; do python3 -c "
import json
d=json.load(open('$OUT/r05_bench_$f.json')); r=d['roofline']
print('$f', round(d['value'],2), 'it/s', round(d['ms_per_step'],4), 'ms  frac', round(r['frac'],3), 'traffic', r.get('traffic'), 'mfma', r.get('mfma_busy_frac'), 'passes', round(r['pass_WtA_ms'],4), round(r['pass_HAt_ms'],4))"; done
