#!/bin/bash
# MFMA utilisation from counters: one --pmc pass per workload -> profiles/mfma_util.json (via gpurun_out/r05)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cp $ROOT/profiles/mfma_util.json $OUT/mfma_util.json 2>/dev/null
pmc() {  # name, steps, workload, kernel needle, key
  local name=$1 steps=$2 wl=$3 needle=$4 key=$5
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_$name -o x -- python3 $ROOT/bench.py --no-cpu-baseline --workload $wl --steps $steps --warmup 2 > $OUT/pmc_${name}.log 2>&1
  local DB=$(find $OUT/pmc_$name -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/pmc_mfma.py "$DB" "$needle" $key $OUT/mfma_util.json > $OUT/r05_mfma_${name}.txt
  rm -rf $OUT/pmc_$name
}
pmc c4 3 c4 bigprod_f3_kernel c4_n1
pmc c3 10 c3 "bigprod_kernel<" c3_n1
pmc c2 50 c2 bigprod_f3 c2_n1
