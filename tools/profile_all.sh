#!/bin/bash
# One entry point for the measurements of a round (replaces tools/gpu_round4.sh and the six tools/gpu_round5_*.sh scripts).
#   bash tools/profile_all.sh <stage> [tag, default r06]     writes into gpurun_out/<tag>/ ; copy what is to be judged into profiles/
# stages:
#   evidence   passive-set histograms + device counters of block pivoting (tools/nnls_sets.py), PMC passes on the NNLS kernels
#   bench      the bench lines of the round (C4 default, C3, C2, the sparse workloads, the checked variants, --api-path)
#   tables     rocprofv3 --kernel-trace --stats kernel tables (C4 whole, C3, C2, s_1m, s_reuters)
#   traffic    separate --pmc FETCH_SIZE / WRITE_SIZE passes -> hbm_traffic.json entries (C4, C3, s_1m)
#   mfma       MFMA utilisation from counters -> mfma_util.json (C4, C3, C2)
#   suite      pytest -m gpu
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
STAGE=${1:-bench}
TAG=${2:-r06}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline"
kt() {   # name, command...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o x -- "$@" > $OUT/${name}_run.log 2>&1
  local DB=$(find $OUT/kt_$name -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/${TAG}_${name}_kernel_stats.md > /dev/null
  rm -rf $OUT/kt_$name
}
pmc() {  # name, counters, command...   (counters in their own run: --kernel-trace + --pmc only)
  local name=$1 ctr=$2; shift 2
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_$name -o x -- "$@" > $OUT/pmc_${name}.log 2>&1
  local DB=$(find $OUT/pmc_$name -name '*.db' | head -1)
  [ -n "$DB" ] && cp "$DB" $OUT/pmc_$name.db
  rm -rf $OUT/pmc_$name
}
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQB="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
SQC="SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"

case $STAGE in
evidence)
  cd $ROOT
  for w in s_1m s_reuters c2 c4s_uniform c4s_planted mid32_uniform mid32_planted; do
    timeout 600 python3 tools/nnls_sets.py $w 20 > $OUT/${TAG}_nnls_sets_$w.txt 2>&1
  done
  cd /tmp
  for w in s_1m c4s c2 s_reuters; do
    case $w in s_1m|s_reuters) steps="--steps 10 --warmup 3";; c4s) steps="--steps 5 --warmup 2";; c2) steps="--steps 50 --warmup 5";; esac
    pmc ${w}_sqa "$SQA" $B --workload $w $steps
    pmc ${w}_sqb "$SQB" $B --workload $w $steps
    pmc ${w}_sqc "$SQC" $B --workload $w $steps
    pmc ${w}_fetch FETCH_SIZE $B --workload $w $steps
    pmc ${w}_write WRITE_SIZE $B --workload $w $steps
  done
  {
    echo "# block-pivoting kernels: counter passes (rocprofv3 --kernel-trace --pmc <group>, one group per run; averages per launch)"
    for w in s_1m c4s c2 s_reuters; do
      echo; echo "## workload $w"
      for g in sqa sqb sqc fetch write; do
        [ -f $OUT/pmc_${w}_$g.db ] && python3 $ROOT/tools/pmc_dump.py $OUT/pmc_${w}_$g.db nnls_bpp
      done
    done
  } > $OUT/${TAG}_nnls_counters_raw.txt 2>&1
  # s_1m gather product traffic
  python3 $ROOT/tools/pmc_dump.py $OUT/pmc_s_1m_fetch.db spmm > $OUT/${TAG}_s_1m_spmm_fetch.txt 2>&1
  python3 $ROOT/tools/pmc_dump.py $OUT/pmc_s_1m_write.db spmm >> $OUT/${TAG}_s_1m_spmm_fetch.txt 2>&1
  rm -f $OUT/pmc_*.db $OUT/pmc_*.log
  ;;
bench)
  cd $ROOT
  python3 bench.py 2> $OUT/bench_c4.err | tail -1 > $OUT/${TAG}_bench_c4.json
  python3 bench.py --workload c3 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c3.json
  python3 bench.py --workload c2 --steps 200 --warmup 20 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c2.json
  for w in s_reuters s_reuters_hals s_1m; do
    case $w in s_1m) steps="--steps 20 --warmup 3";; *) steps="--steps 200 --warmup 20";; esac
    python3 bench.py --workload $w $steps 2>/dev/null | tail -1 > $OUT/${TAG}_bench_$w.json
  done
  $B --workload c4 --check-every-iteration 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c4_checked.json
  $B --workload c3 --steps 20 --warmup 3 --check-every-iteration 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c3_checked.json
  $B --workload c2 --steps 200 --warmup 20 --check-every-iteration 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c2_checked.json
  $B --workload s_reuters --steps 200 --warmup 20 --check-every-iteration 2>/dev/null | tail -1 > $OUT/${TAG}_bench_s_reuters_checked.json
  for n in 2 4 8; do $B --emulate-world $n 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c4_emulate$n.json; done
  for w in c2 c3 c4s; do python3 bench.py --no-cpu-baseline --api-path --workload $w 2>/dev/null | tail -1 > $OUT/${TAG}_bench_${w}_api_path.json; done
  ;;
tables)
  kt c4full_bpp_f32 $B --workload c4 --steps 5 --warmup 2
  kt c3_hals_bf16 $B --workload c3 --steps 20 --warmup 3
  kt c2_bpp_f32 $B --workload c2 --steps 50 --warmup 5
  kt s_1m $B --workload s_1m --steps 10 --warmup 3
  kt s_reuters $B --workload s_reuters --steps 50 --warmup 5
  kt s_reuters_hals $B --workload s_reuters_hals --steps 50 --warmup 5
  kt c4_rank0_of_8 $B --emulate-world 8 --steps 10 --warmup 3
  ;;
traffic)
  pmc c4_fetch FETCH_SIZE $B --workload c4 --steps 3 --warmup 1
  pmc c4_write WRITE_SIZE $B --workload c4 --steps 3 --warmup 1
  pmc c3_fetch FETCH_SIZE $B --workload c3 --steps 10 --warmup 2
  pmc c3_write WRITE_SIZE $B --workload c3 --steps 10 --warmup 2
  pmc s_1m_fetch FETCH_SIZE $B --workload s_1m --steps 5 --warmup 2
  pmc s_1m_write WRITE_SIZE $B --workload s_1m --steps 5 --warmup 2
  for w in s_reuters s_reuters_hals; do
    pmc ${w}_fetch FETCH_SIZE $B --workload $w --steps 50 --warmup 5
    pmc ${w}_write WRITE_SIZE $B --workload $w --steps 50 --warmup 5
  done
  cp $ROOT/profiles/hbm_traffic.json $OUT/hbm_traffic.json 2>/dev/null
  python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_c4_fetch.db $OUT/pmc_c4_write.db bigprod_f3 c4_n1 $OUT/hbm_traffic.json > /dev/null
  python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_c3_fetch.db $OUT/pmc_c3_write.db bigprod_kernel c3_n1 $OUT/hbm_traffic.json > /dev/null
  python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_s_1m_fetch.db $OUT/pmc_s_1m_write.db spmm_ s_1m_n1 $OUT/hbm_traffic.json > /dev/null
  # (the Reuters shape: the main gather kernel only; its long-column fix-up launch moves a few hundred KB more)
  python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_s_reuters_fetch.db $OUT/pmc_s_reuters_write.db spmm_seg_kernel s_reuters_n1 $OUT/hbm_traffic.json > /dev/null
  python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_s_reuters_hals_fetch.db $OUT/pmc_s_reuters_hals_write.db spmm_seg_kernel s_reuters_hals_n1 $OUT/hbm_traffic.json > /dev/null
  rm -f $OUT/pmc_*.db $OUT/pmc_*.log
  ;;
mfma)
  # MFMA utilisation from counters (north_star asks for it by name): one --pmc pass per workload -> mfma_util.json, keyed to bigprod.hip
  cp $ROOT/profiles/mfma_util.json $OUT/mfma_util.json 2>/dev/null
  mf() {  # name, steps, workload, kernel needle, key
    local name=$1 steps=$2 wl=$3 needle=$4 key=$5
    timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_$name -o x -- python3 $ROOT/bench.py --no-cpu-baseline --workload $wl --steps $steps --warmup 2 > $OUT/pmc_${name}.log 2>&1
    local DB=$(find $OUT/pmc_$name -name '*.db' | head -1)
    [ -n "$DB" ] && python3 $ROOT/tools/pmc_mfma.py "$DB" "$needle" $key $OUT/mfma_util.json > $OUT/${TAG}_mfma_${name}.txt
    rm -rf $OUT/pmc_$name $OUT/pmc_${name}.log
  }
  mf c4 3 c4 bigprod_f3_kernel c4_n1
  mf c3 10 c3 "bigprod_kernel<" c3_n1
  mf c2 50 c2 bigprod_f3 c2_n1
  ;;
suite)
  cd $ROOT
  timeout 3000 python3 -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_suite.txt 2>&1
  tail -5 $OUT/${TAG}_gpu_suite.txt
  ;;
*) echo "unknown stage $STAGE"; exit 2;;
esac
ls -la $OUT | tail -40
