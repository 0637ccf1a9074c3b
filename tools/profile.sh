#!/bin/bash
# The judged artefacts of a round, regenerated on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   bash tools/profile.sh [tag, default r04] [quick]      quick: bench lines, the C4 / C3 / C2 / C5 kernel tables and the HBM traffic passes only
#   bench JSON lines: C4 (the default bench, exactly as the driver runs it), C3, C2, a C4 shard, rank 0 of 2 / 4 / 8 emulated
#   kernel tables (rocprofv3 --kernel-trace --stats): C4 whole, C3, C2, rank-0-of-8, the C5-shaped HierNMF2 run, a root-sized
#     RANK2 iteration, block pivoting at k = 192 and k = 512 (the general path) + its times per iteration beside MU's
#   HBM traffic of the streaming kernels (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, C4 and C3) -> hbm_traffic.json
#     with the hash of the kernel source it was measured on
#   counter passes for the kernels the round-2 review had no counter evidence for: the rank-2 gather product, the k = 64
#     NNLS (nnls_bpp_inv_kernel) and the fused HALS W sweep (FETCH_SIZE, WRITE_SIZE, SQ instruction / wait / LDS counters)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
QUICK=${2:-}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
kt() {   # name, command...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o x -- "$@" > $OUT/${name}_run.log 2>&1
  local DB=$(find $OUT/kt_$name -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/${TAG}_${name}_kernel_stats.md > /dev/null
  [ "$name" = "c4full_bpp_f32" ] && [ -n "$DB" ] && cp "$DB" $OUT/${TAG}_c4full_bpp_f32_rocprofv3.db
  rm -rf $OUT/kt_$name
}
pmc() {  # name, counters, command...
  local name=$1 ctr=$2; shift 2
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_$name -o x -- "$@" > $OUT/pmc_${name}.log 2>&1
  local DB=$(find $OUT/pmc_$name -name '*.db' | head -1)
  [ -n "$DB" ] && cp "$DB" $OUT/pmc_$name.db
  rm -rf $OUT/pmc_$name
}
B="python3 $ROOT/bench.py --no-cpu-baseline"
cd $ROOT
# ---- bench lines (un-profiled) ----
python3 $ROOT/bench.py 2> $OUT/bench_c4.err | tail -1 > $OUT/${TAG}_bench_c4.json
python3 $ROOT/bench.py --workload c3 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c3.json
python3 $ROOT/bench.py --workload c2 --steps 200 --warmup 20 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c2.json
$B --workload c4s --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c4s.json
for n in 2 4 8; do $B --emulate-world $n 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c4_emulate$n.json; done
# ---- kernel tables ----
kt c4full_bpp_f32 $B --workload c4 --steps 5 --warmup 2
kt c3_hals_bf16 $B --workload c3 --steps 20 --warmup 3
kt c2_bpp_f32 $B --workload c2 --steps 50 --warmup 5
kt c4_rank0_of_8 $B --emulate-world 8 --steps 10 --warmup 3
kt c5_hiernmf2_1M python3 $ROOT/tools/c5_hier.py 1000000 16 8
SMK_CLUST_TIMING=1 SMK_R2P_PROFILE=1 python3 $ROOT/tools/c5_hier.py 1000000 16 8 2>&1 | grep "smk_clust\|hier_nmf2\|purity\|r2p" > $OUT/${TAG}_c5_hiernmf2_1M_timing.txt
if [ "$QUICK" != "quick" ]; then
kt rank2_iteration_1M python3 $ROOT/tools/r2_iter.py 1000000 16 30
kt wide_bpp_k192 python3 $ROOT/tools/wide_run.py 16384 8192 192 BPP 12 1
kt wide_bpp_k512 python3 $ROOT/tools/wide_run.py 16384 8192 512 BPP 12 1
for k in 100 160 192 256 384 512; do (cd $ROOT && for alg in BPP MU HALS; do python3 tools/wide_run.py 16384 8192 $k $alg 12 1 2>/dev/null | tail -1; done) >> $OUT/${TAG}_wide_rank_times.txt; done
(cd $ROOT && SMK_NSPLIT=8 python3 bench.py --workload c4s --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1) > $OUT/${TAG}_bench_c4s_accurate_form.json
fi
# ---- HBM traffic of the streaming kernels (separate passes) ----
pmc c4_fetch FETCH_SIZE $B --workload c4 --steps 3 --warmup 1
pmc c4_write WRITE_SIZE $B --workload c4 --steps 3 --warmup 1
pmc c3_fetch FETCH_SIZE $B --workload c3 --steps 10 --warmup 2
pmc c3_write WRITE_SIZE $B --workload c3 --steps 10 --warmup 2
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_c4_fetch.db $OUT/pmc_c4_write.db bigprod_f3 c4_n1 $OUT/hbm_traffic.json > /dev/null
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_c3_fetch.db $OUT/pmc_c3_write.db bigprod_kernel c3_n1 $OUT/hbm_traffic.json > /dev/null
cp $OUT/pmc_c4_fetch.db $OUT/${TAG}_c4_pmc_fetch_size.db; cp $OUT/pmc_c4_write.db $OUT/${TAG}_c4_pmc_write_size.db
[ "$QUICK" = "quick" ] && exit 0
# ---- counters for the rank-2 gather product, the k = 64 NNLS and the fused HALS W sweep ----
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQB="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
pmc r2_fetch FETCH_SIZE python3 $ROOT/tools/r2_iter.py 1000000 16 10
pmc r2_write WRITE_SIZE python3 $ROOT/tools/r2_iter.py 1000000 16 10
pmc r2_sqa "$SQA" python3 $ROOT/tools/r2_iter.py 1000000 16 10
pmc r2_sqb "$SQB" python3 $ROOT/tools/r2_iter.py 1000000 16 10
pmc c4s_fetch FETCH_SIZE $B --workload c4s --steps 5 --warmup 2
pmc c4s_write WRITE_SIZE $B --workload c4s --steps 5 --warmup 2
pmc c4s_sqa "$SQA" $B --workload c4s --steps 5 --warmup 2
pmc c4s_sqb "$SQB" $B --workload c4s --steps 5 --warmup 2
pmc c3_sqa "$SQA" $B --workload c3 --steps 10 --warmup 2
pmc c3_sqb "$SQB" $B --workload c3 --steps 10 --warmup 2
{
  echo "# counter passes (rocprofv3 --kernel-trace --pmc ..., one group per run; averages per launch)"
  for tag in r2 c4s c3; do
    case $tag in r2) needle=spmm;; c4s) needle=nnls_bpp_inv;; c3) needle=hals_w_fused;; esac
    echo; echo "## $needle  (workload: $tag)"
    for f in $OUT/pmc_${tag}_fetch.db $OUT/pmc_${tag}_write.db $OUT/pmc_${tag}_sqa.db $OUT/pmc_${tag}_sqb.db; do
      [ -f $f ] && python3 $ROOT/tools/pmc_dump.py $f $needle
    done
  done
} > $OUT/${TAG}_small_kernel_counters.txt 2>&1
rm -f $OUT/pmc_*.db $OUT/*_run.log $OUT/pmc_*.log
ls -la $OUT
for f in $OUT/${TAG}_bench_*.json; do echo $f; python3 -c "
import json,sys
j=json.loads(open('$f').read()); r=j['roofline']; print('  it/s %.2f ms/step %.4f bigprod %.4f ms %.0f GB/s frac %.3f windows %d traffic %s'%(j['value'],j['ms_per_step'],r['avg_launch_ms'],r['achieved'],r['frac'],j['windows'],r.get('traffic')))
if 'cpu_baseline' in j: print('  cpu', j['cpu_baseline']['value'], j['cpu_baseline']['sample_ms'])"; done
cat $OUT/${TAG}_c5_hiernmf2_1M_timing.txt | tail -3
