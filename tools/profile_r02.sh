#!/bin/bash
# Round-2 judged artefacts, regenerated on the GPU box into gpurun_out/r02/ (copied to profiles/ afterwards):
#   kernel tables (rocprofv3 --kernel-trace --stats) for C2, C3, a C4 shard, C4 whole, the C5-shaped HierNMF2 run
#   bench JSON lines for the same workloads
#   HBM traffic of the dominant kernels from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (C3, C4 shard)
#   SQ counters of the k = 64 fp32 streaming kernel, round-1 kernel (variant 21) vs round-2 kernel (variant 108)
#   clock / power samples while the k = 64 fp32 streaming kernel runs (package power cap evidence)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
kt() {   # name, command...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o x -- "$@" > $OUT/${name}_run.log 2>&1
  local DB=$(find $OUT/kt_$name -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r02_${name}_kernel_stats.md > /dev/null
  [ "$name" = "c3_hals_bf16" ] && [ -n "$DB" ] && cp "$DB" $OUT/r02_c3_hals_bf16_rocprofv3.db
  [ "$name" = "c4full_bpp_f32" ] && [ -n "$DB" ] && cp "$DB" $OUT/r02_c4full_bpp_f32_rocprofv3.db
  rm -rf $OUT/kt_$name
}
pmc() {  # name, counters, kernel-substring, command...
  local name=$1 ctr=$2; shift 2
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_$name -o x -- "$@" > $OUT/pmc_${name}.log 2>&1
  local DB=$(find $OUT/pmc_$name -name '*.db' | head -1)
  [ -n "$DB" ] && cp "$DB" $OUT/pmc_$name.db
  rm -rf $OUT/pmc_$name
}
B="python3 $ROOT/bench.py --no-cpu-baseline"
# ---- bench lines (un-profiled) ----
for w in c3 c2 c4s c4; do
  # c4 = the default bench (run exactly as the driver runs it: no flags); c3 keeps its CPU baseline too
  if [ $w = c4 ]; then python3 $ROOT/bench.py 2> $OUT/bench_$w.err | tail -1 > $OUT/r02_bench_$w.json
  else python3 $ROOT/bench.py --workload $w --steps 20 --warmup 3 $([ $w = c3 ] || echo --no-cpu-baseline) 2> $OUT/bench_$w.err | tail -1 > $OUT/r02_bench_$w.json; fi
done
SMK_NSPLIT=2 python3 $ROOT/bench.py --workload c4s --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_bench_c4s_fast2term.json
SMK_NSPLIT=3 python3 $ROOT/bench.py --workload c4s --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_bench_c4s_bf16x3.json
# ---- kernel tables ----
kt c3_hals_bf16 $B --workload c3 --steps 20 --warmup 3
kt c2_bpp_f32 $B --workload c2 --steps 50 --warmup 5
kt c4shard_bpp_f32 $B --workload c4s --steps 10 --warmup 3
kt c4full_bpp_f32 $B --workload c4 --steps 5 --warmup 2
cd $ROOT && kt c5_hiernmf2_1M python3 $ROOT/tools/c5_hier.py 1000000 16 8; cd /tmp
# ---- HBM traffic (separate passes) ----
pmc c3_fetch FETCH_SIZE $B --workload c3 --steps 10 --warmup 2
pmc c3_write WRITE_SIZE $B --workload c3 --steps 10 --warmup 2
pmc c4s_fetch FETCH_SIZE $B --workload c4s --steps 5 --warmup 2
pmc c4s_write WRITE_SIZE $B --workload c4s --steps 5 --warmup 2
pmc c4_fetch FETCH_SIZE $B --workload c4 --steps 3 --warmup 1
pmc c4_write WRITE_SIZE $B --workload c4 --steps 3 --warmup 1
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_c3_fetch.db $OUT/pmc_c3_write.db bigprod_kernel c3_n1 $OUT/hbm_traffic.json > /dev/null
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_c4s_fetch.db $OUT/pmc_c4s_write.db bigprod_f3 c4s_n1 $OUT/hbm_traffic.json > /dev/null
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_c4_fetch.db $OUT/pmc_c4_write.db bigprod_f3 c4_n1 $OUT/hbm_traffic.json > /dev/null
cp $OUT/pmc_c4_fetch.db $OUT/r02_c4_pmc_fetch_size.db; cp $OUT/pmc_c4_write.db $OUT/r02_c4_pmc_write_size.db
# ---- SQ counters, k = 64 fp32 streaming kernel: round-1 (21, bf16x3), round-2 bf16x3 (108) and fp16 two-term (108, MB_NSPLIT=4) ----
SW="$ROOT/tools/mb/mb_bp_sweep 64 8192 262144 0"
for v in 21 108; do
  export MB_NSPLIT=3
  pmc sq_a_v$v "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" $SW $v
  pmc sq_b_v$v "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" $SW $v
done
export MB_NSPLIT=4
pmc sq_a_v125_f16x2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" $SW 125
pmc sq_b_v125_f16x2 "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" $SW 125
unset MB_NSPLIT
for f in $OUT/pmc_sq_*.db; do echo "== $f"; python3 $ROOT/tools/pmc_dump.py $f bigprod; done > $OUT/r02_k64_f32_sq_counters.txt 2>&1
# ---- power / clock while the streaming kernels run ----
bash $ROOT/tools/power_clock.sh > /dev/null 2>&1
rm -f $OUT/long_*.log $OUT/pmc_sq_*.db $OUT/pmc_c4s_*.db $OUT/pmc_c4_*.db
ls -la $OUT
for f in $OUT/r02_bench_*.json; do echo $f; python3 -c "
import json,sys
j=json.loads(open('$f').read()); r=j['roofline']; print('  it/s %.2f ms/step %.4f bigprod %.4f ms %.0f GB/s frac %.3f windows %d'%(j['value'],j['ms_per_step'],r['avg_launch_ms'],r['achieved'],r['frac'],j['windows']))"; done
