#!/bin/bash
# round 3, call w: what the accurate-form product (bigprod_f64_kernel) waits for -- counters on HALS at k = 192
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03w; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pmc() {  # name, counters
  local name=$1 ctr=$2
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_$name -o x -- python3 $ROOT/tools/wide_run.py 16384 8192 192 HALS 3 1 > $OUT/pmc_${name}.log 2>&1
  local DB=$(find $OUT/pmc_$name -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/pmc_dump.py $DB bigprod_f64 >> $OUT/counters.txt 2>&1
  rm -rf $OUT/pmc_$name
}
pmc a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
pmc b "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
pmc c "FETCH_SIZE"
pmc d "WRITE_SIZE"
pmc e "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_MISC"
pmc f "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
