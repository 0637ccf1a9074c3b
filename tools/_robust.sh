set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT gpurun_out/poison
: > $OUT/r06_poisoned_suite.txt
for f in tests/test_gpu_*.py tests/test_reference_callers.py; do
  b=$(basename $f .py)
  SMK_POISON=1 timeout 1200 python3 -m pytest $f -q -m gpu -rf > gpurun_out/poison/$b.log 2>&1
  echo "$b: $(grep -E "passed|failed|deselected|Aborted|Fatal" gpurun_out/poison/$b.log | tail -2 | tr '\n' ' ')" >> $OUT/r06_poisoned_suite.txt
  grep -E "^FAILED" gpurun_out/poison/$b.log | cut -c1-200 >> $OUT/r06_poisoned_suite.txt
done
timeout 900 python3 tools/fuzz_adversarial.py 250 61 2>&1 | tail -25 > $OUT/r06_fuzz_adversarial_seed61.log
timeout 600 python3 tools/fuzz_small_k_bpp.py 150 62 2>&1 | tail -12 > $OUT/r06_fuzz_small_k_bpp_seed62.log
timeout 600 python3 tools/fuzz_parity.py 600 63 2>&1 | tail -4 > $OUT/r06_fuzz_parity_600_cases.log
timeout 600 python3 tools/fuzz_hier.py 60 64 2>&1 | tail -5 > $OUT/r06_fuzz_hier_60_cases.log
cat $OUT/r06_poisoned_suite.txt
tail -6 $OUT/r06_fuzz_adversarial_seed61.log | cut -c1-300; tail -4 $OUT/r06_fuzz_small_k_bpp_seed62.log | cut -c1-300; tail -3 $OUT/r06_fuzz_parity_600_cases.log | cut -c1-300; tail -3 $OUT/r06_fuzz_hier_60_cases.log | cut -c1-300
