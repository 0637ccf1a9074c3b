set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 2400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_dist.py tests/test_gpu_flatclust.py tests/test_reference_callers.py tests/test_gpu_sparse.py -x -q -k "stopping or progress or check_every or sweep or rule or callers or tolerance or converge" 2>&1 | tail -15 > $OUT/t_call7.txt
python3 tools/fuzz_parity.py 150 11 2>&1 | tail -3 >> $OUT/t_call7.txt
SMK_PROGRESS_DEPTH=1 python3 tools/fuzz_parity.py 80 12 2>&1 | tail -2 >> $OUT/t_call7.txt
B="python3 bench.py --no-cpu-baseline"
for d in 1 2 3; do
export SMK_PROGRESS_DEPTH=$d
$B --workload c2 --steps 200 --warmup 20 --check-every-iteration 2>/dev/null | tail -1 > $OUT/ck3_c2_d$d.json
$B --workload c3 --steps 20 --warmup 3 --check-every-iteration 2>/dev/null | tail -1 > $OUT/ck3_c3_d$d.json
done
unset SMK_PROGRESS_DEPTH
$B --workload c4 --check-every-iteration 2>/dev/null | tail -1 > $OUT/ck3_c4_d3.json
SMK_PROGRESS_FUSED=0 $B --workload c2 --steps 200 --warmup 20 --check-every-iteration 2>/dev/null | tail -1 > $OUT/ck3_c2_d3_f0.json
$B --workload c2 --steps 200 --warmup 20 2>/dev/null | tail -1 > $OUT/ck3_c2_unchecked.json
$B --workload c3 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/ck3_c3_unchecked.json
$B --workload s_reuters --steps 200 --warmup 20 --check-every-iteration 2>/dev/null | tail -1 > $OUT/ck3_s_reuters_checked.json
cat $OUT/t_call7.txt
for f in $OUT/ck3_*.json; do echo $f; python3 -c "
import json
j=json.loads(open('$f').read()); print('  it/s %.2f ms/step %.4f'%(j['value'],j['ms_per_step']))"; done
