set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 2400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_dist.py tests/test_gpu_flatclust.py tests/test_reference_callers.py -x -q -k "stopping or progress or check_every or sweep or rule or callers or tolerance or converge or shard_geometry or upload" 2>&1 | tail -15 > $OUT/t_call8.txt
python3 tools/fuzz_parity.py 150 21 2>&1 | tail -2 >> $OUT/t_call8.txt
SMK_PROGRESS_DEPTH=3 python3 tools/fuzz_parity.py 100 22 2>&1 | tail -2 >> $OUT/t_call8.txt
SMK_BPP_GRADW=1 python3 tools/fuzz_parity.py 60 23 2>&1 | tail -2 >> $OUT/t_call8.txt
B="python3 bench.py --no-cpu-baseline"
$B --workload c2 --steps 200 --warmup 20 --check-every-iteration 2>/dev/null | tail -1 > $OUT/ck4_c2.json
SMK_BPP_GRADW=1 $B --workload c2 --steps 200 --warmup 20 --check-every-iteration 2>/dev/null | tail -1 > $OUT/ck4_c2_gradw.json
$B --workload c4 --check-every-iteration 2>/dev/null | tail -1 > $OUT/ck4_c4.json
$B --workload s_reuters --steps 200 --warmup 20 --check-every-iteration 2>/dev/null | tail -1 > $OUT/ck4_s_reuters.json
$B --workload c2 --steps 200 --warmup 20 2>/dev/null | tail -1 > $OUT/ck4_c2_unchecked.json
timeout 300 python3 tools/upload_rate.py 65536 16384 bf16 3 2>&1 | tail -1 >> $OUT/t_call8.txt
cat $OUT/t_call8.txt
for f in $OUT/ck4_*.json; do echo $f; python3 -c "
import json
j=json.loads(open('$f').read()); print('  it/s %.2f ms/step %.4f'%(j['value'],j['ms_per_step']))"; done
