set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
timeout 600 python3 tools/quick_parity.py 2>&1 | tail -1
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
B="python3 bench.py --no-cpu-baseline"
$B --workload s_1m --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/gm_s_1m.json
$B --workload b32 --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/gm_b32.json
$B --workload c4s --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/gm_c4s.json
$B --workload s_reuters --steps 200 --warmup 20 2>/dev/null | tail -1 > $OUT/gm_sr.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_gm -o x -- python3 $R/bench.py --no-cpu-baseline --workload s_1m --steps 10 --warmup 3 > /dev/null 2>&1
DB=$(find $OUT/kt_gm -name '*.db' | head -1)
[ -n "$DB" ] && python3 $R/tools/prof_summary.py "$DB" $OUT/gm_s_1m_kernel_stats.md > /dev/null
rm -rf $OUT/kt_gm
for f in $OUT/gm_*.json; do echo $f; python3 -c "
import json
j=json.loads(open('$f').read()); print('  it/s %.2f ms/step %.4f'%(j['value'],j['ms_per_step']))"; done
grep gram_mfma $OUT/gm_s_1m_kernel_stats.md | cut -c1-30,90-170
