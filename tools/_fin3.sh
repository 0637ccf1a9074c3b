set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06
cd $R
mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_gpu_round6.py tests/test_gpu_dist.py -q -k "round6 or checked or kernel_names or counters or shard_geometry" 2>&1 | tail -8
for w in s_reuters s_reuters_hals s_1m; do
  case $w in s_1m) steps="--steps 20 --warmup 3";; *) steps="--steps 200 --warmup 20";; esac
  python3 bench.py --workload $w $steps 2>/dev/null | tail -1 > $OUT/r06_bench_$w.json
done
python3 bench.py --no-cpu-baseline --workload s_reuters --steps 200 --warmup 20 --check-every-iteration 2>/dev/null | tail -1 > $OUT/r06_bench_s_reuters_checked.json
python3 bench.py --no-cpu-baseline --workload b32 --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/r06_bench_b32.json
for f in $OUT/r06_bench_s_*.json $OUT/r06_bench_b32.json; do echo $f; python3 -c "
import json
j=json.loads(open('$f').read()); r=j['roofline']; print('  it/s %.2f ms/step %.4f frac %.3f nnls share %s nnls ms %s'%(j['value'],j['ms_per_step'], r['frac'], r.get('nnls_share_of_step'), r.get('nnls_avg_launch_ms')))"; done
head -8 $OUT/r06_s_1m_kernel_stats.md | cut -c1-170
