set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
for w in s_reuters s_reuters_hals s_1m; do
  case $w in s_1m) steps="--steps 20 --warmup 3";; *) steps="--steps 200 --warmup 20";; esac
  python3 bench.py --workload $w $steps 2>/dev/null | tail -1 > $OUT/r06_bench_$w.json
done
python3 bench.py --workload c3 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/r06_bench_c3.json
bash tools/profile_all.sh tables r06 > /dev/null 2>&1
for f in $OUT/r06_bench_s_*.json $OUT/r06_bench_c3.json; do echo $f; python3 -c "
import json
j=json.loads(open('$f').read()); print('  it/s %.2f ms/step %.4f frac %.3f'%(j['value'],j['ms_per_step'], j['roofline']['frac']))"; done
bash tools/profile_all.sh suite r06 | tail -3
