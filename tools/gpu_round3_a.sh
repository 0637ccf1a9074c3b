#!/bin/bash
# round 3, run a: the chunk pipeline -- dist tests, the default bench, and the per-rank time of an N-rank run
# measured on one GPU (bench.py --emulate-world N: rank 0's shard, blocks and chunk geometry, RCCL with one rank)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03a
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 > $OUT/dist_tests.log
#python bench.py 2> $OUT/bench_c4.err | tail -1 > $OUT/bench_c4.json
for n in 2 4 8; do
  python bench.py --emulate-world $n --no-cpu-baseline 2> $OUT/emu$n.err | tail -1 > $OUT/bench_c4_emulate$n.json
done
SMK_COMM_CHUNKS=1 python bench.py --emulate-world 8 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_c4_emulate8_c1.json
SMK_COMM_CHUNKS=8 python bench.py --emulate-world 8 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_c4_emulate8_c8.json
python bench.py --workload c4s --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_c4s.json
python bench.py --workload c3 --emulate-world 8 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_c3_emulate8.json
for f in $OUT/*.json; do echo $f; python3 -c "
import json
j=json.loads(open('$f').read()); r=j['roofline']; pr=(j.get('per_rank') or [None])[0]
print('  it/s %.2f ms/step %.4f bigprod %.4f ms %.0f GB/s frac %.3f windows %d'%(j['value'],j['ms_per_step'],r['avg_launch_ms'],r['achieved'],r['frac'],j['windows']))
if pr: print('  ', {k:(round(v,4) if isinstance(v,float) else v) for k,v in pr.items()})
if 'cpu_baseline' in j: print('  cpu', j['cpu_baseline']['value'], j['cpu_baseline']['sample_ms'])
"; done
cat $OUT/dist_tests.log
