#!/bin/bash
# C4 shard and C4 whole: fold depth of the fp16 two-term streaming kernel
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
run() { python3 bench.py --workload $2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('%-28s %s it/s %.2f  ms/iter %.3f  bigprod %.3f ms  frac %.3f'%('$1', '$2', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac']))"; }
for rep in 1 2; do
  run default c4s
  SMK_BP_VARIANT=127 run "variant 127 (fold 16)" c4s
  SMK_BP_VARIANT=108 run "variant 108 (fold 4)" c4s
done
run default c4
SMK_BP_VARIANT=127 run "variant 127 (fold 16)" c4
