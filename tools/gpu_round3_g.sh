#!/bin/bash
# C2: streaming-kernel variants and row-split counts (latency-bound at this size)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
run() { python3 bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('%-40s it/s %.1f  us/iter %.1f  bigprod %.1f us (W.A %.1f, H.At %.1f)'%('$1', j['value'], j['ms_per_step']*1e3, j['roofline']['avg_launch_ms']*1e3, j['roofline']['pass_WtA_ms']*1e3, j['roofline']['pass_HAt_ms']*1e3))"; }
run default
for v in 108 126 110 111 115; do SMK_BP_VARIANT=$v run "variant $v"; done
for s in 4 8 16 32; do SMK_BP_SPLITS=$s run "splits $s"; done
for s in 8 32; do SMK_BP_VARIANT=126 SMK_BP_SPLITS=$s run "variant 126 splits $s"; done
