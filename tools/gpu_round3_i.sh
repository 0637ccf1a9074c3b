#!/bin/bash
# rank 0 of 8 (emulated): row-split counts of the chunked passes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
run() { python3 bench.py --emulate-world 8 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; p=j['per_rank'][0]; print('%-24s ms/iter %.3f  W.A %.3f  H.At %.3f ms  outside %.3f'%('$1', j['ms_per_step'], r['pass_WtA_ms'], r['pass_HAt_ms'], p['outside_products_ms_per_step']))"; }
run default
for s in 2 4 8 16; do SMK_BP_SPLITS=$s run "splits $s"; done
SMK_COMM_CHUNKS=2 run "2 chunks"
SMK_COMM_CHUNKS=3 run "3 chunks"
python -m pytest tests/test_gpu_dist.py -x -q -k "bench_two_processes" 2>&1 | tail -2
