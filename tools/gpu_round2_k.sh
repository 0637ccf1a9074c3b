#!/bin/bash
# fp16 two-term products inside the solver: C4 shard / C2 / 32768x8192 k=32 with SMK_NSPLIT=4 and 3, microbench, quick parity
cd /root/repo
summ() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('  it/s %.2f ms/step %.4f bigprod %.4f ms (WtA %.4f HAt %.4f) %.0f GB/s frac %.3f'%(j['value'],j['ms_per_step'],r['avg_launch_ms'],r['pass_WtA_ms'],r['pass_HAt_ms'],r['achieved'],r['frac']))"; }
for ns in 4 3; do
for w in c4s c2 b32; do
  echo "$w nsplit $ns"; SMK_NSPLIT=$ns python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | summ
done
done
echo c3; python3 bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | summ
MB_NSPLIT=4 tools/mb/mb_bp_sweep 64 262144 8192 0 108 111
MB_NSPLIT=4 tools/mb/mb_bp_sweep 64 8192 262144 0 108 111
SMK_NSPLIT=4 python tools/quick_parity.py
