#!/bin/bash
# C4 whole: forced row-split counts (SMK_BP_SPLITS) -- does the W'A pass gain from splits? (it does not)
cd /root/repo
summ() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('  it/s %.2f ms/step %.4f bigprod %.4f ms (WtA %.4f HAt %.4f) %.0f GB/s frac %.3f'%(j['value'],j['ms_per_step'],r['avg_launch_ms'],r['pass_WtA_ms'],r['pass_HAt_ms'],r['achieved'],r['frac']))"; }
for sp in 0 2 4 8 16; do
  echo "c4 SMK_BP_SPLITS=$sp"; SMK_BP_SPLITS=$sp python3 bench.py --workload c4 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | summ
done
