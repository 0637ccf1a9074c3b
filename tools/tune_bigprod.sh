#!/bin/bash
# sweep streaming-kernel variants on the C3 bench; prints avg launch ms / HBM fraction per variant
for v in ${VARIANTS:-0 1 2 3 4 5 6}; do
  SMK_BP_VARIANT=$v python tools/quick_parity.py 2>&1 | tail -1
  for rep in 1 2; do
  SMK_BP_VARIANT=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('  variant $v splits ${SMK_BP_SPLITS:-auto}: it/s %.1f ms/step %.3f bigprod avg %.4f ms  %.0f GB/s frac %.3f' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac']))
"
  done
done
