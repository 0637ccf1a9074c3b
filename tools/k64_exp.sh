#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in c2 b32 c4s; do
timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['value'], d['ms_per_step'], r['achieved'])"
done
timeout 300 python tools/quick_parity.py 2>&1 | tail -1
