#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python bench.py --workload c4s --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'])"
timeout 300 python tools/quick_parity.py 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -3
