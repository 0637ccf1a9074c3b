#!/bin/bash
# NNLS k = 64: solve bounds in steps of 4 (default) vs steps of 8 (SMK_NNLS_INV=4), per-kernel times from rocprofv3
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in 1 4 1 4; do
  export SMK_NNLS_INV=$mode
  rm -rf /tmp/kt
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o x -- python3 $R/bench.py --workload c4s --steps 10 --warmup 3 --no-cpu-baseline > /tmp/run.log 2>&1
  DB=$(find /tmp/kt -name '*.db' | head -1)
  echo "== SMK_NNLS_INV=$mode"; python3 $R/tools/prof_summary.py "$DB" /tmp/s.md > /dev/null; grep -E "nnls_bpp_inv|bigprod" /tmp/s.md | cut -c1-60,100-200
  tail -1 /tmp/run.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('  it/s %.2f'%j['value'])"
done
