#!/bin/bash
# every GPU test file with poisoned workspaces (SMK_POISON=1): reads of memory nobody wrote become NaN / -1 in every run
cd /root/repo
mkdir -p gpurun_out/poison
for f in tests/test_*.py; do
  b=$(basename $f .py)
  SMK_POISON=1 timeout 900 python -m pytest $f -q -m gpu -rf > gpurun_out/poison/$b.log 2>&1
  echo "$b: $(grep -E "passed|failed|deselected|Aborted|Fatal" gpurun_out/poison/$b.log | tail -2 | tr '\n' ' ')"
  grep -E "^FAILED" gpurun_out/poison/$b.log | cut -c1-200
done
