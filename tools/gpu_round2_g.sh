#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_nnls.py -x -q 2>&1 | tail -3 > gpurun_out/r2g.log
for mode in 1 2 3; do
  SMK_NNLS_INV=$mode python bench.py --workload c4s --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('nnls mode $mode: it/s %.2f ms/step %.3f bigprod avg %.3f ms'%(j['value'],j['ms_per_step'],r['avg_launch_ms']))
" >> gpurun_out/r2g.log
done
bash tools/prof_workload.sh c4s 6 > /dev/null 2>&1
head -10 gpurun_out/prof_c4s/kernel_stats.md >> gpurun_out/r2g.log
cat gpurun_out/r2g.log
