#!/bin/bash
# first GPU pass of round 2: new tests, then C4-shard bench + kernel table
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_nnls.py tests/test_gpu_ref_sparse.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q --durations=8 2>&1 | tail -20
timeout 1200 python -m pytest tests/test_gpu_c5.py -x -q --durations=4 2>&1 | tail -20
} > gpurun_out/r2a_tests.log 2>&1
python bench.py --workload c4s --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2a_bench_c4s.log 2>&1
SMK_NNLS_INV=0 python bench.py --workload c4s --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2a_bench_c4s_old.log 2>&1
bash tools/prof_workload.sh c4s 6 > gpurun_out/r2a_prof_c4s.log 2>&1
tail -5 gpurun_out/r2a_tests.log
