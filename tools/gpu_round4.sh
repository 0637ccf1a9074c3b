#!/bin/bash
# round 4: the one-call GPU scripts, by step (gpurun -- 'bash tools/gpu_round4.sh <step>'); outputs under gpurun_out/r04<step>/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
STEP=${1:-a}
OUT=$ROOT/gpurun_out/r04$STEP; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
case $STEP in
a)  # whole GPU suite (every failure, not only the first), smoke, achieved errors of the dense clustering rows, default bench
    python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/gpu_tests.txt
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke >> $OUT/gpu_tests.txt
    python3 tools/dense_clust_errors.py > $OUT/dense_clust_errors.txt 2>&1
    python3 bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err
    ;;
b)  # the resident RANK2 kernel against the launch-per-kernel loop (results, iteration counts, time per iteration), the sparse
    # RANK2 / HierNMF2 tests, then the C5-shaped run
    SMK_R2P_PROFILE=1 timeout 600 python3 tools/r2_persist_check.py > $OUT/r2_persist_check.txt 2>&1
    timeout 900 python -m pytest tests/test_sparse.py tests/test_gpu_hierclust.py tests/test_gpu_c5.py tests/test_gpu_flatclust.py -m gpu -q --tb=short 2>&1 | tail -30 > $OUT/tests.txt
    timeout 300 python -m pytest tests/test_gpu_flatclust.py -m gpu -q -x --tb=long -k facade 2>&1 | grep -v "^$" | tail -60 > $OUT/facade_test.txt
    SMK_CLUST_TIMING=1 SMK_R2P_PROFILE=1 timeout 600 python3 tools/c5_hier.py > $OUT/c5.txt 2>&1
    ;;
c)  # C5-shaped run with the resident kernel for every node size, and with the default size limit
    for mode in 2 1 0; do
        SMK_R2_PERSIST=$mode SMK_CLUST_TIMING=1 timeout 600 python3 tools/c5_hier.py 2>&1 | grep "smk_clust\|hier_nmf2:" > $OUT/c5_persist$mode.txt
    done
    ;;
d)  # run-time guard cases, the dist tests (row-sharded W under the accurate form, product-form agreement), C4 bench with the guard
    for alg in BPP MU HALS; do for kind in ill well; do SMK_GUARD_EVERY=4 SMK_GUARD_VERBOSE=1 timeout 300 python3 tools/guard_case.py $kind $alg 8 40; done; done > $OUT/guard_cases.txt 2>&1
    timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_variants.py -m gpu -q --tb=short 2>&1 | tail -30 > $OUT/tests.txt
    python3 bench.py --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
    python3 bench.py --no-cpu-baseline --workload c2 > $OUT/bench_c2.json 2>> $OUT/bench_c4.err
    python3 bench.py --no-cpu-baseline --workload c3 > $OUT/bench_c3.json 2>> $OUT/bench_c4.err
    ;;
e)  # guard test after the opt-in change, clustering tests with examples/pyclust.py, then the round's judged artefacts (quick set)
    timeout 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_flatclust.py tests/test_examples.py -m gpu -q --tb=short 2>&1 | tail -15 > $OUT/tests.txt
    bash tools/profile.sh r04 quick > $OUT/profile.log 2>&1
    ;;
esac
