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
f)  # guard test again; the rank-tier anomalies: kernel tables of HALS at k = 80 / 100 / 128, first iterations against steady state
    for pert in 0.05 0.1 0.2; do for ev in 2 0; do SMK_GUARD_EVERY=$ev timeout 300 python3 tools/guard_case.py ill BPP 8 40 $pert 2>/dev/null | tail -1; done; done > $OUT/guard_ill_by_pert.txt
    timeout 600 python -m pytest tests/test_gpu_variants.py -m gpu -q --tb=short -k guard 2>&1 | tail -8 > $OUT/tests.txt
    cd /tmp
    for k in 80 100 128; do
        timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_hals$k -o x -- python3 $ROOT/tools/wide_run.py 16384 8192 $k HALS 12 1 > $OUT/hals${k}_run.log 2>&1
        DB=$(find $OUT/kt_hals$k -name '*.db' | head -1)
        [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r04_hals_k${k}_kernel_stats.md > /dev/null
        rm -rf $OUT/kt_hals$k
    done
    cd $ROOT
    for k in 192 512; do python3 tools/iter_times.py 16384 8192 $k BPP 16; done > $OUT/r04_bpp_first_iterations.txt 2>&1
    for k in 80 100 128; do python3 tools/iter_times.py 16384 8192 $k HALS 12; done >> $OUT/r04_bpp_first_iterations.txt 2>&1
    ;;
g)  # streaming rate of the accurate rank-2 product
    python3 tools/r2_dense_rate.py 65536 16384 bf16 > $OUT/r2_dense_rate.txt 2>&1
    python3 tools/r2_dense_rate.py 65536 16384 f32 >> $OUT/r2_dense_rate.txt 2>&1
    python3 tools/r2_dense_rate.py 8192 4096 f32 >> $OUT/r2_dense_rate.txt 2>&1
    SMK_NSPLIT=3 python3 tools/r2_dense_rate.py 65536 16384 bf16 >> $OUT/r2_dense_rate.txt 2>&1
    SMK_NSPLIT=3 python3 tools/r2_dense_rate.py 65536 16384 f32 >> $OUT/r2_dense_rate.txt 2>&1
    ;;
h)  # compact P columns for k <= 16: parity suite + C2 bench and kernel table
    timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nnls.py tests/test_gpu_variants.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py tests/test_gpu_hierclust.py -m gpu -q --tb=short -x 2>&1 | tail -15 > $OUT/tests.txt
    python3 bench.py --no-cpu-baseline --workload c2 --steps 200 --warmup 20 > $OUT/bench_c2.json 2> $OUT/bench.err
    python3 bench.py --no-cpu-baseline --workload c1 --steps 200 --warmup 20 > $OUT/bench_c1.json 2>> $OUT/bench.err
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_c2 -o x -- python3 $ROOT/bench.py --no-cpu-baseline --workload c2 --steps 50 --warmup 5 > $OUT/c2_run.log 2>&1
    DB=$(find $OUT/kt_c2 -name '*.db' | head -1); [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r04_c2_bpp_f32_kernel_stats.md > /dev/null; rm -rf $OUT/kt_c2
    ;;
i)  # evidence on the current build: whole GPU suite, the poisoned suite, randomised sweeps (NMF parity, HierNMF2, wide BPP)
    python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|^FAILED|^ERROR" | tail -20 > $OUT/gpu_suite.txt
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke >> $OUT/gpu_suite.txt
    bash tools/gpu_poison.sh > $OUT/poison.txt 2>&1
    timeout 1500 python3 tools/fuzz_hier.py 120 7 > $OUT/fuzz_hier_120.log 2>&1
    timeout 1500 python3 tools/fuzz_parity.py 600 11 1500 > $OUT/fuzz_parity_600.log 2>&1
    timeout 900 python3 tools/fuzz_wide_bpp.py 60 3 > $OUT/fuzz_wide_bpp_60.log 2>&1
    ;;
j)  # FETCH_SIZE calibration for 16-byte gathers, then the counter on the resident RANK2 kernel; HALS exchange A/B on C3
    cd /tmp
    for mode in stream gather16 gather16L2; do
        timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_$mode -o x -- $ROOT/tools/mb/mb_gather $mode 1024 16 > $OUT/mb_$mode.log 2>&1
        DB=$(find $OUT/pmc_$mode -name '*.db' | head -1); [ -n "$DB" ] && python3 $ROOT/tools/pmc_dump.py "$DB" _kernel > $OUT/pmc_$mode.txt; rm -rf $OUT/pmc_$mode
    done
    for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_r2p_$ctr -o x -- python3 $ROOT/tools/r2_iter.py 1000000 16 20 > $OUT/r2p_$ctr.log 2>&1
        DB=$(find $OUT/pmc_r2p_$ctr -name '*.db' | head -1); [ -n "$DB" ] && python3 $ROOT/tools/pmc_dump.py "$DB" rank2_persist > $OUT/pmc_r2p_$ctr.txt; rm -rf $OUT/pmc_r2p_$ctr
    done
    cd $ROOT
    { for f in stream gather16 gather16L2; do echo "== $f"; grep -v amdgpu $OUT/mb_$f.log | tail -2; cat $OUT/pmc_$f.txt; done; echo "== resident RANK2 kernel, 1 M x 1 M, 16 M entries, 20 iterations per launch"; cat $OUT/pmc_r2p_FETCH_SIZE.txt $OUT/pmc_r2p_WRITE_SIZE.txt; grep -v amdgpu $OUT/r2p_FETCH_SIZE.log | tail -2; } > $OUT/r04_fetch_size_calibration_16B_gathers.txt 2>&1
    python3 bench.py --no-cpu-baseline --workload c3 > $OUT/bench_c3_two_level.json 2> $OUT/bench.err
    SMK_HALS_EXCHANGE=1 python3 bench.py --no-cpu-baseline --workload c3 > $OUT/bench_c3_flat.json 2>> $OUT/bench.err
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_c3 -o x -- python3 $ROOT/bench.py --no-cpu-baseline --workload c3 --steps 20 --warmup 3 > $OUT/c3_run.log 2>&1
    DB=$(find $OUT/kt_c3 -name '*.db' | head -1); [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r04_c3_hals_bf16_kernel_stats.md > /dev/null; rm -rf $OUT/kt_c3
    cd $ROOT
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_fullsize.py tests/test_hals_blocked_reference.py -m gpu -q -k "hals or HALS or c3" 2>&1 | grep -E "passed|failed|^FAILED" > $OUT/tests_hals.txt
    ;;
k)  # FETCH_SIZE / WRITE_SIZE of the resident RANK2 kernel: a root-sized matrix (16 M entries) and a 2 M-entry one, 20 iterations each
    cd /tmp
    for shape in "1000000 16" "200000 10"; do
        tag=$(echo $shape | tr ' ' '_')
        for ctr in FETCH_SIZE WRITE_SIZE; do
            timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_$tag$ctr -o x -- python3 $ROOT/tools/r2_iter.py $shape 20 > $OUT/r2p_${tag}_$ctr.log 2>&1
            DB=$(find $OUT/pmc_$tag$ctr -name '*.db' | head -1); [ -n "$DB" ] && python3 $ROOT/tools/pmc_dump.py "$DB" rank2_persist > $OUT/pmc_r2p_${tag}_$ctr.txt; rm -rf $OUT/pmc_$tag$ctr
        done
        { echo "== resident RANK2 kernel, $shape (nodes, degree), 20 iterations per launch"; cat $OUT/pmc_r2p_${tag}_FETCH_SIZE.txt $OUT/pmc_r2p_${tag}_WRITE_SIZE.txt; grep "iterations" $OUT/r2p_${tag}_FETCH_SIZE.log | tail -2; } >> $OUT/r2p_counters.txt 2>&1
    done
    ;;
l)  # HierNMF2 on 2 / 4 / 8 device contexts (all on this one GPU): identical trees (tests), accepted speculative steps and the projected critical path on the C5-shaped run
    true
    for d in 1 2 4 8; do
        echo "== SMK_CLUST_DEVICES=$d (contexts on one GPU)" >> $OUT/c5_devices.txt
        SMK_CLUST_DEVICES=$d SMK_SHARDS_ON_ONE_GPU=1 SMK_CLUST_SERIALIZE=1 SMK_CLUST_TIMING=1 timeout 900 python3 tools/c5_hier.py 2>&1 | grep "subset\|hier_nmf2:\|purity" >> $OUT/c5_devices.txt
    done
    ;;
m)  # last build of the round: whole GPU suite + smoke, then the judged artefacts (quick set) on it
    python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|^FAILED|^ERROR" | tail -20 > $OUT/gpu_suite.txt
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke >> $OUT/gpu_suite.txt
    bash tools/profile.sh r04 quick > $OUT/profile.log 2>&1
    ;;
n)  # drift of the product forms over long runs (up to 500 iterations) at k <= 64
    timeout 3000 python3 tools/long_runs_500.py > $OUT/r04_long_runs_500_iterations.txt 2>&1
    ;;
esac
