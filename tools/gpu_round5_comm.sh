#!/bin/bash
# First-real-run diagnostics for N > 1 without a node: measured exposure of the exchange.
#  - two shards on ONE GPU through the in-process stand-in (the collectives are real copies on the second stream, competing with
#    the products for the same device): exposure must be non-zero and consistent with the step time;
#  - rank 0 of an emulated world of 8 (RCCL calls with one rank: device-local copies): exposure ~ the brackets themselves;
#  - the same for C3 / HALS and C4-sized MU.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd $ROOT
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pr=d.get("per_rank") or []
print(json.dumps({"workload": d["config"]["workload"], "parallelism": d["config"]["parallelism"], "ms_per_step": d["ms_per_step"], "per_rank": pr}, indent=1))'
SMK_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --single-process --workload c4s --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r05_bench_c4s_2shards_standin.json
SMK_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --single-process --workload c3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r05_bench_c3_2shards_standin.json
python3 bench.py --emulate-world 8 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r05_bench_c4_emulate8.json
python3 bench.py --emulate-world 8 --workload c3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r05_bench_c3_hals_emulate8.json
SMK_BENCH_ALG=MU python3 bench.py --emulate-world 8 --workload c4mu --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r05_bench_c4_mu_emulate8.json
for f in c4s_2shards_standin c3_2shards_standin c4_emulate8 c3_hals_emulate8 c4_mu_emulate8; do echo "== $f"; python3 -c "$pick" < $OUT/r05_bench_$f.json; done > $OUT/r05_exposed_comm.txt 2>&1
