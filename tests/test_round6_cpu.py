"""CPU-side checks of round 6's additions to the measurement path (no GPU call): the arithmetic of the 8-GPU projection that
bench.py prints beside the C4 line, the gather ceiling lookup of the sparse lines, the new bench flags, the declarations of the new
C-ABI entries."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_projection_arithmetic_matches_design_section_7():
    import bench
    p = bench.projection_8gpu(262144, 65536, 64, "BPP", 22.8)
    pay = p["payload_bytes"]
    assert pay["reduce_scatter_of_AHt_in"] == 262144 * 64 * 8 == 128 * 2 ** 20           # 128 MiB of fp64 partial sums per rank
    assert pay["reduce_scatter_out_per_rank"] * 8 == pay["reduce_scatter_of_AHt_in"]
    assert pay["all_gather_of_packed_W"] == 262144 * 64 * 4
    assert p["assumed_bus_GBps"] == 300.0 and "NOT A MEASUREMENT" in p["for"]
    # one reduce-scatter chunk + one all-gather chunk + two latency-bound all-reduces stay exposed
    exp = (128 * 2 ** 20 / 4 + 262144 * 64 * 4 / 4) / 300e9 * 1e3 + 0.06
    assert abs(p["exposed_ms_with_the_chunk_pipeline"] - exp) < 1e-9
    if p["compute_ms"]:
        assert abs(p["projected_speedup_at_8"] - 22.8 / (p["compute_ms"] + exp)) < 1e-9
        assert p["projected_speedup_at_8"] < p["compute_only_speedup_at_8"] <= 8.5


def test_gather_ceiling_lookup(tmp_path, monkeypatch):
    import bench_sparse
    f = tmp_path / "ceil.json"
    f.write_text(json.dumps({"256": [[2, 9000.0], [256, 6000.0], [1024, 3000.0]]}))
    monkeypatch.setattr(bench_sparse, "GATHER_CEILING_FILE", str(f))
    assert bench_sparse.gather_peak(256, 1 << 20)[0] == 9000.0                # fits the L2-sized table
    assert bench_sparse.gather_peak(256, 256 << 20)[0] == 6000.0              # the 10^6 x 32 factor: Infinity Cache
    assert bench_sparse.gather_peak(256, 4 << 30)[0] == 3000.0                # beyond every table measured: the largest
    assert bench_sparse.gather_peak(64, 1 << 20) == (None, None)              # no measurement for this row size
    monkeypatch.setattr(bench_sparse, "GATHER_CEILING_FILE", str(tmp_path / "missing.json"))
    assert bench_sparse.gather_peak(256, 1 << 20) == (None, None)


def test_bench_flags_of_round_6():
    import bench
    a = bench.parse_args(["--check-every-iteration", "--api-path", "--workload", "c2"])
    assert a.check_every_iteration and a.api_path and a.workload == "c2"
    a = bench.parse_args([])
    assert not a.check_every_iteration and not a.api_path and a.workload == "c4" and a.gpus == 1
    for w in ("c4s2", "c3s2"):                       # two ranks of these = the per-rank shard of the 8-GPU run
        m, n, k, alg, storage, _ = bench.WORKLOADS[w]
        full = bench.WORKLOADS["c4" if w == "c4s2" else "c3"]
        assert (m, k, alg, storage) == (full[0], full[2], full[3], full[4]) and n * 4 == full[1]


def test_new_abi_entries_are_declared_with_their_reference_anchor():
    h = open(os.path.join(ROOT, "include", "smallk_amd.h")).read()
    for name, anchor in (("smk_solver_iterate_checked", "nmf_solve_generic.hpp:98-121"), ("smk_solver_kernel_name", "roofline"),
                         ("smk_debug_nnls_stats", "nnls.hpp:192-241")):
        i = h.index(f"int {name}(")
        assert anchor in h[max(0, i - 900):i], name
