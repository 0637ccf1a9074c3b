#!/usr/bin/env python3
"""MFMA utilisation of one kernel from a rocprofv3 PMC pass (north_star: "rocprof HBM GB/s and MFMA utilisation"):
   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py ...
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) -- rocprofiler-sdk's own MfmaUtil
(/opt/rocm/share/rocprofiler-sdk/counter_defs.yaml: reduce(SQ_VALU_MFMA_BUSY_CYCLES,sum)/(reduce(GRBM_GUI_ACTIVE,max)*SIMD_NUM)),
per launch, averaged over the launches of the kernel.  Stored in profiles/mfma_util.json keyed by workload, with the hash of
bigprod.hip it was measured on (bench.py prints roofline.mfma_busy_frac only while that file is unchanged).
Usage: pmc_mfma.py <results.db> <kernel-substring> [key] [mfma_util.json]"""
import hashlib
import json
import os
import sqlite3
import sys

SIMDS = 1024


def per_launch(db_path, needle):
    db = sqlite3.connect(db_path)
    rows = db.execute("select dispatch_id, counter_name, value, duration from counters_collection where kernel_name like ?", (f"%{needle}%",)).fetchall()
    by = {}
    for did, name, val, dur in rows:
        by.setdefault(did, {}).setdefault(name, 0.0)
        by[did][name] += val
        by[did]["_duration_ns"] = float(dur or 0)
    out = []
    for l in by.values():
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (one GRBM per XCD): the busy time of the chip is the per-XCD value --
        # rocprofiler-sdk's MfmaUtil takes reduce(GRBM_GUI_ACTIVE, max).  Recognised by the clock it implies (active cycles /
        # kernel duration): ~2 GHz per XCD, ~17 GHz for the sum.
        g, d = l.get("GRBM_GUI_ACTIVE", 0.0), l.get("_duration_ns", 0.0)
        if g > 0 and d > 0 and g / d > 4.0:
            l["GRBM_GUI_ACTIVE_sum_over_xcds"] = g
            l["GRBM_GUI_ACTIVE"] = g / 8.0
        if g > 0 and d > 0:
            l["effective_clock_ghz"] = l["GRBM_GUI_ACTIVE"] / d
        out.append(l)
    return out


def main():
    db, needle = sys.argv[1:3]
    launches = [l for l in per_launch(db, needle) if l.get("GRBM_GUI_ACTIVE", 0) > 0]
    if not launches:
        raise SystemExit(f"no launches of '{needle}' with counters in {db}")
    n = len(launches)
    avg = lambda c: sum(l.get(c, 0.0) for l in launches) / n
    out = {
        "mfma_busy_frac": sum(l.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (l["GRBM_GUI_ACTIVE"] * SIMDS) for l in launches) / n,
        "sq_busy_frac_of_active": avg("SQ_BUSY_CYCLES") / avg("GRBM_GUI_ACTIVE") if avg("SQ_BUSY_CYCLES") else None,
        "launches_sampled": n,
        "counters_avg_per_launch": {c: avg(c) for c in sorted({k for l in launches for k in l})},
        "source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE "
                  "(its own pass); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs) = rocprofiler-sdk's MfmaUtil / 100",
    }
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smallk_amd", "csrc", "bigprod.hip")
    out["kernel_source_sha16"] = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    out["kernel"] = needle
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 4:
        key, path = sys.argv[3], sys.argv[4]
        try:
            j = json.load(open(path))
        except Exception:
            j = {}
        j[key] = out
        json.dump(j, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
