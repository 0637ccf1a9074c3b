#!/usr/bin/env python3
"""HBM traffic of one kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
SEPARATE runs, as MI355X_MICROARCH.md prescribes).  Units: both counters are KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled.
Usage: pmc_traffic.py <fetch.db> <write.db> <kernel-substring> [key] [hbm_traffic.json]"""
import hashlib
import json
import os
import sqlite3
import sys


def avg_counter(db_path, counter, needle):
    db = sqlite3.connect(db_path)
    rows = db.execute("select value from counters_collection where counter_name=? and kernel_name like ?",
                      (counter, f"%{needle}%")).fetchall()
    vals = [r[0] for r in rows]
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_db, write_db, needle = sys.argv[1:4]
    f, nf = avg_counter(fetch_db, "FETCH_SIZE", needle)
    w, nw = avg_counter(write_db, "WRITE_SIZE", needle)
    total = 2.0 * f * 1024.0 + w * 1024.0
    out = {"bytes_per_launch": total, "fetch_size_kib_raw": f, "write_size_kib_raw": w, "launches_sampled": [nf, nw],
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); "
                     "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of a wide streaming read)"}
    # bench.py prints the figure only while the streaming kernels' source is the one it was measured on
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smallk_amd", "csrc", "bigprod.hip")
    out["kernel_source_sha16"] = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 5:
        key, path = sys.argv[4], sys.argv[5]
        try:
            j = json.load(open(path))
        except Exception:
            j = {}
        j[key] = out
        json.dump(j, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
