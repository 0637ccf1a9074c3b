#!/usr/bin/env python3
"""print per-kernel averages of every PMC counter in a rocprofv3 rocpd db: pmc_dump.py <db> [kernel-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
needle = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                  "where kernel_name like ? group by kernel_name, counter_name order by kernel_name, counter_name",
                  (f"%{needle}%",)).fetchall()
cur = None
for k, c, n, v in rows:
    if k != cur:
        print("\n" + k[:110]); cur = k
    print(f"   {c:32s} n={n:3d} avg={v:,.1f}")
