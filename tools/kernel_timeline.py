#!/usr/bin/env python3
"""Launches of the kernels whose name contains <pattern>, in launch order, from a rocprofv3 --kernel-trace .db:
   kernel_timeline.py <results.db> <pattern> [out.txt]   -> one line per kernel name: durations in us, in order"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, duration from kernels where name like ? order by start", (f"%{sys.argv[2]}%",)).fetchall()
by = {}
for name, start, dur in rows:
    by.setdefault(name, []).append(dur / 1e3)
lines = []
for name, ds in by.items():
    lines.append(f"{name[:110]}\n   calls {len(ds)}  total {sum(ds) / 1e3:.3f} ms  avg {sum(ds) / len(ds):.1f} us  max {max(ds):.1f} us")
    lines.append("   us in launch order: " + " ".join(f"{d:.0f}" for d in ds))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
