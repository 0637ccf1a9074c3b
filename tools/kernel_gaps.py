#!/usr/bin/env python3
"""inter-kernel gaps of a rocprofv3 --kernel-trace run: kernel_gaps.py <results.db> [skip] [count]
prints, for `count` consecutive dispatches after the first `skip`, duration and the idle gap before each; then the average gap
in front of every kernel name"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 24
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end from kernels order by start").fetchall()
prev_end = None
gaps = collections.defaultdict(list)
for i, (name, st, en) in enumerate(rows):
    if prev_end is not None and i >= skip:
        gaps[name[:60]].append((st - prev_end) / 1e3)
    if skip <= i < skip + count and prev_end is not None:
        print(f"{(st - prev_end)/1e3:8.2f} us gap | {(en - st)/1e3:8.2f} us  {name[:90]}")
    prev_end = en
print()
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) > 10:
        print(f"{sum(v)/len(v):7.2f} us average gap before ({len(v):6d} x)  {k}")
