#!/bin/bash
# ceiling of a pure row gather for rows of 64 .. 512 bytes (the factor rows of the sparse products at KP = 8 .. 64) from tables that sit in
# one XCD's L2 (2 MB), in the Infinity Cache (64 / 256 MB) and in HBM (1024 MB): profiles/r06_row_gather_ceiling.{txt,json}
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
cd $ROOT/tools/mb && hipcc --offload-arch=gfx950 -O3 -std=c++17 mb_gather.hip -o mb_gather 2>/dev/null
: > $OUT/r06_row_gather_ceiling.txt
for rb in 64 128 256 512; do for mb in 2 64 256 1024; do ./mb_gather gatherrow$rb $mb 16 | tail -1 >> $OUT/r06_row_gather_ceiling.txt; done; done
./mb_gather stream 1024 | tail -1 >> $OUT/r06_row_gather_ceiling.txt
python3 - <<PY
import re, json
out = {}
for l in open("$OUT/r06_row_gather_ceiling.txt"):
    m = re.match(r"gatherrow(\d+): .* of (\d+) B from a (\d+) MB table .* = ([\d.]+) TB/s gathered", l)
    if m:
        out.setdefault(m.group(2), []).append([int(m.group(3)), float(m.group(4)) * 1000.0])
json.dump(out, open("$OUT/r06_row_gather_ceiling.json", "w"), indent=1)
print(out)
PY
cat $OUT/r06_row_gather_ceiling.txt
