#!/usr/bin/env python3
"""Kernel timeline of the LAST call of a rocprofv3 kernel trace (tools/ktrace.sh <tag> ...):  python tools/timeline.py <tag> [n_kernels]
start (us, relative), duration (us), stream/queue, grid, kernel -- for reading overlap between the main stream and the lanes."""
import csv, glob, sys
tag = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = [r for f in glob.glob(f"gpurun_out/ktrace_{tag}/trace/*/*kernel_trace.csv") for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("nyxhip::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    print("%9.1f %8.1f q%-3s g%-7s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?")), name[:60]))
