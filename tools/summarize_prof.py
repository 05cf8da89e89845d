"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(root, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield f, r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f, r in rows("trace/**/*kernel_stats.csv"):
    print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})

print("== per-dispatch kernel trace: roi_features_kernel, metric workload only (largest grid) ==")
d = []
for f, r in rows("trace/**/*kernel_trace.csv"):
    if "roi_features" in r.get("Kernel_Name", ""):
        d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), int(r.get("Grid_Size_X") or 0), r.get("Workgroup_Size_X")))
metric_grid = max((x[4] for x in d), default=0)
d = [x for x in d if x[4] == metric_grid]
if d:
    ns = [x[0] for x in d]
    print(f"dispatches {len(d)} avg {sum(ns)/len(ns)/1e6:.3f} ms min {min(ns)/1e6:.3f} max {max(ns)/1e6:.3f}  vgpr {d[0][1]} sgpr {d[0][2]} lds {d[0][3]} grid_x {d[0][4]} wg_x {d[0][5]}")

# the launch group bench.py times with HIP events = roi_features_kernel (+ glcm_features_kernel when the GLCM features run
# as their own launch): average duration of each on the metric workload, and their sum
g = []
for f, r in rows("trace/**/*kernel_trace.csv"):
    if "glcm_features_kernel" in r.get("Kernel_Name", ""):
        g.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r.get("Grid_Size_X") or 0)))
if g:
    gmax = max(x[1] for x in g)
    gn = [x[0] for x in g if x[1] == gmax]
    k1 = sum(ns) / len(ns) / 1e6 if d else 0.0
    k2 = sum(gn) / len(gn) / 1e6
    print(f"== launch group: roi_features_kernel {k1:.3f} ms + glcm_features_kernel {k2:.3f} ms = {k1 + k2:.3f} ms ==")

print("== PMC (per dispatch of roi_features_kernel, averaged) ==")
acc = defaultdict(list)
for f, r in rows("pmc_*/**/*counter_collection.csv"):
    if "roi_features" in r.get("Kernel_Name", "") and int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0) >= metric_grid:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k}: n={len(v)} mean={sum(v)/len(v):.6g}")
if "FETCH_SIZE" in acc:
    f = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"])
    print(f"FETCH_SIZE KB -> bytes: {f*1024:.4g}; x2 gfx950 correction for wide coalesced reads: {2*f*1024:.4g}")
if "WRITE_SIZE" in acc:
    w = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
    print(f"WRITE_SIZE KB -> bytes: {w*1024:.4g}")
