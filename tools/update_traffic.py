"""profiles/hbm_traffic.json from a tools/profile_bench.sh run: python tools/update_traffic.py <tag>  (reads gpurun_out/prof_<tag>/summary.txt).
bench.py reports the value as `roofline.traffic` only while the sources of the metric kernels (roi_features.hip, glcm_rows.h, device_math.h, roi_kernel.h) still hash to what was measured."""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
txt = open(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "summary.txt")).read()
fetch = float(re.search(r"FETCH_SIZE: n=\d+ mean=([0-9.e+]+)", txt).group(1))
write = float(re.search(r"WRITE_SIZE: n=\d+ mean=([0-9.e+]+)", txt).group(1))
kern = re.search(r"launch group: roi_features_kernel ([0-9.]+) ms \+ glcm_features_kernel ([0-9.]+) ms", txt)
rec = {"round": tag, "kernel": "roi_features_kernel_occ8<1, 0>", "tiles": 1000, "gray_depth": 8,
       "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
       "hbm_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
       "rocprof_kernel_ms": [float(kern.group(1)), float(kern.group(2))] if kern else None,
       "kernel_source_sha256": hashlib.sha256(b"".join(open(os.path.join(ROOT, "nyxus_amd", "csrc", f_), "rb").read()
                                                       for f_ in ("roi_features.hip", "glcm_rows.h", "device_math.h", "roi_kernel.h"))).hexdigest(),
       "collected": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_bench.sh {tag}), bench.py --steps 5 --warmup 2; "
                    f"metric-workload dispatches only (profiles/{tag}_summary.txt)"}
# the tile path's traffic per tile from the same round's PMC pass over tools/tile_ab.py (tools/final_prof_r06.sh): all kernels of the call
tp = os.path.join(ROOT, "gpurun_out", f"{tag}_tilepath_traffic.txt")
if os.path.exists(tp):
    tot = 0.0
    for line in open(tp):
        m = re.search(r"dispatches (\d+) FETCH_GB\(x2 gfx950\) ([0-9.]+) WRITE_GB ([0-9.]+)", line)
        if m:
            tot += float(m.group(2)) + float(m.group(3))
    rec["tile_path_bytes_per_tile"] = tot * 1e9 / 1000
    rec["tile_path_source"] = (f"profiles/{tag}_tilepath_traffic.txt (tools/pmc.sh traffic all -- tools/tile_ab.py 1000 8): sum over the path's kernels of "
                               "FETCH_SIZE x 2 + WRITE_SIZE per call / 1000 tiles")
json.dump(rec, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(rec["hbm_bytes_per_launch"], rec["kernel_source_sha256"][:12])
