"""Large-ROI (global workspace) probe: each family alone against the oracle, one subprocess per case."""
import subprocess
import sys

if len(sys.argv) == 1:
    for which in ("all", "rect", "ellipse"):
        for gd in (8, -20):
            for fam in (4, 8, 16, 32, 64, 127):
                r = subprocess.run([sys.executable, __file__, which, str(gd), str(fam)], capture_output=True, text=True)
                tail = (r.stdout + r.stderr).strip().splitlines()
                msg = [l for l in tail if "mismatch" in l or "fault" in l or "Error" in l]
                print(which, gd, fam, "rc", r.returncode, msg[:2], flush=True)
    sys.exit(0)

import numpy as np
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import parity
from tests.test_parity_gpu import _large_rois

which, gd, fam = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rois = _large_rois()
if which == "rect":
    rois = [rois[3]]
elif which == "ellipse":
    rois = [rois[-1]]
ctx = _lib.Context(0)
b = _abi.batch_from_rois(rois)
s = _abi.default_settings(gd)
G = ctx.featurize_host(b, fam, s)
O = po.oracle_featurize(b, fam, s)
bad = parity.compare_tables(G, O, _lib.column_names(fam, s))
print("mismatches:", len(bad), bad[:3], flush=True)
