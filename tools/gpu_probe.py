"""Developer probe: HIP path vs the CPU checkers on a handful of batches."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import synth, parity

ctx = _lib.Context(0)
for gd, ibsi, mode in [(8, False, "rand"), (64, False, "rand"), (-16, False, "rand"), (100, False, "rand"), (20, True, "ibsi"), (8, False, "tile")]:
    s = _abi.default_settings(gd, ibsi)
    if mode == "tile":
        b = synth.tile_batch(0)
    else:
        rois = synth.random_rois(80, seed=1, slide=True)
        if mode == "ibsi":
            rois = [dict(r, inten=(np.asarray(r["inten"]) % 7).astype(np.uint32)) for r in rois]
        b = _abi.batch_from_rois(rois)
    names = _lib.column_names(3, s)
    t0 = time.time(); G = ctx.featurize_host(b, 3, s); t1 = time.time()
    O = po.oracle_featurize(b, 3, s)
    bad = parity.compare_tables(G, O, names)
    ex = ((G == O) | (np.isnan(G) & np.isnan(O))).mean()
    print(f"gd={gd} ibsi={ibsi} {mode}: n_roi={b.n_roi} hip {1e3*(t1-t0):.1f} ms  bit-exact frac {ex:.4f}  mismatches {len(bad)}")
    for m in bad[:12]:
        print("   ", m)
