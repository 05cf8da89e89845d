import sys, numpy as np
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import synth
ctx = _lib.Context(0)
rois = synth.random_rois(100, seed=21)
b = _abi.batch_from_rois(rois)
s = _abi.default_settings(8, False)
names = _lib.column_names(28, s)
k = names.index("GLSZM_ZP")
O = po.oracle_featurize(b, 28, s)
for rep in range(3):
    G = ctx.featurize_host(b, 28, s)
    bad = np.nonzero(~np.isclose(G[:, k], O[:, k], rtol=1e-9))[0]
    print("rep", rep, "bad rois", bad[:10], [(int(b.bbox_w[i]), int(b.bbox_h[i]), int(b.px_offset[i+1]-b.px_offset[i])) for i in bad[:5]], "got", G[bad[:3], k], "want", O[bad[:3], k])
    allbad = np.nonzero(~np.isclose(G, O, rtol=1e-5, equal_nan=True).all(axis=1))[0]
    print("   any-col bad rows:", allbad[:10])
print("dims of first 5:", [(int(b.bbox_w[i]), int(b.bbox_h[i])) for i in range(5)], "max side", int(max(b.bbox_w.max(), b.bbox_h.max())))
