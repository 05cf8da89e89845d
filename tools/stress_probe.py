"""Repeated calls of mixed sizes / family sets through one context: results must repeat bit for bit and device memory must not
grow once the largest workspace has been seen (the context keeps its scratch between calls)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from tests import synth

ctx = _lib.Context(0)
s = _abi.default_settings(8)
batches = [(_abi.batch_from_rois(synth.random_rois(n, seed=sd, rmax=r)), m)
           for (n, sd, r, m) in [(50, 1, 12, _abi.FAM_ALL), (400, 2, 30, _abi.FAM_NORTH_STAR), (5, 3, 60, _abi.FAM_ALL),
                                 (1000, 4, 8, 3), (120, 5, 40, _abi.FAM_ALL & ~_abi.FAM_GABOR)]]
first = [ctx.featurize_host(b, m, s) for b, m in batches]
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
bad = 0
worst = [0.0]
seen = {}
for it in range(60):
    for k, (b, m) in enumerate(batches):
        G = ctx.featurize_host(b, m, s)
        same = (G == first[k]) | (np.isnan(G) & np.isnan(first[k]))
        if not same.all():
            bad += 1
            names = _lib.column_names(m, s)
            cols = sorted({names[c] for c in np.nonzero(~same)[1]})
            rel = np.nanmax(np.abs(G - first[k])[~same] / np.maximum(np.abs(first[k])[~same], 1e-300))
            worst[0] = max(worst[0], float(rel))
            for c in cols: seen[c] = seen.get(c, 0) + 1
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("columns that ever differed:", sorted(seen.items(), key=lambda t: -t[1])[:40])
print("largest relative difference:", worst[0])
print("non-repeatable results:", bad, "| free device memory before / after 300 calls:", free0 >> 20, "/", free1 >> 20, "MiB")
