"""The eight-wave launch of the 64-level GLCM kernel against the four-wave one (NYXHIP_G16_W4=1), same batch, two processes: the tables must be
the same bits (the feature pass on two waves per angle reproduces the one-wave reductions).  python tools/g16_w8_check.py"""
import os, subprocess, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from nyxus_amd import _abi, _lib
from tests import synth
ctx = _lib.Context(0)
h = []
for gd in (64, 33, 17):
    for sym in (0, 1):
        s = _abi.default_settings(gd); s.glcm_symmetric = sym
        rois = synth.random_rois(60, seed=gd + sym, rmax=45, value_modes=(4096, 256, 60000, 8))
        b = _abi.batch_from_rois(rois)
        for mask in (2, 3):
            T = ctx.featurize_host(b, mask, s)
            h.append(T.tobytes())
b = synth.tile_batch(2)
T = ctx.featurize_host(b, 3, _abi.default_settings(64)); h.append(T.tobytes())
import hashlib
print(hashlib.sha256(b"".join(h)).hexdigest())
''' % ROOT
outs = []
for w4 in ("0", "1"):
    env = dict(os.environ, NYXHIP_G16_W4=w4)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-2000:]); sys.exit(1)
    outs.append(r.stdout.strip().splitlines()[-1])
print("eight waves:", outs[0][:16], " four waves:", outs[1][:16], " ->", "identical" if outs[0] == outs[1] else "DIFFERENT")
sys.exit(0 if outs[0] == outs[1] else 2)
