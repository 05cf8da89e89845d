#!/bin/bash
for sd in 21 22; do NYXHIP_GABOR_MODE=2 python tools/gabor_fuzz.py save /tmp/a.npy $sd 20000 > /dev/null; python tools/gabor_fuzz.py save /tmp/b.npy $sd 20000 > /dev/null; echo "gabor_fuzz seed $sd: $(python tools/gabor_fuzz.py cmp /tmp/a.npy /tmp/b.npy)"; done
export GABOR_FUZZ_MAX_SIDE=16; NYXHIP_GABOR_MODE=2 python tools/gabor_fuzz.py save /tmp/a.npy 23 30000 > /dev/null; python tools/gabor_fuzz.py save /tmp/b.npy 23 30000 > /dev/null; echo "gabor_fuzz small seed 23: $(python tools/gabor_fuzz.py cmp /tmp/a.npy /tmp/b.npy)"
export GABOR_FUZZ_MAX_SIDE=125; NYXHIP_GABOR_MODE=2 python tools/gabor_fuzz.py save /tmp/a.npy 24 5000 > /dev/null; python tools/gabor_fuzz.py save /tmp/b.npy 24 5000 > /dev/null; echo "gabor_fuzz large seed 24: $(python tools/gabor_fuzz.py cmp /tmp/a.npy /tmp/b.npy)"
unset GABOR_FUZZ_MAX_SIDE
echo "bank fuzz: $(timeout 900 python tools/gabor_bank_fuzz.py 31 150 2>&1 | tail -1)"
echo "family fuzz: $(timeout 900 python tools/family_fuzz.py 41 20 2>&1 | tail -1)"
echo "subset fuzz: $(timeout 900 python tools/subset_fuzz.py 42 30 2>&1 | tail -1)"
echo "tile fuzz: $(timeout 900 python tools/tile_fuzz.py 43 20 2>&1 | tail -1)"
echo "glcm fuzz: $(timeout 900 python tools/glcm_fuzz.py 44 30 2>&1 | tail -1)"
echo "ltex fuzz: $(timeout 900 python tools/ltex_fuzz.py 45 40 2>&1 | tail -1)"
