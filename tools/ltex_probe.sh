#!/bin/bash
# per-kernel times of the large-ROI texture path by family subset (mixed batch, lanes off so that kernels run alone)
export NYXHIP_NO_LANES=1
for fam in 4 8 16 28; do
  echo "== families $fam"
  tools/ktrace.sh ltex_f$fam python3 tools/size_legs.py --families $fam --no-sweep | grep -E "ltex|roi_texture_kernel<true"
done
