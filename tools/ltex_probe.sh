#!/bin/bash
# per-kernel times of the large-ROI texture path by family subset (mixed batch, lanes off so that kernels run alone); then the
# timeline of the last call with all three families
export NYXHIP_NO_LANES=1
for fam in ${LTEX_FAMS:-4 8 16 28}; do
  echo "== families $fam"
  tools/ktrace.sh ltex_f$fam python3 tools/size_legs.py --families $fam --no-sweep | grep -E "ltex|roi_texture_kernel<true"
done
python tools/timeline.py ltex_f28 48 | grep -E "ltex|fill"
