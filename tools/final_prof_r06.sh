#!/bin/bash
# end-of-round-6 profiles (run on the GPU box): everything tools/final_prof.sh collects, then what this round changed --
# the grey-depth-64 launch pair (kernel times + instruction mix), the tile path's traffic, the mixed batch with all twelve families.
TAG=${1:-r06z}
bash tools/final_prof.sh $TAG
bash tools/kstats.sh --no-extras --gray-depth 64 > gpurun_out/${TAG}_gd64_kernels.txt 2>&1
bash tools/pmc.sh mix roi_features ${TAG}gd64mix -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 --gray-depth 64 >> gpurun_out/${TAG}_gd64_kernels.txt 2>&1
bash tools/pmc.sh traffic all ${TAG}tile -- python3 tools/tile_ab.py 1000 8 > gpurun_out/${TAG}_tilepath_traffic.txt 2>&1
bash tools/ktrace.sh ${TAG}tile python3 tools/tile_ab.py 1000 8 >> gpurun_out/${TAG}_tilepath_traffic.txt 2>&1
bash tools/ktrace.sh ${TAG}allfam python3 tools/size_legs.py --families 4095 --no-sweep > gpurun_out/${TAG}_mixed_sizes_all_families_kernels.txt 2>&1
bash tools/pmc.sh traffic roi_texture ${TAG}textraffic -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 --families 28 > gpurun_out/${TAG}_texture_traffic.txt 2>&1
# the pair-driven GLCM of the smallest size class at grey depth 64 against the workgroup kernel on the same ROIs (49 / 253 px): time + instruction mix
{
for r in 4 9; do
  python3 tools/small_probe.py $r 2 64; NYXHIP_NO_SMALL=1 python3 tools/small_probe.py $r 2 64
  python3 tools/small_probe.py $r 3 64; NYXHIP_NO_SMALL=1 python3 tools/small_probe.py $r 3 64
  bash tools/pmc.sh mix roi_small ${TAG}sg$r -- python3 tools/small_probe.py $r 2 64 2>&1 | tail -1
  NYXHIP_NO_SMALL=1 bash tools/pmc.sh mix roi_features_kernel_g16 ${TAG}wg$r -- python3 tools/small_probe.py $r 2 64 2>&1 | tail -1
done
for r in 4 6 9; do python3 tools/small_probe.py $r 3 8; NYXHIP_NO_SMALL=1 python3 tools/small_probe.py $r 3 8; done
for r in 4 9; do for g in 8 64; do python3 tools/small_mixed_probe.py $r 3 $g; NYXHIP_NO_ADAPT=1 python3 tools/small_mixed_probe.py $r 3 $g; done; done
} > gpurun_out/${TAG}_small_glcm64.txt 2>&1
bash tools/gap_probe.sh > gpurun_out/${TAG}_headline_gaps.txt 2>&1
bash tools/timeline_probe.sh python3 tools/size_legs.py --families 4095 --no-sweep > gpurun_out/${TAG}_all_families_timeline.txt 2>&1
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo rc=$? >> gpurun_out/${TAG}_bench.err
ls -la gpurun_out/${TAG}_*
