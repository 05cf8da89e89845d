#!/bin/bash
# end-of-round profiles: metric group trace + PMC, every kernel, family times, large-ROI texture kernels, mixed batch kernels
TAG=${1:-r05z}
tools/profile_bench.sh $TAG --no-extras --tile-path-tiles 0 > gpurun_out/${TAG}_profile_bench.log 2>&1
tools/profile_all.sh $TAG > gpurun_out/${TAG}_profile_all.log 2>&1
cp gpurun_out/prof_all_$TAG/summary.txt gpurun_out/${TAG}_all_kernels_summary.txt
bash tools/fam_bench.sh 32 64 3072 28 896 127 4095 > gpurun_out/${TAG}_family_ms.txt 2>&1
{ export NYXHIP_NO_COOP_TEX_SC2=1; bash tools/ltex_probe.sh; } > gpurun_out/${TAG}_large_texture_by_family.txt 2>&1
tools/ktrace.sh ${TAG}_mixed4 python3 tools/size_legs.py --families 31 --no-sweep > /dev/null 2>&1
cp gpurun_out/ktrace_${TAG}_mixed4.txt gpurun_out/${TAG}_mixed_sizes_config4_kernels.txt
tools/ktrace.sh ${TAG}_mixed python3 tools/size_legs.py --no-sweep > /dev/null 2>&1
cp gpurun_out/ktrace_${TAG}_mixed.txt gpurun_out/${TAG}_mixed_sizes_kernels.txt
ls -la gpurun_out/${TAG}_*
