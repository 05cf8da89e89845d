#!/bin/bash
# early-exit builds of the texture kernel (-DNYX_TEX_EXIT_AT=k -> gpurun_scratch/libtex_<k>.so) for tools/tex_phases.sh and
# tools/tex_phase_insts.sh; the other translation units are compiled once into /tmp/objs
set -e
cd "$(dirname "$0")/../nyxus_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
mkdir -p /tmp/objs ../../gpurun_scratch
for s in roi_features roi_shape roi_dependence roi_moments tile_assembly nyxhip_api; do
  if [ ! -f /tmp/objs/$s.o ] || [ $s.hip -nt /tmp/objs/$s.o ] || [ device_math.h -nt /tmp/objs/$s.o ] || [ roi_kernel.h -nt /tmp/objs/$s.o ]; then
    /opt/rocm/bin/hipcc $F -c -o /tmp/objs/$s.o $s.hip 2>/dev/null &
  fi
done
wait
for k in ${@:-0 1 2 3 4 5 6 7 8 9}; do
  ( /opt/rocm/bin/hipcc $F -DNYX_TEX_EXIT_AT=$k -c -o /tmp/objs/tex_$k.o roi_texture.hip 2>/dev/null &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../gpurun_scratch/libtex_$k.so /tmp/objs/roi_features.o /tmp/objs/roi_shape.o /tmp/objs/roi_dependence.o /tmp/objs/roi_moments.o /tmp/objs/tile_assembly.o /tmp/objs/nyxhip_api.o /tmp/objs/tex_$k.o ) &
  if [ $((k % 4)) = 3 ]; then wait; fi
done
wait
ls ../../gpurun_scratch/
