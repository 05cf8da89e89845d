#!/bin/bash
# Register / LDS / scratch figures of the kernels of one translation unit (code-object metadata of its object file):
#   tools/kernel_regs.sh <unit: roi_features | roi_texture | roi_large | ...> [kernel-name substring]
U=$1; K=$2
OBJ=$(dirname $0)/../nyxus_amd/csrc/obj/$U.o
T=$(mktemp -d); cp $OBJ $T/u.o
( cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading u.o > /dev/null 2>&1 )
CO=$(ls $T/u.o.*gfx950* 2>/dev/null | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$CO" | python3 -c "
import sys, re, subprocess
name = None; d = {}
for line in sys.stdin:
    m = re.match(r'\s+-?\s*\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|agpr_count):\s+(.*)', line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == 'name':
        if v.startswith('_Z'): name = v
        continue
    if name: d.setdefault(name, {})[k] = v
for n, kv in sorted(d.items()):
    dn = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    if '$K' in dn:
        print(dn[:100].ljust(100), 'vgpr', kv.get('vgpr_count'), 'agpr', kv.get('agpr_count'), 'sgpr', kv.get('sgpr_count'), 'vspill', kv.get('vgpr_spill_count'), 'sspill', kv.get('sgpr_spill_count'), 'scratch', kv.get('private_segment_fixed_size'), 'lds', kv.get('group_segment_fixed_size'))
"
rm -rf $T
