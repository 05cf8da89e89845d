import subprocess, sys, re, os
so = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
# extract fat binary code objects
data = open(so, 'rb').read()
# find ELF images for amdgcn inside the .so (clang offload bundle): search for "\x7fELF" with e_machine = 224 (EM_AMDGPU)
outs = []
i = 0
while True:
    i = data.find(b"\x7fELF", i + 1)
    if i < 0: break
    if data[i+18:i+20] == b"\xe0\x00":
        outs.append(i)
for n, off in enumerate(outs):
    # determine size through section header table
    import struct
    shoff = struct.unpack_from("<Q", data, off + 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", data, off + 0x3A)
    size = shoff + shentsize * shnum
    fn = f"/tmp/t/co_{n}.elf"
    open(fn, "wb").write(data[off:off+size])
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", fn], capture_output=True, text=True).stdout
    cur = {}
    for line in txt.splitlines():
        m = re.match(r"\s+\.(\w+):\s+(.*)", line)
        if not m: continue
        k, v = m.groups()
        if k == "name" and not v.startswith("'") and cur.get("_k"):
            pass
        cur[k] = v
        if k == "wavefront_size":
            name = cur.get("name", "")
            if pat in name:
                print(name[:110], "vgpr", cur.get("vgpr_count"), "sgpr", cur.get("sgpr_count"), "vspill", cur.get("vgpr_spill_count"), "sspill", cur.get("sgpr_spill_count"), "scratch", cur.get("private_segment_fixed_size"), "lds", cur.get("group_segment_fixed_size"))
            cur = {}
