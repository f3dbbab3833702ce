#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel in a HIP object / shared library (the gfx950 code object's metadata).
usage: python tools/kernel_resources.py [ray3d_amd/libray3d_hip.so]"""
import re
import struct
import subprocess
import sys
import tempfile
import zlib

path = sys.argv[1] if len(sys.argv) > 1 else "ray3d_amd/libray3d_hip.so"
data = open(path, "rb").read()
out = []
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
pos = 0
while True:
    i = data.find(MAGIC, pos)
    if i < 0:
        break
    n, = struct.unpack_from("<Q", data, i + 24)
    p = i + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        triple = data[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        if "gfx950" in triple and size:
            out.append(data[i + off:i + off + size])
    pos = i + 24
if not out and b"CCOB" in data:
    i = data.find(b"CCOB")
    # compressed bundle: header {magic, version u16, method u16, ...}; try zlib/zstd from the first plausible offset
    raise SystemExit("compressed offload bundle (CCOB): pass an uncompressed object (-Xclang -no-offload-compress?)")
for k, blob in enumerate(out):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(blob)
        name = f.name
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", name], capture_output=True, text=True).stdout
    cur = {}
    for line in txt.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key in ("agpr_count", "args") and cur.get("name"):
            pass
        if key == "name" and "args" not in cur.get("_ctx", ""):
            pass
        cur[key] = val
        if key == "wavefront_size":      # last field of a kernel's record
            if "symbol" in cur:
                print("%-28s vgpr %3s agpr %3s sgpr %3s  vgpr_spill %3s sgpr_spill %3s  scratch %5s B  lds %6s B  max_wg %s" % (
                    cur.get("symbol", "?").replace(".kd", ""), cur.get("vgpr_count"), cur.get("agpr_count"), cur.get("sgpr_count"),
                    cur.get("vgpr_spill_count"), cur.get("sgpr_spill_count"), cur.get("private_segment_fixed_size"),
                    cur.get("group_segment_fixed_size"), cur.get("max_flat_workgroup_size")))
            cur = {}
