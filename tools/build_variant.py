#!/usr/bin/env python
"""Development tool: build a variant of the library with extra -D flags on chosen sources into tools/bin/<name>.so
(use it through AUDIOCAPTION_HIP_LIB).   python tools/build_variant.py NAME source.hip:-DFLAG[,-DFLAG2] [source2.hip:...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiocaption_amd import build as B

name = sys.argv[1]
extra = {}
for a in sys.argv[2:]:
    src, flags = a.split(":", 1)
    extra[src] = flags.split(",")
out_dir = os.path.join(ROOT, "tools", "bin", "var_" + name)
os.makedirs(out_dir, exist_ok=True)
objs, procs = [], []
for src in B.SOURCES:
    obj = os.path.join(out_dir, src.replace(".hip", ".o"))
    if src not in extra:   # reuse the shipped object when present
        shipped = os.path.join(B.HERE, "build", src.replace(".hip", ".o"))
        if os.path.exists(shipped):
            objs.append(shipped)
            continue
    cmd = [B._hipcc(), "-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj] + B.FLAGS + B.NO_PACKED_F32 + extra.get(src, [])
    procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    objs.append(obj)
for p in procs:
    out = p.communicate()[0].decode()
    assert p.returncode == 0, out
so = os.path.join(ROOT, "tools", "bin", f"lib{name}.so")
subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so] + objs)
print("built", so)
