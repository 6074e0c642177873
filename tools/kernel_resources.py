#!/usr/bin/env python3
"""VGPR / spill / scratch / LDS of every kernel in a HIP object (from the code object's metadata notes).
Usage: python tools/kernel_resources.py a.o [b.o]   (two objects: side by side, differences flagged)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def res(obj):
    d = tempfile.mkdtemp()
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"],
                   check=True, capture_output=True)
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: re.search(r"\." + k + r":\s*(\S+)", blk)
        name = g("name").group(1)
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::|void |\(GemmP, SkP\)|\(GemmP\)", "", name)
        out[name] = tuple(int(g(k).group(1)) for k in ("vgpr_count", "agpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"))
    return out


a = res(sys.argv[1])
b = res(sys.argv[2]) if len(sys.argv) > 2 else None
print(f"{'kernel':70s} vgpr agpr spill scratch lds")
for k in sorted(set(a) | set(b or {})):
    x, y = a.get(k), (b or {}).get(k)
    if b is None:
        print(f"{k[:70]:70s} {x}")
    elif x != y:
        print(f"{k[:70]:70s} {x} -> {y}")
