#!/usr/bin/env python3
"""Per-kernel ISA comparison of two builds of one HIP object (round-6 prune check: a source clean-up must leave the shipped kernels' machine code
unchanged).  Usage: python tools/isa_diff.py old.o new.o   -> lists kernels whose disassembly differs / that exist on one side only."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    d = tempfile.mkdtemp()
    co = os.path.join(d, "co")
    fat = os.path.join(d, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"],
                   check=True, capture_output=True)
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in txt.split("\n"):
        m = re.match(r"^[0-9a-f]* ?<(.+)>:$", line.strip())
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None and line.strip():
            out[cur].append(re.sub(r"\s*//.*", "", line.strip()))     # (branch targets are symbolic offsets: unchanged code disassembles identically)
    dem = {}
    for k, v in out.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
        dem[name] = (len(v), hashlib.sha256("\n".join(v).encode()).hexdigest()[:12])
    return dem


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
same = 0
for k in sorted(set(a) | set(b)):
    if k not in a:
        print("only new:", k, b[k])
    elif k not in b:
        print("only old:", k, a[k])
    elif a[k] != b[k]:
        print("DIFFERS :", k, a[k], "->", b[k])
    else:
        same += 1
print(f"{same} kernels identical")
