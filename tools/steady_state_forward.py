#!/usr/bin/env python3
"""Kernel list of ONE steady-state U-Net forward out of a rocprofv3 --kernel-trace CSV of bench.py (the whole-run summary
cannot tell load-time weight packing from the forward).  A forward starts with its two ew_sinusoid_embed_f16 launches; the segment from the start
of the LAST-BUT-ONE forward to the start of the last one is one denoise step = forward + the fused Euler / CFG kernel.  Prints per-kernel counts
and time of that segment and lists every kernel that is not one of this library's (at::native::*, rocclr copies, Tensile GEMMs).
Usage: python tools/steady_state_forward.py kernel_trace.csv"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:90]


rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a forward opens with TWO back-to-back sinusoid launches (timestep, added time ids); the single ones are the cached time_pos_embed of the first forward
starts = [i for i in range(len(names) - 1) if "sinusoid_kernel" in names[i] and "sinusoid_kernel" in names[i + 1]]
if len(starts) < 2:
    raise SystemExit(f"need at least two forwards in the trace (found {len(starts)} sinusoid pairs)")
# with three or more forwards take the step between the SECOND and the THIRD (both inside the denoise loop of one clip: bench.py's per-kernel
# breakdown forward and its end-of-clip checks come after the last step)
a, b = (starts[1], starts[2]) if len(starts) >= 3 else (starts[-2], starts[-1])
seg = rows[a:b]
agg = collections.OrderedDict()
for r in seg:
    t = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0])
    t[0] += 1
    t[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
OURS = ("gemm", "conv_small_n", "ff320", "attn_", "gn_", "ln_kernel", "sinusoid", "euler_cfg", "nchw_to_nhwc", "nhwc_to_nchw", "_GLOBAL__N_")
foreign = {k: v for k, v in agg.items() if not any(o in k for o in OURS)}
tot = sum(v[1] for v in agg.values())
print(f"# one steady-state denoise step (dispatches {a}..{b - 1} of {len(rows)}): {len(seg)} launches, {tot:.2f} ms of kernel time")
print(f"# kernels not from libevoworld_hip.so in that segment: {sum(v[0] for v in foreign.values())} launches, {sum(v[1] for v in foreign.values()):.3f} ms"
      + ("" if foreign else "  (none: no at::native::* / copy / Tensile kernel in a steady-state forward)"))
for k, (n, ms) in foreign.items():
    print(f"#   FOREIGN {k}: {n} launches, {ms:.3f} ms")
print("| kernel | launches | ms |\n|---|---|---|")
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"| {k} | {n} | {ms:.3f} |")
