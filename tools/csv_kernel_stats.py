#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV (kernel_trace.csv).  Usage: python tools/csv_kernel_stats.py <csv> [n_forwards]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:80]


rows = list(csv.DictReader(open(sys.argv[1])))
nf = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = {}
for r in rows:
    k = short(r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += d
tot = sum(a[1] for a in agg.values())
print(f"# {len(rows)} dispatches, sum of kernel durations {tot / 1e3:.2f} ms ({tot / 1e3 / nf:.2f} ms per forward over {nf:g})")
print("| kernel | calls/fwd | ms/fwd | avg us | % |")
print("|---|---|---|---|---|")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| {k} | {a[0] / nf:.1f} | {a[1] / 1e3 / nf:.3f} | {a[1] / a[0]:.1f} | {100 * a[1] / tot:.1f} |")
