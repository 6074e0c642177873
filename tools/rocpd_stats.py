#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (markdown/CSV-ish).
Usage: python tools/rocpd_stats.py <results.db> [--skip-first-ms X]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels order by start").fetchall() if "name" in cols else []
    agg = {}
    for name, s, e in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    span = (rows[-1][2] - rows[0][1]) / 1e3 if rows else 0
    print(f"# kernels: {len(rows)} dispatches, sum of durations {tot/1e3:.2f} ms, first-start..last-end span {span/1e3:.2f} ms")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {a[0]} | {a[1]/1e3:.3f} | {a[1]/a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} | {100*a[1]/tot:.1f} |")


if __name__ == "__main__":
    main()
