#!/bin/bash
# HBM-side traffic of the U-Net forward per kernel, from the L2's fabric counters (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and
# WRITE_SIZE need separate passes (TCC has 4 counter slots: FETCH_SIZE takes 3, WRITE_SIZE 2).  Run on the GPU box:
#   bash tools/pmc_traffic.sh > gpurun_out/traffic.txt
# Prints per kernel: launches, summed raw counter (KB as rocprofv3 reports it).  Corrections are applied by the reader:
# FETCH_SIZE x2 on gfx950 for wide coalesced reads; WRITE_SIZE calibrated on ln_kernel (writes rows*C*2 bytes).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_t
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_t -o p -- python $REPO/bench.py --steps 1 --warmup 0 --denoise-steps 1 --no-cpu-baseline > /tmp/pmc_t.log 2>&1
  python - $c <<'PY'
import csv, glob, collections, sys
c = sys.argv[1]
for f in glob.glob("/tmp/pmc_t/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"][:60]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    tot = sum(v[1] for v in agg.values())
    print(f"## {c}: total {tot:.4g} (raw units) over {sum(v[0] for v in agg.values())} dispatches")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"{k:62s} {n:5d} {v:14.5g}")
PY
done
