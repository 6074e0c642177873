#!/bin/bash
# per-dispatch FETCH_SIZE / WRITE_SIZE (KB) of the GEMM launches of a script: usage tools/pmc_fetch_one.sh <script.py>
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_f
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_f -o p -- python $1 > /tmp/pmc_f.log 2>&1
  python - $c <<'PY'
import csv, glob, sys
c = sys.argv[1]
for f in glob.glob("/tmp/pmc_f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "gemm" in r["Kernel_Name"]:
            print(c, r["Kernel_Name"][:60], f'{float(r["Counter_Value"]) / 1e3:10.1f} MB (raw)')
PY
done
