#!/bin/bash
# usage: tools/pmc_run.sh <python script> <kernel-name substring ...>   (run on the GPU box; prints summed counters per kernel)
SCRIPT=$1; shift
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES"; do
  rm -rf /tmp/pmc
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $SCRIPT > /dev/null 2>&1
  python - "$@" <<'PY'
import csv,glob,collections,sys
pats=sys.argv[1:]
for f in glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in agg.items():
        if any(p in k for p in pats): print(k, {a:f"{b:.3g}" for a,b in v.items()})
PY
done
