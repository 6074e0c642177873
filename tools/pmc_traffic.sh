#!/bin/bash
# HBM-side traffic of the U-Net forward per kernel, from the L2's fabric counters (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and
# WRITE_SIZE need separate passes (TCC has 4 counter slots: FETCH_SIZE takes 3, WRITE_SIZE 2); --pmc is combined with
# --kernel-trace only.  Run on the GPU box:   bash tools/pmc_traffic.sh > gpurun_out/traffic.txt
# The profiled command runs TWO full-size forwards (one denoise step + bench.py's per-kernel breakdown forward).
# Prints per kernel: launches, summed raw counter (KB as rocprofv3 reports it) and writes gpurun_out/$EW_ROUND_hbm_traffic.json (EW_ROUND, default r06) with
# the corrected per-forward totals: FETCH_SIZE x2 on gfx950 for wide coalesced reads (guide), WRITE_SIZE x1; the factors are
# re-checked on ln_kernel<1,4> (known bytes: 460800 x 320 rows, hi + lo read, fp16 written).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export EW_ROUND=${EW_ROUND:-r06}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_t
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_t -o p -- python $REPO/bench.py --steps 1 --warmup 0 --denoise-steps 1 --no-cpu-baseline --no-fp16-stream > /tmp/pmc_t.log 2>&1
  python - $c $REPO <<'PY'
import csv, glob, collections, sys, json, os
c, repo = sys.argv[1], sys.argv[2]
for f in glob.glob("/tmp/pmc_t/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"][:70]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    tot = sum(v[1] for v in agg.values())
    print(f"## {c}: total {tot:.4g} (raw KB) over {sum(v[0] for v in agg.values())} dispatches")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:26]:
        print(f"{k:72s} {n:5d} {v:14.5g}")
    ln = [(k, n, v) for k, (n, v) in agg.items() if "ln_kernelILi1ELi4" in k]
    out = os.path.join(repo, "gpurun_out", os.environ["EW_ROUND"] + "_hbm_traffic_%s.json" % c)
    json.dump({"counter": c, "raw_total_kb": tot, "dispatches": sum(v[0] for v in agg.values()),
               "ln_kernel_1_4": [{"launches": n, "raw_kb": v} for _, n, v in ln]}, open(out, "w"))
PY
done
python - $REPO <<'PY'
import json, os, sys
repo = sys.argv[1]
f = json.load(open(os.path.join(repo, "gpurun_out", os.environ["EW_ROUND"] + "_hbm_traffic_FETCH_SIZE.json")))
w = json.load(open(os.path.join(repo, "gpurun_out", os.environ["EW_ROUND"] + "_hbm_traffic_WRITE_SIZE.json")))
forwards = 2
read_b = f["raw_total_kb"] * 1024 * 2.0 / forwards      # gfx950: FETCH_SIZE counts 64 B per 128-B request
write_b = w["raw_total_kb"] * 1024 * 1.0 / forwards
# calibration on ln_kernel<1,4>: 25 launches per forward on 460800 x 320 rows: read hi (2 B) + lo8 (1 B), write 2 B per element
ln_elems = 460800 * 320
cal = {}
if f["ln_kernel_1_4"]:
    n = f["ln_kernel_1_4"][0]["launches"]; cal["fetch_factor_measured"] = n * ln_elems * 3 / (f["ln_kernel_1_4"][0]["raw_kb"] * 1024)
if w["ln_kernel_1_4"]:
    n = w["ln_kernel_1_4"][0]["launches"]; cal["write_factor_measured"] = n * ln_elems * 2 / (w["ln_kernel_1_4"][0]["raw_kb"] * 1024)
sys.path.insert(0, repo)
import bench
out = {"recorded_at": {"source_fingerprint": bench.source_fingerprint(), "residual_stream": os.environ.get("EW_RESIDUAL", "split"),
                       "round": int(os.environ["EW_ROUND"][1:])},
       "bytes_per_forward": read_b + write_b, "read_bytes_per_forward": read_b, "written_bytes_per_forward": write_b,
       "fetch_correction": 2.0, "write_correction": 1.0, "calibration_on_ln_kernel": cal,
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 1 --warmup 0 --denoise-steps 1 (two forwards); tools/pmc_traffic.sh"}
json.dump(out, open(os.path.join(repo, "gpurun_out", os.environ["EW_ROUND"] + "_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out))
PY
