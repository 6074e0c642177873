#!/bin/bash
# Effective graphics clock per kernel of the U-Net forward: one `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` pass over
# tools/prof_forward.py (three full-size forwards); clock = GRBM_GUI_ACTIVE cycles of a dispatch / its duration.  The counter ticks in the
# graphics clock domain while the GPU is busy, so the ratio is the clock the kernel actually ran at under the power limit (DVFS).
# Writes gpurun_out/<tag>_clock.json + prints a table.   Usage on the GPU box: bash tools/pmc_clock.sh [tag]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c
EW_PROF_FORWARDS=3 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_c -o p -- python $REPO/tools/prof_forward.py > /tmp/pmc_c.log 2>&1
python - $REPO $TAG <<'PY'
import csv, glob, collections, sys, json, os
repo, tag = sys.argv[1], sys.argv[2]
cc = glob.glob("/tmp/pmc_c/**/*counter_collection.csv", recursive=True)
kt = glob.glob("/tmp/pmc_c/**/*kernel_trace.csv", recursive=True)
if not cc:
    print("no counter_collection.csv:", open("/tmp/pmc_c.log").read()[-2000:]); sys.exit(1)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r.get("Dispatch_Id") or r.get("Correlation_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    if "Start_Timestamp" in r and r["Start_Timestamp"]:
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    else:
        ns = dur.get(r["Dispatch_Id"], 0)
    if ns <= 0:
        continue
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    if k.startswith("_ZN12_GLOBAL__N_1"):
        k = k[len("_ZN12_GLOBAL__N_1"):].lstrip("0123456789")
    k = (k[:k.find("(")] if "(" in k else k)[:48]
    a = agg[k]
    a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += ns
# GRBM_GUI_ACTIVE comes back summed over the XCDs (8 instances on MI355X): one forward's kernels gave 17 "GHz" raw
XCDS = 8
for a in agg.values():
    a[1] /= XCDS
tot_c = sum(a[1] for a in agg.values()); tot_ns = sum(a[2] for a in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
print(f"# effective clock = GRBM_GUI_ACTIVE / {XCDS} XCDs / duration, three forwards under rocprofv3 --pmc (kernels serialised).  The counter window is a few us")
print(f"# longer than the timestamped kernel: only launches of >= 300 us are listed (short ones read 5-30 % high)")
print(f"{'kernel':64s} {'n':>5s} {'ms':>9s} {'GHz':>6s}")
out = {"overall_ghz": tot_c / tot_ns, "kernels": []}
for k, (n, c, ns) in rows[:40]:
    if ns / n < 300e3:
        continue
    print(f"{k:64s} {n:5d} {ns / 1e6:9.2f} {c / ns:6.3f}")
    out["kernels"].append({"kernel": k, "launches": n, "ms": ns / 1e6, "ghz": c / ns})
print(f"{'ALL':64s} {sum(a[0] for a in agg.values()):5d} {tot_ns / 1e6:9.2f} {tot_c / tot_ns:6.3f}")
json.dump(out, open(os.path.join(repo, "gpurun_out", f"{tag}_clock.json"), "w"), indent=1)
PY
