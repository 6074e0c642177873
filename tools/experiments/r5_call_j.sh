cd $GRAFT_REPO_ROOT; R=$PWD; O=$R/gpurun_out/r5j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_gen3.py -x -q -k "conv3x3" > $O/halo_test.log 2>&1; tail -4 $O/halo_test.log
cd /tmp && export TMPDIR=/tmp
for h in 0 1; do
  rm -rf /tmp/pmc_t
  EW_G3_HALO=$h rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_t -o p -- python $R/bench.py --steps 1 --warmup 0 --denoise-steps 1 --no-cpu-baseline --no-fp16-stream > /tmp/pmc_t.log 2>&1
  python - $h <<'PY' | tee $O/fetch_halo_$h.txt
import csv, glob, collections, sys
h = sys.argv[1]
for f in glob.glob("/tmp/pmc_t/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        k = r["Kernel_Name"][:70]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    tot = sum(v[1] for v in agg.values())
    print(f"## EW_G3_HALO={h}  FETCH_SIZE total {tot * 1024 * 2 / 2 / 1e9:.1f} GB per forward (raw KB x 1024 x 2 correction / 2 forwards) over {sum(v[0] for v in agg.values())} dispatches")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"{k:72s} {n:5d} {v * 1024 * 2 / 2 / 1e9:9.2f} GB / forward")
PY
done
