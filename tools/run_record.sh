# Round-5 record run (one gpurun call): PMC traffic of the final sources first (so that bench.py's roofline.traffic carries a
# matching fingerprint), then the default bench.py line, the rocprofv3 kernel statistics of the same command, episode and
# reprojection benches.  Outputs land in gpurun_out/ and are copied to profiles/ by hand.
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
bash $R/tools/pmc_traffic.sh > $R/gpurun_out/r05_j_hbm_traffic.txt 2>&1
tail -3 $R/gpurun_out/r05_j_hbm_traffic.txt
cp $R/gpurun_out/r05_hbm_traffic.json $R/profiles/r05_hbm_traffic.json
cd $R
python bench.py > gpurun_out/r05_j_bench.json 2> gpurun_out/r05_j_bench.err
tail -c 900 gpurun_out/r05_j_bench.json
cd /tmp
rm -rf /tmp/prof_f
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -- python $R/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-fp16-stream > /tmp/prof_f.log 2>&1
f=$(find /tmp/prof_f -name "*kernel_trace.csv" | head -1)
python $R/tools/csv_kernel_stats.py $f 3 > $R/gpurun_out/r05_j_three_forwards_kernel_stats.md
s=$(find /tmp/prof_f -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && head -40 $s > $R/gpurun_out/r05_j_rocprofv3_kernel_stats.csv
head -12 $R/gpurun_out/r05_j_three_forwards_kernel_stats.md
cd $R
python bench_episode.py > gpurun_out/r05_j_bench_episode.json 2> gpurun_out/r05_j_bench_episode.err; tail -c 700 gpurun_out/r05_j_bench_episode.json; tail -3 gpurun_out/r05_j_bench_episode.err
python bench_reproject.py > gpurun_out/r05_j_bench_reproject.json 2>/dev/null; tail -c 300 gpurun_out/r05_j_bench_reproject.json
# per-shape breakdown -> practical-floor table; effective clock of the long kernels
EW_BENCH_FULL_BREAKDOWN=1 EW_BENCH_BY_SHAPE=1 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream > gpurun_out/r05_j_bd.json 2> gpurun_out/r05_j_bd.txt
python tools/floor_table.py gpurun_out/r05_j_bd.txt > gpurun_out/r05_j_floor_table.md; tail -12 gpurun_out/r05_j_floor_table.md
bash tools/pmc_clock.sh r05 > gpurun_out/r05_j_effective_clock.txt 2>&1; tail -12 gpurun_out/r05_j_effective_clock.txt
