# Round record run (one gpurun call): PMC traffic of the final sources first (so that bench.py's roofline.traffic carries a
# matching fingerprint), then the default bench.py line, the rocprofv3 kernel statistics of the same command, episode and
# reprojection benches, per-shape breakdown -> floor table (true-peak + yardstick columns, vendor rates of the same call), effective clock.
# Usage on the GPU box: bash tools/run_record.sh <tag>   (tag e.g. r06_d; outputs in gpurun_out/<tag>_*, copied to profiles/ by hand)
set -x
R=$GRAFT_REPO_ROOT
T=${1:-r06_x}
export EW_ROUND=${T%%_*}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
bash $R/tools/pmc_traffic.sh > $R/gpurun_out/${T}_hbm_traffic.txt 2>&1
tail -3 $R/gpurun_out/${T}_hbm_traffic.txt
cp $R/gpurun_out/${EW_ROUND}_hbm_traffic.json $R/profiles/${EW_ROUND}_hbm_traffic.json
bash $R/tools/pmc_clock.sh $EW_ROUND > $R/gpurun_out/${T}_effective_clock.txt 2>&1; tail -12 $R/gpurun_out/${T}_effective_clock.txt
cp $R/gpurun_out/${EW_ROUND}_clock.json $R/profiles/${EW_ROUND}_clock.json
cd $R
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 900 gpurun_out/${T}_bench.json
cd /tmp
rm -rf /tmp/prof_f
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -- python $R/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-fp16-stream > /tmp/prof_f.log 2>&1
f=$(find /tmp/prof_f -name "*kernel_trace.csv" | head -1)
python $R/tools/csv_kernel_stats.py $f 3 > $R/gpurun_out/${T}_three_forwards_kernel_stats.md
python $R/tools/steady_state_forward.py $f > $R/gpurun_out/${T}_steady_state_forward.md 2>&1; head -6 $R/gpurun_out/${T}_steady_state_forward.md
s=$(find /tmp/prof_f -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && head -40 $s > $R/gpurun_out/${T}_rocprofv3_kernel_stats.csv
head -12 $R/gpurun_out/${T}_three_forwards_kernel_stats.md
cd $R
python bench_episode.py > gpurun_out/${T}_bench_episode.json 2> gpurun_out/${T}_bench_episode.err; tail -c 700 gpurun_out/${T}_bench_episode.json; tail -3 gpurun_out/${T}_bench_episode.err
python bench_reproject.py > gpurun_out/${T}_bench_reproject.json 2> gpurun_out/${T}_bench_reproject.err; tail -c 900 gpurun_out/${T}_bench_reproject.json
timeout 600 python tools/experiments/exp30_vs_hipblaslt.py > gpurun_out/${T}_vs_hipblaslt.txt 2>&1
EW_BENCH_FULL_BREAKDOWN=1 EW_BENCH_BY_SHAPE=1 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream > gpurun_out/${T}_bd.json 2> gpurun_out/${T}_bd.txt
CLK=$(python -c "import bench; print(round(bench.committed_clock()['ghz'], 3))" 2>/dev/null)     # time-weighted over the launches of >= 300 us (short ones read high)
python tools/floor_table.py gpurun_out/${T}_bd.txt ${CLK:+--clock $CLK} --vendor gpurun_out/${T}_vs_hipblaslt.txt > gpurun_out/${T}_floor_table.md; tail -12 gpurun_out/${T}_floor_table.md
