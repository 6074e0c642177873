set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python bench.py > gpurun_out/r03_c_bench.json 2> gpurun_out/r03_c_bench.err
tail -c 600 gpurun_out/r03_c_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -- python $R/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-fp16-stream > /tmp/prof_c.log 2>&1
f=$(find /tmp/prof_c -name "*kernel_trace.csv" | head -1)
python $R/tools/csv_kernel_stats.py $f 3 > $R/gpurun_out/r03_c_three_forwards_kernel_stats.md
head -12 $R/gpurun_out/r03_c_three_forwards_kernel_stats.md
bash $R/tools/pmc_traffic.sh > $R/gpurun_out/r03_c_hbm_traffic.txt 2>&1
tail -3 $R/gpurun_out/r03_c_hbm_traffic.txt
cd $R
python bench_episode.py > gpurun_out/r03_c_bench_episode.json 2> gpurun_out/r03_c_bench_episode.err; tail -c 700 gpurun_out/r03_c_bench_episode.json; tail -3 gpurun_out/r03_c_bench_episode.err
python bench_reproject.py > gpurun_out/r03_c_bench_reproject.json 2>/dev/null; tail -c 300 gpurun_out/r03_c_bench_reproject.json
