cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1; do
rm -rf /tmp/pf$m
rocprofv3 --kernel-trace --output-format csv -d /tmp/pf$m -- python $R/tools/prof_forward.py fused_ff=$m > /tmp/pf$m.log 2>&1
f=$(find /tmp/pf$m -name "*kernel_trace.csv" | head -1)
python $R/tools/csv_kernel_stats.py $f 3 > $R/gpurun_out/r03_e_fwd_fusedff$m.md
head -24 $R/gpurun_out/r03_e_fwd_fusedff$m.md | cut -c1-150
done
