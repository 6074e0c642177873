cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a
./tools/experiments/mb_feed > gpurun_out/r4a/mb_feed.txt 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r4a/hipblaslt -o hb --output-format csv -- python tools/experiments/exp40_hipblaslt_names.py > gpurun_out/r4a/exp40.txt 2>&1
ls -R gpurun_out/r4a/hipblaslt | head; 
tail -40 gpurun_out/r4a/mb_feed.txt
