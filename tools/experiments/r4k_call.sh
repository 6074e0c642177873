cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4k
./tools/experiments/mb_epi > gpurun_out/r4k/mb_epi.txt 2>&1; cat gpurun_out/r4k/mb_epi.txt
./tools/experiments/mb_epi >> gpurun_out/r4k/mb_epi.txt 2>&1; tail -9 gpurun_out/r4k/mb_epi.txt
