cd $GRAFT_REPO_ROOT; O=gpurun_out/r5h; mkdir -p $O
for L in ffa112 ffa96; do EW_LIB_PATH=$PWD/evoworld_amd/libevoworld_hip_$L.so timeout 600 python -m pytest tests/test_gpu_ff_fused.py -x -q > $O/fftest_$L.log 2>&1; echo "$L: $(tail -1 $O/fftest_$L.log)"; done
for r in 1 2; do for L in "" ffa96 ffa112; do P=""; [ -n "$L" ] && P=$PWD/evoworld_amd/libevoworld_hip_$L.so; EW_LIB_PATH=$P timeout 300 python tools/experiments/exp46_ff_agpr.py 2>&1 | grep " us" ; done; done | tee $O/exp46_ff_agpr.txt
timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -s -k "split_operands or fp32_weights" 2>&1 | grep -E "rel-L2|passed|failed" | tee $O/unet_split.log
EW_FULL_FP32_WEIGHTS=1 timeout 1500 python -m pytest tests/test_gpu_unet.py -x -q -s -k full_size_forward > $O/fullsize_fp32w.log 2>&1; grep -E "rel-L2|passed|failed|oracle" $O/fullsize_fp32w.log
