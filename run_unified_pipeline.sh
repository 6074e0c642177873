#!/bin/bash
# N-segment loop-consistency run (generation -> reprojection -> 3D memory -> next segment) on the MI355X-native hot path: counterpart of the reference's run_unified_pipeline.sh
# (same variables, same flags).  With no checkpoint under $CKPT the U-Net is randomly initialised (RANDOM_INIT=true) and the
# VAE / CLIP / VGGT stages are the synthetic stand-ins of evoworld_amd.stages (override with STAGES=pkg.mod:factory).
set -e
cd "$(dirname "$0")"
export HSA_ENABLE_IPC_MODE_LEGACY=0

CKPT=${CKPT:-MODELS/evoworld_curve_unity}
BASE_FOLDER=${BASE_FOLDER:-example/case_000}
OUTPUT_ROOT=${OUTPUT_ROOT:-output}
SAVE_DIR=$OUTPUT_ROOT/$(basename $CKPT)/unified_demo
START_IDX=${START_IDX:-0}
NUM_DATA_PER_GPU=${NUM_DATA_PER_GPU:-1}
NUM_SEGMENTS=${NUM_SEGMENTS:-3}
CURVE_PATH=${CURVE_PATH:-true}
NUM_GPUS=${NUM_GPUS:-1}
STEPS=${STEPS:-25}

[ -d "$CKPT" ] || RANDOM_INIT=true
make -s -C evoworld_amd/csrc

CMD="unified_loop_consistency.py --unet_path $CKPT --svd_path $CKPT --base_folder $BASE_FOLDER --save_dir $SAVE_DIR \
 --num_data $((NUM_DATA_PER_GPU * NUM_GPUS)) --start_idx $START_IDX --num_segments $NUM_SEGMENTS --num_frames 25 \
 --num_inference_steps $STEPS --save_frames"
[ "$CURVE_PATH" = true ] && CMD="$CMD --curve_path"
[ "$RANDOM_INIT" = true ] && CMD="$CMD --random_init"
[ -n "$STAGES" ] && CMD="$CMD --stages $STAGES"

if [ "$NUM_GPUS" -gt 1 ]; then
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NUM_GPUS" --master-addr 127.0.0.1 --master-port ${MASTER_PORT:-29511} $CMD
else
  python $CMD
fi
echo "Unified pipeline completed: $SAVE_DIR"
