#!/bin/bash
# hipcc_agpr.sh <agprs> <src.hip> <out.o> [extra hipcc flags...]
# Compiles one HIP source like `hipcc -c`, but with the AccVGPR budget of every kernel in it set to <agprs> registers (the rest of the
# wave's register file stays arch VGPRs).  hipcc has no source-level spelling for this: its attributor stamps "amdgpu-agpr-alloc"="0" on
# kernels without inline-asm AGPR operands (all MFMA accumulators in arch VGPRs), and any kernel that does touch AGPRs gets an even
# 128 / 128 split.  So the device side goes through LLVM IR: emit bitcode, rewrite the attribute, and finish with the same lld (device LTO
# code generation) / clang-offload-bundler / host-compile steps `hipcc -###` shows.
# Why: with the accumulators in arch VGPRs an MFMA and another wave's VALU instructions do not overlap on this chip
# (tools/experiments/mb_mfma_valu.hip: 8 MFMA + 32 v_fma = 209 ns, the SUM of 124 + 77; 177 ns with AccVGPR accumulators).
set -e
AGPR=$1; SRC=$2; OUT=$3; shift 3
ROCM=${ROCM_PATH:-/opt/rocm}; LLVM=$ROCM/lib/llvm/bin; ARCH=${ARCH:-gfx950}
T=$(mktemp -d); trap 'rm -rf $T' EXIT
$ROCM/bin/hipcc --offload-arch=$ARCH "$@" --cuda-device-only -emit-llvm -c $SRC -o $T/dev.bc
$LLVM/llvm-dis $T/dev.bc -o $T/dev.ll
grep -q '"amdgpu-agpr-alloc"="0"' $T/dev.ll || { echo "hipcc_agpr.sh: no \"amdgpu-agpr-alloc\"=\"0\" attribute in the device IR of $SRC (compiler changed?)" >&2; exit 1; }
sed -i "s/\"amdgpu-agpr-alloc\"=\"0\"/\"amdgpu-agpr-alloc\"=\"$AGPR\"/" $T/dev.ll
$LLVM/opt -passes=verify $T/dev.ll -o $T/dev2.bc                       # (.ll -> bitcode: there is no llvm-as in this ROCm)
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -plugin-opt=-amdgpu-internalize-symbols --lto-partitions=8 -plugin-opt=mcpu=$ARCH -plugin-opt=O3 --lto-CGO3 --whole-archive -o $T/dev.out $T/dev2.bc --no-whole-archive
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
$ROCM/bin/hipcc --offload-arch=$ARCH "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c $SRC -o $OUT
