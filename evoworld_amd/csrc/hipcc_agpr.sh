#!/bin/bash
# hipcc_agpr.sh <agprs> <src.hip> <out.o> [--kernel <mangled-name substring>] [--vgpr-form] [extra hipcc flags...]
# Compiles one HIP source like `hipcc -c`, but with the AccVGPR budget of its kernels (all of them, or those whose mangled name contains the
# --kernel substring) set to <agprs> registers; the rest of the wave's register file stays arch VGPRs.  hipcc has no source-level spelling for
# this: its attributor stamps "amdgpu-agpr-alloc"="0" on kernels without inline-asm AGPR operands (all MFMA accumulators in arch VGPRs), and a
# kernel that does touch AGPRs gets an even 128 / 128 split.  So the device side goes through LLVM IR: emit bitcode, rewrite the attribute, and
# finish with the same lld (device LTO code generation) / clang-offload-bundler / host-compile steps `hipcc -###` shows.  --vgpr-form keeps
# the builtin MFMAs in VGPR form (-amdgpu-mfma-vgpr-form) so that only inline-asm MFMAs with "a" operands use the AccVGPRs.
# Why: with the accumulators in arch VGPRs an MFMA and another wave's VALU instructions do not overlap on this chip
# (tools/experiments/mb_mfma_valu.hip: 8 MFMA + 32 v_fma = 209 ns, the SUM of 124 + 77; 177 ns with AccVGPR accumulators).
set -e
AGPR=$1; SRC=$2; OUT=$3; shift 3
KERNEL=""; VFORM=""
while [ "$1" = "--kernel" ] || [ "$1" = "--vgpr-form" ]; do
  if [ "$1" = "--kernel" ]; then KERNEL=$2; shift 2; else VFORM="-plugin-opt=-amdgpu-mfma-vgpr-form"; shift; fi
done
ROCM=${ROCM_PATH:-/opt/rocm}; LLVM=$ROCM/lib/llvm/bin; ARCH=${ARCH:-gfx950}
T=$(mktemp -d); trap 'rm -rf $T' EXIT
$ROCM/bin/hipcc --offload-arch=$ARCH "$@" --cuda-device-only -emit-llvm -c $SRC -o $T/dev.bc
$LLVM/llvm-dis $T/dev.bc -o $T/dev.ll
python3 - $T/dev.ll "$AGPR" "$KERNEL" <<'PY'
import re, sys
path, agpr, kern = sys.argv[1], sys.argv[2], sys.argv[3]
s = open(path).read()
groups = dict(re.findall(r'^attributes #(\d+) = \{(.*)\}$', s, re.M))
new_id = max(int(g) for g in groups) + 1
made = {}
n = 0
def fix(m):
    global new_id, n
    name, gid = m.group(2), m.group(4)
    if kern and kern not in name:
        return m.group(0)
    body = groups[gid]
    if '"amdgpu-agpr-alloc"' in body and '"amdgpu-agpr-alloc"="0"' not in body:
        return m.group(0)                      # already carries a budget of its own
    if gid not in made:
        made[gid] = new_id
        new_id += 1
    n += 1
    return m.group(1) + name + m.group(3) + '#%d' % made[gid] + m.group(5)
s = re.sub(r'^(define [^\n]*?@)([\w$.]+)(\([^\n]*?\)[^\n#]*)#(\d+)([^\n]*\{)$', fix, s, flags=re.M)
if not n:
    sys.exit('hipcc_agpr.sh: no kernel matched "%s"' % kern)
def with_budget(body):                         # a kernel with inline-asm "a" operands carries no attribute at all (hipcc then splits the file evenly)
    if '"amdgpu-agpr-alloc"="0"' in body:
        return body.replace('"amdgpu-agpr-alloc"="0"', '"amdgpu-agpr-alloc"="%s"' % agpr)
    return body.rstrip() + ' "amdgpu-agpr-alloc"="%s" ' % agpr
s += ''.join('\nattributes #%d = {%s}' % (nid, with_budget(groups[gid])) for gid, nid in made.items()) + '\n'
open(path, 'w').write(s)
print('hipcc_agpr.sh: %d kernel(s) -> %s AccVGPRs' % (n, agpr), file=sys.stderr)
PY
$LLVM/opt -passes=verify $T/dev.ll -o $T/dev2.bc                       # (.ll -> bitcode: there is no llvm-as in this ROCm)
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -plugin-opt=-amdgpu-internalize-symbols --lto-partitions=8 -plugin-opt=mcpu=$ARCH -plugin-opt=O3 --lto-CGO3 $VFORM --whole-archive -o $T/dev.out $T/dev2.bc --no-whole-archive
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
$ROCM/bin/hipcc --offload-arch=$ARCH "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c $SRC -o $OUT
