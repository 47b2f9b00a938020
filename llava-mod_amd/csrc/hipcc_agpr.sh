#!/bin/bash
# hipcc_agpr.sh <src.hip> <out.o> <agprs> [extra hipcc flags]
# Compile a HIP source whose kernels own a FIXED number of accumulator registers through inline asm.
# hipcc has no source-level spelling for LLVM's "amdgpu-agpr-alloc" function attribute; without it a kernel that mentions
# accumulator registers gets its register budget split evenly (128 VGPR + 128 AGPR at 2 waves per SIMD).  So the device
# side is built in the driver's own steps (see `hipcc -###`): device IR -> attribute added -> code object -> fat binary ->
# host object that embeds it.
set -euo pipefail
src=$1; out=$2; agprs=$3; shift 3
HIPCC=${HIPCC:-hipcc}
ARCH=${ARCH:-gfx950}
LLVM=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
FLAGS="--offload-arch=$ARCH -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-spill-vgpr-to-agpr=0 $*"
$HIPCC $FLAGS --cuda-device-only -emit-llvm -S -o "$tmp/dev.ll" "$src" 2> >(grep -v 'hip-link' >&2 || true)
WPE=${WPE:-2}        # waves per SIMD the kernels' __launch_bounds__ ask for: (512, 2) -> 2, (256, 1) -> 1
grep -q "\"amdgpu-waves-per-eu\"=\"$WPE\"" "$tmp/dev.ll" || { echo "hipcc_agpr.sh: no kernel with amdgpu-waves-per-eu=$WPE found in $src" >&2; exit 1; }
sed -i "s/\"amdgpu-waves-per-eu\"=\"$WPE\"/\"amdgpu-waves-per-eu\"=\"$WPE\" \"amdgpu-agpr-alloc\"=\"$agprs\"/" "$tmp/dev.ll"
$LLVM/clang -x ir -target amdgcn-amd-amdhsa -mcpu=$ARCH -O3 -fPIC -mllvm -amdgpu-spill-vgpr-to-agpr=0 -c -o "$tmp/dev.o" "$tmp/dev.ll"
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$tmp/dev.out" "$tmp/dev.o"
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH \
    -input=/dev/null -input="$tmp/dev.out" -output="$tmp/dev.hipfb"
$HIPCC $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$tmp/dev.hipfb" -c -o "$out" "$src" 2> >(grep -v 'hip-link' >&2 || true)
if [ -n "${KEEP_ISA:-}" ]; then
  $LLVM/clang -x ir -target amdgcn-amd-amdhsa -mcpu=$ARCH -O3 -mllvm -amdgpu-spill-vgpr-to-agpr=0 -S -o "$KEEP_ISA" "$tmp/dev.ll"
fi
