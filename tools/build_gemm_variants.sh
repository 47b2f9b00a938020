#!/bin/bash
# alt_libs/liblmod_<name>.so: the library with gemm.hip rebuilt under extra -D flags (main-loop schedule knobs, timing
# ablations: G256_STAGE_POS, G256_M32, G256_ABL, G256_PRIO, G256_STAGGER — see gemm.hip).
# usage: build_gemm_variants.sh name1:"-DG256_STAGE_POS=2" name2:"-DG256_M32=1" ...     (run `make` first: other objects are reused)
set -e
cd "$(dirname "$0")/../llava-mod_amd/csrc"
export PATH=/opt/rocm/bin:$PATH
mkdir -p ../../alt_libs build
objs=$(ls build/*.o | grep -v "build/gemm")
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $flags -c gemm.hip -o build/gemm_$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC $objs build/gemm_$name.o -o ../../alt_libs/liblmod_$name.so &&
    rm build/gemm_$name.o && echo built $name ) &
done
wait
