#!/bin/bash
# alt_libs/liblmod_<name>.so: the library with attn_bwd2.hip (or SRC=attn_fwd2 AGPRS=64 WPE=2) rebuilt under extra -D flags
# (timing ablations, tuning knobs).  usage: build_b2_variants.sh name1:"-DB2_ABL=1" name2:"-DB2_DEPTH=3" ...
set -e
cd "$(dirname "$0")/../llava-mod_amd/csrc"
export PATH=/opt/rocm/bin:$PATH
mkdir -p ../../alt_libs build
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  src=${SRC:-attn_bwd2}
  WPE=${WPE:-1} ./hipcc_agpr.sh $src.hip build/${src}_$name.o ${AGPRS:-256} $flags
  objs=$(ls build/*.o | grep -v "build/$src")
  hipcc --offload-arch=gfx950 -shared -fPIC $objs build/${src}_$name.o -o ../../alt_libs/liblmod_$name.so
  rm build/${src}_$name.o
  echo built $name
done
