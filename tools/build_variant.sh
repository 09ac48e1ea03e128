#!/bin/bash
# build_var/<name>.so = the library with attention.hip (or another source, $3) compiled with extra flags:
#   tools/build_variant.sh v1 "-fno-slp-vectorize" [attention]
set -e
name=$1; flags=$2; src=${3:-attention}
cd "$(dirname "$0")/../ccnet_amd/csrc"
mkdir -p ../../build_var
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value $flags -Rpass-analysis=kernel-resource-usage -c $src.hip -o ../../build_var/${src}_$name.o 2> ../../build_var/${src}_$name.res
grep -E "ScratchSize \[bytes/lane\]: [1-9]" ../../build_var/${src}_$name.res && echo "SPILL in variant $name" || true
objs=""
for s in gemm_conv norm attention elementwise train fused_ff fused_xattn; do
  if [ $s = $src ]; then objs="$objs ../../build_var/${src}_$name.o"; else objs="$objs $s.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_var/lib_$name.so $objs
echo built build_var/lib_$name.so
