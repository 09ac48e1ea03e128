#!/bin/bash
# A/B the committed library against the working-tree build on the conv micro-benchmark (same box, same process order)
for L in build_ab/lib_old.so ccnet_amd/libccnet_hip.so; do echo "== $L"; CCNET_HIP_LIB=$PWD/$L timeout -k 10 300 python tools/bench_conv.py --cfgs=-1 --cold --iters 24 "$@" 2>&1 | grep -v amdgpu; done
