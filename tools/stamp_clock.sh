#!/bin/bash
# in-kernel clock + phase breakdown of a few G1 layers (diagnostic -DCCNET_STAMP build)
export CCNET_HIP_LIB=$PWD/build_ab/lib_stamp.so
for a in "32 32 640 640 3 23" "32 64 320 320 3 23" "32 16 1280 1280 3 23" "32 64 320 2560 1 20 0 2" "32 64 320 320 1 23 1" "32 32 640 640 1 23 1"; do
  echo "== N H C1 Cout R cfg [res] [act]: $a"; timeout -k 10 120 python tools/stamp_one.py $a 2>&1 | grep -v amdgpu | head -12
done
