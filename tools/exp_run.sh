#!/bin/bash
# GPU box: time several builds of the library (exp/libf3dgs_hip_<name>.so, produced from patched copies of csrc/) with the same bench.
#   tools/exp_run.sh "c3 c2" base w5 ...
CFGS=$1; shift
LIB=feature-3dgs_amd/csrc/libf3dgs_hip.so
cp $LIB /tmp/lib_orig.so
for v in "$@"; do
  cp exp/libf3dgs_hip_$v.so $LIB
  for c in $CFGS; do echo "== $v $c"; python tools/quick_bench.py $c 20 "" "" 2>&1 | grep -v amdgpu.ids; done
done
cp /tmp/lib_orig.so $LIB
