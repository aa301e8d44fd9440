#!/bin/bash
# Development aid: build a named variant of the product library from the CURRENT csrc/ into exp/libf3dgs_hip_<name>.so
# (git-ignored; tools/exp_run.sh times the variants against each other on the GPU box).
#   tools/exp_build.sh maxilp RENDER_FLAGS="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/expb/$NAME; D=$B/pkg/csrc
rm -rf $B && mkdir -p $D $B/include && cp -p $ROOT/include/*.h $B/include/
cp -p $ROOT/feature-3dgs_amd/csrc/*.{hip,h,cpp} $ROOT/feature-3dgs_amd/csrc/Makefile $D/
# objects of files this experiment does not touch (same flags): taken from the in-tree build when they are newer than their source
for o in $ROOT/feature-3dgs_amd/csrc/*.o; do cp -p $o $D/; done
for f in ${EXP_TOUCH:-render_bwd_pl.hip render_fwd.hip render_bwd.hip}; do touch $D/$f; done
make -C $D -j8 "$@" libf3dgs_hip.so > $B/build.log 2>&1 || { tail -30 $B/build.log; exit 1; }
mkdir -p $ROOT/exp && cp $D/libf3dgs_hip.so $ROOT/exp/libf3dgs_hip_$NAME.so
echo "built exp/libf3dgs_hip_$NAME.so"
