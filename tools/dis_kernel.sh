#!/bin/bash
# Development aid: dump the gfx950 ISA of one kernel of a csrc/*.hip file.
#   tools/dis_kernel.sh render_bwd.hip 'render_backward_kernelILi32ELi64ELb1ELi4' /tmp/k.s [extra hipcc flags]
set -e
SRC=/root/repo/feature-3dgs_amd/csrc/$1; PAT=$2; OUT=$3; shift 3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -fno-slp-vectorize "$@" -S --cuda-device-only -o $OUT.all $SRC 2>/dev/null
awk -v pat="$PAT" '$0 ~ "^_Z.*"pat".*:" {f=1} f{print} f&&/^; Occupancy/{f=0}' $OUT.all > $OUT
grep -n "NumVgprs\|NumAgprs\|Occupancy\|ScratchSize" $OUT
