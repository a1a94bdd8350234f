#!/bin/bash
# builds an experimental copy of the product library: tools/_exp/lib_<tag>.so = md_kernels.hip compiled with the extra flags given + the product's other objects
# usage: tools/exp_build.sh <tag> [extra hipcc flags, e.g. -DMD_TRACE -DMD_INTER8_ONLY]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p tools/_exp
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 ${MD_OPT:--O3} ${MD_SCHED--mllvm -amdgpu-sched-strategy=max-ilp} -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -Wno-align-mismatch -Wno-unused-variable \
    -c svt-hevc_amd/csrc/md_kernels.hip -o tools/_exp/md_$tag.o
objs=$(ls svt-hevc_amd/build/*.o | grep -v md_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_exp/lib_$tag.so tools/_exp/md_$tag.o $objs
echo tools/_exp/lib_$tag.so
