#!/bin/bash
# Development aid: build a variant of the library with extra flags for ONE translation unit
# (default kernels.hip) into tools/ab/lib_<name>.so; the other objects come from csrc/build.
#   tools/variant.sh noplan -DSMI_EXP_NOPLAN=1
#   SRC=fused_conv.hip tools/variant.sh conv_x -DFOO=1
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=${SRC:-kernels.hip}
cd "$R/scarlet_amd/csrc"
mkdir -p /tmp/var "$R/tools/ab"
extra="-ffp-contract=off"
case "$SRC" in fused_conv*) extra="-ffp-contract=off -ffp-contract=fast -fno-slp-vectorize -fno-signed-zeros";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $extra -Wall -Wno-unused-function "$@" -c "$SRC" -o /tmp/var/$name.o
objs=$(ls build/*.o | grep -v "build/$SRC.o")
/opt/rocm/bin/hipcc $objs /tmp/var/$name.o -shared -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib -o "$R/tools/ab/lib_$name.so"
echo built tools/ab/lib_$name.so
