#!/bin/bash
# Development aid: build libscarlet_amd.so with extra compiler flags into tools/ab/lib_<name>.so
# (git-ignored; travels to the GPU box) for A/B runs through SCARLET_AMD_LIB.
#   tools/build_variant.sh <name> "<extra flags>" [git-rev] ["<flags for fused_conv.hip only>"]
#   (git-rev: build that revision's csrc; "" = the working tree)
set -e
name=$1; extra=$2; rev=$3; conv=$4
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/smi_variant_$name
rm -rf $tmp; mkdir -p $tmp/scarlet_amd $tmp/include $root/tools/ab
if [ -n "$rev" ]; then
  (cd $root && git archive $rev scarlet_amd/csrc include) | tar -x -C $tmp
else
  cp -r $root/scarlet_amd/csrc $tmp/scarlet_amd/; cp $root/include/*.h $tmp/include/
  rm -rf $tmp/scarlet_amd/csrc/build
fi
make -C $tmp/scarlet_amd/csrc -j8 EXTRA="$extra" CONV_EXTRA="$conv" > $tmp/build.log 2>&1 || { tail -20 $tmp/build.log; exit 1; }
cp $tmp/scarlet_amd/libscarlet_amd.so $root/tools/ab/lib_$name.so
echo "built tools/ab/lib_$name.so"
