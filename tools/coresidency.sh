#!/bin/bash
# Development aid: does a 512-thread convolution kernel leave room for update
# workgroups of another range on the same CU, and does that pay?  One bench line per
# combination of convolution workgroup size, update packing and number of ranges.
#   gpurun -- 'bash tools/coresidency.sh r04'
set -u
TAG=${1:-r04}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
run() {  # label, env..., -- bench args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env "${envs[@]}" python "$R/bench.py" --no-cpu "$@" 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['roofline']['phases_ms']
        print('%-44s %-28s %9.0f blend-it/s  %.4f ms/it  conv %.4f update %.4f  sum-wall %.4f  ranges %d' % (
            '$label', ' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], p['conv'], p['update'],
            p['total'] - d['ms_per_step'], d['config']['sub_ranges_per_gpu']))
" "$@" | tee -a "$OUT/coresidency.txt"
}
for blends in 1024 256 128; do
  for steps in 20 100; do
    A="--blends $blends --steps $steps --warmup 5"
    run "conv1024 default" X=1 -- $A
    run "conv512 default" SMI_CONV_WORKGROUP=512 -- $A
    run "conv512 pack4" SMI_CONV_WORKGROUP=512 SMI_UPDATE_PACK=4 -- $A
    run "conv512 pack4 ranges4" SMI_CONV_WORKGROUP=512 SMI_UPDATE_PACK=4 -- $A --sub-ranges 4
    run "conv512 pack4 ranges2" SMI_CONV_WORKGROUP=512 SMI_UPDATE_PACK=4 -- $A --sub-ranges 2
    run "conv512 nostage pack1" SMI_CONV_WORKGROUP=512 SMI_STAGE_PLAN=0 SMI_UPDATE_PACK=1 -- $A
    run "conv1024 pack4" SMI_UPDATE_PACK=4 -- $A
    run "conv1024 pack8" SMI_UPDATE_PACK=8 -- $A
  done
done
