#!/bin/bash
# Development aid: the same bench lines for several library builds (tools/ab/lib_<name>.so), alternating.
#   tools/ab_bench.sh out.txt "name1 name2 ..." "bench args" [repeats]
out=$1; names=$2; args=$3; rep=${4:-2}
for r in $(seq 1 $rep); do
  for n in $names; do
    lib=tools/ab/lib_$n.so
    line=$(SCARLET_AMD_LIB=$PWD/$lib timeout 600 python bench.py $args --no-cpu --no-counters 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['roofline']['phases_ms']
print('%10.0f blend-it/s  %.4f ms/it  conv %.4f update %.4f' % (d['value'], d['ms_per_step'], p['conv'], p['update']))")
    echo "$n | $args | $line" >> $out
  done
done
