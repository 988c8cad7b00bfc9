#!/bin/bash
# Development aid: one bench line per environment setting.
#   tools/ab_env.sh out.txt "bench args" "ENV1=a ENV2=b" "ENV1=c" ...
out=$1; args=$2; shift 2
for e in "$@"; do
  line=$(env $e timeout 600 python bench.py $args --no-cpu --no-counters 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['roofline']['phases_ms']
print('%10.0f blend-it/s  %.4f ms/it  conv %.4f update %.4f ranges %s' % (d['value'], d['ms_per_step'], p['conv'], p['update'], d['config']['sub_ranges_per_gpu']))")
  echo "$e | $args | $line" >> $out
done
