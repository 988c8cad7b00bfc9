#!/bin/bash
# retry wrapper: tools/_gpu.sh <timeout> <logfile>   (runs tools/_r.sh on the GPU box)
T=${1:-1500}; L=${2:-/tmp/gpu.log}
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- 'bash tools/_r.sh' > $L 2>&1
  rc=$?
  if grep -q "status=transient" $L; then sleep 60; continue; fi
  break
done
echo "done rc=$rc" >> $L
