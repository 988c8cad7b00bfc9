#!/bin/bash
# Development aid: the counter passes of tools/collect_profiles.sh only (one range, 20 iterations),
# reduced by tools/hbm_counters.py:  gpurun -- 'bash tools/pmc_quick.sh q1'
set -u
TAG=${1:-q}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
pmc() {
    local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/pmc_$name" -o run --output-format csv -- \
        python "$R/bench.py" --steps 20 --warmup 2 --no-cpu --sub-ranges 1 --no-counters > /dev/null 2> "$OUT/pmc_$name.err"
}
pmc valu SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
pmc busy SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU
pmc lds2 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA
python "$R/tools/hbm_counters.py" "$OUT" "$TAG" 1024
grep -E 'fused_conv|update_kernel_reg' "$OUT/${TAG}_counters.csv"
find "$OUT" -name 'pmc_*' -type d -exec rm -rf {} + 2>/dev/null
