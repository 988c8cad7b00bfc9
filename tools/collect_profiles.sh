#!/bin/bash
# Collect the rocprofv3 summaries behind bench.py's roofline object on the GPU box and leave
# them under gpurun_out/<tag>/ (copy the CSVs / JSON you want judged into profiles/).
#   gpurun -- 'bash tools/collect_profiles.sh r06'
# Kernel trace and counters are separate runs (gpurun refuses --pmc with trace domains other
# than --kernel-trace; counters in passes of their own as MI355X_MICROARCH.md prescribes).
set -u
TAG=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # what bench.py sets for itself (the profiler starts HIP first)

stats() {  # name, bench arguments
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o run --output-format csv -- \
        python "$R/bench.py" "$@" --no-counters > "$OUT/$name.json" 2> "$OUT/$name.err"
    cp "$(find "$OUT/$name" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats_$name.csv"
    grep '^{' "$OUT/$name.json" | tail -1 > "$OUT/${TAG}_bench_$name.json"
}
stats single_range --steps 100 --warmup 10 --no-cpu --sub-ranges 1
stats default --steps 100 --warmup 10 --no-cpu
stats cfg1 --config cfg1 --steps 100 --warmup 10 --no-cpu --sub-ranges 1
stats cfg1_default --config cfg1 --steps 100 --warmup 10 --no-cpu   # two ranges, class streams
stats cfg4 --config cfg4 --steps 100 --warmup 10 --no-cpu
stats cfg5 --config cfg5 --steps 40
stats driver --steps 20 --warmup 5   # the command the driver runs at round end
stats shard128 --steps 20 --warmup 5 --no-cpu --blends 128   # one GPU's shard of an 8-GPU job
stats shard128_100 --steps 100 --warmup 10 --no-cpu --blends 128
stats shard256 --steps 20 --warmup 5 --no-cpu --blends 256
stats shard512 --steps 20 --warmup 5 --no-cpu --blends 512
# the driver's command as the driver runs it (no profiler around it: the bench measures its own
# HBM counters in passes of its own)
python "$R/bench.py" --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_live_counters.json" 2> "$OUT/driver_live.err"
# one GPU's shard without the profiler around it, and one scene through the Python API
for nbl in 128 256 512; do
    python "$R/bench.py" --steps 20 --warmup 5 --no-cpu --no-counters --blends $nbl > "$OUT/${TAG}_bench_shard${nbl}_noprof.json" 2> /dev/null
done
python "$R/bench.py" --config cfg5 --steps 40 > "$OUT/${TAG}_bench_cfg5_noprof.json" 2> /dev/null
python "$R/tools/single_scene.py" 2> "$OUT/single_scene.err" | grep '^{' > "$OUT/${TAG}_bench_single_scene.json"
# the driver's window cold (no clock ramp) beside the default, and the ramp itself
python "$R/bench.py" --steps 20 --warmup 5 --ramp-ms 0 --no-cpu --no-counters > "$OUT/${TAG}_bench_driver_cold.json" 2> /dev/null
python "$R/tools/clock_ramp.py" > "$OUT/${TAG}_clock_ramp.txt" 2> /dev/null
# initialisation beside the fit, a small shard's enqueue time
python "$R/tools/init_time.py" --scenes 4 --out "$OUT/${TAG}_init.json" > /dev/null 2> "$OUT/init.err"
python "$R/tools/launch_overhead.py" --blends 128 > "$OUT/${TAG}_launch_overhead_128.txt" 2> /dev/null
for cfg in cfg1 cfg4; do
    python "$R/bench.py" --config $cfg --steps 100 --warmup 10 --no-cpu --no-counters > "$OUT/${TAG}_bench_${cfg}_noprof.json" 2> /dev/null
done
# the path a scarlet script calls: Blend objects in, fit_blends, fitted objects out
python "$R/bench.py" --facade --blends 1024 --steps 100 > "$OUT/${TAG}_bench_facade.json" 2> "$OUT/facade.err"

pmc() {  # name, counters...
    local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/pmc_$name" -o run --output-format csv -- \
        python "$R/bench.py" --steps 20 --warmup 2 --no-cpu --sub-ranges 1 --no-counters > /dev/null 2> "$OUT/pmc_$name.err"
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc valu SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
pmc busy SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU

python "$R/tools/hbm_counters.py" "$OUT" "$TAG" 1024
