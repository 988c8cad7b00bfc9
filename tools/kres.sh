#!/bin/bash
# Register / scratch / occupancy summary of the kernels of one HIP source (compile only).
#   tools/kres.sh scarlet_amd/csrc/kernels.hip [filter] [extra flags...]
src=$1; filt=${2:-.}; shift; shift
cd "$(dirname "$src")"
extra="-ffp-contract=off"
case "$(basename "$src")" in fused_conv*) extra="-ffp-contract=fast -fno-slp-vectorize -fno-signed-zeros";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $extra "$@" -Rpass-analysis=kernel-resource-usage \
    -c "$(basename "$src")" -o /dev/null 2>&1 |
  awk '/Function Name/ {name=$(NF-1)} /TotalSGPRs/ {s=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {print name, "vgpr", v, "sgpr", s, "scratch", sc, "occ", o, "lds", $(NF-1)}' |
  c++filt | sed 's/smi::(anonymous namespace):://; s/(smi::BatchView[^)]*)//' | grep -E "$filt"
