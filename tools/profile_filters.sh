#!/bin/bash
# Kernel stats of the operators outside the headline step (groves, Fourier destripe, lagoon
# branch, D8, box mean) and SQ counters of the groves kernel; outputs under
# gpurun_out/profiles_<tag>/.  usage: bash tools/profile_filters.sh <tag>
set -eo pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/fstats" -o filters -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --cpu-sample 0 > "$OUT/filters_under_rocprof.json"
cp "$(find "$OUT/fstats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_filters_kernel_stats.csv"
: > "$OUT/${TAG}_groves_pmc_sq.csv"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY"; do
    d="$OUT/gsq_$(echo $set | cut -d' ' -f2)"
    rocprofv3 --output-format csv --pmc $set -d "$d" -o groves -- python3 "$ROOT/tools/groves_time.py" 16384 > /dev/null 2>&1
    python3 "$ROOT/tools/summarize_pmc.py" "$(find "$d" -name '*counter_collection.csv' | head -1)" \
        | grep -E "Kernel_Name|groves_stream_kernel" >> "$OUT/${TAG}_groves_pmc_sq.csv"
done
python3 "$ROOT/tools/sweep_sizes.py" "$OUT/${TAG}_size_sweep.json"
ls -la "$OUT"
