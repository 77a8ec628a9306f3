#!/bin/bash
# Profiles of one round, taken in one session so that they agree with each other (run on the
# GPU box through gpurun; outputs under gpurun_out/profiles_<tag>/, copied to profiles/ by hand):
#   kernel stats   rocprofv3 --kernel-trace --stats of bench.py
#   FETCH_SIZE     } separate --pmc passes of the same command (they do not fit one pass);
#   WRITE_SIZE     } tools/record_traffic.py turns them into profiles/<tag>_fill_traffic.json
#   SQ counters    three --pmc passes of tools/fill_once.py (VALU busy, wait states, LDS)
# usage: bash tools/profile_round.sh <tag>
set -eo pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 3 --no-filters --cpu-sample 0"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o bench -- $BENCH > "$OUT/bench_under_rocprof.json"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --output-format csv --pmc $c -d "$OUT/pmc_$c" -o bench -- $BENCH > /dev/null
    python3 "$ROOT/tools/summarize_pmc.py" "$(find "$OUT/pmc_$c" -name '*counter_collection.csv' | head -1)" \
        > "$OUT/${TAG}_bench_pmc_$(echo $c | tr 'A-Z' 'a-z').csv"
done
: > "$OUT/${TAG}_fill_pmc_sq.csv"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    d="$OUT/sq_$(echo $set | cut -d' ' -f2)"
    rocprofv3 --output-format csv --pmc $set -d "$d" -o fill -- python3 "$ROOT/tools/fill_once.py" 16384 3 > /dev/null 2>&1
    python3 "$ROOT/tools/summarize_pmc.py" "$(find "$d" -name '*counter_collection.csv' | head -1)" \
        | grep -E "Kernel_Name|fill_async_kernel<false, 0>" >> "$OUT/${TAG}_fill_pmc_sq.csv"
done
python3 "$ROOT/bench.py" --steps 20 --warmup 5 > "$OUT/${TAG}_bench.json"
ls -la "$OUT"
