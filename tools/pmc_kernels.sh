#!/bin/bash
# SQ counters of the kernels of one step (exploration): three --pmc passes of tools/step_phases.py,
# condensed per kernel.  usage (GPU box): bash tools/pmc_kernels.sh <out-prefix> [size]
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$1
SIZE=${2:-16384}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/summary.csv"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
    d="$OUT/$(echo $set | cut -d' ' -f2)"
    rocprofv3 --output-format csv --pmc $set -d "$d" -o k -- python3 "$ROOT/tools/step_phases.py" $SIZE > /dev/null 2>&1
    python3 "$ROOT/tools/summarize_pmc.py" "$(find "$d" -name '*counter_collection.csv' | head -1)" \
        | grep -E "Kernel_Name|hub_dist|fill_async|certify|hub_edges" >> "$OUT/summary.csv"
done
cut -c1-60,200- "$OUT/summary.csv" | sed 's/(anonymous namespace):://' | head -80
