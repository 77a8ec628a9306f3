#!/bin/bash
# SQ counters of every kernel of one command (exploration): passes of rocprofv3 --pmc, condensed
# per kernel.  usage (GPU box): bash tools/pmc_cmd.sh <out-dir-under-gpurun_out> <grep-pattern> -- python3 script args...
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$1; PAT=$2; shift 3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/summary.csv"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
    d="$OUT/$(echo $set | cut -d' ' -f2)"
    rocprofv3 --output-format csv --pmc $set -d "$d" -o k -- "$@" > /dev/null 2>&1
    python3 "$ROOT/tools/summarize_pmc.py" "$(find "$d" -name '*counter_collection.csv' | head -1)" \
        | grep -E "$PAT" >> "$OUT/summary.csv" || true
done
python3 - "$OUT/summary.csv" <<'PY'
import csv, sys
d = {}
for r in csv.reader(open(sys.argv[1])):
    if len(r) < 5: continue
    d.setdefault(r[0][:70], {})[r[1]] = float(r[4])
for k, v in d.items():
    print(k)
    print("   " + "  ".join(f"{c}={x:.4g}" for c, x in sorted(v.items())))
PY
