#!/bin/bash
# tools/pmc_sweep.sh <tag> <sweep.py args...>: one rocprofv3 --pmc pass per
# counter group over the variants of tools/sweep.py --pmc 4 (GPU box only).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CSVS=""
if [ -z "${PMC_GROUPS:-}" ]; then
  PMC_GROUPS="FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum|TCC_EA0_RDREQ_sum TCC_BUBBLE_sum|GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"
fi
IFS='|' read -ra GROUPS_ARR <<< "$PMC_GROUPS"
for grp in "${GROUPS_ARR[@]}"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/sweep.py" --pmc 4 "$@" > "$OUT/$name.log" 2>&1
  CSVS="$CSVS $(find "$OUT/$name" -name '*counter_collection.csv' | head -1)"
  LOG="$OUT/$name.log"
done
python3 "$ROOT/tools/pmc_report.py" "$LOG" $CSVS | tee "$OUT/report.txt"
