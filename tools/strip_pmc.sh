#!/bin/bash
# tools/strip_pmc.sh: PMC counters (one rocprofv3 --pmc pass per counter
# group and strip shape; GPU box only) for config 5 through the row-group
# kernel and through the strip kernel (family 8) in a few shapes.
#   -> gpurun_out/strip_pmc/report.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/strip_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SHAPES=${STRIP_SHAPES:-"8,1,4,2,8 14,1,4,2,14 8,2,4,3,8"}
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE")
for shape in $SHAPES; do
  for grp in "${GROUPS_[@]}"; do
    name=$(echo "${shape}_$grp" | tr ' ,' '__' | cut -c1-48)
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/strip_sweep.py" --reps 1 --shapes $shape ${STRIP_ARGS:-} > "$OUT/$name.log" 2>&1
  done
done
python3 - "$OUT" <<'PY' | tee "$OUT/report.txt"
import collections, csv, glob, os, sys
out = sys.argv[1]
table = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, '*/'))):
    shape = os.path.basename(d.rstrip('/')).split('_')[0:5]
    shape = ','.join(shape)
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            k = r['Kernel_Name']
            fam = 'strip ' + shape if 'spmm_strip' in k else 'rowgroup' if 'spmm_rowgroup' in k else None
            if fam:
                per[(fam, r['Counter_Name'])].append(float(r['Counter_Value']))
        for (fam, c), v in per.items():
            table[fam][c] = v[-1]          # the last (timed) launch
counters = sorted({c for t in table.values() for c in t})
print('kernel'.ljust(24) + ''.join(c[:22].rjust(24) for c in counters))
for fam in sorted(table):
    print(fam.ljust(24) + ''.join(f'{table[fam].get(c, float("nan")):24.5g}' for c in counters))
PY
