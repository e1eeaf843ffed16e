#!/bin/bash
# tools/pmc_one.sh <tag> <bench.py args...>: PMC counters of one bench.py
# workload (one rocprofv3 --pmc pass per counter group; GPU box only), the
# per-kernel averages printed as a table -> gpurun_out/pmc_<tag>/report.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES")
n=0
for grp in "${GROUPS_[@]}"; do
  n=$((n+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$n" -- python3 "$ROOT/bench.py" --no-cpu --no-extra --steps 10 --warmup 3 "$@" > "$OUT/g$n.log" 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu --no-extra --steps 10 --warmup 3 "$@" > "$OUT/trace.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/report.txt"
import collections, csv, glob, os, sys
out = sys.argv[1]
table = collections.defaultdict(dict)
for path in glob.glob(os.path.join(out, 'g*', '**', '*counter_collection.csv'), recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name']
        if 'spmm_' not in k:
            continue
        per[(k.split('<')[0].split('::')[-1], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in per.items():
        v = v[len(v) // 2:]                     # the timed launches
        table[k][c] = sum(v) / len(v)
for path in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True):
    for r in csv.DictReader(open(path)):
        if 'spmm_' in r['Name']:
            table[r['Name'].split('<')[0].split('::')[-1]]['avg_ns'] = float(r['AverageNs'])
for k, t in table.items():
    print(k)
    for c in sorted(t):
        print(f'   {c:32s} {t[c]:16.6g}')
PY
rm -rf "$OUT"/g*/ "$OUT"/trace
