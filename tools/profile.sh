#!/bin/bash
# Collect the rocprofv3 evidence for the round (GPU box only).
#   tools/profile.sh <tag> [workload ...]      default workloads: config3
# 1. kernel trace + stats of THE DRIVER'S COMMAND (`bench.py --gpus 1 --steps
#    20 --warmup 5`): its per-dispatch rows give the average duration of the
#    metric kernel inside the timed region, to set beside the live number;
# 2. per workload, separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not
#    fit one pass on gfx950) over `bench.py --workload W --no-extra --no-cpu`.
# The program stands directly behind `--` (no env/bash hop: the profiler's
# preloaded library initialises the GPU before the program starts).
set -u
TAG=$1; shift
WORKLOADS=${*:-config3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
for W in $WORKLOADS; do
  SETS=3; STEPS=12
  case $W in config5|config4|headline) SETS=1; STEPS=6;; esac
  ARGS="--workload $W --steps $STEPS --warmup 2 --sets $SETS --no-cpu --no-extra"
  GROUPS_W="FETCH_SIZE|WRITE_SIZE"
  [ "$W" = config3 ] && GROUPS_W="FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
  IFS='|' read -ra GARR <<< "$GROUPS_W"
  for grp in "${GARR[@]}"; do
    name=$(echo $grp | tr ' ' '_')
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_${W}_$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_${W}_$name.json" 2> "$OUT/pmc_${W}_$name.err"
  done
done
find "$OUT" -name "*.csv" | sed "s|$ROOT/||"
