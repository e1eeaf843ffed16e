#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration (GPU box only).
#   tools/profile.sh <tag> [bench.py args...]
# Separate passes: kernel trace + stats, then one --pmc pass per counter group
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
# the trace pass runs the DEFAULT bench (same steps / warm-up as the live
# number) so its per-kernel average is comparable with roofline.kernel_ms_mean
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu --no-extra $* > "$OUT/trace.log" 2>&1
ARGS="--steps 20 --warmup 3 --no-cpu --no-extra $*"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1
done
find "$OUT" -name "*.csv" | head -40
