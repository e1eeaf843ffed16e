#!/bin/bash
# tools/xaux_ab.sh: cache policy of the X loads (raw_buffer_load aux operand:
# 1 sc0, 2 nt, 16 sc1 and sums), one diagnostic library per policy
# (tools/build_diag.py xaux<N>), the product library first and last.
# GPU box only.  -> gpurun_out/xaux_ab.log
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
LOG=gpurun_out/xaux_ab.log
: > $LOG
for W in config5 headline config3; do
  ORD="group:32:8"; TUNE="10,1,2,1,3"; SETS=1; REPS=4
  [ $W != config5 ] && { ORD="group:1073741824:4"; TUNE="10,0,1,1"; SETS=2; REPS=10; }
  for V in product xaux1 xaux2 xaux3 xaux16 xaux17 product; do
    LIB=pyremap_amd/_lib/libremap_hip.so
    [ $V != product ] && LIB=tools/_build/libremap_hip_$V.so
    [ -f $LIB ] || continue
    echo "== $W $V" >> $LOG
    REMAP_HIP_LIB=$LIB python tools/sweep.py --workload $W --locality mesh --orders $ORD --tunes "$TUNE" --rounds 4 --reps $REPS --sets $SETS 2>&1 | tail -1 >> $LOG
  done
done
cat $LOG
