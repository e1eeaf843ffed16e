#!/bin/bash
# One GPU-box pass: the parity suite, rocprofv3 profiles, the multi-rank
# rehearsals on this one GPU, the bench line of every workload.
#   gpurun -- 'bash tools/round_check.sh r02'      then, back in the build
#   container:  python tools/summarize_profile.py r02 config3 headline config4 config5
# (delete gpurun_out/prof_<tag> first: gpurun merges, it does not replace)
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out
# the DPP read hazard (inline-asm DPP operations: LLVM does not check them)
python tools/dpp_hazard_scan.py > gpurun_out/dpp_hazard_scan.log 2>&1 || { cat gpurun_out/dpp_hazard_scan.log; echo "FAIL: DPP hazard"; exit 1; }
tail -1 gpurun_out/dpp_hazard_scan.log
# no kernel of the library spills VGPRs (hipcc present: the box has ROCm)
KERNEL_REGS=--fail-on-spill python tools/kernel_regs.py > gpurun_out/kernel_regs.log 2>&1 || { tail -3 gpurun_out/kernel_regs.log; echo "FAIL: VGPR spills"; exit 1; }
echo "kernel_regs: $(grep -c vgpr gpurun_out/kernel_regs.log) kernels, no VGPR spill"
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
bash tools/profile.sh "$TAG" config3 headline config4 config5 > "gpurun_out/profile_$TAG.log" 2>&1
# N > 1 as the driver launches it, rehearsed with gloo and the ranks sharing
# this GPU (RCCL refuses several ranks on one device); and the distributed
# code path on one rank over RCCL
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo > gpurun_out/b_gloo2.json 2> gpurun_out/b_gloo2.err
python bench.py --steps 20 --warmup 5 --force-dist --no-cpu > gpurun_out/b_fd.json 2> gpurun_out/b_fd.err
for w in config2 config4 config5 headline; do
  python bench.py --workload $w --steps 30 --warmup 5 --sets 1 --no-cpu --no-extra > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err ) 2> gpurun_out/bench_n1.time
cp gpurun_out/bench_extra.json gpurun_out/bench_n1_extra.json
# the driver keeps an 8 KB tail of stdout: a longer line is an unmeasured round
BYTES=$(tail -n 1 gpurun_out/bench_n1.json | wc -c)
if [ "$BYTES" -ge 4096 ]; then echo "FAIL: bench line is $BYTES bytes (limit 4096)"; exit 1; fi
echo "bench line: $BYTES bytes; $(grep real gpurun_out/bench_n1.time)"
python - <<'PY'
import glob
import json
for f in sorted(glob.glob('gpurun_out/bench_*.json')) + \
        ['gpurun_out/b_gloo2.json', 'gpurun_out/b_fd.json']:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['n_gpus'], d['config'].get('schedule', {}).get('family'),
              round(d['roofline']['kernel_ms_mean'], 4),
              round(d['roofline']['frac'], 4))
    except Exception as e:   # noqa: BLE001
        print(f, 'ERR', e)
PY
# the committed summary must be younger than the kernels it describes
SUM=profiles/${TAG}_summary.md
if [ -f "$SUM" ] && command -v git > /dev/null && [ -d .git ]; then
  KERN=$(git log -1 --format=%ct -- pyremap_amd/csrc 2>/dev/null || echo 0)
  SUMT=$(git log -1 --format=%ct -- "$SUM" 2>/dev/null || echo 0)
  if [ "${SUMT:-0}" -lt "${KERN:-0}" ]; then
    echo "FAIL: $SUM is older than the last commit under pyremap_amd/csrc: regenerate it (tools/summarize_profile.py $TAG ...)"; exit 1
  fi
fi
