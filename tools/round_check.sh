#!/bin/bash
# One GPU-box pass: parity suite, the bench line of every workload, profile.
#   tools/round_check.sh <tag>
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
for w in config2 headline config4 config5; do
  python bench.py --workload $w --steps 30 --warmup 5 --sets 1 --no-cpu --no-extra > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
bash tools/profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['config'].get('schedule', {}).get('family'), d['ms_per_step'], d['roofline']['frac'])
    except Exception as e:
        print(f, 'ERR', e)
PY
