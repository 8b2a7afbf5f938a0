#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest orb gpu (pyramid kernel)"
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_ref_orb.py -m gpu -x -q --timeout 600 2>&1 | tail -4 | tee gpurun_out/pytest_orb.log
echo "== orb kernel times (1024 frames): pyramid, FAST, octree, orient+brief"
timeout 300 python tools/orb_time.py 2>&1 | tail -1 | tee gpurun_out/orb_time2.log
echo "== force-dist"
timeout 600 python bench.py --steps 4 --warmup 1 --force-dist --no-extras --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/fd.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/fd.log').read().strip().splitlines()[-1]); print("force-dist value", d["value"], d.get("rccl"))
PY
echo "== nsplit / batch sweep"
for cfg in "6144 3" "6144 4" "6144 6" "6144 8" "8192 4" "8192 8" "4096 4"; do
  set -- $cfg
  timeout 600 python bench.py --steps 6 --warmup 2 --batch $1 --nsplit $2 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $1 nsplit $2 ->', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" | tee -a gpurun_out/sweep.log
done
exit 0
