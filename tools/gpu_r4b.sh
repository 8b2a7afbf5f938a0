#!/bin/bash
# round 4, second box: the staged k_lsd_rects / stage-parallel k_lsd_improve -- parity, headline with / without the screen, ADV
# headline, rocprofv3 kernel statistics of both refine levels.  gpurun_out/r4b/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=gpurun_out/r4b
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_line.py -m gpu -x -q --timeout 600 2>&1 | tail -4 | tee $O/tests_line.txt
timeout 900 python -m pytest tests/test_soak_gpu.py -m gpu -x -q -s --timeout 800 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/tests_soak_adv.txt
for flag in "" "--no-screen" "--refine adv"; do
echo -n "[$flag] headline: " | tee -a $O/ab.txt
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify $flag 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_launch'])" | tee -a $O/ab.txt
done
cd /tmp
for mode in std adv; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/stats_$mode" -o st -- \
  python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-verify --refine $mode > "$ROOT/$O/bench_stats_$mode.log" 2>&1
f=$(find "$ROOT/$O/stats_$mode" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$ROOT/$O/kernel_stats_$mode.csv" && head -12 "$f"
rm -rf "$ROOT/$O/stats_$mode"
done
exit 0
