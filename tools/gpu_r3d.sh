#!/bin/bash
# full-residency counters of k_lsd_grow (VERDICT r2 item 2's SQ_WAIT_ANY figure) and kernel stats of the small-batch paths
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=gpurun_out/r3d
mkdir -p $O
export TMPDIR=/tmp
bash tools/pmc_grow_detail.sh 6144 2>&1 | grep -v amdgpu.ids | tee $O/grow_full_residency.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/share512" -o st -- \
  python "$ROOT/bench.py" --batch 512 --nsplit 1 --rows 376 --cols 1241 --nfeatures 2000 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-verify > "$ROOT/$O/share512.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/one" -o st -- \
  python "$ROOT/tools/mw_one.py" 1 -1 8 > "$ROOT/$O/one.log" 2>&1
cd "$ROOT"
for d in share512 one; do f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/${d}_kernel_stats.csv && head -6 "$f"; done
tail -1 $O/share512.log | cut -c1-300
exit 0
