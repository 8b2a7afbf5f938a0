#!/bin/bash
# round-end evidence on one box: GPU suite, smoke, default bench; rocprofv3 kernel statistics of the headline, of LSD_REFINE_ADV and of
# the configs[4] share; PMC traffic (calibrated per access pattern) and instruction counters on 1536 resident frames (1.2 GB of
# records: outside the Infinity Cache); SQ counters of k_lsd_grow at full residency.  Everything lands in gpurun_out/final/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=gpurun_out/final
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
# the PMC profiles first: the bench line only carries PMC-derived fields (roofline.traffic, valu_issue) if they were collected on
# the build it runs, and it reads them from profiles/
bash tools/pmc_traffic.sh 1536 > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log; cp gpurun_out/pmc/traffic.json $O/hbm_traffic.json 2>/dev/null && cp $O/hbm_traffic.json profiles/hbm_traffic.json
bash tools/pmc_insts.sh 1536 > $O/pmc_insts.txt 2>&1; head -12 $O/pmc_insts.txt; cp gpurun_out/pmcinst/insts.json $O/insts.json 2>/dev/null && cp $O/insts.json profiles/r05_insts.json
bash tools/pmc_grow_detail.sh > $O/pmc_grow_detail.txt 2>&1; tail -12 $O/pmc_grow_detail.txt
timeout 1500 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/bench.json'))
print('value', d['value'], d['config']['lsd_refine']['level'], 'box', (d.get('box') or {}).get('probe_ms'), 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['ms_per_launch'], d['roofline'].get('ms_per_launch_alone'), 'verified', d.get('verified',{}).get('exact'), d.get('verified',{}).get('frames'))
print('kernels', d['kernel_ms_per_launch'])
print('latency', {k: v for k, v in d.get('latency_ms_single_frame', {}).items() if k != 'note'})
s=d.get('secondary',{})
print('secondary', s.get('value'), 'share512', s.get('configs4_share_512',{}).get('value'), s.get('error'))
a=s.get('refine_std',{}) or s.get('refine_adv',{})
print('other level', a.get('level'), a.get('value'), a.get('vs_headline'), 'share', a.get('share_512',{}).get('value'), 'lat', {k: v for k, v in (a.get('latency_ms_single_frame') or {}).items() if k != 'note'}, 'ver', (a.get('verified') or {}).get('exact'))
c=d.get('cpu_baseline',{})
print('streaming', d.get('streaming',{}).get('value'), 'cpu', c.get('value'), c.get('cores'), c.get('legs'))
PY
cd /tmp
for mode in std adv; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/stats_$mode" -o st -- \
  python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-verify --refine $mode > "$ROOT/$O/bench_stats_$mode.log" 2>&1
f=$(find "$ROOT/$O/stats_$mode" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$ROOT/$O/kernel_stats_$mode.csv" && head -6 "$f" | cut -c1-160
rm -rf "$ROOT/$O/stats_$mode"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/stats_share" -o st -- \
  python "$ROOT/bench.py" --batch 512 --nsplit 1 --rows 376 --cols 1241 --nfeatures 2000 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify > "$ROOT/$O/bench_stats_share.log" 2>&1
f=$(find "$ROOT/$O/stats_share" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$ROOT/$O/kernel_stats_share512.csv" && head -5 "$f" | cut -c1-160
rm -rf "$ROOT/$O/stats_share"
cd "$ROOT"
[ "${GROWMEM:-0}" = 1 ] && timeout 700 bash tools/pmc_grow_mem.sh 6144 > $O/pmc_grow_mem.txt 2>&1; grep k_lsd_grow $O/pmc_grow_mem.txt | cut -c1-200
exit 0
