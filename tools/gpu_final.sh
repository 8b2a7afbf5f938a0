#!/bin/bash
# round-end evidence on one box: GPU suite, smoke; PMC traffic (calibrated per access pattern) and instruction counters on 1536 resident
# frames (1.2 GB of records: outside the Infinity Cache); SQ counters of k_lsd_grow at full residency; the default bench; rocprofv3 kernel
# statistics of both refine levels and of the configs[4] share; the residency curve; the N > 1 legs rehearsed on one GPU; the per-frame
# searches (batch rate, adaptor call latency); the 1024-frame soaks.  Everything lands in gpurun_out/final/; what is judged is copied to
# profiles/ by hand.   PART=tests|pmc|bench|stats|curve|track|soak (default: all but the soak)
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
PARTS=${PART:-tests pmc bench stats curve track}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if has tests; then
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 1200 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
fi
if has pmc; then
# the PMC profiles first: the bench line only carries PMC-derived fields (roofline.traffic, valu_issue) if they were collected on
# the build it runs, and it reads them from profiles/
bash tools/pmc_traffic.sh 1536 > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log; cp gpurun_out/pmc/traffic.json $O/hbm_traffic.json 2>/dev/null && cp $O/hbm_traffic.json profiles/hbm_traffic.json
bash tools/pmc_insts.sh 1536 > $O/pmc_insts.txt 2>&1; head -12 $O/pmc_insts.txt; cp gpurun_out/pmcinst/insts.json $O/insts.json 2>/dev/null && cp $O/insts.json profiles/sq_insts.json
bash tools/pmc_grow_detail.sh > $O/pmc_grow_detail.txt 2>&1; tail -12 $O/pmc_grow_detail.txt
bash tools/pmc_tracking.sh 1024 > $O/pmc_tracking.txt 2>&1; head -10 $O/pmc_tracking.txt; cp gpurun_out/pmctrack/insts.json $O/insts_tracking.json 2>/dev/null && cp $O/insts_tracking.json profiles/sq_insts_tracking.json
fi
if has bench; then
timeout 2400 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/bench.json'))
print('value', d['value'], d['config']['lsd_refine']['level'], 'box', (d.get('box') or {}).get('probe_ms'), 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['ms_per_launch'], d['roofline'].get('ms_per_launch_alone'), 'traffic', d['roofline'].get('traffic'), 'verified', d.get('verified',{}).get('exact'), d.get('verified',{}).get('frames'))
print('kernels', d['kernel_ms_per_launch'])
print('latency', {k: v for k, v in d.get('latency_ms_single_frame', {}).items() if k != 'note'})
s=d.get('secondary',{})
print('secondary', s.get('value'), 'share512', s.get('configs4_share_512',{}).get('value'), s.get('error'))
a=s.get('refine_std',{}) or s.get('refine_adv',{})
print('other level', a.get('level'), a.get('value'), a.get('vs_headline'), 'share', a.get('share_512',{}).get('value'), 'ver', (a.get('verified') or {}).get('exact'))
dd=s.get('distinct_frames',{})
print('distinct', dd.get('value'), dd.get('vs_headline'), (dd.get('verified') or {}).get('exact'), dd.get('error'))
t=s.get('tracking',{})
print('tracking', t.get('value'), t.get('unit'), (t.get('verified') or {}).get('exact'), t.get('error'))
c=d.get('cpu_baseline',{})
print('streaming', d.get('streaming',{}).get('value'), 'cpu', c.get('value'), c.get('cores'), 'extras s', d.get('extras_seconds'))
PY
# the N > 1 legs (RCCL gather per step, the strong-scaling configs[4] job in the same line) rehearsed with a one-rank communicator
timeout 900 python bench.py --force-dist --steps 6 --warmup 2 --no-cpu-baseline --no-extras --strong-leg 2>$O/bench_dist.err | tail -1 > $O/bench_force_dist.json
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/final/bench_force_dist.json'))
    s=(d.get('secondary') or {}).get('configs4_strong') or d.get('configs4_strong')
    print('force-dist value', d['value'], 'gather', d.get('gather'), 'configs4_strong', s if not isinstance(s, dict) else {k: s[k] for k in list(s)[:8]})
except Exception as e:
    print('force-dist: ', e)
PY
fi
if has stats; then
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
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/stats_track" -o st -- \
  python "$ROOT/tools/tracking_bench.py" --pairs 1024 --distinct 32 --steps 5 > "$ROOT/$O/tracking_stats.log" 2>&1
f=$(find "$ROOT/$O/stats_track" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$ROOT/$O/kernel_stats_tracking.csv" && head -8 "$f" | cut -c1-160
rm -rf "$ROOT/$O/stats_track"
cd "$ROOT"
python tools/kernel_resources.py > $O/kernel_resources.txt 2>&1
fi
if has curve; then
# frames per GPU against rate: the mid-residency regime the 8-GPU strong-scaling job of configs[4] lives in
{
echo "# resident frames per GPU -> frames/s (bench.py --no-extras --no-cpu-baseline --verify-frames 4; nsplit 1 below 2048 frames, else 1536-frame sub-batches)"
for shape in "376 1241 2000" "480 640 1000"; do
set -- $shape
for b in 256 512 1024 2048 3072 6144; do
ns=1; [ $b -ge 3072 ] && ns=$((b / 1536)); [ $b -eq 2048 ] && ns=2
timeout 400 python bench.py --batch $b --nsplit $ns --rows $1 --cols $2 --nfeatures $3 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --verify-frames 4 --verify-full none 2>/dev/null | tail -1 | \
  python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%dx%d  %5d frames (nsplit %d): %9.1f frames/s  %8.3f ms/step  verified %s  grow %s ms/launch' % ($2, $1, $b, $ns, d['value'], d['ms_per_step'], d.get('verified',{}).get('exact'), d['roofline'].get('ms_per_launch')))"
done
done
} 2>&1 | tee $O/residency_curve.txt
fi
if has track; then
timeout 900 python tools/tracking_bench.py --pairs 1024 --distinct 32 --steps 5 --json > $O/tracking_bench.json 2>$O/tracking_bench.err; tail -c 600 $O/tracking_bench.json
timeout 900 python tools/adaptor_latency.py 30 > $O/adaptor_call_latency.txt 2>$O/adaptor_latency.err; tail -25 $O/adaptor_call_latency.txt
fi
if has soak; then
PLSLAM_SOAK_FRAMES=1024 timeout 5000 python -m pytest tests/test_soak_gpu.py -m gpu -x -q -s --timeout 4000 2>&1 | grep -v amdgpu.ids | grep -E "soak|passed|failed|Error|error|waves" | tee $O/soak_1024.txt
fi
[ "${GROWMEM:-0}" = 1 ] && timeout 700 bash tools/pmc_grow_mem.sh 6144 > $O/pmc_grow_mem.txt 2>&1; grep k_lsd_grow $O/pmc_grow_mem.txt 2>/dev/null | cut -c1-200
exit 0
