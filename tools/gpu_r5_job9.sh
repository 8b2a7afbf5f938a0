#!/bin/bash
# Round 5: line tests, per-kernel durations alone at 1536 frames per launch (rocprofv3 kernel trace of bench.py --serial, ADV),
# one bench line with the extras (latency, 512-frame share, STD secondary).  gpurun_out/r5j9/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=$ROOT/gpurun_out/r5j9
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest ${TESTSEL:-tests/test_line.py tests/test_e2e_gpu.py tests/test_soak_gpu.py tests/test_ref_line.py} -m gpu -x -q --timeout 1200 2>&1 | grep -v "amdgpu.ids\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 | tee $O/tests.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o s -- python $ROOT/bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify > $O/serial.log 2>&1)
python - <<PY | tee $O/serial_stats.txt
import csv, glob
for f in glob.glob("$O/serial/**/*kernel_stats.csv", recursive=True):
    print("# kernels alone, 1536 frames per launch (bench.py --serial, LSD_REFINE_ADV), ms per launch")
    for r in csv.DictReader(open(f)):
        n=r["Name"].split("(")[0].replace("plh::","").replace("void ","")
        print("%-28s calls %4s  mean %9.3f ms  min %9.3f  max %9.3f" % (n[:28], r["Calls"], float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6, float(r["MaxNs"])/1e6))
PY
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_extras.json 2>$O/bench_extras.err
python - <<PY | tee $O/bench_extras.txt
import json
d=json.loads(open("$O/bench_extras.json").read().strip().split("\n")[-1])
print("value", d["value"], d["config"]["lsd_refine"]["level"], "verified", d["verified"]["exact"], d["verified"]["frames"], "frames; box", d.get("box"))
print("latency", d.get("latency_ms_single_frame"))
s=d.get("secondary",{})
print("secondary 1241x376", s.get("value"), "share512", (s.get("configs4_share_512") or {}).get("value"))
for k in ("refine_std","refine_adv"):
    if k in s: print(k, {kk:(vv.get("value") if isinstance(vv,dict) else vv) for kk,vv in s[k].items() if kk in ("resident_6144","configs4_share_512","latency_ms_single_frame")})
print("streaming", (d.get("streaming") or {}).get("value"))
PY
exit 0
