#!/bin/bash
# Per-access-pattern correction of FETCH_SIZE on this GPU: known bytes / counter for 16-byte, dword and byte streams and for
# random 4-byte gathers (tools/ubench/pmc_patterns.hip) -> gpurun_out/pmc_cal/calibration.json (copied to profiles/ by hand).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=$ROOT/gpurun_out/pmc_cal
mkdir -p $O
export TMPDIR=/tmp
MIB=${1:-2048}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_patterns tools/ubench/pmc_patterns.hip || exit 1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o cal -- /tmp/pmc_patterns $MIB > $O/$c.log 2>&1
done
cd $ROOT
python - "$O" "$MIB" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
root, mib = sys.argv[1], int(sys.argv[2])
bytes_ = mib << 20
words = 196608
frames = bytes_ // 4 // words
# streams: every byte of the buffer is needed once.  Random 4-byte gathers: each of the frames x 3 x words lane-loads asks the memory side
# for the 64-byte sector it falls in (the window, 0.8 MB per wavefront, is read ~3 times over in sectors that hold 16 words), so the
# bytes the fabric has to move are requests x 64 -- THAT is what FETCH_SIZE is compared with (round 4's file divided the window bytes
# by the counter and printed 0.021, which is the sector re-use, not a correction)
requests = frames * 3 * words
known = {"k_stream16": bytes_, "k_stream4": bytes_, "k_stream1": bytes_, "k_gather4": requests * 64}
extra = {"k_gather4": {"lane_loads": requests, "window_bytes_total": frames * words * 4,
                       "known_bytes_is": "lane-loads x 64-byte sectors (one request each; windows far beyond the caches)"}}
out = {"buffer_MiB": mib, "what": "known bytes / FETCH_SIZE (KiB x 1024) per access pattern; buffers far beyond the 256 MiB Infinity Cache, read once",
       "patterns": {}}
for f in glob.glob(os.path.join(root, "FETCH_SIZE", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "FETCH_SIZE":
            per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024)
    for k, v in per.items():
        if k in known:
            # k_stream16 runs as flush + measured in every round: all its launches read `bytes_`
            vals = sorted(v)
            med = vals[len(vals) // 2]
            out["patterns"][k] = {"known_bytes": known[k], "fetch_size_bytes_median": med, "launches": len(v),
                                  "correction": round(known[k] / med, 3) if med else None}
            out["patterns"][k].update(extra.get(k, {}))
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "calibration.json"), "w"), indent=1)
PY
