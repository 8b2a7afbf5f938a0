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
known = {"k_stream16": bytes_, "k_stream4": bytes_, "k_stream1": bytes_, "k_gather4": frames * words * 4}
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
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "calibration.json"), "w"), indent=1)
PY
