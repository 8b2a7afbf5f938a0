#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh) into bytes per frame per kernel.

Units and corrections (MI355X_MICROARCH.md, "HBM"): both counters are reported in KiB of memory-side (fabric) traffic of
the L2, Infinity-Cache hits included.  On gfx950 FETCH_SIZE tallies a 128-byte request at 64 bytes.  Calibrated on this code
base's own access patterns (tools/ubench/pmc_patterns.hip, profiles/r05_pmc_calibration.json, buffers far beyond the Infinity
Cache): consecutive-lane streams of 16-byte, 4-byte AND 1-byte loads all coalesce into 128-byte requests -> x 2.0; random 4-byte
gathers issue one 64-byte request per load and are counted at face value -> x 1.0.  A kernel gets the correction of the pattern
that dominates its reads (GATHER_KERNELS below); `fetch_x1` and `fetch_x2` give both readings.  WRITE_SIZE calibrates exactly on
k_remap_u8's output plane (as is)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def fold(dirname, counter):
    tot, calls = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("plh::", "").replace("void ", "").split("<")[0]
            tot[k] += float(r["Counter_Value"])
            calls[k] += 1
    return tot, calls


# kernels whose memory-side reads are dominated by random 4-byte gathers (the level-line records of region growing)
GATHER_KERNELS = ("k_lsd_grow", "k_lsd_grow_lone", "k_lsd_grow_mw", "k_lsd_grow_mw16")


def main():
    root, batch = sys.argv[1], int(sys.argv[2])
    fetch, fc = fold(os.path.join(root, "fetch"), "FETCH_SIZE")
    write, wc = fold(os.path.join(root, "write"), "WRITE_SIZE")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    # the library build the counters were collected on: bench.py reports them only for the same build
    out = {"batch": batch, "unit": "bytes per frame per launch",
           "fetch_correction": {"streams (16-, 4-, 1-byte consecutive lanes)": 2.0, "random 4-byte gathers": 1.0,
                                "calibration": "profiles/r05_pmc_calibration.json (tools/pmc_calibrate.sh)", "gather_kernels": list(GATHER_KERNELS)},
           "build": g._lib_id(g.LIB), "kernels": {}}
    steps = max(fc.get("k_lsd_grow", 0), wc.get("k_lsd_grow", 0), 1)   # one launch per front-end step
    for k in sorted(set(fetch) | set(write)):
        n = max(fc.get(k, 0), wc.get(k, 0), 1)
        launches_f, launches_w = max(fc.get(k, 0), 1), max(wc.get(k, 0), 1)
        corr = 1.0 if k in GATHER_KERNELS else 2.0
        f1 = fetch.get(k, 0.0) * 1024 / launches_f / batch
        fb = f1 * corr
        wb = write.get(k, 0.0) * 1024 / launches_w / batch
        out["kernels"][k] = {"fetch": round(fb), "write": round(wb), "total": round(fb + wb), "fetch_correction": corr,
                             "fetch_x1": round(f1), "fetch_x2": round(2 * f1), "launches": n,
                             "launches_per_step": max(1, round(n / steps))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
