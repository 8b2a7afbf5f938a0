#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh) into bytes per frame per kernel.

Units and corrections (MI355X_MICROARCH.md, "HBM"): both counters are reported in KiB of memory-side (fabric) traffic of
the L2, Infinity-Cache hits included.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at
64 bytes, i.e. reports half the bytes -> doubled here; narrower accesses and WRITE_SIZE are uncalibrated (taken as is)."""
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


def main():
    root, batch = sys.argv[1], int(sys.argv[2])
    fetch, fc = fold(os.path.join(root, "fetch"), "FETCH_SIZE")
    write, wc = fold(os.path.join(root, "write"), "WRITE_SIZE")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    # the library build the counters were collected on: bench.py reports them only for the same build
    out = {"batch": batch, "unit": "bytes per frame per launch", "fetch_correction": 2.0, "build": g._lib_id(g.LIB), "kernels": {}}
    steps = max(fc.get("k_lsd_grow", 0), wc.get("k_lsd_grow", 0), 1)   # one launch per front-end step
    for k in sorted(set(fetch) | set(write)):
        n = max(fc.get(k, 0), wc.get(k, 0), 1)
        launches_f, launches_w = max(fc.get(k, 0), 1), max(wc.get(k, 0), 1)
        fb = fetch.get(k, 0.0) * 1024 * 2.0 / launches_f / batch
        wb = write.get(k, 0.0) * 1024 / launches_w / batch
        out["kernels"][k] = {"fetch": round(fb), "write": round(wb), "total": round(fb + wb), "launches": n,
                             "launches_per_step": max(1, round(n / steps))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
