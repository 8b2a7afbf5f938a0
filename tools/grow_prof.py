#!/usr/bin/env python3
"""Phase breakdown of k_lsd_grow (debug build with -DPLH_GROW_PROF, see lsd_grow.hip).

    hipcc ... -DPLH_GROW_PROF=2 -o pl-slam_amd/libplslam_hip_prof.so pl-slam_amd/csrc/*.hip
    PLSLAM_HIP_LIB=pl-slam_amd/libplslam_hip_prof.so python tools/grow_prof.py [--batch 1024]

Prints the s_memtime cycle totals of every phase summed over all waves (one wave per frame), per frame.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

NAMES = ["total", "#region2rect pixels", "region_grow", "grow:load wait", "grow:resolve", "region2rect", "refine", "#reduce_radius steps",
         "#steps", "#accepted", "#cands", "#grow calls", "#passes", "#mispredicts", "#region2rect after grow", "#refine"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    P, S = _util.plslam(), _util.synth()
    lib = P.load()
    prof = hasattr(lib, "plh_debug_grow_prof")   # only in a PLH_GROW_PROF build (set PLSLAM_HIP_LIB); else timing only
    B = a.batch
    frames = S.make_frames(2, B, 480, 640, unique=32)
    d = torch.from_numpy(frames).cuda()
    K = [517.306408, 516.469215, 318.643040, 255.313989]
    D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
    le = P.LINEextractor(1, 1.2, 200, 0.0, rows=480, cols=640, max_batch=B, device=0, K=K, D=D)
    cap = le.capacity
    kl = torch.zeros((B, cap, 17), dtype=torch.float32, device="cuda")
    ld = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    fn = torch.zeros((B, cap, 3), dtype=torch.float64, device="cuda")
    nl = torch.zeros((B,), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    out = (C.c_ulonglong * 40)()
    le.extract_batch_dev(d, B, 480 * 640, kl, ld, fn, nl, s)
    torch.cuda.synchronize()
    if prof:
        lib.plh_debug_grow_prof(out, 1)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        le.extract_batch_dev(d, B, 480 * 640, kl, ld, fn, nl, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print("line extract %.2f ms / batch of %d = %.0f frames/s" % (dt * 1e3, B, B / dt))
    if not prof:
        return
    lib.plh_debug_grow_prof(out, 0)
    n = B * a.reps
    res = {NAMES[i] if NAMES[i] != "-" else "c%d" % i: out[i] / n for i in range(16)}
    for k, v in res.items():
        print("  %-16s %14.1f per frame" % (k, v))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
