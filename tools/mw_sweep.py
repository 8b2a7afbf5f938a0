#!/usr/bin/env python3
"""k_lsd_grow_mw (several wavefronts per frame) against k_lsd_grow (one): region-growing time per launch (HIP events of the
library's own stage profile) and whole line-extractor time, for a list of batch sizes and wavefront counts; every
configuration's KeyLines / LBD bytes are compared with the one-wavefront result of the same batch (development aid).

    python tools/mw_sweep.py [--batches 1,8,64,512] [--waves 0,2,4,8,16] [--rows 480 --cols 640] [--reps 5]
"""
import argparse, ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util, torch
P, S = _util.plslam(), _util.synth()
ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="1,8,64,512")
ap.add_argument("--waves", default="0,2,4,8,16")
ap.add_argument("--rows", type=int, default=480)
ap.add_argument("--cols", type=int, default=640)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
K = [517.306408, 516.469215, 318.643040, 255.313989]; D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
dev = torch.device("cuda", 0)
for B in [int(x) for x in a.batches.split(",")]:
    frames = S.make_frames(2, B, a.rows, a.cols, unique=min(B, 32))
    d_img = torch.from_numpy(frames).to(dev)
    ex = P.LINEextractor(1, 1.2, 200, 0.0, rows=a.rows, cols=a.cols, max_batch=B, K=K, D=D)
    cap = ex.capacity
    bufs = [torch.zeros((B, cap, 17), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
            torch.zeros((B, cap, 3), dtype=torch.float64, device=dev), torch.zeros((B,), dtype=torch.int32, device=dev)]
    st = torch.cuda.current_stream().cuda_stream
    ref = None
    for W in [int(x) for x in a.waves.split(",")]:
        if W * 64 > 1024:
            continue
        ex.set_grow_waves(W)
        for t in bufs: t.zero_()
        ex.extract_batch_dev(d_img, B, a.rows * a.cols, *bufs, st)   # warm-up (allocates the workspace)
        torch.cuda.synchronize()
        ex.lib.plh_line_set_profiling(ex.h, 1)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            ex.extract_batch_dev(d_img, B, a.rows * a.cols, *bufs, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        ms, n = C.c_double(0), C.c_int(0)
        ex.lib.plh_line_kernel_ms(ex.h, 1, C.byref(ms), C.byref(n))
        ex.lib.plh_line_set_profiling(ex.h, 0)
        got = [t.cpu().numpy().copy() for t in bufs]
        nl = got[3]
        for b in range(B):   # rows beyond n are stale
            got[0][b, nl[b]:] = 0; got[1][b, nl[b]:] = 0; got[2][b, nl[b]:] = 0
        if ref is None:
            ref = got
        same = all((x.view(np.uint8) == y.view(np.uint8)).all() for x, y in zip(got, ref))
        print("batch %5d  waves %2d : grow %8.3f ms  extract %8.3f ms  %8.0f frames/s  flags %d  %s" %
              (B, W, ms.value / max(n.value, 1), dt * 1e3, B / dt, ex.status(), "== waves %s" % a.waves.split(",")[0] if same else "DIFFERS"),
              flush=True)
    ex.close()
