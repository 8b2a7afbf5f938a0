#!/usr/bin/env python3
"""Schedule of k_lsd_grow_mw on ONE frame from its event trace (debug build with -DPLH_GROW_PROF=3):

    PLSLAM_HIP_LIB=pl-slam_amd/libplslam_hip_trace.so python tools/mw_trace.py [--waves 8]

Per wavefront: time running transactions / posting / draining / re-running / with nothing to do (and why) / in between."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util, torch
ap = argparse.ArgumentParser()
ap.add_argument("--waves", type=int, default=8)
ap.add_argument("--dump", default="")
a = ap.parse_args()
P, S = _util.plslam(), _util.synth()
lib = P.load()
K = [517.306408, 516.469215, 318.643040, 255.313989]; D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
frames = S.make_frames(2, 2, 480, 640)
le = P.LINEextractor(1, 1.2, 200, 0.0, rows=480, cols=640, max_batch=1, K=K, D=D)
le.set_grow_waves(a.waves)
le(frames[0]); le(frames[1])
cap = 1 << 20
buf = np.zeros((cap, 2), np.uint64)
lib.plh_debug_mw_trace.argtypes = [C.c_void_p, C.c_uint, C.c_int]
lib.plh_debug_mw_trace(None, 0, 1)
kl, _, _ = le(frames[0])
n = lib.plh_debug_mw_trace(buf.ctypes.data, cap, 1)
ev = buf[:n]
kind = (ev[:, 0] & 0xff).astype(int); wv = ((ev[:, 0] >> 8) & 0xff).astype(int); arg = (ev[:, 0] >> 16).astype(np.int64)
t = ev[:, 1].astype(np.int64)
t0, t1 = t.min(), t.max()
print("%d events, %d lines, span %.3f ms (clock ticks: %d)" % (n, len(kl), (t1 - t0) / 100e3, t1 - t0))   # s_memtime: 100 MHz
if a.dump:
    np.save(a.dump, ev)
tick = 1e-2   # us per tick at 100 MHz
names = {1: "popped", 2: "run", 3: "run end", 4: "post", 5: "drain", 6: "drain end", 7: "idle", 8: "rerun", 9: "rerun end"}
tot = {}
for w in sorted(set(wv)):
    m = wv == w
    k, tt, ar = kind[m], t[m], arg[m]
    o = np.argsort(tt, kind="stable")
    k, tt, ar = k[o], tt[o], ar[o]
    acc = {"run": 0, "pop->run": 0, "run->post": 0, "post->next": 0, "drain": 0, "idle": 0, "other": 0}
    why = {}
    for i in range(len(k) - 1):
        d = tt[i + 1] - tt[i]
        if k[i] == 2: acc["run"] += d
        elif k[i] == 1: acc["pop->run"] += d
        elif k[i] == 3: acc["run->post"] += d
        elif k[i] == 4: acc["post->next"] += d
        elif k[i] in (5, 8, 9): acc["drain"] += d
        elif k[i] == 7:
            acc["idle"] += d
            why[int(ar[i])] = why.get(int(ar[i]), 0) + d
        else: acc["other"] += d
    print("wave %2d: %5d txns | us: " % (w, int((k == 1).sum())) + "  ".join("%s %.0f" % (kk, v * tick) for kk, v in acc.items()) +
          " | idle why (1 seeds queued, 2 arena, 4 lag, 8 scan done): " + str({kk: round(v * tick) for kk, v in why.items()}))
    for kk, v in acc.items(): tot[kk] = tot.get(kk, 0) + v
print("all: " + "  ".join("%s %.0f" % (kk, v * tick) for kk, v in tot.items()))
# drain sessions
dm = kind == 5
print("drain sessions: %d, re-runs: %d" % (dm.sum(), (kind == 8).sum()))
# run durations by size
rb = np.where(kind == 2)[0]
