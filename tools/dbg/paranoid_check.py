"""A -DPLH_MW_PARANOID build (the commit re-runs every posted transaction exactly and compares it with the post whenever the post passed
validation): how many validated posts differ from the sequential run?  usage: paranoid_check.py LIB.so [frames [rows cols]]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import _util, torch
import test_soak_gpu as T
P, S = _util.plslam(), _util.synth()
lib = os.path.join(ROOT, "pl-slam_amd", sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rows, cols = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (480, 640)
frames = T.soak_frames(S, rows, cols, N)
ref = [g[3] for g in T._gpu_lines(P, frames, 0, 1)]
L = P.load(lib)
out = (C.c_uint * 16)()
for run, (waves, chunk) in enumerate([(-1, N), (-1, N), (16, 256), (-1, 256)]):
    L.plh_debug_mw_paranoid(out, 1)
    bad = []
    for k in range(0, N, chunk):
        got = [g[3] for g in T._gpu_lines(P, frames[k:k + chunk], waves, 1, lib=lib)]
        bad += [k + i for i, (a, b) in enumerate(zip(got, ref[k:k + chunk])) if not (len(a) == len(b) and (a == b).all())]
    L.plh_debug_mw_paranoid(out, 0)
    o = list(out)
    print("%s, %dx%d, %d frames in launches of %d, waves %d: %d validated inline + %d validated general posts re-run exactly, %d differ from the exact run; "
          "frames whose segments differ from the product's: %d %s" % (sys.argv[1], cols, rows, N, chunk, waves, o[0], o[10], o[1], len(bad), bad[:8]), flush=True)
    if o[1]:
        print("   first offender: frame %d, sequence %d, seed (%d, %d), flags 0x%x, posted final region %d px (keep %d), exact %d px (keep %d)" %
              (o[2], o[3], o[4] & 0xffff, o[4] >> 16, o[5], o[6], o[8], o[7], o[9]), flush=True)
