#!/usr/bin/env python
"""How does the oracle's native front end (oracle/frontend.cc, bench.py's cpu_baseline) scale over host threads, and where does the
time go when it does not?  Per thread count: frames/s, parallel efficiency, user and system CPU seconds of the process
(getrusage), voluntary / involuntary context switches, minor page faults.  TEST INFRASTRUCTURE: runs the oracle only.

    python tools/cpu_scaling.py [threads ...]        # default 1 8 32 64 128 256 (clipped to the host's thread count)
"""
import os, resource, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import bench
sys.path.insert(0, os.path.join(bench.ROOT, "tests"))
import _util
O, S = _util.oracle(), _util.synth()
V = _util._load("plslam_amd_vocab", os.path.join(bench.ROOT, "pl-slam_amd", "vocab.py"))
frames = S.make_frames(2, 64, 480, 640, unique=32)
voc = V.Vocabulary.synthetic(102, k=10, L=6, synth=S, idf=True)
L = O.lib()
L.plo_frontend_batch.restype = C.c_double
L.plo_frontend_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 8 + \
                                [C.c_int, C.c_int, C.c_int, C.c_void_p]
K, D = bench.TUM1_K, bench.TUM1_D
mx = np.zeros((480, 640), np.float32); my = np.zeros((480, 640), np.float32)
L.plo_undistort_maps(O._p(np.asarray(K, np.float32)), O._p(np.asarray(D, np.float32)), 640, 480, O._p(mx), O._p(my))
ww = np.ascontiguousarray(voc.word_weight(), np.float64)
frames = np.ascontiguousarray(frames)


def run(th, per):
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    dt = L.plo_frontend_batch(O._p(frames), len(frames), 480, 640, 1000, 8, 200, 0, O._p(mx), O._p(my), O._p(voc.node_desc),
                              O._p(voc.child_start), O._p(voc.child_count), O._p(voc.word_id), O._p(voc.weight), O._p(ww), voc.L, th, per, None)
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    return dt, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, r1.ru_nvcsw - r0.ru_nvcsw, r1.ru_nivcsw - r0.ru_nivcsw, r1.ru_minflt - r0.ru_minflt


hw = os.cpu_count() or 1
ths = [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 128, 256]
base = None
print("host threads %d, physical cores %d" % (hw, bench.physical_cores()))
print("%8s %10s %8s %10s %10s %10s %10s %12s" % ("threads", "frames/s", "eff", "user s", "sys s", "vol.csw", "invol.csw", "minor faults"))
for th in ths:
    if th > hw:
        continue
    per = 8
    dt, u, s, v, iv, mf = run(th, per)
    r = th * per / dt
    if base is None:
        base = r / th
    print("%8d %10.1f %8.3f %10.1f %10.1f %10d %10d %12d" % (th, r, r / (base * th), u, s, v, iv, mf), flush=True)
