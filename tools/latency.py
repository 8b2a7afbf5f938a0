#!/usr/bin/env python3
"""Single-frame latency of the host-buffer entry points (development aid; bench.py reports the same as latency_ms_single_frame)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util, torch
P, S = _util.plslam(), _util.synth()
K = [517.306408, 516.469215, 318.643040, 255.313989]; D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
frames = S.make_frames(2, 8, 480, 640)
orb = P.ORBextractor(1000, 1.2, 8, 20, 7, rows=480, cols=640, max_batch=1)
line = P.LINEextractor(1, 1.2, 200, 0.0, rows=480, cols=640, max_batch=1, K=K, D=D)
line.lib.plh_line_set_profiling.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_int]
def timed(fn, n=10):
    fn(frames[0]); t0 = time.perf_counter()
    for i in range(n): fn(frames[(i + 1) % 8])
    return (time.perf_counter() - t0) / n * 1e3
print("orb %.3f ms  line %.3f ms" % (timed(orb), timed(line)))
import ctypes as C
line.lib.plh_line_set_profiling(line.h, 1)
for i in range(4): line(frames[i])
line.lib.plh_line_kernel_ms.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
for k, name in enumerate(["prep", "k_lsd_grow", "k_keylines", "LBD"]):
    ms, n = C.c_double(0), C.c_int(0)
    line.lib.plh_line_kernel_ms(line.h, k, C.byref(ms), C.byref(n))
    print("  %-12s %.3f ms per frame" % (name, ms.value / max(n.value, 1)))
