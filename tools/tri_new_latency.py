#!/usr/bin/env python3
"""Wall microseconds of LSDmatcher::SearchForTriangulationNew (src/LSDmatcher.cpp:780-832) through the reference's own CPU code (two
std::threads, one per direction) and through the drop-in class on the GPU, on the same posed KeyFrames (oracle/_ref/libadaptor_hip.so:
adx_local_mapping_line_triangulation_new times the two method calls; results asserted equal).
    python tools/tri_new_latency.py [reps] > profiles/r06c_search_for_triangulation_new_latency.txt          (GPU box)
    PLH_LATENCY_EMU=1 python tools/tri_new_latency.py 2                                                      (plumbing check, no GPU)"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    emu = os.environ.get("PLH_LATENCY_EMU") == "1"
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(ROOT, "tools", "gen_golden_ref.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    S, P = _util.synth(), _util.plslam()
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libadaptor_emu.so" if emu else "libadaptor_hip.so"))
    V, I = C.c_void_p, C.c_int
    R.adx_local_mapping_line_triangulation_new.argtypes = [V, V, V, V, I, V, V, V, V, I, V, V, V, I, V, V, V]
    p = lambda a: a.ctypes.data_as(V)
    print("# LSDmatcher::SearchForTriangulationNew(pKF1, pKF2, vMatchedPairs, isDouble = true): wall us per call, %d calls each (first = cold)" % reps)
    for n in (200, 600, 2000):
        x = G.lnew_inputs(S, 18, n, n, 0.1)
        ks = []
        for seg in (x["seg1"], x["seg2"]):
            k = np.zeros(len(seg), P.KL_DTYPE)
            k["startPointX"], k["startPointY"], k["endPointX"], k["endPointY"] = seg[:, 0], seg[:, 1], seg[:, 2], seg[:, 3]
            ks.append(k)
        Ts = []
        for pose in (x["pose1"], x["pose2"]):
            T = np.eye(4, dtype=np.float32); T[:3, :3] = pose[:9].reshape(3, 3); T[:3, 3] = pose[9:]
            Ts.append(np.ascontiguousarray(T))
        K4 = np.array([517.3, 516.5, 318.6, 255.3], np.float32)
        o_ref, o_hip, n_ref, us = np.zeros(n, np.int32), np.zeros(n, np.int32), C.c_int(0), (C.c_double * 2)()
        t = []
        for _ in range(reps):
            nm = R.adx_local_mapping_line_triangulation_new(p(ks[0]), p(x["d1"]), p(x["func1"]), p(x["ml1"]), n, p(ks[1]), p(x["d2"]), p(x["func2"]),
                                                            p(x["ml2"]), n, p(Ts[0]), p(Ts[1]), p(K4), 1, p(o_ref), p(o_hip), C.byref(n_ref))
            assert nm == n_ref.value and (o_ref == o_hip).all()
            R.adx_local_mapping_line_triangulation_new_us(us)
            t.append((us[0], us[1]))
        t = np.array(t)
        w = t[1:] if len(t) > 1 else t
        print("%5d lines per KeyFrame (%4d pairs):  reference CPU  cold %9.1f  warm median %9.1f   |   drop-in on the GPU  cold %9.1f  warm median %9.1f   (x %.1f)"
              % (n, nm, t[0, 0], np.median(w[:, 0]), t[0, 1], np.median(w[:, 1]), np.median(w[:, 0]) / np.median(w[:, 1])))
    print("(every call's result equals the reference's: asserted)")


if __name__ == "__main__":
    main()
