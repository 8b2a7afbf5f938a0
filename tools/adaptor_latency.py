#!/usr/bin/env python3
"""Wall microseconds per call of the tracker's per-frame searches and of ComputeBoW through the reference's own CPU code
(oracle/_ref/libframe_ref.so: src/ORBmatcher.cc, src/LSDmatcher.cpp, DBoW2 compiled from /root/reference) and through the drop-in
adaptor classes on the GPU (oracle/_ref/libadaptor_hip.so: the same harness, the same objects), VERDICT r5 item 3.

Calls (real Frame / KeyFrame / MapPoint / MapLine objects, oracle/ref/ref_frame.cc; the harness times the matcher call alone and restores
the state it modifies in front of every repetition):
    ORBmatcher(0.9, true).SearchByProjection(Cur, Last, th, mono)     Tracking.cc:1321-1357  TrackWithMotionModel
    ORBmatcher(0.8).SearchByProjection(F, local MapPoints, th)        Tracking.cc:1792-1800  SearchLocalPoints (after isInFrustum)
    LSDmatcher().SearchByProjection(F, local MapLines, th)            Tracking.cc:1825-1849  SearchLocalLines
    ORBmatcher(0.7, true).SearchByBoW(pKF, F, matches)                Tracking.cc:1151-1159  TrackReferenceKeyFrame
    Frame::ComputeBoW()                                               Frame.cc:906-913
Run 0 of the adaptor is the cold call (the frame goes to the device: hip::FrameResidency miss), the median of the others the warm one.

    python tools/adaptor_latency.py [reps] > profiles/r06_adaptor_call_latency.txt        (GPU box)
    PLH_LATENCY_EMU=1 python tools/adaptor_latency.py 2                                    (plumbing check on the emulator, no GPU)"""
import ctypes as C
import importlib.util
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def timing(R):
    out = (C.c_double * 256)()
    n = R.ref_get_timing(out, 256)
    return np.array(out[:n])


def fmt(t):
    if len(t) == 0:
        return "      (not run)"
    warm = np.median(t[1:]) if len(t) > 1 else float("nan")
    return "cold %8.1f us   warm median %8.1f us   min %8.1f us" % (t[0], warm, t[1:].min() if len(t) > 1 else float("nan"))


def tile(d, k):
    return {key: np.ascontiguousarray(np.concatenate([v] * k)) for key, v in d.items()}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    emu = os.environ.get("PLH_LATENCY_EMU") == "1"
    G, S, P = _gen(), _util.synth(), _util.plslam()
    TF = G._test_module("test_frame_search")
    VM = _util._load("plslam_amd_vocab", os.path.join(ROOT, "pl-slam_amd", "vocab.py"))
    if not emu:
        P.load()   # the product library first (one HIP runtime in the process)
    libs = [("reference CPU (libframe_ref.so)", G.ref_frame_lib()),
            ("adaptor %s (%s)" % ("emulator" if emu else "GPU", "libadaptor_emu.so" if emu else "libadaptor_hip.so"),
             G.ref_frame_lib(os.path.join(ROOT, "oracle", "_ref", "libadaptor_emu.so" if emu else "libadaptor_hip.so")))]
    for _, R in libs:
        R.ref_set_timing.argtypes = [C.c_int]
        R.ref_get_timing.argtypes = [C.c_void_p, C.c_int]
        R.ref_set_timing(reps)
    print("# wall microseconds per call, %d repetitions behind the cold one; host: %d CPUs" % (reps, os.cpu_count() or 0))
    for n, nl, mapk in ((1000, 200, 1), (1000, 200, 6), (2000, 200, 3)):
        f2, gp, view, nlv, pts, lns, occ_p, occ_l = G.track_inputs(S, P, TF, 11, n, nl, False)
        flags, q = G.track_last_inputs(S, P, TF, 11, n, nl, False)
        big, bigl = tile(pts, mapk), tile(lns, mapk)
        print("\n== frame of %d keypoints / %d lines; local map of %d points / %d lines" % (n, len(f2["keylines"]), len(big["min_dist"]), len(bigl["min_dist"])))
        res = {}
        for name, R in libs:
            c, a, o = G.reference_track_last(R, P, TF, f2, gp, view, nlv, pts, flags, q, occ_p, 15.0)
            t_last = timing(R)
            n_, nl_, g = len(f2["kps"]), len(f2["keylines"]), P._gp_array(gp)
            p = G.p
            h = R.ref_frame_create(p(f2["kps"]), n_, p(f2["keylines"]), p(f2["linefn"]), nl_, p(g))
            op, ap = occ_p.copy(), np.zeros(max(n_, 1), np.int32)
            cp = R.ref_track_local_points(h, p(f2["desc"]), p(view), nlv, p(TF.SCALE), p(op), len(big["min_dist"]), p(big["pos"]), p(big["normal"]),
                                          p(big["min_dist"]), p(big["max_dist"]), p(big["desc"]), p(big["hasobs"]), 3.0, p(ap))
            t_pts = timing(R)
            ol, al = occ_l.copy(), np.zeros(max(nl_, 1), np.int32)
            cl = R.ref_track_local_lines(h, p(f2["ldesc"]), p(view), p(TF.SCALE), nlv, p(ol), len(bigl["min_dist"]), p(bigl["pos"]), p(bigl["normal"]),
                                         p(bigl["min_dist"]), p(bigl["max_dist"]), p(bigl["desc"]), p(bigl["hasobs"]), 3.0, p(al))
            t_lns = timing(R)
            R.ref_frame_destroy(h)
            res[name] = (c, a.copy(), cp, ap.copy(), cl, al.copy())
            print("  %-42s SearchByProjection(Cur, Last)      %s   (%d matches)" % (name, fmt(t_last), c))
            print("  %-42s SearchByProjection(F, MapPoints)   %s   (%d matches)" % (name, fmt(t_pts), cp))
            print("  %-42s SearchByProjection(F, MapLines)    %s   (%d matches)" % (name, fmt(t_lns), cl))
        (c0, a0, p0, ap0, l0, al0), (c1, a1, p1, ap1, l1, al1) = res[libs[0][0]], res[libs[1][0]]
        assert c0 == c1 and (a0 == a1).all() and p0 == p1 and (ap0 == ap1).all() and l0 == l1 and (al0 == al1).all(), "adaptor and reference disagree"
    for k, Lv, n in ((10, 4, 1000), (10, 6, 2000)):
        voc, kf, fr = G.bowtrack_inputs(S, P, VM, 650 + Lv, k, Lv, n)
        vocc, desc = G.computebow_inputs(S, VM, 750 + Lv, k, Lv, 0.01, True, n)
        print("\n== vocabulary k = %d, L = %d (%d nodes), %d features" % (k, Lv, voc.n_nodes, n))
        out = []
        for name, R in libs:
            c, m = G.reference_bowtrack(R, voc, kf, fr, 0.7, 1, tempfile.gettempdir())
            t_bow = timing(R)
            bw, bv, fn, ff, eq, adp = G.reference_computebow(R, vocc, desc, False, 0, 0, tempfile.gettempdir())
            # (ref_compute_bow times Frame::ComputeBoW; the KeyFrame / second-vocabulary calls behind it are untimed: read now)
            t_cb = timing(R)
            out.append((c, m.copy(), bw.copy(), bv.copy()))
            print("  %-42s SearchByBoW(pKF, F)                %s   (%d matches)" % (name, fmt(t_bow), c))
            print("  %-42s Frame::ComputeBoW                  %s   (%d words)" % (name, fmt(t_cb), len(bw)))
        assert out[0][0] == out[1][0] and (out[0][1] == out[1][1]).all() and (out[0][2] == out[1][2]).all() and (out[0][3] == out[1][3]).all()
    for _, R in libs:
        R.ref_set_timing(0)
    print("\n(every adaptor result above equals the reference's: asserted)")


if __name__ == "__main__":
    main()
