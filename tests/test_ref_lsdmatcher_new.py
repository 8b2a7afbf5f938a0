"""LSDmatcher's "New" line matcher for triangulation, pinned against the REFERENCE's own code:

    FrameBFMatchNew(ldesc1, ldesc2, LineMatches, kls1, kls2, kls2func, F, TH)      src/LSDmatcher.cpp:488-548
    mutualOverlap(collinear_points)                                                :550-625
    SearchForTriangulationNew(pKF1, pKF2, vMatchedPairs, isDouble)                 :780-832   (call site LocalMapping.cc:960, commented out)

oracle/_ref/liblsdmatcher_ref.so is src/LSDmatcher.cpp compiled from the source where it lies (oracle/ref/build_ref.sh) against the
stand-ins of oracle/ref/slam_stub.h / oracle/ref/stub; oracle/ref/ref_lsdmatcher.cc fills two stand-in KeyFrames (descriptors, KeyLines,
line equations, poses, K, MapLines) from flat arrays and calls the methods.  What is pinned: the nearest-neighbour choice, the epipolar
carry-over of the end points, the |w| > 1e-12 test, the overlap score and its 0.8 gate, the TH / ratio tests, the mutual check, the MapLine
gate -- bit for bit.  What is not: cv::Mat's float algebra itself (the stand-in's products, cross product, `/=` and norm; OpenCV is not in
the image) and, for the same reason, the values of the fundamental matrices -- they come out of the reference's ComputeF12 through the
stand-in's 3 x 3 inverse and are INPUTS of the oracle and of the library (stored in the golden file).
tools/gen_golden_ref.py lnew committed the reference outputs as tests/golden/ref_lsdmatcher_new.npz; the oracle (CPU), the emulator build
of the kernels (CPU) and the GPU kernels (`-m gpu`) must reproduce them; in the build container the reference is also run live."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLDEN = os.path.join(_util.ROOT, "tests", "golden", "ref_lsdmatcher_new.npz")
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "liblsdmatcher_ref.so")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _oracle_bf(O, x, F, th, ratio):
    n1 = len(x["d1"])
    m = np.full(max(n1, 1), -1, np.int32)
    F = np.ascontiguousarray(F, np.float32)
    O.lib().plo_line_bfmatch_new(O._p(x["d1"]), n1, O._p(x["d2"]), len(x["d2"]), O._p(x["seg1"]), O._p(x["seg2"]), O._p(x["func2"]), O._p(F),
                                 C.c_float(th), C.c_float(ratio), O._p(m))
    return m[:n1]


def _oracle_tri(O, x, F21, F12, ratio, dbl):
    n1 = len(x["d1"])
    m = np.full(max(n1, 1), -1, np.int32)
    F21, F12 = np.ascontiguousarray(F21, np.float32), np.ascontiguousarray(F12, np.float32)
    c = O.lib().plo_line_search_for_triangulation_new(O._p(x["d1"]), n1, O._p(x["d2"]), len(x["d2"]), O._p(x["seg1"]), O._p(x["seg2"]),
                                                      O._p(x["func1"]), O._p(x["func2"]), O._p(F21), O._p(F12), O._p(x["ml1"]), O._p(x["ml2"]),
                                                      C.c_float(50.0), C.c_float(ratio), dbl, O._p(m))
    return c, m[:n1]


def test_golden_file_present():
    assert os.path.exists(GOLDEN)
    g = np.load(GOLDEN)
    assert int(g["tri_18_1_n"]) > 100 and int((g["bf_11_0"] >= 0).sum()) > 50      # the gates leave something and take something away
    assert int((g["bf_11_2"] >= 0).sum()) == 0                                      # F = 0: every intersection has w = 0


def test_oracle_reproduces_reference_lsdmatcher_new(oracle, synth):
    G = _gen()
    g = np.load(GOLDEN)
    for seed, n1, n2, flip, ratio in G.LNEW_CASES:
        x = G.lnew_inputs(synth, seed, n1, n2, flip)
        F21, F12 = g["F21_%d" % seed], g["F12_%d" % seed]
        for dbl in (0, 1):
            c, m = _oracle_tri(oracle, x, F21, F12, ratio, dbl)
            assert c == int(g["tri_%d_%d_n" % (seed, dbl)]) and (m == g["tri_%d_%d_m" % (seed, dbl)]).all(), "SearchForTriangulationNew %d %d" % (seed, dbl)
        for k, (th, sc) in enumerate(G.LNEW_BF):
            assert (_oracle_bf(oracle, x, F21 * np.float32(sc), th, ratio) == g["bf_%d_%d" % (seed, k)]).all(), "FrameBFMatchNew %d %d" % (seed, k)


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_lsdmatcher_new_live(oracle, synth):
    G = _gen()
    R = G.ref_lsdmatcher_lib()
    for seed, n1, n2, flip, ratio in [(31, 210, 190, 0.08, 0.8), (32, 45, 300, 0.18, 0.7), (33, 640, 3, 0.1, 0.9), (34, 1200, 1100, 0.15, 0.75)]:
        x = G.lnew_inputs(synth, seed, n1, n2, flip)
        for dbl in (0, 1):
            c, m, F21, F12 = G.reference_ltri_new(R, x, ratio, dbl)
            rc, rm = _oracle_tri(oracle, x, F21, F12, ratio, dbl)
            assert c == rc and (m == rm).all(), "live SearchForTriangulationNew %d %d" % (seed, dbl)
        for th, sc in ((50.0, 1.0), (80.0, 0.37), (25.0, 1.0)):
            F = F21 * np.float32(sc)
            assert (G.reference_lbf_new(R, x, F, th, ratio) == _oracle_bf(oracle, x, F, th, ratio)).all(), "live FrameBFMatchNew %d" % seed


def _check_device(P, synth, lib):
    G = _gen()
    g = np.load(GOLDEN)
    for seed, n1, n2, flip, ratio in G.LNEW_CASES:
        x = G.lnew_inputs(synth, seed, n1, n2, flip)
        F21, F12 = g["F21_%d" % seed], g["F12_%d" % seed]
        m = P.LSDmatcher(ratio, True, lib=lib)
        for dbl in (0, 1):
            c, got = m.SearchForTriangulationNew(x["d1"], x["d2"], x["seg1"], x["seg2"], x["func1"], x["func2"], F21, F12, x["ml1"], x["ml2"],
                                                 isDouble=bool(dbl))
            assert c == int(g["tri_%d_%d_n" % (seed, dbl)]) and (got == g["tri_%d_%d_m" % (seed, dbl)]).all(), "SearchForTriangulationNew %d %d" % (seed, dbl)
        for k, (th, sc) in enumerate(G.LNEW_BF):
            got = m.FrameBFMatchNew(x["d1"], x["d2"], x["seg1"], x["seg2"], x["func2"], F21 * np.float32(sc), TH=th)
            assert (got == g["bf_%d_%d" % (seed, k)]).all(), "FrameBFMatchNew %d %d" % (seed, k)
    # empty sets: nothing to match, nothing touched
    m = P.LSDmatcher(0.7, True, lib=lib)
    e8, e4, e3 = np.zeros((0, 32), np.uint8), np.zeros((0, 4), np.float32), np.zeros((0, 3))
    x = G.lnew_inputs(synth, 12, 5, 5, 0.1)
    c, got = m.SearchForTriangulationNew(x["d1"], e8, x["seg1"], e4, x["func1"], e3, np.eye(3), np.eye(3), x["ml1"], np.zeros(0, np.uint8))
    assert c == 0 and (got == -1).all() and len(got) == 5
    c, got = m.SearchForTriangulationNew(e8, x["d2"], e4, x["seg2"], e3, x["func2"], np.eye(3), np.eye(3), np.zeros(0, np.uint8), x["ml2"])
    assert c == 0 and len(got) == 0


def _check_errors(P, synth, lib):
    """Error behaviour of the two entry points: negative counts and missing arrays are refused (PLH_ERR_INVALID = a non-zero status, nothing
    written), empty sets are not an error."""
    G = _gen()
    L = P.load(lib)
    x = G.lnew_inputs(synth, 12, 6, 7, 0.1)
    p, F = P._p, np.ascontiguousarray(np.eye(3, dtype=np.float32).reshape(9))
    m, c = np.full(6, 5, np.int32), C.c_int(-3)
    assert L.plh_line_frame_bfmatch_new(p(x["d1"]), -1, p(x["d2"]), 7, p(x["seg1"]), p(x["seg2"]), p(x["func2"]), p(F), 50.0, 0.7, p(m), 0) != 0
    assert L.plh_line_frame_bfmatch_new(p(x["d1"]), 6, p(x["d2"]), 7, None, p(x["seg2"]), p(x["func2"]), p(F), 50.0, 0.7, p(m), 0) != 0
    assert L.plh_line_frame_bfmatch_new(p(x["d1"]), 6, p(x["d2"]), 7, p(x["seg1"]), p(x["seg2"]), None, p(F), 50.0, 0.7, p(m), 0) != 0
    assert L.plh_line_frame_bfmatch_new(p(x["d1"]), 6, p(x["d2"]), 7, p(x["seg1"]), p(x["seg2"]), p(x["func2"]), None, 50.0, 0.7, p(m), 0) != 0
    assert (m == 5).all()
    assert L.plh_line_frame_bfmatch_new(p(x["d1"]), 6, None, 0, p(x["seg1"]), None, None, p(F), 50.0, 0.7, p(m), 0) == 0 and (m == -1).all()
    m[:] = 5
    args = [p(x["d1"]), 6, p(x["d2"]), 7, p(x["seg1"]), p(x["seg2"]), p(x["func1"]), p(x["func2"]), p(F), p(F), p(x["ml1"]), p(x["ml2"]), 50.0, 0.7, 1,
            p(m), C.byref(c), 0]
    for k in (4, 6, 8, 10, 11):      # seg1, func1, F21, has_ml1, has_ml2 missing
        bad = list(args); bad[k] = None
        assert L.plh_line_search_for_triangulation_new(*bad) != 0, k
    bad = list(args); bad[16] = None
    assert L.plh_line_search_for_triangulation_new(*bad) != 0
    bad = list(args); bad[3] = -2
    assert L.plh_line_search_for_triangulation_new(*bad) != 0


def test_emu_lsdmatcher_new_error_behaviour(plslam, synth, emu_lib):
    _check_errors(plslam, synth, emu_lib)


@pytest.mark.gpu
def test_gpu_lsdmatcher_new_error_behaviour(plslam, synth):
    _check_errors(plslam, synth, None)


def test_emu_reproduces_reference_lsdmatcher_new(plslam, synth, emu_lib):
    """The HIP sources compiled for the host emulator (tests/hipemu), through the C ABI."""
    _check_device(plslam, synth, emu_lib)


@pytest.mark.gpu
def test_gpu_reproduces_reference_lsdmatcher_new(plslam, synth):
    _check_device(plslam, synth, None)
