"""Pinning the line matcher against the REFERENCE's own code.  oracle/_ref/liblsdmatcher_ref.so is the reference's
src/LSDmatcher.cpp -- every function -- compiled from the source where it lies (oracle/ref/build_ref.sh) against the
stand-ins of oracle/ref/slam_stub.h (Frame, KeyFrame, MapLine holding just the members the matcher touches, filled from
flat arrays by oracle/ref/ref_lsdmatcher.cc).  cv::BFMatcher::knnMatch underneath is the oracle's knn2 (OpenCV itself is
not in the tree); the grid lookup behind Frame::GetFeaturesInAreaForLine is the oracle's.  Entry points:

    FrameBFMatch (+ lineDescriptorMAD)     SearchDouble(Frame&, Frame&, LineMatches)
    SearchByProjection(Cur, Last, th)      SearchByProjection(F, vpMapLines, th)
    Fuse(pKF, vpMapLines, th)  -- keyframe pose = identity; the projected segments come back from the harness' copy of
                                  the reference's expressions; KeyFrame::GetLinesInArea underneath is restated (KeyFrame.cc
                                  cannot be compiled alone), so what is pinned is the loop: level band, the candidate rows
                                  read from pKF->mDescriptors, best / TH_LOW, and the `return false` that leaves the whole
                                  function at the first line with an endpoint behind the camera (LSDmatcher.cpp:893-894)

What this pins: the MAD thresholds (median selection, the 1.4826 factor), the ratio / TH / nn12 acceptance tests, the
mutual-consistency pass, and the greedy projection searches (candidate order, best / second best, occupancy rules,
TH_HIGH, length and angle gates) -- bit for bit.  SearchDouble(KeyFrame*, Frame&) and SearchForTriangulation are
FrameBFMatch at TH_LOW / TH_HIGH plus bookkeeping, covered through the FrameBFMatch goldens at 50 and 80.
tools/gen_golden_ref.py committed the reference outputs as tests/golden/ref_lsdmatcher.npz; the oracle (CPU) and the GPU
kernels (`-m gpu`) must reproduce them, and in the build container the reference is also run live."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLDEN = os.path.join(_util.ROOT, "tests", "golden", "ref_lsdmatcher.npz")
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "liblsdmatcher_ref.so")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _oracle_proj(O, L, P, TF, f2, gp, q, occ0, variant, th, nn):
    p = O._p
    g = TF._gpa(P, gp)
    _, (lcs, lci) = TF._oracle_grids(O, P, f2, gp)
    n2 = len(f2["keylines"])
    ro, ra = occ0.copy(), np.zeros(max(n2, 1), np.int32)
    if variant == "ml":
        rc = L.plo_line_search_by_projection_ml(p(f2["keylines"]), p(f2["ldesc"]), p(f2["linefn"]), n2, p(g), p(lcs), p(lci), p(ro),
                                                len(q["valid"]), p(q["valid"]), p(q["seg"]), p(q["viewcos"]), p(q["desc"]),
                                                p(q["hasobs"]), th, nn, p(ra))
    else:
        rc = L.plo_line_search_by_projection_frame(p(f2["keylines"]), p(f2["ldesc"]), p(f2["linefn"]), n2, p(g), p(lcs), p(lci), p(ro),
                                                   len(q["valid"]), p(q["valid"]), p(q["seg"]), p(q["length"]), p(q["desc"]),
                                                   p(q["hasobs"]), th, p(ra))
    return rc, ra[:n2], ro


def _oracle_lfuse(O, L, f2, q, seg, valid, level, th):
    nl, n2 = len(level), len(f2["keylines"])
    L.plo_line_fuse_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.plo_line_fuse_search.restype = C.c_int
    rb = np.zeros(max(nl, 1), np.int32)
    rc = L.plo_line_fuse_search(O._p(f2["keylines"]), O._p(f2["ldesc"]), n2, O._p(_SFL), nl, O._p(valid), O._p(seg), O._p(level),
                                O._p(q["desc"]), th, 0.998, 50, O._p(rb))
    return rc, rb[:nl]


_SFL = np.array([1.0, 1.4142135, 2.0, 2.828427], np.float32)


def _oracle_bf(O, a, b, th, ratio):
    m = np.zeros(max(len(a), 1), np.int32)
    O.lib().plo_line_bfmatch(O._p(a), len(a), O._p(b), len(b), C.c_float(th), C.c_float(ratio), O._p(m))
    return m[:len(a)]


def test_golden_file_present():
    assert os.path.exists(GOLDEN)


def test_oracle_reproduces_reference_lsdmatcher(oracle, plslam, synth):
    G = _gen()
    TM, TF = G._test_module("test_match"), G._test_module("test_frame_search")
    g = np.load(GOLDEN)
    L = TF._olib(oracle)
    for seed, n1, n2, flip, ratio in G.LDOUBLE_CASES:
        a, b = G.ldouble_inputs(synth, seed, n1, n2, flip)
        rc, ref = TM._oracle_double(oracle, a, b, 50.0, ratio)
        assert rc == int(g["dbl_%d_n" % seed]) and (ref[:n1] == g["dbl_%d_m" % seed]).all(), "SearchDouble %d" % seed
        for th in G.LBF_TH:
            assert (_oracle_bf(oracle, a, b, th, ratio) == g["bf_%d_%d" % (seed, int(th))]).all(), "FrameBFMatch %d %g" % (seed, th)
    for seed, nl, dist in G.LPROJ_CASES:
        for k, (variant, th, nn) in enumerate(G.LPROJ_VARIANTS):
            f2, gp, q, occ0 = G.lproj_inputs(synth, plslam, TF, seed, nl, dist, variant)
            rc, ra, ro = _oracle_proj(oracle, L, plslam, TF, f2, gp, q, occ0, variant, th, nn)
            key = "proj_%d_%d" % (seed, k)
            assert rc == int(g[key + "_n"]) and (ra == g[key + "_asg"]).all() and (ro == g[key + "_occ"]).all(), key
    for seed, nl, dist, nb in G.LFUSE_CASES:
        f2, gp, q, pos, fl, kfml, level = G.lfuse_inputs(synth, plslam, TF, seed, nl, dist, nb)
        for k, th in enumerate(G.LFUSE_TH):
            rc, rb = _oracle_lfuse(oracle, L, f2, q, np.ascontiguousarray(g["lfuse_%d_seg" % seed]),
                                   np.ascontiguousarray(g["lfuse_%d_valid" % seed]), level, th)
            assert (rb == g["lfuse_%d_%d_best" % (seed, k)]).all(), "Fuse %d %g" % (seed, th)
            assert int(g["lfuse_%d_%d_n" % (seed, k)]) == (0 if int(g["lfuse_%d_stopped" % seed]) else rc)


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_lsdmatcher_live(oracle, plslam, synth):
    G = _gen()
    TM, TF = G._test_module("test_match"), G._test_module("test_frame_search")
    R = G.ref_lsdmatcher_lib()
    L = TF._olib(oracle)
    for seed, n1, n2, flip, ratio in [(71, 200, 230, 0.15, 0.8), (72, 33, 31, 0.22, 0.7), (73, 640, 3, 0.1, 0.9)]:
        a, b = G.ldouble_inputs(synth, seed, n1, n2, flip)
        c, m = G.reference_ldouble(R, a, b, ratio)
        rc, ref = TM._oracle_double(oracle, a, b, 50.0, ratio)
        assert c == rc and (m == ref[:n1]).all(), "live SearchDouble %d" % seed
        for th in (80.0, 40.0):
            assert (G.reference_lbf(R, a, b, th, ratio) == _oracle_bf(oracle, a, b, th, ratio)).all(), "live FrameBFMatch %d" % seed
    for seed, nl, dist in [(81, 150, True), (82, 420, False), (83, 0, False)]:
        for variant, th, nn in G.LPROJ_VARIANTS:
            f2, gp, q, occ0 = G.lproj_inputs(synth, plslam, TF, seed, nl, dist, variant)
            c, a, o = G.reference_lproj(R, plslam, f2, gp, q, occ0, variant, th, nn)
            rc, ra, ro = _oracle_proj(oracle, L, plslam, TF, f2, gp, q, occ0, variant, th, nn)
            assert c == rc and (a == ra).all() and (o == ro).all(), "live SearchByProjection %s %d" % (variant, seed)
    for seed, nl, dist, nb in [(91, 250, True, 0), (92, 90, False, 1)]:
        f2, gp, q, pos, fl, kfml, level = G.lfuse_inputs(synth, plslam, TF, seed, nl, dist, nb)
        c, best, seg, valid, stopped = G.reference_lfuse(R, plslam, f2, gp, q, pos, fl, kfml, level, 5.0)
        rc, rb = _oracle_lfuse(oracle, L, f2, q, seg, valid, level, 5.0)
        assert (best == rb).all() and c == (0 if stopped else rc), "live Fuse %d" % seed


def _check_device(P, synth, lib):
    G = _gen()
    TF = G._test_module("test_frame_search")
    g = np.load(GOLDEN)
    for seed, n1, n2, flip, ratio in G.LDOUBLE_CASES:
        a, b = G.ldouble_inputs(synth, seed, n1, n2, flip)
        m = P.LSDmatcher(ratio, True, lib=lib)
        c, got = m.SearchDouble(a, b)
        assert c == int(g["dbl_%d_n" % seed]) and (got == g["dbl_%d_m" % seed]).all(), "SearchDouble %d" % seed
        for th in G.LBF_TH:
            assert (m.FrameBFMatch(a, b, TH=th) == g["bf_%d_%d" % (seed, int(th))]).all(), "FrameBFMatch %d %g" % (seed, th)
    for seed, nl, dist in G.LPROJ_CASES:
        for k, (variant, th, nn) in enumerate(G.LPROJ_VARIANTS):
            f2, gp, q, occ0 = G.lproj_inputs(synth, P, TF, seed, nl, dist, variant)
            fs = P.FrameSearch(gp, TF.SCALE, [f2], lib=lib)
            if variant == "ml":
                asg, cnt, occ = fs.LineSearchByProjectionMapLines([q], [occ0], th=th, nnratio=nn)
            else:
                asg, cnt, occ = fs.LineSearchByProjectionLastFrame([q], [occ0], th=th)
            n2_ = len(f2["keylines"])
            key = "proj_%d_%d" % (seed, k)
            assert cnt[0] == int(g[key + "_n"]) and (asg[0, :n2_] == g[key + "_asg"]).all() and (occ[0, :n2_] == g[key + "_occ"]).all(), key
    for seed, nl, dist, nb in G.LFUSE_CASES:
        f2, gp, q, pos, fl, kfml, level = G.lfuse_inputs(synth, P, TF, seed, nl, dist, nb)
        fs = P.FrameSearch(gp, TF.SCALE, [f2], lib=lib)
        qd = dict(valid=np.ascontiguousarray(g["lfuse_%d_valid" % seed]), seg=np.ascontiguousarray(g["lfuse_%d_seg" % seed]), level=level,
                  desc=q["desc"])
        for k, th in enumerate(G.LFUSE_TH):
            best, nf = fs.LineFuseSearch([qd], _SFL, th=th, cos_th=0.998)
            assert (best[0, :nl] == g["lfuse_%d_%d_best" % (seed, k)]).all(), "Fuse %d %g" % (seed, th)
            assert int(g["lfuse_%d_%d_n" % (seed, k)]) == (0 if int(g["lfuse_%d_stopped" % seed]) else nf[0])


def test_emu_reproduces_reference_lsdmatcher(plslam, synth, emu_lib):
    """The HIP sources compiled for the host emulator (tests/hipemu), through the C ABI."""
    _check_device(plslam, synth, emu_lib)


@pytest.mark.gpu
def test_gpu_reproduces_reference_lsdmatcher(plslam, synth):
    _check_device(plslam, synth, None)
