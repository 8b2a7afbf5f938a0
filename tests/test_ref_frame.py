"""Pinning the spatial index under every windowed search against the REFERENCE's own code.  oracle/_ref/libframe_ref.so is
the reference's include/Frame.h + src/Frame.cc compiled as they are (oracle/ref/build_ref.sh) -- with the real MapPoint,
MapLine, ORBextractor, LINEextractor, DBoW2 and lineIterator sources around them and stand-ins for KeyFrame / Map /
Converter only.  The harness (oracle/ref/ref_frame.cc) fills a default-constructed Frame from flat arrays and calls

    Frame::AssignFeaturesToGrid (+ PosInGrid)     Frame::AssignFeaturesToGridForLine (with the real LineIterator)
    Frame::GetFeaturesInArea                      Frame::GetFeaturesInAreaForLine
    KeyFrame::GetFeaturesInArea                   KeyFrame::GetLinesInArea      (src/KeyFrame.cc, a real KeyFrame built from
                                                  the Frame: its image bounds are `const int`, i.e. truncated -- the
                                                  lookup still returns what the Frame-style lookup without level filter
                                                  returns, which is how the Fuse / Sim3 kernels generate candidates)

Pinned: which cell a keypoint lands in (round, not floor), every cell a line crosses, the cell ranges a window covers, the
level filter, the order in which candidates come back (cell-major, insertion order inside a cell; the line lookup's
three probe points and its de-duplication), the distance / direction tests.  The ORBmatcher / LSDmatcher pins
(tests/test_ref_orbmatcher*.py, test_ref_lsdmatcher.py) use the oracle's lookups underneath; this file closes that gap.

tools/gen_golden_ref.py committed the reference outputs as tests/golden/ref_framegrid.npz: the oracle reproduces grids and
lookups, the GPU kernels (`-m gpu`) and the HIP sources on the host emulator reproduce the grids (the lookups are internal
to the search kernels, which the matcher pins cover); in the build container the reference also runs live."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLDEN = os.path.join(_util.ROOT, "tests", "golden", "ref_framegrid.npz")
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "libframe_ref.so")
V, I, F = C.c_void_p, C.c_int, C.c_float


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _oracle_all(O, P, TF, f2, gp, pq, lv, seg, lr, lth):
    L = TF._olib(O)
    L.plo_features_in_area.argtypes = [V, V, V, V, F, F, F, I, I, V, I]
    L.plo_features_in_area.restype = I
    L.plo_features_in_area_for_line.argtypes = [V, V, I, V, V, V, F, F, F, F, F, F, V, I]
    L.plo_features_in_area_for_line.restype = I
    L.plo_keyframe_lines_in_area.argtypes = [V, I, F, F, F, F, F, F, V, I]
    L.plo_keyframe_lines_in_area.restype = I
    g = TF._gpa(P, gp)
    (cs, ci), (lcs, lci) = TF._oracle_grids(O, P, f2, gp)
    n, nl = len(f2["kps"]), len(f2["keylines"])
    buf = np.zeros(max(n, nl, 1) + 1, np.int32)
    pa, la = [], []
    for q in range(len(pq)):
        k = L.plo_features_in_area(O._p(f2["kps"]), O._p(g), O._p(cs), O._p(ci), float(pq[q, 0]), float(pq[q, 1]), float(pq[q, 2]),
                                   int(lv[q, 0]), int(lv[q, 1]), O._p(buf), len(buf))
        pa.append(buf[:k].copy())
        k = L.plo_features_in_area_for_line(O._p(f2["keylines"]), O._p(f2["linefn"]), nl, O._p(g), O._p(lcs), O._p(lci), float(seg[q, 0]),
                                            float(seg[q, 1]), float(seg[q, 2]), float(seg[q, 3]), float(lr[q]), float(lth[q]), O._p(buf),
                                            len(buf))
        la.append(buf[:k].copy())
    ka, kla = [], []   # KeyFrame::GetFeaturesInArea = the Frame lookup without level filter; KeyFrame::GetLinesInArea = brute force
    for q in range(len(pq)):
        k = L.plo_features_in_area(O._p(f2["kps"]), O._p(g), O._p(cs), O._p(ci), float(pq[q, 0]), float(pq[q, 1]), float(pq[q, 2]), -1, -1,
                                   O._p(buf), len(buf))
        ka.append(buf[:k].copy())
        k = L.plo_keyframe_lines_in_area(O._p(f2["keylines"]), nl, float(seg[q, 0]), float(seg[q, 1]), float(seg[q, 2]), float(seg[q, 3]),
                                         float(lr[q]) * 4, float(lth[q]), O._p(buf), len(buf))
        kla.append(buf[:k].copy())
    return cs, ci, lcs, lci, pa, la, ka, kla


def _same_lists(got, flat, offs):
    return len(got) == len(offs) - 1 and all((got[q] == flat[offs[q]:offs[q + 1]]).all() and len(got[q]) == offs[q + 1] - offs[q]
                                             for q in range(len(got)))


def test_golden_file_present():
    assert os.path.exists(GOLDEN)


def test_oracle_reproduces_reference_frame_index(oracle, plslam, synth):
    G = _gen()
    TF = G._test_module("test_frame_search")
    g = np.load(GOLDEN)
    for seed, n, nl, dist in G.FRAMEGRID_CASES:
        f2, gp, pq, lv, seg, lr, lth = G.framegrid_inputs(synth, plslam, TF, seed, n, nl, dist)
        cs, ci, lcs, lci, pa, la, ka, kla = _oracle_all(oracle, plslam, TF, f2, gp, pq, lv, seg, lr, lth)
        assert _same_lists(ka, g["ka_%d" % seed], g["ko_%d" % seed]), "KeyFrame::GetFeaturesInArea %d" % seed
        assert _same_lists(kla, g["kla_%d" % seed], g["klo_%d" % seed]), "KeyFrame::GetLinesInArea %d" % seed
        assert (cs == g["cs_%d" % seed]).all() and (ci[:cs[-1]] == g["ci_%d" % seed][:cs[-1]]).all(), "AssignFeaturesToGrid %d" % seed
        assert (lcs == g["lcs_%d" % seed]).all() and (lci[:lcs[-1]] == g["lci_%d" % seed]).all(), "AssignFeaturesToGridForLine %d" % seed
        assert _same_lists(pa, g["pa_%d" % seed], g["po_%d" % seed]), "GetFeaturesInArea %d" % seed
        assert _same_lists(la, g["la_%d" % seed], g["lo_%d" % seed]), "GetFeaturesInAreaForLine %d" % seed


def _device(P, synth, lib):
    G = _gen()
    TF = G._test_module("test_frame_search")
    g = np.load(GOLDEN)
    for seed, n, nl, dist in G.FRAMEGRID_CASES:
        f2, gp = G.framegrid_inputs(synth, P, TF, seed, n, nl, dist)[:2]
        (cs, ci), (lcs, lci) = P.FrameSearch(gp, TF.SCALE, [f2], lib=lib).grids()
        k, kl = int(g["cs_%d" % seed][-1]), int(g["lcs_%d" % seed][-1])
        assert (cs[0] == g["cs_%d" % seed]).all() and (ci[0, :k] == g["ci_%d" % seed][:k]).all(), "point grid %d" % seed
        assert (lcs[0] == g["lcs_%d" % seed]).all() and (lci[0, :kl] == g["lci_%d" % seed]).all(), "line grid %d" % seed


def test_emu_reproduces_reference_frame_grids(plslam, synth, emu_lib):
    _device(plslam, synth, emu_lib)


@pytest.mark.gpu
def test_gpu_reproduces_reference_frame_grids(plslam, synth):
    _device(plslam, synth, None)


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_frame_index_live(oracle, plslam, synth):
    G = _gen()
    TF = G._test_module("test_frame_search")
    R = G.ref_frame_lib()
    for seed, n, nl, dist in [(11, 1500, 150, True), (12, 64, 9, False), (13, 0, 0, False)]:
        f2, gp, pq, lv, seg, lr, lth = G.framegrid_inputs(synth, plslam, TF, seed, n, nl, dist)
        rcs, rci, rlcs, rlci, (rpa, rpo), (rla, rlo), (rka, rko), (rkla, rklo) = G.reference_framegrid(R, plslam, f2, gp, pq, lv, seg, lr, lth)
        cs, ci, lcs, lci, pa, la, ka, kla = _oracle_all(oracle, plslam, TF, f2, gp, pq, lv, seg, lr, lth)
        assert _same_lists(ka, rka, rko) and _same_lists(kla, rkla, rklo), "live keyframe lookups %d" % seed
        assert (cs == rcs).all() and (ci[:cs[-1]] == rci[:cs[-1]]).all() and (lcs == rlcs).all() and (lci[:lcs[-1]] == rlci[:lcs[-1]]).all()
        assert _same_lists(pa, rpa, rpo) and _same_lists(la, rla, rlo), "live lookups %d" % seed
