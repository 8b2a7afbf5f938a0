"""The local-map search of Tracking end to end against the REFERENCE's own object model.  oracle/_ref/libframe_ref.so
compiles src/Frame.cc, KeyFrame.cc, MapPoint.cc, MapLine.cpp, ORBmatcher.cc, LSDmatcher.cpp (and the extractors, DBoW2,
lineIterator) against their OWN headers -- stand-ins only for Map / KeyFrameDatabase / Converter and the OpenCV / Eigen
types.  The harness (oracle/ref/ref_frame.cc) builds a real Frame (keypoints, lines, grids via AssignFeaturesToGrid*), a
local map of real MapPoint / MapLine objects, and runs the core of Tracking::SearchLocalPoints / SearchLocalLines
(src/Tracking.cc:1772-1800, 1825-1849):

    for every local map element:  if (mCurrentFrame.isInFrustum(p, 0.5)) ...
    ORBmatcher(0.8).SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th)      LSDmatcher().SearchByProjection(..., th)

i.e. real isInFrustum + PredictScale -> mTrackProj* / mnTrackScaleLevel / mTrackViewCos -> real GetFeaturesInArea[ForLine]
-> the real matching loops writing mvpMapPoints / mvpMapLines.  The product does the same with two calls per feature
type: plh_frame_is_in_frustum_* -> plh_*_search_by_projection_{mp,ml}.  The same library runs TrackWithMotionModel's
search -- the real ORBmatcher(0.9, true).SearchByProjection(Cur, Last, th, mono) on two real Frames (src/ORBmatcher.cc:
1441-1585) -- against plh_frame_project_points (form 0) -> plh_orb_search_by_projection_frame, and
TrackReferenceKeyFrame's: Frame / KeyFrame::ComputeBoW with a real ORBVocabulary (DBoW2 text file written by
pl-slam_amd/vocab.py), then the real ORBmatcher::SearchByBoW(pKF, F) -- against plh_bow_transform -> plh_orb_search_by_bow.  Committed reference outputs:
tests/golden/ref_track.npz (camera without rotation, see tests/test_frustum.py); the oracle chain, the HIP sources on the
host emulator and the GPU (`-m gpu`) must reproduce which map element ends up on which keypoint / line."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLDEN = os.path.join(_util.ROOT, "tests", "golden", "ref_track.npz")
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "libframe_ref.so")
V, I, F = C.c_void_p, C.c_int, C.c_float


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _oracle_chain(O, P, TF, G, f2, gp, view, nlv, pts, lns, occ_p, occ_l, th):
    FR = G._test_module("test_frustum")
    L = TF._olib(O)
    g = TF._gpa(P, gp)
    (cs, ci), (lcs, lci) = TF._oracle_grids(O, P, f2, gp)
    n, nl = len(f2["kps"]), len(f2["keylines"])
    p = O._p
    valid, uv, level, vc = FR._oracle(O, view, nlv, pts, 0, 0.5)
    valid, uv, level, vc = (np.ascontiguousarray(a) for a in (valid, uv, level, vc))
    op, ap = occ_p.copy(), np.zeros(max(n, 1), np.int32)
    cp = L.plo_orb_search_by_projection_mp(p(f2["kps"]), p(f2["desc"]), n, p(g), p(cs), p(ci), p(TF.SCALE), p(op), len(valid), p(valid), p(uv),
                                           p(level), p(vc), p(pts["desc"]), p(pts["hasobs"]), th, 0.8, p(ap))
    valid, seg, level, vc = (np.ascontiguousarray(a) for a in FR._oracle(O, view, nlv, lns, 1, 0.5))
    ol, al = occ_l.copy(), np.zeros(max(nl, 1), np.int32)
    cl = L.plo_line_search_by_projection_ml(p(f2["keylines"]), p(f2["ldesc"]), p(f2["linefn"]), nl, p(g), p(lcs), p(lci), p(ol), len(valid),
                                            p(valid), p(seg), p(vc), p(lns["desc"]), p(lns["hasobs"]), th, 0.7, p(al))
    return (cp, ap[:n], op), (cl, al[:nl], ol)


def _device_chain(P, TF, G, lib, f2, gp, view, nlv, pts, lns, occ_p, occ_l, th):
    FR = G._test_module("test_frustum")
    n, nl = len(f2["kps"]), len(f2["keylines"])
    rec = np.array([FR._view_record(P, view, nlv)], P.VIEW_DTYPE)
    fs = P.FrameSearch(gp, TF.SCALE, [f2], lib=lib)
    q = P.is_in_frustum(rec, [pts], 0.5, lines=False, lib=lib)[0]
    q = dict(valid=q["valid"], xy=q["uv"], level=q["level"], viewcos=q["viewcos"], desc=pts["desc"], hasobs=pts["hasobs"])
    ap, cp, op = fs.SearchByProjectionMapPoints([q], [occ_p], th=th, nnratio=0.8)
    q = P.is_in_frustum(rec, [lns], 0.5, lines=True, lib=lib)[0]
    q = dict(valid=q["valid"], seg=q["seg"], viewcos=q["viewcos"], desc=lns["desc"], hasobs=lns["hasobs"])
    al, cl, ol = fs.LineSearchByProjectionMapLines([q], [occ_l], th=th, nnratio=0.7)
    return (cp[0], ap[0, :n], op[0, :max(n, 1)][:len(occ_p)]), (cl[0], al[0, :nl], ol[0, :max(nl, 1)][:len(occ_l)])


def _motion_oracle(O, P, TF, G, f2, gp, view, nlv, pts, flags, q, occ, th):
    FR = G._test_module("test_frustum")
    L = TF._olib(O)
    g = TF._gpa(P, gp)
    (cs, ci), _ = TF._oracle_grids(O, P, f2, gp)
    n, p = len(f2["kps"]), O._p
    front, uv = FR._oracle_project(O, view, 0, pts["pos"])
    valid = np.ascontiguousarray(flags["mp"] & (1 - flags["outlier"]) & front)
    uv = np.ascontiguousarray(uv)
    o, a = occ.copy(), np.zeros(max(n, 1), np.int32)
    c = L.plo_orb_search_by_projection_frame(p(f2["kps"]), p(f2["desc"]), n, p(g), p(cs), p(ci), p(TF.SCALE), p(o), len(valid), p(valid), p(uv),
                                             p(q["octave"]), p(q["angle"]), p(pts["desc"]), p(pts["hasobs"]), th, 0, 1, p(a))
    return c, a[:n], o


def _motion_device(P, TF, G, lib, f2, gp, view, nlv, pts, flags, q, occ, th):
    FR = G._test_module("test_frustum")
    n = len(f2["kps"])
    rec = np.array([FR._view_record(P, view, nlv)], P.VIEW_DTYPE)
    front, uv = P.project_points(rec, [pts["pos"]], 0, lib=lib)[0]
    qd = dict(valid=(flags["mp"] & (1 - flags["outlier"]) & front).astype(np.uint8), uv=uv, octave=q["octave"], angle=q["angle"],
              desc=pts["desc"], hasobs=pts["hasobs"])
    a, c, o = P.FrameSearch(gp, TF.SCALE, [f2], lib=lib).SearchByProjectionLastFrame([qd], [occ], th=th, mode=0, checkOri=True)
    return c[0], a[0, :n], o[0, :max(n, 1)][:len(occ)]


def _check_motion(run, G, P, S):
    TF = G._test_module("test_frame_search")
    g = np.load(GOLDEN)
    for seed, n, nl, dist, _ in G.TRACK_CASES:
        f2, gp, view, nlv, pts, lns, occ_p, occ_l = G.track_inputs(S, P, TF, seed, n, nl, dist)
        flags, q = G.track_last_inputs(S, P, TF, seed, n, nl, dist)
        c, a, o = run(TF, f2, gp, view, nlv, pts, flags, q, occ_p, 15.0 if seed != 2 else 7.0)
        assert c == int(g["m_%d_n" % seed]) and (a == g["m_%d_asg" % seed]).all() and (o == g["m_%d_occ" % seed]).all(), "motion model %d" % seed
        assert c > n // 3


def test_oracle_chain_reproduces_reference_motion_model_search(oracle, plslam, synth):
    G = _gen()
    _check_motion(lambda TF, *a: _motion_oracle(oracle, plslam, TF, G, *a), G, plslam, synth)


def test_emu_chain_reproduces_reference_motion_model_search(plslam, synth, emu_lib):
    G = _gen()
    _check_motion(lambda TF, *a: _motion_device(plslam, TF, G, emu_lib, *a), G, plslam, synth)


@pytest.mark.gpu
def test_gpu_chain_reproduces_reference_motion_model_search(plslam, synth):
    G = _gen()
    _check_motion(lambda TF, *a: _motion_device(plslam, TF, G, None, *a), G, plslam, synth)


def _bow_device(P, VMod, lib, voc, kf, fr, nn, chk):
    nid, _ = P.bow_transform([kf["desc"], fr["desc"]], voc, levelsup=4, lib=lib)
    n1, n2 = len(kf["desc"]), len(fr["desc"])
    a = dict(desc=kf["desc"], angle=kf["kps"]["angle"].astype(np.float32), node=np.ascontiguousarray(nid[0, :n1]), valid=kf["valid"])
    b = dict(desc=fr["desc"], angle=fr["kps"]["angle"].astype(np.float32), node=np.ascontiguousarray(nid[1, :n2]))
    m, c = P.ORBmatcher(nn, bool(chk), lib=lib).SearchByBoWBatch([a], [b])
    return c[0], m[0, :n2]


def _check_bow(run, G, P, S, cases=None):
    VM = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    g = np.load(GOLDEN)
    for seed, k, Lv, n, nn, chk in (cases or G.BOWTRACK_CASES):
        voc, kf, fr = G.bowtrack_inputs(S, P, VM, seed, k, Lv, n)
        c, m = run(VM, voc, kf, fr, nn, chk)
        assert c == int(g["b_%d_n" % seed]) and (m == g["b_%d_m" % seed]).all(), "reference keyframe search %d" % seed
        assert c > n // 3


def _bow_oracle(O, G, voc, kf, fr, nn, chk):
    TB, TM = G._test_module("test_bow"), G._test_module("test_match")
    a = dict(desc=kf["desc"], angle=kf["kps"]["angle"].astype(np.float32), node=np.ascontiguousarray(TB._oracle_transform(O, kf["desc"], voc, 4)[0]),
             valid=kf["valid"])
    b = dict(desc=fr["desc"], angle=fr["kps"]["angle"].astype(np.float32), node=np.ascontiguousarray(TB._oracle_transform(O, fr["desc"], voc, 4)[0]))
    c, m = TM._oracle_bow(O, a, b, 50, nn, bool(chk))
    return c, m[:len(fr["desc"])]


def test_oracle_chain_reproduces_reference_keyframe_search(oracle, plslam, synth):
    G = _gen()
    _check_bow(lambda VM, *a: _bow_oracle(oracle, G, *a), G, plslam, synth, G.BOWTRACK_CASES[:2])   # the 10^6-node one: GPU only


def test_emu_chain_reproduces_reference_keyframe_search(plslam, synth, emu_lib):
    G = _gen()
    _check_bow(lambda VM, *a: _bow_device(plslam, VM, emu_lib, *a), G, plslam, synth, G.BOWTRACK_CASES[:2])   # the 10^6-node one: GPU only


@pytest.mark.gpu
def test_gpu_chain_reproduces_reference_keyframe_search(plslam, synth):
    G = _gen()
    _check_bow(lambda VM, *a: _bow_device(plslam, VM, None, *a), G, plslam, synth)


def _check(run, G, P, S):
    TF = G._test_module("test_frame_search")
    g = np.load(GOLDEN)
    for seed, n, nl, dist, th in G.TRACK_CASES:
        args = G.track_inputs(S, P, TF, seed, n, nl, dist)
        (cp, ap, op), (cl, al, ol) = run(TF, *args, th)
        assert cp == int(g["p_%d_n" % seed]) and (ap == g["p_%d_asg" % seed]).all() and (op == g["p_%d_occ" % seed]).all(), "points %d" % seed
        assert cl == int(g["l_%d_n" % seed]) and (al == g["l_%d_asg" % seed]).all() and (ol == g["l_%d_occ" % seed]).all(), "lines %d" % seed
        assert cp > n // 3 and cl >= nl // 3


def test_golden_file_present():
    assert os.path.exists(GOLDEN)


def test_oracle_chain_reproduces_reference_local_map_search(oracle, plslam, synth):
    G = _gen()
    _check(lambda TF, *a: _oracle_chain(oracle, plslam, TF, G, *a), G, plslam, synth)


def test_emu_chain_reproduces_reference_local_map_search(plslam, synth, emu_lib):
    G = _gen()
    _check(lambda TF, *a: _device_chain(plslam, TF, G, emu_lib, *a), G, plslam, synth)


@pytest.mark.gpu
def test_gpu_chain_reproduces_reference_local_map_search(plslam, synth):
    G = _gen()
    _check(lambda TF, *a: _device_chain(plslam, TF, G, None, *a), G, plslam, synth)


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_local_map_search_live(oracle, plslam, synth):
    G = _gen()
    TF = G._test_module("test_frame_search")
    R = G.ref_frame_lib()
    for seed, n, nl, dist, th in [(21, 1500, 150, True, 1.0), (22, 300, 40, False, 5.0), (23, 0, 0, False, 1.0)]:
        args = G.track_inputs(synth, plslam, TF, seed, n, nl, dist)
        ref = G.reference_track(R, plslam, TF, *args, th)
        got = _oracle_chain(oracle, plslam, TF, G, *args, th)
        for (rc, ra, ro), (c, a, o) in zip(ref, got):
            assert rc == c and (ra == a).all() and (ro == o).all(), "live local map search %d" % seed
