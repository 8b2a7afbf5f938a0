"""Pinning the ORB matcher against the REFERENCE's own code.  oracle/_ref/libmatcher_ref.so is the reference's
src/ORBmatcher.cc -- every function -- compiled from the source where it lies (oracle/ref/build_ref.sh).  The SLAM object
model it is written against (Frame, KeyFrame, MapPoint pull in the whole system) is replaced by stand-ins holding just the
members the matcher touches (oracle/ref/slam_stub.h, filled from flat arrays by oracle/ref/ref_matcher.cc); the grid
lookup behind Frame::GetFeaturesInArea is the oracle's.  Entry points:

    SearchByBoW(KeyFrame*, Frame&)   SearchByBoW(KeyFrame*, KeyFrame*)   SearchForInitialization
    SearchByProjection(Frame&, vpMapPoints, th)          (+ ComputeThreeMaxima, DescriptorDistance underneath)
    SearchByProjection(Cur, Last, th, bMono)  -- the tracking search: mono / forward / backward level bands
    SearchByProjection(Cur, pKF, sAlreadyFound, th, ORBdist)  -- relocalisation

The last two project map points inside the function; they are driven with the current pose = identity (Rcw*x+tcw then
returns x exactly) and the harness hands back the (u, v) its copy of the reference's projection expression produced,
which is what the flat-array searches take as q_uv.

What this pins: the greedy matching loops themselves -- candidate order, best / second-best bookkeeping, ratio tests,
occupancy rules, the rotation histogram and its three maxima -- bit for bit.  tools/gen_golden_ref.py committed the
reference outputs as tests/golden/ref_orbmatcher.npz; the oracle (CPU) and the GPU kernels (`-m gpu`) must reproduce them,
and in the build container the reference is also run live."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLDEN = os.path.join(_util.ROOT, "tests", "golden", "ref_orbmatcher.npz")
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "libmatcher_ref.so")
V, I, F = C.c_void_p, C.c_int, C.c_float


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _mods(G):
    return G._test_module("test_match"), G._test_module("test_frame_search")


def _oracle_pose(O, LL, P, TF, f2, gp, q, uv, valid, occ0, form, th, a, chk):
    n = len(f2["kps"])
    ga = TF._gpa(P, gp)
    (cs, ci), _ = TF._oracle_grids(O, P, f2, gp)
    uv, valid = np.ascontiguousarray(uv), np.ascontiguousarray(valid)
    ro, ra = occ0.copy(), np.zeros(max(n, 1), np.int32)
    if form == "frame":
        rc = LL.plo_orb_search_by_projection_frame(O._p(f2["kps"]), O._p(f2["desc"]), n, O._p(ga), O._p(cs), O._p(ci), O._p(TF.SCALE),
                                                   O._p(ro), n, O._p(valid), O._p(uv), O._p(q["octave"]), O._p(q["angle"]),
                                                   O._p(q["desc"]), O._p(q["hasobs"]), th, a, chk, O._p(ra))
    else:
        ones = np.ones(max(n, 1), np.uint8)
        rc = LL.plo_orb_search_by_projection_kf(O._p(f2["kps"]), O._p(f2["desc"]), n, O._p(ga), O._p(cs), O._p(ci), O._p(TF.SCALE),
                                                O._p(ro), n, O._p(valid), O._p(uv), O._p(q["octave"]), O._p(q["angle"]),
                                                O._p(q["desc"]), O._p(ones), th, a, chk, O._p(ra))
    return rc, ra[:n], ro


def test_golden_file_present():
    assert os.path.exists(GOLDEN)


def test_oracle_reproduces_reference_orbmatcher(oracle, plslam, synth):
    G = _gen()
    TM, TF = _mods(G)
    g = np.load(GOLDEN)
    O, L = oracle, oracle.lib()
    for seed, n, nodes, nn, chk in G.BOW_CASES:
        kf, fr = G.bow_inputs(synth, TM, seed, n, nodes)
        rc, ref = TM._oracle_bow(O, kf, fr, 50, nn, bool(chk))
        assert rc == int(g["bow_%d_n" % seed]) and (ref[:n] == g["bow_%d_m" % seed]).all(), "SearchByBoW %d" % seed
    L.plo_orb_search_by_bow_kfkf.argtypes = [V, V, V, V, I, V, V, V, V, I, I, F, I, V]
    L.plo_orb_search_by_bow_kfkf.restype = I
    for seed, n, nodes in G.KFKF_CASES:
        kf, fr, v2 = G.kfkf_inputs(synth, TM, seed, n, nodes)
        ref = np.zeros(n, np.int32)
        rc = L.plo_orb_search_by_bow_kfkf(O._p(kf["desc"]), O._p(kf["angle"]), O._p(kf["node"]), O._p(kf["valid"]), n, O._p(fr["desc"]),
                                          O._p(fr["angle"]), O._p(fr["node"]), O._p(v2), n, 50, 0.8, 1, O._p(ref))
        assert rc == int(g["kfkf_%d_n" % seed]) and (ref == g["kfkf_%d_m" % seed]).all(), "SearchByBoW(KF,KF) %d" % seed
    LL = TF._olib(O)
    for seed, n, dist in G.FRAME_CASES:
        f1, f2, gp, q, occ0 = G.frame_inputs(synth, plslam, TF, seed, n, dist)
        ga = TF._gpa(plslam, gp)
        (cs, ci), _ = TF._oracle_grids(O, plslam, f2, gp)
        prev = np.stack([f1["kps"]["x"], f1["kps"]["y"]], 1).astype(np.float32)
        ref = np.zeros(n, np.int32)
        rc = LL.plo_orb_search_for_initialization(O._p(f1["kps"]), O._p(f1["desc"]), n, O._p(f2["kps"]), O._p(f2["desc"]), n, O._p(ga),
                                                  O._p(cs), O._p(ci), O._p(prev), 100, 0.9, 1, O._p(ref))
        assert rc == int(g["init_%d_n" % seed]) and (ref == g["init_%d_m" % seed]).all() and (prev == g["init_%d_prev" % seed]).all()
        ro, ra = occ0.copy(), np.zeros(n, np.int32)
        rc = LL.plo_orb_search_by_projection_mp(O._p(f2["kps"]), O._p(f2["desc"]), n, O._p(ga), O._p(cs), O._p(ci), O._p(TF.SCALE),
                                                O._p(ro), len(q["valid"]), O._p(q["valid"]), O._p(q["xy"]), O._p(q["level"]),
                                                O._p(q["viewcos"]), O._p(q["desc"]), O._p(q["hasobs"]), 3.0, 0.8, O._p(ra))
        assert rc == int(g["proj_%d_n" % seed]) and (ra == g["proj_%d_asg" % seed]).all() and (ro == g["proj_%d_occ" % seed]).all()
    LL.plo_orb_search_by_projection_kf.argtypes = [V, V, I, V, V, V, V, V, I, V, V, V, V, V, V, F, I, I, V]
    LL.plo_orb_search_by_projection_kf.restype = I
    for seed, n, dist in G.POSE_CASES:
        f2, gp, q, xyz, fl, occ_f, occ_k = G.pose_inputs(synth, plslam, TF, seed, n, dist)
        uv = g["pose_%d_uv" % seed]
        for k, (mode, th, chk) in enumerate(G.POSE_FRAME_VARIANTS):
            rc, ra, ro = _oracle_pose(O, LL, plslam, TF, f2, gp, q, uv, g["pf_%d_valid" % seed], occ_f, "frame", th, mode, chk)
            key = "pf_%d_%d" % (seed, k)
            assert rc == int(g[key + "_n"]) and (ra == g[key + "_asg"]).all() and (ro == g[key + "_occ"]).all(), key
        for k, (orb_dist, chk) in enumerate(G.POSE_KF_VARIANTS):
            rc, ra, ro = _oracle_pose(O, LL, plslam, TF, f2, gp, q, uv, g["pk_%d_valid" % seed], occ_k, "kf", 10.0, orb_dist, chk)
            key = "pk_%d_%d" % (seed, k)
            assert rc == int(g[key + "_n"]) and (ra == g[key + "_asg"]).all() and (ro == g[key + "_occ"]).all(), key


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_orbmatcher_live(oracle, plslam, synth):
    G = _gen()
    TM, TF = _mods(G)
    R = G.ref_matcher_lib()
    for seed, n, nodes, nn, chk in [(310, 1200, 60, 0.6, 1), (311, 500, 3, 0.75, 1), (312, 64, 64, 0.9, 0), (313, 1, 1, 0.7, 1)]:
        kf, fr = G.bow_inputs(synth, TM, seed, n, nodes)
        c, m = G.reference_bow(R, kf, fr, nn, chk)
        rc, ref = TM._oracle_bow(oracle, kf, fr, 50, nn, bool(chk))
        assert c == rc and (m == ref[:n]).all(), "live SearchByBoW %d" % seed
    LL = TF._olib(oracle)
    for seed, n, dist in [(41, 900, False), (42, 120, True)]:
        f1, f2, gp, q, occ0 = G.frame_inputs(synth, plslam, TF, seed, n, dist)
        ga = TF._gpa(plslam, gp)
        (cs, ci), _ = TF._oracle_grids(oracle, plslam, f2, gp)
        c, m, pv = G.reference_init(R, plslam, f1, f2, gp)
        prev = np.stack([f1["kps"]["x"], f1["kps"]["y"]], 1).astype(np.float32)
        ref = np.zeros(n, np.int32)
        rc = LL.plo_orb_search_for_initialization(oracle._p(f1["kps"]), oracle._p(f1["desc"]), n, oracle._p(f2["kps"]),
                                                  oracle._p(f2["desc"]), n, oracle._p(ga), oracle._p(cs), oracle._p(ci), oracle._p(prev),
                                                  100, 0.9, 1, oracle._p(ref))
        assert c == rc and (m == ref).all() and (pv == prev).all(), "live SearchForInitialization %d" % seed
    LL.plo_orb_search_by_projection_kf.argtypes = [V, V, I, V, V, V, V, V, I, V, V, V, V, V, V, F, I, I, V]
    LL.plo_orb_search_by_projection_kf.restype = I
    for seed, n, dist in [(61, 1500, True), (62, 333, False), (63, 0, False)]:
        f2, gp, q, xyz, fl, occ_f, occ_k = G.pose_inputs(synth, plslam, TF, seed, n, dist)
        for mode, th, chk in [(0, 15.0, 1), (1, 7.0, 1), (2, 30.0, 0)]:
            c, a, o, uv, valid = G.reference_pose_frame(R, plslam, TF, f2, gp, q, xyz, fl, occ_f, mode, th, chk)
            rc, ra, ro = _oracle_pose(oracle, LL, plslam, TF, f2, gp, q, uv, valid, occ_f, "frame", th, mode, chk)
            assert c == rc and (a == ra).all() and (o == ro).all(), "live SearchByProjection(Cur, Last) %d mode %d" % (seed, mode)
        for orb_dist, chk in [(100, 1), (30, 0)]:
            c, a, o, uv, valid = G.reference_pose_kf(R, plslam, TF, f2, gp, q, xyz, fl, occ_k, orb_dist, chk)
            rc, ra, ro = _oracle_pose(oracle, LL, plslam, TF, f2, gp, q, uv, valid, occ_k, "kf", 10.0, orb_dist, chk)
            assert c == rc and (a == ra).all() and (o == ro).all(), "live SearchByProjection(Cur, pKF) %d" % seed


def _check_device(P, synth, lib):
    G = _gen()
    TM, TF = _mods(G)
    g = np.load(GOLDEN)
    for seed, n, nodes, nn, chk in G.BOW_CASES:
        kf, fr = G.bow_inputs(synth, TM, seed, n, nodes)
        got, cnt = P.ORBmatcher(nn, bool(chk), lib=lib).SearchByBoWBatch([kf], [fr])
        assert cnt[0] == int(g["bow_%d_n" % seed]) and (got[0, :n] == g["bow_%d_m" % seed]).all(), "SearchByBoW %d" % seed
    for seed, n, dist in G.FRAME_CASES:
        f1, f2, gp, q, occ0 = G.frame_inputs(synth, P, TF, seed, n, dist)
        fs = P.FrameSearch(gp, TF.SCALE, [f2], lib=lib)
        prev = np.stack([f1["kps"]["x"], f1["kps"]["y"]], 1).astype(np.float32)
        m12, cnt, pm = fs.SearchForInitialization([f1], [prev], 100, 0.9, True)
        assert cnt[0] == int(g["init_%d_n" % seed]) and (m12[0, :n] == g["init_%d_m" % seed]).all()
        assert (pm[0, :n] == g["init_%d_prev" % seed]).all()
        asg, cnt, occ = fs.SearchByProjectionMapPoints([q], [occ0], th=3.0, nnratio=0.8)
        assert cnt[0] == int(g["proj_%d_n" % seed]) and (asg[0, :n] == g["proj_%d_asg" % seed]).all()
        assert (occ[0, :n] == g["proj_%d_occ" % seed]).all()
    for seed, n, dist in G.POSE_CASES:
        f2, gp, q, xyz, fl, occ_f, occ_k = G.pose_inputs(synth, P, TF, seed, n, dist)
        fs = P.FrameSearch(gp, TF.SCALE, [f2], lib=lib)
        uv = np.ascontiguousarray(g["pose_%d_uv" % seed])
        qf = dict(valid=g["pf_%d_valid" % seed], uv=uv, octave=q["octave"], angle=q["angle"], desc=q["desc"], hasobs=q["hasobs"])
        for k, (mode, th, chk) in enumerate(G.POSE_FRAME_VARIANTS):
            asg, cnt, occ = fs.SearchByProjectionLastFrame([qf], [occ_f], th=th, mode=mode, checkOri=bool(chk))
            key = "pf_%d_%d" % (seed, k)
            assert cnt[0] == int(g[key + "_n"]) and (asg[0, :n] == g[key + "_asg"]).all() and (occ[0, :n] == g[key + "_occ"]).all(), key
        qk = dict(valid=g["pk_%d_valid" % seed], uv=uv, level=q["octave"], angle=q["angle"], desc=q["desc"], hasobs=np.ones(n, np.uint8))
        for k, (orb_dist, chk) in enumerate(G.POSE_KF_VARIANTS):
            asg, cnt, occ = fs.SearchByProjectionKeyFrame([qk], [occ_k], th=10.0, ORBdist=orb_dist, checkOri=bool(chk))
            key = "pk_%d_%d" % (seed, k)
            assert cnt[0] == int(g[key + "_n"]) and (asg[0, :n] == g[key + "_asg"]).all() and (occ[0, :n] == g[key + "_occ"]).all(), key


def test_emu_reproduces_reference_orbmatcher(plslam, synth, emu_lib):
    """The HIP sources compiled for the host emulator (tests/hipemu), through the C ABI."""
    _check_device(plslam, synth, emu_lib)


@pytest.mark.gpu
def test_gpu_reproduces_reference_orbmatcher(plslam, synth):
    _check_device(plslam, synth, None)
