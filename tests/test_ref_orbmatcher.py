"""Pinning the ORB matcher against the REFERENCE's own code.  oracle/_ref/libmatcher_ref.so is the reference's
src/ORBmatcher.cc -- every function -- compiled from the source where it lies (oracle/ref/build_ref.sh).  The SLAM object
model it is written against (Frame, KeyFrame, MapPoint pull in the whole system) is replaced by stand-ins holding just the
members the matcher touches (oracle/ref/slam_stub.h, filled from flat arrays by oracle/ref/ref_matcher.cc); the grid
lookup behind Frame::GetFeaturesInArea is the oracle's.  Entry points exist for the searches without pose algebra:

    SearchByBoW(KeyFrame*, Frame&)   SearchByBoW(KeyFrame*, KeyFrame*)   SearchForInitialization
    SearchByProjection(Frame&, vpMapPoints, th)          (+ ComputeThreeMaxima, DescriptorDistance underneath)

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


@pytest.mark.gpu
def test_gpu_reproduces_reference_orbmatcher(plslam, synth):
    G = _gen()
    TM, TF = _mods(G)
    g = np.load(GOLDEN)
    P = plslam
    for seed, n, nodes, nn, chk in G.BOW_CASES:
        kf, fr = G.bow_inputs(synth, TM, seed, n, nodes)
        got, cnt = P.ORBmatcher(nn, bool(chk)).SearchByBoWBatch([kf], [fr])
        assert cnt[0] == int(g["bow_%d_n" % seed]) and (got[0, :n] == g["bow_%d_m" % seed]).all(), "SearchByBoW %d" % seed
    for seed, n, dist in G.FRAME_CASES:
        f1, f2, gp, q, occ0 = G.frame_inputs(synth, P, TF, seed, n, dist)
        fs = P.FrameSearch(gp, TF.SCALE, [f2])
        prev = np.stack([f1["kps"]["x"], f1["kps"]["y"]], 1).astype(np.float32)
        m12, cnt, pm = fs.SearchForInitialization([f1], [prev], 100, 0.9, True)
        assert cnt[0] == int(g["init_%d_n" % seed]) and (m12[0, :n] == g["init_%d_m" % seed]).all()
        assert (pm[0, :n] == g["init_%d_prev" % seed]).all()
        asg, cnt, occ = fs.SearchByProjectionMapPoints([q], [occ0], th=3.0, nnratio=0.8)
        assert cnt[0] == int(g["proj_%d_n" % seed]) and (asg[0, :n] == g["proj_%d_asg" % seed]).all()
        assert (occ[0, :n] == g["proj_%d_occ" % seed]).all()
