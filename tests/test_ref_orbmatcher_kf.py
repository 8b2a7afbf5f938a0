"""Pinning the KeyFrame-side ORB searches against the REFERENCE's own code (second part of tests/test_ref_orbmatcher.py;
same library, oracle/_ref/libmatcher_ref.so = the reference's src/ORBmatcher.cc compiled from where it lies):

    Fuse(pKF, vpMapPoints, th)                       Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)
    SearchByProjection(pKF, Scw, vpPoints, vpMatched, th)          SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)
    SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, false)  (+ CheckDistEpipolarLine)

These functions transform map points with the keyframe pose / Sim3 before they search.  The harness
(oracle/ref/ref_matcher.cc) drives them with identity poses -- R*x+t then returns x exactly, nothing rests on how the
stand-in's float algebra rounds -- and hands back the projections its copy of the reference's expressions produced
(`1/z` in float or `1.0/z` in double, as each function writes it), which are the q_uv of the flat-array searches.  For
the two Fuse overloads the keypoint each query settled on is read back from the KeyFrame::GetMapPoint(bestIdx) call the
function makes; the replace / add bookkeeping on the map stays with the caller, as in the product API.  The Sim3 overload
of Fuse has no chi-square gate: it is plh_orb_fuse_search with inv_level_sigma2 = 0.

tools/gen_golden_ref.py committed the reference outputs as tests/golden/ref_orbmatcher_kf.npz; the oracle (CPU), the HIP
sources on the host emulator (CPU) and the GPU kernels (`-m gpu`) must reproduce them; in the build container the
reference is also run live on further inputs."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLDEN = os.path.join(_util.ROOT, "tests", "golden", "ref_orbmatcher_kf.npz")
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "libmatcher_ref.so")
V, I, F = C.c_void_p, C.c_int, C.c_float


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _olib(O, TF):
    L = TF._olib(O)
    L.plo_orb_fuse_search.argtypes = [V, V, I, V, V, V, V, V, I, V, V, V, V, F, I, V]
    L.plo_orb_search_by_projection_sim3.argtypes = [V, V, I, V, V, V, V, V, I, V, V, V, V, F, I, V]
    L.plo_orb_search_by_sim3.argtypes = [V, V, I, V, V, V, V, I, V, V, V, V, V, V, V, V, V, V, V, V, F, I, V, V, V]
    L.plo_orb_search_for_triangulation.argtypes = [V, V, V, V, I, V, V, V, V, I, V, F, F, V, V, I, I, V]
    for f in ("plo_orb_fuse_search", "plo_orb_search_by_projection_sim3", "plo_orb_search_by_sim3", "plo_orb_search_for_triangulation"):
        getattr(L, f).restype = I
    return L


class _Oracle:
    """The oracle's flat-array searches behind the same call shapes as the device wrapper below."""

    def __init__(self, O, P, TF):
        self.O, self.P, self.TF, self.L = O, P, TF, _olib(O, TF)

    def fuse(self, f2, gp, q, uv, valid, inv, th):
        O, n = self.O, len(f2["kps"])
        (cs, ci), _ = self.TF._oracle_grids(O, self.P, f2, gp)
        rb = np.zeros(max(n, 1), np.int32)
        rc = self.L.plo_orb_fuse_search(O._p(f2["kps"]), O._p(f2["desc"]), n, O._p(self.P._gp_array(gp)), O._p(cs), O._p(ci), O._p(self.TF.SCALE),
                                        O._p(inv), n, O._p(valid), O._p(uv), O._p(q["octave"]), O._p(q["desc"]), th, 50, O._p(rb))
        return rc, rb[:n]

    def s3p(self, f2, gp, q, uv, valid, occ_in, th):
        O, n = self.O, len(f2["kps"])
        (cs, ci), _ = self.TF._oracle_grids(O, self.P, f2, gp)
        ro, ra = occ_in.copy(), np.zeros(max(n, 1), np.int32)
        rc = self.L.plo_orb_search_by_projection_sim3(O._p(f2["kps"]), O._p(f2["desc"]), n, O._p(self.P._gp_array(gp)), O._p(cs), O._p(ci),
                                                      O._p(self.TF.SCALE), O._p(ro), n, O._p(valid), O._p(uv), O._p(q["octave"]),
                                                      O._p(q["desc"]), float(th), 50, O._p(ra))
        return rc, ra[:n], ro

    def sim3(self, f1, f2, gp, sides, uv12, v12, uv21, v21, th):
        O, n = self.O, len(f1["kps"])
        (cs1, ci1), _ = self.TF._oracle_grids(O, self.P, f1, gp)
        (cs2, ci2), _ = self.TF._oracle_grids(O, self.P, f2, gp)
        s1, s2 = sides
        r1, r2, r12 = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        rc = self.L.plo_orb_search_by_sim3(O._p(f1["kps"]), O._p(f1["desc"]), n, O._p(cs1), O._p(ci1), O._p(f2["kps"]), O._p(f2["desc"]), n,
                                           O._p(cs2), O._p(ci2), O._p(self.P._gp_array(gp)), O._p(self.TF.SCALE), O._p(v12), O._p(uv12),
                                           O._p(s1["level"]), O._p(s1["desc"]), O._p(v21), O._p(uv21), O._p(s2["level"]), O._p(s2["desc"]),
                                           th, 100, O._p(r1), O._p(r2), O._p(r12))
        return rc, r12[:n]

    def tri(self, a, b, F12, epi, sf, sig2, chk):
        O, n = self.O, len(a["desc"])
        ref = np.zeros(max(n, 1), np.int32)
        rc = self.L.plo_orb_search_for_triangulation(O._p(a["kps"]), O._p(a["desc"]), O._p(a["node"]), O._p(a["has_mp"]), n, O._p(b["kps"]),
                                                     O._p(b["desc"]), O._p(b["node"]), O._p(b["has_mp"]), n, O._p(F12), float(epi[0]),
                                                     float(epi[1]), O._p(sf), O._p(sig2), 50, chk, O._p(ref))
        return rc, ref[:n]


class _Device:
    """The product's C ABI (GPU library, or the HIP sources on the host emulator)."""

    def __init__(self, P, TF, lib):
        self.P, self.TF, self.lib = P, TF, lib

    def fuse(self, f2, gp, q, uv, valid, inv, th):
        n = len(f2["kps"])
        fs = self.P.FrameSearch(gp, self.TF.SCALE, [f2], lib=self.lib)
        best, nf = fs.FuseSearch([dict(valid=valid, uv=uv, level=q["octave"], desc=q["desc"])], inv, th=th)
        return nf[0], best[0, :n]

    def s3p(self, f2, gp, q, uv, valid, occ_in, th):
        n = len(f2["kps"])
        fs = self.P.FrameSearch(gp, self.TF.SCALE, [f2], lib=self.lib)
        qd = dict(valid=valid, uv=uv, level=q["octave"], desc=q["desc"], hasobs=np.ones(len(valid), np.uint8))
        asg, cnt, occ = fs.SearchByProjectionSim3([qd], [occ_in], th=th)
        return cnt[0], asg[0, :n], occ[0, :max(n, 1)]

    def sim3(self, f1, f2, gp, sides, uv12, v12, uv21, v21, th):
        n = len(f1["kps"])
        fs1 = self.P.FrameSearch(gp, self.TF.SCALE, [f1], lib=self.lib)
        fs2 = self.P.FrameSearch(gp, self.TF.SCALE, [f2], lib=self.lib, cap=fs1.cap)
        s1, s2 = sides
        m12, nf, _, _ = fs1.SearchBySim3(fs2, [dict(valid=v12, uv=uv12, level=s1["level"], desc=s1["desc"])],
                                         [dict(valid=v21, uv=uv21, level=s2["level"], desc=s2["desc"])], th=th)
        return nf[0], m12[0, :n]

    def tri(self, a, b, F12, epi, sf, sig2, chk):
        n = len(a["desc"])
        got, cnt = self.P.ORBmatcher(0.6, bool(chk), lib=self.lib).SearchForTriangulationBatch([a], [b], F12, (float(epi[0]), float(epi[1])),
                                                                                              sf, sig2)
        return cnt[0], got[0, :n]


def _check_against_golden(X, G, P, S):
    TM, TF = G._test_module("test_match"), G._test_module("test_frame_search")
    g = np.load(GOLDEN)
    c_ = np.ascontiguousarray
    inv = (np.float32(1.0) / (TF.SCALE * TF.SCALE)).astype(np.float32)
    for seed, n, dist in G.POSE_CASES:
        f2, gp, q, xyz, fl, occ_k, kfmp, slot = G.kf_inputs(S, P, TF, seed, n, dist)
        for k, th in enumerate(G.KF_TH["fuse"]):
            rc, rb = X.fuse(f2, gp, q, c_(g["fuse_%d_uv" % seed]), c_(g["fuse_%d_valid" % seed]), inv, th)
            assert rc == int(g["fuse_%d_%d_n" % (seed, k)]) and (rb == g["fuse_%d_%d_best" % (seed, k)]).all(), "Fuse %d %g" % (seed, th)
        for k, th in enumerate(G.KF_TH["fuse3"]):
            rc, rb = X.fuse(f2, gp, q, c_(g["fuse3_%d_uv" % seed]), c_(g["fuse3_%d_valid" % seed]), np.zeros_like(inv), th)
            assert rc == int(g["fuse3_%d_%d_n" % (seed, k)]) and (rb == g["fuse3_%d_%d_best" % (seed, k)]).all(), "Fuse/Sim3 %d %g" % (seed, th)
        for k, th in enumerate(G.KF_TH["s3p"]):
            rc, ra, ro = X.s3p(f2, gp, q, c_(g["s3p_%d_uv" % seed]), c_(g["s3p_%d_valid" % seed]), c_(g["s3p_%d_occin" % seed]), th)
            key = "s3p_%d_%d" % (seed, k)
            assert rc == int(g[key + "_n"]) and (ra == g[key + "_asg"]).all() and (ro == g[key + "_occ"]).all(), key
        f1, f2, gp, sides, already = G.sim3_inputs(S, P, TF, seed, n, dist)
        for k, th in enumerate(G.KF_TH["sim3"]):
            rc, r12 = X.sim3(f1, f2, gp, sides, c_(g["sim3_%d_uv12" % seed]), c_(g["sim3_%d_v12" % seed]), c_(g["sim3_%d_uv21" % seed]),
                             c_(g["sim3_%d_v21" % seed]), th)
            assert rc == int(g["sim3_%d_%d_n" % (seed, k)]) and (r12 == g["sim3_%d_%d_m12" % (seed, k)]).all(), "SearchBySim3 %d %g" % (seed, th)
    sig2 = (G.TRI_SF * G.TRI_SF).astype(np.float32)
    for seed, n, nodes in G.TRI_CASES:
        a, b = TM._tri_case(P, S, seed, n, nodes)
        for k, (F12, cw, chk) in enumerate(G.TRI_VARIANTS):
            rc, m = X.tri(a, b, F12, g["tri_%d_epi" % k], G.TRI_SF, sig2, chk)
            assert rc == int(g["tri_%d_%d_n" % (seed, k)]) and (m == g["tri_%d_%d_m" % (seed, k)]).all(), "SearchForTriangulation %d %d" % (seed, k)


def test_golden_file_present():
    assert os.path.exists(GOLDEN)


def test_oracle_reproduces_reference_keyframe_searches(oracle, plslam, synth):
    G = _gen()
    _check_against_golden(_Oracle(oracle, plslam, G._test_module("test_frame_search")), G, plslam, synth)


def test_emu_reproduces_reference_keyframe_searches(plslam, synth, emu_lib):
    G = _gen()
    _check_against_golden(_Device(plslam, G._test_module("test_frame_search"), emu_lib), G, plslam, synth)


@pytest.mark.gpu
def test_gpu_reproduces_reference_keyframe_searches(plslam, synth):
    G = _gen()
    _check_against_golden(_Device(plslam, G._test_module("test_frame_search"), None), G, plslam, synth)


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_keyframe_searches_live(oracle, plslam, synth):
    G = _gen()
    TM, TF = G._test_module("test_match"), G._test_module("test_frame_search")
    R, X, P, S = G.ref_matcher_lib(), _Oracle(oracle, plslam, TF), plslam, synth
    inv = (np.float32(1.0) / (TF.SCALE * TF.SCALE)).astype(np.float32)
    for seed, n, dist in [(61, 1500, True), (62, 333, False), (63, 0, False)]:
        f2, gp, q, xyz, fl, occ_k, kfmp, slot = G.kf_inputs(S, P, TF, seed, n, dist)
        c, best, uv, valid = G.reference_fuse(R, P, TF, f2, gp, q, xyz, fl, kfmp, 4.0)
        rc, rb = X.fuse(f2, gp, q, uv, valid, inv, 4.0)
        assert c == rc and (best == rb).all(), "live Fuse %d" % seed
        c, best, uv, valid = G.reference_fuse3(R, P, TF, f2, gp, q, xyz, fl, kfmp, slot, 4.0)
        rc, rb = X.fuse(f2, gp, q, uv, valid, np.zeros_like(inv), 4.0)
        assert c == rc and (best == rb).all(), "live Fuse/Sim3 %d" % seed
        c, asg, occ, uv, valid, occ_in = G.reference_s3p(R, P, TF, f2, gp, q, xyz, fl, occ_k, slot, 7)
        rc, ra, ro = X.s3p(f2, gp, q, uv, valid, occ_in, 7)
        assert c == rc and (asg == ra).all() and (occ == ro).all(), "live SearchByProjection(pKF, Scw) %d" % seed
        f1, f2, gp, sides, already = G.sim3_inputs(S, P, TF, seed, n, dist)
        c, m12, uv12, v12, uv21, v21 = G.reference_sim3(R, P, TF, f1, f2, gp, sides, already, 5.0)
        rc, r12 = X.sim3(f1, f2, gp, sides, uv12, v12, uv21, v21, 5.0)
        assert c == rc and (m12 == r12).all(), "live SearchBySim3 %d" % seed
    sig2 = (G.TRI_SF * G.TRI_SF).astype(np.float32)
    for seed, n, nodes in [(520, 900, 50), (521, 64, 64)]:
        a, b = TM._tri_case(P, S, seed, n, nodes)
        for F12, cw, chk in G.TRI_VARIANTS:
            c, m, epi = G.reference_tri(R, a, b, F12, cw, chk)
            rc, ref = X.tri(a, b, F12, epi, G.TRI_SF, sig2, chk)
            assert c == rc and (m == ref).all(), "live SearchForTriangulation %d" % seed
