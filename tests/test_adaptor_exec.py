"""The C++ drop-in boundary EXECUTED: the reference's own src/Frame.cc, compiled from where it lies with the adaptor
headers of pl-slam_amd/adaptor ahead of the reference's include directory (oracle/ref/build_adaptor.sh),
linked to libplslam_hip.so (GPU box, `-m gpu`) or to the emulator build of the same kernel sources (CPU suite).

  * test_adaptor_executes_frame_constructor: `Frame(imGray, t, ORBextractor*, LINEextractor*, voc, K, distCoef, bf, thDepth, mask)`
    (src/Frame.cc:193-276) runs unchanged -- Frame::ExtractORB and Frame::ExtractLSD on two threads call the adaptor classes -- and
    its members mvKeys, mDescriptors, mvKeylinesUn, mLdesc, mvKeyLineFunctions equal what the reference's own ORBextractor.cc /
    LineExtractor.cpp produced on the same image (tests/golden/ref_orb_*.npz, ref_line_*.npz); mvKeysUn, the image bounds and both
    grids equal the oracle's.
  * test_adaptor_executes_tracking_searches: the real ORBmatcher / LSDmatcher call sites of Tracking (SearchLocalPoints /
    SearchLocalLines, TrackWithMotionModel, TrackReferenceKeyFrame) on real Frame / KeyFrame / MapPoint / MapLine objects, with
    `ORBmatcher` / `LSDmatcher` now being the adaptor classes; results equal tests/golden/ref_track.npz, which the reference's own
    src/ORBmatcher.cc / src/LSDmatcher.cpp produced.
  * test_adaptor_executes_compute_bow: Frame::ComputeBoW / KeyFrame::ComputeBoW with `ORBVocabulary` being the drop-in class
    (adaptor/ORBVocabulary.h: the reference's loaders, the GPU's transform); BowVector doubles and FeatureVector lists equal what the
    same methods leave over the reference's own DBoW2 (tests/golden/ref_computebow.npz).
  * test_adaptor_executes_initialization_matchers: SearchForInitialization and SearchDouble on two constructed Frames vs the oracle.
The libraries are built in the build container (the reference is not on the GPU box) and travel with the snapshot."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLD = os.path.join(_util.ROOT, "tests", "golden")
LIB_HIP = os.path.join(_util.ROOT, "oracle", "_ref", "libadaptor_hip.so")
LIB_EMU = os.path.join(_util.ROOT, "oracle", "_ref", "libadaptor_emu.so")
V, I, F = C.c_void_p, C.c_int, C.c_float


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _lib(path):
    if not os.path.exists(path):
        pytest.skip("%s not built (oracle/ref/build_adaptor.sh needs /root/reference)" % os.path.basename(path))
    G = _gen()
    if path == LIB_HIP:
        # the product library first, through the mirror: it imports torch so that the process holds ONE HIP runtime
        # (libadaptor_hip.so would otherwise pull /opt/rocm's in before PyTorch brings its own)
        _util.plslam().load()
    R = G.ref_frame_lib(path)
    R.adx_tracker_create.restype = V
    R.adx_tracker_create.argtypes = [I, F, I, I, I, I, C.c_double]
    R.adx_tracker_destroy.argtypes = [V]
    R.adx_tracker_set_refine.argtypes = [V, I]
    R.adx_frame_create.restype = V
    R.adx_frame_create.argtypes = [V, V, I, I, V, V, V, C.c_char_p, I]
    R.adx_frame_destroy.argtypes = [V]
    R.adx_frame_counts.argtypes = [V, V, V]
    R.adx_frame_read.argtypes = [V] * 8
    R.adx_search_for_initialization.argtypes = [V, V, V, I, F, I, V]
    R.adx_search_double.argtypes = [V, V, F, V]
    R.adx_descriptor_distance.argtypes = [V, V]
    assert R.adx_uses_adaptor_classes() == 1
    return G, R


class _Frame:
    def __init__(self, R, P, tracker, img, K, D, mask=None):
        self.R = R
        img = np.ascontiguousarray(img)
        err = C.create_string_buffer(512)
        Kf, Df = np.asarray(K, np.float32), np.asarray(D, np.float32)
        mp = mask.ctypes.data_as(V) if mask is not None else None
        self.h = R.adx_frame_create(tracker, img.ctypes.data_as(V), img.shape[0], img.shape[1], Kf.ctypes.data_as(V), Df.ctypes.data_as(V),
                                    mp, err, 512)
        assert self.h, "Frame constructor threw: " + err.value.decode()
        n, nl = C.c_int(0), C.c_int(0)
        R.adx_frame_counts(self.h, C.byref(n), C.byref(nl))
        self.N, self.NL = n.value, nl.value
        self.keys, self.keys_un = np.zeros(max(self.N, 1), P.KP_DTYPE), np.zeros(max(self.N, 1), P.KP_DTYPE)
        self.desc = np.zeros((max(self.N, 1), 32), np.uint8)
        self.kl, self.ldesc = np.zeros(max(self.NL, 1), P.KL_DTYPE), np.zeros((max(self.NL, 1), 32), np.uint8)
        self.fn, self.bounds = np.zeros((max(self.NL, 1), 3)), np.zeros(6, np.float32)
        p = lambda a: a.ctypes.data_as(V)
        R.adx_frame_read(self.h, p(self.keys), p(self.keys_un), p(self.desc), p(self.kl), p(self.ldesc), p(self.fn), p(self.bounds))
        for a in ("keys", "keys_un", "desc"):
            setattr(self, a, getattr(self, a)[:self.N])
        for a in ("kl", "ldesc", "fn"):
            setattr(self, a, getattr(self, a)[:self.NL])

    def grids(self):
        R = self.R
        cs, ci = np.zeros(64 * 48 + 1, np.int32), np.zeros(max(self.N, 1), np.int32)
        R.ref_frame_grid_points(self.h, cs.ctypes.data_as(V), ci.ctypes.data_as(V), len(ci))
        lcs, lci = np.zeros(64 * 48 + 1, np.int32), np.zeros(max(self.NL, 1) * 64, np.int32)
        R.ref_frame_grid_lines(self.h, lcs.ctypes.data_as(V), lci.ctypes.data_as(V), len(lci))
        return (cs, ci), (lcs, lci)

    def close(self):
        if self.h:
            self.R.adx_frame_destroy(self.h)
            self.h = None


def _ulps(a, b):
    ia, ib = a.astype(np.float32).view(np.int32).astype(np.int64), b.astype(np.float32).view(np.int32).astype(np.int64)
    ia, ib = np.where(ia < 0, -(ia & 0x7fffffff), ia), np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def _check_frame_vs_reference(fr, go, gl, what, exact_angle=False):
    """mvKeys / mDescriptors vs the reference's ORBextractor.cc, mvKeylinesUn / mLdesc / mvKeyLineFunctions vs its LineExtractor.cpp
    (or vs the oracle: exact_angle)."""
    assert fr.N == len(go["kps"]), "%s: %d keypoints, reference %d" % (what, fr.N, len(go["kps"]))
    for f in go["kps"].dtype.names:
        assert (fr.keys[f] == go["kps"][f]).all(), "%s: mvKeys.%s" % (what, f)
    assert (fr.desc == go["desc"]).all(), what + ": mDescriptors"
    assert fr.NL == len(gl["keylines"]), "%s: %d keylines, reference %d" % (what, fr.NL, len(gl["keylines"]))
    for f in gl["keylines"].dtype.names:
        if f == "angle" and not exact_angle:   # toolchain-dependent atan2 overload inside the reference (see tests/test_ref_line.py)
            assert _ulps(fr.kl[f], gl["keylines"][f]).max(initial=0) <= 1
        else:
            assert (fr.kl[f] == gl["keylines"][f]).all(), "%s: mvKeylinesUn.%s" % (what, f)
    assert (fr.ldesc == gl["desc"]).all(), what + ": mLdesc"
    assert (fr.fn == gl["linefn"]).all(), what + ": mvKeyLineFunctions"


def _frame_constructor(P, S, O, path, cases):
    G, R = _lib(path)
    TF = G._test_module("test_frame_search")
    for orb_name, line_name, K in cases:
        go, gl = np.load(os.path.join(GOLD, "ref_orb_%s.npz" % orb_name)), np.load(os.path.join(GOLD, "ref_line_%s.npz" % line_name))
        rows, cols, seed = int(go["rows"]), int(go["cols"]), int(go["seed"])
        assert (int(gl["rows"]), int(gl["cols"]), int(gl["seed"])) == (rows, cols, seed) and not bool(gl["masked"])
        img = S.make_frame(seed, rows, cols)
        trk = R.adx_tracker_create(int(go["nfeatures"]), float(go["scale"]), int(go["nlevels"]), int(go["ini"]), int(go["mn"]),
                                   int(gl["nfeatures"]), float(gl["min_len"]))
        try:
            # --- no distortion (KITTI-style calibration): LSD sees the image itself.  The harness is built as INTEGRATION.md section 2
            # tells a maintainer to (-DPLH_LSD_REFINE_DEFAULT=1: LSD_REFINE_ADV, what the reference's linked opencv_contrib runs): the
            # lines of the Frame equal the oracle's at that level; with SetRefine(LSD_REFINE_STD) they equal what the reference's own
            # LineExtractor.cpp + the (un-linked, STD) twin in its tree produced (ref_line_*.npz).  ORB: the reference's ORBextractor.cc.
            for refine in (1, 0):
                R.adx_tracker_set_refine(trk, refine)
                if refine:
                    ak, ad, af = O.line_extract(img, int(gl["nfeatures"]), float(gl["min_len"]), refine=1)
                    gla = dict(keylines=ak, desc=ad, linefn=af)
                for rep in range(2):   # the extractor objects persist across frames like Tracking's
                    fr = _Frame(R, P, trk, img, K, [0, 0, 0, 0, 0])
                    _check_frame_vs_reference(fr, go, gla if refine else gl, "%s (frame %d, refine %d)" % (orb_name, rep, refine),
                                              exact_angle=bool(refine))
                    assert all((fr.keys_un[f] == fr.keys[f]).all() for f in fr.keys.dtype.names)      # Frame.cc:917-921
                    assert tuple(fr.bounds[:4]) == (0.0, 0.0, float(cols), float(rows))
                    gp = P.grid_params(cols, rows)
                    assert fr.bounds[4] == gp.inv_w and fr.bounds[5] == gp.inv_h
                    (cs, ci), (lcs, lci) = fr.grids()
                    (rcs, rci), (rlcs, rlci) = TF._oracle_grids(O, P, dict(kps=fr.keys_un, keylines=fr.kl), gp)
                    assert (cs == rcs).all() and (ci[:rcs[-1]] == rci[:rcs[-1]]).all(), "mGrid"
                    assert (lcs == rlcs).all() and (lci[:rlcs[-1]] == rlci[:rlcs[-1]]).all(), "mGridForLine"
                    fr.close()
            R.adx_tracker_set_refine(trk, 1)   # back to the build's level for the rest
            # --- with distortion: ORB still runs on the raw image (Frame.cc:224), LSD on the remapped one (:221-225)
            D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]       # TUM1.yaml:13-17
            fr = _Frame(R, P, trk, img, K, D)
            assert fr.N == len(go["kps"]) and (fr.desc == go["desc"]).all()
            assert all((fr.keys[f] == go["kps"][f]).all() for f in go["kps"].dtype.names)
            Kf, Df = np.asarray(K, np.float32), np.asarray(D, np.float32)
            un = np.zeros(fr.N, P.KP_DTYPE)
            O.lib().plo_undistort_keypoints.argtypes = [V, I, V, V, V]
            O.lib().plo_undistort_keypoints(O._p(np.ascontiguousarray(fr.keys)), fr.N, O._p(Kf), O._p(Df), O._p(un))
            assert all((fr.keys_un[f] == un[f]).all() for f in un.dtype.names), "mvKeysUn"
            mx, my = np.zeros((rows, cols), np.float32), np.zeros((rows, cols), np.float32)
            O.lib().plo_undistort_maps(O._p(Kf), O._p(Df), cols, rows, O._p(mx), O._p(my))
            und = np.zeros_like(img)
            O.lib().plo_remap_linear_u8(O._p(img), cols, rows, cols, O._p(mx), O._p(my), O._p(und), cols)
            rk, rd, rf = O.line_extract(und, int(gl["nfeatures"]), float(gl["min_len"]))   # (the build's level: LSD_REFINE_ADV)
            assert fr.NL == len(rk) and (fr.ldesc == rd).all() and (fr.fn == rf).all()
            assert all((fr.kl[f] == rk[f]).all() for f in rk.dtype.names), "mvKeylinesUn on the undistorted image"
            # Frame::ComputeImageBounds (Frame.cc:947-974): the undistorted image corners
            cn = np.zeros(4, P.KP_DTYPE)
            cn["x"], cn["y"] = [0, cols, 0, cols], [0, 0, rows, rows]
            cu = np.zeros(4, P.KP_DTYPE)
            O.lib().plo_undistort_keypoints(O._p(cn), 4, O._p(Kf), O._p(Df), O._p(cu))
            want = (min(cu["x"][0], cu["x"][2]), min(cu["y"][0], cu["y"][1]), max(cu["x"][1], cu["x"][3]), max(cu["y"][2], cu["y"][3]))
            assert tuple(fr.bounds[:4]) == tuple(np.float32(w) for w in want) and tuple(fr.bounds[:4]) != (0.0, 0.0, float(cols), float(rows))
            fr.close()
        finally:
            R.adx_tracker_destroy(trk)


SMALL = [("s3_320x240", "s3_320x240_minlen", [258.6, 258.2, 159.3, 127.6])]
FULL = SMALL + [("s1_640x480", "s1_640x480", [517.306408, 516.469215, 318.643040, 255.313989]),
                ("kitti_1241x376", "kitti_1241x376", [718.856, 718.856, 607.1928, 185.2157])]


def test_adaptor_executes_frame_constructor_emu(plslam, synth, oracle, emu_lib):
    _frame_constructor(plslam, synth, oracle, LIB_EMU, SMALL)


@pytest.mark.gpu
def test_adaptor_executes_frame_constructor(plslam, synth, oracle):
    _frame_constructor(plslam, synth, oracle, LIB_HIP, FULL)


def _tracking_searches(P, S, path, track_cases, bow_cases):
    G, R = _lib(path)
    TF = G._test_module("test_frame_search")
    VM = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    g = np.load(os.path.join(GOLD, "ref_track.npz"))
    for seed, n, nl, dist, th in track_cases:
        f2, gp, view, nlv, pts, lns, occ_p, occ_l = G.track_inputs(S, P, TF, seed, n, nl, dist)
        (cp, ap, op), (cl, al, ol) = G.reference_track(R, P, TF, f2, gp, view, nlv, pts, lns, occ_p, occ_l, th)
        assert cp == int(g["p_%d_n" % seed]) and (ap == g["p_%d_asg" % seed]).all() and (op == g["p_%d_occ" % seed]).all(), \
            "ORBmatcher::SearchByProjection(F, MapPoints) case %d" % seed
        assert cl == int(g["l_%d_n" % seed]) and (al == g["l_%d_asg" % seed]).all() and (ol == g["l_%d_occ" % seed]).all(), \
            "LSDmatcher::SearchByProjection(F, MapLines) case %d" % seed
        flags, q = G.track_last_inputs(S, P, TF, seed, n, nl, dist)
        c, a, o = G.reference_track_last(R, P, TF, f2, gp, view, nlv, pts, flags, q, occ_p, 15.0 if seed != 2 else 7.0)
        assert c == int(g["m_%d_n" % seed]) and (a == g["m_%d_asg" % seed]).all() and (o == g["m_%d_occ" % seed]).all(), \
            "ORBmatcher::SearchByProjection(Cur, Last) case %d" % seed
    import tempfile
    for seed, k, Lv, n, nn, chk in bow_cases:
        voc, kf, fr = G.bowtrack_inputs(S, P, VM, seed, k, Lv, n)
        c, m = G.reference_bowtrack(R, voc, kf, fr, nn, chk, tempfile.gettempdir())
        assert c == int(g["b_%d_n" % seed]) and (m == g["b_%d_m" % seed]).all(), "ORBmatcher::SearchByBoW(KF, F) case %d" % seed


def test_adaptor_executes_tracking_searches_emu(plslam, synth, emu_lib):
    G = _gen()
    _tracking_searches(plslam, synth, LIB_EMU, [c for c in G.TRACK_CASES if c[1] <= 800], [c for c in G.BOWTRACK_CASES if c[3] <= 400])


@pytest.mark.gpu
def test_adaptor_executes_tracking_searches(plslam, synth):
    G = _gen()
    _tracking_searches(plslam, synth, LIB_HIP, G.TRACK_CASES, G.BOWTRACK_CASES)


def _compute_bow(S, path, cases, live_reference):
    """Frame::ComputeBoW and KeyFrame::ComputeBoW (src/Frame.cc:906-913, src/KeyFrame.cc:76-83) as the reference compiled them, with
    `ORBVocabulary` being the product's drop-in class (pl-slam_amd/adaptor/ORBVocabulary.h: the reference's loaders build the host
    tree, transform() descends on the device): mBowVec -- words and DOUBLES, `==` -- and mFeatVec -- node ids and per-node feature lists
    in order -- equal what the same two methods leave over the reference's own DBoW2 (tests/golden/ref_computebow.npz, made by
    libframe_ref.so; with /root/reference built here that library is also run live beside it)."""
    import tempfile
    G, R = _lib(path)
    VM = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    g = np.load(os.path.join(GOLD, "ref_computebow.npz"))
    ref_path = os.path.join(_util.ROOT, "oracle", "_ref", "libframe_ref.so")
    Rref = G.ref_frame_lib() if (live_reference and os.path.exists(ref_path)) else None
    for seed, k, Lv, stop, idf, n, binary, sc, wt in cases:
        voc, desc = G.computebow_inputs(S, VM, seed, k, Lv, stop, idf, n)
        bw, bv, fn, ff, eq, adp = G.reference_computebow(R, voc, desc, binary, sc, wt, tempfile.gettempdir())
        assert adp == 1, "ORBVocabulary in this library is not the drop-in class"
        what = "ComputeBoW case %d" % seed
        assert len(bw) == len(g["w_%d" % seed]) and (bw == g["w_%d" % seed]).all(), what + ": BowVector words"
        assert (bv == g["v_%d" % seed]).all(), what + ": BowVector values (doubles compared with ==)"
        assert len(ff) == len(g["ff_%d" % seed]) and (fn == g["fn_%d" % seed]).all() and (ff == g["ff_%d" % seed]).all(), what + ": FeatureVector"
        assert eq == 1, what + ": KeyFrame::ComputeBoW / a copied vocabulary disagree with Frame::ComputeBoW"
        if Rref is not None:
            rw, rv, rn, rf, req, radp = G.reference_computebow(Rref, voc, desc, binary, sc, wt, tempfile.gettempdir())
            assert radp == 0 and req == 1
            assert (rw == bw).all() and (rv == bv).all() and (rn == fn).all() and (rf == ff).all(), what + ": live reference"


def test_adaptor_executes_compute_bow_emu(synth, emu_lib):
    G = _gen()
    _compute_bow(synth, LIB_EMU, [c for c in G.COMPUTEBOW_CASES if c[5] <= 900], True)


@pytest.mark.gpu
def test_adaptor_executes_compute_bow(synth):
    G = _gen()
    _compute_bow(synth, LIB_HIP, G.COMPUTEBOW_CASES, False)


def _initialization_matchers(P, S, O, path, rows, cols, nfeat):
    G, R = _lib(path)
    TF = G._test_module("test_frame_search")
    L = TF._olib(O)
    K = [0.8 * cols, 0.8 * cols, cols / 2.0 - 0.7, rows / 2.0 + 0.4]
    frames = S.make_frames(900, 2, rows, cols, unique=1)              # frame 1 = frame 0 shifted by 3 rows, other exposure
    trk = R.adx_tracker_create(nfeat, 1.2, 4, 20, 7, 80, 0.0)
    f1 = _Frame(R, P, trk, frames[0], K, [0, 0, 0, 0, 0])
    f2 = _Frame(R, P, trk, frames[1], K, [0, 0, 0, 0, 0])
    try:
        assert f1.N > 100 and f2.N > 100 and f1.NL > 10 and f2.NL > 10
        # ORBmatcher(0.9, true).SearchForInitialization(F1, F2, prev, matches, 100)   (Tracking.cc:706-708)
        prev = np.ascontiguousarray(np.stack([f1.keys_un["x"], f1.keys_un["y"]], 1).astype(np.float32))
        m12 = np.zeros(f1.N, np.int32)
        pm = prev.copy()
        c = R.adx_search_for_initialization(f1.h, f2.h, pm.ctypes.data_as(V), 100, 0.9, 1, m12.ctypes.data_as(V))
        gp = P.grid_params(cols, rows)
        g = TF._gpa(P, gp)
        (rcs, rci), _ = TF._oracle_grids(O, P, dict(kps=f2.keys_un, keylines=f2.kl), gp)
        rp, ref = prev.copy(), np.zeros(f1.N, np.int32)
        k1, k2 = np.ascontiguousarray(f1.keys_un), np.ascontiguousarray(f2.keys_un)
        d1, d2 = np.ascontiguousarray(f1.desc), np.ascontiguousarray(f2.desc)
        rc = L.plo_orb_search_for_initialization(O._p(k1), O._p(d1), f1.N, O._p(k2), O._p(d2), f2.N, O._p(g), O._p(rcs), O._p(rci), O._p(rp),
                                                 100, 0.9, 1, O._p(ref))
        assert c == rc and (m12 == ref).all() and (pm == rp).all() and rc > 20, "SearchForInitialization (%d matches)" % rc
        # LSDmatcher(0.7).SearchDouble(InitialFrame, CurrentFrame, LineMatches)      (Tracking.cc:711)
        ml = np.full(f1.NL, -7, np.int32)
        cl = R.adx_search_double(f1.h, f2.h, 0.7, ml.ctypes.data_as(V))
        rl = np.zeros(f1.NL, np.int32)
        l1, l2 = np.ascontiguousarray(f1.ldesc), np.ascontiguousarray(f2.ldesc)
        rcl = O.lib().plo_line_search_double(O._p(l1), f1.NL, O._p(l2), f2.NL, 50.0, 0.7, O._p(rl))
        assert cl == rcl and (ml == rl).all() and rcl > 3, "SearchDouble (%d matches)" % rcl
        # LSDmatcher(0.7).SerachForInitialize(InitialFrame, CurrentFrame, LineMatches)   (Tracking.cc:710, commented out): the reference's method
        # and the drop-in's on the same Frames
        o_ref, o_hip, n_ref = np.full(f1.NL, -7, np.int32), np.full(f1.NL, -9, np.int32), C.c_int(0)
        R.adx_serach_for_initialize.argtypes = [V, V, F, V, V, V]
        ci = R.adx_serach_for_initialize(f1.h, f2.h, 0.7, o_ref.ctypes.data_as(V), o_hip.ctypes.data_as(V), C.byref(n_ref))
        assert ci == n_ref.value and (o_ref == o_hip).all() and ci >= rcl and ci > 3, "SerachForInitialize (%d / %d matches)" % (ci, n_ref.value)
        # LSDmatcher(0.7).SearchByProjection(CurrentFrame, LastFrame) -- the two-argument overload (LSDmatcher.cpp:19-70, no caller): the
        # reference's method and the drop-in's, the last frame's MapLines carried over to the same lines of the current frame
        hm = (np.arange(f1.NL) % 4 != 1).astype(np.uint8)
        o_ref, o_hip = np.full(f2.NL, -7, np.int32), np.full(f2.NL, -9, np.int32)
        R.adx_line_search_by_projection_two_arg.argtypes = [V, V, V, F, V, V, V]
        cp = R.adx_line_search_by_projection_two_arg(f2.h, f1.h, hm.ctypes.data_as(V), 0.7, o_ref.ctypes.data_as(V), o_hip.ctypes.data_as(V),
                                                     C.byref(n_ref))
        assert cp == n_ref.value and (o_ref == o_hip).all() and 0 < cp <= rcl and cp == int((o_hip >= 0).sum()), "SearchByProjection(Cur, Last) (%d / %d)" % (cp, n_ref.value)
        a, b = np.ascontiguousarray(d1[0]), np.ascontiguousarray(d2[0])
        dist = int(np.unpackbits(a ^ b).sum())
        assert R.adx_descriptor_distance(a.ctypes.data_as(V), b.ctypes.data_as(V)) == dist * 1001
    finally:
        f1.close()
        f2.close()
        R.adx_tracker_destroy(trk)


def test_adaptor_executes_initialization_matchers_emu(plslam, synth, oracle, emu_lib):
    _initialization_matchers(plslam, synth, oracle, LIB_EMU, 200, 280, 400)


@pytest.mark.gpu
def test_adaptor_executes_initialization_matchers(plslam, synth, oracle):
    _initialization_matchers(plslam, synth, oracle, LIB_HIP, 480, 640, 1000)


# ------------------------------------------------------------------------------------------------------------------------------
# Back-end call sites (LoopClosing / LocalMapping) behind the class names: the adaptor overloads against the reference's own
# methods (ORBmatcherCPU = src/ORBmatcher.cc), both on real KeyFrame / MapPoint objects built from the same arrays, poses with
# rotation and scale (oracle/ref/ref_adaptor.cc: adx_loop_search_by_projection, adx_local_mapping_fuse, adx_loop_search_by_bow).
# ------------------------------------------------------------------------------------------------------------------------------
def _rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)


def _backend_scene(P, seed, n=700, npts=900, sim3_scale=1.0):
    """A KeyFrame (keypoints over a 640x480 image on 8 levels, descriptors, pose) and candidate map points: most of them sit where
    a keypoint looks (projection within a pixel or two, descriptor a few bits off, distance range that predicts the keypoint's
    level), the rest are behind the camera, outside the image, out of range, facing away, or carry a foreign descriptor."""
    rng = np.random.RandomState(seed)
    K4 = np.array([517.3, 516.5, 318.6, 255.3], np.float32)
    gp = np.array([0.0, 0.0, 640.0, 480.0, 64 / 640.0, 48 / 480.0], np.float32)
    kps = np.zeros(n, P.KP_DTYPE)
    kps["x"] = rng.uniform(5, 635, n).astype(np.float32)
    kps["y"] = rng.uniform(5, 475, n).astype(np.float32)
    kps["octave"] = rng.randint(0, 8, n)
    kps["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    kps["size"] = 31.0 * 1.2 ** kps["octave"]
    kps["response"] = rng.uniform(20, 200, n).astype(np.float32)
    kps["class_id"] = -1
    desc = rng.randint(0, 256, (n, 32)).astype(np.uint8)
    R = _rot(0.05, -0.12, 0.03)
    t = np.array([0.3, -0.2, 0.5], np.float32)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3], T[:3, 3] = R, t
    S = np.eye(4, dtype=np.float32)
    S[:3, :3], S[:3, 3] = sim3_scale * R, sim3_scale * t      # the function divides both by the scale again
    Ow = -(R.T @ t)
    pos, normal = np.zeros((npts, 3), np.float32), np.zeros((npts, 3), np.float32)
    dmin, dmax = np.zeros(npts, np.float32), np.zeros(npts, np.float32)
    mdesc = np.zeros((npts, 32), np.uint8)
    for i in range(npts):
        k = rng.randint(0, n)
        z = rng.uniform(2.0, 12.0)
        u, v = kps["x"][k] + rng.uniform(-2.5, 2.5), kps["y"][k] + rng.uniform(-2.5, 2.5)
        Xc = np.array([(u - K4[2]) / K4[0] * z, (v - K4[3]) / K4[1] * z, z])
        kind = rng.randint(0, 12)
        if kind == 0: Xc[2] = -Xc[2]                              # behind the camera
        if kind == 1: Xc[0] += 3 * z                              # outside the image
        Xw = R.T.astype(np.float64) @ (Xc - t)
        PO = Xw - Ow
        d = np.linalg.norm(PO)
        pos[i] = Xw
        normal[i] = (PO / d if kind != 2 else -PO / d)            # 2: seen from behind
        lvl = int(kps["octave"][k])
        dmax[i] = d * 1.2 ** (lvl - 0.5) if kind != 3 else d * 0.5   # 3: beyond the invariance region
        dmin[i] = dmax[i] / 1.2 ** 8 * 0.8
        dd = desc[k].copy()
        flips = rng.randint(0, 256, rng.randint(0, 40 if kind != 4 else 160))   # 4: a descriptor too far off
        for b in flips:
            dd[b >> 3] ^= 1 << (b & 7)
        mdesc[i] = dd
    return dict(K4=K4, gp=gp, kps=kps, desc=desc, T=T, S=S, pos=pos, normal=normal, dmin=dmin, dmax=dmax, mdesc=mdesc, rng=rng)


def _sim3_scene(P, seed, s12, n=600):
    """Two KeyFrames related by a similarity (p_c1 = s12 R12 p_c2 + t12).  Slot i of KeyFrame 1 carries a point that KeyFrame 2 sees
    near its keypoint pair[i]; slot pair[i] of KeyFrame 2 mostly carries one that KeyFrame 1 sees near keypoint i (the two searches
    agree), sometimes near another keypoint (they do not); some points are behind the camera, outside the image, out of range."""
    rng = np.random.RandomState(seed)
    K4 = np.array([517.3, 516.5, 318.6, 255.3], np.float32)
    gp = np.array([0.0, 0.0, 640.0, 480.0, 64 / 640.0, 48 / 480.0], np.float32)
    def kf():
        k = np.zeros(n, P.KP_DTYPE)
        k["x"], k["y"] = rng.uniform(5, 635, n).astype(np.float32), rng.uniform(5, 475, n).astype(np.float32)
        k["octave"], k["angle"] = rng.randint(0, 8, n), rng.uniform(0, 360, n).astype(np.float32)
        k["size"], k["response"], k["class_id"] = 31.0 * 1.2 ** k["octave"], rng.uniform(20, 200, n).astype(np.float32), -1
        return k, rng.randint(0, 256, (n, 32)).astype(np.uint8)
    (k1, d1), (k2, d2) = kf(), kf()
    def pose(a, t):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3], T[:3, 3] = _rot(*a), t
        return T
    T1, T2 = pose((0.05, -0.12, 0.03), (0.3, -0.2, 0.5)), pose((-0.2, 0.3, 0.1), (-1.0, 0.4, 0.2))
    R12, t12 = _rot(0.02, 0.05, -0.04), np.array([0.2, -0.1, 0.15], np.float32)
    pair = rng.permutation(n)
    inv = np.argsort(pair)
    def points(kt, dt, target, to1):
        """points whose image in the *other* KeyFrame is near that KeyFrame's keypoint target[i]"""
        pos, dmin, dmax = np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        md = np.zeros((n, 32), np.uint8)
        for i in range(n):
            k = int(target[i]) if rng.uniform() < 0.8 else rng.randint(0, n)
            z = rng.uniform(2.0, 12.0)
            u, v = kt["x"][k] + rng.uniform(-2.5, 2.5), kt["y"][k] + rng.uniform(-2.5, 2.5)
            Xt = np.array([(u - K4[2]) / K4[0] * z, (v - K4[3]) / K4[1] * z, z])       # in the other KeyFrame's camera
            kind = rng.randint(0, 14)
            if kind == 0: Xt[2] = -Xt[2]
            if kind == 1: Xt[0] += 3 * z
            d = np.linalg.norm(Xt)
            if to1:      # the point belongs to KeyFrame 2: camera 1 -> camera 2 -> world through T2
                Xc = (R12.T.astype(np.float64) @ (Xt - t12)) / s12
                Xw = T2[:3, :3].T.astype(np.float64) @ (Xc - T2[:3, 3])
            else:        # the point belongs to KeyFrame 1: camera 2 -> camera 1 -> world through T1
                Xc = s12 * (R12.astype(np.float64) @ Xt) + t12
                Xw = T1[:3, :3].T.astype(np.float64) @ (Xc - T1[:3, 3])
            pos[i] = Xw
            lvl = int(kt["octave"][k])
            dmax[i] = d * 1.2 ** (lvl - 0.5) if kind != 3 else d * 0.5
            dmin[i] = dmax[i] / 1.2 ** 8 * 0.8
            dd = dt[k].copy()
            for b in rng.randint(0, 256, rng.randint(0, 40 if kind != 4 else 160)):
                dd[b >> 3] ^= 1 << (b & 7)
            md[i] = dd
        return pos, dmin, dmax, md
    pts1 = points(k2, d2, pair, False)       # KeyFrame 1's points look at KeyFrame 2's keypoints
    pts2 = points(k1, d1, inv, True)
    has1, has2 = (rng.uniform(size=n) < 0.85).astype(np.uint8), (rng.uniform(size=n) < 0.85).astype(np.uint8)
    pre = np.full(n, -1, np.int32)
    for i in np.nonzero(rng.uniform(size=n) < 0.08)[0]:
        j = rng.randint(0, n)
        if has2[j]: pre[i] = j
    return dict(K4=K4, gp=gp, k1=k1, d1=d1, k2=k2, d2=d2, T1=T1, T2=T2, R12=R12, t12=t12, pts1=pts1, pts2=pts2, has1=has1, has2=has2, pre=pre)


def _line_fuse_scene(P, seed, n=500, nl=260, nc=400, behind_at=None):
    """A KeyFrame with nl keylines (and n ORB rows) and nc candidate MapLines: most project onto a keyline (endpoints within a pixel,
    predicted level = the keyline's octave, descriptor a few bits from the row the reference compares with -- the ORB row of the
    line's index, LSDmatcher.cpp:963), the rest fail one of the gates.  behind_at: position in the list of a line with an endpoint
    behind the camera (the reference returns there)."""
    rng = np.random.RandomState(seed)
    K4 = np.array([517.3, 516.5, 318.6, 255.3], np.float32)
    gp = np.array([0.0, 0.0, 640.0, 480.0, 64 / 640.0, 48 / 480.0], np.float32)
    kps = np.zeros(n, P.KP_DTYPE)
    kps["x"], kps["y"] = rng.uniform(5, 635, n).astype(np.float32), rng.uniform(5, 475, n).astype(np.float32)
    kps["octave"], kps["class_id"] = rng.randint(0, 8, n), -1
    desc = rng.randint(0, 256, (n, 32)).astype(np.uint8)
    kl = np.zeros(nl, P.KL_DTYPE)
    mx, my = rng.uniform(80, 560, nl), rng.uniform(80, 400, nl)
    ang, half = rng.uniform(0, np.pi, nl), rng.uniform(20, 70, nl)
    kl["startPointX"], kl["startPointY"] = mx - half * np.cos(ang), my - half * np.sin(ang)
    kl["endPointX"], kl["endPointY"] = mx + half * np.cos(ang), my + half * np.sin(ang)
    kl["pt_x"], kl["pt_y"] = (kl["startPointX"] + kl["endPointX"]) / 2, (kl["startPointY"] + kl["endPointY"]) / 2
    kl["octave"], kl["lineLength"], kl["class_id"] = rng.randint(0, 3, nl), 2 * half, np.arange(nl)
    kl["angle"] = ang
    ldesc = rng.randint(0, 256, (nl, 32)).astype(np.uint8)
    T = np.eye(4, dtype=np.float32)
    R, t = _rot(0.04, -0.1, 0.02), np.array([0.2, -0.1, 0.4], np.float32)
    T[:3, :3], T[:3, 3] = R, t
    Ow = -(R.T @ t)
    pos6, normal = np.zeros((nc, 6), np.float32), np.zeros((nc, 3), np.float32)
    dmin, dmax, cdesc = np.zeros(nc, np.float32), np.zeros(nc, np.float32), np.zeros((nc, 32), np.uint8)
    back = lambda u, v, z: R.T.astype(np.float64) @ (np.array([(u - K4[2]) / K4[0] * z, (v - K4[3]) / K4[1] * z, z]) - t)
    for i in range(nc):
        k = rng.randint(0, nl)
        kind = rng.randint(0, 12)
        j = rng.uniform(-1.0, 1.0, 4) * (1.0 if kind != 5 else 12.0)                # 5: direction off by more than acos(0.998)
        z1, z2 = rng.uniform(2.0, 10.0), rng.uniform(2.0, 10.0)
        u1, v1, u2, v2 = kl["startPointX"][k] + j[0], kl["startPointY"][k] + j[1], kl["endPointX"][k] + j[2], kl["endPointY"][k] + j[3]
        if kind == 1: u1 += 900                                                     # start point outside the image
        S3, E3 = back(u1, v1, z1), back(u2, v2, z2)
        OM = 0.5 * (S3 + E3) - Ow
        d = np.linalg.norm(OM)
        pos6[i, :3], pos6[i, 3:] = S3, E3
        normal[i] = OM / d if kind != 2 else -OM / d
        lvl = int(kl["octave"][k]) + (1 if kind == 6 else 0)                        # 6: predicted level one above the keyline's
        dmax[i] = d * 1.2 ** (lvl - 0.5) if kind != 3 else d * 0.5
        dmin[i] = dmax[i] / 1.2 ** 8 * 0.8
        dd = desc[k].copy()
        for b in rng.randint(0, 256, rng.randint(0, 40 if kind != 4 else 160)):
            dd[b >> 3] ^= 1 << (b & 7)
        cdesc[i] = dd
    order = np.r_[rng.permutation(nc), rng.randint(0, nc, 40), [-1, -1]].astype(np.int32)
    rng.shuffle(order)
    if behind_at is not None:
        i = int(order[behind_at])
        S3, E3 = back(300, 200, 3.0), back(320, 230, -2.0)
        pos6[i, :3], pos6[i, 3:] = S3, E3
    return dict(K4=K4, gp=gp, kps=kps, desc=desc, kl=kl, ldesc=ldesc, T=T, pos6=pos6, normal=normal, dmin=dmin, dmax=dmax, cdesc=cdesc,
                order=order, rng=rng)


def _backend_calls(P, S, path, tmp_path):
    G, R = _lib(path)
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(V)
    n_ref = C.c_int(0)
    # ---- LoopClosing: SearchByProjection(pKF, Scw, vpPoints, vpMatched, th)
    for seed, scale, th in ((11, 1.0, 10), (12, 1.7, 10), (13, 0.6, 4)):
        sc = _backend_scene(P, seed, sim3_scale=scale)
        n, npts = len(sc["kps"]), len(sc["pos"])
        matched0 = (sc["rng"].uniform(size=n) < 0.15).astype(np.uint8)
        o_ref, o_hip = np.zeros(n, np.int32), np.zeros(n, np.int32)
        R.adx_loop_search_by_projection.argtypes = [V, V, I, V, V, V, I, F, V, V, I, V, V, V, V, V, I, V, V, V]
        nm = R.adx_loop_search_by_projection(p(sc["kps"]), p(sc["desc"]), n, p(sc["gp"]), p(sc["T"]), p(sc["K4"]), 8, 1.2, p(sc["S"]),
                                             p(matched0), npts, p(sc["pos"]), p(sc["normal"]), p(sc["dmin"]), p(sc["dmax"]), p(sc["mdesc"]),
                                             th, p(o_ref), p(o_hip), C.byref(n_ref))
        assert nm == n_ref.value and nm > 100, (seed, nm, n_ref.value)
        assert (o_ref == o_hip).all(), "SearchByProjection(KF, Scw) seed %d: vpMatched differs from the reference's" % seed
    # ---- LocalMapping: Fuse(pKF, vpMapPoints, th)
    for seed, th in ((21, 3.0), (22, 3.0), (23, 6.0)):
        sc = _backend_scene(P, seed)
        n, npts = len(sc["kps"]), len(sc["pos"])
        rng = sc["rng"]
        kf_obs = np.where(rng.uniform(size=n) < 0.4, rng.randint(1, 9, n), 0).astype(np.int32)
        cand_obs = rng.randint(0, 9, npts).astype(np.int32)
        order = np.concatenate([rng.permutation(npts), rng.randint(0, npts, 60), -np.ones(15)]).astype(np.int32)   # repeats and NULLs
        rng.shuffle(order)
        outs = [np.zeros(n, np.int32), np.zeros(npts, np.int32), np.zeros(n, np.int32), np.zeros(npts, np.int32)]
        R.adx_local_mapping_fuse.argtypes = [V, V, I, V, V, V, I, F, V, I, V, V, V, V, V, V, V, I, F, V, V, V, V, V]
        nf = R.adx_local_mapping_fuse(p(sc["kps"]), p(sc["desc"]), n, p(sc["gp"]), p(sc["T"]), p(sc["K4"]), 8, 1.2, p(kf_obs), npts,
                                      p(sc["pos"]), p(sc["normal"]), p(sc["dmin"]), p(sc["dmax"]), p(sc["mdesc"]), p(cand_obs), p(order),
                                      len(order), th, p(outs[0]), p(outs[1]), p(outs[2]), p(outs[3]), C.byref(n_ref))
        assert nf == n_ref.value and nf > 100, (seed, nf, n_ref.value)
        assert (outs[0] == outs[2]).all(), "Fuse seed %d: the KeyFrame's map points differ from the reference's" % seed
        assert (outs[1] == outs[3]).all(), "Fuse seed %d: isBad / IsInKeyFrame / Observations of the candidates differ" % seed
        assert (outs[1] & 1).sum() > 10 and (outs[0] >= 0).sum() > (kf_obs > 0).sum()     # points were replaced and added
    # ---- Tracking::Relocalization: SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist)
    for seed, th, odist in ((51, 10.0, 100), (52, 3.0, 64)):
        sc = _backend_scene(P, seed)
        n, npts = len(sc["kps"]), len(sc["pos"])
        rng = sc["rng"]
        cur_has = (rng.uniform(size=n) < 0.2).astype(np.uint8)
        found = (rng.uniform(size=npts) < 0.1).astype(np.uint8)
        kf_angle = rng.uniform(0, 360, npts).astype(np.float32)
        o_ref, o_hip = np.zeros(n, np.int32), np.zeros(n, np.int32)
        R.adx_relocalization_search.argtypes = [V, V, I, V, V, V, I, F, V, I, V, V, V, V, V, V, F, I, V, V, V]
        nm = R.adx_relocalization_search(p(sc["kps"]), p(sc["desc"]), n, p(sc["gp"]), p(sc["T"]), p(sc["K4"]), 8, 1.2, p(cur_has), npts,
                                         p(kf_angle), p(sc["pos"]), p(sc["dmin"]), p(sc["dmax"]), p(sc["mdesc"]), p(found), th, odist,
                                         p(o_ref), p(o_hip), C.byref(n_ref))
        assert nm == n_ref.value and nm > 30, (seed, nm, n_ref.value)
        assert (o_ref == o_hip).all(), "relocalisation SearchByProjection seed %d: mvpMapPoints differ from the reference's" % seed
    # ---- LoopClosing: Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)
    for seed, scale, th in ((41, 1.0, 4.0), (42, 1.4, 4.0), (43, 0.8, 8.0)):
        sc = _backend_scene(P, seed, sim3_scale=scale)
        n, npts = len(sc["kps"]), len(sc["pos"])
        kf_has = (sc["rng"].uniform(size=n) < 0.5).astype(np.uint8)
        outs = [np.zeros(npts, np.int32), np.zeros(n, np.int32), np.zeros(npts, np.int32), np.zeros(n, np.int32)]
        R.adx_loop_fuse.argtypes = [V, V, I, V, V, V, I, F, V, V, I, V, V, V, V, V, F, V, V, V, V, V]
        nf = R.adx_loop_fuse(p(sc["kps"]), p(sc["desc"]), n, p(sc["gp"]), p(sc["T"]), p(sc["K4"]), 8, 1.2, p(sc["S"]), p(kf_has), npts,
                             p(sc["pos"]), p(sc["normal"]), p(sc["dmin"]), p(sc["dmax"]), p(sc["mdesc"]), th, p(outs[0]), p(outs[1]), p(outs[2]),
                             p(outs[3]), C.byref(n_ref))
        assert nf == n_ref.value and nf > 100, (seed, nf, n_ref.value)
        assert (outs[0] == outs[2]).all(), "Fuse(Scw) seed %d: vpReplacePoint differs from the reference's" % seed
        assert (outs[1] == outs[3]).all(), "Fuse(Scw) seed %d: the KeyFrame's map points differ from the reference's" % seed
        assert (outs[0] != -1).sum() > 20 and (outs[1] >= 0).sum() > 20      # replacements proposed and points added
    # ---- LocalMapping::CreateNewMapLines: LSDmatcher::SearchForTriangulation, both overloads
    for seed, mode in ((71, 0), (72, 1), (73, 2)):
        a, b, _ = S.make_descriptor_sets(seed, 300, flip_p=0.05)
        rng = np.random.RandomState(seed)
        k1, k2 = np.zeros(len(a), P.KL_DTYPE), np.zeros(len(b), P.KL_DTYPE)
        h1, h2 = (rng.uniform(size=len(a)) < 0.3).astype(np.uint8), (rng.uniform(size=len(b)) < 0.3).astype(np.uint8)
        o_ref, o_hip = np.zeros(len(a), np.int32), np.zeros(len(a), np.int32)
        R.adx_local_mapping_line_triangulation.argtypes = [V, V, V, I, V, V, V, I, I, V, V, V]
        nm = R.adx_local_mapping_line_triangulation(p(k1), p(a), p(h1), len(a), p(k2), p(b), p(h2), len(b), mode, p(o_ref), p(o_hip),
                                                    C.byref(n_ref))
        assert nm == n_ref.value and nm > 30, (seed, mode, nm, n_ref.value)
        assert (o_ref == o_hip).all(), "LSDmatcher::SearchForTriangulation mode %d differs from the reference's" % mode
    # ---- LocalMapping.cc:960 (commented out in the reference): LSDmatcher::SearchForTriangulationNew -- FrameBFMatchNew both ways over the
    # reference's own ComputeF12, the overlap gate, the mutual check, the MapLine gate -- on two posed KeyFrames
    G = _gen()
    for seed, n1, n2, dbl in ((11, 300, 280, 0), (18, 900, 900, 1), (15, 37, 411, 1)):
        x = G.lnew_inputs(S, seed, n1, n2, 0.1)
        ks = []
        for seg in (x["seg1"], x["seg2"]):
            k = np.zeros(len(seg), P.KL_DTYPE)
            k["startPointX"], k["startPointY"], k["endPointX"], k["endPointY"] = seg[:, 0], seg[:, 1], seg[:, 2], seg[:, 3]
            k["class_id"] = np.arange(len(seg))
            ks.append(k)
        Ts = []
        for pose in (x["pose1"], x["pose2"]):
            T = np.eye(4, dtype=np.float32); T[:3, :3] = pose[:9].reshape(3, 3); T[:3, 3] = pose[9:]
            Ts.append(np.ascontiguousarray(T))
        K4 = np.array([517.3, 516.5, 318.6, 255.3], np.float32)
        o_ref, o_hip = np.zeros(n1, np.int32), np.zeros(n1, np.int32)
        R.adx_local_mapping_line_triangulation_new.argtypes = [V, V, V, V, I, V, V, V, V, I, V, V, V, I, V, V, V]
        nm = R.adx_local_mapping_line_triangulation_new(p(ks[0]), p(x["d1"]), p(x["func1"]), p(x["ml1"]), n1, p(ks[1]), p(x["d2"]), p(x["func2"]),
                                                        p(x["ml2"]), n2, p(Ts[0]), p(Ts[1]), p(K4), dbl, p(o_ref), p(o_hip), C.byref(n_ref))
        assert nm == n_ref.value and (nm > 30 or n1 < 100), (seed, dbl, nm, n_ref.value)
        assert (o_ref == o_hip).all(), "LSDmatcher::SearchForTriangulationNew seed %d differs from the reference's" % seed
    # ---- LocalMapping::SearchLineInNeighbors: LSDmatcher::Fuse(pKF, vpMapLines, th)
    for seed, th, behind in ((81, 3.0, None), (82, 6.0, 380)):
        sc = _line_fuse_scene(P, seed, behind_at=behind)
        n, nl, nc = len(sc["kps"]), len(sc["kl"]), len(sc["pos6"])
        rng = sc["rng"]
        kf_obs = np.where(rng.uniform(size=nl) < 0.5, rng.randint(1, 6, nl), 0).astype(np.int32)
        cand_obs = rng.randint(1, 6, nc).astype(np.int32)
        outs = [np.zeros(nl, np.int32), np.zeros(nc, np.int32), np.zeros(nl, np.int32), np.zeros(nc, np.int32)]
        R.adx_local_mapping_line_fuse.argtypes = [V, V, I, V, V, I, V, V, V, F, V, I, V, V, V, V, V, V, V, I, F, V, V, V, V, V]
        nf = R.adx_local_mapping_line_fuse(p(sc["kps"]), p(sc["desc"]), n, p(sc["kl"]), p(sc["ldesc"]), nl, p(sc["gp"]), p(sc["T"]),
                                           p(sc["K4"]), 1.2, p(kf_obs), nc, p(sc["pos6"]), p(sc["normal"]), p(sc["dmin"]), p(sc["dmax"]),
                                           p(sc["cdesc"]), p(cand_obs), p(sc["order"]), len(sc["order"]), th, p(outs[0]), p(outs[1]),
                                           p(outs[2]), p(outs[3]), C.byref(n_ref))
        assert nf == n_ref.value, (seed, nf, n_ref.value)
        assert (nf == 0) if behind is not None else (nf > 100), (seed, nf)      # `return false` after the line behind the camera
        assert (outs[0] == outs[2]).all(), "LSDmatcher::Fuse seed %d: the KeyFrame's map lines differ from the reference's" % seed
        assert (outs[1] == outs[3]).all(), "LSDmatcher::Fuse seed %d: isBad / IsInKeyFrame / Observations of the candidates differ" % seed
        assert (outs[1] & 1).sum() > 10 and (outs[0] >= 0).sum() > 20                     # lines were replaced and added
    # ---- LoopClosing::ComputeSim3: SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)
    for seed, s12, th in ((61, 1.0, 7.5), (62, 1.4, 7.5), (63, 0.7, 3.0)):
        sc = _sim3_scene(P, seed, s12)
        n = len(sc["k1"])
        o_ref, o_hip = np.zeros(n, np.int32), np.zeros(n, np.int32)
        R.adx_loop_search_by_sim3.argtypes = [V, V, I, V, V, V, V, V, V, V, V, I, V, V, V, V, V, V, V, V, F, V, V, V, F, V, V, V]
        nf = R.adx_loop_search_by_sim3(p(sc["k1"]), p(sc["d1"]), n, p(sc["T1"]), p(sc["has1"]), *[p(x) for x in sc["pts1"]], p(sc["k2"]),
                                       p(sc["d2"]), n, p(sc["T2"]), p(sc["has2"]), *[p(x) for x in sc["pts2"]], p(sc["gp"]), p(sc["K4"]),
                                       s12, p(sc["R12"]), p(sc["t12"]), p(sc["pre"]), th, p(o_ref), p(o_hip), C.byref(n_ref))
        assert nf == n_ref.value and nf > 100, (seed, nf, n_ref.value)
        assert (o_ref == o_hip).all(), "SearchBySim3 seed %d: vpMatches12 differs from the reference's" % seed
    # ---- LoopClosing: SearchByBoW(pKF1, pKF2, vpMatches12)
    VM = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    voc = VM.Vocabulary.synthetic(41, k=10, L=4, synth=S)
    vpath = str(tmp_path / "voc.txt")
    voc.save_text(vpath)
    for seed in (31, 32):
        a, b, _ = S.make_descriptor_sets(seed, 800, flip_p=0.06)
        rng = np.random.RandomState(seed)
        def kps_of(m):
            k = np.zeros(m, P.KP_DTYPE)
            k["x"], k["y"] = rng.uniform(5, 635, m).astype(np.float32), rng.uniform(5, 475, m).astype(np.float32)
            k["octave"], k["angle"], k["class_id"] = rng.randint(0, 8, m), rng.uniform(0, 360, m).astype(np.float32), -1
            return k
        k1, k2 = kps_of(len(a)), kps_of(len(b))
        k2["angle"] = (k1["angle"][np.argsort(np.argsort(rng.uniform(size=len(b))))] + 10).astype(np.float32) % 360
        h1, h2 = (rng.uniform(size=len(a)) < 0.8).astype(np.uint8), (rng.uniform(size=len(b)) < 0.8).astype(np.uint8)
        o_ref, o_hip = np.zeros(len(a), np.int32), np.zeros(len(a), np.int32)
        gp = np.array([0.0, 0.0, 640.0, 480.0, 0.1, 0.1], np.float32)
        R.adx_loop_search_by_bow.argtypes = [C.c_char_p, V, V, V, I, V, V, V, I, V, V, V, V]
        nm = R.adx_loop_search_by_bow(vpath.encode(), p(k1), p(a), p(h1), len(a), p(k2), p(b), p(h2), len(b), p(gp), p(o_ref), p(o_hip),
                                      C.byref(n_ref))
        assert nm == n_ref.value and nm >= 0, (seed, nm, n_ref.value)
        assert (o_ref == o_hip).all(), "SearchByBoW(KF, KF) seed %d differs from the reference's" % seed
        assert (o_ref >= 0).sum() > 50
    # ---- LocalMapping::CreateNewMapPoints: SearchForTriangulation(pKF1, pKF2, F12, vMatchedIndices, false)
    for seed, chk in ((35, 0), (36, 1)):
        a, b, perm = S.make_descriptor_sets(seed, 800, flip_p=0.06)
        rng = np.random.RandomState(seed)
        k1 = np.zeros(len(a), P.KP_DTYPE)
        k1["x"], k1["y"] = rng.uniform(60, 620, len(a)).astype(np.float32), rng.uniform(20, 460, len(a)).astype(np.float32)
        k1["octave"], k1["angle"], k1["class_id"] = rng.randint(0, 8, len(a)), rng.uniform(0, 360, len(a)).astype(np.float32), -1
        k2 = k1[perm].copy() if len(perm) == len(b) else k1[:len(b)].copy()
        k2["x"] = (k2["x"] - rng.uniform(2, 40, len(b))).astype(np.float32)              # disparity along the epipolar line
        k2["y"] = (k2["y"] + rng.uniform(-3.0, 3.0, len(b))).astype(np.float32)          # some violate the 3.84 sigma^2 gate
        k2["angle"] = ((k2["angle"] + rng.choice([5.0, 5.0, 5.0, 90.0], len(b))) % 360).astype(np.float32)
        h1, h2 = (rng.uniform(size=len(a)) < 0.3).astype(np.uint8), (rng.uniform(size=len(b)) < 0.3).astype(np.uint8)
        T1 = np.eye(4, dtype=np.float32)
        T2 = np.eye(4, dtype=np.float32); T2[:3, 3] = (-0.4, 0.0, 0.35)     # the epipole lands inside the image: its gate is exercised
        F12 = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32)            # l = (0, -1, y1): distance^2 = (y2 - y1)^2
        K4 = np.array([500, 500, 320, 240], np.float32)
        gp = np.array([0.0, 0.0, 640.0, 480.0, 0.1, 0.1], np.float32)
        o_ref, o_hip = np.zeros(len(a), np.int32), np.zeros(len(a), np.int32)
        R.adx_local_mapping_triangulation.argtypes = [C.c_char_p, V, V, V, I, V, V, V, I, V, V, V, V, V, I, V, V, V]
        nm = R.adx_local_mapping_triangulation(vpath.encode(), p(k1), p(a), p(h1), len(a), p(k2), p(b), p(h2), len(b), p(gp), p(T1), p(T2),
                                               p(K4), p(F12), chk, p(o_ref), p(o_hip), C.byref(n_ref))
        assert nm == n_ref.value and nm > 50, (seed, nm, n_ref.value)
        assert (o_ref == o_hip).all(), "SearchForTriangulation seed %d differs from the reference's" % seed


def test_adaptor_executes_backend_searches_emu(plslam, synth, emu_lib, tmp_path):
    _backend_calls(plslam, synth, LIB_EMU, tmp_path)


@pytest.mark.gpu
def test_adaptor_executes_backend_searches(plslam, synth, tmp_path):
    _backend_calls(plslam, synth, LIB_HIP, tmp_path)
