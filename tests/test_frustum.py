"""Frame::isInFrustum for map points and map lines (+ MapPoint / MapLine::PredictScale) -- the visibility test in front of
SearchByProjection(F, MapPoints / MapLines) (reference src/Frame.cc:560-623, 625-711; src/MapPoint.cc:413-428;
src/MapLine.cpp:395-404).  C ABI: plh_frame_is_in_frustum_{points,lines}_batch_dev.

* the oracle against hand-derived known answers;
* the oracle against the REFERENCE's own Frame.cc / MapPoint.cc / MapLine.cpp (oracle/_ref/libframe_ref.so, real Frame,
  MapPoint and MapLine objects) for poses without rotation -- the case in which the stand-in cv::Mat algebra and OpenCV's
  gemm provably give the same floats; committed as tests/golden/ref_frustum.npz.  With a rotation the result depends on
  how cv::gemm rounds `mRcw*P+mtcw` (one double-accumulated product, the definition pinned in oracle/plo.h): PARITY
  UNPINNED there, the GPU is held to the oracle;
* the HIP kernels (host emulator, and `-m gpu` on the GPU) against the oracle for rotated poses and against the goldens."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLDEN = os.path.join(_util.ROOT, "tests", "golden", "ref_frustum.npz")
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "libframe_ref.so")
V, I, F = C.c_void_p, C.c_int, C.c_float


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _oracle(O, view, nlv, e, lines, cos):
    L = O.lib()
    L.plo_frame_is_in_frustum_points.argtypes = [V, I, I, V, V, V, V, F, V, V, V, V]
    L.plo_frame_is_in_frustum_lines.argtypes = [V, I, V, V, V, V, F, V, V, V, V]
    n = len(e["min_dist"])
    valid, proj = np.zeros(max(n, 1), np.uint8), np.zeros((max(n, 1), 4 if lines else 2), np.float32)
    level, vc = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float32)
    p = O._p
    if lines:
        L.plo_frame_is_in_frustum_lines(p(view), n, p(e["pos"]), p(e["normal"]), p(e["min_dist"]), p(e["max_dist"]), cos, p(valid), p(proj),
                                        p(level), p(vc))
    else:
        L.plo_frame_is_in_frustum_points(p(view), nlv, n, p(e["pos"]), p(e["normal"]), p(e["min_dist"]), p(e["max_dist"]), cos, p(valid),
                                         p(proj), p(level), p(vc))
    return valid[:n], proj[:n], level[:n], vc[:n]


def _view_record(P, view, nlv):
    r = np.zeros(1, P.VIEW_DTYPE)
    r["Rcw"], r["tcw"], r["Ow"] = view[:9], view[9:12], view[12:15]
    for k, name in enumerate(("fx", "fy", "cx", "cy", "min_x", "min_y", "max_x", "max_y", "log_scale_factor")):
        r[name] = view[15 + k]
    r["n_scale_levels"] = nlv
    return r[0]


def _same(a, b):
    return all((x == y).all() for x, y in zip(a, b))


def test_oracle_known_answers(oracle):
    # camera at the origin looking down +z, fx = fy = 500, principal point (320, 240), image 640 x 480, 8 levels of 1.2
    view = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 500, 500, 320, 240, 0, 0, 640, 480, np.log(np.float32(1.2))], np.float32)
    e = dict(pos=np.array([[0, 0, 2], [0.2, -0.1, 2], [0, 0, -2], [2, 0, 2], [0, 0, 2], [0, 0, 2], [0, 0, 2]], np.float32),
             normal=np.array([[0, 0, 1]] * 5 + [[1, 0, 0], [0, 0, 1]], np.float32),
             min_dist=np.array([0.5] * 6 + [0.5], np.float32), max_dist=np.array([4.0, 4.0, 4.0, 4.0, 1.0, 4.0, 2.0], np.float32))
    valid, uv, level, vc = _oracle(oracle, view, 8, e, 0, 0.5)
    assert list(valid) == [1, 1, 0, 0, 0, 0, 1]            # behind / outside the image / beyond 1.2 max / seen at 90 degrees
    assert (uv[0] == [320, 240]).all() and (uv[1] == [370, 215]).all() and vc[0] == 1.0
    assert level[0] == 4 and level[6] == 0                 # ceil(log(4/2) / log 1.2) = ceil(3.80) = 4 ; ratio 1 -> 0
    le = dict(pos=np.array([[-0.2, 0, 2, 0.2, 0, 2], [-0.2, 0, 2, 0.2, 0, -1]], np.float32), normal=np.array([[0, 0, 1]] * 2, np.float32),
              min_dist=np.array([0.5, 0.5], np.float32), max_dist=np.array([64.0, 4.0], np.float32))
    valid, seg, level, vc = _oracle(oracle, view, 8, le, 1, 0.5)
    assert list(valid) == [1, 0] and (seg[0] == [270, 240, 370, 240]).all()
    assert level[0] == 20                                   # MapLine::PredictScale does not clamp: ceil(log 32 / log 1.2) = 20


def test_golden_file_present():
    assert os.path.exists(GOLDEN)


def test_oracle_reproduces_reference_frustum(oracle, plslam, synth):
    G = _gen()
    TF = G._test_module("test_frame_search")
    g = np.load(GOLDEN)
    for seed, n, dist in G.FRUSTUM_CASES:
        view, nlv = G.frustum_view(synth, plslam, TF, seed, dist, rotate=False)
        for lines in (0, 1):
            e = G.frustum_elems(synth, seed, n, view, lines)
            key = "%s_%d" % ("l" if lines else "p", seed)
            assert _same(_oracle(oracle, view, nlv, e, lines, G.FRUSTUM_COS), [g[key + s] for s in ("_valid", "_proj", "_level", "_vc")]), key


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_frustum_live(oracle, plslam, synth):
    G = _gen()
    TF = G._test_module("test_frame_search")
    R = G.ref_frame_lib()
    for seed, n, dist in [(11, 2000, True), (12, 77, False)]:
        view, nlv = G.frustum_view(synth, plslam, TF, seed, dist, rotate=False)
        for lines in (0, 1):
            e = G.frustum_elems(synth, seed, n, view, lines)
            ref = G.reference_frustum(R, view, nlv, e, lines)
            assert ref[0].sum() > n // 5 and _same(_oracle(oracle, view, nlv, e, lines, G.FRUSTUM_COS), ref), (seed, lines)


def _device(P, O, synth, lib, sizes):
    G = _gen()
    TF = G._test_module("test_frame_search")
    g = np.load(GOLDEN)
    for lines in (0, 1):
        # rotated poses, several frames per launch, ragged element counts: against the oracle
        views, elems, refs = [], [], []
        for b, n in enumerate(sizes):
            view, nlv = G.frustum_view(synth, P, TF, 40 + b, b % 2 == 1, rotate=True)
            e = G.frustum_elems(synth, 40 + b, n, view, lines)
            views.append(_view_record(P, view, nlv)); elems.append(e)
            refs.append(_oracle(O, view, nlv, e, lines, 0.5))
        got = P.is_in_frustum(np.array(views, P.VIEW_DTYPE), elems, 0.5, lines=bool(lines), lib=lib)
        for b, (r, q) in enumerate(zip(refs, got)):
            assert _same(r, (q["valid"], q["seg" if lines else "uv"], q["level"], q["viewcos"])), "frame %d %s" % (b, "lines" if lines else "points")
            assert r[0].sum() > len(r[0]) // 6 or len(r[0]) < 10
        # the committed reference outputs
        for seed, n, dist in G.FRUSTUM_CASES:
            view, nlv = G.frustum_view(synth, P, TF, seed, dist, rotate=False)
            e = G.frustum_elems(synth, seed, n, view, lines)
            q = P.is_in_frustum(np.array([_view_record(P, view, nlv)], P.VIEW_DTYPE), [e], G.FRUSTUM_COS, lines=bool(lines), lib=lib)[0]
            key = "%s_%d" % ("l" if lines else "p", seed)
            assert _same((q["valid"], q["seg" if lines else "uv"], q["level"], q["viewcos"]),
                         [g[key + s] for s in ("_valid", "_proj", "_level", "_vc")]), key


def test_emu_frustum(plslam, oracle, synth, emu_lib):
    _device(plslam, oracle, synth, emu_lib, [400, 0, 37, 1])


@pytest.mark.gpu
def test_gpu_frustum(plslam, oracle, synth):
    _device(plslam, oracle, synth, None, [5000, 0, 37, 1, 2048, 999])


# ---- the inline projection of the pose-driven searches (plh_frame_project_points_batch_dev)
def _oracle_project(O, view, form, pos):
    L = O.lib()
    L.plo_frame_project_points.argtypes = [V, I, I, V, V, V]
    n = len(pos)
    front, uv = np.zeros(max(n, 1), np.uint8), np.zeros((max(n, 1), 2), np.float32)
    L.plo_frame_project_points(O._p(view), form, n, O._p(pos), O._p(front), O._p(uv))
    return front[:n], uv[:n]


def _identity_view(G):
    return np.concatenate([np.eye(3).reshape(9), np.zeros(6), G.POSE_K, [0, 0, 640, 480], [np.log(np.float32(1.2))]]).astype(np.float32)


def _project_goldens(G, P, S):
    """(form, world points, golden uv, golden front or None) from the matcher pins: the uv there were produced by the harness'
    copy of each reference expression, compiled next to the reference code that consumed them."""
    TF = G._test_module("test_frame_search")
    go = np.load(os.path.join(_util.ROOT, "tests", "golden", "ref_orbmatcher.npz"))
    gk = np.load(os.path.join(_util.ROOT, "tests", "golden", "ref_orbmatcher_kf.npz"))
    out = []
    for seed, n, dist in G.POSE_CASES:
        f2, gp, q, xyz, fl, occ_f, occ_k = G.pose_inputs(S, P, TF, seed, n, dist)
        out.append((0, xyz, go["pose_%d_uv" % seed], None, (go["pf_%d_valid" % seed], fl["mp"] & (1 - fl["outlier"]))))
        out.append((1, xyz, gk["fuse_%d_uv" % seed], None, None))
        out.append((1, xyz, gk["s3p_%d_uv" % seed], None, None))
        out.append((2, xyz, gk["fuse3_%d_uv" % seed], None, None))
        f1, f2, gp, sides, already = G.sim3_inputs(S, P, TF, seed, n, dist)
        out.append((2, sides[0]["xyz"], gk["sim3_%d_uv12" % seed], None, None))
        out.append((2, sides[1]["xyz"], gk["sim3_%d_uv21" % seed], None, None))
    return out


def test_oracle_projection_reproduces_matcher_goldens(oracle, plslam, synth):
    G = _gen()
    view = _identity_view(G)
    for form, xyz, uv, _, vf in _project_goldens(G, plslam, synth):
        front, got = _oracle_project(oracle, view, form, xyz)
        assert (got == uv).all(), "form %d" % form
        if vf is not None:
            assert (vf[0] == (vf[1] & front)).all()


def _device_project(P, O, synth, lib):
    G = _gen()
    TF = G._test_module("test_frame_search")
    ident = _view_record(P, _identity_view(G), 8)
    for form in (0, 1, 2):
        views, poss, refs = [], [], []
        for b, n in enumerate([700, 0, 33, 1]):
            view, nlv = G.frustum_view(synth, P, TF, 60 + b, False, rotate=True)
            pos = G.frustum_elems(synth, 60 + b, n, view, 0)["pos"]
            views.append(_view_record(P, view, nlv)); poss.append(pos); refs.append(_oracle_project(O, view, form, pos))
        got = P.project_points(np.array(views, P.VIEW_DTYPE), poss, form, lib=lib)
        for (rf, ru), (gf, gu) in zip(refs, got):
            assert (rf == gf).all() and (ru == gu).all(), "form %d" % form
    for form, xyz, uv, _, _ in _project_goldens(G, P, synth):
        gf, gu = P.project_points(np.array([ident], P.VIEW_DTYPE), [xyz], form, lib=lib)[0]
        assert (gu == uv).all(), "golden form %d" % form


def test_emu_projection(plslam, oracle, synth, emu_lib):
    _device_project(plslam, oracle, synth, emu_lib)


@pytest.mark.gpu
def test_gpu_projection(plslam, oracle, synth):
    _device_project(plslam, oracle, synth, None)


# ---- the gates in front of the back end's pose-driven searches (plh_map_point_gates)
GATE_FORMS = {"relocalisation": 2, "loop_closing": 1 | 4 | 8 | 32, "fuse": 1 | 4 | 8 | 32, "fuse_sim3": 1 | 2 | 4 | 8 | 32,
              "search_by_sim3": 1 | 2 | 4 | 8 | 16 | 64}


def _oracle_gates(O, view, nlv, flags, R2t2, pos, normal, dmin, dmax, raw, pre):
    L = O.lib()
    L.plo_map_point_gates.argtypes = [V, I, V, I, I, V, V, V, V, V, V, V, V, V]
    L.plo_map_point_gates.restype = None
    n = len(pos)
    m = max(n, 1)
    valid = np.ascontiguousarray(pre, np.uint8).copy() if n else np.zeros(1, np.uint8)
    uv, dist, level = np.zeros((m, 2), np.float32), np.zeros(m, np.float32), np.zeros(m, np.int32)
    L.plo_map_point_gates(O._p(view), nlv, O._p(R2t2), flags, n, O._p(pos), O._p(normal), O._p(dmin), O._p(dmax),
                          O._p(raw) if raw is not None else None, O._p(valid), O._p(uv), O._p(dist), O._p(level))
    return valid[:n], uv[:n], dist[:n], level[:n]


def _gate_case(G, S, P, TF, seed, n):
    view, nlv = G.frustum_view(S, P, TF, seed, False, rotate=True)
    e = G.frustum_elems(S, seed, n, view, 0)
    rng = np.random.RandomState(seed)
    # a similarity close to the identity as the second transform (SearchBySim3's sR21 / t21)
    a = rng.uniform(-0.05, 0.05, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
    Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
    R2 = (rng.uniform(0.9, 1.1) * Rx @ Ry).astype(np.float32)
    t2 = rng.uniform(-0.1, 0.1, 3).astype(np.float32)
    raw = np.ascontiguousarray(e["max_dist"], np.float32)
    dmin = (np.float32(0.8) * np.ascontiguousarray(e["min_dist"], np.float32)).astype(np.float32)
    dmax = (np.float32(1.2) * raw).astype(np.float32)
    pre = (rng.uniform(size=n) < 0.9).astype(np.uint8)
    return view, nlv, e, np.concatenate([R2.reshape(9), t2]).astype(np.float32), dmin, dmax, raw, pre


def _device_gates(P, O, S, lib, sizes):
    G = _gen()
    TF = G._test_module("test_frame_search")
    passed = 0
    for k, n in enumerate(sizes):
        view, nlv, e, R2t2, dmin, dmax, raw, pre = _gate_case(G, S, P, TF, 80 + k, n)
        rec = _view_record(P, view, nlv)
        for name, flags in GATE_FORMS.items():
            for with_raw in (True, False):
                ref = _oracle_gates(O, view, nlv, flags, R2t2, e["pos"], e["normal"], dmin, dmax, raw if with_raw else None, pre)
                got = P.map_point_gates(rec, flags, e["pos"], e["normal"], dmin, dmax, raw if with_raw else None, pre, R2t2[:9], R2t2[9:], lib=lib)
                assert _same(ref, got), "%s, %d points, raw %s" % (name, n, with_raw)
                passed += int(ref[0].sum())
    assert passed > 0   # (the gates do let points through on this content)


def test_emu_map_point_gates(plslam, oracle, synth, emu_lib):
    _device_gates(plslam, oracle, synth, emu_lib, [600, 0, 1, 65])


@pytest.mark.gpu
def test_gpu_map_point_gates(plslam, oracle, synth):
    _device_gates(plslam, oracle, synth, None, [6000, 0, 1, 257, 2048])
