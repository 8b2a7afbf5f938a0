"""Grid ("windowed") searches of the tracking front end: Frame::AssignFeaturesToGrid*, GetFeaturesInArea*,
ORBmatcher::SearchForInitialization / SearchByProjection, LSDmatcher::SearchByProjection.
Oracle known answers (CPU), the kernel sources under hipemu (CPU), GPU parity.  Everything is integer / index work:
the bar is exact equality with the oracle."""
import ctypes as C

import numpy as np
import pytest

import _util

V, I, F = C.c_void_p, C.c_int, C.c_float


def _olib(O):
    L = O.lib()
    if getattr(L, "_fs_ready", False):
        return L
    L.plo_frame_assign_grid.argtypes = [V, I, V, V, V]
    L.plo_frame_assign_grid_lines.argtypes = [V, I, V, V, V, I]
    L.plo_features_in_area.argtypes = [V, V, V, V, F, F, F, I, I, V, I]
    L.plo_features_in_area_for_line.argtypes = [V, V, I, V, V, V, F, F, F, F, F, F, V, I]
    L.plo_orb_search_for_initialization.argtypes = [V, V, I, V, V, I, V, V, V, V, I, F, I, V]
    L.plo_orb_search_by_projection_mp.argtypes = [V, V, I, V, V, V, V, V, I, V, V, V, V, V, V, F, F, V]
    L.plo_orb_search_by_projection_frame.argtypes = [V, V, I, V, V, V, V, V, I, V, V, V, V, V, V, F, I, I, V]
    L.plo_line_search_by_projection_frame.argtypes = [V, V, V, I, V, V, V, V, I, V, V, V, V, V, F, V]
    L.plo_line_search_by_projection_ml.argtypes = [V, V, V, I, V, V, V, V, I, V, V, V, V, V, F, F, V]
    for f in ("plo_frame_assign_grid", "plo_frame_assign_grid_lines", "plo_features_in_area", "plo_features_in_area_for_line",
              "plo_orb_search_for_initialization", "plo_orb_search_by_projection_mp", "plo_orb_search_by_projection_frame",
              "plo_line_search_by_projection_frame", "plo_line_search_by_projection_ml"):
        getattr(L, f).restype = I
    L._fs_ready = True
    return L


SCALE = np.cumprod(np.r_[np.float32(1.0), np.full(7, np.float32(1.2))]).astype(np.float32)


def _gp(P, cols=640, rows=480, distorted=False):
    if distorted:   # undistorted image bounds stick out of the sensor (Frame::ComputeImageBounds)
        return P.grid_params(cols, rows, -18.5, -11.25, cols + 21.75, rows + 14.5)
    return P.grid_params(cols, rows)


def _gpa(P, gp):
    return P._gp_array(gp)


def make_frame_pair(P, S, seed, n, cols=640, rows=480, move=6.0, nl=60):
    """Two synthetic frames with known correspondences: keypoints of frame 2 = permuted, displaced, noisy copies."""
    rng = S.SplitMix64(seed)
    d1, d2, perm = S.make_descriptor_sets(seed + 1, max(n, 1), 0.07)
    d1, d2 = d1[:n], d2[:n]
    k1 = np.zeros(n, P.KP_DTYPE)
    k1["x"] = rng.uniform(n, 2, cols - 2).astype(np.float32)
    k1["y"] = rng.uniform(n, 2, rows - 2).astype(np.float32)
    k1["octave"] = np.minimum(7, (rng.uniform(n) ** 2 * 8).astype(np.int32))   # mostly fine levels
    k1["angle"] = rng.uniform(n, 0, 360).astype(np.float32)
    k1["size"] = 31.0 * SCALE[k1["octave"]]
    k1["response"] = rng.randint(n, 20, 200).astype(np.float32)
    k1["class_id"] = -1
    k2 = k1.copy()
    if n:
        k2 = np.zeros(n, P.KP_DTYPE)
        src = perm                      # b = a[perm] (see synth.make_descriptor_sets): row j of set 2 comes from perm[j]
        k2[:] = k1[src]
        k2["x"] = (k2["x"] + rng.uniform(n, -move, move)).astype(np.float32)
        k2["y"] = (k2["y"] + rng.uniform(n, -move, move)).astype(np.float32)
        k2["angle"] = ((k2["angle"] + 10.0 + rng.uniform(n, -4, 4)) % 360).astype(np.float32)
        wrong = rng.uniform(n) < 0.1
        k2["angle"][wrong] = rng.uniform(int(wrong.sum()), 0, 360).astype(np.float32)
        k2["octave"] = np.clip(k2["octave"] + (rng.uniform(n) < 0.2) * rng.randint(n, -1, 2), 0, 7).astype(np.int32)
    # lines
    kl1 = np.zeros(nl, P.KL_DTYPE)
    sx, sy = rng.uniform(nl, 5, cols - 5), rng.uniform(nl, 5, rows - 5)
    ang, ln = rng.uniform(nl, 0, np.pi), rng.uniform(nl, 15, 220)
    ex, ey = np.clip(sx + ln * np.cos(ang), 0, cols - 1), np.clip(sy + ln * np.sin(ang), 0, rows - 1)

    def fill(kl, sx, sy, ex, ey):
        kl["startPointX"], kl["startPointY"], kl["endPointX"], kl["endPointY"] = sx, sy, ex, ey
        kl["sPointInOctaveX"], kl["sPointInOctaveY"], kl["ePointInOctaveX"], kl["ePointInOctaveY"] = sx, sy, ex, ey
        kl["lineLength"] = np.hypot(ex - sx, ey - sy).astype(np.float32)
        kl["pt_x"], kl["pt_y"] = ((kl["startPointX"] + kl["endPointX"]) / 2).astype(np.float32), \
            ((kl["startPointY"] + kl["endPointY"]) / 2).astype(np.float32)
        kl["class_id"] = np.arange(len(kl))
        fn = np.zeros((len(kl), 3))
        s = np.stack([kl["startPointX"], kl["startPointY"], np.ones(len(kl))], 1).astype(np.float64)
        e = np.stack([kl["endPointX"], kl["endPointY"], np.ones(len(kl))], 1).astype(np.float64)
        fn = np.cross(s, e)
        nrm = np.hypot(fn[:, 0], fn[:, 1])
        nrm[nrm == 0] = 1
        return fn / nrm[:, None]
    fn1 = fill(kl1, sx, sy, ex, ey)
    ld1, ld2, lperm = S.make_descriptor_sets(seed + 2, max(nl, 1), 0.06)
    ld1, ld2 = ld1[:nl], ld2[:nl]
    kl2 = kl1[lperm].copy() if nl else kl1.copy()
    if nl:
        j = rng.uniform(nl, -3, 3)
        shrink = rng.uniform(nl, 0.7, 1.0)
        mx, my = (kl2["startPointX"] + kl2["endPointX"]) / 2, (kl2["startPointY"] + kl2["endPointY"]) / 2
        hx, hy = (kl2["endPointX"] - kl2["startPointX"]) / 2 * shrink, (kl2["endPointY"] - kl2["startPointY"]) / 2 * shrink
        fn2 = fill(kl2, np.clip(mx - hx + j, 0, cols - 1), np.clip(my - hy + j, 0, rows - 1), np.clip(mx + hx + j, 0, cols - 1),
                   np.clip(my + hy + j, 0, rows - 1))
    else:
        fn2 = fn1
    f1 = dict(kps=k1, desc=d1, keylines=kl1, ldesc=ld1, linefn=fn1)
    f2 = dict(kps=k2, desc=d2, keylines=kl2, ldesc=ld2, linefn=np.ascontiguousarray(fn2))
    return f1, f2, perm, (lperm if nl else None)


def _oracle_grids(O, P, f, gp):
    L, g = _olib(O), _gpa(P, gp)
    cs, ci = np.zeros(64 * 48 + 1, np.int32), np.zeros(max(len(f["kps"]), 1), np.int32)
    L.plo_frame_assign_grid(O._p(f["kps"]), len(f["kps"]), O._p(g), O._p(cs), O._p(ci))
    nl = len(f["keylines"])
    lcs, lci = np.zeros(64 * 48 + 1, np.int32), np.zeros(max(nl, 1) * 64, np.int32)
    L.plo_frame_assign_grid_lines(O._p(f["keylines"]), nl, O._p(g), O._p(lcs), O._p(lci), len(lci))
    return (cs, ci), (lcs, lci)


def _queries_points(P, S, seed, f_last, f_cur, variant):
    """Queries as Tracking builds them: projections of the last frame's map points near their true new position."""
    rng = S.SplitMix64(seed)
    n = len(f_last["kps"])
    k = f_last["kps"]
    q = dict(valid=(rng.uniform(n) < 0.85).astype(np.uint8), desc=f_last["desc"].copy(), hasobs=(rng.uniform(n) < 0.9).astype(np.uint8))
    xy = np.stack([k["x"] + rng.uniform(n, -4, 4), k["y"] + rng.uniform(n, -4, 4)], 1).astype(np.float32)
    far = rng.uniform(n) < 0.05          # some projections fall outside the image
    xy[far] += 900
    if variant == "mp":
        q.update(xy=xy, level=k["octave"].astype(np.int32), viewcos=rng.uniform(n, 0.99, 1.0).astype(np.float32))
    else:
        q.update(uv=xy, octave=k["octave"].astype(np.int32), angle=k["angle"].astype(np.float32))
    return q


def _queries_lines(P, S, seed, f_last, variant):
    rng = S.SplitMix64(seed)
    kl = f_last["keylines"]
    nl = len(kl)
    seg = np.stack([kl["startPointX"], kl["startPointY"], kl["endPointX"], kl["endPointY"]], 1).astype(np.float32)
    seg += rng.uniform(nl * 4, -2.5, 2.5).reshape(nl, 4).astype(np.float32)
    q = dict(valid=(rng.uniform(nl) < 0.9).astype(np.uint8), seg=seg, desc=f_last["ldesc"].copy(),
             hasobs=(rng.uniform(nl) < 0.9).astype(np.uint8))
    if variant == "ml":
        q["viewcos"] = rng.uniform(nl, 0.99, 1.0).astype(np.float32)
    else:
        q["length"] = kl["lineLength"].astype(np.float32)
    return q


def _run_all(P, O, S, lib, seeds, n, nl, distorted=False):
    """Every search on a batch of frame pairs through the C ABI vs the oracle; returns the number of matches found."""
    L = _olib(O)
    gp = _gp(P, distorted=distorted)
    g = _gpa(P, gp)
    pairs = [make_frame_pair(P, S, s, n if i % 3 else max(0, n - 37 * i), nl=nl if i % 2 == 0 else max(0, nl - 11)) for i, s in enumerate(seeds)]
    lasts, curs = [p[0] for p in pairs], [p[1] for p in pairs]
    fs = P.FrameSearch(gp, SCALE, curs, lib=lib)
    (cs, ci), (lcs, lci) = fs.grids()
    total = 0
    for b, (f1, f2) in enumerate(zip(lasts, curs)):
        (rcs, rci), (rlcs, rlci) = _oracle_grids(O, P, f2, gp)
        assert (cs[b] == rcs).all() and (ci[b, :rcs[-1]] == rci[:rcs[-1]]).all(), "point grid %d" % b
        assert (lcs[b] == rlcs).all() and (lci[b, :rlcs[-1]] == rlci[:rlcs[-1]]).all(), "line grid %d" % b
    # ---- SearchForInitialization
    prev = [np.stack([f["kps"]["x"], f["kps"]["y"]], 1).astype(np.float32) for f in lasts]
    m12, cnt, pm = fs.SearchForInitialization(lasts, prev, 100, 0.9, True)
    for b, (f1, f2) in enumerate(zip(lasts, curs)):
        (rcs, rci), _ = _oracle_grids(O, P, f2, gp)
        n1 = len(f1["kps"])
        rp = prev[b].copy()
        ref = np.zeros(max(n1, 1), np.int32)
        rc = L.plo_orb_search_for_initialization(O._p(f1["kps"]), O._p(f1["desc"]), n1, O._p(f2["kps"]), O._p(f2["desc"]), len(f2["kps"]),
                                                 O._p(g), O._p(rcs), O._p(rci), O._p(rp), 100, 0.9, 1, O._p(ref))
        assert cnt[b] == rc and (m12[b, :n1] == ref[:n1]).all() and (pm[b, :n1] == rp).all(), "SearchForInitialization %d" % b
        total += rc
    # ---- ORB SearchByProjection, both forms
    for variant in ("mp", "frame"):
        qs = [_queries_points(P, S, 900 + b, f1, f2, variant) for b, (f1, f2) in enumerate(zip(lasts, curs))]
        occ0 = [(S.SplitMix64(77 + b).uniform(len(f2["kps"])) < 0.1).astype(np.uint8) for b, f2 in enumerate(curs)]
        if variant == "mp":
            asg, cnt, occ = fs.SearchByProjectionMapPoints(qs, occ0, th=3.0, nnratio=0.8)
        else:
            asg, cnt, occ = fs.SearchByProjectionLastFrame(qs, occ0, th=15.0, mode=0, checkOri=True)
        for b, (f1, f2) in enumerate(zip(lasts, curs)):
            (rcs, rci), _ = _oracle_grids(O, P, f2, gp)
            n2, q = len(f2["kps"]), qs[b]
            ro, ra = occ0[b].copy(), np.zeros(max(n2, 1), np.int32)
            if variant == "mp":
                rc = L.plo_orb_search_by_projection_mp(O._p(f2["kps"]), O._p(f2["desc"]), n2, O._p(g), O._p(rcs), O._p(rci), O._p(SCALE),
                                                       O._p(ro), len(q["valid"]), O._p(q["valid"]), O._p(q["xy"]), O._p(q["level"]),
                                                       O._p(q["viewcos"]), O._p(q["desc"]), O._p(q["hasobs"]), 3.0, 0.8, O._p(ra))
            else:
                rc = L.plo_orb_search_by_projection_frame(O._p(f2["kps"]), O._p(f2["desc"]), n2, O._p(g), O._p(rcs), O._p(rci),
                                                          O._p(SCALE), O._p(ro), len(q["valid"]), O._p(q["valid"]), O._p(q["uv"]),
                                                          O._p(q["octave"]), O._p(q["angle"]), O._p(q["desc"]), O._p(q["hasobs"]),
                                                          15.0, 0, 1, O._p(ra))
            assert cnt[b] == rc and (asg[b, :n2] == ra[:n2]).all() and (occ[b, :n2] == ro[:n2]).all(), "%s %d" % (variant, b)
            total += rc
    # ---- relocalisation form: caller's distance threshold, every assignment occupies its keypoint
    L.plo_orb_search_by_projection_kf.argtypes = [V, V, I, V, V, V, V, V, I, V, V, V, V, V, V, F, I, I, V]
    L.plo_orb_search_by_projection_kf.restype = I
    for orb_dist in (64, 100):
        qs = []
        for b, (f1, f2) in enumerate(zip(lasts, curs)):
            q = _queries_points(P, S, 930 + b, f1, f2, "frame")
            qs.append(dict(valid=q["valid"], uv=q["uv"], level=q["octave"], angle=q["angle"], desc=q["desc"],
                           hasobs=np.ones(len(q["valid"]), np.uint8)))
        occ0 = [(S.SplitMix64(99 + b).uniform(len(f2["kps"])) < 0.05).astype(np.uint8) for b, f2 in enumerate(curs)]
        asg, cnt, occ = fs.SearchByProjectionKeyFrame(qs, occ0, th=10.0, ORBdist=orb_dist, checkOri=True)
        for b, (f1, f2) in enumerate(zip(lasts, curs)):
            (rcs, rci), _ = _oracle_grids(O, P, f2, gp)
            n2, q = len(f2["kps"]), qs[b]
            ro, ra = occ0[b].copy(), np.zeros(max(n2, 1), np.int32)
            rc = L.plo_orb_search_by_projection_kf(O._p(f2["kps"]), O._p(f2["desc"]), n2, O._p(g), O._p(rcs), O._p(rci), O._p(SCALE),
                                                   O._p(ro), len(q["valid"]), O._p(q["valid"]), O._p(q["uv"]), O._p(q["level"]),
                                                   O._p(q["angle"]), O._p(q["desc"]), O._p(q["hasobs"]), 10.0, orb_dist, 1, O._p(ra))
            assert cnt[b] == rc and (asg[b, :n2] == ra[:n2]).all() and (occ[b, :n2] == ro[:n2]).all(), "reloc %d %d" % (orb_dist, b)
            total += rc
    # ---- the search inside Fuse and the loop-closing projection search
    L.plo_orb_fuse_search.argtypes = [V, V, I, V, V, V, V, V, I, V, V, V, V, F, I, V]
    L.plo_orb_fuse_search.restype = I
    L.plo_orb_search_by_projection_sim3.argtypes = [V, V, I, V, V, V, V, V, I, V, V, V, V, F, I, V]
    L.plo_orb_search_by_projection_sim3.restype = I
    INVSIG2 = (np.float32(1.0) / (SCALE * SCALE)).astype(np.float32)
    qs = []
    for b, (f1, f2) in enumerate(zip(lasts, curs)):
        q = _queries_points(P, S, 960 + b, f1, f2, "frame")
        qs.append(dict(valid=q["valid"], uv=q["uv"], level=q["octave"], desc=q["desc"], hasobs=np.ones(len(q["valid"]), np.uint8)))
    best, nf = fs.FuseSearch(qs, INVSIG2, th=3.0)
    occ0 = [(S.SplitMix64(111 + b).uniform(len(f2["kps"])) < 0.05).astype(np.uint8) for b, f2 in enumerate(curs)]
    asg, cnt, occ = fs.SearchByProjectionSim3(qs, occ0, th=10)
    for b, (f1, f2) in enumerate(zip(lasts, curs)):
        (rcs, rci), _ = _oracle_grids(O, P, f2, gp)
        n2, q = len(f2["kps"]), qs[b]
        nq = len(q["valid"])
        rb = np.zeros(max(nq, 1), np.int32)
        rc = L.plo_orb_fuse_search(O._p(f2["kps"]), O._p(f2["desc"]), n2, O._p(g), O._p(rcs), O._p(rci), O._p(SCALE), O._p(INVSIG2), nq,
                                   O._p(q["valid"]), O._p(q["uv"]), O._p(q["level"]), O._p(q["desc"]), 3.0, 50, O._p(rb))
        assert nf[b] == rc and (best[b, :nq] == rb[:nq]).all(), "fuse %d" % b
        total += rc
        ro, ra = occ0[b].copy(), np.zeros(max(n2, 1), np.int32)
        rc = L.plo_orb_search_by_projection_sim3(O._p(f2["kps"]), O._p(f2["desc"]), n2, O._p(g), O._p(rcs), O._p(rci), O._p(SCALE),
                                                 O._p(ro), nq, O._p(q["valid"]), O._p(q["uv"]), O._p(q["level"]), O._p(q["desc"]),
                                                 10.0, 50, O._p(ra))
        assert cnt[b] == rc and (asg[b, :n2] == ra[:n2]).all() and (occ[b, :n2] == ro[:n2]).all(), "sim3 %d" % b
        total += rc
    # ---- SearchBySim3: both one-way searches and the agreement check
    L.plo_orb_search_by_sim3.argtypes = [V, V, I, V, V, V, V, I, V, V, V, V, V, V, V, V, V, V, V, V, F, I, V, V, V]
    L.plo_orb_search_by_sim3.restype = I
    fs1 = P.FrameSearch(gp, SCALE, lasts, lib=lib, cap=fs.cap)
    fs2 = fs if fs.cap == fs1.cap else P.FrameSearch(gp, SCALE, curs, lib=lib, cap=fs1.cap)
    q12, q21 = [], []
    for b, (f1, f2) in enumerate(zip(lasts, curs)):
        a, c = _queries_points(P, S, 970 + b, f1, f2, "frame"), _queries_points(P, S, 980 + b, f2, f1, "frame")
        q12.append(dict(valid=a["valid"], uv=a["uv"], level=a["octave"], desc=a["desc"]))
        q21.append(dict(valid=c["valid"], uv=c["uv"], level=c["octave"], desc=c["desc"]))
    m12, nf, m1, m2 = fs1.SearchBySim3(fs2, q12, q21, th=7.5)
    for b, (f1, f2) in enumerate(zip(lasts, curs)):
        (cs1, ci1), _ = _oracle_grids(O, P, f1, gp)
        (cs2, ci2), _ = _oracle_grids(O, P, f2, gp)
        n1, n2, a, c = len(f1["kps"]), len(f2["kps"]), q12[b], q21[b]
        r1, r2, r12 = np.zeros(max(n1, 1), np.int32), np.zeros(max(n2, 1), np.int32), np.zeros(max(n1, 1), np.int32)
        rc = L.plo_orb_search_by_sim3(O._p(f1["kps"]), O._p(f1["desc"]), n1, O._p(cs1), O._p(ci1), O._p(f2["kps"]), O._p(f2["desc"]), n2,
                                      O._p(cs2), O._p(ci2), O._p(g), O._p(SCALE), O._p(a["valid"]), O._p(a["uv"]), O._p(a["level"]),
                                      O._p(a["desc"]), O._p(c["valid"]), O._p(c["uv"]), O._p(c["level"]), O._p(c["desc"]), 7.5, 100,
                                      O._p(r1), O._p(r2), O._p(r12))
        assert nf[b] == rc and (m12[b, :n1] == r12[:n1]).all() and (m1[b, :n1] == r1[:n1]).all() and \
            (m2[b, :n2] == r2[:n2]).all(), "SearchBySim3 %d" % b
        assert (m12[b, n1:] == -1).all()
        assert rc > min(n1, n2) // 8 or min(n1, n2) < 50, "SearchBySim3 agrees on too few (%d of %d)" % (rc, n1)
        total += rc
    # ---- LSD SearchByProjection, both forms
    for variant in ("ml", "frame"):
        qs = [_queries_lines(P, S, 950 + b, f1, variant) for b, f1 in enumerate(lasts)]
        occ0 = [(S.SplitMix64(88 + b).uniform(len(f2["keylines"])) < 0.1).astype(np.uint8) for b, f2 in enumerate(curs)]
        if variant == "ml":
            asg, cnt, occ = fs.LineSearchByProjectionMapLines(qs, occ0, th=3.0, nnratio=0.9)
        else:
            asg, cnt, occ = fs.LineSearchByProjectionLastFrame(qs, occ0, th=12.0)
        for b, (f1, f2) in enumerate(zip(lasts, curs)):
            _, (rlcs, rlci) = _oracle_grids(O, P, f2, gp)
            n2, q = len(f2["keylines"]), qs[b]
            ro, ra = occ0[b].copy(), np.zeros(max(n2, 1), np.int32)
            if variant == "ml":
                rc = L.plo_line_search_by_projection_ml(O._p(f2["keylines"]), O._p(f2["ldesc"]), O._p(f2["linefn"]), n2, O._p(g),
                                                        O._p(rlcs), O._p(rlci), O._p(ro), len(q["valid"]), O._p(q["valid"]),
                                                        O._p(q["seg"]), O._p(q["viewcos"]), O._p(q["desc"]), O._p(q["hasobs"]), 3.0,
                                                        0.9, O._p(ra))
            else:
                rc = L.plo_line_search_by_projection_frame(O._p(f2["keylines"]), O._p(f2["ldesc"]), O._p(f2["linefn"]), n2, O._p(g),
                                                           O._p(rlcs), O._p(rlci), O._p(ro), len(q["valid"]), O._p(q["valid"]),
                                                           O._p(q["seg"]), O._p(q["length"]), O._p(q["desc"]), O._p(q["hasobs"]), 12.0,
                                                           O._p(ra))
            assert cnt[b] == rc and (asg[b, :n2] == ra[:n2]).all() and (occ[b, :n2] == ro[:n2]).all(), "line %s %d" % (variant, b)
            total += rc
    # ---- the search inside LSDmatcher::Fuse
    L.plo_line_fuse_search.argtypes = [V, V, I, V, I, V, V, V, V, F, F, I, V]
    L.plo_line_fuse_search.restype = I
    SFL = np.ones(4, np.float32)
    qs = []
    for b, f1 in enumerate(lasts):
        q = _queries_lines(P, S, 970 + b, f1, "ml")
        qs.append(dict(valid=q["valid"], seg=q["seg"], level=np.zeros(len(q["valid"]), np.int32), desc=q["desc"]))
    best, nf = fs.LineFuseSearch(qs, SFL, th=6.0, cos_th=0.998)
    for b, f2 in enumerate(curs):
        nl2, q = len(f2["keylines"]), qs[b]
        nq = len(q["valid"])
        rb = np.zeros(max(nq, 1), np.int32)
        rc = L.plo_line_fuse_search(O._p(f2["keylines"]), O._p(f2["ldesc"]), nl2, O._p(SFL), nq, O._p(q["valid"]), O._p(q["seg"]),
                                    O._p(q["level"]), O._p(q["desc"]), 6.0, 0.998, 50, O._p(rb))
        assert nf[b] == rc and (best[b, :nq] == rb[:nq]).all(), "line fuse %d" % b
        assert rc > 0 or nl2 < 5, "line fuse finds nothing"
        total += rc
    return total


# ------------------------------------------------------------------ oracle known answers (CPU)
def test_oracle_grid_known_answers(oracle, plslam):
    P, O, L = plslam, oracle, _olib(oracle)
    gp = P.grid_params(640, 480)
    g = P._gp_array(gp)
    assert abs(gp.inv_w - 0.1) < 1e-7 and abs(gp.inv_h - 0.1) < 1e-7
    k = np.zeros(5, P.KP_DTYPE)
    k["x"] = [0.0, 4.9, 5.1, 639.0, 320.0]      # PosInGrid rounds: 4.9 -> cell 0, 5.1 -> cell 1, 639 -> 64 = out of range
    k["y"] = [0.0, 0.0, 0.0, 100.0, 474.9]      # 474.9 * 0.1 = 47.49 -> 47 (last row)
    cs, ci = np.zeros(3073, np.int32), np.zeros(5, np.int32)
    assert L.plo_frame_assign_grid(O._p(k), 5, O._p(g), O._p(cs), O._p(ci)) == 4
    cells = {c: ci[cs[c]:cs[c + 1]].tolist() for c in range(3072) if cs[c + 1] > cs[c]}
    assert cells == {0: [0, 1], 48: [2], 32 * 48 + 47: [4]}
    out = np.zeros(8, np.int32)
    # window |dx| < r, |dy| < r strictly; level band (min, max)
    assert L.plo_features_in_area(O._p(k), O._p(g), O._p(cs), O._p(ci), 2.0, 0.0, 3.0, -1, -1, O._p(out), 8) == 2
    assert out[:2].tolist() == [0, 1]
    assert L.plo_features_in_area(O._p(k), O._p(g), O._p(cs), O._p(ci), 2.0, 0.0, 2.0, -1, -1, O._p(out), 8) == 0   # |0-2| < 2 fails
    k["octave"] = [0, 3, 0, 0, 0]
    assert L.plo_features_in_area(O._p(k), O._p(g), O._p(cs), O._p(ci), 2.0, 0.0, 3.5, 0, 0, O._p(out), 8) == 2   # octave-3 keypoint 1 is out
    assert out[:2].tolist() == [0, 2]
    # a horizontal line crosses every cell of its row once (LineIterator), a point-like line occupies one cell
    kl = np.zeros(2, P.KL_DTYPE)
    kl["startPointX"], kl["startPointY"], kl["endPointX"], kl["endPointY"] = [5.0, 300.0], [105.0, 200.0], [205.0, 300.5], [105.0, 200.2]
    lcs, lci = np.zeros(3073, np.int32), np.zeros(128, np.int32)
    assert L.plo_frame_assign_grid_lines(O._p(kl), 2, O._p(g), O._p(lcs), O._p(lci), 128) == 21 + 1
    occ = [c for c in range(3072) if lcs[c + 1] > lcs[c]]
    assert occ == [ix * 48 + 10 for ix in range(0, 21)] + [30 * 48 + 20]


def test_oracle_search_recovers_correspondences(oracle, plslam, synth):
    P, O, S, L = plslam, oracle, synth, _olib(oracle)
    f1, f2, perm, _ = make_frame_pair(P, S, 5, 400, nl=0)
    gp = _gp(P)
    g = P._gp_array(gp)
    (cs, ci), _ = _oracle_grids(O, P, f2, gp)
    prev = np.stack([f1["kps"]["x"], f1["kps"]["y"]], 1).astype(np.float32)
    m = np.zeros(400, np.int32)
    c = L.plo_orb_search_for_initialization(O._p(f1["kps"]), O._p(f1["desc"]), 400, O._p(f2["kps"]), O._p(f2["desc"]), 400, O._p(g),
                                            O._p(cs), O._p(ci), O._p(prev), 100, 0.9, 1, O._p(m))
    inv = np.empty(400, np.int64)
    inv[perm] = np.arange(400)             # feature i of frame 1 sits at row inv[i] of frame 2
    ok = m >= 0
    assert c == ok.sum() and c > 60        # only level-0 keypoints take part
    assert (f1["kps"]["octave"][ok] == 0).all() and (m[ok] == inv[ok]).mean() > 0.95
    assert (prev[ok] == np.stack([f2["kps"]["x"][m[ok]], f2["kps"]["y"][m[ok]]], 1)).all()


# ------------------------------------------------------------------ HIP sources under hipemu (CPU)
def test_emu_frame_search(plslam, oracle, synth, emu_lib):
    assert _run_all(plslam, oracle, synth, emu_lib, [11, 12, 13], 260, 40) > 100
    assert _run_all(plslam, oracle, synth, emu_lib, [14, 15], 90, 25, distorted=True) > 20


def test_emu_frame_search_empty(plslam, oracle, synth, emu_lib):
    _run_all(plslam, oracle, synth, emu_lib, [21], 0, 0)


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
def test_gpu_frame_search_2000(plslam, oracle, synth):
    assert _run_all(plslam, oracle, synth, None, [31, 32, 33, 34, 35, 36], 2000, 201) > 3000
    assert _run_all(plslam, oracle, synth, None, [41, 42], 1000, 201, distorted=True) > 500


def _large_local_map(P, O, S, lib, nq_total=9000, n=700):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) with a local map larger than a frame: Tracking::SearchLocalPoints
    hands over mvpLocalMapPoints, which routinely holds more than 6000 points (the per-query LDS records exist only for the
    rotation-histogram forms).  Queries outside [0, nlevels) are skipped."""
    L = _olib(O)
    gp = _gp(P)
    g = _gpa(P, gp)
    f1, f2, _, _ = make_frame_pair(P, S, 4242, n, nl=0)
    fs = P.FrameSearch(gp, SCALE, [f2], lib=lib)
    rng = S.SplitMix64(99)
    src = (rng.uniform(nq_total) * n).astype(np.int64) % n          # map points re-observing the frame's features
    k = f1["kps"][src]
    pv = np.where(np.arange(nq_total) < 6100, 0.03, 0.6)              # most of the early queries are not in view
    q = dict(valid=(rng.uniform(nq_total) < pv).astype(np.uint8), desc=f1["desc"][src].copy(),
             hasobs=(rng.uniform(nq_total) < 0.9).astype(np.uint8),
             xy=np.stack([k["x"] + rng.uniform(nq_total, -4, 4), k["y"] + rng.uniform(nq_total, -4, 4)], 1).astype(np.float32),
             level=k["octave"].astype(np.int32), viewcos=rng.uniform(nq_total, 0.99, 1.0).astype(np.float32))
    flip = rng.uniform(nq_total) < 0.3                                # noisy descriptors so that not everything matches
    q["desc"][flip] ^= rng.randint(int(flip.sum()) * 32, 0, 256).astype(np.uint8).reshape(-1, 32) & 0x11
    occ0 = (S.SplitMix64(5).uniform(n) < 0.1).astype(np.uint8)
    asg, cnt, occ = fs.SearchByProjectionMapPoints([q], [occ0], th=3.0, nnratio=0.8)
    (rcs, rci), _ = _oracle_grids(O, P, f2, gp)
    ro, ra = occ0.copy(), np.zeros(n, np.int32)
    rc = L.plo_orb_search_by_projection_mp(O._p(f2["kps"]), O._p(f2["desc"]), n, O._p(g), O._p(rcs), O._p(rci), O._p(SCALE), O._p(ro),
                                           nq_total, O._p(q["valid"]), O._p(q["xy"]), O._p(q["level"]), O._p(q["viewcos"]),
                                           O._p(q["desc"]), O._p(q["hasobs"]), 3.0, 0.8, O._p(ra))
    assert cnt[0] == rc and (asg[0, :n] == ra).all() and (occ[0, :n] == ro).all()
    assert rc > 100 and ra.max() >= 6000, "the late queries must take part (max assigned query %d)" % ra.max()
    # out-of-range predicted levels are skipped, exactly as if the caller had cleared their valid flag
    q2 = dict(q)
    q2["level"] = q["level"].copy()
    bad = rng.uniform(nq_total) < 0.2
    q2["level"][bad] = np.where(rng.uniform(int(bad.sum())) < 0.5, -1, len(SCALE)).astype(np.int32)
    q3 = dict(q2)
    q3["valid"] = (q["valid"] * ~bad).astype(np.uint8)
    q3["level"] = np.where(bad, 0, q["level"]).astype(np.int32)
    a2, c2, o2 = fs.SearchByProjectionMapPoints([q2], [occ0], th=3.0, nnratio=0.8)
    a3, c3, o3 = fs.SearchByProjectionMapPoints([q3], [occ0], th=3.0, nnratio=0.8)
    assert c2[0] == c3[0] and (a2 == a3).all() and (o2 == o3).all()


def test_emu_large_local_map(plslam, oracle, synth, emu_lib):
    _large_local_map(plslam, oracle, synth, emu_lib, nq_total=7000, n=300)


@pytest.mark.gpu
def test_gpu_large_local_map(plslam, oracle, synth):
    _large_local_map(plslam, oracle, synth, None, nq_total=20000, n=2000)


def _host_buffer_forms(P, O, S, lib, n=1500, nl=180):
    """The one-call-per-reference-call entry points (host buffers through the calling thread's staging arena, grid rebuilt inside)
    agree with the oracle -- and so do the RESIDENT forms (round 6: plh_frame_points / plh_frame_lines hold keypoints, descriptors
    and grid on the device; a search uploads its queries only), called twice on the same handles."""
    L = _olib(O)
    H = P.load(lib)
    gp = _gp(P)
    g = _gpa(P, gp)
    f1, f2, _, _ = make_frame_pair(P, S, 51, n, nl=nl)
    (cs, ci), (lcs, lci) = _oracle_grids(O, P, f2, gp)
    n1, n2, nl = len(f1["kps"]), len(f2["kps"]), len(f2["keylines"])
    p = P._p
    # resident handles of both frames
    H.plh_frame_points_create.argtypes = [V, V, I, V, I, V]
    H.plh_frame_lines_create.argtypes = [V, V, V, I, V, I, V]
    H.plh_frame_points_destroy.argtypes = [V]
    H.plh_frame_lines_destroy.argtypes = [V]
    H.plh_frame_points_count.argtypes = [V]
    R1, R2, RL2, RL1 = V(), V(), V(), V()
    P._check(H, H.plh_frame_points_create(p(f1["kps"]), p(f1["desc"]), n1, C.byref(gp), 0, C.byref(R1)), "resident points 1")
    P._check(H, H.plh_frame_points_create(p(f2["kps"]), p(f2["desc"]), n2, C.byref(gp), 0, C.byref(R2)), "resident points 2")
    P._check(H, H.plh_frame_lines_create(p(f2["keylines"]), p(f2["ldesc"]), p(f2["linefn"]), nl, C.byref(gp), 0, C.byref(RL2)), "resident lines 2")
    P._check(H, H.plh_frame_lines_create(p(f1["keylines"]), p(f1["ldesc"]), p(f1["linefn"]), len(f1["keylines"]), C.byref(gp), 0, C.byref(RL1)),
             "resident lines 1")
    assert H.plh_frame_points_count(R2) == n2
    cnt = C.c_int(0)
    # SearchForInitialization
    prev = np.stack([f1["kps"]["x"], f1["kps"]["y"]], 1).astype(np.float32)
    rp, ref = prev.copy(), np.zeros(n1, np.int32)
    rc = L.plo_orb_search_for_initialization(O._p(f1["kps"]), O._p(f1["desc"]), n1, O._p(f2["kps"]), O._p(f2["desc"]), n2, O._p(g),
                                             O._p(cs), O._p(ci), O._p(rp), 100, 0.9, 1, O._p(ref))
    H.plh_orb_search_for_initialization.argtypes = [V, V, I, V, V, I, V, V, I, F, I, V, V, I]
    H.plh_orb_search_for_initialization_resident.argtypes = [V, V, V, I, F, I, V, V]
    for form in ("host", "resident", "resident"):
        got, pm = np.full(n1, 7, np.int32), prev.copy()
        if form == "host":
            P._check(H, H.plh_orb_search_for_initialization(p(f1["kps"]), p(f1["desc"]), n1, p(f2["kps"]), p(f2["desc"]), n2, C.byref(gp),
                                                            p(pm), 100, 0.9, 1, p(got), C.byref(cnt), 0), "init")
        else:
            P._check(H, H.plh_orb_search_for_initialization_resident(R1, R2, p(pm), 100, 0.9, 1, p(got), C.byref(cnt)), "init resident")
        assert cnt.value == rc and (got == ref).all() and (pm == rp).all() and rc > n // 15, form
    # ORB SearchByProjection(Cur, Last)
    q = _queries_points(P, S, 901, f1, f2, "frame")
    ro, ra = np.zeros(n2, np.uint8), np.zeros(n2, np.int32)
    rc = L.plo_orb_search_by_projection_frame(O._p(f2["kps"]), O._p(f2["desc"]), n2, O._p(g), O._p(cs), O._p(ci), O._p(SCALE), O._p(ro),
                                              n1, O._p(q["valid"]), O._p(q["uv"]), O._p(q["octave"]), O._p(q["angle"]), O._p(q["desc"]),
                                              O._p(q["hasobs"]), 15.0, 0, 1, O._p(ra))
    H.plh_orb_search_by_projection_frame.argtypes = [V, V, I, V, V, I, V, I, V, V, V, V, V, V, F, I, I, V, V, I]
    H.plh_orb_search_by_projection_frame_resident.argtypes = [V, V, I, V, I, V, V, V, V, V, V, F, I, I, V, V]
    for form in ("host", "resident", "resident"):
        got, occ = np.full(n2, 7, np.int32), np.zeros(n2, np.uint8)
        if form == "host":
            P._check(H, H.plh_orb_search_by_projection_frame(p(f2["kps"]), p(f2["desc"]), n2, C.byref(gp), p(SCALE), len(SCALE), p(occ), n1,
                                                             p(q["valid"]), p(q["uv"]), p(q["octave"]), p(q["angle"]), p(q["desc"]),
                                                             p(q["hasobs"]), 15.0, 0, 1, p(got), C.byref(cnt), 0), "proj frame")
        else:
            P._check(H, H.plh_orb_search_by_projection_frame_resident(R2, p(SCALE), len(SCALE), p(occ), n1, p(q["valid"]), p(q["uv"]),
                                                                      p(q["octave"]), p(q["angle"]), p(q["desc"]), p(q["hasobs"]), 15.0, 0, 1,
                                                                      p(got), C.byref(cnt)), "proj frame resident")
        assert cnt.value == rc and (got == ra).all() and (occ == ro).all() and rc > n // 5, form
    # ... and with the projection on the device (plh_orb_search_by_projection_frame_resident_world): world points behind and in front of a
    # rotated pose; the oracle's projection (form 0) + the oracle's search on its (valid && front, uv) is the reference
    rng = S.SplitMix64(905)
    ang = 0.05
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = np.array([0.05, -0.02, 0.1], np.float32)
    K4 = [520.0, 518.0, 320.0, 240.0]
    z = rng.uniform(n1, 1.5, 9.0).astype(np.float32)
    z[rng.uniform(n1) < 0.05] *= -1                      # some behind the camera: invzc < 0
    uvw = q["uv"].astype(np.float64)
    cam = np.stack([(uvw[:, 0] - K4[2]) / K4[0] * z, (uvw[:, 1] - K4[3]) / K4[1] * z, z], 1)
    world = ((cam - tcw.astype(np.float64)) @ Rcw.astype(np.float64)).astype(np.float32)   # Rcw^T (cam - tcw)
    view = np.concatenate([Rcw.reshape(9), tcw, np.zeros(3), K4, [0, 0, 640, 480], [np.log(np.float32(1.2))]]).astype(np.float32)
    L.plo_frame_project_points.argtypes = [V, I, I, V, V, V]
    front, uv2 = np.zeros(max(n1, 1), np.uint8), np.zeros((max(n1, 1), 2), np.float32)
    L.plo_frame_project_points(O._p(view), 0, n1, O._p(world), O._p(front), O._p(uv2))
    v2 = (q["valid"] & front[:n1]).astype(np.uint8)
    ro, ra = np.zeros(n2, np.uint8), np.zeros(n2, np.int32)
    rc = L.plo_orb_search_by_projection_frame(O._p(f2["kps"]), O._p(f2["desc"]), n2, O._p(g), O._p(cs), O._p(ci), O._p(SCALE), O._p(ro),
                                              n1, O._p(v2), O._p(uv2), O._p(q["octave"]), O._p(q["angle"]), O._p(q["desc"]),
                                              O._p(q["hasobs"]), 15.0, 0, 1, O._p(ra))
    vrec = np.zeros(1, P.VIEW_DTYPE)
    vrec["Rcw"][0], vrec["tcw"][0] = Rcw.reshape(9), tcw
    vrec["fx"], vrec["fy"], vrec["cx"], vrec["cy"] = K4
    vrec["max_x"], vrec["max_y"], vrec["log_scale_factor"], vrec["n_scale_levels"] = 640, 480, np.log(np.float32(1.2)), len(SCALE)
    H.plh_orb_search_by_projection_frame_resident_world.argtypes = [V, V, I, V, I, V, V, V, V, V, V, V, F, I, I, V, V]
    got, occ = np.full(n2, 7, np.int32), np.zeros(n2, np.uint8)
    P._check(H, H.plh_orb_search_by_projection_frame_resident_world(R2, p(SCALE), len(SCALE), p(occ), n1, p(vrec), p(q["valid"]), p(world),
                                                                    p(q["octave"]), p(q["angle"]), p(q["desc"]), p(q["hasobs"]), 15.0, 0, 1,
                                                                    p(got), C.byref(cnt)), "proj frame resident world")
    assert cnt.value == rc and (got == ra).all() and (occ == ro).all() and rc > n // 8 and 0 < int(front[:n1].sum()) < n1
    # ORB SearchByProjection(F, MapPoints)
    q = _queries_points(P, S, 903, f1, f2, "mp")
    occ0 = (S.SplitMix64(6).uniform(n2) < 0.1).astype(np.uint8)
    ro, ra = occ0.copy(), np.zeros(n2, np.int32)
    rc = L.plo_orb_search_by_projection_mp(O._p(f2["kps"]), O._p(f2["desc"]), n2, O._p(g), O._p(cs), O._p(ci), O._p(SCALE), O._p(ro), n1,
                                           O._p(q["valid"]), O._p(q["xy"]), O._p(q["level"]), O._p(q["viewcos"]), O._p(q["desc"]),
                                           O._p(q["hasobs"]), 3.0, 0.8, O._p(ra))
    H.plh_orb_search_by_projection_mp.argtypes = [V, V, I, V, V, I, V, I, V, V, V, V, V, V, F, F, V, V, I]
    H.plh_orb_search_by_projection_mp_resident.argtypes = [V, V, I, V, I, V, V, V, V, V, V, F, F, V, V]
    for form in ("host", "resident"):
        got, occ = np.full(n2, 7, np.int32), occ0.copy()
        if form == "host":
            P._check(H, H.plh_orb_search_by_projection_mp(p(f2["kps"]), p(f2["desc"]), n2, C.byref(gp), p(SCALE), len(SCALE), p(occ), n1,
                                                          p(q["valid"]), p(q["xy"]), p(q["level"]), p(q["viewcos"]), p(q["desc"]), p(q["hasobs"]),
                                                          3.0, 0.8, p(got), C.byref(cnt), 0), "proj mp")
        else:
            P._check(H, H.plh_orb_search_by_projection_mp_resident(R2, p(SCALE), len(SCALE), p(occ), n1, p(q["valid"]), p(q["xy"]), p(q["level"]),
                                                                   p(q["viewcos"]), p(q["desc"]), p(q["hasobs"]), 3.0, 0.8, p(got),
                                                                   C.byref(cnt)), "proj mp resident")
        assert cnt.value == rc and (got == ra).all() and (occ == ro).all() and rc > n // 10, form
    # LSD SearchByProjection(F, MapLines) and (Cur, Last)
    H.plh_line_search_by_projection_ml.argtypes = [V, V, V, I, V, V, I, V, V, V, V, V, F, F, V, V, I]
    H.plh_line_search_by_projection_ml_resident.argtypes = [V, V, I, V, V, V, V, V, F, F, V, V]
    H.plh_line_search_by_projection_frame.argtypes = [V, V, V, I, V, V, I, V, V, V, V, V, F, V, V, I]
    H.plh_line_search_by_projection_frame_resident.argtypes = [V, V, I, V, V, V, V, V, F, V, V]
    for variant in ("ml", "frame"):
        q = _queries_lines(P, S, 951, f1, variant)
        aux = q["viewcos"] if variant == "ml" else q["length"]
        ro, ra = np.zeros(nl, np.uint8), np.zeros(nl, np.int32)
        if variant == "ml":
            rc = L.plo_line_search_by_projection_ml(O._p(f2["keylines"]), O._p(f2["ldesc"]), O._p(f2["linefn"]), nl, O._p(g), O._p(lcs),
                                                    O._p(lci), O._p(ro), len(q["valid"]), O._p(q["valid"]), O._p(q["seg"]), O._p(aux),
                                                    O._p(q["desc"]), O._p(q["hasobs"]), 3.0, 0.9, O._p(ra))
        else:
            rc = L.plo_line_search_by_projection_frame(O._p(f2["keylines"]), O._p(f2["ldesc"]), O._p(f2["linefn"]), nl, O._p(g), O._p(lcs),
                                                       O._p(lci), O._p(ro), len(q["valid"]), O._p(q["valid"]), O._p(q["seg"]), O._p(aux),
                                                       O._p(q["desc"]), O._p(q["hasobs"]), 12.0, O._p(ra))
        for form in ("host", "resident", "resident"):
            got, occ = np.full(nl, 7, np.int32), np.zeros(nl, np.uint8)
            a = (p(f2["keylines"]), p(f2["ldesc"]), p(f2["linefn"]), nl, C.byref(gp)) if form == "host" else (RL2,)
            qa = (p(occ), len(q["valid"]), p(q["valid"]), p(q["seg"]), p(aux), p(q["desc"]), p(q["hasobs"]))
            if variant == "ml":
                fn = H.plh_line_search_by_projection_ml if form == "host" else H.plh_line_search_by_projection_ml_resident
                P._check(H, fn(*a, *qa, 3.0, 0.9, p(got), C.byref(cnt), *((0,) if form == "host" else ())), "line ml " + form)
            else:
                fn = H.plh_line_search_by_projection_frame if form == "host" else H.plh_line_search_by_projection_frame_resident
                P._check(H, fn(*a, *qa, 12.0, p(got), C.byref(cnt), *((0,) if form == "host" else ())), "line frame " + form)
            assert cnt.value == rc and (got == ra).all() and (occ == ro).all() and rc > nl // 8, (variant, form)
    # LSDmatcher::SearchDouble: resident == host form (the host form is held to the reference by tests/test_ref_lsdmatcher.py)
    H.plh_line_search_double.argtypes = [V, I, V, I, F, F, V, V, I]
    H.plh_line_search_double_resident.argtypes = [V, V, F, F, V, V]
    nl1 = len(f1["keylines"])
    mh, mr, ch, cr = np.zeros(max(nl1, 1), np.int32), np.full(max(nl1, 1), 7, np.int32), C.c_int(0), C.c_int(0)
    P._check(H, H.plh_line_search_double(p(f1["ldesc"]), nl1, p(f2["ldesc"]), nl, 50.0, 0.7, p(mh), C.byref(ch), 0), "search double")
    P._check(H, H.plh_line_search_double_resident(RL1, RL2, 50.0, 0.7, p(mr), C.byref(cr)), "search double resident")
    assert ch.value == cr.value and (mh[:nl1] == mr[:nl1]).all() and ch.value > nl // 8
    # ORBmatcher::SearchByBoW(KF, F): resident == host form (held to the reference by tests/test_ref_orbmatcher.py)
    rng = S.SplitMix64(77)
    node1 = rng.randint(n1, 0, 60).astype(np.int32)
    node2 = rng.randint(n2, 0, 60).astype(np.int32)
    node2[rng.uniform(n2) < 0.05] = -1
    valid1 = (rng.uniform(n1) < 0.8).astype(np.uint8)
    H.plh_frame_points_set_nodes.argtypes = [V, V]
    P._check(H, H.plh_frame_points_set_nodes(R1, p(node1)), "set nodes 1")
    P._check(H, H.plh_frame_points_set_nodes(R2, p(node2)), "set nodes 2")
    H.plh_orb_search_by_bow.argtypes = [V, V, V, V, I, V, V, V, I, I, F, I, V, V, I]
    H.plh_orb_search_by_bow_resident.argtypes = [V, V, V, I, F, I, V, V]
    a1, a2 = np.ascontiguousarray(f1["kps"]["angle"]), np.ascontiguousarray(f2["kps"]["angle"])
    mh, mr = np.zeros(n2, np.int32), np.full(n2, 7, np.int32)
    P._check(H, H.plh_orb_search_by_bow(p(f1["desc"]), p(a1), p(node1), p(valid1), n1, p(f2["desc"]), p(a2), p(node2), n2, 50, 0.7, 1, p(mh),
                                        C.byref(ch), 0), "bow")
    P._check(H, H.plh_orb_search_by_bow_resident(R1, p(valid1), R2, 50, 0.7, 1, p(mr), C.byref(cr)), "bow resident")
    assert ch.value == cr.value and (mh == mr).all()
    for h in (R1, R2):
        H.plh_frame_points_destroy(h)
    for h in (RL1, RL2):
        H.plh_frame_lines_destroy(h)


def test_emu_host_buffer_and_resident_forms(plslam, oracle, synth, emu_lib):
    _host_buffer_forms(plslam, oracle, synth, emu_lib, n=500, nl=90)


@pytest.mark.gpu
def test_gpu_host_buffer_forms(plslam, oracle, synth):
    _host_buffer_forms(plslam, oracle, synth, None)


# ------------------------------------------------------------------ UndistortKeyPoints / ComputeDistinctiveDescriptors
TUM1_K = [517.306408, 516.469215, 318.643040, 255.313989]
TUM1_D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]


def _check_post(P, O, S, lib, nk, sizes):
    L = O.lib()
    L.plo_undistort_keypoints.argtypes = [V, I, V, V, V]
    L.plo_distinctive_descriptor.argtypes = [V, I]
    L.plo_distinctive_descriptor.restype = I
    Kf, Df = np.asarray(TUM1_K, np.float32), np.asarray(TUM1_D, np.float32)
    frames = [make_frame_pair(P, S, 60 + i, n, nl=0)[0]["kps"] for i, n in enumerate(nk)]
    for D in (Df, np.zeros(5, np.float32)):
        got = P.undistort_keypoints(frames, Kf, D, lib=lib)
        for k, g in zip(frames, got):
            ref = np.zeros(len(k), P.KP_DTYPE)
            L.plo_undistort_keypoints(O._p(k), len(k), O._p(Kf), O._p(D), O._p(ref))
            assert all((g[f] == ref[f]).all() for f in ref.dtype.names)
            if D[0] != 0 and len(k) > 50:
                assert np.abs(g["x"] - k["x"]).max() > 0.5      # the distortion really moves border points
    rng = S.SplitMix64(5)
    sets = []
    for n in sizes:
        base = S.make_descriptor_sets(200 + n, 1, 0.0)[0][0]
        flips = (rng.uniform(n * 256) < 0.1).reshape(n, 256)
        sets.append(base[None, :] ^ np.packbits(flips, axis=1, bitorder="little") if n else np.zeros((0, 32), np.uint8))
    got = P.distinctive_descriptors(sets, lib=lib)
    for s, g in zip(sets, got):
        s = np.ascontiguousarray(s)
        assert g == L.plo_distinctive_descriptor(O._p(s), len(s))


def test_emu_undistort_and_distinctive(plslam, oracle, synth, emu_lib):
    _check_post(plslam, oracle, synth, emu_lib, [150, 0, 37], [1, 2, 3, 8, 0, 33, 70])


@pytest.mark.gpu
def test_gpu_undistort_and_distinctive(plslam, oracle, synth):
    _check_post(plslam, oracle, synth, None, [2000, 1000, 0, 1], [1, 2, 3, 5, 8, 0, 33, 64, 65, 200, 700])


# ------------------------------------------------------------------ round 6: prepass + ordered resolve against the one-wavefront kernels
def _contention_queries(P, S, seed, f_last, f_cur, variant, dup, spread):
    """Queries that fight for the same keypoints: every last-frame feature `dup` times (jittered by `spread` pixels), a third of the
    map elements without observations (a later query overwrites their assignment), descriptors lightly corrupted so that the ratio
    test of the map-point form decides both ways."""
    rng = S.SplitMix64(seed)
    q0 = _queries_points(P, S, seed + 1, f_last, f_cur, variant)
    n = len(q0["valid"])
    idx = np.concatenate([np.arange(n)] * dup)
    rng_perm = np.argsort(rng.uniform(len(idx)), kind="stable")
    idx = idx[rng_perm]
    q = {k: np.ascontiguousarray(v[idx]) for k, v in q0.items()}
    key = "xy" if variant == "mp" else "uv"
    q[key] = (q[key] + rng.uniform(len(idx) * 2, -spread, spread).reshape(-1, 2)).astype(np.float32)
    q["hasobs"] = (rng.uniform(len(idx)) < 0.66).astype(np.uint8)
    flip = rng.uniform(len(idx)) < 0.5
    q["desc"][flip] ^= (rng.randint(int(flip.sum()) * 32, 0, 256).astype(np.uint8).reshape(-1, 32) & 0x03)
    return q


def _parallel_equals_serial(P, O, S, lib, n, nl, dups):
    H = P.load(lib)
    H.plh_debug_set_proj_serial.argtypes = [I]
    gp = _gp(P)
    try:
        for seed, dup in dups:
            f1, f2, _, _ = make_frame_pair(P, S, seed, n, nl=nl, move=3.0)
            fs = P.FrameSearch(gp, SCALE, [f2, f2], lib=lib)
            occ0 = (S.SplitMix64(seed + 9).uniform(len(f2["kps"])) < 0.15).astype(np.uint8)
            locc0 = (S.SplitMix64(seed + 10).uniform(len(f2["keylines"])) < 0.15).astype(np.uint8)
            qm = [_contention_queries(P, S, seed + 20 + k, f1, f2, "mp", dup, 3.0 + 4 * k) for k in range(2)]
            qf = [_contention_queries(P, S, seed + 30 + k, f1, f2, "frame", dup, 2.0 + 6 * k) for k in range(2)]
            ql = []
            for k in range(2):
                base = _queries_lines(P, S, seed + 40 + k, f1, "ml")
                rep = np.concatenate([np.arange(len(base["valid"]))] * dup)
                qq = {kk: np.ascontiguousarray(v[rep]) for kk, v in base.items()}
                qq["hasobs"] = (S.SplitMix64(seed + 50 + k).uniform(len(rep)) < 0.6).astype(np.uint8)
                qq["length"] = np.ascontiguousarray(f1["keylines"]["lineLength"][rep].astype(np.float32))
                ql.append(qq)
            res = {}
            for serial in (1, 0, 2):   # 2: candidate lists of two entries -- contended queries run out of list and take the slow path
                H.plh_debug_set_proj_serial(serial)
                out = [fs.SearchByProjectionMapPoints(qm, [occ0, occ0], th=3.0, nnratio=0.8),
                       fs.SearchByProjectionMapPoints(qm, [occ0, occ0], th=1.0, nnratio=0.9),
                       fs.SearchByProjectionLastFrame(qf, [occ0, occ0], th=15.0, mode=0, checkOri=True),
                       fs.SearchByProjectionLastFrame(qf, [occ0, occ0], th=7.0, mode=1, checkOri=False),
                       fs.SearchByProjectionKeyFrame([dict(q, level=q["octave"]) for q in qf], [occ0, occ0], th=10.0, ORBdist=64, checkOri=True),
                       fs.SearchByProjectionSim3([dict(q, level=q["octave"]) for q in qf], [occ0, occ0], th=10, TH_LOW=50),
                       fs.FuseSearch([dict(q, level=q["octave"]) for q in qf], np.float32(1.0) / (SCALE * SCALE), th=3.0, TH_LOW=50),
                       fs.LineSearchByProjectionMapLines(ql, [locc0, locc0], th=3.0, nnratio=0.9),
                       fs.LineSearchByProjectionLastFrame(ql, [locc0, locc0], th=12.0)]
                res[serial] = out
            for mode in (0, 2):
                for k, (a, b) in enumerate(zip(res[1], res[mode])):
                    for u, v in zip(a, b):
                        assert (np.asarray(u) == np.asarray(v)).all(), ("search %d, seed %d, %d-fold queries: prepass + resolve (mode %d) differs from "
                                                                        "the sequential kernel" % (k, seed, dup, mode))
            assert res[0][0][1].min() > 0 and res[0][2][1].min() > 0 and res[0][7][1].min() > 0
            # the line prepass has a small-launch form (eight lanes per query, at most 4096 queries per launch) and a batch form (one lane
            # per query): the same frame and queries as a launch of enough pairs for the batch form, against the two-pair launch above
            if dup == dups[-1][1]:
                H.plh_debug_set_proj_serial(0)
                npairs = 4096 // max(len(ql[0]["valid"]), 1) + 2
                fsn = P.FrameSearch(gp, SCALE, [f2] * npairs, lib=lib)
                big = [fsn.LineSearchByProjectionMapLines([ql[0]] * npairs, [locc0] * npairs, th=3.0, nnratio=0.9),
                       fsn.LineSearchByProjectionLastFrame([ql[0]] * npairs, [locc0] * npairs, th=12.0)]
                for k, (got, ref) in enumerate(zip(big, (res[0][7], res[0][8]))):
                    for u, v in zip(got, ref):
                        u, v = np.asarray(u), np.asarray(v)
                        assert all((u[b] == v[0]).all() for b in range(npairs)), "line search %d: batch form differs from the small-launch form" % k
    finally:
        H.plh_debug_set_proj_serial(0)


def test_emu_proj_prepass_resolve_equals_sequential(plslam, oracle, synth, emu_lib):
    _parallel_equals_serial(plslam, oracle, synth, emu_lib, 300, 60, [(71, 1), (72, 5), (73, 13)])


@pytest.mark.gpu
def test_gpu_proj_prepass_resolve_equals_sequential(plslam, oracle, synth):
    _parallel_equals_serial(plslam, oracle, synth, None, 900, 200, [(81, 1), (82, 4), (83, 13)])   # (13 x 900 < 12000: the bound of the forms with a rotation histogram)
