"""Hamming matching: oracle known answers (CPU), kernel sources under hipemu (CPU), and GPU parity."""
import numpy as np
import pytest

import _util


def _cases(synth, seed, n1, n2, flip=0.08):
    a, b, perm = synth.make_descriptor_sets(seed, max(n1, n2), flip)
    return a[:n1].copy(), b[:n2].copy()


def _bow_sets(synth, seed, n, nodes=100, valid_p=0.8):
    a, b, perm = synth.make_descriptor_sets(seed, n, 0.06)
    rng = synth.SplitMix64(seed + 7)
    node_a = rng.randint(n, 0, nodes).astype(np.int32)
    node_b = node_a[perm].copy()
    # a few features land in other nodes / have no word
    flip = rng.uniform(n) < 0.1
    node_b[flip] = rng.randint(int(flip.sum()), 0, nodes).astype(np.int32)
    node_b[rng.uniform(n) < 0.02] = -1
    ang_a = rng.uniform(n, 0, 360).astype(np.float32)
    ang_b = ((ang_a[perm] + 12.0 + rng.uniform(n, -3, 3)) % 360).astype(np.float32)
    wrong = rng.uniform(n) < 0.15
    ang_b[wrong] = rng.uniform(int(wrong.sum()), 0, 360).astype(np.float32)
    valid = (rng.uniform(n) < valid_p).astype(np.uint8)
    kf = dict(desc=a, angle=ang_a, node=node_a, valid=valid)
    fr = dict(desc=b, angle=ang_b, node=node_b)
    return kf, fr


def _bow_sets_contended(synth, seed, n, nodes, protos=5):
    """As _bow_sets, with every descriptor one of `protos` prototypes + at most two flipped bits: under a node many KeyFrame features
    want the same few Frame features at equal or nearly equal distances -- the first-come claims decide, the candidate lists of the
    prepass run out (round 6: k_search_by_bow's lane-per-group resolve and its exact fallback), ties go to the first in order."""
    kf, fr = _bow_sets(synth, seed, n, nodes, valid_p=0.9)
    rng = synth.SplitMix64(seed + 77)
    base = kf["desc"][:protos].copy()
    for d in (kf["desc"], fr["desc"]):
        pick = rng.randint(len(d), 0, protos)
        d[:] = base[pick]
        for _ in range(2):
            hit = rng.uniform(len(d)) < 0.5
            byte, bit = rng.randint(len(d), 0, 32), rng.randint(len(d), 0, 8)
            d[np.arange(len(d))[hit], byte[hit]] ^= (1 << bit[hit]).astype(np.uint8)
    return kf, fr


def _oracle_bow(O, kf, fr, th_low, nnratio, check):
    n1, n2 = len(kf["desc"]), len(fr["desc"])
    out = np.zeros(max(n2, 1), np.int32)
    p = O._p
    c = O.lib().plo_orb_search_by_bow(p(kf["desc"]), p(kf["angle"]), p(kf["node"]), p(kf["valid"]), n1, p(fr["desc"]),
                                      p(fr["angle"]), p(fr["node"]), n2, th_low, nnratio, int(check), p(out))
    return c, out[:n2]


def _oracle_double(O, d1, d2, th, ratio):
    m = np.zeros(max(len(d1), 1), np.int32)
    c = O.lib().plo_line_search_double(O._p(d1), len(d1), O._p(d2), len(d2), th, ratio, O._p(m))
    return c, m[:len(d1)]


# ------------------------------------------------------------------ oracle known answers (CPU)
def test_descriptor_distance_known(oracle):
    L = oracle.lib()
    z, f = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert L.plo_descriptor_distance(oracle._p(z), oracle._p(f)) == 256
    assert L.plo_descriptor_distance(oracle._p(z), oracle._p(z)) == 0
    a = np.zeros(32, np.uint8); a[3] = 0b10110000; a[31] = 1
    assert L.plo_descriptor_distance(oracle._p(a), oracle._p(z)) == 4
    rng = np.random.default_rng(0)
    x, y = rng.integers(0, 256, (2, 32)).astype(np.uint8)
    assert L.plo_descriptor_distance(oracle._p(x), oracle._p(y)) == int(np.unpackbits(x ^ y).sum())


def test_knn2_ties_and_bruteforce(oracle):
    q = np.zeros((1, 32), np.uint8)
    t = np.zeros((4, 32), np.uint8); t[0, 0] = 3; t[1, 0] = 1; t[2, 0] = 2; t[3, 0] = 4   # distances 2,1,1,1
    idx, dist = oracle.knn2(q, t)
    assert idx.tolist() == [[1, 2]] and dist.tolist() == [[1, 1]]     # ties keep the lower train index first
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (50, 32)).astype(np.uint8); t = rng.integers(0, 256, (70, 32)).astype(np.uint8)
    idx, dist = oracle.knn2(q, t)
    full = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
    assert (np.sort(full, axis=1)[:, :2] == dist).all()
    assert (full[np.arange(50), idx[:, 0]] == dist[:, 0]).all()


def test_search_double_recovers_permutation(oracle, synth):
    a, b, perm = synth.make_descriptor_sets(103, 200, 0.08)
    c, m = _oracle_double(oracle, a, b, 50.0, 0.7)
    inv = np.empty(200, np.int64); inv[perm] = np.arange(200)      # a[i] corresponds to b[inv[i]]
    ok = m >= 0
    assert c == ok.sum() and c > 100
    assert (m[ok] == inv[ok]).all()


# ------------------------------------------------------------------ HIP sources under hipemu (CPU)
def test_emu_knn2(plslam, oracle, synth, emu_lib):
    a, b = _cases(synth, 100, 300, 333)
    idx, dist = plslam.hamming_knn2(a, b, lib=emu_lib)
    ri, rd = oracle.knn2(a, b)
    assert (idx == ri).all() and (dist == rd).all()
    idx, dist = plslam.hamming_knn2(a[:5], b[:1], lib=emu_lib)       # nt < 2: second slot empty
    assert (idx[:, 1] == -1).all() and (dist[:, 1] == 2**31 - 1).all()


def test_emu_line_search_double(plslam, oracle, synth, emu_lib):
    m = plslam.LSDmatcher(0.7, True, lib=emu_lib)
    sets1, sets2 = [], []
    for k, (n1, n2) in enumerate([(200, 201), (57, 120), (3, 2), (0, 10), (10, 0), (1, 1)]):
        a, b = _cases(synth, 110 + k, n1, n2)
        sets1.append(a); sets2.append(b)
    got, cnt = m.SearchDoubleBatch(sets1, sets2)
    for p, (a, b) in enumerate(zip(sets1, sets2)):
        c, ref = _oracle_double(oracle, a, b, 50.0, 0.7)
        assert cnt[p] == c and (got[p, :len(a)] == ref).all(), p
    a, b = _cases(synth, 130, 64, 80)
    ref = np.zeros(64, np.int32)
    oracle.lib().plo_line_bfmatch(oracle._p(a), 64, oracle._p(b), 80, 50.0, 0.7, oracle._p(ref))
    assert (m.FrameBFMatch(a, b) == ref).all()
    # SerachForInitialize (LSDmatcher.cpp:340-373): only the gap test of FrameBFMatch -- both thresholds at infinity
    oracle.lib().plo_line_bfmatch(oracle._p(a), 64, oracle._p(b), 80, float("inf"), float("inf"), oracle._p(ref))
    c, got = m.SerachForInitialize(a, b)
    assert (got == ref).all() and c == int((ref >= 0).sum()) and c >= int((m.FrameBFMatch(a, b) >= 0).sum())


def test_emu_search_by_bow(plslam, oracle, synth, emu_lib):
    om = plslam.ORBmatcher(0.7, True, lib=emu_lib)
    for seed, n, nodes in [(200, 300, 20), (201, 150, 5), (202, 64, 64), (203, 200, 2)]:   # last: > 64 candidates per node
        kf, fr = _bow_sets(synth, seed, n, nodes)
        c, got = om.SearchByBoW(kf, fr)
        rc, ref = _oracle_bow(oracle, kf, fr, 50, 0.7, True)
        assert c == rc and (got == ref).all() and c > 10
    for seed, n, nodes, nn in [(210, 300, 12, 0.7), (211, 300, 12, 1.01), (212, 120, 1, 1.01), (213, 500, 40, 1.01)]:
        kf, fr = _bow_sets_contended(synth, seed, n, nodes)
        omc = plslam.ORBmatcher(nn, True, lib=emu_lib)
        c, got = omc.SearchByBoW(kf, fr)
        rc, ref = _oracle_bow(oracle, kf, fr, 50, nn, True)
        assert c == rc and (got == ref).all(), (seed, c, rc)
        assert nn < 1 or c > 20
    om2 = plslam.ORBmatcher(0.9, False, lib=emu_lib)
    kf, fr = _bow_sets(synth, 203, 200, 10)
    c, got = om2.SearchByBoW(kf, fr)
    rc, ref = _oracle_bow(oracle, kf, fr, 50, 0.9, False)
    assert c == rc and (got == ref).all()


# ------------------------------------------------------------------ GPU parity (config 4: 2 x 2000 descriptors)
@pytest.mark.gpu
def test_gpu_knn2_2000(plslam, oracle, synth):
    a, b, _ = synth.make_descriptor_sets(100, 2000)
    idx, dist = plslam.hamming_knn2(a, b)
    ri, rd = oracle.knn2(a, b)
    assert (idx == ri).all() and (dist == rd).all()
    # ties: many duplicate rows
    t = np.repeat(b[:50], 8, axis=0)
    idx, dist = plslam.hamming_knn2(a[:300], t)
    ri, rd = oracle.knn2(a[:300], t)
    assert (idx == ri).all() and (dist == rd).all()


@pytest.mark.gpu
def test_gpu_line_search_double(plslam, oracle, synth):
    m = plslam.LSDmatcher(0.7, True)
    sets1, sets2 = [], []
    for k, (n1, n2) in enumerate([(200, 201), (201, 200), (57, 120), (3, 2), (0, 10), (10, 0), (1, 1), (199, 64)] * 4):
        a, b = _cases(synth, 103 + k, n1, n2, flip=0.05 + 0.01 * (k % 5))
        sets1.append(a); sets2.append(b)
    got, cnt = m.SearchDoubleBatch(sets1, sets2)
    for p, (a, b) in enumerate(zip(sets1, sets2)):
        c, ref = _oracle_double(oracle, a, b, 50.0, 0.7)
        assert cnt[p] == c and (got[p, :len(a)] == ref).all(), p


@pytest.mark.gpu
def test_gpu_search_by_bow_2000(plslam, oracle, synth):
    om = plslam.ORBmatcher(0.7, True)
    kfs, frs = [], []
    for seed, n, nodes in [(300, 2000, 100), (301, 2000, 1000), (302, 1500, 10), (303, 64, 3), (304, 1, 1)]:
        kf, fr = _bow_sets(synth, seed, n, nodes)
        kfs.append(kf); frs.append(fr)
    got, cnt = om.SearchByBoWBatch(kfs, frs)
    for p, (kf, fr) in enumerate(zip(kfs, frs)):
        rc, ref = _oracle_bow(oracle, kf, fr, 50, 0.7, True)
        assert cnt[p] == rc and (got[p, :len(fr["desc"])] == ref).all(), p
    assert cnt[0] > 500
    # contended sets (a handful of prototype descriptors): first-come claims, exhausted candidate lists, ties; one pair alone (sixteen
    # wavefronts per pair) and as a batch (four)
    for nn in (0.7, 1.01):
        omc = plslam.ORBmatcher(nn, True)
        kfs, frs = [], []
        for seed, n, nodes in [(310, 2000, 100), (311, 2000, 40), (312, 1000, 12), (313, 300, 1), (314, 2000, 700)]:
            kf, fr = _bow_sets_contended(synth, seed, n, nodes)
            kfs.append(kf); frs.append(fr)
        for rep in (1, 40):   # 200 pairs: the 512-thread launch
            got, cnt = omc.SearchByBoWBatch(kfs * rep, frs * rep)
            for p, (kf, fr) in enumerate(zip(kfs * rep, frs * rep)):
                if p >= 5 and p < len(kfs) * rep - 5:
                    continue
                rc, ref = _oracle_bow(oracle, kf, fr, 50, nn, True)
                assert cnt[p] == rc and (got[p, :len(fr["desc"])] == ref).all(), (nn, rep, p)
        c1, g1 = omc.SearchByBoW(kfs[0], frs[0])
        rc, ref = _oracle_bow(oracle, kfs[0], frs[0], 50, nn, True)
        assert c1 == rc and (g1 == ref).all()


@pytest.mark.gpu
def test_gpu_search_by_bow_large_capacity(plslam, oracle, synth):
    """3000 features per set: the kernel's LDS request (87 KiB) goes beyond the 64 KiB default."""
    om = plslam.ORBmatcher(0.7, True)
    kf, fr = _bow_sets(synth, 305, 3000, 150)
    got, cnt = om.SearchByBoWBatch([kf], [fr])
    rc, ref = _oracle_bow(oracle, kf, fr, 50, 0.7, True)
    assert cnt[0] == rc and (got[0, :3000] == ref).all() and rc > 800


@pytest.mark.gpu
def test_gpu_match_golden(plslam, synth):
    """GPU vs the committed golden vectors (tests/golden/match_*.npz, tools/gen_golden.py)."""
    import glob
    import os
    for path in sorted(glob.glob(os.path.join(_util.ROOT, "tests", "golden", "match_*.npz"))):
        g = np.load(path)
        n = int(g["n"])
        a, b, perm = synth.make_descriptor_sets(int(g["seed"]), n, 0.08)
        idx, dist = plslam.hamming_knn2(a, b)
        assert (idx == g["knn_idx"]).all() and (dist == g["knn_dist"]).all()
        c, m = plslam.LSDmatcher(0.7, True).SearchDouble(a, b)
        assert c == int(g["double_n"]) and (m == g["double_m"]).all()
        kf = dict(desc=a, angle=g["ang_a"], node=g["node_a"], valid=np.ones(n, np.uint8))
        fr = dict(desc=b, angle=g["ang_b"], node=g["node_b"])
        cb, mb = plslam.ORBmatcher(0.7, True).SearchByBoW(kf, fr)
        assert cb == int(g["bow_n"]) and (mb == g["bow_m"]).all()


# ------------------------------------------------------------------ SearchByBoW(KeyFrame, KeyFrame)
def _kfkf_case(plslam, synth, seed, n, nodes):
    kf, fr = _bow_sets(synth, seed, n, nodes, valid_p=0.85)
    rng = synth.SplitMix64(seed + 31)
    k1 = np.zeros(n, plslam.KP_DTYPE); k1["angle"] = kf["angle"]
    k2 = np.zeros(n, plslam.KP_DTYPE); k2["angle"] = fr["angle"]
    a = dict(desc=kf["desc"], kps=k1, node=kf["node"], valid=kf["valid"], angle=kf["angle"])
    b = dict(desc=fr["desc"], kps=k2, node=fr["node"], valid=(rng.uniform(n) < 0.85).astype(np.uint8), angle=fr["angle"])
    return a, b


def _check_kfkf(plslam, oracle, synth, lib, cases):
    import ctypes as C
    L = oracle.lib()
    L.plo_orb_search_by_bow_kfkf.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_int,
                                                                                                C.c_void_p]
    L.plo_orb_search_by_bow_kfkf.restype = C.c_int
    om = plslam.ORBmatcher(0.8, True, lib=lib)
    sets = [_kfkf_case(plslam, synth, *c) for c in cases]
    got, cnt = om.SearchByBoWKeyFramesBatch([s[0] for s in sets], [s[1] for s in sets])
    tot = 0
    for p, (a, b) in enumerate(sets):
        n1, n2 = len(a["desc"]), len(b["desc"])
        ref = np.zeros(max(n1, 1), np.int32)
        p_ = oracle._p
        rc = L.plo_orb_search_by_bow_kfkf(p_(a["desc"]), p_(a["angle"]), p_(a["node"]), p_(a["valid"]), n1, p_(b["desc"]),
                                          p_(b["angle"]), p_(b["node"]), p_(b["valid"]), n2, 50, 0.8, 1, p_(ref))
        assert cnt[p] == rc and (got[p, :n1] == ref[:n1]).all(), p
        tot += rc
    return tot


def test_emu_search_by_bow_keyframes(plslam, oracle, synth, emu_lib):
    assert _check_kfkf(plslam, oracle, synth, emu_lib, [(400, 300, 20), (401, 120, 5), (402, 64, 64)]) > 100


@pytest.mark.gpu
def test_gpu_search_by_bow_keyframes(plslam, oracle, synth):
    assert _check_kfkf(plslam, oracle, synth, None, [(410, 2000, 100), (411, 2000, 1000), (412, 1500, 10), (413, 1, 1)]) > 1000


# ------------------------------------------------------------------ SearchForTriangulation
def _tri_case(plslam, synth, seed, n, nodes):
    """Two keyframes of a pure x-translation stereo pair: F12 = [t]_x up to scale, epipolar lines are horizontal."""
    kf, fr = _bow_sets(synth, seed, n, nodes)
    rng = synth.SplitMix64(seed + 17)
    perm = synth.make_descriptor_sets(seed, n, 0.06)[2]
    k1 = np.zeros(n, plslam.KP_DTYPE)
    k1["x"], k1["y"] = rng.uniform(n, 20, 620).astype(np.float32), rng.uniform(n, 20, 460).astype(np.float32)
    k1["octave"] = rng.randint(n, 0, 8).astype(np.int32)
    k1["angle"] = kf["angle"]
    k2 = k1[perm].copy()
    k2["x"] = (k2["x"] - rng.uniform(n, 2, 40)).astype(np.float32)          # disparity along the epipolar line
    k2["y"] = (k2["y"] + rng.uniform(n, -3.0, 3.0)).astype(np.float32)       # some violate the 3.84 sigma^2 gate
    k2["angle"] = fr["angle"]
    a = dict(desc=kf["desc"], kps=k1, node=kf["node"], has_mp=(rng.uniform(n) < 0.3).astype(np.uint8))
    b = dict(desc=fr["desc"], kps=k2, node=fr["node"], has_mp=(rng.uniform(n) < 0.3).astype(np.uint8))
    return a, b


def _check_tri(plslam, oracle, synth, lib, cases):
    import ctypes as C
    L = oracle.lib()
    V, I, F = C.c_void_p, C.c_int, C.c_float
    L.plo_orb_search_for_triangulation.argtypes = [V, V, V, V, I, V, V, V, V, I, V, F, F, V, V, I, I, V]
    L.plo_orb_search_for_triangulation.restype = I
    F12 = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32)     # l = (0, -1, y1): distance^2 = (y2 - y1)^2
    epi = (-5000.0, 240.0)
    sf = np.cumprod(np.r_[np.float32(1.0), np.full(7, np.float32(1.2))]).astype(np.float32)
    sig2 = (sf * sf).astype(np.float32)
    om = plslam.ORBmatcher(0.6, True, lib=lib)
    sets = [_tri_case(plslam, synth, *c) for c in cases]
    got, cnt = om.SearchForTriangulationBatch([s[0] for s in sets], [s[1] for s in sets], F12, epi, sf, sig2)
    tot = 0
    p_ = oracle._p
    for p, (a, b) in enumerate(sets):
        n1, n2 = len(a["desc"]), len(b["desc"])
        ref = np.zeros(max(n1, 1), np.int32)
        rc = L.plo_orb_search_for_triangulation(p_(a["kps"]), p_(a["desc"]), p_(a["node"]), p_(a["has_mp"]), n1, p_(b["kps"]),
                                                p_(b["desc"]), p_(b["node"]), p_(b["has_mp"]), n2, p_(F12), epi[0], epi[1], p_(sf),
                                                p_(sig2), 50, 1, p_(ref))
        assert cnt[p] == rc and (got[p, :n1] == ref[:n1]).all(), p
        tot += rc
    return tot


def test_emu_search_for_triangulation(plslam, oracle, synth, emu_lib):
    assert _check_tri(plslam, oracle, synth, emu_lib, [(500, 300, 20), (501, 120, 5), (502, 64, 64)]) > 50


@pytest.mark.gpu
def test_gpu_search_for_triangulation(plslam, oracle, synth):
    assert _check_tri(plslam, oracle, synth, None, [(510, 2000, 100), (511, 2000, 1000), (512, 1500, 10), (513, 1, 1)]) > 500
