"""CPU tests: the oracle against hand-derivable known answers and the committed golden vectors.
(The reference ships no tests or fixtures for this path -- SURVEY.md 4 -- so these are ours.)"""
import glob
import os

import numpy as np
import pytest

import _util


def test_tables_match_reference_constants(oracle):
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    # umax of the 31-px circular patch (ORBextractor.cc:454-469), derived by hand in SURVEY.md 8a
    assert o.umax().tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    # mnFeaturesPerLevel (ORBextractor.cc:435-446)
    assert o.features_per_level().tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert oracle.OrbOracle(2000, 1.2, 8, 20, 7).features_per_level().tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    sf = o.scale_table(0)
    assert sf[0] == 1.0 and sf[1] == np.float32(1.2)
    assert [int(31 * s) for s in sf] == [31, 37, 44, 53, 64, 77, 92, 111]


def test_pyramid_sizes(oracle, synth):
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    o.extract(synth.make_frame(3, 480, 640, n_rect=30, n_line=10))
    assert [o.level(l).shape for l in range(8)] == [(480, 640), (400, 533), (333, 444), (278, 370), (231, 309),
                                                     (193, 257), (161, 214), (134, 179)]


def test_gaussian_kernels_q8(oracle):
    assert oracle.gaussian_kernel_q8(7, 2.0).tolist() == [18, 34, 49, 55, 49, 34, 18]
    assert oracle.gaussian_kernel_q8(7, 0.75).tolist() == [0, 4, 56, 136, 56, 4, 0]
    assert oracle.gaussian_kernel_q8(5, 1.0).tolist() == [14, 63, 103, 63, 14]


def test_blur_constant_and_reflect(oracle):
    img = np.full((20, 24), 100, np.uint8)
    out = oracle.gaussian_blur(img, 7, 2.0)
    # kernel sums to 257 per pass (not renormalised): 100*257*257 / 65536 = 100.78 -> 101
    assert (out == 101).all()


def test_fast_atan2_cardinals(oracle):
    f = oracle.lib().plo_fast_atan2
    assert f(0.0, 1.0) == 0.0 and f(0.0, 0.0) == 0.0
    assert abs(f(1.0, 0.0) - 90.0) < 1e-3 and abs(f(0.0, -1.0) - 180.0) < 1e-3 and abs(f(-1.0, 0.0) - 270.0) < 1e-3
    assert abs(f(1.0, 1.0) - 45.0) < 0.3
    rng = np.random.default_rng(0)
    for y, x in rng.normal(size=(200, 2)):
        a = f(float(y), float(x))
        assert abs(((a - np.degrees(np.arctan2(y, x))) + 180) % 360 - 180) < 0.3


def test_fast_atan2_error_bound_behind_the_direction_test(oracle):
    """k_lsd_grow decides alignments by directions when a pixel is further than LSD_ALIGN_MARGIN_DEG = 0.05 degrees from the
    tolerance (lsd_grow.hip, lsd_classify); that is only exact if fastAtan2 stays well inside the margin of the true angle."""
    f = oracle.lib().plo_fast_atan2
    rng = np.random.default_rng(7)
    worst = 0.0
    ang = np.concatenate([np.linspace(0, 360, 20001), rng.uniform(0, 360, 20000)])
    for a in ang:
        r = float(rng.uniform(0.5, 3.0e5))
        y, x = np.float32(r * np.sin(np.radians(a))), np.float32(r * np.cos(np.radians(a)))
        d = f(float(y), float(x)) - np.degrees(np.arctan2(float(y), float(x))) % 360.0
        worst = max(worst, abs((d + 180.0) % 360.0 - 180.0))
    assert worst < 0.012, worst


def test_cv_round_half_even(oracle):
    r = oracle.lib().plo_cv_round_f
    assert [r(0.5), r(1.5), r(2.5), r(-0.5), r(-1.5), r(2.4999), r(2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_fast_synthetic_corner(oracle):
    img = np.full((15, 15), 50, np.uint8)
    img[7:, 7:] = 200          # a bright quadrant: pixels at its apex see a 9+ arc of much darker ring pixels
    kps = oracle.fast9_16(img, 20, False)
    assert (7, 7) in {(int(k["x"]), int(k["y"])) for k in kps}
    # score = largest threshold that keeps it a corner = |200-50| - 1
    assert oracle.lib().plo_fast_score(img.ctypes.data + 7 * 15 + 7, 15, 20) == 149
    # all apex pixels tie at 149 and the 3x3 NMS is STRICT (>): a plateau suppresses itself entirely
    assert len(oracle.fast9_16(img, 20, True)) == 0
    img[7, 7] = 210            # break the tie
    k2 = oracle.fast9_16(img, 20, True)
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in k2] == [(7, 7, 159)]
    assert len(oracle.fast9_16(np.full((15, 15), 77, np.uint8), 7, True)) == 0


def test_fast_score_threshold_independent(oracle):
    """cornerScore is independent of the threshold for corners -> one score map serves iniTh and minTh."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 40)).astype(np.uint8)
    k20 = oracle.fast9_16(img, 20, False)
    k7 = oracle.fast9_16(img, 7, False)
    s7 = {(int(k["x"]), int(k["y"])) for k in k7}
    assert all((int(k["x"]), int(k["y"])) in s7 for k in k20)
    L = oracle.lib()
    for k in k20[:50]:
        p = img.ctypes.data + int(k["y"]) * 40 + int(k["x"])
        assert L.plo_fast_score(p, 40, 20) == L.plo_fast_score(p, 40, 7)


def test_resize_identity_and_range(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (60, 80)).astype(np.uint8)
    assert (oracle.resize_linear(img, 80, 60) == img).all()
    small = oracle.resize_linear(np.full((60, 80), 200, np.uint8), 67, 50)
    assert (small == 200).all()


def test_orb_empty_and_flat(oracle):
    o = oracle.OrbOracle(500, 1.2, 4, 20, 7)
    k, d = o.extract(np.full((120, 160), 90, np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(_util.ROOT, "tests", "golden", "orb_*.npz"))))
def test_oracle_matches_golden(oracle, synth, path):
    g = np.load(path)
    img = synth.make_frame(int(g["seed"]), int(g["rows"]), int(g["cols"]), n_rect=int(g["n_rect"]), n_line=int(g["n_line"]))
    assert int(img.astype(np.int64).sum()) == int(g["img_sum"]), "synthetic generator drifted"
    o = oracle.OrbOracle(int(g["nfeatures"]), 1.2, int(g["nlevels"]), int(g["ini"]), int(g["mn"]))
    kps, desc = o.extract(img)
    assert len(kps) == len(g["kps"])
    for f in kps.dtype.names:
        assert (kps[f] == g["kps"][f]).all(), f
    assert (desc == g["desc"]).all()


def test_orb_output_invariants(oracle, synth):
    img = synth.make_frame(11, 240, 320, n_rect=120, n_line=60)
    o = oracle.OrbOracle(500, 1.2, 6, 20, 7)
    kps, desc = o.extract(img)
    assert (np.diff(kps["octave"]) >= 0).all()          # level-major concatenation
    per = np.bincount(kps["octave"], minlength=6)
    assert (per <= o.features_per_level() + 3).all()
    assert ((kps["angle"] >= 0) & (kps["angle"] < 360)).all()
    sf = o.scale_table(0)
    lx = kps["x"] / sf[kps["octave"]]
    assert (lx >= 19 - 1e-3).all()


# ------------------------------------------------------------------ golden vectors: lines and matchers
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(_util.ROOT, "tests", "golden", "line_*.npz"))))
def test_oracle_matches_golden_lines(oracle, synth, path):
    g = np.load(path)
    img = synth.make_frame(int(g["seed"]), int(g["rows"]), int(g["cols"]), n_rect=int(g["n_rect"]), n_line=int(g["n_line"]))
    assert int(img.astype(np.int64).sum()) == int(g["img_sum"])          # the generator itself is pinned
    refine = int(g["refine"]) if "refine" in g.files else 0      # (files of rounds 1-4: LSD_REFINE_STD; line_adv_*: LSD_REFINE_ADV)
    segs = oracle.lsd_detect(img, refine=refine)
    kl, desc, fn = oracle.line_extract(img, int(g["nfeature"]), float(g["minlen"]), refine=refine)
    assert segs.shape == g["segs"].shape and (segs == g["segs"]).all()
    assert len(kl) == len(g["keylines"]) and all((kl[f] == g["keylines"][f]).all() for f in kl.dtype.names)
    assert (desc == g["desc"]).all() and (fn == g["linefn"]).all()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(_util.ROOT, "tests", "golden", "match_*.npz"))))
def test_oracle_matches_golden_matchers(oracle, synth, path):
    import ctypes as C
    g = np.load(path)
    n = int(g["n"])
    a, b, perm = synth.make_descriptor_sets(int(g["seed"]), n, 0.08)
    idx, dist = oracle.knn2(a, b)
    assert (idx == g["knn_idx"]).all() and (dist == g["knn_dist"]).all()
    m = np.zeros(n, np.int32)
    L = oracle.lib()
    c = L.plo_line_search_double(oracle._p(a), n, oracle._p(b), n, C.c_float(50.0), C.c_float(0.7), oracle._p(m))
    assert c == int(g["double_n"]) and (m == g["double_m"]).all()
    mb = np.zeros(n, np.int32)
    valid = np.ones(n, np.uint8)
    cb = L.plo_orb_search_by_bow(oracle._p(a), oracle._p(g["ang_a"]), oracle._p(g["node_a"]), oracle._p(valid), n, oracle._p(b),
                                 oracle._p(g["ang_b"]), oracle._p(g["node_b"]), n, 50, C.c_float(0.7), 1, oracle._p(mb))
    assert cb == int(g["bow_n"]) and (mb == g["bow_m"]).all()
