"""GPU parity tests (run with -m gpu on an MI355X): HIP path through the C ABI vs the CPU oracle
and vs the committed golden vectors.  Bar: BIT-EXACT keypoints (x, y, size, angle, response, octave)
and 256-bit descriptors."""
import glob
import os

import numpy as np
import pytest

import _util

pytestmark = pytest.mark.gpu


def _assert_same(kps, desc, rk, rd, what=""):
    assert len(kps) == len(rk), "%s count %d vs %d" % (what, len(kps), len(rk))
    for f in rk.dtype.names:
        bad = np.nonzero(kps[f] != rk[f])[0]
        assert len(bad) == 0, "%s field %s differs at %s" % (what, f, bad[:5])
    assert (desc == rd).all(), "%s descriptors differ in %d rows" % (what, (desc != rd).any(axis=1).sum())


def test_native_library_loaded(plslam):
    lib = plslam.load()
    assert lib.plh_device_count() >= 1
    assert b"gfx950" in lib.plh_version()


def test_stage_taps_s1(plslam, oracle, synth):
    img = synth.make_frame(1)
    ref = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    rk, rd = ref.extract(img)
    ex = plslam.ORBextractor(1000, 1.2, 8, 20, 7, rows=480, cols=640, max_batch=1)
    kps, desc = ex(img)
    for l in range(8):
        lv = ref.level(l)
        assert (ex.read_level(0, l, lv.shape) == lv).all(), "pyramid level %d" % l
        cr, cg = ref.candidates(l), ex.read_candidates(0, l)
        assert len(cr) == len(cg) and all((cr[f] == cg[f]).all() for f in ("x", "y", "response")), "FAST level %d" % l
    _assert_same(kps, desc, rk, rd, "S1")
    ex.close()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(_util.ROOT, "tests", "golden", "orb_*.npz"))))
def test_golden(plslam, synth, path):
    g = np.load(path)
    img = synth.make_frame(int(g["seed"]), int(g["rows"]), int(g["cols"]), n_rect=int(g["n_rect"]), n_line=int(g["n_line"]))
    ex = plslam.ORBextractor(int(g["nfeatures"]), 1.2, int(g["nlevels"]), int(g["ini"]), int(g["mn"]),
                             rows=int(g["rows"]), cols=int(g["cols"]), max_batch=1)
    kps, desc = ex(img)
    _assert_same(kps, desc, g["kps"], g["desc"], os.path.basename(path))
    ex.close()


def test_batch_s2_vs_oracle(plslam, oracle, synth):
    """Config 3 inputs (seeds 2..): a 24-frame batch through the batch entry point, every frame vs the oracle."""
    B = 24
    frames = synth.make_frames(2, B, 480, 640)
    ex = plslam.ORBextractor(1000, 1.2, 8, 20, 7, rows=480, cols=640, max_batch=B)
    kps, desc, n = ex.extract_batch(frames)
    ref = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    for b in range(B):
        rk, rd = ref.extract(frames[b])
        _assert_same(kps[b, :n[b]], desc[b, :n[b]], rk, rd, "frame %d" % b)
    # batch result is independent of batch composition: frame 5 alone == frame 5 in the batch
    k1, d1 = ex(frames[5])
    _assert_same(k1, d1, kps[5, :n[5]], desc[5, :n[5]], "single vs batch")
    ex.close()


def test_kitti_2000_features(plslam, oracle, synth):
    frames = synth.make_frames(1001, 3, 376, 1241)
    ex = plslam.ORBextractor(2000, 1.2, 8, 20, 7, rows=376, cols=1241, max_batch=3)
    kps, desc, n = ex.extract_batch(frames)
    ref = oracle.OrbOracle(2000, 1.2, 8, 20, 7)
    for b in range(3):
        rk, rd = ref.extract(frames[b])
        _assert_same(kps[b, :n[b]], desc[b, :n[b]], rk, rd, "kitti %d" % b)
    ex.close()


@pytest.mark.parametrize("nf,nl,ini,mn", [(300, 8, 20, 7), (1000, 8, 40, 30), (1500, 5, 12, 5), (50, 8, 20, 7)])
def test_parameter_sweep(plslam, oracle, synth, nf, nl, ini, mn):
    img = synth.make_frame(77 + nf)
    ex = plslam.ORBextractor(nf, 1.2, nl, ini, mn, rows=480, cols=640, max_batch=1)
    kps, desc = ex(img)
    rk, rd = oracle.OrbOracle(nf, 1.2, nl, ini, mn).extract(img)
    _assert_same(kps, desc, rk, rd, "sweep")
    ex.close()


def test_edge_cases(plslam, oracle):
    ex = plslam.ORBextractor(500, 1.2, 4, 20, 7, rows=120, cols=160, max_batch=2)
    k, d = ex(np.full((120, 160), 90, np.uint8))            # flat: no corners -> descriptors.release()
    assert len(k) == 0 and d.shape == (0, 32)
    k, d = ex(np.zeros((0, 0), np.uint8))                   # empty image -> silent return
    assert len(k) == 0
    with pytest.raises(plslam.PlhError):
        ex(np.zeros((100, 160), np.uint8))                  # not the planned size
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, (120, 160)).astype(np.uint8)   # maximum corner density
    k, d = ex(noise)
    rk, rd = oracle.OrbOracle(500, 1.2, 4, 20, 7).extract(noise)
    _assert_same(k, d, rk, rd, "noise")
    ex.close()


def test_full_size_properties(plslam, synth):
    """Size-independent properties on a large batch (no oracle): determinism, ordering, bounds."""
    B = 64
    frames = synth.make_frames(300, B, 480, 640, unique=8)
    ex = plslam.ORBextractor(1000, 1.2, 8, 20, 7, rows=480, cols=640, max_batch=B)
    k1, d1, n1 = ex.extract_batch(frames)
    k2, d2, n2 = ex.extract_batch(frames)
    assert (n1 == n2).all()
    for b in range(B):   # idempotent (rows past n[b] are unspecified)
        assert k1[b, :n1[b]].tobytes() == k2[b, :n1[b]].tobytes() and d1[b, :n1[b]].tobytes() == d2[b, :n1[b]].tobytes()
    per = ex.features_per_level()
    sf = ex.GetScaleFactors()
    for b in range(B):
        k = k1[b, :n1[b]]
        assert (np.diff(k["octave"]) >= 0).all()
        assert (np.bincount(k["octave"], minlength=8) <= per + 3).all()
        assert ((k["angle"] >= 0) & (k["angle"] < 360)).all()
        lx, ly = k["x"] / sf[k["octave"]], k["y"] / sf[k["octave"]]
        assert (lx > 18.9).all() and (ly > 18.9).all()
    # permuting the batch permutes the results
    perm = np.random.default_rng(0).permutation(B)
    k3, d3, n3 = ex.extract_batch(frames[perm])
    assert (n3 == n1[perm]).all()
    for b in range(B):
        assert d3[b, :n3[b]].tobytes() == d1[perm[b], :n3[b]].tobytes()
        assert k3[b, :n3[b]].tobytes() == k1[perm[b], :n3[b]].tobytes()
    ex.close()
