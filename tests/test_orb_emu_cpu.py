"""CPU: the HIP kernel SOURCES executed under tests/hipemu (fiber emulation) vs the oracle.
This is a logic/debug aid that runs without a GPU -- parity evidence comes from tests marked gpu."""
import numpy as np
import pytest


def _cmp(P, O, img, nf, nl, lib, ini=20, mn=7):
    ref = O.OrbOracle(nf, 1.2, nl, ini, mn)
    rk, rd = ref.extract(img)
    ex = P.ORBextractor(nf, 1.2, nl, ini, mn, rows=img.shape[0], cols=img.shape[1], max_batch=1, lib=lib)
    try:
        kps, desc = ex(img)
        for l in range(nl):
            lv = ref.level(l)
            assert (ex.read_level(0, l, lv.shape) == lv).all(), "pyramid level %d" % l
            cr, cg = ref.candidates(l), ex.read_candidates(0, l)
            assert len(cr) == len(cg) and all((cr[f] == cg[f]).all() for f in ("x", "y", "response")), "FAST level %d" % l
        assert len(kps) == len(rk)
        for f in rk.dtype.names:
            assert (kps[f] == rk[f]).all(), f
        assert (desc == rd).all()
    finally:
        ex.close()
    return len(kps)


def test_emu_small_frame(plslam, oracle, synth, emu_lib):
    img = synth.make_frame(7, 120, 160, n_rect=40, n_line=20)
    assert _cmp(plslam, oracle, img, 200, 3, emu_lib) > 100


def test_emu_octree_binds(plslam, oracle, synth, emu_lib):
    # few features on a busy frame: the quad-tree's "largest first" phase and the N-break are exercised
    img = synth.make_frame(9, 200, 280, n_rect=150, n_line=60)
    n = _cmp(plslam, oracle, img, 120, 4, emu_lib)
    assert 100 <= n <= 140


def test_emu_threshold_fallback(plslam, oracle, synth, emu_lib):
    # high iniTh: most cells fall back to minTh
    img = synth.make_frame(10, 120, 160, n_rect=40, n_line=20)
    _cmp(plslam, oracle, img, 300, 2, emu_lib, ini=90, mn=10)


def test_emu_flat_image_no_keypoints(plslam, emu_lib):
    ex = plslam.ORBextractor(100, 1.2, 2, 20, 7, rows=100, cols=120, max_batch=1, lib=emu_lib)
    k, d = ex(np.full((100, 120), 33, np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    k, d = ex(np.zeros((0, 0), np.uint8))        # reference: empty image -> silent return
    assert len(k) == 0
    ex.close()


def test_emu_plan_errors(plslam, emu_lib):
    with pytest.raises(plslam.PlhError):
        plslam.ORBextractor(100, 1.2, 8, 20, 7, rows=60, cols=80, max_batch=1, lib=emu_lib)   # top level < 38 px
    ex = plslam.ORBextractor(100, 1.2, 2, 20, 7, rows=100, cols=120, max_batch=1, lib=emu_lib)
    with pytest.raises(plslam.PlhError):
        ex(np.zeros((90, 120), np.uint8))        # size does not match the plan
    ex.close()


def test_emu_wide_odd_pitch_frame(plslam, oracle, synth, emu_lib):
    # width 403: level-0 rows are not dword aligned (funnel-shifted staging) and a cell row is split into two FAST strips
    img = synth.make_frame(12, 110, 403, n_rect=60, n_line=30)
    assert _cmp(plslam, oracle, img, 300, 2, emu_lib) > 150


def test_emu_dense_corners_chunked_nms(plslam, oracle, synth, emu_lib, monkeypatch):
    # white noise at low thresholds: a third of the pixels are corners, many of them possible on both sides; the corner list
    # is processed in chunks of 5 rows (the path real frames take only above ~40 % corner density)
    rng = synth.SplitMix64(77)
    img = rng.randint(130 * 170, 0, 256).astype(np.uint8).reshape(130, 170)
    monkeypatch.setenv("PLH_EMU_FAST_ROWS", "5")
    assert _cmp(plslam, oracle, img, 400, 2, emu_lib, ini=9, mn=3) > 300
    monkeypatch.delenv("PLH_EMU_FAST_ROWS")
    assert _cmp(plslam, oracle, img, 400, 2, emu_lib, ini=9, mn=3) > 300
