"""Line half of the front end (LSD + KeyLine selection + LBD): oracle sanity (CPU), HIP sources under hipemu (CPU),
GPU parity.  The bar is BIT-EXACT: segments, KeyLines (every field), LBD bytes and line equations equal the oracle's.
The north star would allow a float tolerance for this half ("LSD endpoints/LBD within a stated float tolerance"); it is
only available behind an explicit escape, PLSLAM_LINE_TOLERANCE=1, which downgrades a mismatch to
  * |endpoint difference| <= 1e-3 px on >= 99% of the lines, <= 2 differing LBD bits on >= 99% of those, equal
    numOfPixels, line equations within 1e-3
and says so on stdout.  Without the flag any difference fails the test."""
import os
import numpy as np
import pytest

import _util

TUM1_K = [517.306408, 516.469215, 318.643040, 255.313989]           # Examples/Monocular/TUM1.yaml:8-11
TUM1_D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]       # TUM1.yaml:13-17


def _oracle_line(O, img, nf, minlen, K=None, D=None, mask=None, refine=None):
    """(keylines, LBD, line equations, LSD segments) of the oracle; refine=None: the reference's level = the library's default
    (LSD_REFINE_ADV, oracle/plo.py REFERENCE_REFINE)."""
    src = img
    if K is not None:
        rows, cols = img.shape
        mx = np.zeros((rows, cols), np.float32)
        my = np.zeros((rows, cols), np.float32)
        Kf = np.asarray(K, np.float32)
        Df = np.asarray(D, np.float32)
        O.lib().plo_undistort_maps(O._p(Kf), O._p(Df), cols, rows, O._p(mx), O._p(my))
        src = np.zeros_like(img)
        O.lib().plo_remap_linear_u8(O._p(img), cols, rows, cols, O._p(mx), O._p(my), O._p(src), cols)
    return O.line_extract(src, nf, minlen, mask, refine=refine) + (O.lsd_detect(src, refine=refine),)


def _exact(kl, desc, fn, rk, rd, rf):
    return (len(kl) == len(rk) and all((kl[f] == rk[f]).all() for f in rk.dtype.names) and (desc == rd).all()
            and (fn == rf).all())


def _match(kl, desc, fn, rk, rd, rf, what=""):
    """Exactness is the assertion; the tolerance is an explicit, reported escape."""
    if _exact(kl, desc, fn, rk, rd, rf):
        return True
    if os.environ.get("PLSLAM_LINE_TOLERANCE") == "1":
        print("PLSLAM_LINE_TOLERANCE=1: %s is NOT bit-exact, checking the float tolerance instead" % what)
        _close(kl, desc, fn, rk, rd, rf, what)
        return False
    assert len(kl) == len(rk), "%s: %d vs %d keylines" % (what, len(kl), len(rk))
    bad = [f for f in rk.dtype.names if not (kl[f] == rk[f]).all()]
    raise AssertionError("%s: not bit-exact (keyline fields %s, %d descriptor bytes, %d line-equation terms differ)" %
                         (what, bad, int((desc != rd).sum()), int((fn != rf).sum())))


def _close(kl, desc, fn, rk, rd, rf, what=""):
    assert len(kl) == len(rk), "%s: %d vs %d keylines" % (what, len(kl), len(rk))
    ep = np.stack([kl[f] for f in ("startPointX", "startPointY", "endPointX", "endPointY")], 1)
    er = np.stack([rk[f] for f in ("startPointX", "startPointY", "endPointX", "endPointY")], 1)
    good = np.abs(ep - er).max(axis=1) <= 1e-3
    assert good.mean() >= 0.99, "%s: only %.3f of the lines within 1e-3 px" % (what, good.mean())
    bits = np.unpackbits(desc ^ rd, axis=1).sum(axis=1)
    assert (bits[good] <= 2).mean() >= 0.99, "%s: LBD differs (max %d bits)" % (what, bits[good].max())
    assert (kl["numOfPixels"][good] == rk["numOfPixels"][good]).all()
    assert np.abs(fn[good] - rf[good]).max() <= 1e-3


# ------------------------------------------------------------------ oracle sanity (CPU)
def test_oracle_lsd_finds_a_drawn_edge(oracle):
    img = np.full((120, 160), 60, np.uint8)
    img[:, 80:] = 190                                      # one long vertical step edge at x = 80
    segs = oracle.lsd_detect(img)
    assert len(segs) >= 1
    ln = np.hypot(segs[:, 0] - segs[:, 2], segs[:, 1] - segs[:, 3])
    s = segs[np.argmax(ln)]
    assert ln.max() > 90 and abs(s[0] - 80) < 1.5 and abs(s[2] - 80) < 1.5
    assert len(oracle.lsd_detect(np.full((120, 160), 128, np.uint8))) == 0


def test_oracle_line_extract_invariants(oracle, synth):
    img = synth.make_frame(21, 240, 320, n_rect=120, n_line=60)
    kl, desc, fn = oracle.line_extract(img, 60, 0.0)
    assert len(kl) == 61                                   # LineExtractor.cpp:64 keeps index+1 = nLSDFeature+1 lines
    assert (np.diff(kl["response"]) <= 0).all() and (kl["class_id"] == np.arange(61)).all()
    assert np.allclose(np.hypot(fn[:, 0], fn[:, 1]), 1.0)
    for a, b in (("startPointX", "startPointY"), ("endPointX", "endPointY")):   # endpoints lie on l.(x,y,1) = 0
        assert np.abs(fn[:, 0] * kl[a] + fn[:, 1] * kl[b] + fn[:, 2]).max() < 1e-3
    assert (kl["numOfPixels"] >= 1).all() and (kl["octave"] == 0).all()
    k2, _, _ = oracle.line_extract(img, 60, 30.0)          # min_line_length cuts the tail
    assert 0 < len(k2) <= 61 and (k2["lineLength"][:-1] >= 30.0).all()
    mask = np.zeros_like(img)                              # a line is dropped only if BOTH endpoints are on mask == 0
    mask[:, :160] = 255
    km, _, _ = oracle.line_extract(img, 1000, 0.0, mask)
    ka, _, _ = oracle.line_extract(img, 1000, 0.0)
    both_out = (ka["startPointX"].astype(int) >= 160) & (ka["endPointX"].astype(int) >= 160)
    assert len(km) == (~both_out).sum()


def test_oracle_lsd_refine_adv_known_answers(oracle, synth):
    """LSD_REFINE_ADV (oracle/lsd.cc): nfa() against values computed by hand here, and what the level does to a frame."""
    import math
    L = oracle.lib()
    w, h = 512, 384
    logNT = 5 * (math.log10(w) + math.log10(h)) / 2 + math.log10(11.0)
    p = 0.125
    # trivial cases of the published code: no points / no aligned points -> -log NT; all points aligned -> -log NT - n log10 p
    assert L.plo_lsd_nfa(w, h, 0, 0, p) == -logNT and L.plo_lsd_nfa(w, h, 17, 0, p) == -logNT
    assert L.plo_lsd_nfa(w, h, 40, 40, p) == -logNT - 40.0 * math.log10(p)
    # log_gamma: Lanczos below 15, Windschitl above; both within 1e-9 of ln Gamma
    for x in (1.0, 2.0, 5.0, 11.5, 15.0, 15.5, 40.0, 300.0):
        assert abs(L.plo_lsd_log_gamma(x) - math.lgamma(x)) < 1e-9 * max(1.0, abs(math.lgamma(x))), x
    assert abs(L.plo_lsd_log_gamma(5.0) - math.log(24.0)) < 1e-12           # Gamma(5) = 4!
    # general case, by hand: the published first term is (n + 1) -- NOT log Gamma(n + 1) --, then the tail is summed until the
    # remaining terms are below 10 % of it
    def nfa_published(n, k):
        log1 = (n + 1.0) - math.lgamma(k + 1.0) - math.lgamma(n - k + 1.0) + k * math.log(p) + (n - k) * math.log(1.0 - p)
        term = math.exp(log1)
        if term == 0.0:                 # double_equal(term, 0): the first term underflows
            return -log1 / math.log(10.0) - logNT if k > n * p else -logNT
        tail = term
        for i in range(k + 1, n + 1):
            bin_term = (n - i + 1.0) / i
            mult = bin_term * p / (1 - p)
            term *= mult
            tail += term
            if bin_term < 1:
                err = term * ((1 - mult ** (n - i + 1.0)) / (1 - mult) - 1)
                if err < 0.1 * abs(-math.log10(tail) - logNT) * tail:
                    break
        return -math.log10(tail) - logNT
    for n, k in ((20, 10), (60, 25), (200, 60), (200, 40), (1000, 180)):
        assert abs(L.plo_lsd_nfa(w, h, n, k, p) - nfa_published(n, k)) < 1e-9 * max(1.0, abs(nfa_published(n, k))), (n, k)
    # ... which is NOT the binomial tail of the original algorithm (documented quirk): 20 points, 10 aligned
    true_tail = sum(math.comb(20, i) * p ** i * (1 - p) ** (20 - i) for i in range(10, 21))
    assert abs(L.plo_lsd_nfa(w, h, 20, 10, p) - (-math.log10(true_tail) - logNT)) > 1.0
    # on a frame: ADV only ever removes or adjusts STD's rectangles
    img = synth.make_frame(3, 240, 320)
    std, adv = oracle.lsd_detect(img, refine=0), oracle.lsd_detect(img, refine=1)
    assert 0 < len(adv) <= len(std)
    sstd = {tuple(x) for x in std}
    assert sum(tuple(x) in sstd for x in adv) >= 0.9 * len(adv)
    assert len(oracle.lsd_detect(np.full((120, 160), 128, np.uint8), refine=1)) == 0


def test_oracle_lbd_determinism_and_norm(oracle, synth):
    img = synth.make_frame(22, 240, 320, n_rect=120, n_line=60)
    kl, desc, _ = oracle.line_extract(img, 30, 0.0)
    d2, f72 = oracle.lbd_compute(img, kl)
    assert (d2 == desc).all()
    assert np.allclose(np.linalg.norm(f72, axis=1), 1.0, atol=1e-4)   # final re-normalisation


# ------------------------------------------------------------------ HIP sources under hipemu (CPU)
@pytest.mark.parametrize("seed,rows,cols,nf,minlen,undist", [(7, 120, 160, 50, 0.0, False), (8, 120, 160, 20, 15.0, False),
                                                            (10, 120, 160, 50, 0.0, True), (11, 118, 203, 40, 0.0, False),
                                                            (12, 120, 160, 40, 0.0, "outside")])
def test_emu_line_extract(plslam, oracle, synth, emu_lib, seed, rows, cols, nf, minlen, undist):
    img = synth.make_frame(seed, rows, cols, n_rect=40, n_line=20)
    K, D = ([150.0, 150.0, 80.0, 60.0], TUM1_D) if undist else (None, None)
    if undist == "outside":   # principal point off the image: the map leaves the frame on two sides (taps outside read 0, per axis)
        K, D = [150.0, 150.0, 20.0, 95.0], [0.4, -0.9, 0.01, -0.008, 1.1]
    rk, rd, rf, rs = _oracle_line(oracle, img, nf, minlen, K, D)
    ex = plslam.LINEextractor(1, 1.2, nf, minlen, rows=rows, cols=cols, max_batch=1, lib=emu_lib, K=K, D=D)
    kl, desc, fn = ex(img)
    gs = ex.read_segments(0)
    ex.close()
    assert len(gs) == len(rs) and (gs == rs).all()
    assert _exact(kl, desc, fn, rk, rd, rf)


def test_emu_line_extract_refine_std(plslam, oracle, synth, emu_lib):
    """plh_line_set_refine(PLH_LSD_REFINE_STD): the level of the un-linked twin in the reference's tree (the default is ADV)."""
    img = synth.make_frame(7, 120, 160, n_rect=40, n_line=20)
    rk, rd, rf, rs = _oracle_line(oracle, img, 50, 0.0, refine=0)
    ex = plslam.LINEextractor(1, 1.2, 50, 0.0, rows=120, cols=160, max_batch=1, lib=emu_lib)
    assert plslam.load(emu_lib).plh_lsd_refine_default() == 1
    ex.set_refine(0)
    kl, desc, fn = ex(img)
    gs = ex.read_segments(0)
    ex.close()
    assert len(gs) == len(rs) and (gs == rs).all() and len(rs) > len(oracle.lsd_detect(img, refine=1))
    assert _exact(kl, desc, fn, rk, rd, rf)


@pytest.mark.parametrize("refine", [1, 0], ids=["adv", "std"])
def test_emu_line_few_blocks_per_frame(plslam, oracle, synth, emu_lib, refine, monkeypatch):
    """The launch shape of large batches -- a few one-wavefront blocks per frame in the seed ordering, k_lsd_rects and the LSD_REFINE_ADV
    kernels, each looping over several chunks (lsd_blocks_per_frame) -- on a small frame: the same records."""
    monkeypatch.setenv("PLH_BLOCKS_PER_FRAME_MIN", "1")
    img = synth.make_frame(9, 118, 203, n_rect=40, n_line=20)
    rk, rd, rf, rs = _oracle_line(oracle, img, 40, 0.0, refine=refine)
    ex = plslam.LINEextractor(1, 1.2, 40, 0.0, rows=118, cols=203, max_batch=1, lib=emu_lib)
    ex.set_refine(refine)
    kl, desc, fn = ex(img)
    gs = ex.read_segments(0)
    ex.close()
    assert len(gs) == len(rs) and (gs == rs).all() and len(rs) > 40
    assert _exact(kl, desc, fn, rk, rd, rf)


def _sawtooth(rows, cols, period=40, slope=6):
    """Ramps of constant gradient: every tooth is ONE region of period x rows aligned pixels -- thousands of queue entries, far
    beyond the 512-entry LDS mirror of the region queue (the k_lsd_grow paths that read the queue back from global memory),
    with a wide rectangle behind it."""
    x = np.arange(cols)
    row = ((x % period) * slope + 8).astype(np.uint8)
    img = np.repeat(row[None, :], rows, axis=0).copy()
    img[::7, ::5] += 1                                     # a little texture so that not every pixel ties
    return img


def test_emu_line_large_regions(plslam, oracle, emu_lib):
    img = _sawtooth(120, 200)
    rk, rd, rf, rs = _oracle_line(oracle, img, 50, 0.0)
    ex = plslam.LINEextractor(1, 1.2, 50, 0.0, rows=120, cols=200, max_batch=1, lib=emu_lib)
    kl, desc, fn = ex(img)
    gs = ex.read_segments(0)
    ex.close()
    assert len(rs) >= 3                                    # the teeth are found at all
    assert len(gs) == len(rs) and (gs == rs).all()
    assert _exact(kl, desc, fn, rk, rd, rf)


@pytest.mark.gpu
def test_gpu_line_large_regions(plslam, oracle):
    img = _sawtooth(480, 640, period=42, slope=6)
    rk, rd, rf, rs = _oracle_line(oracle, img, 200, 0.0)
    ex = plslam.LINEextractor(1, 1.2, 200, 0.0, rows=480, cols=640, max_batch=1)
    kl, desc, fn = ex(img)
    gs = ex.read_segments(0)
    ex.close()
    assert len(gs) == len(rs) and (gs == rs).all()
    _match(kl, desc, fn, rk, rd, rf, "sawtooth")


@pytest.mark.parametrize("waves", [4, 8])
def test_emu_line_multi_wavefront_growing(plslam, oracle, synth, emu_lib, waves):
    """k_lsd_grow_mw: several wavefronts grow regions of one frame as optimistic transactions with in-order commit; the
    segments are the one-wavefront kernel's (and the oracle's), whatever the schedule."""
    img = synth.make_frame(7, 120, 160, n_rect=40, n_line=20)
    rk, rd, rf, rs = _oracle_line(oracle, img, 50, 0.0)
    ex = plslam.LINEextractor(1, 1.2, 50, 0.0, rows=120, cols=160, max_batch=1, lib=emu_lib)
    ex.set_grow_waves(waves)
    kl, desc, fn = ex(img)
    gs = ex.read_segments(0)
    assert ex.status() == 0
    assert len(gs) == len(rs) and (gs == rs).all()
    assert _exact(kl, desc, fn, rk, rd, rf)
    saw = _sawtooth(120, 200)                              # regions of thousands of pixels: logs far beyond the LDS ring
    rk, rd, rf, rs = _oracle_line(oracle, saw, 50, 0.0)
    ex2 = plslam.LINEextractor(1, 1.2, 50, 0.0, rows=120, cols=200, max_batch=1, lib=emu_lib)
    ex2.set_grow_waves(waves)
    kl, desc, fn = ex2(saw)
    assert (ex2.read_segments(0) == rs).all() and _exact(kl, desc, fn, rk, rd, rf)
    with pytest.raises(plslam.PlhError):
        ex.set_grow_waves(17)
    ex.close(); ex2.close()


def test_emu_line_refine_adv(plslam, oracle, synth, emu_lib):
    """plh_line_set_refine(PLH_LSD_REFINE_ADV): rect_improve / rect_nfa / nfa behind refine(), one wavefront per frame and four."""
    img = synth.make_frame(9, 120, 160, n_rect=40, n_line=20)
    rs = oracle.lsd_detect(img, refine=1)
    rk, rd, rf = oracle.line_extract(img, 50, 0.0, refine=1)
    assert 0 < len(rs) < len(oracle.lsd_detect(img, refine=0))      # the level does something on this frame
    for waves in (0, 4):
        ex = plslam.LINEextractor(1, 1.2, 50, 0.0, rows=120, cols=160, max_batch=1, lib=emu_lib)
        ex.set_refine(1)
        ex.set_grow_waves(waves)
        kl, desc, fn = ex(img)
        gs = ex.read_segments(0)
        assert ex.status() == 0
        assert len(gs) == len(rs) and (gs == rs).all(), waves
        assert _exact(kl, desc, fn, rk, rd, rf), waves
        with pytest.raises(plslam.PlhError):
            ex.set_refine(2)
        ex.close()


def test_emu_line_edge_cases(plslam, emu_lib):
    ex = plslam.LINEextractor(1, 1.2, 20, 0.0, rows=64, cols=96, max_batch=1, lib=emu_lib)
    kl, desc, fn = ex(np.full((64, 96), 99, np.uint8))     # flat image: no segments -> empty outputs
    assert len(kl) == 0 and desc.shape == (0, 32) and fn.shape == (0, 3)
    kl, _, _ = ex(np.zeros((0, 0), np.uint8))              # empty image -> silent return
    assert len(kl) == 0
    with pytest.raises(plslam.PlhError):                   # mask size mismatch is the reference's runtime_error
        ex(np.zeros((64, 96), np.uint8), mask=np.zeros((10, 10), np.uint8))
    for no, sc in ((2, 1.2), (2, 3.0), (3, 2.0), (0, 1.2)):   # what the reference itself cannot run: cv::pyrDown's assertion
        with pytest.raises(plslam.PlhError):                  # ((int)scale != 2), undefined behaviour (three octaves)
            plslam.LINEextractor(no, sc, 20, 0.0, rows=64, cols=96, lib=emu_lib)
    ex.close()


@pytest.mark.parametrize("seed,rows,cols,refine,masked", [(7, 120, 160, 0, False), (8, 121, 163, 1, True)])
def test_emu_line_two_octaves(plslam, oracle, synth, emu_lib, seed, rows, cols, refine, masked):
    """LINEextractor(numOctaves = 2, scale = 2): the second octave's Gaussian pyramid level, its LSD, KeyLines of both octaves in
    one response order, LBD on the gradient images of each line's octave -- every record equal to the oracle's."""
    img = synth.make_frame(seed, rows, cols, n_rect=40, n_line=20)
    mask = None
    if masked:
        mask = np.full((rows, cols), 255, np.uint8)
        mask[20:90, 30:120] = 0
    rk, rd, rf = oracle.line_extract(img, 60, 0.0, mask, refine=refine, num_octaves=2, scale=2.0)
    assert (rk["octave"] == 1).sum() > 5 and (rk["octave"] == 0).sum() > 5
    ex = plslam.LINEextractor(2, 2.0, 60, 0.0, rows=rows, cols=cols, max_batch=1, lib=emu_lib)
    ex.set_refine(refine)
    kl, desc, fn = ex(img, mask)
    assert ex.status() == 0
    ex.close()
    assert _exact(kl, desc, fn, rk, rd, rf)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,refine,waves", [(480, 640, 0, -1), (376, 1241, 1, 0), (203, 405, 0, 4)])
def test_gpu_line_two_octaves(plslam, oracle, synth, rows, cols, refine, waves):
    """Two octaves on the GPU: one frame through the host-buffer call and a batch of four through the device entry point."""
    import torch
    frames = np.stack([synth.make_frame(40 + i, rows, cols) for i in range(4)])
    ref = [oracle.line_extract(f, 200, 0.0, refine=refine, num_octaves=2, scale=2.0) for f in frames]
    ex = plslam.LINEextractor(2, 2.0, 200, 0.0, rows=rows, cols=cols, max_batch=4)
    ex.set_refine(refine)
    ex.set_grow_waves(waves)
    kl, desc, fn = ex(frames[0])
    _match(kl, desc, fn, *ref[0], "two octaves, one frame")
    cap = ex.capacity
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(frames).to(dev)
    d_kl = torch.zeros((4, cap, 17), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((4, cap, 32), dtype=torch.uint8, device=dev)
    d_fn = torch.zeros((4, cap, 3), dtype=torch.float64, device=dev)
    d_n = torch.zeros((4,), dtype=torch.int32, device=dev)
    ex.extract_batch_dev(d_img, 4, rows * cols, d_kl, d_desc, d_fn, d_n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ex.status() == 0
    n = d_n.cpu().numpy()
    klb = d_kl.cpu().numpy().view(np.uint8).reshape(4, cap, 68).copy().view(plslam.KL_DTYPE).reshape(4, cap)
    for b in range(4):
        _match(klb[b, :n[b]], d_desc[b, :n[b]].cpu().numpy(), d_fn[b, :n[b]].cpu().numpy(), *ref[b], "two octaves, batch frame %d" % b)
        assert (ref[b][0]["octave"] == 1).sum() > 0
    ex.close()


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("seed,rows,cols,nf,minlen,undist", [(1, 480, 640, 200, 0.0, False), (2, 480, 640, 200, 20.0, True),
                                                            (1000, 376, 1241, 200, 0.0, False), (5, 480, 640, 50, 0.0, False),
                                                            (6, 240, 320, 100, 0.0, "outside")])
@pytest.mark.parametrize("refine", [None, 0], ids=["default-adv", "std"])
def test_gpu_line_extract(plslam, oracle, synth, seed, rows, cols, nf, minlen, undist, refine):
    img = synth.make_frame(seed, rows, cols)
    K, D = (TUM1_K, TUM1_D) if undist else (None, None)
    if undist == "outside":   # principal point off the image: a fifth of the map leaves the frame
        K, D = [300.0, 300.0, 40.0, 190.0], [0.4, -0.9, 0.01, -0.008, 1.1]
    rk, rd, rf, rs = _oracle_line(oracle, img, nf, minlen, K, D, refine=refine)
    ex = plslam.LINEextractor(1, 1.2, nf, minlen, rows=rows, cols=cols, max_batch=1, K=K, D=D)
    if refine is not None:   # (None: what a new handle runs -- the library's default, LSD_REFINE_ADV)
        ex.set_refine(refine)
    kl, desc, fn = ex(img)
    gs = ex.read_segments(0)
    ex.close()
    if os.environ.get("PLSLAM_LINE_TOLERANCE") == "1":
        assert abs(len(gs) - len(rs)) <= max(2, len(rs) // 100)
    else:
        assert len(gs) == len(rs) and (gs == rs).all(), "seed %d: LSD segments differ from the oracle" % seed
    _match(kl, desc, fn, rk, rd, rf, "seed %d" % seed)


@pytest.mark.gpu
@pytest.mark.parametrize("refine", [1, 0], ids=["adv", "std"])
@pytest.mark.parametrize("waves", [0, 2, 3, 4, 8, 16])
def test_gpu_line_grow_waves(plslam, oracle, synth, waves, refine):
    """Region growing with `waves` wavefronts per frame (k_lsd_grow_mw; 0 = k_lsd_grow_lone): segments, KeyLines and LBD bytes
    equal the oracle's on textured, sparse, sawtooth (huge regions) and white-noise frames, repeatedly (the schedule of the
    transactions differs from run to run, the result must not)."""
    rng = np.random.RandomState(5)
    imgs = [synth.make_frame(31, 480, 640), synth.make_frame(32, 480, 640, n_rect=20, n_line=10),
            _sawtooth(480, 640, period=42, slope=6), rng.randint(0, 256, (480, 640)).astype(np.uint8),
            synth.make_frame(33, 480, 640)]
    ex = plslam.LINEextractor(1, 1.2, 200, 0.0, rows=480, cols=640, max_batch=1)
    ex.set_grow_waves(waves)
    ex.set_refine(refine)
    for k, img in enumerate(imgs):
        rk, rd, rf, rs = _oracle_line(oracle, img, 200, 0.0, refine=refine)
        for rep in range(3):
            kl, desc, fn = ex(img)
            gs = ex.read_segments(0)
            assert ex.status() == 0
            assert len(gs) == len(rs) and (gs == rs).all(), "image %d, run %d: LSD segments differ from the oracle" % (k, rep)
            _match(kl, desc, fn, rk, rd, rf, "image %d, run %d" % (k, rep))
    ex.close()


@pytest.mark.gpu
@pytest.mark.parametrize("waves,B", [(4, 24), (2, 48), (8, 5)])
def test_gpu_line_grow_waves_batch(plslam, oracle, synth, waves, B):
    """A batch of frames, several wavefronts per frame: every frame equals the oracle; the mark planes are clean afterwards
    (a second pass over different frames in the same workspace is exact too)."""
    import torch
    ex = plslam.LINEextractor(1, 1.2, 200, 0.0, rows=240, cols=320, max_batch=B)
    ex.set_grow_waves(waves)
    cap = ex.capacity
    dev = torch.device("cuda", 0)
    for seed in (50, 90):
        frames = synth.make_frames(seed, B, 240, 320)
        d_img = torch.from_numpy(frames).to(dev)
        d_kl = torch.zeros((B, cap, 17), dtype=torch.float32, device=dev)
        d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        d_fn = torch.zeros((B, cap, 3), dtype=torch.float64, device=dev)
        d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
        ex.extract_batch_dev(d_img, B, 240 * 320, d_kl, d_desc, d_fn, d_n, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert ex.status() == 0
        n = d_n.cpu().numpy()
        kl = d_kl.cpu().numpy().view(np.uint8).reshape(B, cap, 68).copy().view(plslam.KL_DTYPE).reshape(B, cap)
        desc, fn = d_desc.cpu().numpy(), d_fn.cpu().numpy()
        for b in range(B):
            rk, rd, rf = oracle.line_extract(frames[b], 200, 0.0)
            _match(kl[b, :n[b]], desc[b, :n[b]], fn[b, :n[b]], rk, rd, rf, "seed %d frame %d" % (seed, b))
    ex.close()


@pytest.mark.gpu
@pytest.mark.parametrize("refine", [1, 0], ids=["adv", "std"])
@pytest.mark.parametrize("rows,cols,B", [(376, 1241, 512), (480, 640, 320)], ids=["1241x376x512", "640x480x320"])
def test_gpu_line_mw16_many_frames(plslam, oracle, synth, rows, cols, B, refine):
    """The launch shape of BASELINE configs[4]'s per-GPU share: hundreds of frames with the AUTOMATIC wavefront policy, i.e.
    batch x 8 wavefronts > 2048, which launch_lsd_grow() serves with the 128-register build k_lsd_grow_mw16 (the default soak
    stops at 256 x 8 = 2048 = the roomy k_lsd_grow_mw).  Distinct frames of the soak's mix, both refine levels, every LSD segment,
    KeyLine, LBD byte and line equation against the oracle."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from test_soak_gpu import soak_frames
    assert B * 8 > 2048
    frames = soak_frames(synth, rows, cols, B)

    def one(img):
        return oracle.line_extract(img, 200, 0.0, refine=refine) + (oracle.lsd_detect(img, refine=refine),)
    with ThreadPoolExecutor(os.cpu_count() or 1) as pool:
        ref = list(pool.map(one, list(frames)))
    ex = plslam.LINEextractor(1, 1.2, 200, 0.0, rows=rows, cols=cols, max_batch=B)
    ex.set_refine(refine)           # (grow waves stay automatic: 8 per frame for <= 1024 frames)
    cap = ex.capacity
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(frames).to(dev)
    d_kl = torch.zeros((B, cap, 17), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_fn = torch.zeros((B, cap, 3), dtype=torch.float64, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    nseg = 0
    for rep in range(2):            # (the schedule of the transactions differs from run to run, the result must not)
        d_n.zero_()
        ex.extract_batch_dev(d_img, B, rows * cols, d_kl, d_desc, d_fn, d_n, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert ex.status() == 0
        n = d_n.cpu().numpy()
        kl = d_kl.cpu().numpy().view(np.uint8).reshape(B, cap, 68).copy().view(plslam.KL_DTYPE).reshape(B, cap)
        desc, fn = d_desc.cpu().numpy(), d_fn.cpu().numpy()
        for b in range(B):
            rk, rd, rf, rs = ref[b]
            gs = ex.read_segments(b)
            assert len(gs) == len(rs) and (gs == rs).all(), "run %d, frame %d: LSD segments differ from the oracle" % (rep, b)
            _match(kl[b, :n[b]], desc[b, :n[b]], fn[b, :n[b]], rk, rd, rf, "run %d, frame %d" % (rep, b))
            nseg += len(rs)
    ex.close()
    print("\nmw16 %dx%d x %d frames, refine %d: %d segments bit-exact over 2 runs" % (cols, rows, B, refine, nseg))


@pytest.mark.gpu
@pytest.mark.parametrize("waves", [0, -1, 4])
def test_gpu_line_refine_adv(plslam, oracle, synth, waves):
    """LSD_REFINE_ADV on the GPU (k_lsd_grow_adv / k_lsd_grow_mw_adv) equals the oracle's ADV level: single frames of several
    kinds, and a batch through the device entry point."""
    import torch
    rng = np.random.RandomState(11)
    imgs = [synth.make_frame(61, 480, 640), synth.make_frame(62, 480, 640, n_rect=20, n_line=10), _sawtooth(480, 640, period=42, slope=6),
            rng.randint(0, 256, (480, 640)).astype(np.uint8), synth.make_frame(63, 480, 640, n_rect=400, n_line=200)]
    ex = plslam.LINEextractor(1, 1.2, 200, 0.0, rows=480, cols=640, max_batch=1)
    ex.set_refine(1)
    ex.set_grow_waves(waves)
    nstd = nadv = 0
    for k, img in enumerate(imgs):
        rk, rd, rf = oracle.line_extract(img, 200, 0.0, refine=1)
        rs = oracle.lsd_detect(img, refine=1)
        kl, desc, fn = ex(img)
        gs = ex.read_segments(0)
        assert ex.status() == 0
        assert len(gs) == len(rs) and (gs == rs).all(), "image %d: LSD segments (ADV) differ from the oracle" % k
        _match(kl, desc, fn, rk, rd, rf, "ADV image %d" % k)
        nstd += len(oracle.lsd_detect(img, refine=0)); nadv += len(rs)
    ex.close()
    assert nadv < nstd
    B = 16
    frames = synth.make_frames(70, B, 240, 320)
    exb = plslam.LINEextractor(1, 1.2, 200, 0.0, rows=240, cols=320, max_batch=B)
    exb.set_refine(1)
    exb.set_grow_waves(waves)
    cap = exb.capacity
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(frames).to(dev)
    d_kl = torch.zeros((B, cap, 17), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_fn = torch.zeros((B, cap, 3), dtype=torch.float64, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    exb.extract_batch_dev(d_img, B, 240 * 320, d_kl, d_desc, d_fn, d_n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert exb.status() == 0
    n = d_n.cpu().numpy()
    kl = d_kl.cpu().numpy().view(np.uint8).reshape(B, cap, 68).copy().view(plslam.KL_DTYPE).reshape(B, cap)
    desc, fn = d_desc.cpu().numpy(), d_fn.cpu().numpy()
    for b in range(B):
        rk, rd, rf = oracle.line_extract(frames[b], 200, 0.0, refine=1)
        _match(kl[b, :n[b]], desc[b, :n[b]], fn[b, :n[b]], rk, rd, rf, "ADV batch frame %d" % b)
    exb.close()


@pytest.mark.gpu
def test_gpu_line_batch_and_mask(plslam, oracle, synth):
    import torch
    B = 12
    frames = synth.make_frames(40, B, 480, 640)
    ex = plslam.LINEextractor(1, 1.2, 200, 0.0, rows=480, cols=640, max_batch=B)
    cap = ex.capacity
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(frames).to(dev)
    d_kl = torch.zeros((B, cap, 17), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_fn = torch.zeros((B, cap, 3), dtype=torch.float64, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    ex.extract_batch_dev(d_img, B, 480 * 640, d_kl, d_desc, d_fn, d_n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    kl = d_kl.cpu().numpy().view(np.uint8).reshape(B, cap, 68).copy().view(plslam.KL_DTYPE).reshape(B, cap)
    desc, fn = d_desc.cpu().numpy(), d_fn.cpu().numpy()
    nexact = 0
    for b in range(B):
        rk, rd, rf = oracle.line_extract(frames[b], 200, 0.0)
        nexact += _match(kl[b, :n[b]], desc[b, :n[b]], fn[b, :n[b]], rk, rd, rf, "frame %d" % b)
    print("batch: %d/%d frames bit-exact" % (nexact, B))
    mask = np.zeros((480, 640), np.uint8)                  # mask through the single-frame entry point
    mask[:, :320] = 255
    k1, d1, f1 = ex(frames[0], mask)
    rk, rd, rf = oracle.line_extract(frames[0], 200, 0.0, mask)
    _match(k1, d1, f1, rk, rd, rf, "mask")
    ex.close()


@pytest.mark.gpu
@pytest.mark.parametrize("refine", [1, 0], ids=["adv", "std"])
def test_gpu_keylines_thousands_of_equal_lines(plslam, oracle, refine):
    """KeyLine selection when the responses do not separate the lines: a lattice of identical squares gives thousands of segments
    of a handful of distinct lengths, so the candidate set of k_keylines' histogram step exceeds its LDS list and the selection
    falls back to the repeated arg-max over all keys -- the order among equal responses is the detection index, as in the
    reference's stable sort (LineExtractor.cpp:43)."""
    rows, cols = 480, 640
    img = np.full((rows, cols), 30, np.uint8)
    for y in range(6, rows - 18, 20):
        for x in range(6, cols - 18, 20):
            img[y:y + 12, x:x + 12] = 220
    rs = oracle.lsd_detect(img, refine=refine)
    # k_keylines' candidates: the 256-bin histogram of the responses (length / 640 x 170), bins from the top until 201 keys are in
    length = np.hypot(rs[:, 0] - rs[:, 2], rs[:, 1] - rs[:, 3]).astype(np.float32)
    bins = np.minimum(255, (length / np.float32(cols) * np.float32(170.0)).astype(int))
    cand, cum = 0, 0
    for b in sorted(set(bins.tolist()), reverse=True):
        cum += int((bins == b).sum())
        cand = cum
        if cum >= 201:
            break
    assert cand > 1024, (len(rs), cand)   # more candidates than the kernel's LDS list holds: the fallback runs
    rk, rd, rf = oracle.line_extract(img, 200, 0.0, refine=refine)
    ex = plslam.LINEextractor(1, 1.2, 200, 0.0, rows=rows, cols=cols, max_batch=1)
    ex.set_refine(refine)
    kl, desc, fn = ex(img)
    gs = ex.read_segments(0)
    ex.close()
    assert len(gs) == len(rs) and (gs == rs).all()
    _match(kl, desc, fn, rk, rd, rf, "lattice of squares, refine %d" % refine)


@pytest.mark.gpu
def test_gpu_line_golden(plslam, synth):
    """GPU vs the committed golden vectors (tests/golden/line_*.npz, tools/gen_golden.py)."""
    import glob
    for path in sorted(glob.glob(os.path.join(_util.ROOT, "tests", "golden", "line_*.npz"))):
        g = np.load(path)
        rows, cols = int(g["rows"]), int(g["cols"])
        img = synth.make_frame(int(g["seed"]), rows, cols, n_rect=int(g["n_rect"]), n_line=int(g["n_line"]))
        ex = plslam.LINEextractor(1, 1.2, int(g["nfeature"]), float(g["minlen"]), rows=rows, cols=cols, max_batch=1)
        ex.set_refine(int(g["refine"]) if "refine" in g.files else 0)   # line_adv_*: LSD_REFINE_ADV; the files of rounds 1-4: STD
        kl, desc, fn = ex(img)
        gs = ex.read_segments(0)
        ex.close()
        assert gs.shape == g["segs"].shape and (gs == g["segs"]).all(), os.path.basename(path)
        _match(kl, desc, fn, g["keylines"], g["desc"], g["linefn"], os.path.basename(path))


def test_emu_batch_of_ten_xcd_block_order(plslam, oracle, synth, emu_lib):
    """Ten small frames in one call on the emulator build, where plh_xcd_decode (plh_common.h) switches to the XCD-aware block order
    from eight frames on (the product build: from 64): block L -> frame 8 (L / 8 / blocks per frame) + L % 8, the last group of eight
    ragged (frames 8, 9 + six that do not exist).  Every kernel that decodes its grid that way -- pyramid, orientation + rBRIEF, remap,
    blur, Sobel, LBD, the LSD_REFINE_ADV walks -- must touch every frame exactly once: all records equal the oracle's."""
    B, rows, cols = 10, 120, 160
    frames = np.stack([synth.make_frame(30 + b, rows, cols, n_rect=40, n_line=20) for b in range(B)])
    K, D = [150.0, 150.0, 80.0, 60.0], TUM1_D
    ex = plslam.LINEextractor(1, 1.2, 40, 0.0, rows=rows, cols=cols, max_batch=B, lib=emu_lib, K=K, D=D)
    ex.set_grow_waves(0)
    cap = ex.capacity
    kl = np.zeros((B, cap), plslam.KL_DTYPE)
    desc = np.zeros((B, cap, 32), np.uint8)
    fn = np.zeros((B, cap, 3), np.float64)
    n = np.full(B, -7, np.int32)
    ex.extract_batch_dev(frames, B, rows * cols, kl, desc, fn, n)   # (the emulator's device memory is host memory)
    assert ex.status() == 0
    for b in range(B):
        rk, rd, rf, _ = _oracle_line(oracle, frames[b], 40, 0.0, K, D)
        assert n[b] == len(rk) and _exact(kl[b, :n[b]], desc[b, :n[b]], fn[b, :n[b]], rk, rd, rf), b
    ex.close()
    orb = plslam.ORBextractor(200, 1.2, 3, 20, 7, rows=rows, cols=cols, max_batch=B, lib=emu_lib)
    kps, od, on = orb.extract_batch(frames)
    ref = oracle.OrbOracle(200, 1.2, 3, 20, 7)
    for b in range(B):
        rk, rd = ref.extract(frames[b])
        assert on[b] == len(rk) and (od[b, :on[b]] == rd).all(), b
        assert all((kps[b, :on[b]][f] == rk[f]).all() for f in rk.dtype.names), b
    orb.close()
