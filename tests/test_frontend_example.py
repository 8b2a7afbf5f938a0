"""examples/batch_frontend.cc -- a plain C++ host that drives the whole batch front end through plh_frontend_* (the C++ side
of the throughput path: sub-batches, stream pairs, events live in the library) -- built by __graft_entry__.build() and run
here on the GPU; everything it writes is compared with the oracle, record by record."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _util
from test_line import TUM1_D, TUM1_K


def _entry():
    sys.path.insert(0, _util.ROOT)
    import __graft_entry__ as g
    return g


def test_example_builds_and_links():
    g = _entry()
    exe = g.build_examples()
    assert os.access(exe, os.X_OK)
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    for sym in ("plh_frontend_create", "plh_frontend_step", "plh_frontend_join", "plh_frontend_records_of", "plh_frontend_status",
                "plh_vocab_load_text"):
        assert sym in out, sym


def _read_out(path, P):
    raw = open(path, "rb").read()
    hdr = np.frombuffer(raw, np.int32, 4, 0)
    off = 16
    parts = []
    for _ in range(int(hdr[1])):
        first, B, oc, lc = [int(v) for v in np.frombuffer(raw, np.int32, 4, off)]
        off += 16

        def take(dtype, count, shape):
            nonlocal off
            a = np.frombuffer(raw, dtype, count, off).reshape(shape)
            off += a.nbytes
            return a
        r = {"first": first, "B": B}
        r["n"] = take(np.int32, B, (B,))
        r["kps"] = take(np.uint8, B * oc * 28, (B, oc, 28)).copy().view(P.KP_DTYPE).reshape(B, oc)
        r["desc"] = take(np.uint8, B * oc * 32, (B, oc, 32))
        r["nid"] = take(np.int32, B * oc, (B, oc))
        r["word"] = take(np.int32, B * oc, (B, oc))
        r["bow_n"] = take(np.int32, B, (B,))
        r["bow_word"] = take(np.int32, B * oc, (B, oc))
        r["bow_value"] = take(np.float64, B * oc, (B, oc))
        r["nl"] = take(np.int32, B, (B,))
        r["kl"] = take(np.uint8, B * lc * 68, (B, lc, 68)).copy().view(P.KL_DTYPE).reshape(B, lc)
        r["ldesc"] = take(np.uint8, B * lc * 32, (B, lc, 32))
        r["lfn"] = take(np.float64, B * lc * 3, (B, lc, 3))
        r["nm_orb"] = take(np.int32, B, (B,))
        r["m_orb"] = take(np.int32, B * oc, (B, oc))
        r["nm_line"] = take(np.int32, B, (B,))
        r["m_line"] = take(np.int32, B * lc, (B, lc))
        parts.append(r)
    assert off == len(raw)
    return parts


@pytest.mark.gpu
@pytest.mark.parametrize("undist", [False, True])
def test_cpp_host_matches_the_oracle(plslam, oracle, synth, tmp_path, undist):
    g = _entry()
    exe = g.build_examples()
    sys.path.insert(0, _util.ROOT)
    import bench
    V = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    B, ns, rows, cols, nfeat, nlines = 12, 3, 480, 640, 1000, 200
    frames = synth.make_frames(900 + int(undist), B, rows, cols)
    voc = V.Vocabulary.synthetic(31, k=10, L=4, synth=synth, idf=True)
    fbin, vtxt, out = str(tmp_path / "frames.bin"), str(tmp_path / "voc.txt"), str(tmp_path / "out.bin")
    frames.tofile(fbin)
    voc.save_text(vtxt)
    voc = V.Vocabulary.load_text(vtxt)      # the weights as the text file holds them (what the C++ host loads)
    cmd = [exe, fbin, str(rows), str(cols), str(B), str(ns), str(nfeat), str(nlines), "3", out, vtxt]
    K, D = (TUM1_K, TUM1_D) if undist else (None, None)
    # (without a camera on the command line the example takes the one of the frame size -- TUM1 with its distortion for
    # 640x480 -- so the undistorted case says so: zero distortion = no remap, Frame.cc:917-921)
    cmd += ["%r" % v for v in (K + D if undist else TUM1_K + [0.0] * 5)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "frames/s" in run.stdout
    parts = _read_out(out, plslam)
    assert sum(p["B"] for p in parts) == B and [p["first"] for p in parts] == [0, 4, 8]
    for p in parts:
        idx = list(range(p["first"], p["first"] + p["B"])) + [p["first"]]     # the last frame of a sub-batch is matched against its first
        recs, pairs = bench.oracle_records(oracle, V, frames[idx], voc, nfeat, 8, nlines, K, D)
        bad = bench.verify_records(p, recs[:-1], pairs)
        assert not bad, bad
        assert len(recs) - 1 == p["B"] and len(pairs) == p["B"]


@pytest.mark.gpu
def test_cpp_host_on_real_frames_pgm_and_refine_adv(plslam, oracle, synth, tmp_path):
    """Real images end to end: the reference's masks/tum_mask.png and masks/mask.png (the only real 640x480 images it ships;
    Tracking.cc:83-84) and shifted copies, written as PGM files, through `batch_frontend <directory>` -- the TUM1 camera is
    picked from the frame size -- with --refine adv; every record against the oracle (ADV)."""
    g = _entry()
    exe = g.build_examples()
    sys.path.insert(0, _util.ROOT)
    import bench
    V = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    FIO = _util._load("plslam_amd_frames_io", os.path.join(_util.ROOT, "pl-slam_amd", "frames_io.py"))
    gm = np.load(os.path.join(_util.ROOT, "tests", "golden", "ref_masks.npz"))
    base = [np.unpackbits(gm[k])[:480 * 640].reshape(480, 640).astype(np.uint8) * 255 for k in ("tum_mask", "mask")]
    scene = synth.make_frame(5, 480, 640)
    imgs = []
    for i in range(8):   # the masks as they are, shifted, and blended into a scene (grey levels, not only 0 / 255)
        m = np.roll(base[i % 2], 17 * (i // 2), axis=1)
        imgs.append(m if i < 4 else (m // 2 + scene // 2).astype(np.uint8))
    d = tmp_path / "seq"
    d.mkdir()
    for i, im in enumerate(imgs):
        FIO.write_pgm(str(d / ("frame_%03d.pgm" % i)), im)
    frames = FIO.load_frames(str(d), 480, 640)
    assert frames.shape == (8, 480, 640) and all((frames[i] == imgs[i]).all() for i in range(8))
    voc = V.Vocabulary.synthetic(31, k=10, L=4, synth=synth, idf=True)
    vtxt, out = str(tmp_path / "voc.txt"), str(tmp_path / "out.bin")
    voc.save_text(vtxt)
    voc = V.Vocabulary.load_text(vtxt)
    cmd = [exe, str(d), "480", "640", "8", "2", "1000", "200", "2", out, vtxt, "--refine", "adv"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "TUM1" in run.stdout and "LSD_REFINE_ADV" in run.stdout
    parts = _read_out(out, plslam)
    for p in parts:
        idx = list(range(p["first"], p["first"] + p["B"])) + [p["first"]]
        recs, pairs = bench.oracle_records(oracle, V, frames[idx], voc, 1000, 8, 200, TUM1_K, TUM1_D, refine=1)
        bad = bench.verify_records(p, recs[:-1], pairs)
        assert not bad, bad
    assert sum(int(p["nl"].sum()) for p in parts) > 50      # lines were found on the real images


def test_frames_io_roundtrip(tmp_path):
    FIO = _util._load("plslam_amd_frames_io", os.path.join(_util.ROOT, "pl-slam_amd", "frames_io.py"))
    rng = np.random.RandomState(3)
    a = rng.randint(0, 256, (5, 30, 40)).astype(np.uint8)
    FIO.write_pgm(str(tmp_path / "b.pgm"), a[0])
    (tmp_path / "c.pgm").write_bytes(b"P5\n# a comment\n40 30\n# another\n255\n" + a[1].tobytes())
    a[2:].tofile(str(tmp_path / "d.bin"))
    got = FIO.load_frames(str(tmp_path), 30, 40)
    assert got.shape == (5, 30, 40) and (got == a).all()
    assert (FIO.tile_frames(got, 12)[7] == a[2]).all()
    with pytest.raises(ValueError):
        FIO.load_frames(str(tmp_path / "b.pgm"), 31, 40)
    (tmp_path / "e.pgm").write_bytes(b"P2\n1 1\n255\n0\n")
    with pytest.raises(ValueError):
        FIO.read_pgm(str(tmp_path / "e.pgm"))


def test_emu_frontend_matches_the_oracle(plslam, oracle, synth, emu_lib):
    """plh_frontend_* on the emulator build (CPU): two sub-batches, records in the library's own buffers (host memory there),
    every record against the oracle -- the host logic (slot copies, successor matching, sub-batch offsets) without a GPU."""
    import ctypes as C
    sys.path.insert(0, _util.ROOT)
    import bench
    P = plslam
    V = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    L = P.load(emu_lib)
    B, ns, rows, cols, nfeat, nlines, nlev = 4, 2, 120, 160, 300, 40, 4
    frames = np.ascontiguousarray(synth.make_frames(77, B, rows, cols))
    voc = V.Vocabulary.synthetic(31, k=6, L=3, synth=synth, idf=True)
    hv = P.ORBVocabulary(lib=emu_lib)
    parent, leaf = voc.tree_arrays()
    hv.create(voc.k, voc.L, parent, leaf, voc.node_desc, voc.weight64)
    fp = P.FrontendParams()   # zero-initialised, as a C caller's memset leaves it: lsd_refine = 0 = PLH_FRONTEND_REFINE_LIBRARY
    fp.rows, fp.cols = rows, cols
    fp.orb = P.OrbParams(nfeat, 1.2, nlev, 20, 7)
    fp.line = P.LineParams(1, 1.2, nlines, 0.0)
    fp.bow_levelsup, fp.orb_th_low, fp.orb_nnratio, fp.orb_check_orientation, fp.line_th, fp.line_nnratio = 4, 50, 0.7, 1, 50.0, 0.7
    assert fp.lsd_refine == 0 and L.plh_lsd_refine_default() == 1   # ... which is the library's default: LSD_REFINE_ADV
    h = C.c_void_p()
    # a caller compiled against another header (no / a different struct_size) is refused, not read past its struct
    assert L.plh_frontend_create(C.byref(fp), hv.h, B, ns, 0, C.byref(h)) == 1 and not h.value  # PLH_ERR_INVALID
    assert b"struct_size" in L.plh_last_error()
    fp.struct_size = C.sizeof(P.FrontendParams) - 4
    assert L.plh_frontend_create(C.byref(fp), hv.h, B, ns, 0, C.byref(h)) == 1 and not h.value  # PLH_ERR_INVALID
    fp.struct_size = C.sizeof(P.FrontendParams)
    for wrong in (7, 1, 2):   # 1 = PLH_LSD_REFINE_ADV assigned out of habit (round 4's encoding), 1 / 2 = round 5's: refused, not guessed
        fp.lsd_refine = wrong
        assert L.plh_frontend_create(C.byref(fp), hv.h, B, ns, 0, C.byref(h)) == 1 and not h.value  # PLH_ERR_INVALID
        assert b"PLH_FRONTEND_REFINE" in L.plh_last_error()
    fp.lsd_refine = 0
    P._check(L, L.plh_frontend_create(C.byref(fp), hv.h, B, ns, 0, C.byref(h)), "plh_frontend_create")
    try:
        assert L.plh_frontend_parts(h) == ns
        P._check(L, L.plh_frontend_step(h, P._p(frames), rows * cols, None, 1), "plh_frontend_step")
        f = C.c_int(-1)
        P._check(L, L.plh_frontend_status(h, C.byref(f)), "plh_frontend_status")
        assert f.value == 0
        for part in range(ns):
            r = P.FrontendRecords()
            P._check(L, L.plh_frontend_records_of(h, part, C.byref(r)), "plh_frontend_records_of")
            Bp, oc, lc = r.frames, r.orb_capacity, r.line_capacity
            assert r.first == part * (B // ns) and Bp == B // ns

            def arr(ptr, dtype, shape):
                n = int(np.prod(shape)) * np.dtype(dtype).itemsize
                return np.frombuffer((C.c_uint8 * n).from_address(ptr), np.uint8).view(dtype).reshape(shape).copy()
            res = dict(n=arr(r.n, np.int32, (Bp,)), kps=arr(r.kps, np.uint8, (Bp, oc, 28)).view(P.KP_DTYPE).reshape(Bp, oc),
                       desc=arr(r.desc, np.uint8, (Bp, oc, 32)), nid=arr(r.nid, np.int32, (Bp, oc)), word=arr(r.word, np.int32, (Bp, oc)),
                       bow_n=arr(r.bow_n, np.int32, (Bp,)), bow_word=arr(r.bow_word, np.int32, (Bp, oc)),
                       bow_value=arr(r.bow_value, np.float64, (Bp, oc)), nl=arr(r.nl, np.int32, (Bp,)),
                       kl=arr(r.kl, np.uint8, (Bp, lc, 68)).view(P.KL_DTYPE).reshape(Bp, lc), ldesc=arr(r.ldesc, np.uint8, (Bp, lc, 32)),
                       lfn=arr(r.lfn, np.float64, (Bp, lc, 3)), nm_orb=arr(r.nm_orb, np.int32, (Bp,)), m_orb=arr(r.m_orb, np.int32, (Bp, oc)),
                       nm_line=arr(r.nm_line, np.int32, (Bp,)), m_line=arr(r.m_line, np.int32, (Bp, lc)))
            idx = list(range(r.first, r.first + Bp)) + [r.first]
            recs, pairs = bench.oracle_records(oracle, V, frames[idx], voc, nfeat, nlev, nlines, None, None)
            mism = bench.verify_records(res, recs[:-1], pairs)
            assert not mism, mism
        bad = P.FrontendParams()
        assert L.plh_frontend_create(C.byref(bad), hv.h, 3, 2, 0, C.byref(C.c_void_p())) != 0      # batch not a multiple of nsplit
        assert L.plh_frontend_bind_records(h, 0, C.byref(P.FrontendRecords())) != 0                   # not created for external records
    finally:
        L.plh_frontend_destroy(h)
        hv.close()
