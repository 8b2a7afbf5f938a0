"""GPU end-to-end: the batch front end bench.py times (pl-slam_amd/pipeline.py) vs the oracle, stage by stage."""
import ctypes as C
import os

import numpy as np
import pytest

import _util
from test_line import TUM1_D, TUM1_K, _match, _oracle_line

pytestmark = pytest.mark.gpu


KITTI_K = [718.856, 718.856, 607.1928, 185.2157]      # Examples/Monocular/KITTI00-02.yaml:8-11
KITTI_D = [0.0, 0.0, 0.0, 0.0, 0.0]                   # :13-16 (rectified sequence: no remap in front of LSD)


@pytest.mark.parametrize("rows,cols,nfeat,K,D,seed", [(480, 640, 1000, TUM1_K, TUM1_D, 500), (376, 1241, 2000, KITTI_K, KITTI_D, 1500)],
                         ids=["tum1_640x480_1000", "kitti_1241x376_2000"])
@pytest.mark.parametrize("refine", [None, 0], ids=["default-adv", "std"])   # cv::LineSegmentDetector's level: the library's default
def test_front_end_batch_vs_oracle(plslam, oracle, synth, rows, cols, nfeat, K, D, seed, refine):   # (LSD_REFINE_ADV) and STD
    """The whole batch front end against the oracle, stage by stage, on both frame shapes the north star names:
    TUM1 (Examples/Monocular/TUM1.yaml: 640x480, 1000 features, distorted) and KITTI 00-02
    (Examples/Monocular/KITTI00-02.yaml:8-51: 1241x376, 2000 features, 4 quad-tree roots, no distortion)."""
    import torch
    V = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    PL = _util._load("plslam_amd_pipeline", os.path.join(_util.ROOT, "pl-slam_amd", "pipeline.py"))
    B = 6
    frames = synth.make_frames(seed, B, rows, cols)
    voc = V.Vocabulary.synthetic(102, k=10, L=6, synth=synth, idf=True)
    fe = PL.FrontEndBatch(plslam, voc, B, rows, cols, nfeat, 8, 200, 0.0, K, D)
    if refine is not None:
        fe.line.set_refine(refine)
    d = torch.from_numpy(frames).cuda()
    fe.step(d)
    fe.step(d)          # second pass over the same buffers: results must not depend on stale state
    r = fe.results()
    fe.close()
    O = oracle
    L = O.lib()
    L.plo_bow_transform.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.plo_bow_transform.restype = None
    L.plo_bow_vector.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.plo_bow_vector.restype = C.c_int
    orb = O.OrbOracle(nfeat, 1.2, 8, 20, 7)
    ref = []
    for b in range(B):
        rk, rd = orb.extract(frames[b])
        n = r["n"][b]
        assert n == len(rk) and n >= nfeat
        for f in rk.dtype.names:
            assert (r["kps"][b, :n][f] == rk[f]).all(), (b, f)
        assert (r["desc"][b, :n] == rd).all()
        nid = np.zeros(n, np.int32)
        word = np.zeros(n, np.int32)
        L.plo_bow_transform(O._p(rd), n, O._p(voc.node_desc), O._p(voc.child_start), O._p(voc.child_count), O._p(voc.word_id),
                            O._p(voc.weight), voc.L, 4, O._p(nid), O._p(word))
        assert (r["nid"][b, :n] == nid).all() and (r["word"][b, :n] == word).all()
        bw, bv = np.zeros(n, np.int32), np.zeros(n, np.float64)      # mBowVec (TF_IDF / L1_NORM like ORBvoc)
        m = L.plo_bow_vector(O._p(word), n, O._p(voc.word_weight()), 0, 0, O._p(bw), O._p(bv), n)
        assert r["bow_n"][b] == m and (r["bow_word"][b, :m] == bw[:m]).all() and (r["bow_value"][b, :m] == bv[:m]).all()
        lk, ld, lf, _ = _oracle_line(O, frames[b], 200, 0.0, *((K, D) if any(D) else (None, None)), refine=refine)
        nl = r["nl"][b]
        _match(r["kl"][b, :nl], r["ldesc"][b, :nl], r["lfn"][b, :nl], lk, ld, lf, "frame %d" % b)
        ref.append((rk, rd, nid, ld))
    for b in range(B):       # frame b (KeyFrame) -> frame (b+1) % B
        k1, d1, n1, l1 = ref[b]
        k2, d2, n2, l2 = ref[(b + 1) % B]
        m = np.zeros(len(d2), np.int32)
        valid = np.ones(len(d1), np.uint8)
        a1, a2 = np.ascontiguousarray(k1["angle"]), np.ascontiguousarray(k2["angle"])
        c = L.plo_orb_search_by_bow(O._p(d1), O._p(a1), O._p(n1), O._p(valid), len(d1), O._p(d2), O._p(a2), O._p(n2), len(d2),
                                    50, 0.7, 1, O._p(m))
        assert r["nm_orb"][b] == c and (r["m_orb"][b, :len(d2)] == m).all(), b
        ml = np.zeros(len(l1), np.int32)
        cl = L.plo_line_search_double(O._p(l1), len(l1), O._p(l2), len(l2), 50.0, 0.7, O._p(ml))
        assert r["nm_line"][b] == cl and (r["m_line"][b, :len(l1)] == ml).all(), b


def test_pipelined_sub_batches_match_plain_batches(plslam, synth):
    """FrontEndPipelined (staggered sub-batches, steps not joined) produces exactly what FrontEndBatch produces on each
    sub-batch (which the test above checks against the oracle)."""
    import torch
    V = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    PL = _util._load("plslam_amd_pipeline", os.path.join(_util.ROOT, "pl-slam_amd", "pipeline.py"))
    B, ns = 8, 2
    frames = synth.make_frames(510, B, 480, 640)
    voc = V.Vocabulary.synthetic(102, k=10, L=6, synth=synth)
    d = torch.from_numpy(frames).cuda()
    fp = PL.FrontEndPipelined(plslam, voc, B, 480, 640, 1000, 8, 200, 0.0, TUM1_K, TUM1_D, nsplit=ns)
    for _ in range(3):
        fp.step(d, join=False)
    rp = fp.results()
    fp.close()
    fe = PL.FrontEndBatch(plslam, voc, B // ns, 480, 640, 1000, 8, 200, 0.0, TUM1_K, TUM1_D)
    for k in range(ns):
        fe.step(d[k * (B // ns):(k + 1) * (B // ns)])
        r = fe.results()
        sl = slice(k * (B // ns), (k + 1) * (B // ns))
        Bp = B // ns
        assert (rp["n"][sl] == r["n"]).all() and (rp["nl"][sl] == r["nl"]).all()
        for key in r:       # rows beyond the per-frame counts are unspecified: compare the live ones
            for i in range(Bp):
                cnt = {"kps": r["n"][i], "desc": r["n"][i], "nid": r["n"][i], "word": r["n"][i], "bow_word": r["bow_n"][i],
                       "bow_value": r["bow_n"][i], "kl": r["nl"][i], "ldesc": r["nl"][i],
                       "lfn": r["nl"][i], "m_orb": r["n"][(i + 1) % Bp], "m_line": r["nl"][i]}.get(key)
                a, b = rp[key][sl][i], r[key][i]
                if cnt is not None:
                    a, b = a[:cnt], b[:cnt]
                if a.dtype.names:
                    assert all((a[f] == b[f]).all() for f in a.dtype.names), (k, key, i)
                else:
                    assert np.array_equal(a, b, equal_nan=True), (k, key, i)
    fe.close()


def test_partial_batches_and_blank_frames(plslam, oracle, synth):
    """A handle planned for 8 frames is driven with 3 (ragged use), and a batch that contains feature-less frames (flat /
    nearly flat images: zero keypoints, zero lines) goes through every stage without touching its neighbours."""
    import torch
    B = 8
    frames = synth.make_frames(520, B, 480, 640)
    frames[1] = 77                                        # flat: no keypoints, no lines
    frames[4] = (np.arange(640, dtype=np.uint8) // 64 * 2 + 100)[None, :]   # faint ramp: below every threshold
    d = torch.from_numpy(frames).cuda()
    orb = plslam.ORBextractor(1000, 1.2, 8, 20, 7, rows=480, cols=640, max_batch=B)
    le = plslam.LINEextractor(1, 1.2, 200, 0.0, rows=480, cols=640, max_batch=B)
    ocap, lcap = orb.capacity, le.capacity
    s = torch.cuda.current_stream().cuda_stream
    ref = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    for nb in (3, B):
        kps = torch.zeros((B, ocap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((B, ocap, 32), dtype=torch.uint8, device="cuda")
        n = torch.full((B,), -5, dtype=torch.int32, device="cuda")
        kl = torch.zeros((B, lcap, 17), dtype=torch.float32, device="cuda")
        ld = torch.zeros((B, lcap, 32), dtype=torch.uint8, device="cuda")
        fn = torch.zeros((B, lcap, 3), dtype=torch.float64, device="cuda")
        nl = torch.full((B,), -5, dtype=torch.int32, device="cuda")
        orb.extract_batch_dev(d, nb, 480 * 640, kps, desc, n, s)
        le.extract_batch_dev(d, nb, 480 * 640, kl, ld, fn, nl, s)
        torch.cuda.synchronize()
        n_h, nl_h = n.cpu().numpy(), nl.cpu().numpy()
        assert (n_h[nb:] == -5).all() and (nl_h[nb:] == -5).all()          # frames beyond the call's batch are untouched
        k_h = kps.cpu().numpy().view(np.uint8).reshape(B, ocap, 28).copy().view(plslam.KP_DTYPE).reshape(B, ocap)
        d_h = desc.cpu().numpy()
        kl_h = kl.cpu().numpy().view(np.uint8).reshape(B, lcap, 68).copy().view(plslam.KL_DTYPE).reshape(B, lcap)
        ld_h, fn_h = ld.cpu().numpy(), fn.cpu().numpy()
        for b in range(nb):
            rk, rd = ref.extract(frames[b])
            assert n_h[b] == len(rk) and (d_h[b, :n_h[b]] == rd).all()
            assert all((k_h[b, :n_h[b]][f] == rk[f]).all() for f in rk.dtype.names)
            lk, ldr, lfr = oracle.line_extract(frames[b], 200, 0.0)
            assert nl_h[b] == len(lk)
            _match(kl_h[b, :nl_h[b]], ld_h[b, :nl_h[b]], fn_h[b, :nl_h[b]], lk, ldr, lfr, "frame %d" % b)
        assert n_h[1] == 0 and nl_h[1] == 0 and (nb <= 4 or (n_h[4] == 0 and nl_h[4] == 0))
    orb.close()
    le.close()


def test_full_residency_replicas_are_identical(plslam, oracle, synth):
    """BASELINE's bench size (6144 resident frames, 4 pipelined sub-batches, consecutive steps not joined) through a
    size-independent property: the batch is 384 copies of 16 distinct frames, so every copy must come out bit-identical to the
    first one -- for every record the front end produces -- whatever wavefront slot, sub-batch or memory block it ran in; and the
    first copies equal the oracle.  Exercises the batch build of k_lsd_grow (64 registers, 8 wavefronts per SIMD) at full
    residency and the per-frame arena at its largest."""
    import torch
    V = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    PL = _util._load("plslam_amd_pipeline", os.path.join(_util.ROOT, "pl-slam_amd", "pipeline.py"))
    B, ns, U = 6144, 4, 16
    base = synth.make_frames(530, U, 480, 640)
    d = torch.from_numpy(base).cuda().repeat(B // U, 1, 1).contiguous()
    voc = V.Vocabulary.synthetic(102, k=10, L=6, synth=synth)
    fp = PL.FrontEndPipelined(plslam, voc, B, 480, 640, 1000, 8, 200, 0.0, TUM1_K, TUM1_D, nsplit=ns)
    for _ in range(2):
        fp.step(d, join=False)
    r = fp.results()
    fp.close()
    n, nl = r["n"], r["nl"]
    assert (n.reshape(-1, U) == n[:U]).all() and (nl.reshape(-1, U) == nl[:U]).all()
    assert (r["nm_orb"].reshape(-1, U) == r["nm_orb"][:U]).all() and (r["nm_line"].reshape(-1, U) == r["nm_line"][:U]).all()
    for key, cnt in (("desc", n), ("nid", n), ("word", n), ("ldesc", nl), ("lfn", nl), ("m_line", nl)):
        a = r[key].reshape((B // U, U) + r[key].shape[1:])
        for u in range(U):
            assert np.array_equal(a[:, u, :cnt[u]], np.broadcast_to(a[0, u, :cnt[u]], a[:, u, :cnt[u]].shape), equal_nan=True), (key, u)
    for key, cnt in (("kps", n), ("kl", nl)):
        a = r[key].reshape(B // U, U, -1)
        for u in range(U):
            for f in a.dtype.names:
                assert (a[:, u, :cnt[u]][f] == a[0, u, :cnt[u]][f]).all(), (key, f, u)
    ref = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    for u in (0, 7):                                       # ... and the first copies are the oracle's
        rk, rd = ref.extract(base[u])
        assert n[u] == len(rk) and (r["desc"][u, :n[u]] == rd).all()
        lk, ldr, lfr, _ = _oracle_line(oracle, base[u], 200, 0.0, TUM1_K, TUM1_D)
        assert nl[u] == len(lk) and (r["ldesc"][u, :nl[u]] == ldr).all()



def test_ragged_group_of_eight_xcd_block_order(plslam, oracle, synth):
    """From 64 frames on the kernels that share bytes between a frame's blocks decode a one-dimensional grid the XCD-aware way
    (plh_xcd_decode, plh_common.h: block L -> frame 8 (L / 8 / blocks per frame) + L % 8).  A handle planned for 80 frames is driven
    with 67 -- the last group of eight holds three frames and five that do not exist -- and then with 64 (the smallest batch that
    takes the order): every frame equals the oracle, frames beyond the call's batch are untouched."""
    import torch
    B = 80
    frames = synth.make_frames(640, B, 480, 640)
    d = torch.from_numpy(frames).cuda()
    orb = plslam.ORBextractor(1000, 1.2, 8, 20, 7, rows=480, cols=640, max_batch=B)
    le = plslam.LINEextractor(1, 1.2, 200, 0.0, rows=480, cols=640, max_batch=B, K=TUM1_K, D=TUM1_D)
    ocap, lcap = orb.capacity, le.capacity
    s = torch.cuda.current_stream().cuda_stream
    ref = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    want = {}
    for nb in (67, 64):
        kps = torch.zeros((B, ocap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((B, ocap, 32), dtype=torch.uint8, device="cuda")
        n = torch.full((B,), -5, dtype=torch.int32, device="cuda")
        kl = torch.zeros((B, lcap, 17), dtype=torch.float32, device="cuda")
        ld = torch.zeros((B, lcap, 32), dtype=torch.uint8, device="cuda")
        fn = torch.zeros((B, lcap, 3), dtype=torch.float64, device="cuda")
        nl = torch.full((B,), -5, dtype=torch.int32, device="cuda")
        orb.extract_batch_dev(d, nb, 480 * 640, kps, desc, n, s)
        le.extract_batch_dev(d, nb, 480 * 640, kl, ld, fn, nl, s)
        torch.cuda.synchronize()
        assert orb.status() == 0 and le.status() == 0
        n_h, nl_h = n.cpu().numpy(), nl.cpu().numpy()
        assert (n_h[nb:] == -5).all() and (nl_h[nb:] == -5).all()
        k_h = kps.cpu().numpy().view(np.uint8).reshape(B, ocap, 28).copy().view(plslam.KP_DTYPE).reshape(B, ocap)
        d_h = desc.cpu().numpy()
        kl_h = kl.cpu().numpy().view(np.uint8).reshape(B, lcap, 68).copy().view(plslam.KL_DTYPE).reshape(B, lcap)
        ld_h, fn_h = ld.cpu().numpy(), fn.cpu().numpy()
        for b in range(nb):
            if b not in want:
                want[b] = (ref.extract(frames[b]), _oracle_line(oracle, frames[b], 200, 0.0, TUM1_K, TUM1_D)[:3])
            (rk, rd), (lk, ldr, lfr) = want[b]
            assert n_h[b] == len(rk) and (d_h[b, :n_h[b]] == rd).all(), (nb, b)
            assert all((k_h[b, :n_h[b]][f] == rk[f]).all() for f in rk.dtype.names), (nb, b)
            assert nl_h[b] == len(lk), (nb, b)
            _match(kl_h[b, :nl_h[b]], ld_h[b, :nl_h[b]], fn_h[b, :nl_h[b]], lk, ldr, lfr, "batch %d frame %d" % (nb, b))
    orb.close()
    le.close()
