#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (the reference ships no fixtures).

The oracle is a restatement, so these vectors pin the ORACLE (regression) and the GPU path against
it; they are not outputs of the reference binary (which cannot be built here: OpenCV absent).
    python tools/gen_golden.py [substring]      (only the cases whose name contains `substring`)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

O, S = _util.oracle(), _util.synth()
O.build()
G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""


def wanted(name):
    return ONLY in name


def orb_case(name, seed, rows, cols, nf, nl=8, ini=20, mn=7, **kw):
    if not wanted(name):
        return
    img = S.make_frame(seed, rows, cols, **kw)
    o = O.OrbOracle(nf, 1.2, nl, ini, mn)
    kps, desc = o.extract(img)
    crc = [int(np.uint32(np.bitwise_xor.reduce(o.level(l).astype(np.uint32).ravel() * np.uint32(2654435761) +
                                               np.arange(o.level(l).size, dtype=np.uint32)))) for l in range(nl)]
    ncand = [len(o.candidates(l)) for l in range(nl)]
    np.savez_compressed(os.path.join(G, name + ".npz"), seed=seed, rows=rows, cols=cols, nfeatures=nf, nlevels=nl,
                        ini=ini, mn=mn, n_rect=kw.get("n_rect", 400), n_line=kw.get("n_line", 200),
                        img_sum=int(img.astype(np.int64).sum()), kps=kps, desc=desc, level_crc=np.array(crc, np.uint32),
                        ncand=np.array(ncand))
    print(name, len(kps), ncand)


orb_case("orb_s1_640x480", 1, 480, 640, 1000)
orb_case("orb_kitti_1241x376", 1000, 376, 1241, 2000)
orb_case("orb_small_160x120", 7, 120, 160, 200, nl=3, n_rect=40, n_line=20)


# ---- line extractor (LSD + KeyLine selection + LBD): segments, keylines, descriptors, line equations
# refine: cv::LineSegmentDetector's level (0 = LSD_REFINE_STD, 1 = LSD_REFINE_ADV = the reference's, oracle/plo.py REFERENCE_REFINE);
# the files of rounds 1-4 carry no `refine` field and are LSD_REFINE_STD
def line_case(name, seed, rows, cols, nf, minlen=0.0, refine=0, **kw):
    if not wanted(name):
        return
    img = S.make_frame(seed, rows, cols, **kw)
    segs = O.lsd_detect(img, refine=refine)
    kl, desc, fn = O.line_extract(img, nf, minlen, refine=refine)
    extra = dict(refine=refine) if refine else {}
    np.savez_compressed(os.path.join(G, name + ".npz"), seed=seed, rows=rows, cols=cols, nfeature=nf, minlen=minlen,
                        n_rect=kw.get("n_rect", 400), n_line=kw.get("n_line", 200), img_sum=int(img.astype(np.int64).sum()),
                        segs=segs, keylines=kl, desc=desc, linefn=fn, **extra)
    print(name, len(segs), len(kl))


line_case("line_s1_640x480", 1, 480, 640, 200)
line_case("line_small_160x120", 7, 120, 160, 50, n_rect=40, n_line=20)
line_case("line_adv_s1_640x480", 1, 480, 640, 200, refine=1)
line_case("line_adv_kitti_1241x376", 1000, 376, 1241, 200, refine=1)
line_case("line_adv_small_160x120", 7, 120, 160, 50, refine=1, n_rect=40, n_line=20)


# ---- Hamming matchers on the S3 descriptor sets: knn2 table checksum, SearchDouble, SearchByBoW
def match_case(name, seed, n):
    if not wanted(name):
        return
    import ctypes as C
    a, b, perm = S.make_descriptor_sets(seed, n, 0.08)
    idx, dist = O.knn2(a, b)
    m = np.zeros(n, np.int32)
    L = O.lib()
    c = L.plo_line_search_double(O._p(a), n, O._p(b), n, C.c_float(50.0), C.c_float(0.7), O._p(m))
    rng = S.SplitMix64(seed + 7)
    node_a = rng.randint(n, 0, 100).astype(np.int32)
    node_b = node_a[perm].copy()
    ang_a = rng.uniform(n, 0, 360).astype(np.float32)
    ang_b = ((ang_a[perm] + 12.0) % 360).astype(np.float32)
    valid = np.ones(n, np.uint8)
    mb = np.zeros(n, np.int32)
    cb = L.plo_orb_search_by_bow(O._p(a), O._p(ang_a), O._p(node_a), O._p(valid), n, O._p(b), O._p(ang_b), O._p(node_b), n, 50,
                                 C.c_float(0.7), 1, O._p(mb))
    np.savez_compressed(os.path.join(G, name + ".npz"), seed=seed, n=n, knn_idx=idx, knn_dist=dist, double_n=c, double_m=m,
                        bow_n=cb, bow_m=mb, node_a=node_a, node_b=node_b, ang_a=ang_a, ang_b=ang_b)
    print(name, c, cb)


match_case("match_s3_2000", 100, 2000)
match_case("match_s3_200", 103, 200)
