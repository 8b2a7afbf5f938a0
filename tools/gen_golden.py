#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (the reference ships no fixtures).

The oracle is a restatement, so these vectors pin the ORACLE (regression) and the GPU path against
it; they are not outputs of the reference binary (which cannot be built here: OpenCV absent).
    python tools/gen_golden.py
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


def orb_case(name, seed, rows, cols, nf, nl=8, ini=20, mn=7, **kw):
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
