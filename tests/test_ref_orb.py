"""Pinning the ORB path against the REFERENCE's own code.  oracle/_ref/liborb_ref.so is the reference's
src/ORBextractor.cc + include/ORBextractor.h compiled from the sources where they lie (oracle/ref/build_ref.sh): OpenCV's
types come from a stand-in header, cv::resize / cv::GaussianBlur / cv::FAST / cv::fastAtan2 are the oracle's restatements,
and the library's allocations come from a monotonic arena so that DistributeOctTree's address-ordered tie-break
(ORBextractor.cc:684, run-to-run dependent under malloc) is the "most recently created first" order the oracle pins.

What this pins: the constructor tables, the per-cell FAST + threshold fallback, the quad-tree distribution, IC_Angle, the
steered rBRIEF, level ordering and coordinate scaling -- the reference's own control logic, bit for bit.  What it does not
pin: the OpenCV primitives underneath (restated in oracle/img_ops.cc).

tools/gen_golden_ref.py committed the reference outputs as tests/golden/ref_orb_*.npz:
  * the oracle must reproduce them (CPU), * so must the GPU extractor (`-m gpu`; no reference on the GPU box),
  * and when the reference library is present (build container) it is run live on more images and parameters."""
import glob
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLDEN = sorted(glob.glob(os.path.join(_util.ROOT, "tests", "golden", "ref_orb_*.npz")))
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "liborb_ref.so")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _same(kps, desc, rk, rd, what):
    assert len(kps) == len(rk), "%s: %d keypoints, reference %d" % (what, len(kps), len(rk))
    for f in rk.dtype.names:
        assert (kps[f] == rk[f]).all(), "%s: keypoint field %s differs from the reference" % (what, f)
    assert (desc == rd).all(), "%s: descriptors differ from the reference" % what


def _params(g):
    return int(g["nfeatures"]), float(g["scale"]), int(g["nlevels"]), int(g["ini"]), int(g["mn"])


def test_golden_files_present():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[8:-4] for p in GOLDEN])
def test_oracle_reproduces_reference_orb(oracle, synth, path):
    g = np.load(path)
    img = synth.make_frame(int(g["seed"]), int(g["rows"]), int(g["cols"]))
    ref = oracle.OrbOracle(*_params(g))
    kps, desc = ref.extract(img)
    _same(kps, desc, g["kps"], g["desc"], "oracle")
    sf = ref.scale_table(0) if hasattr(ref, "scale_table") else None
    if sf is not None:
        assert (np.asarray(sf, np.float32)[: len(g["scale_factors"])] == g["scale_factors"]).all()


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_orb_live(oracle, plslam, synth):
    """More images and parameter sets through the reference's ORBextractor right now (build container only)."""
    G = _gen()
    R = G.ref_orb_lib()
    cases = [(2, 480, 640, 1000, 1.2, 8, 20, 7), (5, 376, 1241, 2000, 1.2, 8, 20, 7), (11, 200, 403, 700, 1.2, 5, 20, 7),
             (12, 240, 320, 400, 1.3, 6, 30, 10), (13, 96, 128, 150, 1.2, 3, 20, 7), (14, 480, 640, 2000, 1.1, 8, 15, 5)]
    for seed, rows, cols, nf, scale, nl, ini, mn in cases:
        img = synth.make_frame(seed, rows, cols)
        rk, rd, sf, sg = G.reference_orb(R, plslam, img, nf, scale, nl, ini, mn)
        kps, desc = oracle.OrbOracle(nf, scale, nl, ini, mn).extract(img)
        _same(kps, desc, rk, rd, "oracle vs live reference (seed %d)" % seed)
    # a flat image: no corners anywhere -> both return nothing
    flat = np.full((120, 160), 90, np.uint8)
    rk, rd, _, _ = G.reference_orb(R, plslam, flat, 300, 1.2, 4, 20, 7)
    kps, desc = oracle.OrbOracle(300, 1.2, 4, 20, 7).extract(flat)
    assert len(rk) == 0 and len(kps) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[8:-4] for p in GOLDEN])
def test_gpu_reproduces_reference_orb(plslam, synth, path):
    g = np.load(path)
    rows, cols = int(g["rows"]), int(g["cols"])
    img = synth.make_frame(int(g["seed"]), rows, cols)
    nf, scale, nl, ini, mn = _params(g)
    ex = plslam.ORBextractor(nf, scale, nl, ini, mn, rows=rows, cols=cols, max_batch=1, device=0)
    try:
        kps, desc = ex(img)
    finally:
        ex.close()
    _same(kps, desc, g["kps"], g["desc"], "GPU")
