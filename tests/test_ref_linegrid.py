"""Frame::AssignFeaturesToGridForLine (reference src/Frame.cc:295-320) pinned on the reference's own LineIterator
(src/lineIterator.cpp compiled as it is into oracle/_ref/libmisc_ref.so): tools/gen_golden_ref.py re-enacts the ten-line
loop of Frame.cc around the real iterator and commits the cell lists (tests/golden/ref_linegrid_*.npz).  The oracle (CPU)
and the GPU grid builder (`-m gpu`) must reproduce them, cell by cell, in insertion order."""
import ctypes as C
import glob
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLDEN = sorted(glob.glob(os.path.join(_util.ROOT, "tests", "golden", "ref_linegrid_*.npz")))
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "libmisc_ref.so")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _oracle_grid(O, kl, g):
    L = O.lib()
    L.plo_frame_assign_grid_lines.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.plo_frame_assign_grid_lines.restype = C.c_int
    cs, ci = np.zeros(64 * 48 + 1, np.int32), np.zeros(max(len(kl), 1) * 64, np.int32)
    n = L.plo_frame_assign_grid_lines(O._p(kl), len(kl), O._p(g), O._p(cs), O._p(ci), len(ci))
    assert n <= len(ci)
    return cs, ci[:n]


def test_golden_files_present():
    assert len(GOLDEN) >= 2


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[13:-4] for p in GOLDEN])
def test_oracle_reproduces_reference_line_grid(oracle, plslam, synth, path):
    g = np.load(path)
    kl = _gen().random_keylines(synth, plslam, int(g["seed"]), int(g["n"]))
    cs, ci = _oracle_grid(oracle, kl, np.ascontiguousarray(g["gp"], np.float32))
    assert (cs == g["start"]).all() and (ci == g["items"]).all()


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_line_iterator_live(oracle, plslam, synth):
    G = _gen()
    M = G.ref_misc_lib()
    for seed, n in ((51, 400), (52, 37), (53, 1)):
        kl = G.random_keylines(synth, plslam, seed, max(n, 40))[:n]
        gp = plslam._gp_array(plslam.grid_params(640, 480))
        rs, ri = G.reference_line_grid(M, kl, gp)
        cs, ci = _oracle_grid(oracle, kl, gp)
        assert (cs == rs).all() and (ci == ri).all(), seed


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[13:-4] for p in GOLDEN])
def test_gpu_reproduces_reference_line_grid(plslam, synth, path):
    g = np.load(path)
    kl = _gen().random_keylines(synth, plslam, int(g["seed"]), int(g["n"]))
    gpa = np.ascontiguousarray(g["gp"], np.float32)
    gp = plslam.GridParams(*[float(v) for v in gpa]) if hasattr(plslam, "GridParams") else None
    frame = dict(kps=np.zeros(0, plslam.KP_DTYPE), desc=np.zeros((0, 32), np.uint8), keylines=kl,
                 ldesc=np.zeros((len(kl), 32), np.uint8), linefn=np.zeros((len(kl), 3)))
    scale = np.cumprod(np.r_[np.float32(1.0), np.full(7, np.float32(1.2))]).astype(np.float32)
    fs = plslam.FrameSearch(gp, scale, [frame])
    _, (lcs, lci) = fs.grids()
    assert (lcs[0] == g["start"]).all() and (lci[0, :g["start"][-1]] == g["items"]).all()
