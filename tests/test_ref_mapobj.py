"""Pinning ComputeDistinctiveDescriptors against the REFERENCE's own code.  oracle/_ref/libmapobj_ref.so holds the
reference's include/MapPoint.h + src/MapPoint.cc and include/MapLine.h + src/MapLine.cpp compiled as they are
(oracle/ref/build_ref.sh) against stand-ins for KeyFrame / Frame / Map only (oracle/ref/mapobj_stub.h); the harness
(oracle/ref/ref_mapobj.cc) adds observations to a real MapPoint / MapLine and calls
MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:249-314) / MapLine::ComputeDistinctiveDescriptors
(src/MapLine.cpp:256-326).  Pinned: which observation wins -- the N x N Hamming table, the sorted row's element
0.5*(N-1), first minimum -- and that bad keyframes' rows are left out.  (std::map<KeyFrame*, size_t> iterates by keyframe
address; the harness allocates its keyframes in one array, i.e. in the order the rows are given.)

tools/gen_golden_ref.py committed the chosen descriptors as tests/golden/ref_distinctive.npz; the oracle, the HIP sources
on the host emulator and the GPU kernel (`-m gpu`) must reproduce them; in the build container the reference also runs live."""
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLDEN = os.path.join(_util.ROOT, "tests", "golden", "ref_distinctive.npz")
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "libmapobj_ref.so")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _kept(rows, bad):
    return np.ascontiguousarray(rows[bad == 0])


def _check(choose, cases, g):
    for i, (rows, bad) in enumerate(cases):
        keep = _kept(rows, bad)
        for line in (0, 1):
            rc, want = int(g["d_%d_%d_rc" % (i, line)]), g["d_%d_%d" % (i, line)]
            assert rc == (1 if len(keep) else 0)
            if rc:
                assert (keep[choose(i, keep)] == want).all(), "case %d (%d rows) %s" % (i, len(keep), "line" if line else "point")


def test_golden_file_present():
    assert os.path.exists(GOLDEN)


def test_oracle_reproduces_reference_distinctive(oracle, synth):
    import ctypes as C
    G = _gen()
    L = oracle.lib()
    L.plo_distinctive_descriptor.argtypes = [C.c_void_p, C.c_int]
    L.plo_distinctive_descriptor.restype = C.c_int
    _check(lambda i, keep: L.plo_distinctive_descriptor(oracle._p(keep), len(keep)), G.distinctive_inputs(synth), np.load(GOLDEN))


def _device(P, synth, lib):
    G = _gen()
    cases = G.distinctive_inputs(synth)
    got = P.distinctive_descriptors([_kept(r, b) for r, b in cases], lib=lib)
    _check(lambda i, keep: got[i], cases, np.load(GOLDEN))


def test_emu_reproduces_reference_distinctive(plslam, synth, emu_lib):
    _device(plslam, synth, emu_lib)


@pytest.mark.gpu
def test_gpu_reproduces_reference_distinctive(plslam, synth):
    _device(plslam, synth, None)


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_distinctive_live(oracle, synth):
    import ctypes as C
    G = _gen()
    R, L = G.ref_mapobj_lib(), oracle.lib()
    L.plo_distinctive_descriptor.argtypes = [C.c_void_p, C.c_int]
    L.plo_distinctive_descriptor.restype = C.c_int
    for rows, bad in G.distinctive_inputs(synth, seed=77, sizes=[1, 2, 5, 12, 31, 100], reps=4):
        keep = _kept(rows, bad)
        for line in (0, 1):
            rc, d = G.reference_distinctive(R, rows, bad, line)
            assert rc == (1 if len(keep) else 0)
            if rc:
                assert (d == keep[L.plo_distinctive_descriptor(oracle._p(keep), len(keep))]).all()
