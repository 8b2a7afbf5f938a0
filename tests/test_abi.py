"""C-ABI housekeeping: the build id the library carries, the capacity-flag query of the throughput path and the host
helper plh_descriptor_distance (ORBmatcher::DescriptorDistance, reference src/ORBmatcher.cc:1764-1780;
LSDmatcher::DescriptorDistance, src/LSDmatcher.cpp:654-670)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import _util

LIB = os.path.join(_util.ROOT, "pl-slam_amd", "libplslam_hip.so")


def _graft():
    sys.path.insert(0, _util.ROOT)
    import __graft_entry__ as g
    return g


def _version(path):
    lib = C.CDLL(path)
    lib.plh_version.restype = C.c_char_p
    return lib.plh_version().decode()


def test_library_is_built_from_these_sources():
    """The shipped libplslam_hip.so carries the hash of the sources next to it (a stale binary cannot pass)."""
    g = _graft()
    if not os.path.exists(LIB):
        g.build_hip()
    assert ("build " + g.source_id()) in _version(LIB), "libplslam_hip.so was not built from the current sources"


@pytest.mark.gpu
def test_gpu_library_is_built_from_these_sources(plslam):
    g = _graft()
    v = plslam.load().plh_version().decode()
    assert "gfx950" in v and ("build " + g.source_id()) in v, v


def _bitcount_ref(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def _distance_cases(S):
    rng = S.SplitMix64(2024)
    rows = rng.randint(64 * 32, 0, 256).astype(np.uint8).reshape(64, 32)
    cases = [(np.zeros(32, np.uint8), np.zeros(32, np.uint8), 0), (np.zeros(32, np.uint8), np.full(32, 255, np.uint8), 256)]
    one = np.zeros(32, np.uint8)
    one[31] = 0x80
    cases.append((np.zeros(32, np.uint8), one, 1))
    for i in range(0, 64, 2):
        cases.append((rows[i], rows[i + 1], _bitcount_ref(rows[i], rows[i + 1])))
    return cases


def test_descriptor_distance_product_host_helper(plslam, oracle, synth):
    """plh_descriptor_distance of the product library (a host popcount, no GPU involved) against a bit count and the
    oracle's restatement of the reference's SWAR loop."""
    if not os.path.exists(LIB):
        _graft().build_hip()
    L = plslam.load(LIB)
    O = oracle.lib()
    has_oracle = hasattr(O, "plo_descriptor_distance")
    if has_oracle:
        O.plo_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        O.plo_descriptor_distance.restype = C.c_int
    for a, b, want in _distance_cases(synth):
        assert plslam.descriptor_distance(a, b, LIB) == want
        assert plslam.descriptor_distance(b, a, LIB) == want
        if has_oracle:
            assert O.plo_descriptor_distance(oracle._p(np.ascontiguousarray(a)), oracle._p(np.ascontiguousarray(b))) == want
    # unaligned rows (a cv::Mat row pointer has no 8-byte guarantee)
    buf = np.zeros(80, np.uint8)
    buf[1:33] = 0xF0
    assert L.plh_descriptor_distance(C.c_void_p(buf.ctypes.data + 1), C.c_void_p(buf.ctypes.data + 40)) == 128


def _status_case(P, S, lib):
    img = S.make_frame(3, 120, 160, n_rect=40, n_line=20)
    ex = P.ORBextractor(200, 1.2, 3, 20, 7, rows=120, cols=160, max_batch=1, lib=lib)
    ln = P.LINEextractor(1, 1.2, 50, 0.0, rows=120, cols=160, max_batch=1, lib=lib)
    try:
        k, d = ex(img)
        kl, ld, fn = ln(img)
        assert len(k) > 50 and len(kl) > 5
        assert ex.status() == 0 and ln.status() == 0
    finally:
        ex.close()
        ln.close()


def test_emu_status_query(plslam, synth, emu_lib):
    _status_case(plslam, synth, emu_lib)


@pytest.mark.gpu
def test_gpu_status_query(plslam, synth):
    _status_case(plslam, synth, None)


def _selftest(P, lib):
    import ctypes as C
    L = P.load(lib)
    n = L.plh_selftest_shims()
    per = (C.c_int32 * n)()
    bad = C.c_int(-1)
    assert L.plh_selftest(0, C.byref(bad), per, n) == 0, L.plh_last_error()
    return bad.value, list(per)


def test_emu_selftest_entry_point(plslam, emu_lib):
    bad, per = _selftest(plslam, emu_lib)      # the emulator build has only the portable twins: nothing can differ
    assert bad == 0 and len(per) == 21 and not any(per)


@pytest.mark.gpu
def test_gpu_instruction_shims_match_their_descriptions(plslam):
    """plh_selftest: every gfx950 instruction the kernels reach through plh_shims.h (wave votes incl. inverse_ballot with bits in
    both mask halves, readlane broadcasts from every lane, v_perm / v_alignbyte / v_dot4 / v_dot2 / packed 16-bit, v_fract,
    the division without v_div_scale, the hand-scheduled walk of LSD's region growing) against its portable twin, 4096 rounds
    x 64 lanes each -- the class of bug the CPU emulator cannot see, caught in seconds."""
    bad, per = _selftest(plslam, None)
    assert bad == 0, "shims that differ (mismatching lanes per shim): %s" % per


def test_box_probe_and_line_reserve_need_a_device_or_fail_cleanly(plslam):
    """The two entry points of round 5 are exported and fail like every other one without a GPU (no CPU fallback)."""
    import torch
    L = plslam.load()
    ms = (C.c_float * 2)()
    assert L.plh_box_probe(0, 0, ms) == 1            # PLH_ERR_INVALID: iters <= 0
    assert L.plh_line_reserve(None, 1) == 1          # PLH_ERR_INVALID: no handle
    if not torch.cuda.is_available():
        assert L.plh_box_probe(0, 16, ms) == 2       # PLH_ERR_NO_DEVICE


@pytest.mark.gpu
def test_gpu_box_probe_and_line_reserve(plslam, oracle, synth):
    """plh_box_probe: a fixed VALU launch whose duration scales with its iteration count (the box normaliser of bench.py);
    plh_line_reserve: the workspace of a batch at create time -- the extraction that follows is the oracle's."""
    L = plslam.load()
    ms1, ms2 = (C.c_float * 2)(), (C.c_float * 2)()
    assert L.plh_box_probe(0, 1024, ms1) == 0 and L.plh_box_probe(0, 4096, ms2) == 0
    assert ms1[1] > 0 and 3.0 < ms2[1] / ms1[1] < 5.0, (ms1[1], ms2[1])
    img = synth.make_frame(21, 240, 320, n_rect=120, n_line=60)
    ex = plslam.LINEextractor(1, 1.2, 100, 0.0, rows=240, cols=320, max_batch=4)
    assert L.plh_line_reserve(ex.h, 5) == 1          # beyond the plan
    plslam._check(L, L.plh_line_reserve(ex.h, 4), "plh_line_reserve")
    kl, desc, fn = ex(img)
    ex.close()
    rk, rd, rf = oracle.line_extract(img, 100, 0.0)
    assert len(kl) == len(rk) and all((kl[f] == rk[f]).all() for f in rk.dtype.names) and (desc == rd).all() and (fn == rf).all()
