"""Occupancy points DESIGN.md relies on, read from the code objects inside the built product library (no GPU, nothing loaded):
registers -> wavefronts per SIMD, LDS per block, scratch.  A compiler or source change that silently costs a wave slot or starts
spilling shows up here, not three rounds later in a bench line (tools/kernel_resources.py)."""
import os
import sys

import pytest

import _util

sys.path.insert(0, os.path.join(_util.ROOT, "tools"))
import kernel_resources as KR   # noqa: E402

LIB = os.path.join(_util.ROOT, "pl-slam_amd", "libplslam_hip.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(KR.LLVM, "clang-offload-bundler"))),
                                reason="needs the built library and the ROCm llvm tools")


@pytest.fixture(scope="module")
def res():
    r = KR.kernel_resources(LIB)
    assert len(r) >= 45, sorted(r)          # every translation unit's bundle was found
    return r


def _waves(r):
    return KR.waves_per_simd(KR.unified_vgprs(r))


def test_region_growing_keeps_its_seven_wave_slots(res):
    g = res["k_lsd_grow"]
    assert _waves(g) >= 7, g                 # 72 registers: 7168 resident frames (+ 3.7 % against the 64-register build)
    # round 5: nothing on the per-seed path lives in scratch any more; what is left are two stores per kept region and reloads on
    # refine() / reduce_region_radius() paths (112 bytes before)
    assert g["private_segment_fixed_size"] <= 64 and not g["uses_dynamic_stack"], g
    mw16 = res["k_lsd_grow_mw16"]
    assert _waves(mw16) >= 4 and mw16["private_segment_fixed_size"] <= 144, mw16   # the 128-register build: two 8-wavefront frames per CU
    assert _waves(res["k_lsd_grow_mw"]) >= 3, res["k_lsd_grow_mw"]                # the roomy build: three wavefronts per SIMD


def test_no_other_kernel_spills(res):
    bad = {k: (r["private_segment_fixed_size"], r["uses_dynamic_stack"]) for k, r in res.items()
           if k not in ("k_lsd_grow", "k_lsd_grow_mw16") and (r["private_segment_fixed_size"] != 0 or r["uses_dynamic_stack"])}
    assert not bad, bad


def test_dense_kernels_run_eight_wavefronts_per_simd(res):
    for k in ("k_fast_strips", "k_orient_brief", "k_octree", "k_pyr_down", "k_lbd", "k_lsd_grad", "k_sobel_pack", "k_blur7_u8", "k_remap_u8",
              "k_resize_u8", "k_search_by_bow", "k_bow_transform", "k_lsd_bin_scatter"):
        assert _waves(res[k]) == 8, (k, res[k])


def test_waiting_kernels_of_the_line_chain_hold_little_lds(res):
    # a resident one-wavefront block that waits for memory holds its LDS tile, and LDS is what the two halves share on a
    # CU -- the launches are sized as a few fat blocks per frame AND the tiles are small
    assert res["k_lsd_rects"]["group_segment_fixed_size"] <= 8 * 1024 and res["k_lsd_rects_adv"]["group_segment_fixed_size"] <= 8 * 1024
    assert res["k_adv_first"]["group_segment_fixed_size"] <= 4 * 1024 and res["k_adv_improve"]["group_segment_fixed_size"] <= 2 * 1024
    assert res["k_keylines"]["group_segment_fixed_size"] <= 10 * 1024
