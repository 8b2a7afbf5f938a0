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
    # round 6: k_adv_improve is BUILT for three wavefronts per SIMD (168 of the 235 registers it wants, 248 bytes of spill): as fast
    # alone, + 1.3 - 2.0 % on the headline for the room it leaves the other sub-batches' wavefronts (profiles/r06_adv_improve_registers_ab.txt)
    bad = {k: (r["private_segment_fixed_size"], r["uses_dynamic_stack"]) for k, r in res.items()
           if k not in ("k_lsd_grow", "k_lsd_grow_mw16", "k_adv_improve") and (r["private_segment_fixed_size"] != 0 or r["uses_dynamic_stack"])}
    assert not bad, bad
    imp = res["k_adv_improve"]
    assert _waves(imp) >= 3 and imp["private_segment_fixed_size"] <= 256 and not imp["uses_dynamic_stack"], imp


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


# ---- the miscompilation behind round 5's "memory access fault" of the counter build (profiles/r06_prof_build_mw16_fault_root_cause.txt):
# vector writes the register allocator placed in front of a join block's EXEC restore.  tools/isa_exec_split_check.py finds the shape in
# the ISA; the excerpt of the faulting build is kept as a fixture so that the detector itself is tested.
import isa_exec_split_check as ISA   # noqa: E402


def _excerpt():
    ins = []
    for ln in open(os.path.join(_util.ROOT, "tests", "golden", "r05_counter_build_mw16_join_block.s")):
        m = ISA.INS.match(ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins


def test_detector_finds_the_round5_miscompilation():
    hits = ISA.scan(_excerpt())
    texts = sorted(t for _, t, _, _ in hits)
    # the constant 1 (region growing's private mark), two more constants and a 64-bit copy, re-materialised above `s_or_b64 exec, exec, s[0:1]`
    assert texts == ["v_mov_b32_e32 v46, 1", "v_mov_b32_e32 v51, 0x100", "v_mov_b32_e32 v84, 0xffffff80", "v_mov_b64_e32 v[68:69], v[36:37]"], texts


@pytest.mark.parametrize("lib", ["libplslam_hip.so", "libplslam_hip_prof.so"])
def test_no_vector_write_in_front_of_a_join_blocks_exec_restore(lib):
    import tempfile
    path = os.path.join(_util.ROOT, "pl-slam_amd", lib)
    if not os.path.exists(path):
        pytest.skip(lib + " not built")
    bad = []
    with tempfile.TemporaryDirectory() as td:
        for co in ISA.code_objects(path, td):
            for name, ins in ISA.functions(co):
                bad += [(name, "+0x%x" % (a - ins[0][0]), t) for a, t, _, _ in ISA.scan(ins)]
    assert not bad, bad
