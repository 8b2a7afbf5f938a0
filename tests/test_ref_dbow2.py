"""Pinning against the REFERENCE's own code.  oracle/_ref/libdbow2_ref.so is the reference's vendored DBoW2
(Thirdparty/DBoW2) compiled from the sources where they lie (oracle/ref/build_ref.sh, with a minimal cv::Mat stand-in);
tools/gen_golden_ref.py ran it on synthetic vocabularies and committed the outputs as tests/golden/ref_dbow2_*.npz.

  * the oracle's restatement of FORB::distance and TemplatedVocabulary::transform must reproduce those goldens (CPU),
  * so must the GPU's plh_bow_transform_batch_dev (`-m gpu`; the reference itself is not on the GPU box),
  * and when the reference library is present (build container) it is run live on fresh inputs."""
import ctypes as C
import glob
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLDEN = sorted(glob.glob(os.path.join(_util.ROOT, "tests", "golden", "ref_dbow2_*.npz")))
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "libdbow2_ref.so")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _case(synth, g):
    G = _gen()
    VM = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    return G.make_case(synth, VM, int(g["vseed"]), int(g["k"]), int(g["L"]), float(g["stop"]), int(g["dseed"]), int(g["n"]))


def _oracle_transform(O, desc, voc, levelsup):
    n = len(desc)
    nid, word = np.zeros(n, np.int32), np.zeros(n, np.int32)
    f = O.lib().plo_bow_transform
    f.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    f.restype = None
    f(O._p(desc), n, O._p(voc.node_desc), O._p(voc.child_start), O._p(voc.child_count), O._p(voc.word_id), O._p(voc.weight),
      voc.L, levelsup, O._p(nid), O._p(word))
    return nid, word


def _check_against_reference(nid, word, g):
    """nid / word per feature in this repo's convention (-1 for a stopped word) vs the reference's outputs."""
    live = g["feat_node"] >= 0                      # FeatureVector membership == word not stopped (w > 0)
    assert (g["weight"][live] > 0).all() and (g["weight"][~live] <= 0).all()
    assert (nid[live] == g["feat_node"][live]).all() and (nid[live] == g["node"][live]).all()
    assert (word[live] == g["word"][live]).all()
    assert (word[~live] == -1).all()
    # BowVector of Frame::ComputeBoW: the set of words of the live features (values are normalised tf weights)
    assert sorted(set(word[live].tolist())) == g["bow_word"].tolist()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[10:-4] for p in GOLDEN])
def test_oracle_reproduces_reference_dbow2(oracle, synth, path):
    g = np.load(path)
    voc, desc = _case(synth, g)
    nid, word = _oracle_transform(oracle, desc, voc, int(g["levelsup"]))
    _check_against_reference(nid, word, g)
    dd = oracle.lib().plo_descriptor_distance
    dd.argtypes = [C.c_void_p, C.c_void_p]
    got = np.array([dd(oracle._p(desc[i]), oracle._p(desc[j])) for i, j in zip(g["pair_a"], g["pair_b"])], np.int32)
    assert (got == g["pair_dist"]).all()


def test_golden_files_present():
    assert len(GOLDEN) >= 3


def test_vocabulary_text_round_trip(synth, tmp_path):
    VM = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    voc = VM.Vocabulary.synthetic(3, k=5, L=3, synth=synth, stop_fraction=0.1)
    p = str(tmp_path / "voc.txt")
    voc.save_text(p)
    back = VM.Vocabulary.load_text(p)
    for f in ("node_desc", "child_start", "child_count", "word_id", "weight"):
        assert (getattr(voc, f)[1:] == getattr(back, f)[1:]).all(), f
    assert (back.k, back.L) == (5, 3)


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_library_live(oracle, synth):
    """Fresh inputs through the reference's DBoW2 right now (build container only)."""
    G = _gen()
    VM = _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))
    L = G.ref_lib()
    for vseed, k, Lv, stop, dseed, n, up in [(21, 9, 3, 0.03, 31, 700, 1), (22, 10, 4, 0.0, 32, 400, 3), (23, 3, 5, 0.0, 33, 300, 2)]:
        voc, desc = G.make_case(synth, VM, vseed, k, Lv, stop, dseed, n)
        word, weight, node, fnode, bw, bv = G.reference_transform(L, voc, desc, up)
        nid, w = _oracle_transform(oracle, desc, voc, up)
        _check_against_reference(nid, w, dict(feat_node=fnode, weight=weight, node=node, word=word, bow_word=bw))
        assert abs(bv.sum() - 1.0) < 1e-9            # L1-normalised tf weights (ScoringObject L1)
    dd = oracle.lib().plo_descriptor_distance
    dd.argtypes = [C.c_void_p, C.c_void_p]
    a, b, _ = synth.make_descriptor_sets(77, 500)
    for i in range(500):
        assert dd(oracle._p(a[i]), oracle._p(b[i])) == L.ref_forb_distance(G.p(a[i]), G.p(b[i]))


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[10:-4] for p in GOLDEN])
def test_gpu_reproduces_reference_dbow2(plslam, synth, path):
    g = np.load(path)
    voc, desc = _case(synth, g)
    nid, word = plslam.bow_transform([desc], voc, levelsup=int(g["levelsup"]))
    _check_against_reference(nid[0], word[0], g)
