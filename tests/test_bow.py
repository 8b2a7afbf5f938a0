"""DBoW2 vocabulary-tree transform (8f "next" row 1): oracle sanity, HIP source under hipemu, GPU parity.
The reference's ORBvoc.bin is not in the mount, so a synthetic k=10, L=6 tree is used (pl-slam_amd/vocab.py)."""
import importlib.util
import os

import numpy as np
import pytest

import _util


def _vocab_mod():
    return _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))


def _oracle_transform(O, desc, voc, levelsup=4):
    n = len(desc)
    nid = np.zeros(max(n, 1), np.int32)
    word = np.zeros(max(n, 1), np.int32)
    import ctypes as C
    f = O.lib().plo_bow_transform
    f.argtypes = [C.c_void_p] * 1 + [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    f.restype = None
    f(O._p(desc), n, O._p(voc.node_desc), O._p(voc.child_start), O._p(voc.child_count), O._p(voc.word_id), O._p(voc.weight),
      voc.L, levelsup, O._p(nid), O._p(word))
    return nid[:n], word[:n]


def test_oracle_transform_descends_to_nearest_leaf(oracle, synth):
    voc = _vocab_mod().Vocabulary.synthetic(5, k=4, L=3, synth=synth)
    first_leaf = (4 ** 3 - 1) // 3
    leaves = voc.node_desc[first_leaf:first_leaf + 64]
    nid, word = _oracle_transform(oracle, leaves, voc, levelsup=2)
    # a leaf's own descriptor descends to a leaf at distance 0 whose level-1 ancestor is its own ancestor
    # (ties between identical siblings are impossible here with overwhelming probability)
    assert (word == np.arange(64)).mean() > 0.9
    assert ((nid >= 1) & (nid <= 4)).all()          # level L - levelsup = 1


def test_emu_bow_transform(plslam, oracle, synth, emu_lib):
    voc = _vocab_mod().Vocabulary.synthetic(102, k=10, L=3, synth=synth, stop_fraction=0.05)
    a, b, _ = synth.make_descriptor_sets(7, 300)
    nid, word = plslam.bow_transform([a, b[:123]], voc, levelsup=1, lib=emu_lib)
    for p, d in enumerate([a, b[:123]]):
        rn, rw = _oracle_transform(oracle, d, voc, levelsup=1)
        assert (nid[p, :len(d)] == rn).all() and (word[p, :len(d)] == rw).all()
        assert (nid[p, len(d):] == -1).all()
    assert (word == -1).sum() > 0                    # some stopped words


@pytest.mark.gpu
def test_gpu_bow_transform_orbvoc_shape(plslam, oracle, synth):
    voc = _vocab_mod().Vocabulary.synthetic(102, k=10, L=6, synth=synth, stop_fraction=0.01)
    a, b, _ = synth.make_descriptor_sets(100, 2000)
    nid, word = plslam.bow_transform([a, b], voc, levelsup=4)
    for p, d in enumerate([a, b]):
        rn, rw = _oracle_transform(oracle, d, voc, levelsup=4)
        assert (nid[p] == rn).all() and (word[p] == rw).all()
    assert len(np.unique(nid[0])) > 50               # level-2 nodes: up to 100 distinct
