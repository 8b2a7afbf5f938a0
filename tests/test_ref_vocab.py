"""SURVEY 8f row 1 completed: the DBoW2 vocabulary loaders in the C library and the BowVector of Frame::ComputeBoW, pinned
against the REFERENCE's own DBoW2 (Thirdparty/DBoW2 compiled from where it lies, oracle/ref/ref_dbow2.cc).

tools/gen_golden_ref.py::gen_vocab_files took synthetic trees (idf-like weights with full double mantissas, stopped words),
handed them to the reference as text (file A, tests/golden/ref_voc_*_src.txt), let the reference write them back with ITS
saveToTextFile (file B, *_ref.txt: 6-digit weights, trailing newline) and ITS saveToBinaryFile (file C, *_ref.bin: float
weights), loaded each file with the reference's loader again and recorded, per file, the per-feature word / node and the
BowVector (word ids in std::map order, double values).  Here
  * the oracle's parsers + transform + BowVector restatement reproduce those outputs from the same files (CPU),
  * plh_vocab_load_text / plh_vocab_load_binary + plh_vocab_transform_batch_dev do (emulator on CPU, GPU with -m gpu):
    words, FeatureVector nodes and BowVector values BIT-EXACT (doubles compared with ==),
  * plh_vocab_save_binary writes file C back byte for byte,
  * live, in the build container: the reference loads what plh_vocab_save_binary wrote.
Cases cover TF_IDF/L1 (ORBvoc's setting), TF/L2, BINARY/DOT_PRODUCT and a depth-first numbered file whose children are not
contiguous (the library renumbers internally and reports the reference's NodeIds)."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import _util

GOLD = os.path.join(_util.ROOT, "tests", "golden")
REF_SO = os.path.join(_util.ROOT, "oracle", "_ref", "libdbow2_ref.so")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(_util.ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


G = _gen()
CASES = G.VOC_CASES
IDS = [c[0] for c in CASES]


def _vm():
    return _util._load("plslam_amd_vocab", os.path.join(_util.ROOT, "pl-slam_amd", "vocab.py"))


def _files(name):
    return {"A": os.path.join(GOLD, "ref_voc_%s_src.txt" % name), "B": os.path.join(GOLD, "ref_voc_%s_ref.txt" % name),
            "C": os.path.join(GOLD, "ref_voc_%s_ref.bin" % name)}


def _golden(g, name, which):
    return {k: g["%s_%s_%s" % (name, which, k)] for k in ("word", "weight", "node", "feat_node", "bow_word", "bow_value")}


def _check(nid, word, bow, ref, what):
    live = ref["feat_node"] >= 0
    assert (ref["weight"][live] > 0).all() and (ref["weight"][~live] <= 0).all()
    assert (word[live] == ref["word"][live]).all() and (word[~live] == -1).all(), what + ": word ids"
    assert (nid[live] == ref["feat_node"][live]).all() and (nid[~live] == -1).all(), what + ": FeatureVector nodes"
    bw, bv = bow
    assert len(bw) == len(ref["bow_word"]) and (bw == ref["bow_word"]).all(), what + ": BowVector words"
    assert (bv == ref["bow_value"]).all(), what + ": BowVector values differ (max |d| %.3g)" % np.abs(bv - ref["bow_value"]).max()


# ------------------------------------------------------------------ oracle (CPU)
def _oracle_parse(O, path):
    L = O.lib()
    for f in (L.plo_vocab_parse_text, L.plo_vocab_parse_bin):
        f.argtypes = [C.c_char_p] + [C.c_void_p] * 5 + [C.c_int]
        f.restype = C.c_int
    cap = 4096
    hdr, parent, leaf = np.zeros(4, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.uint8)
    desc, weight = np.zeros((cap, 32), np.uint8), np.zeros(cap, np.float64)
    f = L.plo_vocab_parse_bin if path.endswith(".bin") else L.plo_vocab_parse_text
    n = f(path.encode(), O._p(hdr), O._p(parent), O._p(leaf), O._p(desc), O._p(weight), cap)
    assert n > 1, path
    return hdr, parent[:n], leaf[:n], desc[:n], weight[:n]


def _flat_from_parents(VM, hdr, parent, leaf, desc, weight):
    """Breadth-first flat tree + map flat -> file node id (what plh_vocab builds for a non-contiguous file)."""
    n = len(parent)
    kids = [[] for _ in range(n)]
    for i in range(1, n):
        kids[parent[i]].append(i)
    order = [0]
    for r in order:
        order.extend(kids[r])
    flat_of = np.zeros(n, np.int64)
    flat_of[order] = np.arange(n)
    ref_word = np.full(n, -1, np.int32)
    ref_word[leaf > 0] = np.arange(int((leaf > 0).sum()), dtype=np.int32)
    cs = np.array([flat_of[kids[r][0]] if kids[r] else 0 for r in order], np.int32)
    cc = np.array([len(kids[r]) for r in order], np.int32)
    voc = VM.Vocabulary(desc[order], cs, cc, ref_word[order], weight[order], int(hdr[0]), int(hdr[1]))
    return voc, np.array(order, np.int32), ref_word


def _oracle_outputs(O, VM, path, desc, levelsup):
    hdr, parent, leaf, nd, weight = _oracle_parse(O, path)
    voc, node_id, ref_word = _flat_from_parents(VM, hdr, parent, leaf, nd, weight)
    L = O.lib()
    n = len(desc)
    nid, word = np.zeros(n, np.int32), np.zeros(n, np.int32)
    L.plo_bow_transform.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.plo_bow_transform.restype = None
    L.plo_bow_transform(O._p(desc), n, O._p(voc.node_desc), O._p(voc.child_start), O._p(voc.child_count), O._p(voc.word_id),
                        O._p(voc.weight), voc.L, levelsup, O._p(nid), O._p(word))
    nid = np.where(nid >= 0, node_id[np.maximum(nid, 0)], -1)
    ww = np.zeros(max(1, int((leaf > 0).sum())), np.float64)
    ww[ref_word[leaf > 0]] = weight[leaf > 0]
    L.plo_bow_vector.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.plo_bow_vector.restype = C.c_int
    bw, bv = np.zeros(n, np.int32), np.zeros(n, np.float64)
    m = L.plo_bow_vector(O._p(word), n, O._p(ww), int(hdr[3]), int(hdr[2]), O._p(bw), O._p(bv), n)
    return hdr, nid, word, (bw[:m], bv[:m])


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_reproduces_reference_vocabulary_files(oracle, synth, case):
    name, seed, k, Lv, stop, dfs, dseed, n, up, scoring, weighting = case
    g = np.load(os.path.join(GOLD, "ref_voc.npz"))
    _, desc = G.voc_case_inputs(synth, _vm(), seed, k, Lv, stop, dseed, n)
    for which, path in _files(name).items():
        hdr, nid, word, bow = _oracle_outputs(oracle, _vm(), path, desc, up)
        assert tuple(hdr) == (k, Lv, scoring, weighting)
        _check(nid, word, bow, _golden(g, name, which), "oracle, file %s of %s" % (which, name))


def test_reference_text_files_lose_weight_digits():
    """What makes B differ from A: saveToTextFile prints 6 significant digits (TemplatedVocabulary.h:1458)."""
    g = np.load(os.path.join(GOLD, "ref_voc.npz"))
    a, b = g["k5L3_A_weight"], g["k5L3_B_weight"]
    assert (a != b).any() and np.allclose(a, b, rtol=1e-5)
    assert (g["k5L3_C_weight"] == g["k5L3_A_weight"].astype(np.float32)).all()     # binary files hold floats


# ------------------------------------------------------------------ the C library (emulator on CPU, GPU)
def _product(P, S, lib, case):
    name, seed, k, Lv, stop, dfs, dseed, n, up, scoring, weighting = case
    g = np.load(os.path.join(GOLD, "ref_voc.npz"))
    _, desc = G.voc_case_inputs(S, _vm(), seed, k, Lv, stop, dseed, n)
    files = _files(name)
    for which, path in files.items():
        v = P.ORBVocabulary(lib=lib)
        assert (v.loadFromBinaryFile if which == "C" else v.loadFromTextFile)(path)
        assert (v.info.k, v.info.L, v.info.scoring, v.info.weighting) == (k, Lv, scoring, weighting)
        assert v.info.identity_ids == (0 if dfs else 1)
        assert v.size() == int(g["%s_C_size" % name]) - 1     # the reference's binary reader appends one phantom word
        nid, word, bow = v.transform([desc, desc[: n // 2]], levelsup=up)
        _check(nid[0, :n], word[0, :n], bow[0], _golden(g, name, which), "file %s of %s" % (which, name))
        assert (nid[1, n // 2:] == -1).all() and len(bow[1][0]) <= len(bow[0][0])
        if which == "C":                                      # saveToBinaryFile layout, byte for byte
            out = path + ".roundtrip"
            try:
                v.saveToBinaryFile(out)
                assert open(out, "rb").read() == open(path, "rb").read()
            finally:
                if os.path.exists(out):
                    os.remove(out)
        v.close()


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_emu_vocabulary_files_and_bow_vector(plslam, synth, emu_lib, case):
    _product(plslam, synth, emu_lib, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_gpu_vocabulary_files_and_bow_vector(plslam, synth, case):
    _product(plslam, synth, None, case)


def _create_and_errors(P, S, lib, tmp_path):
    VM = _vm()
    voc = VM.Vocabulary.synthetic(7, k=4, L=2, synth=S, idf=True)
    parent, leaf = voc.tree_arrays()
    v = P.ORBVocabulary(lib=lib)
    v.create(4, 2, parent, leaf, voc.node_desc, voc.weight64)
    a = v.arrays()
    assert (a["child_start"] == voc.child_start).all() and (a["child_count"] == voc.child_count).all()
    assert (a["word_id"] == voc.word_id).all() and (a["weight"][1:] == voc.weight64[1:]).all()
    assert (a["node_id"] == np.arange(voc.n_nodes)).all() and (a["node_desc"][1:] == voc.node_desc[1:]).all()
    v.close()
    bad = tmp_path / "bad.txt"
    bad.write_text("99 3 0 0\n0 1 " + " ".join(["0"] * 32) + " 1.0")
    with pytest.raises(P.PlhError):
        P.ORBVocabulary(lib=lib).loadFromTextFile(str(bad))           # k out of range: the reference rejects it too
    with pytest.raises(P.PlhError):
        P.ORBVocabulary(lib=lib).loadFromTextFile(str(tmp_path / "missing.txt"))
    node = "0 1 " + " ".join(["7"] * 32)
    short = tmp_path / "noweight.txt"                                 # a node line without its weight must not borrow the next line's
    short.write_text("4 2 0 0\n" + node + "\n" + node + " 0.5\n")   # first field as the weight (strtod would skip the line feed)
    with pytest.raises(P.PlhError, match="no weight"):
        P.ORBVocabulary(lib=lib).loadFromTextFile(str(short))
    digits = tmp_path / "digits.txt"                                  # a run of digits no field can hold is refused, not wrapped around
    digits.write_text("4 2 0 0\n0 1 " + "9" * 40 + " " + " ".join(["7"] * 31) + " 0.5\n")
    with pytest.raises(P.PlhError):
        P.ORBVocabulary(lib=lib).loadFromTextFile(str(digits))
    ok = tmp_path / "exp.txt"                                         # exponent and sign forms of `ostream << double` still load
    ok.write_text("4 2 0 0\n" + node + " 1e-05\n" + node + " +2.5\n")
    v = P.ORBVocabulary(lib=lib)
    v.loadFromTextFile(str(ok))
    assert list(v.arrays()["weight"][1:3]) == [1e-05, 2.5]
    v.close()
    with pytest.raises(P.PlhError):
        P.ORBVocabulary(lib=lib).loadFromBinaryFile(_files("k5L3")["A"])   # a text file is not a binary vocabulary


def test_emu_vocab_create_and_errors(plslam, synth, emu_lib, tmp_path):
    _create_and_errors(plslam, synth, emu_lib, tmp_path)


@pytest.mark.gpu
def test_gpu_vocab_create_and_errors(plslam, synth, tmp_path):
    _create_and_errors(plslam, synth, None, tmp_path)


@pytest.mark.gpu
def test_gpu_bow_vector_large_tree(plslam, oracle, synth):
    VM = _vm()
    voc = VM.Vocabulary.synthetic(102, k=10, L=5, synth=synth, stop_fraction=0.01, idf=True)
    parent, leaf = voc.tree_arrays()
    v = plslam.ORBVocabulary()
    v.create(10, 5, parent, leaf, voc.node_desc, voc.weight64)
    sets = []
    for i, n in enumerate((2000, 1, 0, 1337)):
        a, b, _ = synth.make_descriptor_sets(500 + i, max(n, 1))
        rng = synth.SplitMix64(600 + i)
        pick = rng.randint(max(n, 1), (10 ** 5 - 1) // 9, voc.n_nodes)
        sets.append(np.ascontiguousarray(np.where((np.arange(max(n, 1)) % 2 == 0)[:, None], a, voc.node_desc[pick])[:n], np.uint8))
    nid, word, bow = v.transform(sets, levelsup=4)
    v.close()
    L = oracle.lib()
    L.plo_bow_vector.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.plo_bow_vector.restype = C.c_int
    ww = voc.word_weight()
    for i, d in enumerate(sets):
        n = len(d)
        w = np.ascontiguousarray(word[i, :n])
        bw, bv = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
        m = L.plo_bow_vector(oracle._p(w), n, oracle._p(ww), 0, 0, oracle._p(bw), oracle._p(bv), max(n, 1))
        assert len(bow[i][0]) == m and (bow[i][0] == bw[:m]).all() and (bow[i][1] == bv[:m]).all(), i
        assert m == 0 or abs(bow[i][1].sum() - 1.0) < 1e-12


# ------------------------------------------------------------------ live reference (build container only)
@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_reference_loads_what_the_library_writes(plslam, synth, emu_lib, tmp_path):
    """plh_vocab_save_binary -> the reference's loadFromBinaryFile -> same transform results as from the reference's own file."""
    L = G.ref_lib()
    g = np.load(os.path.join(GOLD, "ref_voc.npz"))
    for case in CASES[:2]:
        name, seed, k, Lv, stop, dfs, dseed, n, up, scoring, weighting = case
        _, desc = G.voc_case_inputs(synth, _vm(), seed, k, Lv, stop, dseed, n)
        v = plslam.ORBVocabulary(lib=emu_lib)
        v.loadFromTextFile(_files(name)["A"])
        out = str(tmp_path / (name + ".bin"))
        v.saveToBinaryFile(out)
        v.close()
        h = L.ref_voc_load_binary(out.encode())
        assert h
        try:
            r = G.reference_voc_outputs(L, h, desc, up)
        finally:
            L.ref_voc_free(h)
        ref = _golden(g, name, "C")
        for key in ("word", "feat_node", "bow_word", "bow_value"):
            assert (r[key] == ref[key]).all(), (name, key)
