#!/usr/bin/env python3
"""Golden vectors produced by the REFERENCE's own code (oracle/_ref/libdbow2_ref.so = Thirdparty/DBoW2, liborb_ref.so =
src/ORBextractor.cc on the oracle's OpenCV primitives; compiled from
/root/reference by oracle/ref/build_ref.sh): DBoW2::FORB::distance and TemplatedVocabulary::transform on synthetic
vocabularies written in the reference's text format.  Run in the build container (the reference is not on the GPU box):

    bash oracle/ref/build_ref.sh && python tools/gen_golden_ref.py

Writes tests/golden/ref_dbow2_*.npz and ref_orb_*.npz; tests/test_ref_dbow2.py / test_ref_orb.py check the oracle and the
GPU against them."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402


def ref_lib():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdbow2_ref.so"))
    V, I = C.c_void_p, C.c_int
    L.ref_forb_distance.argtypes = [V, V]
    L.ref_voc_load_text.argtypes = [C.c_char_p]
    L.ref_voc_load_text.restype = V
    L.ref_voc_free.argtypes = [V]
    L.ref_voc_size.argtypes = [V]
    L.ref_voc_transform_each.argtypes = [V, V, I, I, V, V, V]
    L.ref_voc_transform.argtypes = [V, V, I, I, V, V, I, V]
    L.ref_voc_load_binary.argtypes = [C.c_char_p]
    L.ref_voc_load_binary.restype = V
    L.ref_voc_save_binary.argtypes = [V, C.c_char_p]
    L.ref_voc_save_text.argtypes = [V, C.c_char_p]
    return L


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def reference_transform(L, voc, desc, levelsup):
    """(word[n], weight[n], node[n], feat_node[n], bow_word[m], bow_value[m]) from the reference."""
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "voc.txt")
        voc.save_text(path)
        h = L.ref_voc_load_text(path.encode())
        assert h, "reference loader rejected the file"
        try:
            assert L.ref_voc_size(h) == int((voc.word_id >= 0).sum())
            n = len(desc)
            word, node, fnode = (np.zeros(n, np.int32) for _ in range(3))
            weight = np.zeros(n, np.float64)
            L.ref_voc_transform_each(h, p(desc), n, levelsup, p(word), p(weight), p(node))
            bw, bv = np.zeros(n, np.int32), np.zeros(n, np.float64)
            m = L.ref_voc_transform(h, p(desc), n, levelsup, p(bw), p(bv), n, p(fnode))
        finally:
            L.ref_voc_free(h)
    return word, weight, node, fnode, bw[:m], bv[:m]


# ---------------------------------------------------------------------------------------------------------------
# Vocabulary FILES written by the reference (saveToTextFile / saveToBinaryFile) and what its transform makes of them.
#   A  ref_voc_<name>_src.txt   the synthetic tree written by pl-slam_amd/vocab.py (weights with all 17 digits)
#   B  ref_voc_<name>_ref.txt   A loaded by the reference and written back by ITS saveToTextFile (6-digit weights)
#   C  ref_voc_<name>_ref.bin   ... by ITS saveToBinaryFile (float weights)
# and the per-feature word / node plus the BowVector (std::map order, doubles) the reference computes after loading each.
# "dfs": the same kind of tree with its nodes numbered depth first, i.e. children NOT contiguous in the file.
# ---------------------------------------------------------------------------------------------------------------
VOC_CASES = [   # name, seed, k, L, stop fraction, dfs order, descriptor seed, n, levelsup, scoring, weighting
    ("k5L3", 31, 5, 3, 0.05, False, 41, 400, 1, 0, 0),       # L1_NORM / TF_IDF like ORBvoc
    ("dfs_k4L3", 32, 4, 3, 0.0, True, 42, 300, 2, 0, 0),
    ("k6L2_l2_tf", 33, 6, 2, 0.1, False, 43, 250, 1, 1, 1),  # L2_NORM / TF
    ("k3L4_dot_bin", 34, 3, 4, 0.0, False, 44, 200, 2, 5, 3),  # DOT_PRODUCT (no normalisation) / BINARY
]


def write_text_dfs(voc, path, scoring, weighting):
    """The tree of `voc` with depth-first node numbering (a legal DBoW2 text file whose children are not contiguous)."""
    lines = ["%d %d %d %d" % (voc.k, voc.L, scoring, weighting)]
    new_id = {0: 0}

    def visit(node):
        for c in range(int(voc.child_start[node]), int(voc.child_start[node]) + int(voc.child_count[node])):
            new_id[c] = len(new_id)
            lines.append("%d %d %s %r" % (new_id[node], 1 if voc.child_count[c] == 0 else 0,
                                          " ".join(str(int(b)) for b in voc.node_desc[c]), float(voc.weight64[c])))
            visit(c)
    visit(0)
    with open(path, "w") as f:
        f.write("\n".join(lines))


def voc_case_inputs(S, VM, seed, k, Lv, stop, dseed, n):
    voc = VM.Vocabulary.synthetic(seed, k=k, L=Lv, synth=S, stop_fraction=stop, idf=True)
    a, _, _ = S.make_descriptor_sets(dseed, n)
    first_leaf = (k ** Lv - 1) // (k - 1)
    rng = S.SplitMix64(dseed + 1)
    pick = rng.randint(n - n // 3, first_leaf, voc.n_nodes)      # two thirds near leaves (several features per word)
    noisy = voc.node_desc[pick] ^ np.packbits((rng.uniform(len(pick) * 256) < 0.02).reshape(-1, 256), axis=1, bitorder="little")
    return voc, np.ascontiguousarray(np.concatenate([a[: n // 3], noisy]), np.uint8)


def reference_voc_outputs(L, h, desc, levelsup):
    n = len(desc)
    word, node, fnode = (np.zeros(n, np.int32) for _ in range(3))
    weight = np.zeros(n, np.float64)
    L.ref_voc_transform_each(h, p(desc), n, levelsup, p(word), p(weight), p(node))
    bw, bv = np.zeros(n, np.int32), np.zeros(n, np.float64)
    m = L.ref_voc_transform(h, p(desc), n, levelsup, p(bw), p(bv), n, p(fnode))
    return dict(word=word, weight=weight, node=node, feat_node=fnode, bow_word=bw[:m].copy(), bow_value=bv[:m].copy())


def gen_vocab_files(S, VM, L, out):
    g = {}
    for name, seed, k, Lv, stop, dfs, dseed, n, up, scoring, weighting in VOC_CASES:
        voc, desc = voc_case_inputs(S, VM, seed, k, Lv, stop, dseed, n)
        a = os.path.join(out, "ref_voc_%s_src.txt" % name)
        b = os.path.join(out, "ref_voc_%s_ref.txt" % name)
        c = os.path.join(out, "ref_voc_%s_ref.bin" % name)
        if dfs:
            write_text_dfs(voc, a, scoring, weighting)
        else:
            voc.save_text(a, scoring, weighting)
        h = L.ref_voc_load_text(a.encode())
        assert h
        for key, v in reference_voc_outputs(L, h, desc, up).items():
            g["%s_A_%s" % (name, key)] = v
        L.ref_voc_save_text(h, b.encode())
        L.ref_voc_save_binary(h, c.encode())
        L.ref_voc_free(h)
        # the reference's text writer ends every line with a newline, which its own reader turns into a stray node with an
        # uninitialised parent (undefined behaviour): strip the final newline before handing the file back to it
        txt = open(b).read().rstrip("\n")
        stripped = b + ".tmp"
        open(stripped, "w").write(txt)
        h = L.ref_voc_load_text(stripped.encode())
        os.remove(stripped)
        for key, v in reference_voc_outputs(L, h, desc, up).items():
            g["%s_B_%s" % (name, key)] = v
        L.ref_voc_free(h)
        h = L.ref_voc_load_binary(c.encode())
        for key, v in reference_voc_outputs(L, h, desc, up).items():
            g["%s_C_%s" % (name, key)] = v
        g["%s_C_size" % name] = np.int32(L.ref_voc_size(h))
        L.ref_voc_free(h)
        print("vocabulary files", name, "bow entries", len(g["%s_A_bow_word" % name]), len(g["%s_C_bow_word" % name]))
    np.savez_compressed(os.path.join(out, "ref_voc.npz"), **g)


CASES = [   # name, vocabulary seed, k, L, stop fraction, descriptor seed, n, levelsup
    ("k10L4", 102, 10, 4, 0.02, 7, 1500, 2),
    ("k8L3", 5, 8, 3, 0.0, 9, 600, 1),
    ("k10L5_up4", 11, 10, 5, 0.01, 13, 1000, 4),
]


def make_case(S, VM, vseed, k, Lv, stop, dseed, n):
    voc = VM.Vocabulary.synthetic(vseed, k=k, L=Lv, synth=S, stop_fraction=stop)
    a, b, _ = S.make_descriptor_sets(dseed, n)
    # half random descriptors, half noisy copies of leaf descriptors (realistic descents, few ties)
    first_leaf = (k ** Lv - 1) // (k - 1)
    rng = S.SplitMix64(dseed + 1)
    pick = rng.randint(n // 2, first_leaf, voc.n_nodes)
    noisy = voc.node_desc[pick] ^ np.packbits((rng.uniform(n // 2 * 256) < 0.04).reshape(-1, 256), axis=1, bitorder="little")
    desc = np.ascontiguousarray(np.concatenate([a[: n - n // 2], noisy]), np.uint8)
    return voc, desc


# ---------------------------------------------------------------------------------------------------------------
# The reference's ORB extractor (src/ORBextractor.cc compiled into oracle/_ref/liborb_ref.so on top of the oracle's
# restated OpenCV primitives, list nodes from a monotonic arena so that its address-ordered tie-break is reproducible)
# ---------------------------------------------------------------------------------------------------------------
ORB_CASES = [   # name, frame seed, rows, cols, nfeatures, scale, nlevels, iniTh, minTh
    ("s1_640x480", 1, 480, 640, 1000, 1.2, 8, 20, 7),
    ("s3_320x240", 3, 240, 320, 500, 1.2, 6, 20, 7),
    ("s7_200x160_sparse", 7, 160, 200, 300, 1.5, 4, 40, 12),   # every level >= 38 px (the GPU plan refuses smaller ones)
    ("kitti_1241x376", 1000, 376, 1241, 2000, 1.2, 8, 20, 7),  # Examples/Monocular/KITTI00-02.yaml:32-51 (4 root nodes)
]


def ref_orb_lib():
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "liborb_ref.so"))
    R.ref_orb_create.restype = C.c_void_p
    R.ref_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
    R.ref_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    R.ref_orb_destroy.argtypes = [C.c_void_p]
    R.ref_orb_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return R


def reference_orb(R, P, img, nf, scale, nl, ini, mn):
    h = R.ref_orb_create(nf, scale, nl, ini, mn)
    try:
        cap = nf * 2 + 64
        kps, desc = np.zeros(cap, P.KP_DTYPE), np.zeros((cap, 32), np.uint8)
        img = np.ascontiguousarray(img)
        n = R.ref_orb_extract(h, p(img), img.shape[0], img.shape[1], img.shape[1], p(kps), p(desc), cap)
        assert n >= 0
        sf, sg = np.zeros(16, np.float32), np.zeros(16, np.float32)
        nlv = C.c_int(0)
        R.ref_orb_tables(h, p(sf), p(sg), C.byref(nlv))
    finally:
        R.ref_orb_destroy(h)
    return kps[:n].copy(), desc[:n].copy(), sf[:nlv.value].copy(), sg[:nlv.value].copy()


def gen_orb(S, out):
    R, P = ref_orb_lib(), _util.plslam()
    for name, seed, rows, cols, nf, scale, nl, ini, mn in ORB_CASES:
        img = S.make_frame(seed, rows, cols)
        kps, desc, sf, sg = reference_orb(R, P, img, nf, scale, nl, ini, mn)
        np.savez_compressed(os.path.join(out, "ref_orb_%s.npz" % name), seed=seed, rows=rows, cols=cols, nfeatures=nf,
                            scale=scale, nlevels=nl, ini=ini, mn=mn, kps=kps, desc=desc, scale_factors=sf, sigma2=sg)
        print("orb", name, "keypoints", len(kps))


# ---------------------------------------------------------------------------------------------------------------
# Frame::AssignFeaturesToGridForLine (src/Frame.cc:295-320) re-enacted around the reference's real LineIterator
# (src/lineIterator.cpp in oracle/_ref/libmisc_ref.so): cell -> line indices in insertion order.
# ---------------------------------------------------------------------------------------------------------------
def reference_line_grid(M, kl, gp_array):
    inv_w, inv_h = np.float32(gp_array[4]), np.float32(gp_array[5])
    cells = [[] for _ in range(64 * 48)]
    buf = np.zeros(2 * 4096, np.int32)
    for i in range(len(kl)):
        x1, y1 = np.float32(kl["startPointX"][i]) * inv_w, np.float32(kl["startPointY"][i]) * inv_h   # float products
        x2, y2 = np.float32(kl["endPointX"][i]) * inv_w, np.float32(kl["endPointY"][i]) * inv_h
        n = M.ref_line_iterator(float(x1), float(y1), float(x2), float(y2), p(buf), 4096)
        assert n <= 4096
        for j in range(n):
            cx, cy = int(buf[2 * j]), int(buf[2 * j + 1])
            if 0 <= cx < 64 and 0 <= cy < 48:
                cells[cx * 48 + cy].append(i)
    start = np.zeros(64 * 48 + 1, np.int32)
    items = []
    for c in range(64 * 48):
        start[c] = len(items)
        items += cells[c]
    start[64 * 48] = len(items)
    return start, np.asarray(items, np.int32)


def ref_misc_lib():
    M = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libmisc_ref.so"))
    M.ref_line_iterator.argtypes = [C.c_double] * 4 + [C.c_void_p, C.c_int]
    return M


def random_keylines(S, P, seed, n, cols=640, rows=480):
    rng = S.SplitMix64(seed)
    kl = np.zeros(n, P.KL_DTYPE)
    sx, sy = rng.uniform(n, -20, cols + 20), rng.uniform(n, -20, rows + 20)     # a few leave the image on purpose
    ang, ln = rng.uniform(n, 0, 2 * np.pi), rng.uniform(n, 0.5, 300)
    kl["startPointX"], kl["startPointY"] = sx.astype(np.float32), sy.astype(np.float32)
    kl["endPointX"], kl["endPointY"] = (sx + ln * np.cos(ang)).astype(np.float32), (sy + ln * np.sin(ang)).astype(np.float32)
    # exact verticals / horizontals / points / grid-aligned ends
    kl["endPointX"][:10] = kl["startPointX"][:10]
    kl["endPointY"][10:20] = kl["startPointY"][10:20]
    kl["endPointX"][20:25], kl["endPointY"][20:25] = kl["startPointX"][20:25], kl["startPointY"][20:25]
    kl["startPointX"][25:35] = np.round(kl["startPointX"][25:35] / 10) * 10
    return kl


def gen_line_grid(S, out):
    M, P = ref_misc_lib(), _util.plslam()
    for name, seed, n, gp in (("plain", 41, 260, P.grid_params(640, 480)),
                              ("bounds", 42, 200, P.grid_params(640, 480, -18.5, -11.25, 661.75, 494.5))):
        kl = random_keylines(S, P, seed, n)
        start, items = reference_line_grid(M, kl, P._gp_array(gp))
        np.savez_compressed(os.path.join(out, "ref_linegrid_%s.npz" % name), seed=seed, n=n, gp=P._gp_array(gp), start=start, items=items)
        print("line grid", name, "items", len(items))


# ---------------------------------------------------------------------------------------------------------------
# The reference's line path: src/LineExtractor.cpp + the vendored twin of opencv_contrib's line_descriptor
# (oracle/_ref/libline_ref.so), LSD / GaussianBlur / Sobel from the oracle.
# ---------------------------------------------------------------------------------------------------------------
LINE_CASES = [   # name, frame seed, rows, cols, nLSDFeature, min_line_length, masked
    ("s1_640x480", 1, 480, 640, 200, 0.0, False),
    ("s3_320x240_minlen", 3, 240, 320, 300, 25.0, False),
    ("s4_403x200_mask", 4, 200, 403, 120, 0.0, True),
    ("kitti_1241x376", 1000, 376, 1241, 200, 0.0, False),     # BASELINE configs[4]'s frame shape
]
LINE_OCTAVE_CASES = [   # LINEextractor(numOctaves = 2, scale = 2.x): name, frame seed, rows, cols, nLSDFeature, min_line_length, masked, scale
    ("oct2_s6_640x480", 6, 480, 640, 200, 0.0, False, 2.0),
    ("oct2_s8_405x203_mask", 8, 203, 405, 120, 12.0, True, 2.5),   # odd sizes: pyrDown to floor(w / 2) x floor(h / 2); (int)2.5 == 2
]


def ref_line_lib():
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libline_ref.so"))
    R.ref_line_create.restype = C.c_void_p
    R.ref_line_create.argtypes = [C.c_int, C.c_float, C.c_uint, C.c_double]
    R.ref_line_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int]
    R.ref_line_destroy.argtypes = [C.c_void_p]
    return R


def line_mask(S, seed, rows, cols):
    """Blocks of zeros (LSDDetector drops a line only if BOTH end points sit on mask == 0)."""
    rng = S.SplitMix64(seed + 900)
    m = np.full((rows, cols), 255, np.uint8)
    for _ in range(6):
        y, x = int(rng.randint(1, 0, rows - 40)[0]), int(rng.randint(1, 0, cols - 60)[0])
        m[y:y + 40, x:x + 60] = 0
    return m


def reference_lines(R, P, img, nf, min_len, mask=None, num_octaves=1, scale=1.2):
    """The reference's LINEextractor(num_octaves, scale, nf, min_len)::operator(); None when it throws (cv::pyrDown's assertion)."""
    h = R.ref_line_create(num_octaves, scale, nf, min_len)
    try:
        cap = nf + 8
        kl, desc, fn = np.zeros(cap, P.KL_DTYPE), np.zeros((cap, 32), np.uint8), np.zeros((cap, 3))
        img = np.ascontiguousarray(img)
        mp = p(mask) if mask is not None else None
        n = R.ref_line_extract(h, p(img), img.shape[0], img.shape[1], img.shape[1], mp, img.shape[1], p(kl), p(desc), p(fn), cap)
        if n == -3:
            return None
        assert n >= 0
    finally:
        R.ref_line_destroy(h)
    return kl[:n].copy(), desc[:n].copy(), fn[:n].copy()


def gen_lines(S, out):
    R, P = ref_line_lib(), _util.plslam()
    for name, seed, rows, cols, nf, min_len, masked in LINE_CASES:
        img = S.make_frame(seed, rows, cols)
        mask = line_mask(S, seed, rows, cols) if masked else None
        kl, desc, fn = reference_lines(R, P, img, nf, min_len, mask)
        np.savez_compressed(os.path.join(out, "ref_line_%s.npz" % name), seed=seed, rows=rows, cols=cols, nfeatures=nf,
                            min_len=min_len, masked=masked, keylines=kl, desc=desc, linefn=fn)
        print("lines", name, "keylines", len(kl))
    for name, seed, rows, cols, nf, min_len, masked, scale in LINE_OCTAVE_CASES:
        img = S.make_frame(seed, rows, cols)
        mask = line_mask(S, seed, rows, cols) if masked else None
        kl, desc, fn = reference_lines(R, P, img, nf, min_len, mask, num_octaves=2, scale=scale)
        np.savez_compressed(os.path.join(out, "ref_line_%s.npz" % name), seed=seed, rows=rows, cols=cols, nfeatures=nf,
                            min_len=min_len, masked=masked, num_octaves=2, scale=scale, keylines=kl, desc=desc, linefn=fn)
        print("lines", name, "keylines", len(kl), "of them in octave 1:", int((kl["octave"] == 1).sum()))


# ---------------------------------------------------------------------------------------------------------------
# The reference's ORBmatcher (src/ORBmatcher.cc compiled into oracle/_ref/libmatcher_ref.so against stand-ins for
# Frame / KeyFrame / MapPoint): the searches without pose algebra.  Inputs come from the test-suite generators.
# ---------------------------------------------------------------------------------------------------------------
def _test_module(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def ref_matcher_lib():
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libmatcher_ref.so"))
    V, I, F = C.c_void_p, C.c_int, C.c_float
    R.ref_orb_search_by_bow.argtypes = [V, V, V, V, I, V, V, V, I, F, I, V]
    R.ref_orb_search_by_bow_kfkf.argtypes = [V, V, V, V, I, V, V, V, V, I, F, I, V]
    R.ref_orb_search_for_initialization.argtypes = [V, V, I, V, V, I, V, V, I, F, I, V]
    R.ref_orb_search_by_projection_mp.argtypes = [V, V, I, V, V, I, V, I, V, V, V, V, V, V, F, F, V]
    R.ref_orb_search_by_projection_frame.argtypes = [V, V, I, V, V, I, V, I, V, V, V, V, V, V, V, V, F, I, I, V, V, V]
    R.ref_orb_search_by_projection_kf.argtypes = [V, V, I, V, V, I, V, I, V, V, V, V, V, V, V, V, V, F, I, I, V, V]
    R.ref_orb_fuse.argtypes = [V, V, I, V, V, V, I, V, I, V, V, V, V, V, V, V, V, V, F, V, V, V, V]
    R.ref_orb_fuse_sim3.argtypes = [V, V, I, V, V, I, V, I, V, V, V, V, V, V, V, V, F, V, V, V, V]
    R.ref_orb_search_by_projection_sim3.argtypes = [V, V, I, V, V, I, V, I, V, V, V, V, V, V, V, V, I, V, V, V, V]
    R.ref_orb_search_by_sim3.argtypes = [V, V, I, V, V, V, V, V, V, V, V, I, V, V, V, V, V, V, V, V, I, V, V, F] + [V] * 7
    R.ref_orb_search_for_triangulation.argtypes = [V, V, V, V, I, V, V, V, V, I, V, V, V, V, V, I, I, V, V]
    return R


BOW_CASES = [(300, 2000, 100, 0.7, 1), (302, 1500, 10, 0.9, 0), (305, 777, 40, 0.7, 1)]     # seed, n, nodes, nnratio, checkOri
KFKF_CASES = [(400, 1500, 80), (401, 300, 5)]                                              # seed, n, nodes (nnratio 0.8)
FRAME_CASES = [(31, 2000, False), (32, 700, True)]                                         # seed, n, distorted bounds


POSE_CASES = [(51, 2000, False), (52, 600, True), (53, 40, False)]                         # seed, n, distorted bounds
POSE_FRAME_VARIANTS = [(0, 15.0, 1), (0, 7.0, 0), (1, 15.0, 1), (2, 15.0, 1), (2, 7.0, 0)]     # mode, th, checkOri
POSE_KF_VARIANTS = [(64, 1), (100, 1), (100, 0)]                                             # ORBdist, checkOri (th = 10)
POSE_K = np.array([517.3, 516.5, 318.6, 255.3], np.float32)                                # fx fy cx cy


def pose_inputs(S, P, TF, seed, n, distorted):
    """Last-frame / keyframe map points at world positions in front of (a few behind) a current camera with pose = identity."""
    f1, f2, _, _ = TF.make_frame_pair(P, S, seed, n, nl=0)
    gp = TF._gp(P, distorted=distorted)
    q = TF._queries_points(P, S, 900 + seed, f1, f2, "frame")
    rng = S.SplitMix64(seed + 7)
    z = rng.uniform(n, 0.8, 6.0).astype(np.float32)
    behind = rng.uniform(n) < 0.04
    z[behind] = -z[behind]
    K = POSE_K
    xyz = np.ascontiguousarray(np.stack([(q["uv"][:, 0] - K[2]) / K[0] * z, (q["uv"][:, 1] - K[3]) / K[1] * z, z], 1).astype(np.float32))
    flags = {k: (rng.uniform(n) < pr).astype(np.uint8) for k, pr in (("mp", 0.9), ("outlier", 0.07), ("bad", 0.05), ("found", 0.1),
                                                                     ("inrange", 0.95))}
    occ_f = (S.SplitMix64(77 + seed).uniform(n) < 0.1).astype(np.uint8)
    occ_k = (S.SplitMix64(99 + seed).uniform(n) < 0.05).astype(np.uint8)
    return f2, gp, q, xyz, flags, occ_f, occ_k


def reference_pose_frame(R, P, TF, f2, gp, q, xyz, fl, occ0, mode, th, chk):
    n = len(f2["kps"])
    g = P._gp_array(gp)
    uv, front = np.zeros((max(n, 1), 2), np.float32), np.zeros(max(n, 1), np.uint8)
    occ, asg = occ0.copy(), np.zeros(max(n, 1), np.int32)
    c = R.ref_orb_search_by_projection_frame(p(f2["kps"]), p(f2["desc"]), n, p(g), p(TF.SCALE), len(TF.SCALE), p(occ), n, p(fl["mp"]),
                                             p(fl["outlier"]), p(xyz), p(q["octave"]), p(q["angle"]), p(q["desc"]), p(q["hasobs"]),
                                             p(POSE_K), th, mode, chk, p(uv), p(front), p(asg))
    valid = (fl["mp"] & (1 - fl["outlier"]) & front[:n]).astype(np.uint8)
    return c, asg[:n], occ, uv[:n].copy(), valid


def reference_pose_kf(R, P, TF, f2, gp, q, xyz, fl, occ0, orb_dist, chk):
    n = len(f2["kps"])
    g = P._gp_array(gp)
    uv = np.zeros((max(n, 1), 2), np.float32)
    occ, asg = occ0.copy(), np.zeros(max(n, 1), np.int32)
    c = R.ref_orb_search_by_projection_kf(p(f2["kps"]), p(f2["desc"]), n, p(g), p(TF.SCALE), len(TF.SCALE), p(occ), n, p(fl["mp"]),
                                          p(fl["bad"]), p(fl["found"]), p(fl["inrange"]), p(xyz), p(q["octave"]), p(q["angle"]),
                                          p(q["desc"]), p(POSE_K), 10.0, orb_dist, chk, p(uv), p(asg))
    valid = (fl["mp"] & (1 - fl["bad"]) & (1 - fl["found"]) & fl["inrange"]).astype(np.uint8)
    return c, asg[:n], occ, uv[:n].copy(), valid


def bow_inputs(S, TM, seed, n, nodes):
    return TM._bow_sets(S, seed, n, nodes)


def kfkf_inputs(S, TM, seed, n, nodes):
    kf, fr = TM._bow_sets(S, seed, n, nodes, valid_p=0.85)
    v2 = (S.SplitMix64(seed + 5).uniform(n) < 0.85).astype(np.uint8)
    return kf, fr, v2


def frame_inputs(S, P, TF, seed, n, distorted):
    f1, f2, _, _ = TF.make_frame_pair(P, S, seed, n, nl=0)
    gp = TF._gp(P, distorted=distorted)
    q = TF._queries_points(P, S, 900 + seed, f1, f2, "mp")
    occ0 = (S.SplitMix64(77 + seed).uniform(n) < 0.1).astype(np.uint8)
    return f1, f2, gp, q, occ0


def reference_bow(R, kf, fr, nnratio, check):
    n1, n2 = len(kf["desc"]), len(fr["desc"])
    out = np.zeros(max(n2, 1), np.int32)
    c = R.ref_orb_search_by_bow(p(kf["desc"]), p(kf["angle"]), p(kf["node"]), p(kf["valid"]), n1, p(fr["desc"]), p(fr["angle"]),
                                p(fr["node"]), n2, nnratio, check, p(out))
    return c, out[:n2]


def reference_kfkf(R, kf, fr, v2):
    n = len(kf["desc"])
    out = np.zeros(max(n, 1), np.int32)
    c = R.ref_orb_search_by_bow_kfkf(p(kf["desc"]), p(kf["angle"]), p(kf["node"]), p(kf["valid"]), n, p(fr["desc"]), p(fr["angle"]),
                                     p(fr["node"]), p(v2), n, 0.8, 1, p(out))
    return c, out[:n]


def reference_init(R, P, f1, f2, gp):
    n1, n2 = len(f1["kps"]), len(f2["kps"])
    prev = np.stack([f1["kps"]["x"], f1["kps"]["y"]], 1).astype(np.float32)
    out = np.zeros(max(n1, 1), np.int32)
    g = P._gp_array(gp)
    c = R.ref_orb_search_for_initialization(p(f1["kps"]), p(f1["desc"]), n1, p(f2["kps"]), p(f2["desc"]), n2, p(g), p(prev), 100, 0.9, 1,
                                            p(out))
    return c, out[:n1], prev


def reference_proj_mp(R, P, f2, gp, q, occ0, scale):
    n = len(f2["kps"])
    occ, asg = occ0.copy(), np.zeros(max(n, 1), np.int32)
    g = P._gp_array(gp)
    c = R.ref_orb_search_by_projection_mp(p(f2["kps"]), p(f2["desc"]), n, p(g), p(scale), len(scale), p(occ), len(q["valid"]),
                                          p(q["valid"]), p(q["xy"]), p(q["level"]), p(q["viewcos"]), p(q["desc"]), p(q["hasobs"]),
                                          3.0, 0.8, p(asg))
    return c, asg[:n], occ


def gen_matchers(S, out):
    R, P = ref_matcher_lib(), _util.plslam()
    TM, TF = _test_module("test_match"), _test_module("test_frame_search")
    g = {}
    for seed, n, nodes, nn, chk in BOW_CASES:
        kf, fr = bow_inputs(S, TM, seed, n, nodes)
        c, m = reference_bow(R, kf, fr, nn, chk)
        g["bow_%d_n" % seed], g["bow_%d_m" % seed] = c, m
    for seed, n, nodes in KFKF_CASES:
        kf, fr, v2 = kfkf_inputs(S, TM, seed, n, nodes)
        c, m = reference_kfkf(R, kf, fr, v2)
        g["kfkf_%d_n" % seed], g["kfkf_%d_m" % seed] = c, m
    for seed, n, dist in FRAME_CASES:
        f1, f2, gp, q, occ0 = frame_inputs(S, P, TF, seed, n, dist)
        c, m, prev = reference_init(R, P, f1, f2, gp)
        g["init_%d_n" % seed], g["init_%d_m" % seed], g["init_%d_prev" % seed] = c, m, prev
        c, a, o = reference_proj_mp(R, P, f2, gp, q, occ0, TF.SCALE)
        g["proj_%d_n" % seed], g["proj_%d_asg" % seed], g["proj_%d_occ" % seed] = c, a, o
    for seed, n, dist in POSE_CASES:
        f2, gp, q, xyz, fl, occ_f, occ_k = pose_inputs(S, P, TF, seed, n, dist)
        for k, (mode, th, chk) in enumerate(POSE_FRAME_VARIANTS):
            c, a, o, uv, valid = reference_pose_frame(R, P, TF, f2, gp, q, xyz, fl, occ_f, mode, th, chk)
            key = "pf_%d_%d" % (seed, k)
            g[key + "_n"], g[key + "_asg"], g[key + "_occ"] = c, a, o
            g["pose_%d_uv" % seed], g["pf_%d_valid" % seed] = uv, valid
        for k, (orb_dist, chk) in enumerate(POSE_KF_VARIANTS):
            c, a, o, uv, valid = reference_pose_kf(R, P, TF, f2, gp, q, xyz, fl, occ_k, orb_dist, chk)
            key = "pk_%d_%d" % (seed, k)
            g[key + "_n"], g[key + "_asg"], g[key + "_occ"] = c, a, o
            assert (uv == g["pose_%d_uv" % seed]).all()
            g["pk_%d_valid" % seed] = valid
    np.savez_compressed(os.path.join(out, "ref_orbmatcher.npz"), **g)
    print("matchers:", {k: int(v) for k, v in g.items() if k.endswith("_n")})


# ---------------------------------------------------------------------------------------------------------------
# The reference's LSDmatcher (src/LSDmatcher.cpp compiled into oracle/_ref/liblsdmatcher_ref.so against the same
# stand-ins + MapLine): FrameBFMatch, SearchDouble(Frame, Frame), both SearchByProjection forms.
# ---------------------------------------------------------------------------------------------------------------
def ref_lsdmatcher_lib():
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "liblsdmatcher_ref.so"))
    V, I, F = C.c_void_p, C.c_int, C.c_float
    R.ref_line_search_double.argtypes = [V, I, V, I, F, V]
    R.ref_line_bfmatch.argtypes = [V, I, V, I, F, F, V]
    R.ref_line_search_by_projection_frame.argtypes = [V, V, V, I, V, V, I, V, V, V, V, V, F, V]
    R.ref_line_search_by_projection_ml.argtypes = [V, V, V, I, V, V, I, V, V, V, V, V, F, F, V]
    R.ref_line_fuse.argtypes = [V, V, I, V, I, V, V, I, V, V, V, V, V, V, V, V, V, F, V, V, V, V]
    R.ref_line_bfmatch_new.argtypes = [V, I, V, I, V, V, V, V, F, F, V]
    R.ref_line_search_for_triangulation_new.argtypes = [V, I, V, I, V, V, V, V, V, V, V, V, V, V, F, I, V, V, V]
    return R


LDOUBLE_CASES = [(1, 300, 280, 0.06, 0.7), (2, 64, 80, 0.20, 0.9), (4, 500, 7, 0.10, 0.8), (5, 37, 411, 0.27, 0.75), (6, 2, 2, 0.05, 0.7)]
LBF_TH = (50.0, 80.0, 25.0)                                  # TH_LOW, TH_HIGH (SearchForTriangulation), a tight one
LPROJ_CASES = [(21, 60, False), (22, 300, True), (23, 1, False), (24, 700, False)]        # seed, lines, distorted bounds
LPROJ_VARIANTS = [("ml", 3.0, 0.9), ("ml", 6.0, 0.7), ("frame", 12.0, 0.9), ("frame", 4.0, 0.9)]


def ldouble_inputs(S, seed, n1, n2, flip):
    a, b, _ = S.make_descriptor_sets(seed, max(n1, n2), flip)
    return np.ascontiguousarray(a[:n1]), np.ascontiguousarray(b[:n2])


def lproj_inputs(S, P, TF, seed, nl, distorted, variant):
    f1, f2, _, _ = TF.make_frame_pair(P, S, seed, 50, nl=nl)
    gp = TF._gp(P, distorted=distorted)
    q = TF._queries_lines(P, S, 950 + seed, f1, variant)
    occ0 = (S.SplitMix64(88 + seed).uniform(len(f2["keylines"])) < 0.1).astype(np.uint8)
    return f2, gp, q, occ0


def reference_ldouble(R, a, b, ratio):
    m = np.zeros(max(len(a), 1), np.int32)
    c = R.ref_line_search_double(p(a), len(a), p(b), len(b), ratio, p(m))
    return c, m[:len(a)]


def reference_lbf(R, a, b, th, ratio):
    m = np.zeros(max(len(a), 1), np.int32)
    R.ref_line_bfmatch(p(a), len(a), p(b), len(b), th, ratio, p(m))
    return m[:len(a)]


def reference_lproj(R, P, f2, gp, q, occ0, variant, th, nn):
    n2 = len(f2["keylines"])
    g = P._gp_array(gp)
    occ, asg = occ0.copy(), np.zeros(max(n2, 1), np.int32)
    if variant == "ml":
        c = R.ref_line_search_by_projection_ml(p(f2["keylines"]), p(f2["ldesc"]), p(f2["linefn"]), n2, p(g), p(occ), len(q["valid"]),
                                               p(q["valid"]), p(q["seg"]), p(q["viewcos"]), p(q["desc"]), p(q["hasobs"]), th, nn, p(asg))
    else:
        c = R.ref_line_search_by_projection_frame(p(f2["keylines"]), p(f2["ldesc"]), p(f2["linefn"]), n2, p(g), p(occ), len(q["valid"]),
                                                  p(q["valid"]), p(q["seg"]), p(q["length"]), p(q["desc"]), p(q["hasobs"]), th, p(asg))
    return c, asg[:n2], occ


def gen_lsdmatcher(S, out):
    R, P = ref_lsdmatcher_lib(), _util.plslam()
    TF = _test_module("test_frame_search")
    g = {}
    for seed, n1, n2, flip, ratio in LDOUBLE_CASES:
        a, b = ldouble_inputs(S, seed, n1, n2, flip)
        c, m = reference_ldouble(R, a, b, ratio)
        g["dbl_%d_n" % seed], g["dbl_%d_m" % seed] = c, m
        for th in LBF_TH:
            g["bf_%d_%d" % (seed, int(th))] = reference_lbf(R, a, b, th, ratio)
    for seed, nl, dist in LPROJ_CASES:
        for k, (variant, th, nn) in enumerate(LPROJ_VARIANTS):
            f2, gp, q, occ0 = lproj_inputs(S, P, TF, seed, nl, dist, variant)
            c, a, o = reference_lproj(R, P, f2, gp, q, occ0, variant, th, nn)
            g["proj_%d_%d_n" % (seed, k)], g["proj_%d_%d_asg" % (seed, k)], g["proj_%d_%d_occ" % (seed, k)] = c, a, o
    for seed, nl, dist, nb in LFUSE_CASES:
        f2, gp, q, pos, fl, kfml, level = lfuse_inputs(S, P, TF, seed, nl, dist, nb)
        for k, th in enumerate(LFUSE_TH):
            c, best, seg, valid, stopped = reference_lfuse(R, P, f2, gp, q, pos, fl, kfml, level, th)
            g["lfuse_%d_%d_n" % (seed, k)], g["lfuse_%d_%d_best" % (seed, k)] = c, best
            g["lfuse_%d_seg" % seed], g["lfuse_%d_valid" % seed], g["lfuse_%d_stopped" % seed] = seg, valid, int(stopped)
    np.savez_compressed(os.path.join(out, "ref_lsdmatcher.npz"), **g)
    print("lsdmatcher:", {k: int(v) for k, v in g.items() if k.endswith("_n")},
          {k: int((v >= 0).sum()) for k, v in g.items() if k.startswith("bf_")})


# ---- KeyFrame-side searches: Fuse (both overloads), loop-closing SearchByProjection, SearchBySim3, SearchForTriangulation
KF_TH = {"fuse": (3.0, 6.0), "fuse3": (3.0, 6.0), "s3p": (10, 4), "sim3": (7.5, 3.0)}
TRI_CASES = [(510, 2000, 100), (512, 1500, 10), (514, 300, 20), (513, 1, 1)]                # seed, n, nodes
TRI_VARIANTS = [(np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32), np.array([-10.0, 0.03, 1.0], np.float32), 1),
                (np.array([1e-6, 2e-5, -0.004, -2e-5, 1e-6, -1.0, 0.005, 1.0, 0.3], np.float32), np.array([0.1, 0.05, 1.0], np.float32), 0)]
TRI_SF = np.cumprod(np.r_[np.float32(1.0), np.full(7, np.float32(1.2))]).astype(np.float32)


def kf_inputs(S, P, TF, seed, n, distorted):
    f2, gp, q, xyz, fl, occ_f, occ_k = pose_inputs(S, P, TF, seed, n, distorted)
    rng = S.SplitMix64(seed + 13)
    kfmp = np.where(rng.uniform(n) < 0.3, 1, 0).astype(np.uint8)
    kfmp[rng.uniform(n) < 0.03] = 2
    fl = dict(fl, inkf=(rng.uniform(n) < 0.08).astype(np.uint8), viewok=(rng.uniform(n) < 0.93).astype(np.uint8))
    slot = np.full(max(n, 1), -1, np.int32)
    if n:
        pick = np.where(rng.uniform(n) < 0.08)[0]
        tgt = rng.randint(len(pick), 0, n) if len(pick) else []
        used = set()
        for a, b in zip(pick, tgt):
            if int(b) not in used:
                slot[a] = b
                used.add(int(b))
    return f2, gp, q, xyz, fl, occ_k, kfmp, slot


def _bufs(n):
    return np.zeros((max(n, 1), 2), np.float32), np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)


def reference_fuse(R, P, TF, f2, gp, q, xyz, fl, kfmp, th):
    n, g = len(f2["kps"]), P._gp_array(gp)
    inv = (np.float32(1.0) / (TF.SCALE * TF.SCALE)).astype(np.float32)
    (uv, fr, im), best = _bufs(n), np.zeros(max(n, 1), np.int32)
    c = R.ref_orb_fuse(p(f2["kps"]), p(f2["desc"]), n, p(g), p(TF.SCALE), p(inv), len(TF.SCALE), p(kfmp), n, p(fl["mp"]), p(fl["bad"]),
                       p(fl["inkf"]), p(fl["inrange"]), p(fl["viewok"]), p(xyz), p(q["octave"]), p(q["desc"]), p(POSE_K), th, p(uv), p(fr),
                       p(im), p(best))
    valid = (fl["mp"] & (1 - fl["bad"]) & (1 - fl["inkf"]) & fr[:n] & im[:n] & fl["inrange"] & fl["viewok"]).astype(np.uint8)
    return c, best[:n], uv[:n].copy(), valid


def reference_fuse3(R, P, TF, f2, gp, q, xyz, fl, kfmp, slot, th):
    n, g = len(f2["kps"]), P._gp_array(gp)
    (uv, fr, im), best = _bufs(n), np.zeros(max(n, 1), np.int32)
    c = R.ref_orb_fuse_sim3(p(f2["kps"]), p(f2["desc"]), n, p(g), p(TF.SCALE), len(TF.SCALE), p(kfmp), n, p(fl["bad"]), p(slot),
                            p(fl["inrange"]), p(fl["viewok"]), p(xyz), p(q["octave"]), p(q["desc"]), p(POSE_K), th, p(uv), p(fr), p(im),
                            p(best))
    valid = ((1 - fl["bad"]) & (slot[:n] < 0) & fr[:n] & im[:n] & fl["inrange"] & fl["viewok"]).astype(np.uint8)
    return c, best[:n], uv[:n].copy(), valid


def reference_s3p(R, P, TF, f2, gp, q, xyz, fl, occ0, slot, th):
    n, g = len(f2["kps"]), P._gp_array(gp)
    (uv, fr, im), asg, occ = _bufs(n), np.zeros(max(n, 1), np.int32), occ0.copy()
    c = R.ref_orb_search_by_projection_sim3(p(f2["kps"]), p(f2["desc"]), n, p(g), p(TF.SCALE), len(TF.SCALE), p(occ), n, p(fl["bad"]),
                                            p(slot), p(fl["inrange"]), p(fl["viewok"]), p(xyz), p(q["octave"]), p(q["desc"]), p(POSE_K),
                                            int(th), p(uv), p(fr), p(im), p(asg))
    valid = ((1 - fl["bad"]) & (slot[:n] < 0) & fr[:n] & im[:n] & fl["inrange"] & fl["viewok"]).astype(np.uint8)
    occ_in = occ0.copy()
    occ_in[slot[:n][slot[:n] >= 0]] = 1          # vpMatched slots holding a query are occupied too
    return c, asg[:n], occ, uv[:n].copy(), valid, occ_in


def sim3_inputs(S, P, TF, seed, n, distorted):
    f1, f2, _, _ = TF.make_frame_pair(P, S, seed, n, nl=0)
    gp = TF._gp(P, distorted=distorted)
    a, c = TF._queries_points(P, S, 970 + seed, f1, f2, "frame"), TF._queries_points(P, S, 980 + seed, f2, f1, "frame")
    rng = S.SplitMix64(seed + 21)
    K, sides = POSE_K, []
    for q in (a, c):
        z = rng.uniform(n, 0.8, 6.0).astype(np.float32)
        behind = rng.uniform(n) < 0.04
        z[behind] = -z[behind]
        xyz = np.ascontiguousarray(np.stack([(q["uv"][:, 0] - K[2]) / K[0] * z, (q["uv"][:, 1] - K[3]) / K[1] * z, z], 1).astype(np.float32))
        sides.append(dict(xyz=xyz, mp=(rng.uniform(n) < 0.9).astype(np.uint8), bad=(rng.uniform(n) < 0.05).astype(np.uint8),
                          inr=(rng.uniform(n) < 0.95).astype(np.uint8), level=q["octave"], desc=q["desc"]))
    already = np.full(max(n, 1), -1, np.int32)
    if n:
        pick = rng.uniform(n) < 0.1
        already[:n][pick] = rng.randint(int(pick.sum()), 0, n + n // 4 + 1)
    return f1, f2, gp, sides, already


def reference_sim3(R, P, TF, f1, f2, gp, sides, already, th):
    n, g = len(f1["kps"]), P._gp_array(gp)
    s1, s2 = sides
    (uv12, fr12, im12), (uv21, fr21, im21), m12 = _bufs(n), _bufs(n), np.zeros(max(n, 1), np.int32)
    c = R.ref_orb_search_by_sim3(p(f1["kps"]), p(f1["desc"]), n, p(s1["mp"]), p(s1["bad"]), p(s1["xyz"]), p(s1["level"]), p(s1["inr"]),
                                 p(s1["desc"]), p(f2["kps"]), p(f2["desc"]), n, p(s2["mp"]), p(s2["bad"]), p(s2["xyz"]), p(s2["level"]),
                                 p(s2["inr"]), p(s2["desc"]), p(g), p(TF.SCALE), len(TF.SCALE), p(already), p(POSE_K), th, p(uv12), p(fr12),
                                 p(im12), p(uv21), p(fr21), p(im21), p(m12))
    am1 = already[:n] >= 0
    am2 = np.zeros(n, bool)
    idx = already[:n][am1]
    am2[idx[idx < n]] = True
    v12 = (s1["mp"] & (1 - s1["bad"]) & ~am1 & fr12[:n] & im12[:n] & s1["inr"]).astype(np.uint8)
    v21 = (s2["mp"] & (1 - s2["bad"]) & ~am2 & fr21[:n] & im21[:n] & s2["inr"]).astype(np.uint8)
    return c, m12[:n], uv12[:n].copy(), v12, uv21[:n].copy(), v21


def reference_tri(R, a, b, F12, cw, chk):
    n = len(a["desc"])
    epi, m = np.zeros(2, np.float32), np.zeros(max(n, 1), np.int32)
    sig2 = (TRI_SF * TRI_SF).astype(np.float32)
    c = R.ref_orb_search_for_triangulation(p(a["kps"]), p(a["desc"]), p(a["node"]), p(a["has_mp"]), n, p(b["kps"]), p(b["desc"]), p(b["node"]),
                                           p(b["has_mp"]), n, p(F12), p(cw), p(POSE_K), p(TRI_SF), p(sig2), 8, chk, p(epi), p(m))
    return c, m[:n], epi


def gen_keyframe_searches(S, out):
    R, P = ref_matcher_lib(), _util.plslam()
    TM, TF = _test_module("test_match"), _test_module("test_frame_search")
    g = {}
    for seed, n, dist in POSE_CASES:
        f2, gp, q, xyz, fl, occ_k, kfmp, slot = kf_inputs(S, P, TF, seed, n, dist)
        for k, th in enumerate(KF_TH["fuse"]):
            c, best, uv, valid = reference_fuse(R, P, TF, f2, gp, q, xyz, fl, kfmp, th)
            g["fuse_%d_%d_n" % (seed, k)], g["fuse_%d_%d_best" % (seed, k)] = c, best
            g["fuse_%d_uv" % seed], g["fuse_%d_valid" % seed] = uv, valid
        for k, th in enumerate(KF_TH["fuse3"]):
            c, best, uv, valid = reference_fuse3(R, P, TF, f2, gp, q, xyz, fl, kfmp, slot, th)
            g["fuse3_%d_%d_n" % (seed, k)], g["fuse3_%d_%d_best" % (seed, k)] = c, best
            g["fuse3_%d_uv" % seed], g["fuse3_%d_valid" % seed] = uv, valid
        for k, th in enumerate(KF_TH["s3p"]):
            c, asg, occ, uv, valid, occ_in = reference_s3p(R, P, TF, f2, gp, q, xyz, fl, occ_k, slot, th)
            key = "s3p_%d_%d" % (seed, k)
            g[key + "_n"], g[key + "_asg"], g[key + "_occ"] = c, asg, occ
            g["s3p_%d_uv" % seed], g["s3p_%d_valid" % seed], g["s3p_%d_occin" % seed] = uv, valid, occ_in
        f1, f2, gp, sides, already = sim3_inputs(S, P, TF, seed, n, dist)
        for k, th in enumerate(KF_TH["sim3"]):
            c, m12, uv12, v12, uv21, v21 = reference_sim3(R, P, TF, f1, f2, gp, sides, already, th)
            g["sim3_%d_%d_n" % (seed, k)], g["sim3_%d_%d_m12" % (seed, k)] = c, m12
            g["sim3_%d_uv12" % seed], g["sim3_%d_v12" % seed], g["sim3_%d_uv21" % seed], g["sim3_%d_v21" % seed] = uv12, v12, uv21, v21
    for seed, n, nodes in TRI_CASES:
        a, b = TM._tri_case(P, S, seed, n, nodes)
        for k, (F12, cw, chk) in enumerate(TRI_VARIANTS):
            c, m, epi = reference_tri(R, a, b, F12, cw, chk)
            g["tri_%d_%d_n" % (seed, k)], g["tri_%d_%d_m" % (seed, k)], g["tri_%d_epi" % k] = c, m, epi
    np.savez_compressed(os.path.join(out, "ref_orbmatcher_kf.npz"), **g)
    print("keyframe searches:", {k: int(v) for k, v in g.items() if k.endswith("_n")})


# ---- the search inside LSDmatcher::Fuse
LFUSE_CASES = [(21, 60, False, 0), (22, 300, True, 0), (24, 700, False, 1), (26, 201, False, 0)]   # seed, lines, distorted, one line behind
LFUSE_TH = (6.0, 3.0)
LFUSE_SFL = np.array([1.0, 1.4142135, 2.0, 2.828427], np.float32)


def lfuse_inputs(S, P, TF, seed, nl, distorted, nbehind):
    gp = TF._gp(P, distorted=distorted)
    f1, f2, _, _ = TF.make_frame_pair(P, S, seed, 50, nl=nl)
    q = TF._queries_lines(P, S, 970 + seed, f1, "ml")
    rng = S.SplitMix64(seed + 31)
    K = POSE_K
    z1 = rng.uniform(nl, 0.8, 6.0).astype(np.float32)
    z2 = (z1 + rng.uniform(nl, -0.3, 0.3)).astype(np.float32)
    if nbehind and nl:
        z2[int(nl * 0.8)] = -1.0
    seg = q["seg"]
    pos = np.ascontiguousarray(np.stack([(seg[:, 0] - K[2]) / K[0] * z1, (seg[:, 1] - K[3]) / K[1] * z1, z1, (seg[:, 2] - K[2]) / K[0] * z2,
                                         (seg[:, 3] - K[3]) / K[1] * z2, z2], 1).astype(np.float32))
    fl = {k: (rng.uniform(nl) < pr).astype(np.uint8) for k, pr in (("ml", 0.92), ("bad", 0.05), ("inkf", 0.08), ("inrange", 0.95),
                                                                    ("viewok", 0.93))}
    kfml = np.where(rng.uniform(nl) < 0.3, 1, 0).astype(np.uint8)
    kfml[rng.uniform(nl) < 0.03] = 2
    level = rng.randint(nl, 0, 2).astype(np.int32)
    f2["keylines"]["octave"] = rng.randint(len(f2["keylines"]), 0, 2)
    return f2, gp, q, pos, fl, kfml, level


def reference_lfuse(R, P, f2, gp, q, pos, fl, kfml, level, th):
    nl, n2, g = len(level), len(f2["keylines"]), P._gp_array(gp)
    sg, fr, im = np.zeros((max(nl, 1), 4), np.float32), np.zeros(max(nl, 1), np.uint8), np.zeros(max(nl, 1), np.uint8)
    best = np.zeros(max(nl, 1), np.int32)
    c = R.ref_line_fuse(p(f2["keylines"]), p(f2["ldesc"]), n2, p(LFUSE_SFL), 4, p(kfml), p(g), nl, p(fl["ml"]), p(fl["bad"]), p(fl["inkf"]),
                        p(fl["inrange"]), p(fl["viewok"]), p(pos), p(level), p(q["desc"]), p(POSE_K), th, p(sg), p(fr), p(im), p(best))
    pre = (fl["ml"] & (1 - fl["bad"]) & (1 - fl["inkf"])).astype(bool)
    stop = np.where(pre & (fr[:nl] == 0))[0]       # LSDmatcher.cpp:893-894: `return false` leaves the whole function
    alive = np.ones(nl, bool)
    if len(stop):
        alive[stop[0]:] = False
    valid = (pre & alive & im[:nl].astype(bool) & fl["inrange"].astype(bool) & fl["viewok"].astype(bool)).astype(np.uint8)
    return c, best[:nl], sg[:nl].copy(), valid, len(stop) > 0

# ---- MapPoint / MapLine::ComputeDistinctiveDescriptors (the reference's src/MapPoint.cc, src/MapLine.cpp in libmapobj_ref.so)
DISTINCT_SIZES = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 33, 64, 65, 200, 700]


def ref_mapobj_lib():
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libmapobj_ref.so"))
    V, I = C.c_void_p, C.c_int
    R.ref_mappoint_distinctive.argtypes = [V, I, V, V]
    R.ref_mapline_distinctive.argtypes = [V, I, V, V]
    return R


def distinctive_inputs(S, seed=5, sizes=DISTINCT_SIZES, reps=4):
    """Observation sets of one map element: noisy copies of a descriptor; some observing keyframes are bad (dropped)."""
    rng = S.SplitMix64(seed)
    out = []
    for n in sizes:
        for rep in range(reps):
            base = S.make_descriptor_sets(200 + n + rep, 1, 0.0)[0][0]
            flips = (rng.uniform(n * 256) < (0.1 if rep % 2 else 0.02)).reshape(n, 256)
            rows = np.ascontiguousarray(base[None, :] ^ np.packbits(flips, axis=1, bitorder="little"))
            bad = (rng.uniform(n) < (0.2 if rep >= 2 else 0.0)).astype(np.uint8)
            out.append((rows, bad))
    return out


def reference_distinctive(R, rows, bad, line):
    out = np.zeros(32, np.uint8)
    fn = R.ref_mapline_distinctive if line else R.ref_mappoint_distinctive
    rc = fn(p(rows), len(rows), p(bad), p(out))
    return rc, out


def gen_distinctive(S, out):
    R = ref_mapobj_lib()
    g = {}
    for i, (rows, bad) in enumerate(distinctive_inputs(S)):
        for line in (0, 1):
            rc, d = reference_distinctive(R, rows, bad, line)
            g["d_%d_%d_rc" % (i, line)], g["d_%d_%d" % (i, line)] = rc, d
    np.savez_compressed(os.path.join(out, "ref_distinctive.npz"), **g)
    print("distinctive descriptors:", len(g) // 2, "cases,", sum(int(v) for k, v in g.items() if k.endswith("_rc")), "with a result")

# ---- Frame's spatial index: the reference's src/Frame.cc (libframe_ref.so)
FRAMEGRID_CASES = [(1, 2000, 201, False), (2, 700, 60, True), (3, 5, 1, False), (5, 1200, 300, True)]   # seed, n, lines, distorted bounds


def ref_frame_lib(path=None):
    """libframe_ref.so (the reference's classes throughout) or, with `path`, a library with the same harness built differently
    (oracle/_ref/libadaptor_*.so: the adaptor extractor / matcher classes in place of the reference's)."""
    R = C.CDLL(path or os.path.join(ROOT, "oracle", "_ref", "libframe_ref.so"))
    V, I, F = C.c_void_p, C.c_int, C.c_float
    R.ref_frame_create.restype = V
    R.ref_frame_create.argtypes = [V, I, V, V, I, V]
    R.ref_frame_destroy.argtypes = [V]
    R.ref_frame_grid_points.argtypes = [V, V, V, I]
    R.ref_frame_grid_lines.argtypes = [V, V, V, I]
    R.ref_frame_features_in_area.argtypes = [V, F, F, F, I, I, V, I]
    R.ref_frame_features_in_area_for_line.argtypes = [V, F, F, F, F, F, F, V, I]
    R.ref_keyframe_features_in_area.argtypes = [V, F, F, F, V, I]
    R.ref_keyframe_lines_in_area.argtypes = [V, F, F, F, F, F, F, V, I]
    R.ref_track_reference_keyframe.argtypes = [C.c_char_p, V, V, V, I, V, V, I, F, I, V]
    R.ref_track_last_frame.argtypes = [V, V, V, I, V, V, I, V, V, V, V, V, V, V, F, V]
    R.ref_track_local_points.argtypes = [V, V, V, I, V, V, I, V, V, V, V, V, V, F, V]
    R.ref_track_local_lines.argtypes = [V, V, V, V, I, V, I, V, V, V, V, V, V, F, V]
    R.ref_frame_is_in_frustum_points.argtypes = [V, I, I, V, V, V, V, F, V, V, V, V]
    R.ref_frame_is_in_frustum_lines.argtypes = [V, I, V, V, V, V, F, V, V, V, V]
    return R


def framegrid_inputs(S, P, TF, seed, n, nl, distorted):
    f1, f2, _, _ = TF.make_frame_pair(P, S, seed, n, nl=nl)
    gp = TF._gp(P, distorted=distorted)
    rng = S.SplitMix64(seed + 100)
    nq = 300
    pq = np.stack([rng.uniform(nq, -30, 670), rng.uniform(nq, -30, 510), rng.uniform(nq, 1, 60)], 1).astype(np.float32)
    lv = np.array([(-1, -1), (0, 3), (2, -1), (1, 2), (3, 7), (0, 0)], np.int32)[np.arange(nq) % 6]
    # line queries: the frame's own lines, displaced and turned a little (so that the direction test passes for some)
    kl = f2["keylines"]
    if len(kl):
        pick = rng.randint(nq, 0, len(kl))
        seg = np.stack([kl["startPointX"][pick], kl["startPointY"][pick], kl["endPointX"][pick], kl["endPointY"][pick]], 1)
        seg = (seg + rng.uniform(nq * 4, -6, 6).reshape(nq, 4)).astype(np.float32)
    else:
        seg = np.stack([rng.uniform(nq, 0, 640), rng.uniform(nq, 0, 480), rng.uniform(nq, 0, 640), rng.uniform(nq, 0, 480)], 1).astype(np.float32)
    lr = rng.uniform(nq, 1, 20).astype(np.float32)
    lth = np.array([0.998, 0.96, 0.9], np.float32)[np.arange(nq) % 3]
    return f2, gp, pq, lv, seg, lr, lth


def reference_framegrid(R, P, f2, gp, pq, lv, seg, lr, lth):
    n, nl, g = len(f2["kps"]), len(f2["keylines"]), P._gp_array(gp)
    h = R.ref_frame_create(p(f2["kps"]), n, p(f2["keylines"]), p(f2["linefn"]), nl, p(g))
    cs, ci = np.zeros(64 * 48 + 1, np.int32), np.zeros(max(n, 1), np.int32)
    R.ref_frame_grid_points(h, p(cs), p(ci), len(ci))
    lcs, lci = np.zeros(64 * 48 + 1, np.int32), np.zeros(max(nl, 1) * 64, np.int32)
    R.ref_frame_grid_lines(h, p(lcs), p(lci), len(lci))
    buf = np.zeros(max(n, nl, 1) + 1, np.int32)
    pa, la = [], []
    for q in range(len(pq)):
        k = R.ref_frame_features_in_area(h, float(pq[q, 0]), float(pq[q, 1]), float(pq[q, 2]), int(lv[q, 0]), int(lv[q, 1]), p(buf), len(buf))
        pa.append(buf[:k].copy())
        k = R.ref_frame_features_in_area_for_line(h, float(seg[q, 0]), float(seg[q, 1]), float(seg[q, 2]), float(seg[q, 3]), float(lr[q]),
                                                  float(lth[q]), p(buf), len(buf))
        la.append(buf[:k].copy())
    ka, kla = [], []          # the same windows through a real KeyFrame built from this Frame (no level filter; brute-force lines)
    for q in range(len(pq)):
        k = R.ref_keyframe_features_in_area(h, float(pq[q, 0]), float(pq[q, 1]), float(pq[q, 2]), p(buf), len(buf))
        ka.append(buf[:k].copy())
        k = R.ref_keyframe_lines_in_area(h, float(seg[q, 0]), float(seg[q, 1]), float(seg[q, 2]), float(seg[q, 3]), float(lr[q]) * 4,
                                         float(lth[q]), p(buf), len(buf))
        kla.append(buf[:k].copy())
    R.ref_frame_destroy(h)
    flat = lambda v: (np.concatenate(v).astype(np.int32) if sum(len(a) for a in v) else np.zeros(0, np.int32),
                      np.cumsum([0] + [len(a) for a in v]).astype(np.int32))
    return cs, ci, lcs, lci, flat(pa), flat(la), flat(ka), flat(kla)


def gen_framegrid(S, out):
    R, P = ref_frame_lib(), _util.plslam()
    TF = _test_module("test_frame_search")
    g = {}
    for seed, n, nl, dist in FRAMEGRID_CASES:
        f2, gp, pq, lv, seg, lr, lth = framegrid_inputs(S, P, TF, seed, n, nl, dist)
        cs, ci, lcs, lci, (pa, po), (la, lo), (ka, ko), (kla, klo) = reference_framegrid(R, P, f2, gp, pq, lv, seg, lr, lth)
        for k, v in (("cs", cs), ("ci", ci), ("lcs", lcs), ("lci", lci[:lcs[-1]]), ("pa", pa), ("po", po), ("la", la), ("lo", lo),
                     ("ka", ka), ("ko", ko), ("kla", kla), ("klo", klo)):
            g["%s_%d" % (k, seed)] = v
        print("frame grid", seed, "points placed", int(cs[-1]), "line items", int(lcs[-1]), "point hits", len(pa), "line hits", len(la), "keyframe point / line hits", len(ka), len(kla))
    np.savez_compressed(os.path.join(out, "ref_framegrid.npz"), **g)

# ---- Frame::isInFrustum (points and lines) through the reference's Frame.cc / MapPoint.cc / MapLine.cpp (libframe_ref.so)
FRUSTUM_CASES = [(1, 3000, False), (2, 500, True), (3, 1, False)]       # seed, map elements, distorted image bounds
FRUSTUM_COS = 0.5


def frustum_view(S, P, TF, seed, distorted, rotate):
    """plh_frame_view fields as the flat float[24] the oracle takes (+ nlevels).  rotate=False: Rcw = I (what the reference
    harness can be driven with), any translation."""
    rng = S.SplitMix64(seed + 500)
    R = np.eye(3)
    if rotate:
        ax, ay, az = rng.uniform(3, -0.4, 0.4)
        cx_, sx_, cy_, sy_, cz_, sz_ = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
        R = (np.array([[cz_, -sz_, 0], [sz_, cz_, 0], [0, 0, 1]]) @ np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]]) @
             np.array([[1, 0, 0], [0, cx_, -sx_], [0, sx_, cx_]]))
    R = R.astype(np.float32)
    t = rng.uniform(3, -0.5, 0.5).astype(np.float32)
    Ow = (-(R.T.astype(np.float64) @ t.astype(np.float64))).astype(np.float32) if rotate else (-t).astype(np.float32)
    gp = P._gp_array(TF._gp(P, distorted=distorted))
    view = np.concatenate([R.reshape(9), t, Ow, POSE_K, gp[:4], [np.float32(np.log(np.float32(1.2)))]]).astype(np.float32)
    return view, 8


def frustum_elems(S, seed, n, view, lines):
    """World positions scattered around the viewing cone: most in front, some behind / outside / too far / seen edge-on."""
    rng = S.SplitMix64(seed + (900 if lines else 700))
    K = POSE_K
    R, t = view[:9].reshape(3, 3).astype(np.float64), view[9:12].astype(np.float64)

    def world(u, v, z):
        pc = np.stack([(u - K[2]) / K[0] * z, (v - K[3]) / K[1] * z, z], 1)
        return ((pc - t) @ R).astype(np.float32)          # R^T (pc - t)
    u, v = rng.uniform(n, -80, 720), rng.uniform(n, -60, 540)
    z = rng.uniform(n, 0.5, 8.0)
    z[rng.uniform(n) < 0.05] *= -1
    pos = world(u, v, z)
    if lines:
        u2, v2 = u + rng.uniform(n, -90, 90), v + rng.uniform(n, -90, 90)
        z2 = z + rng.uniform(n, -0.4, 0.4)
        pos = np.ascontiguousarray(np.concatenate([pos, world(u2, v2, z2)], 1))
        mid = 0.5 * (pos[:, :3] + pos[:, 3:])
    else:
        mid = pos
    Ow = view[12:15]
    d = np.linalg.norm(mid - Ow, axis=1)
    nrm = (mid - Ow) / np.maximum(d, 1e-6)[:, None]
    nrm = nrm + rng.uniform(n * 3, -0.9, 0.9).reshape(n, 3)            # tilt the mean viewing direction, some beyond 60 deg
    nrm = (nrm / np.maximum(np.linalg.norm(nrm, axis=1), 1e-6)[:, None]).astype(np.float32)
    lvl = rng.randint(n, 0, 8)
    maxd = (d * (1.2 ** lvl) * rng.uniform(n, 0.7, 1.5)).astype(np.float32)       # mfMaxDistance = dist * scale[level] at creation
    mind = (maxd / np.float32(1.2 ** 7)).astype(np.float32)
    return dict(pos=np.ascontiguousarray(pos), normal=np.ascontiguousarray(nrm), min_dist=mind, max_dist=maxd)


def reference_frustum(R, view, nlevels, e, lines):
    n = len(e["min_dist"])
    valid, proj = np.zeros(max(n, 1), np.uint8), np.zeros((max(n, 1), 4 if lines else 2), np.float32)
    level, vc = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float32)
    if lines:
        R.ref_frame_is_in_frustum_lines(p(view), n, p(e["pos"]), p(e["normal"]), p(e["min_dist"]), p(e["max_dist"]), FRUSTUM_COS, p(valid),
                                        p(proj), p(level), p(vc))
    else:
        R.ref_frame_is_in_frustum_points(p(view), nlevels, n, p(e["pos"]), p(e["normal"]), p(e["min_dist"]), p(e["max_dist"]), FRUSTUM_COS,
                                         p(valid), p(proj), p(level), p(vc))
    return valid[:n], proj[:n], level[:n], vc[:n]


def gen_frustum(S, out):
    R, P = ref_frame_lib(), _util.plslam()
    TF = _test_module("test_frame_search")
    g = {}
    for seed, n, dist in FRUSTUM_CASES:
        view, nlv = frustum_view(S, P, TF, seed, dist, rotate=False)
        for lines in (0, 1):
            e = frustum_elems(S, seed, n, view, lines)
            valid, proj, level, vc = reference_frustum(R, view, nlv, e, lines)
            key = "%s_%d" % ("l" if lines else "p", seed)
            g[key + "_valid"], g[key + "_proj"], g[key + "_level"], g[key + "_vc"] = valid, proj, level, vc
            print("frustum", key, "in view", int(valid.sum()), "of", n)
    np.savez_compressed(os.path.join(out, "ref_frustum.npz"), **g)

# ---- Tracking::SearchLocalPoints / SearchLocalLines on real objects: isInFrustum over a local map, then SearchByProjection
TRACK_CASES = [(1, 2000, 201, False, 1.0), (2, 800, 120, True, 5.0), (3, 30, 4, False, 3.0)]   # seed, keypoints, lines, distorted, th


def track_inputs(S, P, TF, seed, n, nl, distorted):
    """Current frame = frame 2 of a synthetic pair; local map = frame 1's features lifted to world points / lines that
    project near their counterparts (camera without rotation, translated), with descriptors, distance ranges that predict
    the counterpart's octave, a few behind the camera / edge-on / out of range."""
    f1, f2, _, _ = TF.make_frame_pair(P, S, seed, n, nl=nl)
    view, nlv = frustum_view(S, P, TF, seed, distorted, rotate=False)
    K, t, Ow = POSE_K, view[9:12].astype(np.float64), view[12:15]
    rng = S.SplitMix64(seed + 1300)

    def lift(u, v, z):
        return (np.stack([(u - K[2]) / K[0] * z, (v - K[3]) / K[1] * z, z], 1) - t).astype(np.float32)

    def ranges(mid, octave, m):
        d = np.linalg.norm(mid - Ow, axis=1)
        nrm = (mid - Ow) / np.maximum(d, 1e-6)[:, None] + rng.uniform(m * 3, -0.5, 0.5).reshape(m, 3)
        nrm = (nrm / np.maximum(np.linalg.norm(nrm, axis=1), 1e-6)[:, None]).astype(np.float32)
        maxd = (d * np.float32(1.2) ** octave * 0.98).astype(np.float32)
        maxd[rng.uniform(m) < 0.04] *= 0.3                      # out of the distance-invariance range
        return np.ascontiguousarray(nrm), (maxd / np.float32(1.2 ** 7)).astype(np.float32), maxd
    k = f1["kps"]
    z = rng.uniform(n, 0.8, 6.0)
    z[rng.uniform(n) < 0.04] *= -1
    pos = lift(k["x"] + rng.uniform(n, -3, 3), k["y"] + rng.uniform(n, -3, 3), z)
    nrm, mind, maxd = ranges(pos, k["octave"], n)
    pts = dict(pos=np.ascontiguousarray(pos), normal=nrm, min_dist=mind, max_dist=maxd, desc=f1["desc"].copy(),
               hasobs=(rng.uniform(n) < 0.9).astype(np.uint8))
    kl = f1["keylines"]
    z1 = rng.uniform(nl, 0.8, 6.0)
    z1[rng.uniform(nl) < 0.04] *= -1
    z2 = z1 + rng.uniform(nl, -0.2, 0.2)
    j = rng.uniform(nl * 4, -2, 2).reshape(nl, 4)
    lp = np.ascontiguousarray(np.concatenate([lift(kl["startPointX"] + j[:, 0], kl["startPointY"] + j[:, 1], z1),
                                              lift(kl["endPointX"] + j[:, 2], kl["endPointY"] + j[:, 3], z2)], 1))
    lnrm, lmind, lmaxd = ranges(0.5 * (lp[:, :3] + lp[:, 3:]), np.zeros(nl), nl)
    lns = dict(pos=lp, normal=lnrm, min_dist=lmind, max_dist=lmaxd, desc=f1["ldesc"].copy(), hasobs=(rng.uniform(nl) < 0.9).astype(np.uint8))
    occ_p = (S.SplitMix64(77 + seed).uniform(n) < 0.1).astype(np.uint8)
    occ_l = (S.SplitMix64(88 + seed).uniform(len(f2["keylines"])) < 0.1).astype(np.uint8)
    return f2, TF._gp(P, distorted=distorted), view, nlv, pts, lns, occ_p, occ_l


def reference_track(R, P, TF, f2, gp, view, nlv, pts, lns, occ_p, occ_l, th):
    n, nl, g = len(f2["kps"]), len(f2["keylines"]), P._gp_array(gp)
    h = R.ref_frame_create(p(f2["kps"]), n, p(f2["keylines"]), p(f2["linefn"]), nl, p(g))
    op, ap = occ_p.copy(), np.zeros(max(n, 1), np.int32)
    cp = R.ref_track_local_points(h, p(f2["desc"]), p(view), nlv, p(TF.SCALE), p(op), len(pts["min_dist"]), p(pts["pos"]), p(pts["normal"]),
                                  p(pts["min_dist"]), p(pts["max_dist"]), p(pts["desc"]), p(pts["hasobs"]), th, p(ap))
    ol, al = occ_l.copy(), np.zeros(max(nl, 1), np.int32)
    cl = R.ref_track_local_lines(h, p(f2["ldesc"]), p(view), p(TF.SCALE), nlv, p(ol), len(lns["min_dist"]), p(lns["pos"]), p(lns["normal"]),
                                 p(lns["min_dist"]), p(lns["max_dist"]), p(lns["desc"]), p(lns["hasobs"]), th, p(al))
    R.ref_frame_destroy(h)
    return (cp, ap[:n], op), (cl, al[:nl], ol)


def reference_track_last(R, P, TF, f2, gp, view, nlv, pts, flags, q, occ, th):
    n, g = len(f2["kps"]), P._gp_array(gp)
    h = R.ref_frame_create(p(f2["kps"]), n, p(f2["keylines"]), p(f2["linefn"]), len(f2["keylines"]), p(g))
    o, a = occ.copy(), np.zeros(max(n, 1), np.int32)
    c = R.ref_track_last_frame(h, p(f2["desc"]), p(view), nlv, p(TF.SCALE), p(o), len(pts["min_dist"]), p(flags["mp"]), p(flags["outlier"]),
                               p(pts["pos"]), p(q["octave"]), p(q["angle"]), p(pts["desc"]), p(pts["hasobs"]), th, p(a))
    R.ref_frame_destroy(h)
    return c, a[:n], o


def track_last_inputs(S, P, TF, seed, n, nl, distorted):
    """TrackWithMotionModel: the last frame's map points (the local map of track_inputs) with its keypoints' octave / angle."""
    f1, _, _, _ = TF.make_frame_pair(P, S, seed, n, nl=nl)
    rng = S.SplitMix64(seed + 1700)
    flags = dict(mp=(rng.uniform(n) < 0.9).astype(np.uint8), outlier=(rng.uniform(n) < 0.07).astype(np.uint8))
    q = dict(octave=f1["kps"]["octave"].astype(np.int32), angle=f1["kps"]["angle"].astype(np.float32))
    return flags, q


BOWTRACK_CASES = [(601, 10, 4, 1500, 0.7, 1), (602, 8, 3, 400, 0.9, 0), (603, 10, 6, 2000, 0.7, 1)]   # seed, k, L, n, nnratio, checkOri


def bowtrack_inputs(S, P, VM, seed, k, Lv, n):
    """A synthetic vocabulary, a keyframe's descriptors (half of them near vocabulary leaves) and a frame observing the same
    features: permuted, noisy copies with slightly turned orientations."""
    voc, d1 = make_case(S, VM, seed, k, Lv, 0.0, seed + 1, n)
    rng = S.SplitMix64(seed + 3)
    perm = np.argsort(rng.uniform(n))
    d2 = np.ascontiguousarray(d1[perm] ^ np.packbits((rng.uniform(n * 256) < 0.05).reshape(n, 256), axis=1, bitorder="little"))
    k1 = np.zeros(n, P.KP_DTYPE)
    k1["x"], k1["y"] = rng.uniform(n, 20, 620).astype(np.float32), rng.uniform(n, 20, 460).astype(np.float32)
    k1["angle"] = rng.uniform(n, 0, 360).astype(np.float32)
    k1["octave"] = rng.randint(n, 0, 8)
    k1["size"], k1["class_id"] = 31, -1
    k2 = k1[perm].copy()
    k2["angle"] = ((k2["angle"] + 12.0 + rng.uniform(n, -4, 4)) % 360).astype(np.float32)
    wrong = rng.uniform(n) < 0.1
    k2["angle"][wrong] = rng.uniform(int(wrong.sum()), 0, 360).astype(np.float32)
    valid = (rng.uniform(n) < 0.8).astype(np.uint8)
    return voc, dict(kps=k1, desc=d1, valid=valid), dict(kps=k2, desc=d2)


def reference_bowtrack(R, voc, kf, fr, nnratio, chk, tmpdir):
    path = os.path.join(tmpdir, "voc_%d.txt" % len(voc.node_desc))
    voc.save_text(path)
    n1, n2 = len(kf["desc"]), len(fr["desc"])
    m = np.zeros(max(n2, 1), np.int32)
    c = R.ref_track_reference_keyframe(path.encode(), p(kf["kps"]), p(kf["desc"]), p(kf["valid"]), n1, p(fr["kps"]), p(fr["desc"]), n2,
                                       nnratio, chk, p(m))
    os.remove(path)
    return c, m[:n2]


def gen_track(S, out):
    R, P = ref_frame_lib(), _util.plslam()
    TF = _test_module("test_frame_search")
    g = {}
    for seed, n, nl, dist, th in TRACK_CASES:
        f2, gp, view, nlv, pts, lns, occ_p, occ_l = track_inputs(S, P, TF, seed, n, nl, dist)
        (cp, ap, op), (cl, al, ol) = reference_track(R, P, TF, f2, gp, view, nlv, pts, lns, occ_p, occ_l, th)
        g["p_%d_n" % seed], g["p_%d_asg" % seed], g["p_%d_occ" % seed] = cp, ap, op
        g["l_%d_n" % seed], g["l_%d_asg" % seed], g["l_%d_occ" % seed] = cl, al, ol
        flags, q = track_last_inputs(S, P, TF, seed, n, nl, dist)
        c, a, o = reference_track_last(R, P, TF, f2, gp, view, nlv, pts, flags, q, occ_p, 15.0 if seed != 2 else 7.0)
        g["m_%d_n" % seed], g["m_%d_asg" % seed], g["m_%d_occ" % seed] = c, a, o
        print("   motion-model search matched", c)
        print("local map search", seed, "points matched", cp, "of", n, "lines matched", cl, "of", nl)
    import tempfile
    VM = _util._load("plslam_amd_vocab", os.path.join(ROOT, "pl-slam_amd", "vocab.py"))
    for seed, k, Lv, n, nn, chk in BOWTRACK_CASES:
        voc, kf, fr = bowtrack_inputs(S, P, VM, seed, k, Lv, n)
        c, m = reference_bowtrack(R, voc, kf, fr, nn, chk, tempfile.gettempdir())
        g["b_%d_n" % seed], g["b_%d_m" % seed] = c, m
        print("reference-keyframe search", seed, "matched", c, "of", n)
    np.savez_compressed(os.path.join(out, "ref_track.npz"), **g)

# ---------------------------------------------------------------------------------------------------------------
# Frame::ComputeBoW / KeyFrame::ComputeBoW on a real Frame with a real ORBVocabulary (oracle/ref/ref_frame.cc: ref_compute_bow);
# the same harness function in libadaptor_*.so runs them through the product's drop-in ORBVocabulary class.
# ---------------------------------------------------------------------------------------------------------------
COMPUTEBOW_CASES = [   # seed, k, L, stop fraction, idf-like weights, n descriptors, binary file, scoring, weighting
    (701, 10, 4, 0.02, True, 1500, False, 0, 0), (702, 8, 3, 0.0, False, 400, False, 0, 0), (703, 10, 5, 0.01, True, 2000, True, 0, 0),
    (704, 9, 3, 0.05, True, 1, False, 0, 0), (705, 10, 4, 0.0, True, 700, False, 1, 1), (706, 6, 4, 0.1, True, 900, True, 5, 3),
]


def write_binary_voc(voc, path, scoring, weighting):
    """DBoW2's binary vocabulary (TemplatedVocabulary::saveToBinaryFile, :1511-1536): u32 nodes, u32 41, k L scoring weighting,
    then per non-root node u32 parent, 32 descriptor bytes, f32 weight, u8 is_leaf."""
    parent, leaf = voc.tree_arrays()
    with open(path, "wb") as f:
        f.write(np.array([voc.n_nodes, 41], np.uint32).tobytes())
        f.write(np.array([voc.k, voc.L, scoring, weighting], np.int32).tobytes())
        for i in range(1, voc.n_nodes):
            f.write(np.uint32(parent[i]).tobytes() + voc.node_desc[i].tobytes() + np.float32(voc.weight64[i]).tobytes() +
                    np.uint8(leaf[i]).tobytes())


def computebow_inputs(S, VM, seed, k, Lv, stop, idf, n):
    voc = VM.Vocabulary.synthetic(seed, k=k, L=Lv, synth=S, stop_fraction=stop, idf=idf)
    a, _, _ = S.make_descriptor_sets(seed + 1, max(n, 2))
    first_leaf = (k ** Lv - 1) // (k - 1)
    rng = S.SplitMix64(seed + 2)
    pick = rng.randint(max(n // 2, 1), first_leaf, voc.n_nodes)
    noisy = voc.node_desc[pick] ^ np.packbits((rng.uniform(len(pick) * 256) < 0.04).reshape(-1, 256), axis=1, bitorder="little")
    desc = np.ascontiguousarray(np.concatenate([a[: n - n // 2], noisy[: n // 2]]), np.uint8)[:n]
    return voc, desc


def reference_computebow(R, voc, desc, binary, scoring, weighting, tmpdir):
    """(BowVector words, values, FeatureVector nodes, features, KeyFrame / copied vocabulary agree, ORBVocabulary is the adaptor class)"""
    path = os.path.join(tmpdir, "cbow_%d_%d.%s" % (os.getpid(), len(voc.node_desc), "bin" if binary else "txt"))
    if binary:
        write_binary_voc(voc, path, scoring, weighting)
    else:
        voc.save_text(path, scoring, weighting)
    n = len(desc)
    bw, bv = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
    fn, ff = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    nfv, eq, adp = C.c_int(0), C.c_int(0), C.c_int(0)
    R.ref_compute_bow.argtypes = [C.c_char_p, C.c_int] + [C.c_void_p, C.c_int] + [C.c_void_p] * 7
    c = R.ref_compute_bow(path.encode(), 1 if binary else 0, p(desc), n, p(bw), p(bv), p(fn), p(ff), C.byref(nfv), C.byref(eq), C.byref(adp))
    os.remove(path)
    assert c >= 0, "the reference did not load %s" % path
    return bw[:c].copy(), bv[:c].copy(), fn[:nfv.value].copy(), ff[:nfv.value].copy(), eq.value, adp.value


def gen_computebow(S, out):
    import tempfile
    R = ref_frame_lib()
    VM = _util._load("plslam_amd_vocab", os.path.join(ROOT, "pl-slam_amd", "vocab.py"))
    g = {}
    for seed, k, Lv, stop, idf, n, binary, sc, wt in COMPUTEBOW_CASES:
        voc, desc = computebow_inputs(S, VM, seed, k, Lv, stop, idf, n)
        bw, bv, fn, ff, eq, adp = reference_computebow(R, voc, desc, binary, sc, wt, tempfile.gettempdir())
        assert eq == 1 and adp == 0
        g["w_%d" % seed], g["v_%d" % seed], g["fn_%d" % seed], g["ff_%d" % seed] = bw, bv, fn, ff
        print("ComputeBoW", seed, "words", len(bw), "listed features", len(ff), "of", n)
    np.savez_compressed(os.path.join(out, "ref_computebow.npz"), **g)


# ---------------------------------------------------------------------------------------------------------------
# LSDmatcher::FrameBFMatchNew / SearchForTriangulationNew (src/LSDmatcher.cpp:488-625, 780-858) -> ref_lsdmatcher_new.npz.
# Two views of n random 3-D segments; set 2's rows are set 1's permuted (make_descriptor_sets), its segments shortened / shifted along
# the line so that the overlap score spreads around the 0.8 gate.  The fundamental matrices come out of the reference's ComputeF12
# through the harness and are part of the golden file: the oracle and the library take them as inputs.
# ---------------------------------------------------------------------------------------------------------------
LNEW_CASES = [(11, 300, 280, 0.06, 0.7), (12, 64, 80, 0.20, 0.9), (14, 500, 7, 0.10, 0.8), (15, 37, 411, 0.27, 0.75), (16, 2, 2, 0.05, 0.7),
              (17, 3, 1, 0.05, 0.7), (18, 2000, 2000, 0.12, 0.8)]       # seed, n1, n2, bit-flip probability, nnratio
LNEW_K = np.array([517.3, 0, 318.6, 0, 516.5, 255.3, 0, 0, 1], np.float32)
LNEW_BF = [(50.0, 1.0), (80.0, 1.0), (50.0, 0.0), (50.0, -3.5)]     # TH, scale of F21 (0: the degenerate matrix, every w = 0)


def lnew_inputs(S, seed, n1, n2, flip):
    n = max(n1, n2)
    a, b, perm = S.make_descriptor_sets(seed, n, flip)
    rng = S.SplitMix64(7000 + seed)
    u = lambda k: rng.uniform(k)
    ay, ax = 0.04 + 0.1 * u(1)[0], 0.03 * (u(1)[0] - 0.5)
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    R2 = (Ry @ Rx).astype(np.float32)
    t2 = np.array([-0.35, 0.02, 0.06], np.float32)
    pose1 = np.r_[np.eye(3, dtype=np.float32).ravel(), np.zeros(3, np.float32)].astype(np.float32)
    pose2 = np.r_[R2.ravel(), t2].astype(np.float32)
    mid = np.c_[3.0 * (u(n) - 0.5), 2.0 * (u(n) - 0.5), 3.0 + 3.0 * u(n)]
    d = np.c_[u(n) - 0.5, u(n) - 0.5, 0.4 * (u(n) - 0.5)]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hl = 0.2 + 0.4 * u(n)
    P, Q = mid - d * hl[:, None], mid + d * hl[:, None]
    K = LNEW_K.reshape(3, 3).astype(np.float64)

    def proj(X, R, t):
        x = (K @ (R.astype(np.float64) @ X.T + t.astype(np.float64)[:, None])).T
        return x[:, :2] / x[:, 2:3]
    s1 = np.c_[proj(P, np.eye(3), np.zeros(3)), proj(Q, np.eye(3), np.zeros(3))]
    p2, q2 = proj(P, R2, t2)[perm], proj(Q, R2, t2)[perm]
    cut0, cut1 = u(n) < 0.6, u(n) < 0.6
    t0, t1 = 0.45 * u(n) * cut0, 1.0 - 0.45 * u(n) * cut1
    sp, ep = p2 + t0[:, None] * (q2 - p2), p2 + t1[:, None] * (q2 - p2)
    swap = u(n) < 0.3
    sp2, ep2 = np.where(swap[:, None], ep, sp), np.where(swap[:, None], sp, ep)
    perp = np.c_[-(q2 - p2)[:, 1], (q2 - p2)[:, 0]]
    perp /= np.maximum(np.linalg.norm(perp, axis=1, keepdims=True), 1e-9)
    off = 1.5 * (u(n) - 0.5)
    s2 = np.c_[sp2 + off[:, None] * perp, ep2 + off[:, None] * perp]
    seg1, seg2 = np.ascontiguousarray(s1[:n1], np.float32), np.ascontiguousarray(s2[:n2], np.float32)

    def funcs(seg):
        sg = seg.astype(np.float64)
        l = np.cross(np.c_[sg[:, 0], sg[:, 1], np.ones(len(sg))], np.c_[sg[:, 2], sg[:, 3], np.ones(len(sg))])
        return np.ascontiguousarray(l / np.maximum(np.hypot(l[:, 0], l[:, 1]), 1e-12)[:, None])
    ml1, ml2 = (u(n) < 0.1).astype(np.uint8)[:n1], (u(n) < 0.1).astype(np.uint8)[:n2]
    return dict(d1=np.ascontiguousarray(a[:n1]), d2=np.ascontiguousarray(b[:n2]), seg1=seg1, seg2=seg2, func1=funcs(seg1), func2=funcs(seg2),
                pose1=pose1, pose2=pose2, K=LNEW_K.copy(), ml1=np.ascontiguousarray(ml1), ml2=np.ascontiguousarray(ml2))


def reference_ltri_new(R, x, ratio, is_double):
    n1, n2 = len(x["d1"]), len(x["d2"])
    m, F21, F12 = np.full(max(n1, 1), -1, np.int32), np.zeros(9, np.float32), np.zeros(9, np.float32)
    c = R.ref_line_search_for_triangulation_new(p(x["d1"]), n1, p(x["d2"]), n2, p(x["seg1"]), p(x["seg2"]), p(x["func1"]), p(x["func2"]),
                                                p(x["pose1"]), p(x["pose2"]), p(x["K"]), p(x["K"]), p(x["ml1"]), p(x["ml2"]), ratio,
                                                1 if is_double else 0, p(m), p(F21), p(F12))
    return c, m[:n1], F21, F12


def reference_lbf_new(R, x, F, th, ratio):
    n1 = len(x["d1"])
    m = np.full(max(n1, 1), -1, np.int32)
    F = np.ascontiguousarray(F, np.float32)
    R.ref_line_bfmatch_new(p(x["d1"]), n1, p(x["d2"]), len(x["d2"]), p(x["seg1"]), p(x["seg2"]), p(x["func2"]), p(F), th, ratio, p(m))
    return m[:n1]


def gen_lsdmatcher_new(S, out):
    R = ref_lsdmatcher_lib()
    g = {}
    for seed, n1, n2, flip, ratio in LNEW_CASES:
        x = lnew_inputs(S, seed, n1, n2, flip)
        for dbl in (0, 1):
            c, m, F21, F12 = reference_ltri_new(R, x, ratio, dbl)
            g["tri_%d_%d_n" % (seed, dbl)], g["tri_%d_%d_m" % (seed, dbl)] = c, m
        g["F21_%d" % seed], g["F12_%d" % seed] = F21, F12
        for k, (th, sc) in enumerate(LNEW_BF):
            g["bf_%d_%d" % (seed, k)] = reference_lbf_new(R, x, F21 * np.float32(sc), th, ratio)
    np.savez_compressed(os.path.join(out, "ref_lsdmatcher_new.npz"), **g)
    print("lsdmatcher New:", {k: int(v) for k, v in g.items() if k.endswith("_n")},
          {k: int((v >= 0).sum()) for k, v in g.items() if k.startswith("bf_")})


def main():
    S = _util.synth()
    if "lnew" in sys.argv[1:]:   # only the file of LSDmatcher's New variants (the other files are not touched)
        return gen_lsdmatcher_new(S, os.path.join(ROOT, "tests", "golden"))
    VM = _util._load("plslam_amd_vocab", os.path.join(ROOT, "pl-slam_amd", "vocab.py"))
    L = ref_lib()
    out = os.path.join(ROOT, "tests", "golden")
    for name, vseed, k, Lv, stop, dseed, n, up in CASES:
        voc, desc = make_case(S, VM, vseed, k, Lv, stop, dseed, n)
        word, weight, node, fnode, bw, bv = reference_transform(L, voc, desc, up)
        # Hamming distances of the reference on a sample of pairs
        ia, ib = np.arange(0, n - 1, 7), np.arange(1, n, 7)[: len(np.arange(0, n - 1, 7))]
        dist = np.array([L.ref_forb_distance(p(desc[i]), p(desc[j])) for i, j in zip(ia, ib)], np.int32)
        np.savez_compressed(os.path.join(out, "ref_dbow2_%s.npz" % name), vseed=vseed, k=k, L=Lv, stop=stop, dseed=dseed, n=n,
                            levelsup=up, word=word, weight=weight, node=node, feat_node=fnode, bow_word=bw, bow_value=bv,
                            pair_a=ia.astype(np.int32), pair_b=ib.astype(np.int32), pair_dist=dist)
        print(name, "words", len(bw), "stopped", int((fnode < 0).sum()), "nodes", len(np.unique(node)))
    gen_vocab_files(S, VM, L, out)
    gen_orb(S, out)
    gen_line_grid(S, out)
    gen_lines(S, out)
    gen_matchers(S, out)
    gen_keyframe_searches(S, out)
    gen_lsdmatcher(S, out)
    gen_lsdmatcher_new(S, out)
    gen_distinctive(S, out)
    gen_framegrid(S, out)
    gen_frustum(S, out)
    gen_track(S, out)
    gen_computebow(S, out)


if __name__ == "__main__":
    main()
