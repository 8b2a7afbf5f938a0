"""Flat DBoW2-style vocabulary tree for the GPU BoW transform (plh_bow_transform_batch_dev).

The reference's `Vocabulary/ORBvoc.bin` is missing from the mount (.MISSING_LARGE_BLOBS), so parity tests and
bench.py use a SYNTHETIC k-ary tree (default k = 10, L = 6 like ORBvoc): node descriptors are derived
hierarchically (each child flips a random ~1/8 of its parent's bits), weights are 1.0.  `from_arrays`
accepts a real tree in the same flat form (children of a node contiguous, root = node 0).
"""
import numpy as np


class Vocabulary:
    def __init__(self, node_desc, child_start, child_count, word_id, weight, k, L):
        self.node_desc = np.ascontiguousarray(node_desc, np.uint8)
        self.child_start = np.ascontiguousarray(child_start, np.int32)
        self.child_count = np.ascontiguousarray(child_count, np.int32)
        self.word_id = np.ascontiguousarray(word_id, np.int32)
        self.weight64 = np.ascontiguousarray(weight, np.float64)        # WordValue is double in DBoW2
        self.weight = self.weight64.astype(np.float32)                  # the descent only tests `w > 0`
        self.weight[(self.weight64 > 0) & ~(self.weight > 0)] = np.finfo(np.float32).tiny
        self.k, self.L = k, L
        self._dev = None

    @property
    def n_nodes(self):
        return len(self.child_start)

    @staticmethod
    def synthetic(seed=102, k=10, L=6, synth=None, stop_fraction=0.0, idf=False):
        """Complete k-ary tree of depth L in BFS numbering (children of node n: k*n+1 .. k*n+k).  idf=True gives the words
        idf-like weights with full double mantissas (ORBvoc's are log(N / n_i)) instead of 1.0."""
        if synth is None:
            raise ValueError("pass the synth module (SplitMix64 generator)")
        rng = synth.SplitMix64(seed)
        n_nodes = (k ** (L + 1) - 1) // (k - 1)
        first_leaf = (k ** L - 1) // (k - 1)
        desc = np.zeros((n_nodes, 32), np.uint8)
        lo, hi = 1, 1 + k
        desc[lo:hi] = rng.u64(k * 4).view(np.uint8).reshape(k, 32)
        for _ in range(2, L + 1):
            nlo, nhi = hi, hi + (hi - lo) * k
            parents = np.repeat(desc[lo:hi], k, axis=0)
            flips = (rng.uniform((nhi - nlo) * 256) < 0.125).reshape(nhi - nlo, 256)
            desc[nlo:nhi] = parents ^ np.packbits(flips, axis=1, bitorder="little")
            lo, hi = nlo, nhi
        idx = np.arange(n_nodes, dtype=np.int64)
        child_start = np.where(idx < first_leaf, k * idx + 1, 0).astype(np.int32)
        child_count = np.where(idx < first_leaf, k, 0).astype(np.int32)
        word_id = np.where(idx >= first_leaf, idx - first_leaf, -1).astype(np.int32)
        weight = np.ones(n_nodes, np.float64)
        if idf:
            weight = np.log(1.0 + 1.0 / (1e-3 + rng.uniform(n_nodes))) * 3.0
        if stop_fraction > 0:
            stopped = rng.uniform(n_nodes) < stop_fraction
            weight[stopped & (idx >= first_leaf)] = 0.0
        return Vocabulary(desc, child_start, child_count, word_id, weight, k, L)

    # ---- DBoW2 text format (the reference's TemplatedVocabulary::loadFromTextFile / saveToTextFile,
    #      Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1350-1462; ORBvoc.txt of ORB-SLAM2 is such a file):
    #      line 1 "k L scoring weighting", then one line per non-root node in id order:
    #      "parent isLeaf d0 .. d31 weight".  Node ids are line numbers, a node's children are the lines naming it as
    #      parent in file order, word ids count the leaves in file order.
    def save_text(self, path, scoring=0, weighting=0):
        """Write the tree so that the reference's loader rebuilds it with identical node and word ids.  Requires the
        flat layout to be in id order already (children contiguous and ascending -- true for `synthetic`)."""
        n = self.n_nodes
        parent = np.zeros(n, np.int64)
        for p in range(n):
            c0, cc = int(self.child_start[p]), int(self.child_count[p])
            parent[c0:c0 + cc] = p
        lines = ["%d %d %d %d" % (self.k, self.L, scoring, weighting)]
        for i in range(1, n):
            lines.append("%d %d %s %r" % (parent[i], 1 if self.child_count[i] == 0 else 0,
                                          " ".join(str(int(b)) for b in self.node_desc[i]), float(self.weight64[i])))
        with open(path, "w") as f:
            f.write("\n".join(lines))   # no trailing newline: the reference's eof loop would read one node too many

    @staticmethod
    def load_text(path):
        """Read a DBoW2 text vocabulary (e.g. ORBvoc.txt) into the flat form, ids as the reference assigns them."""
        with open(path) as f:
            k, L, _scoring, _weighting = (int(v) for v in f.readline().split()[:4])
            rows = [ln.split() for ln in f if ln.strip()]
        n = len(rows) + 1
        parent = np.zeros(n, np.int64)
        leaf = np.zeros(n, bool)
        desc = np.zeros((n, 32), np.uint8)
        weight = np.zeros(n, np.float64)
        for i, r in enumerate(rows, start=1):
            parent[i], leaf[i] = int(r[0]), int(r[1]) > 0
            desc[i] = [int(v) for v in r[2:34]]
            weight[i] = float(r[34])
        # children of a node in file order; the flat form wants them contiguous: renumber breadth first when they are not
        kids = [[] for _ in range(n)]
        for i in range(1, n):
            kids[parent[i]].append(i)
        contiguous = all(len(c) == 0 or c == list(range(c[0], c[0] + len(c))) for c in kids)
        if not contiguous:
            raise ValueError("vocabulary file whose children are not contiguous in id order is not supported yet")
        child_start = np.array([c[0] if c else 0 for c in kids], np.int32)
        child_count = np.array([len(c) for c in kids], np.int32)
        word_id = np.full(n, -1, np.int32)
        word_id[leaf] = np.arange(int(leaf.sum()), dtype=np.int32)
        return Vocabulary(desc, child_start, child_count, word_id, weight, k, L)

    def tree_arrays(self):
        """(parent, is_leaf) in node order, as plh_vocab_create takes them."""
        n = self.n_nodes
        parent = np.zeros(n, np.int32)
        for p in range(n):
            c0, cc = int(self.child_start[p]), int(self.child_count[p])
            parent[c0:c0 + cc] = p
        parent[0] = -1
        return parent, (self.child_count == 0).astype(np.uint8)

    def word_weight(self):
        """weight (f64) by word id."""
        ww = np.zeros(max(1, int((self.word_id >= 0).sum())), np.float64)
        ww[self.word_id[self.word_id >= 0]] = self.weight64[self.word_id >= 0]
        return ww

    def device_arrays(self, dev):
        """Upload once per device helper (`plslam_amd._Dev`)."""
        if self._dev is None:
            self._dev = tuple(dev.put(a) for a in (self.node_desc, self.child_start, self.child_count, self.word_id, self.weight))
        return self._dev
