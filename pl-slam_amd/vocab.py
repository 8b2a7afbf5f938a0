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
        self.weight = np.ascontiguousarray(weight, np.float32)
        self.k, self.L = k, L
        self._dev = None

    @property
    def n_nodes(self):
        return len(self.child_start)

    @staticmethod
    def synthetic(seed=102, k=10, L=6, synth=None, stop_fraction=0.0):
        """Complete k-ary tree of depth L in BFS numbering (children of node n: k*n+1 .. k*n+k)."""
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
        weight = np.ones(n_nodes, np.float32)
        if stop_fraction > 0:
            stopped = rng.uniform(n_nodes) < stop_fraction
            weight[stopped & (idx >= first_leaf)] = 0.0
        return Vocabulary(desc, child_start, child_count, word_id, weight, k, L)

    def device_arrays(self, dev):
        """Upload once per device helper (`plslam_amd._Dev`)."""
        if self._dev is None:
            self._dev = tuple(dev.put(a) for a in (self.node_desc, self.child_start, self.child_count, self.word_id, self.weight))
        return self._dev
