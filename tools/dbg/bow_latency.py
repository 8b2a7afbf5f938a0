"""Wall time per call of plh_orb_search_by_bow_resident (one KeyFrame / Frame pair, both resident) against the feature count and the
number of vocabulary nodes -- what TrackReferenceKeyFrame pays in the library below the adaptor.  usage: bow_latency.py [reps]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import _util
import test_frame_search as TF
P, S = _util.plslam(), _util.synth()
H = P.load()
V, I, F = C.c_void_p, C.c_int, C.c_float
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = lambda a: a.ctypes.data_as(V)
H.plh_frame_points_create.argtypes = [V, V, I, V, I, V]
H.plh_frame_points_set_nodes.argtypes = [V, V]
H.plh_orb_search_by_bow_resident.argtypes = [V, V, V, I, F, I, V, V]
gp = TF._gp(P)
for n, nodes in ((1000, 100), (2000, 100), (2000, 1), (2000, 1000)):
    f1, f2, _, _ = TF.make_frame_pair(P, S, 5, n, nl=50, move=3.0)
    n1, n2 = len(f1["kps"]), len(f2["kps"])
    R = []
    for f in (f1, f2):
        h = V()
        P._check(H, H.plh_frame_points_create(p(f["kps"]), p(f["desc"]), len(f["kps"]), C.byref(gp), 0, C.byref(h)), "create")
        R.append(h)
    rng = S.SplitMix64(77)
    node1, node2 = rng.randint(n1, 0, nodes).astype(np.int32), rng.randint(n2, 0, nodes).astype(np.int32)
    valid1 = np.ones(n1, np.uint8)
    H.plh_frame_points_set_nodes(R[0], p(node1)); H.plh_frame_points_set_nodes(R[1], p(node2))
    m, c = np.zeros(n2, np.int32), C.c_int(0)
    for _ in range(5):
        H.plh_orb_search_by_bow_resident(R[0], p(valid1), R[1], 50, 0.7, 1, p(m), C.byref(c))
    t0 = time.perf_counter()
    for _ in range(reps):
        H.plh_orb_search_by_bow_resident(R[0], p(valid1), R[1], 50, 0.7, 1, p(m), C.byref(c))
    dt = (time.perf_counter() - t0) / reps
    print("%4d x %4d features, %4d nodes: %7.1f us per call (%d matches)" % (n1, n2, nodes, dt * 1e6, c.value), flush=True)
    for h in R:
        H.plh_frame_points_destroy(h)
