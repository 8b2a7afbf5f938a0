"""Round 6: the counter build (-DPLH_GROW_PROF) of k_lsd_grow_mw16 returns other segments than the product build on 10 of the soak's 1024
frames.  Is the set of frames the same from run to run (code generation) or not (a race of the multi-wavefront protocol that the
counter build's timing exposes)?  Product build, automatic policy = reference (it equals the oracle in the soak); then the counter
build three times at 1024 x 8 (k_lsd_grow_mw16) and once as 4 x 256 x 8 (the roomy k_lsd_grow_mw)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import _util, torch
import test_soak_gpu as T
P, S = _util.plslam(), _util.synth()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = T.soak_frames(S, 480, 640, N)
prof = os.path.join(ROOT, "pl-slam_amd", sys.argv[2] if len(sys.argv) > 2 else "libplslam_hip_prof.so")
print("library under test:", os.path.basename(prof), flush=True)


def segs(fr, waves, lib):
    return [g[3] for g in T._gpu_lines(P, fr, waves, 1, lib=lib)]


ref = segs(frames, -1, None)
ref0 = segs(frames, 0, None)
print("product: automatic == one wavefront per frame on %d of %d frames" % (sum(len(a) == len(b) and (a == b).all() for a, b in zip(ref, ref0)), N), flush=True)
for run in range(3):
    got = segs(frames, -1, prof)
    bad = [i for i, (a, b) in enumerate(zip(got, ref)) if not (len(a) == len(b) and (a == b).all())]
    print("build under test, %d x 8 (k_lsd_grow_mw16), run %d: %d frames differ: %s" % (N, run, len(bad), bad), flush=True)
    for i in bad[:2]:
        a, b = got[i], ref[i]
        if len(a) == len(b):
            d = np.where((a != b).any(axis=1))[0]
            print("   frame %d: %d of %d segments differ; first: got %s, product %s" % (i, len(d), len(a), a[d[0]], b[d[0]]), flush=True)
        else:
            print("   frame %d: %d segments, product %d" % (i, len(a), len(b)), flush=True)
bad = []
for k in range(0, N, 256):
    got = segs(frames[k:k + 256], -1, prof)
    bad += [k + i for i, (a, b) in enumerate(zip(got, ref[k:k + 256])) if not (len(a) == len(b) and (a == b).all())]
print("build under test, %d x (256 x 8) (k_lsd_grow_mw, the roomy compile): %d frames differ: %s" % (N // 256, len(bad), bad), flush=True)
got = segs(frames, 0, prof)
bad = [i for i, (a, b) in enumerate(zip(got, ref)) if not (len(a) == len(b) and (a == b).all())]
print("build under test, one wavefront per frame (k_lsd_grow): %d frames differ: %s" % (len(bad), bad), flush=True)
for rep in range(3):
    got = segs(frames, -1, None)
    bad = [i for i, (a, b) in enumerate(zip(got, ref)) if not (len(a) == len(b) and (a == b).all())]
    print("product build again, %d x 8 (k_lsd_grow_mw16), repetition %d: %d frames differ" % (N, rep, len(bad)), flush=True)
