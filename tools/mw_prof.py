#!/usr/bin/env python3
"""Where the wavefronts of k_lsd_grow_mw spend their time (debug build with -DPLH_GROW_PROF):

    hipcc <flags of __graft_entry__.HIPCC_FLAGS> -DPLH_GROW_PROF -o pl-slam_amd/libplslam_hip_prof.so pl-slam_amd/csrc/*.hip
    PLSLAM_HIP_LIB=pl-slam_amd/libplslam_hip_prof.so python tools/mw_prof.py [--batch 1] [--waves 2,4,8,16]

s_memtime cycle totals over all wavefronts, divided by frames x wavefronts (100 MHz constant clock: 1 tick = 10 ns)."""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util, torch
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--waves", default="0,2,4,8,16")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
P, S = _util.plslam(), _util.synth()
lib = P.load()
B = a.batch
frames = S.make_frames(2, B, 480, 640, unique=min(B, 32))
d = torch.from_numpy(frames).cuda()
K = [517.306408, 516.469215, 318.643040, 255.313989]; D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
le = P.LINEextractor(1, 1.2, 200, 0.0, rows=480, cols=640, max_batch=B, device=0, K=K, D=D)
cap = le.capacity
bufs = [torch.zeros((B, cap, 17), dtype=torch.float32, device="cuda"), torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"),
        torch.zeros((B, cap, 3), dtype=torch.float64, device="cuda"), torch.zeros((B,), dtype=torch.int32, device="cuda")]
s = torch.cuda.current_stream().cuda_stream
out = (C.c_ulonglong * 40)()
N = ["total", "rect px", "grow", "grow:load", "grow:resolve", "rect", "refine", "#reduce", "#steps", "#accepted", "#cands", "#grow calls",
     "#passes", "#mispred", "#rect", "#refine", "#txn", "#unused", "#rerun", "wait turn", "scan", "wait seed", "run", "commit"]
for W in [int(x) for x in a.waves.split(",")]:
    le.set_grow_waves(W)
    le.extract_batch_dev(d, B, 480 * 640, *bufs, s); torch.cuda.synchronize()
    lib.plh_debug_grow_prof(out, 1)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        le.extract_batch_dev(d, B, 480 * 640, *bufs, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    lib.plh_debug_grow_prof(out, 0)
    nw = B * a.reps * max(W, 1)
    nf = B * a.reps
    cyc = lambda k: out[k] / nw / 2400.0   # microseconds per wavefront (s_memtime counts shader clocks here, 2.4 GHz)
    cnt = lambda k: out[k] / nf           # per frame
    print("   grow %.0f rect %.0f refine %.0f validate+unmark %.0f post %.0f (us per wavefront); rects %.0f refines %.0f predicted-unused %.0f; drain re-runs: accepted pixel used %.0f, assumed pixel free %.0f (predicted-unused %.0f) per frame" %
          (cyc(2), cyc(5), cyc(6), cyc(26), cyc(27), cnt(14), cnt(15), cnt(28), cnt(29), cnt(30), cnt(31)))
    print("   drain: %.0f posts committed in inline batches, %.0f one by one; re-runs took %.0f us per wavefront" % (cnt(34), cnt(35), cyc(10)))
    print("waves %2d: extract %.2f ms | per wavefront (us): total %.0f  run %.0f  idle %.0f  drain %.0f  scan %.0f | per frame: txn %.0f unused %.0f "
          "rerun own %.0f drain %.0f  drain sessions %.0f  steps %.0f accepted %.0f grow-calls %.0f" %
          (W, dt * 1e3, cyc(0), cyc(22) if W else cyc(2) + cyc(5) + cyc(6), cyc(19), cyc(23), cyc(20), cnt(16), cnt(17), cnt(18), cnt(24), cnt(25),
           cnt(8), cnt(9), cnt(11)), flush=True)
