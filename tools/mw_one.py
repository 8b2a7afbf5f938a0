#!/usr/bin/env python3
"""A few line extractions of a small batch with a fixed wavefront count (for rocprofv3 runs): mw_one.py BATCH WAVES [REPS]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util, torch
P, S = _util.plslam(), _util.synth()
B, W = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
K = [517.306408, 516.469215, 318.643040, 255.313989]; D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
frames = S.make_frames(2, B, 480, 640, unique=min(B, 32))
d = torch.from_numpy(frames).cuda()
le = P.LINEextractor(1, 1.2, 200, 0.0, rows=480, cols=640, max_batch=B, K=K, D=D)
le.set_grow_waves(W)
cap = le.capacity
bufs = [torch.zeros((B, cap, 17), dtype=torch.float32, device="cuda"), torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"),
        torch.zeros((B, cap, 3), dtype=torch.float64, device="cuda"), torch.zeros((B,), dtype=torch.int32, device="cuda")]
for _ in range(reps):
    le.extract_batch_dev(d, B, 480 * 640, *bufs, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("done", int(bufs[3].sum()))
