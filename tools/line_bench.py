#!/usr/bin/env python3
"""Quick stage timing of the line extractor + matchers on a batch (development aid; bench.py is the contract)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util
import torch
P, S = _util.plslam(), _util.synth()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rows, cols = 480, 640
frames = S.make_frames(2, B, rows, cols, unique=16)
dev = torch.device("cuda", 0)
d_img = torch.from_numpy(frames).to(dev)
ex = P.LINEextractor(1, 1.2, 200, 0.0, rows=rows, cols=cols, max_batch=B)
cap = ex.capacity
d_kl = torch.zeros((B, cap, 17), dtype=torch.float32, device=dev)
d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
d_fn = torch.zeros((B, cap, 3), dtype=torch.float64, device=dev)
d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    ex.extract_batch_dev(d_img, B, rows * cols, d_kl, d_desc, d_fn, d_n, st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("line extract batch %d: %.2f ms -> %.0f frames/s (mean lines %.1f)" % (B, dt * 1e3, B / dt, d_n.float().mean().item()))
