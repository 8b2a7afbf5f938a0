#!/usr/bin/env python3
"""Throughput of the two halves of the front end on their own (same resident batch, same sub-batch pipelining as bench.py):
line chains only, ORB chains only, both.  Development aid: tells whether the halves overlap or add up."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util
import torch
P, S = _util.plslam(), _util.synth()
V = _util._load("plslam_amd_vocab", os.path.join(ROOT, "pl-slam_amd", "vocab.py"))
PL = _util._load("plslam_amd_pipeline", os.path.join(ROOT, "pl-slam_amd", "pipeline.py"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
frames = S.make_frames(2, B, 480, 640, unique=32)
d = torch.from_numpy(frames).to(dev)
voc = V.Vocabulary.synthetic(102, k=10, L=6, synth=S, idf=True)
K = [517.306408, 516.469215, 318.643040, 255.313989]
D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
fe = PL.FrontEndPipelined(P, voc, B, 480, 640, 1000, 8, 200, 0.0, K, D, 0, nsplit=ns)
Bp = B // ns
main = torch.cuda.current_stream(dev)
def run(which, steps=8):
    def step():
        fe.ev_start.record(main)
        for k, p in enumerate(fe.parts):
            if which in ("line", "both"): p.enqueue_line(d[k * Bp:(k + 1) * Bp], main, fe.ev_start)
        for k, p in enumerate(fe.parts):
            if which in ("orb", "both"): p.enqueue_orb(d[k * Bp:(k + 1) * Bp], main, fe.ev_start)
    for _ in range(2): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print("%-5s %8.2f ms per %d frames = %8.0f frames/s" % (which, dt * 1e3, B, B / dt))
for w in ("line", "orb", "both"):
    run(w)
