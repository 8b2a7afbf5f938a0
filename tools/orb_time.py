import sys, os, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "tests"))
import _util, torch, numpy as np
P, S = _util.plslam(), _util.synth()
B=1024
frames = S.make_frames(2, B, 480, 640, unique=32)
d = torch.from_numpy(frames).cuda()
orb = P.ORBextractor(1000, 1.2, 8, 20, 7, rows=480, cols=640, max_batch=B)
cap = orb.capacity
kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); n = torch.zeros((B,), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
orb.extract_batch_dev(d, B, 480*640, kps, desc, n, s); torch.cuda.synchronize()
orb.set_profiling(True)
for _ in range(3): orb.extract_batch_dev(d, B, 480*640, kps, desc, n, s)
torch.cuda.synchronize()
print(os.environ.get("PLSLAM_HIP_LIB","default"), [round(orb.kernel_ms(k)[0]/3,3) for k in range(4)])
