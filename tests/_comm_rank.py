"""One rank of a world-size-2 plh_gather_records on ONE device (helper process of tests/test_comm.py):
    python _comm_rank.py RANK UNIQUE_ID_FILE OUT_FILE
Both ranks create an RCCL communicator on device 0 from the same unique id, gather four record blocks to every rank and to
rank 1, and write what they received."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _util  # noqa: E402

rank, idfile, outfile = int(sys.argv[1]), sys.argv[2], sys.argv[3]
import torch  # noqa: E402
P, S = _util.plslam(), _util.synth()
P.load()
if rank == 0:
    uid = P.Comm.unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 60:
            sys.exit(3)
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
try:
    c = P.Comm(uid, rank, 2, device=0)
except P.PlhError as e:
    open(outfile, "w").write("CREATE_FAILED %s" % e)
    sys.exit(0)


def blocks(seed):
    rng = S.SplitMix64(seed)
    return [rng.randint(n, 0, 256).astype(np.uint8) for n in (4, 28 * 1006, 32 * 1006, 68 * 201)]


s = torch.cuda.Stream()
res = {}
for root in (-1, 1):
    send = [torch.from_numpy(b).cuda() for b in blocks(100 * (root + 2) + rank)]
    receives = root < 0 or root == rank
    recv = [torch.zeros((2,) + tuple(b.shape), dtype=torch.uint8, device="cuda") if receives else None for b in send]
    torch.cuda.synchronize()
    c.gather(list(zip(send, recv)), root=root, stream=s.cuda_stream)
    s.synchronize()
    if receives:
        for k, r in enumerate(recv):
            res["root%d_block%d" % (root, k)] = r.cpu().numpy()
c.close()
np.savez(outfile + ".npz", **res)
open(outfile, "w").write("OK")
