"""N > 1 path on CPU: world_size 2, gloo.  Each rank extracts ORB features AND lines for its contiguous shard of a frame
batch (kernel sources under hipemu -- there is no GPU here), all SEVEN fixed-stride record blocks of plh_frontend_gather
(n, kps, desc, nl, kl, ldesc, lfn: PLH_FRONTEND_GATHERED) are all_gather'ed exactly as bench.py does over RCCL, and rank 0
checks the gathered batch -- keypoints, rBRIEF, KeyLines, LBD, line equations -- against the oracle frame by frame."""
import os
import subprocess
import sys

import _util

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import _util
P, S, O = _util.plslam(), _util.synth(), _util.oracle()
D = _util._load("plslam_amd_dist", os.path.join(_util.ROOT, "pl-slam_amd", "dist.py"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
TOTAL, rows, cols = 6, 100, 128
frames = S.make_frames(900, TOTAL, rows, cols, unique=TOTAL)
lo, hi = D.shard_range(TOTAL, rank, world)
assert hi - lo == TOTAL // world
ex = P.ORBextractor(150, 1.2, 2, 20, 7, rows=rows, cols=cols, max_batch=hi - lo, lib=_util.EMU_LIB)
kps, desc, n = ex.extract_batch(frames[lo:hi])
lx = P.LINEextractor(1, 1.2, 30, 0.0, rows=rows, cols=cols, max_batch=1, lib=_util.EMU_LIB)
lcap = lx.capacity
nl = np.zeros(hi - lo, np.int32)
kl = np.zeros((hi - lo, lcap), P.KL_DTYPE)
ldesc = np.zeros((hi - lo, lcap, 32), np.uint8)
lfn = np.zeros((hi - lo, lcap, 3), np.float64)
for i in range(hi - lo):   # the line records of this rank's frames, at the fixed stride of the gather (plh_frontend_records)
    k1, d1, f1 = lx(frames[lo + i])
    nl[i] = len(k1); kl[i, :len(k1)] = k1; ldesc[i, :len(k1)] = d1; lfn[i, :len(k1)] = f1
local = {"n": torch.from_numpy(n), "kps": torch.from_numpy(kps.view(np.uint8).reshape(hi - lo, ex.capacity, 28).copy()),
         "desc": torch.from_numpy(desc), "nl": torch.from_numpy(nl),
         "kl": torch.from_numpy(kl.view(np.uint8).reshape(hi - lo, lcap, 68).copy()), "ldesc": torch.from_numpy(ldesc),
         "lfn": torch.from_numpy(lfn)}
assert tuple(local) == ("n", "kps", "desc", "nl", "kl", "ldesc", "lfn") and len(local) == P.FRONTEND_GATHERED
g = D.all_gather_records(local, world, dist)
if rank == 0:
    ref = O.OrbOracle(150, 1.2, 2, 20, 7)
    gn = g["n"].numpy()
    gk = g["kps"].numpy().reshape(TOTAL, ex.capacity, 28).copy().view(P.KP_DTYPE).reshape(TOTAL, ex.capacity)
    gd = g["desc"].numpy()
    for b in range(TOTAL):
        rk, rd = ref.extract(frames[b])
        assert gn[b] == len(rk), (b, gn[b], len(rk))
        for f in rk.dtype.names:
            assert (gk[b, :gn[b]][f] == rk[f]).all(), (b, f)
        assert (gd[b, :gn[b]] == rd).all(), b
    gnl = g["nl"].numpy()
    gkl = g["kl"].numpy().reshape(TOTAL, lcap, 68).copy().view(P.KL_DTYPE).reshape(TOTAL, lcap)
    gld, gfn = g["ldesc"].numpy(), g["lfn"].numpy()
    for b in range(TOTAL):
        rk, rd, rf = O.line_extract(frames[b], 30, 0.0)
        assert gnl[b] == len(rk) and len(rk) > 0, (b, gnl[b], len(rk))
        for f in rk.dtype.names:
            assert (gkl[b, :gnl[b]][f] == rk[f]).all(), (b, f)
        assert (gld[b, :gnl[b]] == rd).all() and (gfn[b, :gnl[b]] == rf).all(), b
    print("DIST_OK", gn.tolist(), gnl.tolist())
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_range():
    D = _util._load("plslam_amd_dist", os.path.join(_util.ROOT, "pl-slam_amd", "dist.py"))
    for total, world in [(4096, 8), (10, 3), (7, 8), (1, 1)]:
        cover = []
        for r in range(world):
            lo, hi = D.shard_range(total, r, world)
            cover += list(range(lo, hi))
            assert 0 <= hi - lo <= total // world + 1
        assert cover == list(range(total))


def test_two_rank_gloo_shard_and_gather(emu_lib, oracle, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), _util.ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0], outs[0]
