#!/usr/bin/env python3
"""bench.py -- throughput of the PL-SLAM front-end hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one pass of the hot path over one batch of synthetic frames that is already resident in HBM.
Workload (BASELINE.json configs[1]/[2]): 640x480 mono frames, 8-level pyramid, 1000 ORB features
(TUM1.yaml parameters), batch of `--batch` frames per GPU (weak scaling: every rank processes its own batch;
frames are independent), followed -- for N > 1 -- by one RCCL all_gather of the fixed-stride
keypoint/descriptor records, as the north star asks.

One JSON line on stdout (rank 0): metric/value/unit ... plus
  "roofline":     dominant kernel, algorithmic bytes per launch / its mean duration measured live with HIP
                  events on the launch stream, against the 8 TB/s HBM peak
  "cpu_baseline": the CPU oracle (a from-scratch restatement, kind "port") timed on this box's host cores on a
                  bounded sample of the same frames.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def level_sizes(P, rows, cols, nlevels):
    ex_isf = [np.float32(1.0)]
    sf = np.float32(1.0)
    out = []
    for l in range(nlevels):
        if l > 0:
            sf = np.float32(np.float64(sf) * np.float64(np.float32(1.2)))
        isf = np.float32(1.0) / sf
        out.append((int(np.rint(np.float32(cols) * isf)), int(np.rint(np.float32(rows) * isf))))
    return out


def cpu_baseline(O, frames, nfeatures, nlevels, budget_s=15.0):
    """Oracle (port) on all host cores, one frame per task (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1
    handles = [O.OrbOracle(nfeatures, 1.2, nlevels, 20, 7) for _ in range(cores)]

    # calibrate on one frame, then size the sample (frames are cycled) to the time budget
    t0 = time.perf_counter()
    handles[0].extract(frames[0])
    per = time.perf_counter() - t0
    per_thread = int(max(2, min(64, budget_s / max(per, 1e-4))))
    total = per_thread * cores
    nf = len(frames)

    def work(t):
        for k in range(per_thread):
            handles[t].extract(frames[(t * per_thread + k) % nf])
        return per_thread

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d synthetic 640x480 frame extractions (ORB, oracle/ restatement, g++ -O2 -ffp-contract=off), %d host threads x %d frames" % (done, cores, per_thread)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="frames per GPU per step")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--nlevels", type=int, default=8)
    ap.add_argument("--unique", type=int, default=32, help="distinct rasterised frames (rest are cheap variants)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    P, S = _util.plslam(), _util.synth()
    B, rows, cols = args.batch, args.rows, args.cols
    frames = S.make_frames(2 + 100000 * rank, B, rows, cols, unique=args.unique)
    d_imgs = torch.from_numpy(frames).to(dev)
    ex = P.ORBextractor(args.nfeatures, 1.2, args.nlevels, 20, 7, rows=rows, cols=cols, max_batch=B, device=local_rank)
    cap = ex.capacity
    d_kps = torch.empty((B, cap, 7), dtype=torch.float32, device=dev)     # 28-byte records
    d_desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    if world > 1:
        g_kps = torch.empty((world * B, cap, 7), dtype=torch.float32, device=dev)
        g_desc = torch.empty((world * B, cap, 32), dtype=torch.uint8, device=dev)
        g_n = torch.empty((world * B,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        ex.extract_batch_dev(d_imgs, B, rows * cols, d_kps, d_desc, d_n, stream.cuda_stream)
        if world > 1:   # RCCL gather of the fixed-stride records over xGMI
            dist.all_gather_into_tensor(g_n, d_n)
            dist.all_gather_into_tensor(g_kps, d_kps)
            dist.all_gather_into_tensor(g_desc, d_desc)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    ex.set_profiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        names = ["k_pyr_down (7 levels)", "k_fast_cells", "k_octree", "k_orient_brief"]
        kms = [ex.kernel_ms(k) for k in range(4)]
        sizes = level_sizes(P, rows, cols, args.nlevels)
        Ppx = sum(w * h for w, h in sizes)
        WH = rows * cols
        nkp = float(d_n.float().mean().item())
        # algorithmic bytes per frame of each kernel group (DESIGN.md "kernels")
        alg = [(Ppx - sizes[-1][0] * sizes[-1][1]) + (Ppx - WH),       # pyramid: read levels 0..L-2, write levels 1..L-1
               Ppx,                                                    # FAST: every level read once
               0,                                                      # quad-tree: latency-bound list work
               nkp * (43 * 43 + 32 + 28)]                              # orientation + rBRIEF patch gathers
        per_launch_ms = [ms / max(n, 1) for ms, n in kms]
        dom = int(np.argmax(per_launch_ms))
        if alg[dom] == 0:   # report the dominant *streaming* kernel against HBM; the list kernel has no byte roofline
            dom_stream = int(np.argmax([per_launch_ms[k] if alg[k] > 0 else -1 for k in range(4)]))
        else:
            dom_stream = dom
        ach = alg[dom_stream] * B / (per_launch_ms[dom_stream] * 1e-3) / 1e9 if per_launch_ms[dom_stream] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel_group") == dom_stream:
                    traffic = tj["bytes_per_frame"] * B
            except Exception:
                traffic = None
        out = {
            "metric": "frames/s ORB extract (pyramid+FAST+quad-tree+IC-angle+rBRIEF), 640x480 mono",
            "value": round(world * B * args.steps / dt, 2), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%dx%d mono, %d-level pyramid, %d ORB (TUM1.yaml), batch %d frames/GPU resident in HBM"
                                   % (cols, rows, args.nlevels, args.nfeatures, B),
                       "stages": "ORB extract", "mean_keypoints_per_frame": round(nkp, 1),
                       "parallelism": "frames sharded 1 batch/GPU" + (", RCCL all_gather of records" if world > 1 else "")},
            "kernel_ms_per_launch": {names[k]: round(per_launch_ms[k], 4) for k in range(4)},
            "roofline": {"bound": "hbm", "kernel": names[dom_stream], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(alg[dom_stream] * B),
                         "dominant_by_time": names[dom]},
        }
        if not args.no_cpu_baseline:
            O = _util.oracle()
            O.build()
            out["cpu_baseline"] = cpu_baseline(O, frames[:min(B, 256)], args.nfeatures, args.nlevels)
        print(json.dumps(out), flush=True)
    ex.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
