#!/usr/bin/env python3
"""bench.py -- throughput of the PL-SLAM front-end hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one pass of the whole front end over one batch of synthetic frames already resident in HBM
(pl-slam_amd/pipeline.py): ORB extract + LSD/LBD line extract (with the Frame.cc undistortion remap) + BoW feature
vectors + ORBmatcher::SearchByBoW and LSDmatcher::SearchDouble between consecutive frames.
Workload = BASELINE.json configs[2] ("640x480 ORB+LSD+LBD full extract, TUM-style intrinsics, 1000 ORB / 200 lines")
plus the frame-to-frame match of configs[3]; `--batch` frames per GPU (weak scaling: frames are independent,
SURVEY.md 8e), and for N > 1 one RCCL all_gather of the fixed-stride keypoint / descriptor / keyline records.

One JSON line on stdout (rank 0) with, besides the contract fields,
  "roofline":     the dominant kernel: algorithmic bytes per launch / its mean duration measured live with HIP events
                  on the launch stream, against the 8 TB/s HBM peak; "roofline_fast" repeats it for the FAST kernel
                  the north star sets its 60 % goal on;
  "cpu_baseline": the CPU oracle (from-scratch restatement, kind "port") timed on this box's host cores on a
                  bounded sample of the same frames (same stages, all cores, one frame per task).
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
TUM1_K = [517.306408, 516.469215, 318.643040, 255.313989]        # Examples/Monocular/TUM1.yaml:8-11
TUM1_D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]    # TUM1.yaml:13-17


def level_sizes(rows, cols, nlevels):
    sf = np.float32(1.0)
    out = []
    for l in range(nlevels):
        if l > 0:
            sf = np.float32(np.float64(sf) * np.float64(np.float32(1.2)))
        isf = np.float32(1.0) / sf
        out.append((int(np.rint(np.float32(cols) * isf)), int(np.rint(np.float32(rows) * isf))))
    return out


def cpu_baseline(O, V, frames, voc, nfeatures, nlevels, nlines, K, D, budget_s=20.0):
    """The oracle on all host cores: per frame ORB + (remap) + lines + BoW transform + both matchers against the
    previous frame of the same worker (ctypes releases the GIL, so threads run in parallel)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1
    L = O.lib()
    L.plo_bow_transform.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.plo_bow_transform.restype = None
    rows, cols = frames[0].shape
    mx = np.zeros((rows, cols), np.float32)
    my = np.zeros((rows, cols), np.float32)
    Kf, Df = np.asarray(K, np.float32), np.asarray(D, np.float32)
    L.plo_undistort_maps(O._p(Kf), O._p(Df), cols, rows, O._p(mx), O._p(my))

    def one_frame(orb, img, prev):
        kps, desc = orb.extract(img)
        und = np.zeros_like(img)
        L.plo_remap_linear_u8(O._p(img), cols, rows, cols, O._p(mx), O._p(my), O._p(und), cols)
        kl, ldesc, fn = O.line_extract(und, nlines, 0.0)
        n = len(desc)
        nid = np.zeros(max(n, 1), np.int32)
        word = np.zeros(max(n, 1), np.int32)
        L.plo_bow_transform(O._p(desc), n, O._p(voc.node_desc), O._p(voc.child_start), O._p(voc.child_count), O._p(voc.word_id),
                            O._p(voc.weight), voc.L, 4, O._p(nid), O._p(word))
        cur = (desc, np.ascontiguousarray(kps["angle"]), nid, ldesc)
        if prev is not None:
            pd, pa, pn, pl = prev
            valid = np.ones(len(pd), np.uint8)
            m = np.zeros(max(n, 1), np.int32)
            L.plo_orb_search_by_bow(O._p(pd), O._p(pa), O._p(pn), O._p(valid), len(pd), O._p(desc), O._p(cur[1]), O._p(nid), n,
                                    50, 0.7, 1, O._p(m))
            ml = np.zeros(max(len(pl), 1), np.int32)
            L.plo_line_search_double(O._p(pl), len(pl), O._p(ldesc), len(ldesc), 50.0, 0.7, O._p(ml))
        return cur

    orbs = [O.OrbOracle(nfeatures, 1.2, nlevels, 20, 7) for _ in range(cores)]
    t0 = time.perf_counter()
    one_frame(orbs[0], frames[0], one_frame(orbs[0], frames[1 % len(frames)], None))
    per = (time.perf_counter() - t0) / 2
    per_thread = int(max(2, min(48, budget_s / max(per, 1e-4))))
    nf = len(frames)

    def work(t):
        prev = None
        for k in range(per_thread):
            prev = one_frame(orbs[t], frames[(t * per_thread + k) % nf], prev)
        return per_thread

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d synthetic %dx%d frames (ORB + remap + LSD/LBD + BoW + SearchByBoW + SearchDouble), oracle/ restatement "
                      "(g++ -O2 -ffp-contract=off, no OpenCV SIMD), %d host threads x %d frames" % (done, cols, rows, cores, per_thread)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=6144, help="frames per GPU per step (6 k_lsd_grow wavefronts per SIMD = 6144 resident frames)")
    ap.add_argument("--nsplit", type=int, default=4, help="sub-batches pipelined against each other (pl-slam_amd/pipeline.py)")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--nlevels", type=int, default=8)
    ap.add_argument("--nlines", type=int, default=200)
    ap.add_argument("--unique", type=int, default=32, help="distinct rasterised frames (the rest are cheap variants)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fake-gather", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--force-dist", action="store_true", help=argparse.SUPPRESS)   # 1-rank RCCL group: rehearses the N > 1 path
    ap.add_argument("--serial", action="store_true", help="ORB and line halves on one stream (no overlap); used for PMC runs")
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
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # keep stdout to the one JSON line: RCCL's NCCL_DEBUG=VERSION banner (set in this image) goes to a file
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/plslam_bench_rccl_%h_%p.log")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    P, S = _util.plslam(), _util.synth()
    V = _util._load("plslam_amd_vocab", os.path.join(ROOT, "pl-slam_amd", "vocab.py"))
    PL = _util._load("plslam_amd_pipeline", os.path.join(ROOT, "pl-slam_amd", "pipeline.py"))
    B, rows, cols = args.batch, args.rows, args.cols
    tum = (rows, cols) == (480, 640)
    K, D = (TUM1_K, TUM1_D) if tum else (None, None)     # KITTI: zero distortion -> no remap (Frame.cc:917-921)
    frames = S.make_frames(2 + 100000 * rank, B, rows, cols, unique=args.unique)
    d_imgs = torch.from_numpy(frames).to(dev)
    voc = V.Vocabulary.synthetic(102, k=10, L=6, synth=S)
    fe = PL.FrontEndPipelined(P, voc, B, rows, cols, args.nfeatures, args.nlevels, args.nlines, 0.0, K, D, device=local_rank,
                              nsplit=args.nsplit)
    fe.overlap = not args.serial
    Bp = fe.Bp
    DI = _util._load("plslam_amd_dist", os.path.join(ROOT, "pl-slam_amd", "dist.py"))

    gathering = world > 1 or args.fake_gather or args.force_dist
    comm = torch.cuda.Stream(device=dev) if gathering else None
    gdist = dist
    if args.fake_gather and world == 1:   # single-GPU rehearsal of the N > 1 choreography (a copy stands in for RCCL)
        class _Loop:
            @staticmethod
            def all_gather_into_tensor(dst, src):
                dst.copy_(src)
        gdist = _Loop

    def step():
        # consecutive steps are independent batches and overlap (sub-batch pipelining).  N > 1: the fixed-stride records of
        # every sub-batch are all_gather'ed over RCCL / xGMI on a communication stream as soon as that sub-batch is done
        # (pl-slam_amd/dist.py, gloo-tested on CPU); only the sub-batch's own next step waits for its gather
        fe.step(d_imgs, join=False)
        if gathering:
            fe.gather(comm, world, gdist, DI.all_gather_records)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    for part in fe.parts:
        part.orb.set_profiling(True)
        part.line.lib.plh_line_set_profiling.argtypes = [C_VOID, C_INT]
        part.line.lib.plh_line_set_profiling(part.line.h, 1)
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

    def read_kernel_totals():
        """Cumulative (ms, intervals) of the 8 kernel groups since profiling was switched on."""
        import ctypes as C
        tot = [[0.0, 0] for _ in range(8)]
        for part in fe.parts:
            for k in range(4):
                ms, n = part.orb.kernel_ms(k)
                tot[k][0] += ms
                tot[k][1] += n
            lib = part.line.lib
            lib.plh_line_kernel_ms.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
            for k in range(4):
                ms, n = C.c_double(0), C.c_int(0)
                lib.plh_line_kernel_ms(part.line.h, k, C.byref(ms), C.byref(n))
                tot[4 + k][0] += ms.value
                tot[4 + k][1] += n.value
        return [tuple(x) for x in tot]

    if rank == 0:
        names = ["k_pyr_down x7", "k_fast_strips", "k_octree", "k_orient_brief", "line prep (remap/blur/resize/grad/order)",
                 "k_lsd_grow", "k_keylines", "LBD (blur+sobel+k_lbd)"]
        t1 = read_kernel_totals()
        per_ms_timed = [ms / max(n, 1) for ms, n in t1]   # HIP events on the launch streams, over the timed region
        # one more pass with both halves on one stream: per-kernel durations without interference between the halves
        fe.overlap = False
        for _ in range(2):
            fe.step(d_imgs, join=True)
        torch.cuda.synchronize(dev)
        t2 = read_kernel_totals()
        per_ms = [(b[0] - a[0]) / max(b[1] - a[1], 1) for a, b in zip(t1, t2)]
        fe.overlap = not args.serial
        sizes = level_sizes(rows, cols, args.nlevels)
        Ppx = sum(w * h for w, h in sizes)
        WH = rows * cols
        res = fe.results()
        nkp, nln = float(res["n"].mean()), float(res["nl"].mean())
        sWH = int(np.rint(cols * 0.8)) * int(np.rint(rows * 0.8))
        # algorithmic bytes per frame of each kernel group (DESIGN.md "kernels and rooflines")
        alg = [(Ppx - sizes[-1][0] * sizes[-1][1]) + (Ppx - WH),         # pyramid: read levels 0..L-2, write 1..L-1
               Ppx,                                                      # FAST: every level read once
               0,                                                        # quad-tree: latency-bound list work
               nkp * (43 * 43 + 32 + 28),                                # orientation + rBRIEF patch gathers
               (2 * WH if tum else 0) + 2 * WH + WH + sWH + sWH * (1 + 16) + sWH * (16 + 4),   # remap, blur, resize, records, order
               3 * sWH * 9,                                              # region growing: ~3 passes over the 0.64WH field
               0,
               2 * WH + WH + 4 * WH + nln * 63 * 80 * 4]                 # LBD: blur, Sobel read/write, band gathers
        dom = int(np.argmax(per_ms))
        # PMC traffic (FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_traffic.sh -> profiles/hbm_traffic.json), bytes per frame
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and tum and args.nfeatures == 1000:   # the counters were collected on this workload only
            try:
                traffic = json.load(open(tpath)).get("kernels", {})
            except Exception:
                traffic = {}
        pmc_names = {1: ["k_fast_strips"], 5: ["k_lsd_grow"], 0: ["k_pyr_down"], 2: ["k_octree"], 3: ["k_orient_brief"]}

        def roof(k, ms, where):
            ach = alg[k] * Bp / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            t = [traffic[n]["total"] * traffic[n].get("launches_per_step", 1) for n in pmc_names.get(k, []) if n in traffic]
            return {"bound": "hbm", "kernel": names[k], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": int(sum(t) * Bp) if t else None,
                    "algorithmic_bytes_per_launch": int(alg[k] * Bp), "frames_per_launch": Bp, "ms_per_launch": round(ms, 4),
                    "measured": where}

        r_dom = roof(dom, per_ms_timed[dom], "HIP events on the launch stream over the timed region")
        r_dom["ms_per_launch_alone"] = round(per_ms[dom], 4)
        out = {
            "metric": "frames/s ORB+LSD extract+match, %dx%d mono" % (cols, rows),
            "value": round(world * B * args.steps / dt, 2), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%dx%d mono, %d-level pyramid, %d ORB / %d lines (%s parameters), batch %d frames/GPU resident in HBM "
                                   "(%d sub-batches of %d pipelined); extract + BoW + SearchByBoW + line SearchDouble per consecutive "
                                   "frame pair" % (cols, rows, args.nlevels, args.nfeatures, args.nlines,
                                                   "TUM1.yaml" if tum else "KITTI00-02.yaml", B, args.nsplit, Bp),
                       "mean_keypoints_per_frame": round(nkp, 1), "mean_keylines_per_frame": round(nln, 1),
                       "mean_orb_matches_per_pair": round(float(res["nm_orb"].mean()), 1),
                       "mean_line_matches_per_pair": round(float(res["nm_line"].mean()), 1),
                       "vocabulary": "synthetic k=10 L=6 (ORBvoc.bin is not in the mount)",
                       "streams": "line chain on a high-priority stream, ORB + BoW + SearchByBoW on a second stream" if not args.serial else "one stream",
                       "parallelism": "frames sharded 1 batch/GPU" + (", RCCL all_gather of the records per sub-batch on a "
                                                                       "communication stream" if world > 1 else "")},
            "kernel_ms_per_launch": {names[k]: round(per_ms[k], 4) for k in range(8)},
            "kernel_ms_per_launch_timed_region": {names[k]: round(per_ms_timed[k], 4) for k in range(8)},
            "roofline": r_dom,
            "roofline_fast": roof(1, per_ms[1], "HIP events, extra pass after the timed region with both halves on one stream"),
        }
        if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only (rank 0)
            O = _util.oracle()
            O.build()
            out["cpu_baseline"] = cpu_baseline(O, V, frames[:min(B, 64)], voc, args.nfeatures, args.nlevels, args.nlines,
                                               TUM1_K if tum else [718.856, 718.856, 607.1928, 185.2157],
                                               TUM1_D if tum else [0, 0, 0, 0, 0])
        result_line = json.dumps(out)
    fe.close()
    try:   # flush what native libraries (the RCCL version banner) buffered on C stdio, on every rank, BEFORE the result line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if world > 1 or args.force_dist:
        dist.barrier()
    if rank == 0:
        print(result_line, flush=True)   # the one JSON line, last on stdout
    if world > 1 or args.force_dist:
        dist.barrier()
        dist.destroy_process_group()


import ctypes as _C  # noqa: E402
C_VOID, C_INT = _C.c_void_p, _C.c_int

if __name__ == "__main__":
    main()
