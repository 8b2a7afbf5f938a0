#!/usr/bin/env python3
"""bench.py -- throughput of the PL-SLAM front-end hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one pass of the whole front end over one batch of synthetic frames ALREADY RESIDENT IN HBM
(pl-slam_amd/pipeline.py): ORB extract + LSD/LBD line extract (with the Frame.cc undistortion remap) + Frame::ComputeBoW
(FeatureVector and BowVector) + ORBmatcher::SearchByBoW and LSDmatcher::SearchDouble between consecutive frames.
Workload = BASELINE.json configs[2] ("640x480 ORB+LSD+LBD full extract, TUM-style intrinsics, 1000 ORB / 200 lines")
plus the frame-to-frame match of configs[3]; `--batch` frames per GPU (weak scaling: frames are independent,
SURVEY.md 8e), and for N > 1 the RCCL gather of the fixed-stride keypoint / descriptor / keyline records through the C ABI
(plh_gather_records, one grouped launch per sub-batch on a communication stream).

`value` is the resident-batch kernel throughput: the same batch is re-processed every step, nothing crosses PCIe inside the
timed region and consecutive steps overlap.  What a host that streams frames sees is reported next to it ("streaming":
fresh frames uploaded and all records downloaded every step, steps joined).

One JSON line on stdout (rank 0) with, besides the contract fields,
  "roofline":      the dominant kernel: algorithmic bytes per launch / its mean duration measured live with HIP events on the
                   launch stream over the timed region, against the 8 TB/s HBM peak;
  "roofline_fast": the same for the FAST kernel the north star sets its 60 % goal on, plus its VALU-issue roofline (the
                   bound it really runs against);
  "cpu_baseline":  the CPU oracle (from-scratch restatement, kind "port") on this box's host cores -- one native thread, one per
                   physical core, one per hardware thread (oracle/frontend.cc: std::thread shards) -- bounded sample;
  "secondary":     (N = 1) the KITTI 1241x376 / 2000-feature workload of BASELINE configs[4]: one GPU's resident-batch rate,
                   and the literal configs[4] share of 4096 / 8 = 512 frames per GPU;
  "streaming":     (N = 1) the PCIe-inclusive rate described above;
  "latency_ms_single_frame": (N = 1) one Frame() worth of extraction through the host-buffer entry points, ORB and lines on
                   two threads as Frame.cc:224-227 runs them;
  "verified":      the records the timed steps left behind for EVERY frame of one sub-batch (which one rotates with the hour of the
                   run: 1536 of the 6144 frames) and the first 16 frames of every other sub-batch (and of each secondary batch),
                   compared bit for bit with the CPU oracle on the same frames; with N > 1 every rank verifies its own
                   batch and the line carries the per-rank verdicts; a mismatch makes the run exit non-zero.
  "secondary.refine_std": (N = 1) the same workload with cv::LSD_REFINE_STD -- the level of the un-linked twin in the reference's
                   tree; the headline runs LSD_REFINE_ADV, what the linked opencv_contrib LSDDetector creates (`--refine std`
                   makes STD the headline and reports ADV here): resident rate, the 512-frame share, one-frame latency, verified,
                   its own roofline.
  "box":           a fixed VALU probe (tools/ubench/valu_rate, 50 ms) and the clocks / power cap rocm-smi reports in front of the
                   timed region: what a number from another box has to be normalised with.
`--frames <dir | file>` replaces the synthetic frames by real ones (raw 8-bit planes or PGM, pl-slam_amd/frames_io.py), with
the TUM1 / KITTI00-02 camera picked by --rows / --cols.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
VALU_PEAK_GINST = 1024 * 2.4 / 4.2   # wave64 VALU instructions / ns: 1024 SIMDs, 2.4 GHz, 4.2 cycles per instruction of the
                                     # front end's op mix (profiles/r01_valu_issue_rate_gfx950.txt)
TUM1_K = [517.306408, 516.469215, 318.643040, 255.313989]        # Examples/Monocular/TUM1.yaml:8-11
TUM1_D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]    # TUM1.yaml:13-17
KITTI_K = [718.856, 718.856, 607.1928, 185.2157]                  # Examples/Monocular/KITTI00-02.yaml:8-11
NAMES = ["k_pyr_down x7", "k_fast_strips", "k_octree", "k_orient_brief", "line prep (remap/blur/resize/grad/order)",
         "k_lsd_grow", "k_lsd_rects + k_keylines", "LBD (blur+sobel+k_lbd)"]


def level_sizes(rows, cols, nlevels):
    sf = np.float32(1.0)
    out = []
    for l in range(nlevels):
        if l > 0:
            sf = np.float32(np.float64(sf) * np.float64(np.float32(1.2)))
        isf = np.float32(1.0) / sf
        out.append((int(np.rint(np.float32(cols) * isf)), int(np.rint(np.float32(rows) * isf))))
    return out


def physical_cores():
    """Distinct (package, core) pairs of /proc/cpuinfo; falls back to os.cpu_count()."""
    try:
        seen, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
        return len(seen) or (os.cpu_count() or 1)
    except Exception:
        return os.cpu_count() or 1


def cpu_quota():
    """CPUs the container may use at a time: cgroup v2 cpu.max or v1 cfs quota / period; None when unlimited or unreadable."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(O, V, frames, voc, nfeatures, nlevels, nlines, K, D, refine=None, budget_s=8.0):
    """The oracle on the host cores, natively threaded (oracle/frontend.cc plo_frontend_batch: one std::thread per worker, each
    with its own ORB handle and buffers): per frame ORB + (remap) + lines + BoW transform + both matchers against the worker's
    previous frame -- the work of one product step per frame.  Legs: one thread, one per physical core, one per hardware thread,
    and -- when the container is given fewer CPUs than the host has (a cgroup quota, or simply measured: CPU seconds / wall
    seconds of the physical-core leg) -- one per CPU it really gets; `value` is the best of them.  `cores_busy` = the process's
    CPU seconds over the leg's wall seconds: the cores that actually ran the threads."""
    import ctypes as C
    import resource
    L = O.lib()
    L.plo_frontend_batch.restype = C.c_double
    L.plo_frontend_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 8 + \
                                    [C.c_int, C.c_int, C.c_int, C.c_void_p]
    frames = np.ascontiguousarray(frames)
    n, rows, cols = frames.shape
    refine = O.REFERENCE_REFINE if refine is None else int(refine)   # None: the reference's level (LSD_REFINE_ADV, oracle/plo.py)
    mx = my = None
    if K is not None and D is not None and any(D):
        mx = np.zeros((rows, cols), np.float32)
        my = np.zeros((rows, cols), np.float32)
        L.plo_undistort_maps(O._p(np.asarray(K, np.float32)), O._p(np.asarray(D, np.float32)), cols, rows, O._p(mx), O._p(my))
    ww = np.ascontiguousarray(voc.word_weight(), np.float64)

    def run(threads, per_thread):
        chk = C.c_ulonglong(0)
        u0 = resource.getrusage(resource.RUSAGE_SELF)
        dt = L.plo_frontend_batch(O._p(frames), n, rows, cols, nfeatures, nlevels, nlines, int(refine), O._p(mx), O._p(my),
                                  O._p(voc.node_desc), O._p(voc.child_start), O._p(voc.child_count), O._p(voc.word_id), O._p(voc.weight),
                                  O._p(ww), voc.L, threads, per_thread, C.byref(chk))
        u1 = resource.getrusage(resource.RUSAGE_SELF)
        busy = ((u1.ru_utime - u0.ru_utime) + (u1.ru_stime - u0.ru_stime)) / max(dt, 1e-9)
        return threads * per_thread / dt, dt, busy

    hw, phys, quota = os.cpu_count() or 1, physical_cores(), cpu_quota()
    r1, dt1, _ = run(1, 3)
    per = dt1 / 3
    legs = {"1": {"threads": 1, "frames_per_s": round(r1, 2), "ms_per_frame": round(per * 1e3, 2)}}
    best, best_threads = r1, 1
    todo = [("physical_cores", phys), ("hardware_threads", hw)]
    if quota is not None and quota < phys:
        todo.insert(0, ("cgroup_cpu_quota", max(1, int(round(quota)))))
    granted = None
    while todo:
        name, th = todo.pop(0)
        if th <= 1 or str(th) in legs:
            continue
        per_thread = int(max(3, min(32, budget_s * min(th, granted or th) / th / max(per, 1e-4))))
        r, dt, busy = run(th, per_thread)
        legs[str(th)] = {"threads": th, "which": name, "frames_per_s": round(r, 2), "frames": th * per_thread, "seconds": round(dt, 2),
                         "cores_busy": round(busy, 1), "parallel_efficiency": round(r / (r1 * th), 3),
                         "efficiency_per_busy_core": round(r / (r1 * max(busy, 1.0)), 3)}
        if r > best:
            best, best_threads = r, th
        if name == "physical_cores" and granted is None and busy < 0.6 * th:
            # the threads were runnable but ran on `busy` cores: the container gets fewer CPUs than the host shows
            granted = max(1, int(round(busy)))
            todo.insert(0, ("cores_granted_measured", granted))
    return {"value": round(best, 2), "unit": "frames/s", "cores": best_threads, "kind": "port", "physical_cores": phys,
            "hardware_threads": hw, "cgroup_cpu_quota": quota, "cores_granted_measured": granted, "legs": legs,
            "single_thread_ms_per_frame": round(per * 1e3, 1),
            "sample": "synthetic %dx%d frames (ORB + remap + LSD%s/LBD + BoW + SearchByBoW + SearchDouble per frame), oracle/ restatement "
                      "(g++ -O2 -ffp-contract=off, no OpenCV SIMD), native std::thread shards (oracle/frontend.cc); cores = the thread "
                      "count of the best leg" % (cols, rows, " (LSD_REFINE_ADV)" if refine else "")}


def oracle_records(O, V, frames, voc, nfeatures, nlevels, nlines, K, D, refine=None):
    """Everything one step produces for `frames` (consecutive frames of a batch), from the CPU oracle, on all host cores:
    per frame keypoints / rBRIEF / FeatureVector nodes / words / BowVector / keylines / LBD / line equations, and per
    consecutive pair the SearchByBoW and SearchDouble match lists.  The checker of `verify_records`, nothing else."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    L = O.lib()
    L.plo_bow_transform.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.plo_bow_transform.restype = None
    L.plo_bow_vector.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.plo_bow_vector.restype = C.c_int
    rows, cols = frames[0].shape
    undist = K is not None and D is not None and any(D)
    if undist:
        mx = np.zeros((rows, cols), np.float32)
        my = np.zeros((rows, cols), np.float32)
        Kf, Df = np.asarray(K, np.float32), np.asarray(D, np.float32)
        L.plo_undistort_maps(O._p(Kf), O._p(Df), cols, rows, O._p(mx), O._p(my))
    ww = voc.word_weight()

    def one(img):
        orb = O.OrbOracle(nfeatures, 1.2, nlevels, 20, 7)
        kps, desc = orb.extract(img)
        src = img
        if undist:
            src = np.zeros_like(img)
            L.plo_remap_linear_u8(O._p(img), cols, rows, cols, O._p(mx), O._p(my), O._p(src), cols)
        kl, ldesc, fn = O.line_extract(src, nlines, 0.0, refine=refine)
        n = len(desc)
        nid = np.zeros(max(n, 1), np.int32)
        word = np.zeros(max(n, 1), np.int32)
        L.plo_bow_transform(O._p(desc), n, O._p(voc.node_desc), O._p(voc.child_start), O._p(voc.child_count), O._p(voc.word_id),
                            O._p(voc.weight), voc.L, 4, O._p(nid), O._p(word))
        bw, bv = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
        m = L.plo_bow_vector(O._p(word), n, O._p(ww), 0, 0, O._p(bw), O._p(bv), max(n, 1))
        return dict(kps=kps, desc=desc, nid=nid[:n], word=word[:n], bow_word=bw[:m], bow_value=bv[:m], kl=kl, ldesc=ldesc, lfn=fn)

    def pair(ab):
        a, b = ab
        valid = np.ones(len(a["desc"]), np.uint8)
        a1, a2 = np.ascontiguousarray(a["kps"]["angle"]), np.ascontiguousarray(b["kps"]["angle"])
        m = np.zeros(max(len(b["desc"]), 1), np.int32)
        c = L.plo_orb_search_by_bow(O._p(a["desc"]), O._p(a1), O._p(a["nid"]), O._p(valid), len(a["desc"]), O._p(b["desc"]), O._p(a2),
                                    O._p(b["nid"]), len(b["desc"]), 50, 0.7, 1, O._p(m))
        ml = np.zeros(max(len(a["ldesc"]), 1), np.int32)
        cl = L.plo_line_search_double(O._p(a["ldesc"]), len(a["ldesc"]), O._p(b["ldesc"]), len(b["ldesc"]), 50.0, 0.7, O._p(ml))
        return dict(m_orb=m[:len(b["desc"])], nm_orb=c, m_line=ml[:len(a["ldesc"])], nm_line=cl)

    with ThreadPoolExecutor(min(os.cpu_count() or 1, len(frames))) as ex:
        recs = list(ex.map(one, list(frames)))
        pairs = list(ex.map(pair, list(zip(recs[:-1], recs[1:]))))
    return recs, pairs


def verify_records(res, recs, pairs, first=0):
    """Compare frames [first, first + len(recs)) of a step's results (FrontEnd*.results()) with the oracle's: every record, bit for
    bit; pairs[i] = frame first + i -> first + i + 1.  Returns the list of mismatches (empty = exact)."""
    bad = []
    for i, r in enumerate(recs):
        b = first + i
        n, nl = int(res["n"][b]), int(res["nl"][b])
        if n != len(r["desc"]):
            bad.append("frame %d: %d keypoints vs %d" % (b, n, len(r["desc"])))
            continue
        for f in r["kps"].dtype.names:
            if not (res["kps"][b, :n][f] == r["kps"][f]).all():
                bad.append("frame %d: keypoint field %s" % (b, f))
        if not (res["desc"][b, :n] == r["desc"]).all():
            bad.append("frame %d: rBRIEF descriptors" % b)
        if not ((res["nid"][b, :n] == r["nid"]).all() and (res["word"][b, :n] == r["word"]).all()):
            bad.append("frame %d: FeatureVector nodes / words" % b)
        m = len(r["bow_word"])
        if not (int(res["bow_n"][b]) == m and (res["bow_word"][b, :m] == r["bow_word"]).all() and (res["bow_value"][b, :m] == r["bow_value"]).all()):
            bad.append("frame %d: BowVector" % b)
        if nl != len(r["kl"]):
            bad.append("frame %d: %d keylines vs %d" % (b, nl, len(r["kl"])))
            continue
        for f in r["kl"].dtype.names:
            if not (res["kl"][b, :nl][f] == r["kl"][f]).all():
                bad.append("frame %d: keyline field %s" % (b, f))
        if not ((res["ldesc"][b, :nl] == r["ldesc"]).all() and (res["lfn"][b, :nl] == r["lfn"]).all()):
            bad.append("frame %d: LBD descriptors / line equations" % b)
    for i, q in enumerate(pairs):
        b = first + i
        if not (int(res["nm_orb"][b]) == q["nm_orb"] and (res["m_orb"][b, :len(q["m_orb"])] == q["m_orb"]).all()):
            bad.append("pair %d -> %d: SearchByBoW matches" % (b, b + 1))
        if not (int(res["nm_line"][b]) == q["nm_line"] and (res["m_line"][b, :len(q["m_line"])] == q["m_line"]).all()):
            bad.append("pair %d -> %d: SearchDouble matches" % (b, b + 1))
    return bad


def verify_batch(O, V, W, res, voc, per_part, full_part=None):
    """The records workload W's last steps left behind, against the oracle on the same frames: EVERY frame of sub-batch `full_part`
    (None: none) and the first `per_part` frames of every other sub-batch.  {"frames", "pairs", "exact", "mismatches", ...}."""
    nv = min(per_part, W.Bp)
    bad, nf, npairs = [], 0, 0
    for k in range(W.nsplit):
        first = k * W.Bp
        cnt = W.Bp if k == full_part else nv
        recs, pairs = oracle_records(O, V, W.frames[first:first + cnt], voc, W.nfeatures, W.nlevels, W.nlines, W.K, W.D, refine=W.refine)
        bad += verify_records(res, recs, pairs, first)
        nf += len(recs)
        npairs += len(pairs)
    full = "" if full_part is None else "every frame of sub-batch %d (%d frames; the sub-batch rotates with the hour of the run) and " % (full_part, W.Bp)
    return {"frames": nf, "pairs": npairs, "sub_batches": W.nsplit, "full_sub_batch": full_part, "exact": not bad, "mismatches": bad[:8],
            "what": "every record of %sthe first %d frames of each %ssub-batch of the timed batch (keypoints, rBRIEF, FeatureVector, "
                    "BowVector, keylines, LBD, line equations) and the match lists of their consecutive pairs, compared bit for bit with "
                    "the CPU oracle on the same frames" % (full, nv, "other " if full else "")}


def distinct_frames(S, rows, cols, n, scenes=128):
    """`n` frames no two of which share a control-flow trace (the timed batch is 32 rasterised frames + row shifts,
    i.e. 32 traces of region growing): `scenes` rasterised scenes of random density (10 .. 400 rectangles, 5 .. 200 lines), each followed by
    its variants -- shifted in both axes, contrast and exposure changed, sensor noise of random strength added -- which change gradients
    everywhere and so the seeds' order, the regions and the FAST corners.  Consecutive frames are views of one scene (the matchers find
    correspondences).  Deterministic."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.RandomState(20260 + rows + cols)
    scenes = max(1, min(scenes, n))
    dens = [(int(rng.randint(10, 401)), int(rng.randint(5, 201))) for _ in range(scenes)]
    with ThreadPoolExecutor(min(os.cpu_count() or 1, 32)) as ex:
        base = list(ex.map(lambda i: S.make_frame(9000 + i, rows, cols, n_rect=dens[i][0], n_line=dens[i][1]), range(scenes)))
    per = -(-n // scenes)
    out = np.empty((n, rows, cols), np.uint8)
    for i in range(n):
        b, k = base[i // per], i % per
        if k == 0:
            out[i] = b
            continue
        img = np.roll(np.roll(b, int(rng.randint(1, 9)) * k, axis=0), int(rng.randint(1, 9)) * k, axis=1).astype(np.float32)
        img = (img - 128.0) * rng.uniform(0.55, 1.0) + 128.0 + rng.uniform(-12, 12)
        amp = rng.uniform(0.5, 6.0)
        img += rng.standard_normal((rows, cols)).astype(np.float32) * amp
        out[i] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out


def box_probe(P, device):
    """What a rate measured on another box has to be normalised with (the driver's number is taken on a box the builder
    never sees): a fixed VALU-only launch timed by the library (plh_box_probe: 4096 x 256 threads x 20480 x 64 multiply-adds, ~40 ms on
    an MI355X at full clocks; first run from the power state the GPU was in, second at running clocks) and what rocm-smi says about
    clocks and the power cap.  Best effort: missing tools leave fields out, never fail the run."""
    import ctypes as C
    import subprocess
    out = {"what": "fixed VALU-only launch (plh_box_probe, 4096 x 256 threads x 20480 iterations x 64 mads) timed in front of the warm-up; "
                   "rates from boxes whose probe_ms differ compare after scaling by it"}
    try:
        ms = (C.c_float * 2)()
        if P.load().plh_box_probe(int(device), 20480, ms) == 0:
            out["probe_ms_first"], out["probe_ms"] = round(float(ms[0]), 3), round(float(ms[1]), 3)
    except Exception as e:   # noqa: BLE001
        out["probe_error"] = str(e)[:120]
    try:
        txt = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--showmaxpower", "--showpower"], capture_output=True, text=True,
                             timeout=20).stdout
        import re
        for key, pat in (("sclk_mhz", r"sclk clock level: \S+ \((\d+)Mhz\)"), ("mclk_mhz", r"mclk clock level: \S+ \((\d+)Mhz\)"),
                         ("power_cap_w", r"Max Graphics Package Power \(W\): ([0-9.]+)"),
                         ("power_w", r"Current Socket Graphics Package Power \(W\): ([0-9.]+)")):
            m = re.search(pat, txt)
            if m:
                out[key] = float(m.group(1))
    except Exception:   # noqa: BLE001
        pass
    return out


class Workload:
    """One resident batch + the pipelined front end over it."""

    def __init__(self, P, S, V, PL, torch, dev, rank, batch, nsplit, rows, cols, nfeatures, nlevels, nlines, unique, voc, serial=False,
                 shard=None, refine=0, screen=True, real=None, grow_waves=-1):
        self.P, self.torch, self.dev = P, torch, dev
        self.B, self.rows, self.cols, self.nfeatures, self.nlevels, self.nlines = batch, rows, cols, nfeatures, nlevels, nlines
        self.tum = (rows, cols) == (480, 640)
        K, D = (TUM1_K, TUM1_D) if self.tum else (None, None)     # KITTI: zero distortion -> no remap (Frame.cc:917-921)
        self.K, self.D, self.refine = K, D, int(refine)
        if real is not None:  # real frames (--frames): the sequence, cycled to fill the batch; rank r starts r batches in
            idx = (np.arange(batch) + (rank * batch if shard is None else shard[0] * batch)) % len(real)
            self.frames = np.ascontiguousarray(real[idx])
        elif shard is None:   # weak scaling: every rank has its own batch
            self.frames = S.make_frames(2 + 100000 * rank, batch, rows, cols, unique=unique)
        else:                 # strong scaling: this rank's contiguous shard of ONE job of `total` frames
            r, n, total = shard
            self.frames = S.make_frames(2, total, rows, cols, unique=unique, first=r * (total // n), n=total // n)   # (only the shard is made)
        self.d_imgs = torch.from_numpy(self.frames).to(dev)
        self.fe = PL.FrontEndPipelined(P, voc, batch, rows, cols, nfeatures, nlevels, nlines, 0.0, K, D, device=dev.index, nsplit=nsplit,
                                       lsd_refine=self.refine)
        if not screen:        # A/B switch: the exact rectangle behind every density decision (rounds 1-3)
            for part in self.fe.parts:
                part.line.set_screen(0)
        if grow_waves >= 0:
            for part in self.fe.parts:
                part.line.set_grow_waves(grow_waves)
        self.fe.overlap = not serial
        self.serial = serial
        self.nsplit, self.Bp = nsplit, self.fe.Bp

    def set_profiling(self, on):
        for part in self.fe.parts:
            part.orb.set_profiling(on)
            part.line.lib.plh_line_set_profiling.argtypes = [C_VOID, C_INT]
            part.line.lib.plh_line_set_profiling(part.line.h, int(on))

    def kernel_totals(self):
        """Cumulative (ms, intervals) of the 8 kernel groups since profiling was switched on."""
        import ctypes as C
        tot = [[0.0, 0] for _ in range(8)]
        for part in self.fe.parts:
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

    def run(self, steps, warmup, step_fn=None):
        """warmup untimed steps, then `steps` timed ones between device synchronisations; returns seconds."""
        t = self.torch
        step_fn = step_fn or (lambda: self.fe.step(self.d_imgs, join=False))
        for _ in range(warmup):
            step_fn()
        t.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        t.cuda.synchronize(self.dev)
        return time.perf_counter() - t0

    def algorithmic_bytes(self, res):
        """per frame and kernel group (DESIGN.md "kernels and rooflines")"""
        sizes = level_sizes(self.rows, self.cols, self.nlevels)
        Ppx = sum(w * h for w, h in sizes)
        WH = self.rows * self.cols
        nkp, nln = float(res["n"].mean()), float(res["nl"].mean())
        sWH = int(np.rint(self.cols * 0.8)) * int(np.rint(self.rows * 0.8))
        return [(Ppx - sizes[-1][0] * sizes[-1][1]) + (Ppx - WH),         # pyramid: read levels 0..L-2, write 1..L-1
                Ppx,                                                      # FAST: every level read once
                0,                                                        # quad-tree: latency-bound list work
                nkp * (43 * 43 + 32 + 28),                                # orientation + rBRIEF patch gathers
                (2 * WH if self.tum else 0) + 2 * WH + WH + sWH + sWH * (1 + 16) + sWH * (16 + 4),   # remap, blur, resize, records, order
                3 * sWH * 9,                                              # region growing: ~3 passes over the 0.64WH field
                0,
                2 * WH + WH + 4 * WH + nln * 63 * 80 * 4]                 # LBD: blur, Sobel read/write, band gathers

    def close(self):
        if self.fe is not None:
            self.fe.close()
        self.fe = None
        self.d_imgs = None


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            pass
    return {}


def single_frame_latency(P, torch, dev, frames, nfeatures, nlevels, nlines, K, D, reps=12, refine=0):
    """One Frame() worth of extraction through the host-buffer entry points (plh_orb_extract / plh_line_extract: H2D, kernels,
    D2H, blocking), ORB and lines on two host threads as Frame.cc:224-227 runs ExtractORB / ExtractLSD."""
    rows, cols = frames[0].shape
    orb = P.ORBextractor(nfeatures, 1.2, nlevels, 20, 7, rows=rows, cols=cols, max_batch=1, device=dev.index)
    line = P.LINEextractor(1, 1.2, nlines, 0.0, rows=rows, cols=cols, max_batch=1, device=dev.index, K=K, D=D)
    line.set_refine(refine)
    out = {}

    def timed(fn, n):
        fn(frames[0])                        # first call: plan + staging buffers
        t0 = time.perf_counter()
        for i in range(n):
            fn(frames[(i + 1) % len(frames)])
        return (time.perf_counter() - t0) / n * 1e3

    def both(img):
        ta = threading.Thread(target=orb, args=(img,))
        tb = threading.Thread(target=line, args=(img,))
        ta.start(); tb.start(); ta.join(); tb.join()

    out["orb"] = round(timed(orb, reps), 3)
    out["line"] = round(timed(line, reps), 3)
    out["total"] = round(timed(both, reps), 3)
    out["note"] = ("host-buffer calls on one %dx%d frame, %d ORB / %d lines, two host threads (Frame.cc:224-227); the frame's LSD region growing runs as "
                   "optimistic transactions on eight wavefronts (k_lsd_grow_mw), see DESIGN.md 'single-frame latency'" % (cols, rows, nfeatures, nlines))
    orb.close()
    line.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=6144, help="frames per GPU per step (6144 = 6 k_lsd_grow wavefronts per SIMD; the kernel is built for up to 8)")
    ap.add_argument("--nsplit", type=int, default=4, help="sub-batches pipelined against each other (pl-slam_amd/pipeline.py)")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--nlevels", type=int, default=8)
    ap.add_argument("--nlines", type=int, default=200)
    ap.add_argument("--unique", type=int, default=32, help="distinct rasterised frames (the rest are cheap variants)")
    ap.add_argument("--gather", choices=["root", "all"], default="root", help="N > 1: records gathered to rank 0 or to every rank")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch frames per GPU; strong: --total frames in all, contiguous shards of total / N per rank "
                         "(BASELINE configs[4]: --scaling strong --total 4096 --rows 376 --cols 1241)")
    ap.add_argument("--total", type=int, default=4096, help="--scaling strong: frames of the whole job")
    ap.add_argument("--refine", choices=["std", "adv", "default"], default="default",
                    help="cv::LineSegmentDetector's refine level of the headline: LSD_REFINE_STD, LSD_REFINE_ADV, or the library's "
                         "build-time default (PLH_LSD_REFINE_DEFAULT; the other level is reported under `secondary`)")
    ap.add_argument("--no-screen", action="store_true", help="A/B: region growing without the density screen (same records)")
    ap.add_argument("--grow-waves", type=int, default=-1, help="wavefronts per frame of LSD's region growing (-1: the front end's choice "
                                                                "by resident frames; 0: one; 2..16); used by the PMC scripts on small batches")
    ap.add_argument("--frames", default=None, help="real frames instead of synthetic ones: a raw / PGM file or a directory of them "
                                                   "(rows x cols 8-bit planes; pl-slam_amd/frames_io.py), cycled to fill the batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the N = 1 secondary / streaming / latency legs")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of the timed batch's records with the CPU oracle")
    ap.add_argument("--verify-frames", type=int, default=16, help="frames of EVERY sub-batch of the timed batch compared with the oracle (on every rank)")
    ap.add_argument("--verify-full", choices=["rotate", "none"], default="rotate",
                    help="additionally compare EVERY frame of one sub-batch of the timed batch (--verify-full-part says which)")
    ap.add_argument("--verify-full-part", type=int, default=-1,
                    help="the sub-batch compared in full: 0 .. nsplit-1 (on rank r: part + r, modulo nsplit); -1 (default) derives it from "
                         "--steps and --warmup, so that a command line always verifies the same part and CI cycles through all of them "
                         "by varying it; reported as verified.full_sub_batch")
    ap.add_argument("--force-dist", action="store_true", help=argparse.SUPPRESS)   # 1-rank RCCL communicator: rehearses the N > 1 path
    ap.add_argument("--strong-leg", action="store_true", help=argparse.SUPPRESS)   # with --force-dist --no-extras: only the configs[4] strong-scaling leg
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
    gathering = world > 1 or args.force_dist
    rccl_log = "/tmp/plslam_bench_rccl_%d.log" % os.getpid()
    if gathering:
        # keep stdout to the one JSON line: RCCL's log (topology and the transport of every channel) goes to a file,
        # which rank 0 summarises into the JSON afterwards
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH")
        os.environ["NCCL_DEBUG_FILE"] = rccl_log
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # barrier + max-over-ranks + id exchange

    strong = args.scaling == "strong"
    if strong:
        # the literal configs[4] job: a fixed set of frames, rank r owns the contiguous shard [r, r + 1) * total / N, one gather
        if args.total % world:
            raise SystemExit("--total %d is not a multiple of the %d ranks" % (args.total, world))
        args.batch = args.total // world
        if (args.rows, args.cols) == (376, 1241) and args.nfeatures == 1000:
            args.nfeatures = 2000          # KITTI00-02.yaml
        if args.nsplit == 4:               # (the default) sub-batches of at least 1024 frames
            args.nsplit = max(1, min(4, args.batch // 1024))
        while args.batch % args.nsplit:
            args.nsplit -= 1
    P, S = _util.plslam(), _util.synth()
    V = _util._load("plslam_amd_vocab", os.path.join(ROOT, "pl-slam_amd", "vocab.py"))
    PL = _util._load("plslam_amd_pipeline", os.path.join(ROOT, "pl-slam_amd", "pipeline.py"))
    voc = V.Vocabulary.synthetic(102, k=10, L=6, synth=S, idf=True)
    lib_default = int(P.load().plh_lsd_refine_default())
    refine = {"std": 0, "adv": 1, "default": lib_default}[args.refine]
    real = None
    if args.frames:
        FIO = _util._load("plslam_amd_frames_io", os.path.join(ROOT, "pl-slam_amd", "frames_io.py"))
        real = FIO.load_frames(args.frames, args.rows, args.cols)
        if (args.rows, args.cols) == (376, 1241) and args.nfeatures == 1000:
            args.nfeatures = 2000          # KITTI00-02.yaml
    W = Workload(P, S, V, PL, torch, dev, rank, args.batch, args.nsplit, args.rows, args.cols, args.nfeatures, args.nlevels, args.nlines,
                 args.unique, voc, serial=args.serial, shard=(rank, world, args.total) if strong else None, refine=refine,
                 screen=not args.no_screen, real=real, grow_waves=args.grow_waves)
    fe, B, Bp, rows, cols = W.fe, W.B, W.Bp, W.rows, W.cols

    comm = comm_stream = None
    recv = None
    if gathering:
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(P.Comm.unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(uid, 0)
        # RCCL prints a version banner on stdout when a communicator is created (NCCL_DEBUG >= VERSION): stdout is the one
        # JSON line's, so the descriptor points at the log file while ncclCommInitRank runs
        sys.stdout.flush()
        saved = os.dup(1)
        logfd = os.open(rccl_log + ".stdout", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        os.dup2(logfd, 1)
        try:
            comm = P.Comm(bytes(uid.cpu().numpy().tobytes()), rank, world, device=local_rank)
            import ctypes
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            os.close(logfd)
        comm_stream = torch.cuda.Stream(device=dev)
        root = 0 if args.gather == "root" else -1
        recv = fe.alloc_gather_buffers(world, receives=(root < 0 or rank == root))

    def step():
        # consecutive steps are independent batches and overlap (sub-batch pipelining).  N > 1: the fixed-stride records of
        # every sub-batch go over RCCL / xGMI on a communication stream as soon as that sub-batch is done (plh_gather_records,
        # one grouped launch); only the sub-batch's own next step waits for its gather
        fe.step(W.d_imgs, join=False)
        if gathering:
            fe.gather(comm_stream, comm, root, recv)

    box = box_probe(P, dev.index) if rank == 0 else None   # in front of the warm-up: untimed, and the GPU is idle around it
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    W.set_profiling(True)     # HIP events around the kernel groups: the roofline is measured over the timed region itself
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    host_ms = []
    for _ in range(args.steps):
        th = time.perf_counter()
        step()
        host_ms.append((time.perf_counter() - th) * 1e3)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    result_line = None
    failed_verification = []
    t1 = W.kernel_totals() if rank == 0 else None
    res = fe.results()   # what the timed steps left in the result buffers (before anything else runs over them)
    verified = None
    if not args.no_verify:
        # self-verification, on EVERY rank: what the timed kernels left in the result buffers, against the oracle on the same
        # frames -- the first --verify-frames frames of each sub-batch
        O = _util.oracle()
        O.build()
        # one sub-batch in full (--verify-full-part), 16 frames of the others
        # (round 5 picked the part from the wall clock: which frames a run verified was not reproducible -- ADVICE r5)
        full_part = None if args.verify_full == "none" else ((args.verify_full_part if args.verify_full_part >= 0 else args.steps + args.warmup) + rank) % W.nsplit
        verified = verify_batch(O, V, W, res, voc, args.verify_frames, full_part)
        if world > 1:   # a scaling run is a parity run: rank 0 reports every rank's verdict
            mine = {"rank": rank, "exact": verified["exact"], "frames": verified["frames"], "pairs": verified["pairs"],
                    "mismatches": verified["mismatches"]}
            allv = [None] * world
            dist.all_gather_object(allv, mine)
            verified = dict(verified, per_rank=allv, exact=all(v["exact"] for v in allv),
                            frames=sum(v["frames"] for v in allv), pairs=sum(v["pairs"] for v in allv))
    if verified is not None and not verified["exact"]:   # (every rank: a rank whose records differ exits non-zero as well)
        failed_verification.append("timed batch: %s" % (verified.get("per_rank") or verified["mismatches"]))
    if rank == 0:
        per_ms_timed = [ms / max(n, 1) for ms, n in t1]   # HIP events on the launch streams, over the timed region
        # one more pass with both halves on one stream: per-kernel durations without interference between the halves
        fe.overlap = False
        for _ in range(2):
            fe.step(W.d_imgs, join=True)
        torch.cuda.synchronize(dev)
        t2 = W.kernel_totals()
        per_ms = [(b[0] - a[0]) / max(b[1] - a[1], 1) for a, b in zip(t1, t2)]
        fe.overlap = not args.serial
        alg = W.algorithmic_bytes(res)
        dom = int(np.argmax(per_ms))
        # PMC traffic (FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_traffic.sh -> profiles/hbm_traffic.json), bytes per frame, and the
        # SQ instruction counters (tools/pmc_insts.sh -> profiles/sq_insts.json; round 5 wrote r05_insts.json): collected on the headline workload only
        headline = W.tum and args.nfeatures == 1000 and real is None
        # PMC figures are a property of a build: they are reported only when the file carries this library's build id
        build = P.load().plh_version().decode().split("build ")[-1].strip()
        tj, ij = load_profile_json("hbm_traffic.json"), (load_profile_json("sq_insts.json") or load_profile_json("r05_insts.json"))
        pmc_ok = headline and refine == lib_default and not args.no_screen   # (the profiles are collected at the library's default level)
        traffic = tj.get("kernels", {}) if pmc_ok and tj.get("build") == build else {}
        insts = ij.get("kernels", {}) if pmc_ok and ij.get("build") == build else {}
        pmc_note = {"library_build": build, "hbm_traffic.json": tj.get("build"), "insts.json": ij.get("build"),
                    "note": "PMC-derived fields (roofline.traffic, valu_issue) are null unless the profile was collected on this build"}
        pmc_names = {1: ["k_fast_strips"], 5: ["k_lsd_grow"], 0: ["k_pyr_down"], 2: ["k_octree"], 3: ["k_orient_brief"]}

        def roof(k, ms, where, frames_per_launch=Bp, algb=None):
            tr = []
            if algb is None:   # PMC traffic was collected on the headline workload only
                tr = [traffic[n]["total"] * traffic[n].get("launches_per_step", 1) for n in pmc_names.get(k, []) if n in traffic]
            algb = alg if algb is None else algb
            ach = algb[k] * frames_per_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"bound": "hbm", "kernel": NAMES[k], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": int(sum(tr) * frames_per_launch) if tr else None,
                    "algorithmic_bytes_per_launch": int(algb[k] * frames_per_launch), "frames_per_launch": frames_per_launch,
                    "ms_per_launch": round(ms, 4), "measured": where}

        r_dom = roof(dom, per_ms_timed[dom], "HIP events on the launch stream over the timed region")
        r_dom["ms_per_launch_alone"] = round(per_ms[dom], 4)
        if per_ms[dom] > 0:
            r_dom["frac_alone"] = round(alg[dom] * Bp / (per_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        # `frac` divides ONE launch's bytes by that launch's duration, and the launches of the sub-batches are resident together
        # (each the slower for it): the kernel's bytes over the whole step against the step time is the rate the chip sustains
        r_dom["all_launches_of_a_step"] = {
            "launches": args.nsplit, "achieved": round(alg[dom] * B / (dt / args.steps) / 1e9, 1), "unit": "GB/s",
            "frac": round(alg[dom] * B / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
            "what": "algorithmic bytes of the dominant kernel over a whole step / ms_per_step (lower bound of its share of the step)"}
        if NAMES[dom] in insts:
            # what the dominant kernel is actually bound by: the instructions its wavefronts issue.  All of a step's
            # launches of it over the step time, against the VALU issue peak; `wait_share` = s_waitcnt cycles / wave cycles (SQ counters)
            i_d = insts[NAMES[dom]]
            gi = i_d["valu"] * B / (dt / args.steps) / 1e9
            r_dom["valu_issue"] = {"achieved": round(gi, 1), "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instructions/s",
                                   "frac": round(gi / VALU_PEAK_GINST, 4), "valu_wave_instructions_per_frame": i_d["valu"],
                                   "salu_wave_instructions_per_frame": i_d.get("salu"),
                                   "wait_share_lone_wavefronts": round(i_d["wait_any"] / i_d["wave_cycles"], 3) if i_d.get("wave_cycles") else None,
                                   "what": "the kernel's VALU wave-instructions of a whole step / ms_per_step; the whole front end issues "
                                           "%.2f M VALU per frame" % (ij.get("total_valu", 0) / 1e6)}
        r_fast = roof(1, per_ms[1], "HIP events, extra pass after the timed region with both halves on one stream")
        if "k_fast_strips" in insts and per_ms[1] > 0:
            # the bound this kernel actually runs against: VALU issue: wave64 VALU instructions per second against
            # 1024 SIMDs x 2.4 GHz / 4.2 cycles
            gi = insts["k_fast_strips"]["valu"] * Bp / (per_ms[1] * 1e-3) / 1e9
            r_fast["valu_issue"] = {"achieved": round(gi, 1), "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instructions/s",
                                    "frac": round(gi / VALU_PEAK_GINST, 4), "valu_wave_instructions_per_frame": insts["k_fast_strips"]["valu"]}
        out = {
            "metric": "frames/s ORB+LSD extract+match, %dx%d mono" % (cols, rows),
            "value": round(world * B * args.steps / dt, 2), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "host_enqueue_ms_per_step": [round(v, 2) for v in host_ms],   # how far the host runs ahead of the GPU (launch queues)
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "real frames" if real is not None else "synthetic",
            "config": {"workload": "%dx%d mono, %d-level pyramid, %d ORB / %d lines (%s parameters), batch %d frames/GPU resident in HBM "
                                   "(%d sub-batches of %d pipelined, consecutive steps overlap, nothing crosses PCIe in the timed region); "
                                   "extract + ComputeBoW + SearchByBoW + line SearchDouble per consecutive frame pair"
                                   % (cols, rows, args.nlevels, args.nfeatures, args.nlines, "TUM1.yaml" if W.tum else "KITTI00-02.yaml",
                                      B, args.nsplit, Bp),
                       "lsd_refine": {"level": "LSD_REFINE_ADV" if refine else "LSD_REFINE_STD", "library_default": "LSD_REFINE_ADV" if lib_default else "LSD_REFINE_STD",
                                      "note": "src/LineExtractor.cpp:39-40 links the system opencv_contrib LSDDetector, which creates its detector with "
                                              "LSD_REFINE_ADV as published (no OpenCV binary in this image to check; INTEGRATION.md section 2); the "
                                              "un-linked twin in the reference's tree would run STD; the other level is measured under "
                                              "secondary.refine_%s" % ("std" if refine else "adv")},
                       "density_screen": not args.no_screen,
                       "mean_keypoints_per_frame": round(float(res["n"].mean()), 1), "mean_keylines_per_frame": round(float(res["nl"].mean()), 1),
                       "mean_orb_matches_per_pair": round(float(res["nm_orb"].mean()), 1),
                       "mean_line_matches_per_pair": round(float(res["nm_line"].mean()), 1),
                       "mean_bow_words_per_frame": round(float(res["bow_n"].mean()), 1),
                       "frames": ("%d real frames from %s, cycled to fill the batch" % (len(real), os.path.basename(args.frames.rstrip("/")))) if real is not None else
                                 "%d rasterised synthetic frames per GPU + cheap variants (row shift, exposure); busy by construction: "
                                 "~2/3 of the 0.8x-scaled pixels end up in LSD regions, ~10%% of the pyramid pixels are FAST corners at "
                                 "minThFAST -- a stress case, real TUM frames are sparser" % args.unique,
                       "vocabulary": "synthetic k=10 L=6, idf-like weights (ORBvoc.bin is not in the mount)",
                       "streams": "line chain on a high-priority stream, ORB + BoW + SearchByBoW on a second stream" if not args.serial else "one stream",
                       "parallelism": ("one job of %d frames, contiguous shards of %d per GPU" % (args.total, B) if strong else
                                       "frames sharded 1 batch/GPU") + (
                           ", records gathered %s through plh_gather_records (RCCL, one grouped launch per sub-batch on a communication "
                           "stream)" % ("to rank 0" if args.gather == "root" else "to every rank") if gathering else "")},
            "kernel_ms_per_launch": {NAMES[k]: round(per_ms[k], 4) for k in range(8)},
            "kernel_ms_per_launch_timed_region": {NAMES[k]: round(per_ms_timed[k], 4) for k in range(8)},
            "roofline": r_dom,
            "roofline_fast": r_fast,
            "pmc_provenance": pmc_note,
            "front_end_valu_issue": ({"achieved": round(ij["total_valu"] * world * B * args.steps / dt / 1e9, 1), "peak": round(VALU_PEAK_GINST, 1),
                                      "unit": "G wave-instructions/s", "frac": round(ij["total_valu"] * B * args.steps / dt / 1e9 / VALU_PEAK_GINST, 4),
                                      "valu_wave_instructions_per_frame": ij["total_valu"],
                                      "what": "all kernels' VALU wave-instructions per frame (SQ counters of this build) x frames/s per GPU "
                                              "against 1024 SIMDs x 2.4 GHz / 4.2 cycles: how close the front end runs to its own instruction floor"}
                                     if insts and ij.get("total_valu") else None),
            "front_end_traffic": (lambda tot: {
                "pmc_bytes_per_frame": int(tot), "algorithmic_bytes_per_frame": int(sum(alg)), "ratio": round(tot / max(sum(alg), 1), 3),
                "achieved": round(tot * B * args.steps / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(tot * B * args.steps / dt / 1e9 / HBM_PEAK_GBS, 4),
                "what": "memory-side bytes of ALL kernels of a step per frame (PMC FETCH_SIZE / WRITE_SIZE of this build, "
                        "profiles/hbm_traffic.json) against the algorithmic bytes of the stages (DESIGN.md 3), and that traffic at the "
                        "measured rate against the HBM peak"})(
                    sum(v["total"] * v.get("launches_per_step", 1) for v in traffic.values())) if traffic else None,
            "verified": verified,
            "box": box,
        }
        if gathering:
            tr = {}
            try:
                import glob
                import re
                for f in glob.glob(rccl_log + "*") + glob.glob("/tmp/plslam_bench_rccl_*.log*"):
                    for ln in open(f, errors="ignore"):
                        m = re.search(r"via (P2P/\S+|SHM\S*|NET/\S+|direct\S*)", ln)
                        if m:
                            tr[m.group(1)] = tr.get(m.group(1), 0) + 1
                        if "XGMI" in ln.upper() and "xgmi_lines" not in tr:
                            tr["xgmi_lines"] = ln.strip()[-160:]
            except Exception:
                pass
            out["rccl"] = {"version": comm.rccl_version(), "gather": args.gather, "channel_transports": tr,
                           "bytes_per_rank_per_step": int(sum(b.numel() * b.element_size() for part in recv["send"] for b in part))}
        W.set_profiling(False)

    # ---- N = 1 extras (rank 0 only, after the timed region; none of this is in `value`)
    if rank == 0 and world == 1 and not args.no_extras:
        extras_t0 = time.perf_counter()
        # streaming: fresh frames uploaded from pinned host memory and every record downloaded, each step joined
        try:
            host = torch.from_numpy(W.frames).pin_memory()
            d_alt = [torch.empty_like(W.d_imgs), torch.empty_like(W.d_imgs)]
            copy_s = torch.cuda.Stream(device=dev)
            ev = [torch.cuda.Event(), torch.cuda.Event()]
            keys = ("n", "kps", "desc", "nid", "bow_n", "bow_word", "bow_value", "nl", "kl", "ldesc", "lfn", "m_orb", "nm_orb", "m_line", "nm_line")
            pinned = [{k: torch.empty_like(getattr(p, k)[:p.B], device="cpu").pin_memory() for k in keys} for p in fe.parts]

            def upload(i):
                with torch.cuda.stream(copy_s):
                    d_alt[i & 1].copy_(host, non_blocking=True)
                    ev[i & 1].record(copy_s)

            nst = 3
            upload(0)
            torch.cuda.synchronize(dev)
            ts = time.perf_counter()
            for i in range(nst):
                if i + 1 < nst:
                    upload(i + 1)                       # next batch crosses PCIe underneath this step's kernels
                torch.cuda.current_stream(dev).wait_event(ev[i & 1])
                fe.step(d_alt[i & 1], join=True)
                for p, dst in zip(fe.parts, pinned):    # D2H of everything one step produced
                    for k in keys:
                        dst[k].copy_(getattr(p, k)[:p.B], non_blocking=True)
                torch.cuda.synchronize(dev)
            tstream = (time.perf_counter() - ts) / nst
            up = W.frames.nbytes
            down = sum(v.numel() * v.element_size() for d in pinned for v in d.values())
            out["streaming"] = {"value": round(B / tstream, 1), "unit": "frames/s", "ms_per_step": round(tstream * 1e3, 2), "steps": nst,
                                "h2d_bytes_per_step": int(up), "d2h_bytes_per_step": int(down),
                                "what": "per step: %d fresh frames H2D from pinned memory (overlapped with the previous step's kernels), one "
                                        "joined pass, all records D2H; not the contract's `value`" % B}
            del host, d_alt, pinned
        except Exception as e:   # the extras never take the headline down
            out["streaming"] = {"error": repr(e)[:200]}
        lat_frames = W.frames[:8].copy()
        W.close()
        fe = None
        torch.cuda.empty_cache()
        try:
            out["latency_ms_single_frame"] = single_frame_latency(P, torch, dev, lat_frames, args.nfeatures, args.nlevels, args.nlines,
                                                                  TUM1_K if W.tum else None, TUM1_D if W.tum else None, refine=refine)
        except Exception as e:
            out["latency_ms_single_frame"] = {"error": repr(e)[:200]}
        if headline and isinstance(out["latency_ms_single_frame"], dict) and "error" not in out["latency_ms_single_frame"]:
            # ... and the other frame shape the north star names, at both refine levels
            try:
                kf = S.make_frames(2, 4, 376, 1241, unique=4)
                for lvl, name in ((refine, "kitti_1241x376"), (1 - refine, "kitti_1241x376_refine_%s" % ("std" if refine else "adv"))):
                    r2 = single_frame_latency(P, torch, dev, kf, 2000, 8, 200, None, None, reps=6, refine=lvl)
                    out["latency_ms_single_frame"][name] = {k: v for k, v in r2.items() if k != "note"}
            except Exception as e:
                out["latency_ms_single_frame"]["kitti_1241x376"] = {"error": repr(e)[:200]}
        # secondary workload: KITTI 1241x376 / 2000 features (BASELINE configs[4]'s frame shape)
        if headline:
            try:
                sec = {}
                def leg(b2, ns2, r2, c2, nf2, uniq2, refine2, label, frames=None):
                    """One resident-batch leg after the timed region: rate, verified, per-kernel rooflines."""
                    W2 = Workload(P, S, V, PL, torch, dev, rank, b2, ns2, r2, c2, nf2, 8, 200, uniq2, voc, refine=refine2, screen=not args.no_screen,
                                  real=frames)
                    n2 = 3 if b2 > 1024 else 8   # the small share needs a few steps to reach its steady state
                    dt2 = W2.run(n2, 1 if b2 > 1024 else 2)
                    W2.set_profiling(True)
                    W2.fe.overlap = False
                    for _ in range(2):
                        W2.fe.step(W2.d_imgs, join=True)
                    torch.cuda.synchronize(dev)
                    tk = W2.kernel_totals()
                    pm2 = [ms / max(n, 1) for ms, n in tk]
                    res2 = W2.fe.results()
                    ver2 = None
                    if not args.no_verify:
                        ver2 = verify_batch(_util.oracle(), V, W2, res2, voc, args.verify_frames)
                        if not ver2["exact"]:
                            failed_verification.append("secondary %s: %s" % (label, ver2["mismatches"]))
                    alg2 = W2.algorithmic_bytes(res2)
                    d2 = int(np.argmax(pm2))
                    r = {"value": round(b2 * n2 / dt2, 1), "ms_per_step": round(dt2 / n2 * 1e3, 3), "steps": n2, "batch": b2, "nsplit": ns2,
                         "mean_keypoints_per_frame": round(float(res2["n"].mean()), 1), "mean_keylines_per_frame": round(float(res2["nl"].mean()), 1),
                         "verified": ver2,
                         "roofline": roof(d2, pm2[d2], "HIP events, extra pass with both halves on one stream", W2.Bp, alg2),
                         "roofline_fast": roof(1, pm2[1], "HIP events, extra pass with both halves on one stream", W2.Bp, alg2),
                         "kernel_ms_per_launch": {NAMES[k]: round(pm2[k], 4) for k in range(8)}}
                    W2.close()
                    del W2
                    torch.cuda.empty_cache()
                    return r

                sec = {}
                for label, b2, ns2 in (("resident_6144", 6144, 4), ("configs4_share_512", 512, 1)):
                    sec[label] = leg(b2, ns2, 376, 1241, 2000, 16, refine, label)
                # the other refine level of cv::LineSegmentDetector on the headline workload (which one the reference's OpenCV runs is
                # a property of that build: INTEGRATION.md section 2)
                other = 1 - refine
                oname = "refine_adv" if other else "refine_std"
                ro = {"level": "LSD_REFINE_ADV" if other else "LSD_REFINE_STD", "unit": "frames/s",
                      "resident_6144": leg(6144, 4, 480, 640, 1000, args.unique, other, oname + " resident_6144"),
                      "share_512": leg(512, 1, 480, 640, 1000, args.unique, other, oname + " share_512")}
                ro["value"] = ro["resident_6144"]["value"]
                ro["vs_headline"] = round(ro["value"] / out["value"], 3)
                ro["verified"] = ro["resident_6144"]["verified"]
                ro["roofline"] = ro["resident_6144"]["roofline"]
                try:
                    ro["latency_ms_single_frame"] = single_frame_latency(P, torch, dev, lat_frames, args.nfeatures, args.nlevels, args.nlines,
                                                                         TUM1_K, TUM1_D, reps=8, refine=other)
                except Exception as e:
                    ro["latency_ms_single_frame"] = {"error": repr(e)[:200]}
                out["secondary"] = {"workload": "1241x376 mono (KITTI00-02.yaml: 2000 ORB, 8 levels, no distortion) + 200 lines, same pipeline; "
                                                "BASELINE configs[4] shards 4096 such frames over 8 GPUs = 512 per GPU",
                                    "unit": "frames/s", "value": sec["resident_6144"]["value"], "ms_per_step": sec["resident_6144"]["ms_per_step"],
                                    "roofline": sec["resident_6144"]["roofline"], "resident_6144": sec["resident_6144"],
                                    "configs4_share_512": sec["configs4_share_512"], oname: ro,
                                    "note": "512 resident frames cannot fill the GPU with one wavefront per frame, so region growing runs 8 "
                                            "wavefronts per frame there (k_lsd_grow_mw, same segments): the per-GPU rate of the literal configs[4] "
                                            "job is the configs4_share_512 figure, the 6144-frame one is what a GPU sustains on a long sequence"}
                # the N = 1 point of the strong-scaling configs[4] job (at N > 1 the same field is the sharded, gathered job, below)
                try:
                    s1 = leg(4096, 4, 376, 1241, 2000, 16, refine, "configs4_strong (N = 1)")
                    out["secondary"]["configs4_strong"] = {
                        "value": s1["value"], "unit": "frames/s", "scaling": "strong", "n_gpus": 1, "total_frames": 4096, "frames_per_gpu": 4096,
                        "sub_batches": 4, "steps": s1["steps"], "ms_per_step": s1["ms_per_step"], "gather": None, "verified": s1["verified"],
                        "workload": "BASELINE configs[4]: 4096 frames of 1241x376 (KITTI00-02.yaml: 2000 ORB, 8 levels) + 200 lines as one job; "
                                    "one GPU: no gather (with --gpus N the same field holds the sharded job with its RCCL gather)"}
                except Exception as e:
                    out["secondary"]["configs4_strong"] = {"error": repr(e)[:300]}
                # the headline workload on 1024 DISTINCT frames (cycled six times to the same 6144-frame batch): 1024 control-flow traces of
                # region growing / FAST instead of the timed batch's 32
                try:
                    t0d = time.perf_counter()
                    dfr = distinct_frames(S, args.rows, args.cols, 1024)
                    gen_s = time.perf_counter() - t0d
                    dl = leg(6144, 4, args.rows, args.cols, args.nfeatures, 1024, refine, "distinct_frames", frames=dfr)
                    dl["vs_headline"] = round(dl["value"] / out["value"], 3)
                    dl["frames"] = ("1024 distinct synthetic frames (128 rasterised scenes of random density x 8 views each: shifted, contrast / "
                                    "exposure changed, Gaussian sensor noise of sigma 0.5 .. 6 added), cycled six times to fill the batch; "
                                    "generated in %.1f s" % gen_s)
                    out["secondary"]["distinct_frames"] = dl
                    del dfr
                except Exception as e:
                    out["secondary"]["distinct_frames"] = {"error": repr(e)[:300]}
            except Exception as e:
                out["secondary"] = {"error": repr(e)[:300]}
        # the tracker's per-frame searches (TrackWithMotionModel + SearchLocalPoints / SearchLocalLines) as a resident batch: what a
        # running tracker calls for EVERY frame (SearchByBoW, which the headline times, only after a keyframe or a loss)
        if headline:
            try:
                TB = _util._load("plslam_tracking_bench", os.path.join(ROOT, "tools", "tracking_bench.py"))
                tr = TB.run(pairs=1024, distinct=32, steps=5, warmup=1, quiet=True)
                if not args.no_verify and not tr["verified"]["exact"]:
                    failed_verification.append("secondary tracking: the searches differ from the oracle chain")
                out.setdefault("secondary", {})["tracking"] = tr
            except Exception as e:
                out.setdefault("secondary", {})["tracking"] = {"error": repr(e)[:300]}
        out["extras_seconds"] = round(time.perf_counter() - extras_t0, 1)
    else:
        W.close()

    # ---- N > 1: the literal BASELINE configs[4] job in the same line as the weak-scaling headline: 4096 frames of
    # 1241x376 / 2000 ORB / 200 lines as ONE job, rank r owns the contiguous shard [r, r + 1) * 4096 / N, records gathered to rank 0.
    # A curve over N from `value` alone is linear by construction (every rank brings its own 6144 frames); this one is strong scaling.
    if (world > 1 or (args.force_dist and (args.strong_leg or not args.no_extras))) and not strong and real is None:
        sleg = None
        try:
            tot = 4096
            b2 = tot // world
            ns2 = max(1, min(4, b2 // 1024))
            while b2 % ns2:
                ns2 -= 1
            W2, recv2, err2 = None, None, None
            try:
                W2 = Workload(P, S, V, PL, torch, dev, rank, b2, ns2, 376, 1241, 2000, 8, 200, 16, voc, shard=(rank, world, tot), refine=refine,
                              screen=not args.no_screen)
                recv2 = W2.fe.alloc_gather_buffers(world, receives=(root < 0 or rank == root))
            except Exception as e:   # noqa: BLE001
                err2 = repr(e)[:300]
            if world > 1:   # every rank enters the collectives below, or none does
                flag = torch.tensor([0 if err2 is None else 1], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if int(flag.item()) and err2 is None:
                    err2 = "another rank could not set the job up"
            if err2 is not None:
                if W2 is not None:
                    W2.close()
                raise RuntimeError(err2)

            def step2():
                W2.fe.step(W2.d_imgs, join=False)
                W2.fe.gather(comm_stream, comm, root, recv2)
            for _ in range(2):
                step2()
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            ts = time.perf_counter()
            n2 = 6
            for _ in range(n2):
                step2()
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            dt2 = time.perf_counter() - ts
            if world > 1:
                t = torch.tensor([dt2], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt2 = float(t.item())
            ver2 = None
            if not args.no_verify:
                ver2 = verify_batch(_util.oracle(), V, W2, W2.fe.results(), voc, 4)
                ok2 = ver2["exact"]
                if world > 1:
                    allok = [None] * world
                    dist.all_gather_object(allok, bool(ok2))
                    ok2 = all(allok)
                if not ok2:
                    failed_verification.append("configs4_strong: records differ from the oracle")
                ver2 = {"frames_per_rank": ver2["frames"], "exact": bool(ok2)}
            sleg = {"value": round(tot * n2 / dt2, 1), "unit": "frames/s", "scaling": "strong", "n_gpus": world, "total_frames": tot,
                    "frames_per_gpu": b2, "sub_batches": ns2, "steps": n2, "ms_per_step": round(dt2 / n2 * 1e3, 3), "gather": args.gather,
                    "verified": ver2,
                    "workload": "BASELINE configs[4]: 4096 frames of 1241x376 (KITTI00-02.yaml: 2000 ORB, 8 levels) + 200 lines as one job, "
                                "contiguous shards, records gathered over RCCL"}
            W2.close()
        except Exception as e:
            sleg = {"error": repr(e)[:300]}
        if rank == 0:
            out.setdefault("secondary", {})["configs4_strong"] = sleg

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only (rank 0)
            O = _util.oracle()
            O.build()
            fr = real[:64] if real is not None else S.make_frames(2, 64, rows, cols, unique=min(args.unique, 64))
            out["cpu_baseline"] = cpu_baseline(O, V, fr, voc, args.nfeatures, args.nlevels, args.nlines,
                                               TUM1_K if (rows, cols) == (480, 640) else KITTI_K,
                                               TUM1_D if (rows, cols) == (480, 640) else [0, 0, 0, 0, 0], refine=refine)
        result_line = json.dumps(out)
    if comm is not None:
        comm.close()
    try:   # flush what native libraries buffered on C stdio, on every rank, BEFORE the result line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(result_line, flush=True)   # the one JSON line, last on stdout
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if failed_verification:   # a fast kernel whose results differ from the reference's is not done
        sys.stderr.write("bench.py: records differ from the oracle: %s\n" % "; ".join(failed_verification))
        sys.exit(1)


import ctypes as _C  # noqa: E402
C_VOID, C_INT = _C.c_void_p, _C.c_int

if __name__ == "__main__":
    main()
