#!/usr/bin/env python3
"""Throughput of the tracker's PER-FRAME searches on the GPU -- what a running tracker calls for every frame, against which round 5 had
no timing of any kind:

    TrackWithMotionModel (src/Tracking.cc:1321-1357)
        ORBmatcher(0.9, true).SearchByProjection(Cur, Last, th, mono)   src/ORBmatcher.cc:1441-1585
            = plh_frame_project_points (form 0)  ->  plh_orb_search_by_projection_frame
        LSDmatcher().SearchByProjection(Cur, Last, th)                  src/LSDmatcher.cpp:72-176      (isInFrustum per last-frame MapLine)
            = plh_frame_is_in_frustum_lines      ->  plh_line_search_by_projection_frame
    SearchLocalPoints / SearchLocalLines (src/Tracking.cc:1792-1855), Frame::isInFrustum src/Frame.cc:560-711
        ORBmatcher(0.8).SearchByProjection(F, local MapPoints, th)      src/ORBmatcher.cc:56-144
            = plh_frame_is_in_frustum_points     ->  plh_orb_search_by_projection_mp
        LSDmatcher().SearchByProjection(F, local MapLines, th)          src/LSDmatcher.cpp:221-338
            = plh_frame_is_in_frustum_lines      ->  plh_line_search_by_projection_ml

A batch of `pairs` independent (current frame, last frame, local map) triples -- `distinct` different ones, tiled -- stays on the device:
frames (1000 keypoints, 200 lines, grids built once: a Frame's are built in its constructor), last-frame map points, a local map of
6000 points / 600 lines.  A step runs the eight calls above for the whole batch; time = HIP events on the launch stream.  The first
`distinct` triples are checked against the oracle chain (oracle/frame_search.cc) after the timed region; cpu_baseline = that oracle
chain on one host thread.

    python tools/tracking_bench.py [--pairs 1024] [--steps 10] [--json]"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(ROOT, "tools", "gen_golden_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def tile(d, k, keys=None):
    return {key: np.ascontiguousarray(np.concatenate([v] * k)) for key, v in d.items() if keys is None or key in keys}


def run(pairs=1024, distinct=32, steps=10, warmup=2, n=1000, nl=200, mapk=6, linek=3, cpu_frames=8, lib=None, quiet=False):
    import torch
    G, S, P, O = _gen(), _util.synth(), _util.plslam(), _util.oracle()
    O.build()
    TF, FR, TR = G._test_module("test_frame_search"), G._test_module("test_frustum"), G._test_module("test_ref_track")
    L = P.load(lib)
    dev = torch.device("cuda", 0)
    cases = []
    for d in range(distinct):
        f2, gp, view, nlv, pts, lns, occ_p, occ_l = G.track_inputs(S, P, TF, 2000 + d, n, nl, False)
        flags, q = G.track_last_inputs(S, P, TF, 2000 + d, n, nl, False)
        cases.append(dict(f2=f2, gp=gp, view=view, nlv=nlv, pts=pts, lns=lns, occ_p=occ_p, occ_l=occ_l, flags=flags, q=q,
                          mapp=tile(pts, mapk), mapl=tile(lns, linek)))
    gp, nlv = cases[0]["gp"], cases[0]["nlv"]
    reps = (pairs + distinct - 1) // distinct

    def stack(fn, dtype, width=None):
        a = [np.ascontiguousarray(fn(c)) for c in cases]
        cap = max(len(x) for x in a)
        out = np.zeros((distinct, cap) + (() if width is None else (width,)), dtype)
        for i, x in enumerate(a):
            out[i, :len(x)] = x.reshape((len(x),) + (() if width is None else (width,)))
        return out, np.array([len(x) for x in a], np.int32)

    def dput(a):
        t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(a.shape[0], -1) if a.dtype.names else np.ascontiguousarray(a)).to(dev)
        return t.repeat((reps,) + (1,) * (t.dim() - 1))[:pairs].contiguous()

    p = P._p
    # ---- the frames (current): keypoints, descriptors, lines, grids (built once, as the Frame constructor does)
    kps, nk = stack(lambda c: c["f2"]["kps"], P.KP_DTYPE)
    cap = kps.shape[1]
    d_kps, d_nk = dput(kps), dput(nk)
    d_desc = dput(stack(lambda c: c["f2"]["desc"], np.uint8, 32)[0])
    kl, nkl = stack(lambda c: c["f2"]["keylines"], P.KL_DTYPE)
    lcap = kl.shape[1]
    d_kl, d_nkl = dput(kl), dput(nkl)
    d_ld = dput(stack(lambda c: c["f2"]["ldesc"], np.uint8, 32)[0])
    d_fn = dput(stack(lambda c: c["f2"]["linefn"], np.float64, 3)[0])
    item_cap = lcap * 64
    d_cs, d_ci = torch.zeros((pairs, 64 * 48 + 1), dtype=torch.int32, device=dev), torch.zeros((pairs, cap), dtype=torch.int32, device=dev)
    d_lcs, d_lci = torch.zeros((pairs, 64 * 48 + 1), dtype=torch.int32, device=dev), torch.zeros((pairs, item_cap), dtype=torch.int32, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P._check(L, L.plh_frame_assign_grid_batch_dev(p(d_kps), p(d_nk), cap, pairs, C.byref(gp), p(d_cs), p(d_ci), s), "grid")
    P._check(L, L.plh_frame_assign_grid_lines_batch_dev(p(d_kl), p(d_nkl), lcap, pairs, C.byref(gp), p(d_lcs), p(d_lci), item_cap, s), "line grid")
    views = np.array([FR._view_record(P, c["view"], c["nlv"]) for c in cases], P.VIEW_DTYPE)
    d_views = dput(views.view(np.uint8).reshape(distinct, P.VIEW_DTYPE.itemsize))
    SC = TF.SCALE

    def elems(key, lines):
        w = 6 if lines else 3
        pos, nq = stack(lambda c: c[key]["pos"].reshape(-1, w), np.float32, w)
        return dict(nq=dput(nq), qcap=pos.shape[1], pos=dput(pos), normal=dput(stack(lambda c: c[key]["normal"], np.float32, 3)[0]),
                    mind=dput(stack(lambda c: c[key]["min_dist"], np.float32)[0]), maxd=dput(stack(lambda c: c[key]["max_dist"], np.float32)[0]),
                    desc=dput(stack(lambda c: c[key]["desc"], np.uint8, 32)[0]), hasobs=dput(stack(lambda c: c[key]["hasobs"], np.uint8)[0]))

    last_p, last_l, map_p, map_l = elems("pts", False), elems("lns", True), elems("mapp", False), elems("mapl", True)
    d_flag = dput(stack(lambda c: (c["flags"]["mp"] & (1 - c["flags"]["outlier"])).astype(np.uint8), np.uint8)[0])
    d_oct = dput(stack(lambda c: c["q"]["octave"], np.int32)[0])
    d_ang = dput(stack(lambda c: c["q"]["angle"], np.float32)[0])
    # last frame's keylines carry the query length of the line search (LastFrame.mvKeylinesUn[i].lineLength): frame 1 of the pair
    f1s = [TF.make_frame_pair(P, S, 2000 + d, n, nl=nl)[0] for d in range(distinct)]
    d_len = dput(np.stack([np.pad(f["keylines"]["lineLength"].astype(np.float32), (0, last_l["qcap"] - len(f["keylines"]))) for f in f1s]))
    occ_p0, occ_l0 = dput(stack(lambda c: c["occ_p"], np.uint8)[0]), dput(stack(lambda c: c["occ_l"], np.uint8)[0])
    if occ_p0.shape[1] < cap:
        occ_p0 = torch.nn.functional.pad(occ_p0, (0, cap - occ_p0.shape[1]))
    if occ_l0.shape[1] < lcap:
        occ_l0 = torch.nn.functional.pad(occ_l0, (0, lcap - occ_l0.shape[1]))
    occ_p, occ_l = occ_p0.clone(), occ_l0.clone()

    def z(shape, dt):
        return torch.zeros(shape, dtype=dt, device=dev)
    qp = last_p["qcap"]
    front, uv = z((pairs, qp), torch.uint8), z((pairs, qp, 2), torch.float32)
    asg_m, cnt_m = z((pairs, cap), torch.int32), z((pairs,), torch.int32)
    ql = last_l["qcap"]
    lv, lseg, llev, lvc = z((pairs, ql), torch.uint8), z((pairs, ql, 4), torch.float32), z((pairs, ql), torch.int32), z((pairs, ql), torch.float32)
    asg_ll, cnt_ll = z((pairs, lcap), torch.int32), z((pairs,), torch.int32)
    mp = map_p["qcap"]
    pv, puv, plev, pvc = z((pairs, mp), torch.uint8), z((pairs, mp, 2), torch.float32), z((pairs, mp), torch.int32), z((pairs, mp), torch.float32)
    asg_p, cnt_p = z((pairs, cap), torch.int32), z((pairs,), torch.int32)
    ml = map_l["qcap"]
    mv, mseg, mlev, mvc = z((pairs, ml), torch.uint8), z((pairs, ml, 4), torch.float32), z((pairs, ml), torch.int32), z((pairs, ml), torch.float32)
    asg_l, cnt_l = z((pairs, lcap), torch.int32), z((pairs,), torch.int32)
    valid_m = z((pairs, qp), torch.uint8)
    ones_l = torch.ones((pairs, ql), dtype=torch.uint8, device=dev)
    sf = np.ascontiguousarray(SC, np.float32)
    F = C.c_float
    marks = {}

    def stage(name, fn, ev):
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        fn()
        if ev is not None:
            e1.record()
            ev.append((name, e0, e1))

    def step(ev=None):
        occ_p.copy_(occ_p0); occ_l.copy_(occ_l0)
        stage("k_project_points", lambda: P._check(L, L.plh_frame_project_points_batch_dev(p(d_views), pairs, p(last_p["nq"]), qp, p(last_p["pos"]), 0, p(front), p(uv), s), "project"), ev)
        torch.bitwise_and(front, d_flag, out=valid_m)
        stage("k_search_proj_points (Cur, Last)", lambda: P._check(L, L.plh_orb_search_by_projection_frame_batch_dev(
            p(d_kps), p(d_desc), p(d_nk), cap, pairs, C.byref(gp), p(d_cs), p(d_ci), p(sf), len(sf), p(occ_p), p(last_p["nq"]), qp, p(valid_m), p(uv),
            p(d_oct), p(d_ang), p(last_p["desc"]), p(last_p["hasobs"]), F(15.0), 0, 1, p(asg_m), p(cnt_m), s), "proj frame"), ev)
        stage("k_frustum_lines (last frame)", lambda: P._check(L, L.plh_frame_is_in_frustum_lines_batch_dev(
            p(d_views), pairs, p(last_l["nq"]), ql, p(last_l["pos"]), p(last_l["normal"]), p(last_l["mind"]), p(last_l["maxd"]), F(0.5), p(lv), p(lseg),
            p(llev), p(lvc), s), "frustum lines"), ev)
        stage("k_search_proj_lines (Cur, Last)", lambda: P._check(L, L.plh_line_search_by_projection_frame_batch_dev(
            p(d_kl), p(d_ld), p(d_fn), p(d_nkl), lcap, pairs, C.byref(gp), p(d_lcs), p(d_lci), item_cap, p(occ_l), p(last_l["nq"]), ql, p(lv), p(lseg),
            p(d_len), p(last_l["desc"]), p(last_l["hasobs"]), F(12.0), p(asg_ll), p(cnt_ll), s), "line frame"), ev)
        stage("k_frustum_points (local map)", lambda: P._check(L, L.plh_frame_is_in_frustum_points_batch_dev(
            p(d_views), pairs, p(map_p["nq"]), mp, p(map_p["pos"]), p(map_p["normal"]), p(map_p["mind"]), p(map_p["maxd"]), F(0.5), p(pv), p(puv),
            p(plev), p(pvc), s), "frustum points"), ev)
        stage("k_search_proj_points (F, MapPoints)", lambda: P._check(L, L.plh_orb_search_by_projection_mp_batch_dev(
            p(d_kps), p(d_desc), p(d_nk), cap, pairs, C.byref(gp), p(d_cs), p(d_ci), p(sf), len(sf), p(occ_p), p(map_p["nq"]), mp, p(pv), p(puv),
            p(plev), p(pvc), p(map_p["desc"]), p(map_p["hasobs"]), F(3.0), F(0.8), p(asg_p), p(cnt_p), s), "proj mp"), ev)
        stage("k_frustum_lines (local map)", lambda: P._check(L, L.plh_frame_is_in_frustum_lines_batch_dev(
            p(d_views), pairs, p(map_l["nq"]), ml, p(map_l["pos"]), p(map_l["normal"]), p(map_l["mind"]), p(map_l["maxd"]), F(0.5), p(mv), p(mseg),
            p(mlev), p(mvc), s), "frustum lines"), ev)
        stage("k_search_proj_lines (F, MapLines)", lambda: P._check(L, L.plh_line_search_by_projection_ml_batch_dev(
            p(d_kl), p(d_ld), p(d_fn), p(d_nkl), lcap, pairs, C.byref(gp), p(d_lcs), p(d_lci), item_cap, p(occ_l), p(map_l["nq"]), ml, p(mv), p(mseg),
            p(mvc), p(map_l["desc"]), p(map_l["hasobs"]), F(3.0), F(0.7), p(asg_l), p(cnt_l), s), "line ml"), ev)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    ev = []
    step(ev)
    torch.cuda.synchronize()
    kernels = {name: round(a.elapsed_time(b), 4) for name, a, b in ev}
    # ---- the first `distinct` triples against the oracle chain; the same chain timed on one host thread = cpu_baseline
    got = dict(m=(cnt_m.cpu().numpy(), asg_m.cpu().numpy()), ll=(cnt_ll.cpu().numpy(), asg_ll.cpu().numpy()),
               p=(cnt_p.cpu().numpy(), asg_p.cpu().numpy()), l=(cnt_l.cpu().numpy(), asg_l.cpu().numpy()))
    Lo = TF._olib(O)
    exact, t_cpu, ncpu, nmatch = True, 0.0, 0, [0, 0, 0, 0]
    for d, c in enumerate(cases[:max(cpu_frames, 1)] if quiet else cases):
        f2, view, pts, lns = c["f2"], c["view"], c["pts"], c["lns"]
        nn, nnl, g = len(f2["kps"]), len(f2["keylines"]), TF._gpa(P, gp)
        (cs, ci), (lcs, lci) = TF._oracle_grids(O, P, f2, gp)       # (the Frame constructor's work: not part of the timed chain)
        po = O._p
        t0 = time.perf_counter()
        cm, am, om = TR._motion_oracle(O, P, TF, G, f2, gp, view, nlv, pts, c["flags"], c["q"], c["occ_p"], 15.0)
        valid, seg, level, vc = (np.ascontiguousarray(a) for a in FR._oracle(O, view, nlv, lns, 1, 0.5))
        ol, al = c["occ_l"].copy(), np.zeros(max(nnl, 1), np.int32)
        ln = np.ascontiguousarray(f1s[d]["keylines"]["lineLength"].astype(np.float32))
        cll = Lo.plo_line_search_by_projection_frame(po(f2["keylines"]), po(f2["ldesc"]), po(f2["linefn"]), nnl, po(g), po(lcs), po(lci), po(ol), len(valid),
                                                     po(valid), po(seg), po(ln), po(lns["desc"]), po(lns["hasobs"]), 12.0, po(al))
        valid2, uv2, level2, vc2 = (np.ascontiguousarray(a) for a in FR._oracle(O, view, nlv, c["mapp"], 0, 0.5))
        op, ap = om.copy(), np.zeros(max(nn, 1), np.int32)
        cp = Lo.plo_orb_search_by_projection_mp(po(f2["kps"]), po(f2["desc"]), nn, po(g), po(cs), po(ci), po(TF.SCALE), po(op), len(valid2), po(valid2),
                                                po(uv2), po(level2), po(vc2), po(c["mapp"]["desc"]), po(c["mapp"]["hasobs"]), 3.0, 0.8, po(ap))
        valid3, seg3, level3, vc3 = (np.ascontiguousarray(a) for a in FR._oracle(O, view, nlv, c["mapl"], 1, 0.5))
        ol2, al2 = ol.copy(), np.zeros(max(nnl, 1), np.int32)
        cl = Lo.plo_line_search_by_projection_ml(po(f2["keylines"]), po(f2["ldesc"]), po(f2["linefn"]), nnl, po(g), po(lcs), po(lci), po(ol2), len(valid3),
                                                 po(valid3), po(seg3), po(vc3), po(c["mapl"]["desc"]), po(c["mapl"]["hasobs"]), 3.0, 0.7, po(al2))
        if d < cpu_frames:
            t_cpu += time.perf_counter() - t0
            ncpu += 1
        ok = (cm == got["m"][0][d] and (am == got["m"][1][d, :nn]).all() and cll == got["ll"][0][d] and (al[:nnl] == got["ll"][1][d, :nnl]).all()
              and cp == got["p"][0][d] and (ap[:nn] == got["p"][1][d, :nn]).all() and cl == got["l"][0][d] and (al2[:nnl] == got["l"][1][d, :nnl]).all())
        exact = exact and bool(ok)
        for k, v in enumerate((cm, cll, cp, cl)):
            nmatch[k] += int(v)
    # copies of a triple give the same answers
    rep_ok = bool((asg_p[:distinct] == asg_p[distinct:2 * distinct]).all().item()) if pairs >= 2 * distinct else True
    nchk = len(cases[:max(cpu_frames, 1)] if quiet else cases)
    out = dict(metric="frames/s of the tracker's per-frame searches (TrackWithMotionModel + SearchLocalPoints / SearchLocalLines), resident batch",
               value=round(pairs / (ms / 1e3), 1), unit="frames/s", ms_per_step=round(ms, 3), pairs=pairs, distinct=distinct, steps=steps,
               workload="%d keypoints / %d lines per frame, last frame %d map points / %d map lines, local map %d points / %d lines" % (n, nl, qp, ql, mp, ml),
               kernel_ms_per_launch=kernels,
               verified=dict(frames=nchk, exact=bool(exact and rep_ok), matches_per_frame=dict(motion_points=nmatch[0] / nchk, motion_lines=nmatch[1] / nchk,
                                                                                           local_points=nmatch[2] / nchk, local_lines=nmatch[3] / nchk)),
               cpu_baseline=dict(value=round(ncpu / t_cpu, 2) if t_cpu > 0 else None, unit="frames/s", cores=1, kind="port",
                                 sample="%d frames, the oracle chain (oracle/frame_search.cc) on one thread" % ncpu))
    # what bounds it: the instruction streams of the prepass lanes.  SQ counters of the same build (tools/pmc_tracking.sh ->
    # profiles/sq_insts_tracking.json), VALU wave-instructions per frame x frames/s against 1024 SIMDs x 2.4 GHz / 4.2 cycles
    try:
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        pj = os.path.join(ROOT, "profiles", "sq_insts_tracking.json")
        if os.path.exists(pj):
            ij = json.load(open(pj))
            if ij.get("build") == g._lib_id(g.LIB) and lib is None:
                peak = 1024 * 2.4e9 / 4.2
                valu = float(ij["total_valu"])
                out["roofline"] = dict(bound="valu_issue", achieved=round(valu * out["value"] / 1e9, 1), peak=round(peak / 1e9, 1), unit="G wave-instructions/s",
                                       frac=round(valu * out["value"] / peak, 4), valu_wave_instructions_per_frame=round(valu),
                                       wait_share=round(float(ij["total_wait_any"]) / max(float(ij["total_wave_cycles"]), 1.0), 3),
                                       per_kernel=ij["kernels"], build=ij.get("build"),
                                       what="all kernels of a step; SQ counters of this build (tools/pmc_tracking.sh)")
    except Exception as e:   # noqa: BLE001
        out["roofline"] = {"error": repr(e)[:160]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1024)
    ap.add_argument("--distinct", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    r = run(a.pairs, a.distinct, a.steps)
    if a.json:
        print(json.dumps(r))
    else:
        print(json.dumps(r, indent=1))
    return 0 if r["verified"]["exact"] else 1


if __name__ == "__main__":
    sys.exit(main())
