"""Batch front end on one GPU: ORB extract + line extract + frame-to-frame matching, all on device buffers.

This is the hot path bench.py times and the end-to-end GPU test checks.  One `step()` processes a batch of B
independent frames that are already resident in HBM:

    ORBextractor::operator()            per frame                      (plh_orb_extract_batch_dev)
    LINEextractor::operator()           per frame, optional undistort  (plh_line_extract_batch_dev)
    Frame::ComputeBoW (mFeatVec + mBowVec) per frame                   (plh_vocab_transform_batch_dev)
    ORBmatcher(0.7).SearchByBoW         frame b (as KeyFrame) -> frame b+1   (plh_orb_search_by_bow_kp_batch_dev)
    LSDmatcher(0.7).SearchDouble        frame b -> frame b+1                 (plh_line_search_double_batch_dev)

Frame B's successor is frame 0 (its records are copied into slot B), so every frame is matched once.
PyTorch only provides device memory and the streams; every computation is a kernel of libplslam_hip.so.

The ORB half (+ BoW + SearchByBoW) and the line half (+ SearchDouble) are independent until the results are
consumed, exactly as the reference runs ExtractORB and ExtractLSD on two threads (Frame.cc:224-227): they are
enqueued on two HIP streams.  LSD region growing is one latency-bound wavefront per frame, so the ORB kernels run in
the SIMD slots it leaves idle.  FrontEndPipelined (below) additionally staggers sub-batches.
"""
import ctypes as C

import numpy as np


class FrontEndBatch:
    def __init__(self, P, vocab, batch, rows=480, cols=640, nfeatures=1000, nlevels=8, n_lines=200, min_line_length=0.0,
                 K=None, D=None, device=0):
        import torch
        self.torch, self.P = torch, P
        self.B, self.rows, self.cols = batch, rows, cols
        self.dev = torch.device("cuda", device)
        self.lib = P.load()
        self.orb = P.ORBextractor(nfeatures, 1.2, nlevels, 20, 7, rows=rows, cols=cols, max_batch=batch, device=device)
        self.line = P.LINEextractor(1, 1.2, n_lines, min_line_length, rows=rows, cols=cols, max_batch=batch, device=device, K=K, D=D)
        self.vocab = vocab
        B1 = batch + 1
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.dev)
        self.ocap, self.lcap = self.orb.capacity, self.line.capacity
        self.kps = z((B1, self.ocap, 7), torch.float32)
        self.desc = z((B1, self.ocap, 32), torch.uint8)
        self.n = z((B1,), torch.int32)
        self.nid = z((B1, self.ocap), torch.int32)
        self.word = z((B1, self.ocap), torch.int32)
        self.bow_word = z((B1, self.ocap), torch.int32)      # mBowVec: word ids in map order ...
        self.bow_value = z((B1, self.ocap), torch.float64)   # ... and their L1-normalised tf-idf weights (WordValue = double)
        self.bow_n = z((B1,), torch.int32)
        self.valid = torch.ones((batch, self.ocap), dtype=torch.uint8, device=self.dev)
        self.m_orb = z((batch, self.ocap), torch.int32)
        self.nm_orb = z((batch,), torch.int32)
        self.kl = z((B1, self.lcap, 17), torch.float32)
        self.ldesc = z((B1, self.lcap, 32), torch.uint8)
        self.lfn = z((B1, self.lcap, 3), torch.float64)
        self.nl = z((B1,), torch.int32)
        self.m_line = z((batch, self.lcap), torch.int32)
        self.nm_line = z((batch,), torch.int32)
        self.ws_bytes = self.lib.plh_line_search_double_workspace(self.lcap, batch)
        self.ws = z((self.ws_bytes,), torch.uint8)
        # the line chain is the critical path (image prep -> region growing -> LBD): high priority, so that its
        # streaming kernels are dispatched first and the ORB kernels fill in underneath the region growing
        # (other layouts -- equal priorities, one stream per part, one ORB stream for all parts -- were measured in rounds 2 and 3
        # and stay within the run-to-run spread or lose)
        self.line_stream = torch.cuda.Stream(device=self.dev, priority=-1)
        self.orb_stream = torch.cuda.Stream(device=self.dev, priority=0)
        self.ev_start = torch.cuda.Event()
        self.ev_orb = torch.cuda.Event()
        self.ev_line = torch.cuda.Event()
        self.overlap = True    # False: both halves on the caller's stream (per-kernel timing without interference)
        self.ev_free = None    # optional event a consumer records when it has read this part's records (see gather())
        # the vocabulary lives in the C library (plh_vocab): one handle per vocabulary object and device
        if getattr(vocab, "_plh_handle", None) is None or vocab._plh_handle.device != device:
            hv = P.ORBVocabulary(device=device)
            parent, leaf = vocab.tree_arrays()
            hv.create(vocab.k, vocab.L, parent, leaf, vocab.node_desc, vocab.weight64)
            vocab._plh_handle = hv
        self.hvoc = vocab._plh_handle
        L = self.lib
        V, I, F, Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        L.plh_orb_search_by_bow_kp_batch_dev.argtypes = [V] * 9 + [I, I, I, F, I, V, V, V]
        L.plh_orb_search_by_bow_kp_batch_dev.restype = I

    def close(self):
        self.orb.close()
        self.line.close()

    def enqueue_line(self, d_imgs, main, ev_start):
        """Line half: LINEextractor batch + SearchDouble against the next frame, on the high-priority stream."""
        P, L, B, t = self.P, self.lib, self.B, self.torch
        p = P._p
        sl = self.line_stream if self.overlap else main
        if self.overlap:
            sl.wait_event(ev_start)
        if self.ev_free is not None:
            sl.wait_event(self.ev_free)       # the previous step's records have been consumed
        sm = sl.cuda_stream
        self.line.extract_batch_dev(d_imgs, B, self.rows * self.cols, self.kl, self.ldesc, self.lfn, self.nl, sm)
        with t.cuda.stream(sl):   # slot B := frame 0 (so that frame B-1 has a successor)
            for buf in (self.kl, self.ldesc, self.nl):
                buf[B].copy_(buf[0], non_blocking=True)
        P._check(L, L.plh_line_search_double_batch_dev(p(self.ldesc), p(self.nl), p(self.ldesc[1:]), p(self.nl[1:]), self.lcap, B,
                                                       50.0, 0.7, p(self.m_line), p(self.nm_line), p(self.ws), self.ws_bytes,
                                                       C.c_void_p(sm)),
                 "plh_line_search_double_batch_dev")
        if self.overlap:
            self.ev_line.record(sl)

    def enqueue_orb(self, d_imgs, main, ev_start):
        """ORB half: ORBextractor batch + BoW feature vectors + SearchByBoW against the next frame, on its own stream."""
        P, L, B, t = self.P, self.lib, self.B, self.torch
        p = P._p
        so = self.orb_stream if self.overlap else main
        if self.overlap:
            so.wait_event(ev_start)
        if self.ev_free is not None:
            so.wait_event(self.ev_free)
        sp = C.c_void_p(so.cuda_stream)
        self.orb.extract_batch_dev(d_imgs, B, self.rows * self.cols, self.kps, self.desc, self.n, so.cuda_stream)
        with t.cuda.stream(so):
            for buf in (self.kps, self.desc, self.n):
                buf[B].copy_(buf[0], non_blocking=True)
        P._check(L, L.plh_vocab_transform_batch_dev(self.hvoc.h, p(self.desc), p(self.n), self.ocap, B + 1, 4, p(self.nid),
                                                    p(self.word), p(self.bow_word), p(self.bow_value), p(self.bow_n), sp),
                 "plh_vocab_transform_batch_dev")
        P._check(L, L.plh_orb_search_by_bow_kp_batch_dev(p(self.desc), p(self.kps), p(self.nid), p(self.valid), p(self.n),
                                                         p(self.desc[1:]), p(self.kps[1:]), p(self.nid[1:]), p(self.n[1:]),
                                                         self.ocap, B, 50, 0.7, 1, p(self.m_orb), p(self.nm_orb), sp),
                 "plh_orb_search_by_bow_kp_batch_dev")
        if self.overlap:
            self.ev_orb.record(so)

    def join(self, main):
        """Make the caller's stream wait for everything this part has enqueued."""
        if self.overlap:
            main.wait_event(self.ev_orb)
            main.wait_event(self.ev_line)

    def step(self, d_imgs, stream=None):
        """Enqueue one pass over the resident batch `d_imgs` (uint8 [B, rows, cols]); complete on the caller's stream."""
        t = self.torch
        main = t.cuda.current_stream(self.dev) if stream is None else t.cuda.ExternalStream(stream, device=self.dev)
        self.ev_start.record(main)
        self.enqueue_line(d_imgs, main, self.ev_start)
        self.enqueue_orb(d_imgs, main, self.ev_start)
        self.join(main)

    def results(self):
        """Host copies of everything one step produced (synchronises)."""
        t, P = self.torch, self.P
        t.cuda.synchronize(self.dev)
        flags = self.orb.status() | self.line.status()   # the _dev path reports truncation here, never silently
        if flags:
            raise P.PlhError("front end: a fixed-capacity buffer overflowed (flags 0x%x)" % flags)
        B = self.B
        kps = self.kps[:B].cpu().numpy().view(np.uint8).reshape(B, self.ocap, 28).copy().view(P.KP_DTYPE).reshape(B, self.ocap)
        kl = self.kl[:B].cpu().numpy().view(np.uint8).reshape(B, self.lcap, 68).copy().view(P.KL_DTYPE).reshape(B, self.lcap)
        return dict(n=self.n[:B].cpu().numpy(), kps=kps, desc=self.desc[:B].cpu().numpy(), nid=self.nid[:B].cpu().numpy(),
                    word=self.word[:B].cpu().numpy(), bow_word=self.bow_word[:B].cpu().numpy(),
                    bow_value=self.bow_value[:B].cpu().numpy(), bow_n=self.bow_n[:B].cpu().numpy(),
                    nl=self.nl[:B].cpu().numpy(), kl=kl, ldesc=self.ldesc[:B].cpu().numpy(), lfn=self.lfn[:B].cpu().numpy(),
                    m_orb=self.m_orb.cpu().numpy(), nm_orb=self.nm_orb.cpu().numpy(), m_line=self.m_line.cpu().numpy(),
                    nm_line=self.nm_line.cpu().numpy())


class _Borrowed:
    """An extractor handle owned by a plh_frontend, seen through the per-handle API (profiling, status, grow waves)."""

    def __init__(self, cls, lib, h, capacity):
        self._obj = cls.__new__(cls)
        self._obj.lib, self._obj.h, self._obj.capacity = lib, h, capacity
        self._obj.close = lambda: None          # the front end destroys it

    def __getattr__(self, k):
        return getattr(self._obj, k)


class _Part:
    pass


class FrontEndPipelined:
    """The same front end over `nsplit` sub-batches, each with its own extractor handles and stream pair -- a binding of the
    library's plh_frontend_* (csrc/frontend_host.hip: the sub-batch handles, the stream pairs with the line chain on the
    high-priority one, the events and the per-sub-batch gather live in C++; a C++ host calls the same four functions).
    LSD region growing is latency-bound while every other kernel is a dense streaming kernel: with the sub-batches staggered,
    the dense kernels of one run underneath the region growing of another, and -- because a sub-batch only depends on its
    own previous step -- consecutive steps overlap as well (the batches of a video stream are independent).
    `step(..., join=True)` restores strict step-by-step completion on the caller's stream.  The record buffers are torch
    tensors bound to the handle (external_records), so that the results are visible to Python without a copy."""

    GATHERED = ("n", "kps", "desc", "nl", "kl", "ldesc", "lfn")   # the records a tracker on another GPU needs

    def __init__(self, P, vocab, batch, rows=480, cols=640, nfeatures=1000, nlevels=8, n_lines=200, min_line_length=0.0,
                 K=None, D=None, device=0, nsplit=2, lsd_refine=-1):
        import torch
        assert batch % nsplit == 0
        self.torch, self.P, self.B, self.nsplit, self.Bp = torch, P, batch, nsplit, batch // nsplit
        self.dev = torch.device("cuda", device)
        self.lib = L = P.load()
        if getattr(vocab, "_plh_handle", None) is None or vocab._plh_handle.device != device:
            hv = P.ORBVocabulary(device=device)
            parent, leaf = vocab.tree_arrays()
            hv.create(vocab.k, vocab.L, parent, leaf, vocab.node_desc, vocab.weight64)
            vocab._plh_handle = hv
        self.hvoc = vocab._plh_handle
        fp = P.FrontendParams()
        fp.struct_size = C.sizeof(P.FrontendParams)
        fp.rows, fp.cols = rows, cols
        fp.orb = P.OrbParams(nfeatures, 1.2, nlevels, 20, 7)
        fp.line = P.LineParams(1, 1.2, n_lines, min_line_length)
        und = K is not None and D is not None and any(float(d) != 0.0 for d in D)
        fp.undistort = int(und)
        if und:
            fp.K = (C.c_float * 4)(*[float(v) for v in K])
            fp.D = (C.c_float * 5)(*[float(v) for v in D])
        fp.bow_levelsup, fp.orb_th_low, fp.orb_nnratio, fp.orb_check_orientation = 4, 50, 0.7, 1
        fp.line_th, fp.line_nnratio = 50.0, 0.7
        fp.external_records = 1
        # this binding's argument: -1 = the library's default (plh_lsd_refine_default()), 0 = LSD_REFINE_STD, 1 = LSD_REFINE_ADV;
        # the C struct: PLH_FRONTEND_REFINE_LIBRARY = 0 (what a zeroed struct holds), _STD = 0x100, _ADV = 0x101
        fp.lsd_refine = 0 if int(lsd_refine) < 0 else 0x100 | int(lsd_refine)
        h = C.c_void_p()
        P._check(L, L.plh_frontend_create(C.byref(fp), self.hvoc.h, batch, nsplit, device, C.byref(h)), "plh_frontend_create")
        self.h = h
        self._overlap = True
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.dev)
        self.parts = []
        for k in range(nsplit):
            r = P.FrontendRecords()
            P._check(L, L.plh_frontend_records_of(self.h, k, C.byref(r)), "plh_frontend_records_of")
            pt = _Part()
            pt.B, pt.ocap, pt.lcap = r.frames, r.orb_capacity, r.line_capacity
            B, B1, oc, lc = pt.B, pt.B + 1, pt.ocap, pt.lcap
            pt.kps, pt.desc, pt.n = z((B1, oc, 7), torch.float32), z((B1, oc, 32), torch.uint8), z((B1,), torch.int32)
            pt.nid, pt.word, pt.bow_word = z((B1, oc), torch.int32), z((B1, oc), torch.int32), z((B1, oc), torch.int32)
            pt.bow_value, pt.bow_n = z((B1, oc), torch.float64), z((B1,), torch.int32)
            pt.kl, pt.ldesc, pt.lfn, pt.nl = z((B1, lc, 17), torch.float32), z((B1, lc, 32), torch.uint8), z((B1, lc, 3), torch.float64), z((B1,), torch.int32)
            pt.m_orb, pt.nm_orb = z((B, oc), torch.int32), z((B,), torch.int32)
            pt.m_line, pt.nm_line = z((B, lc), torch.int32), z((B,), torch.int32)
            for name in ("kps", "desc", "n", "nid", "word", "bow_word", "bow_value", "bow_n", "kl", "ldesc", "lfn", "nl", "m_orb", "nm_orb",
                         "m_line", "nm_line"):
                setattr(r, name, getattr(pt, name).data_ptr())
            P._check(L, L.plh_frontend_bind_records(self.h, k, C.byref(r)), "plh_frontend_bind_records")
            ho, hl = C.c_void_p(), C.c_void_p()
            P._check(L, L.plh_frontend_handles(self.h, k, C.byref(ho), C.byref(hl)), "plh_frontend_handles")
            pt.orb = _Borrowed(P.ORBextractor, L, ho, oc)
            pt.line = _Borrowed(P.LINEextractor, L, hl, lc)
            self.parts.append(pt)

    @property
    def overlap(self):
        return self._overlap

    @overlap.setter
    def overlap(self, v):
        self._overlap = bool(v)
        self.P._check(self.lib, self.lib.plh_frontend_set_overlap(self.h, int(bool(v))), "plh_frontend_set_overlap")

    def close(self):
        if getattr(self, "h", None):
            self.lib.plh_frontend_destroy(self.h)
            self.h = None
        self.parts = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, d_imgs, join=True):
        main = self.torch.cuda.current_stream(self.dev)
        rows_cols = d_imgs.shape[1] * d_imgs.shape[2]
        self.P._check(self.lib, self.lib.plh_frontend_step(self.h, self.P._p(d_imgs), rows_cols, C.c_void_p(main.cuda_stream), int(join)),
                      "plh_frontend_step")

    def join(self):
        main = self.torch.cuda.current_stream(self.dev)
        self.P._check(self.lib, self.lib.plh_frontend_join(self.h, C.c_void_p(main.cuda_stream)), "plh_frontend_join")

    def alloc_gather_buffers(self, world, receives=True):
        """Per sub-batch: the send views (this rank's records) and, on receiving ranks, buffers for `world` ranks' records."""
        send, recv = [], []
        for p in self.parts:
            sv = [getattr(p, k)[:p.B] for k in self.GATHERED]
            send.append(sv)
            recv.append([v.new_empty((world,) + tuple(v.shape)) for v in sv] if receives else [None] * len(sv))
        return {"send": send, "recv": recv}

    def gather(self, comm_stream, comm, root, bufs):
        """N > 1: the fixed-stride records of every sub-batch go over RCCL on `comm_stream` as soon as that sub-batch is done
        (plh_frontend_gather -> plh_gather_records: one grouped launch per sub-batch; root = -1 all ranks receive, else only
        `root`), without joining the step: a sub-batch's next step waits only for its own gather."""
        if not self._overlap:
            comm_stream.wait_stream(self.torch.cuda.current_stream(self.dev))
        ptrs = (C.c_void_p * (self.nsplit * self.P.FRONTEND_GATHERED))()
        for i, rv in enumerate(bufs["recv"]):
            for k, t in enumerate(rv):
                ptrs[i * self.P.FRONTEND_GATHERED + k] = t.data_ptr() if t is not None else None
        self.P._check(self.lib, self.lib.plh_frontend_gather(self.h, comm.h, root, ptrs, C.c_void_p(comm_stream.cuda_stream)),
                      "plh_frontend_gather")

    def results(self):
        """Host copies of everything one step produced (synchronises; raises if a fixed-capacity buffer overflowed)."""
        t, P = self.torch, self.P
        t.cuda.synchronize(self.dev)
        f = C.c_int(0)
        P._check(self.lib, self.lib.plh_frontend_status(self.h, C.byref(f)), "plh_frontend_status")
        if f.value:
            raise P.PlhError("front end: a fixed-capacity buffer overflowed or a launch was abandoned (flags 0x%x)" % f.value)
        rs = []
        for p in self.parts:
            B = p.B
            kps = p.kps[:B].cpu().numpy().view(np.uint8).reshape(B, p.ocap, 28).copy().view(P.KP_DTYPE).reshape(B, p.ocap)
            kl = p.kl[:B].cpu().numpy().view(np.uint8).reshape(B, p.lcap, 68).copy().view(P.KL_DTYPE).reshape(B, p.lcap)
            rs.append(dict(n=p.n[:B].cpu().numpy(), kps=kps, desc=p.desc[:B].cpu().numpy(), nid=p.nid[:B].cpu().numpy(),
                           word=p.word[:B].cpu().numpy(), bow_word=p.bow_word[:B].cpu().numpy(),
                           bow_value=p.bow_value[:B].cpu().numpy(), bow_n=p.bow_n[:B].cpu().numpy(),
                           nl=p.nl[:B].cpu().numpy(), kl=kl, ldesc=p.ldesc[:B].cpu().numpy(), lfn=p.lfn[:B].cpu().numpy(),
                           m_orb=p.m_orb.cpu().numpy(), nm_orb=p.nm_orb.cpu().numpy(), m_line=p.m_line.cpu().numpy(),
                           nm_line=p.nm_line.cpu().numpy()))
        return {k: np.concatenate([r[k] for r in rs], axis=0) for k in rs[0]}
