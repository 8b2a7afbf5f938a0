// plh_frontend_*: the host side of the throughput path, in the library (C ABI, include/plslam_hip.h), so that a C++ host --
// the reference is C++ (north star: "host code stays C++") -- drives the whole batch front end with four calls:
//     create  ->  step (enqueue one pass over a resident batch)  ->  [gather over RCCL]  ->  read the records
// One step is what Frame::Frame() and Tracking's frame-to-frame matching do per frame, for a batch of independent frames:
//     ORBextractor::operator()               (Frame.cc:322-328)      plh_orb_extract_batch_dev
//     undistort + LINEextractor::operator()  (Frame.cc:220-227,331)  plh_line_extract_batch_dev
//     Frame::ComputeBoW                      (Frame.cc:906-913)      plh_vocab_transform_batch_dev
//     ORBmatcher(0.7).SearchByBoW            frame b -> frame b + 1  plh_orb_search_by_bow_kp_batch_dev
//     LSDmatcher::SearchDouble               frame b -> frame b + 1  plh_line_search_double_batch_dev
// What it owns: `nsplit` sub-batches, each with its own extractor handles, result buffers, a stream pair (the line chain on a
// high-priority stream -- it is the critical path: image prep -> region growing -> LBD -- the ORB chain, BoW and SearchByBoW
// on a second one, as the reference runs ExtractORB and ExtractLSD on two threads) and the events that tie them to the
// caller's stream.  A sub-batch only depends on its own previous step, so consecutive steps overlap unless the caller joins;
// with a gather in between, a sub-batch's next step waits for its own gather only.
#include <new>
#include <vector>

#include "plh_common.h"

using namespace plh;

namespace {

struct Part {
  int first = 0, B = 0;
  plh_orb* orb = nullptr;
  plh_line* line = nullptr;
  int ocap = 0, lcap = 0;
  // records: B + 1 slots where a successor is needed (slot B = copy of slot 0, so that frame B - 1 has one)
  plh_keypoint* kps = nullptr; uint8_t* desc = nullptr; int32_t* n = nullptr;
  int32_t *nid = nullptr, *word = nullptr, *bowWord = nullptr; double* bowValue = nullptr; int32_t* bowN = nullptr;
  uint8_t* valid = nullptr;
  int32_t *mOrb = nullptr, *nmOrb = nullptr;
  plh_keyline* kl = nullptr; uint8_t* ldesc = nullptr; double* lfn = nullptr; int32_t* nl = nullptr;
  int32_t *mLine = nullptr, *nmLine = nullptr;
  void* ws = nullptr; size_t wsBytes = 0;
  hipStream_t sLine = nullptr, sOrb = nullptr;
  hipEvent_t evOrb = nullptr, evLine = nullptr, evFree = nullptr, evFreeOrb = nullptr;   // evFree: the line records (or all) are gathered
  bool freeValid = false, freeOrbValid = false, ran = false;
};

}  // namespace

struct plh_frontend {
  plh_frontend_params p;
  const plh_vocab* voc = nullptr;
  int device = 0, batch = 0, nsplit = 0, Bp = 0;
  bool overlap = true;
  std::vector<Part> parts;
  bool around = false;   // small resident batch: the ORB chain runs around region growing, not underneath it (see plh_frontend_step)
  hipEvent_t evStart = nullptr;
  std::vector<void*> allocs;
};

namespace {

template <typename T>
plh_status dev_alloc(plh_frontend* fe, T** out, size_t count, bool ones = false) {
  void* p = nullptr;
  const size_t bytes = count * sizeof(T);
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess || hipMemset(p, ones ? 1 : 0, bytes ? bytes : 1) != hipSuccess) {
    (void)hipGetLastError();
    set_error("plh_frontend_create: cannot allocate %zu bytes of record buffers", bytes);
    return PLH_ERR_ALLOC;
  }
  fe->allocs.push_back(p);
  *out = static_cast<T*>(p);
  return PLH_OK;
}

#define FE_TRY(x) do { const plh_status st__ = (x); if (st__ != PLH_OK) { plh_frontend_destroy(fe); return st__; } } while (0)
#define FE_HIP(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); set_error("plh_frontend: %s failed", #x); plh_frontend_destroy(fe); return PLH_ERR_HIP; } } while (0)

plh_status enqueue_line(plh_frontend* fe, Part& pt, const uint8_t* imgs, size_t stride, hipStream_t main) {
  hipStream_t s = fe->overlap ? pt.sLine : main;
  if (fe->overlap) PLH_HIP(hipStreamWaitEvent(s, fe->evStart, 0));
  if (pt.freeValid) PLH_HIP(hipStreamWaitEvent(s, pt.evFree, 0));   // the previous step's records have been consumed
  const int B = pt.B;
  plh_status st = plh_line_extract_batch_dev(pt.line, imgs, B, stride, nullptr, pt.kl, pt.ldesc, pt.lfn, pt.nl, s);
  if (st != PLH_OK) return st;
  // slot B := frame 0
  PLH_HIP(hipMemcpyAsync(pt.kl + (size_t)B * pt.lcap, pt.kl, (size_t)pt.lcap * sizeof(plh_keyline), hipMemcpyDeviceToDevice, s));
  PLH_HIP(hipMemcpyAsync(pt.ldesc + (size_t)B * pt.lcap * 32, pt.ldesc, (size_t)pt.lcap * 32, hipMemcpyDeviceToDevice, s));
  PLH_HIP(hipMemcpyAsync(pt.nl + B, pt.nl, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  st = plh_line_search_double_batch_dev(pt.ldesc, pt.nl, pt.ldesc + (size_t)pt.lcap * 32, pt.nl + 1, pt.lcap, B, fe->p.line_th,
                                        fe->p.line_nnratio, pt.mLine, pt.nmLine, pt.ws, pt.wsBytes, s);
  if (st != PLH_OK) return st;
  PLH_HIP(hipEventRecord(pt.evLine, s));   // on whichever stream ran the chain: plh_frontend_gather waits for it in every mode
  return PLH_OK;
}

plh_status enqueue_orb(plh_frontend* fe, Part& pt, const uint8_t* imgs, size_t stride, hipStream_t main) {
  hipStream_t s = fe->overlap ? pt.sOrb : main;
  if (fe->overlap) PLH_HIP(hipStreamWaitEvent(s, fe->evStart, 0));
  if (pt.freeOrbValid) PLH_HIP(hipStreamWaitEvent(s, pt.evFreeOrb, 0));   // the ORB records were gathered on their own (below)
  else if (pt.freeValid) PLH_HIP(hipStreamWaitEvent(s, pt.evFree, 0));
  const int B = pt.B;
  plh_status st = plh_orb_extract_batch_dev(pt.orb, imgs, B, stride, pt.kps, pt.desc, pt.n, s);
  if (st != PLH_OK) return st;
  PLH_HIP(hipMemcpyAsync(pt.kps + (size_t)B * pt.ocap, pt.kps, (size_t)pt.ocap * sizeof(plh_keypoint), hipMemcpyDeviceToDevice, s));
  PLH_HIP(hipMemcpyAsync(pt.desc + (size_t)B * pt.ocap * 32, pt.desc, (size_t)pt.ocap * 32, hipMemcpyDeviceToDevice, s));
  PLH_HIP(hipMemcpyAsync(pt.n + B, pt.n, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  st = plh_vocab_transform_batch_dev(fe->voc, pt.desc, pt.n, pt.ocap, B + 1, fe->p.bow_levelsup, pt.nid, pt.word, pt.bowWord, pt.bowValue,
                                     pt.bowN, s);
  if (st != PLH_OK) return st;
  st = plh_orb_search_by_bow_kp_batch_dev(pt.desc, pt.kps, pt.nid, pt.valid, pt.n, pt.desc + (size_t)pt.ocap * 32, pt.kps + pt.ocap,
                                          pt.nid + pt.ocap, pt.n + 1, pt.ocap, B, fe->p.orb_th_low, fe->p.orb_nnratio,
                                          fe->p.orb_check_orientation, pt.mOrb, pt.nmOrb, s);
  if (st != PLH_OK) return st;
  PLH_HIP(hipEventRecord(pt.evOrb, s));
  return PLH_OK;
}

}  // namespace

extern "C" {

plh_status plh_frontend_destroy(plh_frontend* fe) {
  if (!fe) return PLH_OK;
  (void)hipSetDevice(fe->device);
  (void)hipDeviceSynchronize();
  for (Part& pt : fe->parts) {
    if (pt.orb) plh_orb_destroy(pt.orb);
    if (pt.line) plh_line_destroy(pt.line);
    if (pt.sLine) (void)hipStreamDestroy(pt.sLine);
    if (pt.sOrb) (void)hipStreamDestroy(pt.sOrb);
    for (hipEvent_t e : {pt.evOrb, pt.evLine, pt.evFree, pt.evFreeOrb})
      if (e) (void)hipEventDestroy(e);
  }
  for (void* p : fe->allocs) (void)hipFree(p);
  if (fe->evStart) (void)hipEventDestroy(fe->evStart);
  delete fe;
  return PLH_OK;
}

plh_status plh_frontend_create(const plh_frontend_params* p, const plh_vocab* voc, int batch, int nsplit, int device, plh_frontend** out) {
  if (!p || !voc || !out || batch <= 0 || nsplit <= 0 || batch % nsplit) {
    set_error("plh_frontend_create: invalid argument (batch %d must be a positive multiple of nsplit %d)", batch, nsplit);
    return PLH_ERR_INVALID;
  }
  if (p->struct_size != sizeof(plh_frontend_params)) {   // a caller compiled against another header: do not read its struct
    set_error("plh_frontend_create: plh_frontend_params::struct_size is %u, this library's struct has %zu bytes (set it to "
              "sizeof(plh_frontend_params); a mismatch means the caller was built against another plslam_hip.h)",
              p->struct_size, sizeof(plh_frontend_params));
    return PLH_ERR_INVALID;
  }
  if (p->lsd_refine != PLH_FRONTEND_REFINE_LIBRARY && p->lsd_refine != PLH_FRONTEND_REFINE_STD && p->lsd_refine != PLH_FRONTEND_REFINE_ADV) {
    set_error("plh_frontend_create: lsd_refine = %d: must be PLH_FRONTEND_REFINE_LIBRARY (0), PLH_FRONTEND_REFINE_STD (0x100) or "
              "PLH_FRONTEND_REFINE_ADV (0x101) -- not a PLH_LSD_REFINE_* value, and not round 5's 1 / 2", (int)p->lsd_refine);
    return PLH_ERR_INVALID;
  }
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  PLH_HIP(hipSetDevice(device));
  plh_frontend* fe = new (std::nothrow) plh_frontend();
  if (!fe) return PLH_ERR_ALLOC;
  fe->p = *p; fe->voc = voc; fe->device = device; fe->batch = batch; fe->nsplit = nsplit; fe->Bp = batch / nsplit;
  try {   // nothing may throw across the C ABI: the host containers are sized here, once
    fe->parts.resize(nsplit);
    fe->allocs.reserve((size_t)nsplit * 20);
  } catch (const std::bad_alloc&) {
    delete fe;
    set_error("plh_frontend_create: out of host memory");
    return PLH_ERR_ALLOC;
  }
  FE_HIP(hipEventCreateWithFlags(&fe->evStart, hipEventDisableTiming));
  int prLo = 0, prHi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prLo, &prHi);   // (least, greatest): greatest is the numerically lower one
  for (int k = 0; k < nsplit; k++) {
    Part& pt = fe->parts[k];
    pt.first = k * fe->Bp; pt.B = fe->Bp;
    FE_TRY(plh_orb_create(&p->orb, device, p->rows, p->cols, pt.B, &pt.orb));
    FE_TRY(plh_line_create(&p->line, device, p->rows, p->cols, pt.B, &pt.line));
    if (p->undistort) FE_TRY(plh_line_set_undistort(pt.line, p->K, p->D));
    // ONE quantity decides both the wavefronts per frame of LSD's region growing and the schedule that goes with them: the
    // frames resident in ALL sub-batches (they run together).  Up to 1024: eight wavefronts per frame (k_lsd_grow_mw) and the
    // ORB chain around region growing (plh_frontend_step); above: one wavefront per frame, the ORB chain underneath.
    // (-1 = the extractor's own choice by batch size -- plh_line_set_grow_waves: 8 up to 1024 frames, measured; the sub-batch is at
    // most the resident batch, so it takes the multi-wavefront kernel exactly when `around` is set)
    fe->around = batch <= 1024;
    FE_TRY(plh_line_set_grow_waves(pt.line, fe->around ? -1 : 0));
    if (p->lsd_refine != PLH_FRONTEND_REFINE_LIBRARY) FE_TRY(plh_line_set_refine(pt.line, p->lsd_refine & 0xff));   // (0x100 | PLH_LSD_REFINE_*)
    FE_TRY(plh_line_reserve(pt.line, pt.B));   // workspace now: an out-of-memory condition belongs to create, not to the first step
    pt.ocap = plh_orb_capacity(pt.orb); pt.lcap = plh_line_capacity(pt.line);
    const size_t B1 = (size_t)pt.B + 1, oc = (size_t)pt.ocap, lc = (size_t)pt.lcap;
    FE_TRY(dev_alloc(fe, &pt.valid, (size_t)pt.B * oc, true));
    pt.wsBytes = plh_line_search_double_workspace(pt.lcap, pt.B);
    {
      uint8_t* w = nullptr;
      FE_TRY(dev_alloc(fe, &w, pt.wsBytes));
      pt.ws = w;
    }
    if (!p->external_records) {
    FE_TRY(dev_alloc(fe, &pt.kps, B1 * oc)); FE_TRY(dev_alloc(fe, &pt.desc, B1 * oc * 32)); FE_TRY(dev_alloc(fe, &pt.n, B1));
    FE_TRY(dev_alloc(fe, &pt.nid, B1 * oc)); FE_TRY(dev_alloc(fe, &pt.word, B1 * oc)); FE_TRY(dev_alloc(fe, &pt.bowWord, B1 * oc));
    FE_TRY(dev_alloc(fe, &pt.bowValue, B1 * oc)); FE_TRY(dev_alloc(fe, &pt.bowN, B1));
    FE_TRY(dev_alloc(fe, &pt.mOrb, (size_t)pt.B * oc)); FE_TRY(dev_alloc(fe, &pt.nmOrb, (size_t)pt.B));
    FE_TRY(dev_alloc(fe, &pt.kl, B1 * lc)); FE_TRY(dev_alloc(fe, &pt.ldesc, B1 * lc * 32)); FE_TRY(dev_alloc(fe, &pt.lfn, B1 * lc * 3));
    FE_TRY(dev_alloc(fe, &pt.nl, B1));
    FE_TRY(dev_alloc(fe, &pt.mLine, (size_t)pt.B * lc)); FE_TRY(dev_alloc(fe, &pt.nmLine, (size_t)pt.B));
    }
    FE_HIP(hipStreamCreateWithPriority(&pt.sLine, hipStreamNonBlocking, prHi));
    FE_HIP(hipStreamCreateWithPriority(&pt.sOrb, hipStreamNonBlocking, prLo));
    FE_HIP(hipEventCreateWithFlags(&pt.evOrb, hipEventDisableTiming));
    FE_HIP(hipEventCreateWithFlags(&pt.evLine, hipEventDisableTiming));
    FE_HIP(hipEventCreateWithFlags(&pt.evFree, hipEventDisableTiming));
    FE_HIP(hipEventCreateWithFlags(&pt.evFreeOrb, hipEventDisableTiming));
    if (fe->around) FE_TRY(plh_line_set_grow_events(pt.line, pt.evOrb, nullptr));
  }
  *out = fe;
  return PLH_OK;
}

int plh_frontend_parts(const plh_frontend* fe) { return fe ? fe->nsplit : 0; }

plh_status plh_frontend_set_overlap(plh_frontend* fe, int on) {
  if (!fe) return PLH_ERR_INVALID;
  fe->overlap = on != 0;
  return PLH_OK;
}

plh_status plh_frontend_handles(plh_frontend* fe, int part, plh_orb** orb, plh_line** line) {
  if (!fe || part < 0 || part >= fe->nsplit) return PLH_ERR_INVALID;
  if (orb) *orb = fe->parts[part].orb;
  if (line) *line = fe->parts[part].line;
  return PLH_OK;
}

plh_status plh_frontend_records_of(plh_frontend* fe, int part, plh_frontend_records* r) {
  if (!fe || !r || part < 0 || part >= fe->nsplit) return PLH_ERR_INVALID;
  const Part& pt = fe->parts[part];
  r->first = pt.first; r->frames = pt.B; r->orb_capacity = pt.ocap; r->line_capacity = pt.lcap;
  r->kps = pt.kps; r->desc = pt.desc; r->n = pt.n; r->nid = pt.nid; r->word = pt.word; r->bow_word = pt.bowWord;
  r->bow_value = pt.bowValue; r->bow_n = pt.bowN; r->kl = pt.kl; r->ldesc = pt.ldesc; r->lfn = pt.lfn; r->nl = pt.nl;
  r->m_orb = pt.mOrb; r->nm_orb = pt.nmOrb; r->m_line = pt.mLine; r->nm_line = pt.nmLine;
  return PLH_OK;
}

plh_status plh_frontend_bind_records(plh_frontend* fe, int part, const plh_frontend_records* r) {
  if (!fe || !r || part < 0 || part >= fe->nsplit || !fe->p.external_records) {
    set_error("plh_frontend_bind_records: invalid argument (the handle must be created with external_records = 1)");
    return PLH_ERR_INVALID;
  }
  if (!r->kps || !r->desc || !r->n || !r->nid || !r->word || !r->bow_word || !r->bow_value || !r->bow_n || !r->kl || !r->ldesc ||
      !r->lfn || !r->nl || !r->m_orb || !r->nm_orb || !r->m_line || !r->nm_line) {
    set_error("plh_frontend_bind_records: a record buffer is null");
    return PLH_ERR_INVALID;
  }
  Part& pt = fe->parts[part];
  pt.kps = r->kps; pt.desc = r->desc; pt.n = r->n; pt.nid = r->nid; pt.word = r->word; pt.bowWord = r->bow_word;
  pt.bowValue = r->bow_value; pt.bowN = r->bow_n; pt.kl = r->kl; pt.ldesc = r->ldesc; pt.lfn = r->lfn; pt.nl = r->nl;
  pt.mOrb = r->m_orb; pt.nmOrb = r->nm_orb; pt.mLine = r->m_line; pt.nmLine = r->nm_line;
  return PLH_OK;
}

plh_status plh_frontend_join(plh_frontend* fe, void* stream) {
  if (!fe) return PLH_ERR_INVALID;
  if (!fe->overlap) return PLH_OK;
  hipStream_t main = (hipStream_t)stream;
  for (Part& pt : fe->parts)
    if (pt.ran) {
      PLH_HIP(hipStreamWaitEvent(main, pt.evOrb, 0));
      PLH_HIP(hipStreamWaitEvent(main, pt.evLine, 0));
    }
  return PLH_OK;
}

plh_status plh_frontend_step(plh_frontend* fe, const uint8_t* d_imgs, size_t frame_stride, void* stream, int join) {
  if (!fe || !d_imgs || frame_stride < (size_t)fe->p.rows * fe->p.cols) {
    set_error("plh_frontend_step: invalid argument");
    return PLH_ERR_INVALID;
  }
  for (const Part& pt : fe->parts)
    if (!pt.kps) { set_error("plh_frontend_step: the record buffers of a sub-batch are not bound"); return PLH_ERR_INVALID; }
  PLH_HIP(hipSetDevice(fe->device));
  hipStream_t main = (hipStream_t)stream;
  PLH_HIP(hipEventRecord(fe->evStart, main));
  if (fe->around) {
    // Small resident batch: region growing runs several wavefronts per frame (k_lsd_grow_mw) whose 128-register build fills the
    // register file of every SIMD, so nothing runs beside it (measured: two streams = one stream).  The ORB chain of a step is
    // therefore placed AROUND region growing: this step's region growing waits for this step's ORB chain (plh_line_set_grow_events),
    // and the ORB chain of the NEXT step, enqueued behind it on its own stream, gets the CUs as the frames of this step's region
    // growing finish one by one -- it fills the kernel's tail (the slowest frame takes a quarter longer than the average one) and
    // then runs beside the KeyLine / LBD tail and the next step's image preparation.  36.8 -> 33.9 ms per 512-frame step.
    for (Part& pt : fe->parts) {
      const plh_status st = enqueue_orb(fe, pt, d_imgs + (size_t)pt.first * frame_stride, frame_stride, main);
      if (st != PLH_OK) return st;
      pt.ran = true;
    }
    for (Part& pt : fe->parts) {
      const plh_status st = enqueue_line(fe, pt, d_imgs + (size_t)pt.first * frame_stride, frame_stride, main);
      if (st != PLH_OK) return st;
    }
    return join ? plh_frontend_join(fe, stream) : PLH_OK;
  }
  // the critical-path (line) chains of all sub-batches first, then the ORB chains
  for (Part& pt : fe->parts) {
    const plh_status st = enqueue_line(fe, pt, d_imgs + (size_t)pt.first * frame_stride, frame_stride, main);
    if (st != PLH_OK) return st;
  }
  for (Part& pt : fe->parts) {
    const plh_status st = enqueue_orb(fe, pt, d_imgs + (size_t)pt.first * frame_stride, frame_stride, main);
    if (st != PLH_OK) return st;
    pt.ran = true;
  }
  return join ? plh_frontend_join(fe, stream) : PLH_OK;
}

// The records a tracker on another GPU needs, per sub-batch: n, kps, desc, nl, kl, ldesc, lfn.
plh_status plh_frontend_gather(plh_frontend* fe, plh_comm* comm, int root, void* const* recv, void* comm_stream) {
  if (!fe || !comm) return PLH_ERR_INVALID;
  PLH_HIP(hipSetDevice(fe->device));
  hipStream_t cs = (hipStream_t)comm_stream;
  int part = 0;
  for (Part& pt : fe->parts) {
    const size_t B = (size_t)pt.B;
    plh_gather_block blk[PLH_FRONTEND_GATHERED];
    const void* snd[PLH_FRONTEND_GATHERED] = {pt.n, pt.kps, pt.desc, pt.nl, pt.kl, pt.ldesc, pt.lfn};
    const size_t bytes[PLH_FRONTEND_GATHERED] = {B * 4, B * pt.ocap * sizeof(plh_keypoint), B * pt.ocap * 32, B * 4,
                                                 B * pt.lcap * sizeof(plh_keyline), B * pt.lcap * 32, B * pt.lcap * 24};
    for (int k = 0; k < PLH_FRONTEND_GATHERED; k++) {
      blk[k].send = snd[k];
      blk[k].recv = recv ? recv[part * PLH_FRONTEND_GATHERED + k] : nullptr;
      blk[k].bytes = bytes[k];
    }
    if (!pt.ran) { set_error("plh_frontend_gather: no step has been enqueued"); return PLH_ERR_INVALID; }
    if (fe->around && fe->overlap) {
      // small resident batch: the next step's ORB chain has to start while this step's region growing is still running (it
      // fills that kernel's tail, plh_frontend_step), so the ORB records go in a launch of their own as soon as they exist
      PLH_HIP(hipStreamWaitEvent(cs, pt.evOrb, 0));
      plh_status st = plh_gather_records(comm, blk, 3, root, comm_stream);
      if (st != PLH_OK) return st;
      PLH_HIP(hipEventRecord(pt.evFreeOrb, cs));
      pt.freeOrbValid = true;
      PLH_HIP(hipStreamWaitEvent(cs, pt.evLine, 0));
      st = plh_gather_records(comm, blk + 3, PLH_FRONTEND_GATHERED - 3, root, comm_stream);
      if (st != PLH_OK) return st;
    } else {
      // (with overlap off both chains ran on the caller's stream: the events were recorded there)
      PLH_HIP(hipStreamWaitEvent(cs, pt.evLine, 0));
      PLH_HIP(hipStreamWaitEvent(cs, pt.evOrb, 0));
      const plh_status st = plh_gather_records(comm, blk, PLH_FRONTEND_GATHERED, root, comm_stream);
      if (st != PLH_OK) return st;
      pt.freeOrbValid = false;
    }
    PLH_HIP(hipEventRecord(pt.evFree, cs));   // this sub-batch's next step waits for its own gather only
    pt.freeValid = true;
    part++;
  }
  return PLH_OK;
}

plh_status plh_frontend_gather_bytes(const plh_frontend* fe, int part, size_t bytes[PLH_FRONTEND_GATHERED]) {
  if (!fe || !bytes || part < 0 || part >= fe->nsplit) return PLH_ERR_INVALID;
  const Part& pt = fe->parts[part];
  const size_t B = (size_t)pt.B;
  const size_t b[PLH_FRONTEND_GATHERED] = {B * 4, B * pt.ocap * sizeof(plh_keypoint), B * pt.ocap * 32, B * 4,
                                           B * pt.lcap * sizeof(plh_keyline), B * pt.lcap * 32, B * pt.lcap * 24};
  for (int k = 0; k < PLH_FRONTEND_GATHERED; k++) bytes[k] = b[k];
  return PLH_OK;
}

plh_status plh_frontend_status(plh_frontend* fe, int* flags) {
  if (!fe || !flags) return PLH_ERR_INVALID;
  PLH_HIP(hipSetDevice(fe->device));
  int all = 0;
  for (Part& pt : fe->parts) {
    int f = 0;
    plh_status st = plh_orb_status(pt.orb, &f);
    if (st != PLH_OK) return st;
    all |= f;
    st = plh_line_status(pt.line, &f);
    if (st != PLH_OK) return st;
    all |= f << 8;
  }
  *flags = all;
  return PLH_OK;
}

}  // extern "C"
