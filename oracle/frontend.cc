// ORACLE (test infrastructure) -- the whole front end of one frame, and of a batch of frames sharded over std::threads.
//
// bench.py's cpu_baseline leg (SURVEY.md 8d ii: "the reference CPU path timed on the host cores of the same box, core count
// stated").  Rounds 1-3 drove the per-frame oracle calls from a Python thread pool: on the 256 hardware threads of the GPU box
// that ran at 4.9 % parallel efficiency -- interpreter lock and allocator contention, not the oracle.  This is the same work per
// frame as one step of the product path (plh_frontend_step), entirely in native code, one worker per std::thread with its own
// ORB handle and buffers, nothing shared but the read-only inputs:
//     ORBextractor::operator()                     src/ORBextractor.cc:1043-1105       plo_orb_extract
//     remap (Frame.cc:220-222) + LINEextractor     src/LineExtractor.cpp:26-93         plo_remap_linear_u8, plo_line_extract_ex
//     Frame::ComputeBoW                            src/Frame.cc:906-913                plo_bow_transform, plo_bow_vector
//     ORBmatcher::SearchByBoW, LSDmatcher::SearchDouble against the worker's previous frame (Tracking.cc:1151-1159)
// It is a timing harness: the results are discarded (a checksum keeps the calls alive), parity lives in tests/.
#include <malloc.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "plo.h"

namespace {

struct FrontendJob {
  const uint8_t* frames;
  int n, rows, cols;
  int nfeatures, nlevels, nlines, refine;
  const float* mapx;   // undistortion maps or null
  const float* mapy;
  // vocabulary (flat arrays of oracle/plo.py's Vocabulary)
  const uint8_t* node_desc;
  const int32_t *child_start, *child_count, *word_id;
  const float* weight;
  const double* word_weight;
  int L;
};

struct FrameOut {
  std::vector<uint8_t> desc, ldesc;
  std::vector<float> angle;
  std::vector<int32_t> nid;
  int n = 0, nl = 0;
};

unsigned long long run_worker(const FrontendJob& J, int first, int count) {
  plo_orb* orb = plo_orb_create(J.nfeatures, 1.2f, J.nlevels, 20, 7);
  const int cap = J.nfeatures + 16 * J.nlevels + 64, lcap = J.nlines + 1;
  std::vector<plo_keypoint> kps(cap);
  std::vector<plo_keyline> kl(lcap);
  std::vector<double> fn((size_t)lcap * 3), bv(cap);
  std::vector<int32_t> word(cap), bw(cap), m(cap), ml(lcap);
  std::vector<uint8_t> und((size_t)J.rows * J.cols), valid(cap, 1);
  FrameOut a, b;
  FrameOut *cur = &a, *prev = &b;
  for (FrameOut* f : {&a, &b}) { f->desc.resize((size_t)cap * 32); f->ldesc.resize((size_t)lcap * 32); f->angle.resize(cap); f->nid.resize(cap); }
  unsigned long long sum = 0;
  for (int k = 0; k < count; k++) {
    const uint8_t* img = J.frames + (size_t)((first + k) % J.n) * J.rows * J.cols;
    cur->n = plo_orb_extract(orb, img, J.rows, J.cols, (size_t)J.cols, kps.data(), cur->desc.data(), cap);
    if (cur->n < 0) cur->n = 0;
    for (int i = 0; i < cur->n; i++) cur->angle[i] = kps[i].angle;
    const uint8_t* src = img;
    if (J.mapx) {
      plo_remap_linear_u8(img, J.cols, J.rows, (size_t)J.cols, J.mapx, J.mapy, und.data(), (size_t)J.cols);
      src = und.data();
    }
    cur->nl = plo_line_extract_ex(src, J.rows, J.cols, (size_t)J.cols, nullptr, (unsigned)J.nlines, 0.0, kl.data(), cur->ldesc.data(),
                                  fn.data(), lcap, J.refine);
    if (cur->nl < 0) cur->nl = 0;
    plo_bow_transform(cur->desc.data(), cur->n, J.node_desc, J.child_start, J.child_count, J.word_id, J.weight, J.L, 4, cur->nid.data(),
                      word.data());
    sum += (unsigned long long)plo_bow_vector(word.data(), cur->n, J.word_weight, 0, 0, bw.data(), bv.data(), cap);
    if (k > 0) {
      sum += (unsigned long long)plo_orb_search_by_bow(prev->desc.data(), prev->angle.data(), prev->nid.data(), valid.data(), prev->n,
                                                        cur->desc.data(), cur->angle.data(), cur->nid.data(), cur->n, 50, 0.7f, 1, m.data());
      sum += (unsigned long long)plo_line_search_double(prev->ldesc.data(), prev->nl, cur->ldesc.data(), cur->nl, 50.0f, 0.7f, ml.data());
    }
    sum += (unsigned long long)cur->n + (unsigned long long)cur->nl;
    std::swap(cur, prev);
  }
  plo_orb_destroy(orb);
  return sum;
}

}  // namespace

extern "C" {

// `per_thread` frames on each of `nthreads` std::threads (worker t starts at frame t * per_thread of the `n` given, cyclically).
// Returns the wall-clock seconds of the parallel section; *checksum (may be null) keeps the work observable.
double plo_frontend_batch(const uint8_t* frames, int n, int rows, int cols, int nfeatures, int nlevels, int nlines, int refine,
                          const float* mapx, const float* mapy, const uint8_t* node_desc, const int32_t* child_start,
                          const int32_t* child_count, const int32_t* word_id, const float* weight, const double* word_weight, int L,
                          int nthreads, int per_thread, unsigned long long* checksum) {
  FrontendJob J{frames, n, rows, cols, nfeatures, nlevels, nlines, refine, mapx, mapy, node_desc, child_start, child_count, word_id,
                weight, word_weight, L};
  if (nthreads < 1) nthreads = 1;
  // The oracle's stages allocate their images and tables per call (tens of MB per frame, std::vector).  With glibc's defaults
  // every such block is its own mmap / munmap and page-faults in afresh: hundreds of threads then queue on the process's address-
  // space lock and the timing measures the kernel's mm, not the oracle (round 4's first native run: 9.7 % parallel efficiency on
  // 128 cores).  Keep the blocks in the threads' malloc arenas instead.
  // (glibc refuses an mmap threshold above HEAP_MAX_SIZE / 2 = 32 MiB, and setting any of these switches its own adaptive
  // threshold off: the first version of this asked for 1 GiB, was refused, and left every block >= 128 KiB on mmap.)
  mallopt(M_MMAP_THRESHOLD, 32 << 20);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_TOP_PAD, 16 << 20);
  std::vector<unsigned long long> sums((size_t)nthreads, 0);
  const auto t0 = std::chrono::steady_clock::now();
  if (nthreads == 1) {
    sums[0] = run_worker(J, 0, per_thread);
  } else {
    std::vector<std::thread> th;
    th.reserve((size_t)nthreads);
    for (int t = 0; t < nthreads; t++) th.emplace_back([&, t] { sums[(size_t)t] = run_worker(J, t * per_thread, per_thread); });
    for (auto& x : th) x.join();
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  // back to glibc's documented defaults (128 KiB thresholds, no padding) and the arenas' free tops returned: the tuning above is
  // for this call, not for whatever the process measures afterwards.  (glibc's adaptive mmap threshold stays off once
  // any of these has been set; the static defaults are what it starts from.)
  mallopt(M_MMAP_THRESHOLD, 128 << 10);
  mallopt(M_TRIM_THRESHOLD, 128 << 10);
  mallopt(M_TOP_PAD, 0);
  malloc_trim(0);
  unsigned long long s = 0;
  for (unsigned long long v : sums) s += v;
  if (checksum) *checksum = s;
  return dt;
}

}  // extern "C"
