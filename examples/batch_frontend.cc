// A C++ host of the batch front end: nothing but include/plslam_hip.h and the HIP runtime.
//
//   batch_frontend <frames.bin> <rows> <cols> <batch> <nsplit> <nfeatures> <nlines> <steps> <out.bin> <vocabulary.txt>
//                  [K0 K1 K2 K3 D0 D1 D2 D3 D4]
//
// frames.bin: batch x rows x cols bytes; vocabulary.txt: a DBoW2 text vocabulary.  Runs `steps` un-joined steps over the resident batch, joins, and writes for every
// frame: n, keypoints, descriptors, FeatureVector nodes, words, BowVector, keylines, LBD, line equations and both match
// lists -- tests/test_frontend_example.py compares them with the oracle.  This is what replaces the reference's per-frame
//   Frame::Frame(...) { thread(ExtractORB); thread(ExtractLSD); ... ComputeBoW }   (Frame.cc:193-276, 906-913)
// for a host that has a batch of frames (a recorded sequence, a camera rig) instead of one.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/plslam_hip.h"

#define CHECK(x)                                                                       \
  do {                                                                                 \
    if ((x) != PLH_OK) {                                                               \
      std::fprintf(stderr, "%s failed: %s\n", #x, plh_last_error());                   \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)
#define HIPCHECK(x)                                                                    \
  do {                                                                                 \
    if ((x) != hipSuccess) {                                                           \
      std::fprintf(stderr, "%s failed\n", #x);                                         \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

template <typename T>
static bool dump(FILE* f, const T* dptr, size_t count) {
  std::vector<T> h(count);
  if (hipMemcpy(h.data(), dptr, count * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) return false;
  return std::fwrite(h.data(), sizeof(T), count, f) == count;
}

int main(int argc, char** argv) {
  if (argc < 11) {
    std::fprintf(stderr, "usage: %s frames.bin rows cols batch nsplit nfeatures nlines steps out.bin vocabulary.txt [K x4 D x5]\n", argv[0]);
    return 2;
  }
  const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]), batch = std::atoi(argv[4]), nsplit = std::atoi(argv[5]);
  const int nfeatures = std::atoi(argv[6]), nlines = std::atoi(argv[7]), steps = std::atoi(argv[8]);
  std::vector<uint8_t> frames((size_t)batch * rows * cols);
  {
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(frames.data(), 1, frames.size(), f) != frames.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    std::fclose(f);
  }
  // the vocabulary: a DBoW2 text file (ORBvoc.txt's format, TemplatedVocabulary.h:1350-1438), as System.cc:66-84 loads it
  plh_vocab* voc = nullptr;
  CHECK(plh_vocab_load_text(argv[10], 0, &voc));

  plh_frontend_params p;
  std::memset(&p, 0, sizeof(p));
  p.rows = rows; p.cols = cols;
  p.orb.nfeatures = nfeatures; p.orb.scale_factor = 1.2f; p.orb.nlevels = 8; p.orb.ini_th_fast = 20; p.orb.min_th_fast = 7;
  p.line.num_octaves = 1; p.line.scale = 1.2f; p.line.n_lsd_feature = (uint32_t)nlines; p.line.min_line_length = 0.0;
  if (argc >= 20) {
    p.undistort = 1;
    for (int i = 0; i < 4; i++) p.K[i] = (float)std::atof(argv[11 + i]);
    for (int i = 0; i < 5; i++) p.D[i] = (float)std::atof(argv[15 + i]);
  }
  p.bow_levelsup = 4; p.orb_th_low = 50; p.orb_nnratio = 0.7f; p.orb_check_orientation = 1; p.line_th = 50.f; p.line_nnratio = 0.7f;
  plh_frontend* fe = nullptr;
  CHECK(plh_frontend_create(&p, voc, batch, nsplit, 0, &fe));

  uint8_t* d_imgs = nullptr;
  HIPCHECK(hipMalloc((void**)&d_imgs, frames.size()));
  HIPCHECK(hipMemcpy(d_imgs, frames.data(), frames.size(), hipMemcpyHostToDevice));
  hipStream_t stream;
  HIPCHECK(hipStreamCreate(&stream));
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  CHECK(plh_frontend_step(fe, d_imgs, (size_t)rows * cols, stream, 1));   // warm-up (first-use allocations)
  HIPCHECK(hipStreamSynchronize(stream));
  HIPCHECK(hipEventRecord(e0, stream));
  for (int s = 0; s < steps; s++) CHECK(plh_frontend_step(fe, d_imgs, (size_t)rows * cols, stream, 0));   // un-joined: steps overlap
  CHECK(plh_frontend_join(fe, stream));
  HIPCHECK(hipEventRecord(e1, stream));
  HIPCHECK(hipStreamSynchronize(stream));
  float ms = 0;
  HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
  int flags = 0;
  CHECK(plh_frontend_status(fe, &flags));
  if (flags) { std::fprintf(stderr, "capacity flags 0x%x\n", flags); return 1; }
  std::printf("%d frames x %d steps in %.3f ms: %.0f frames/s\n", batch, steps, ms, 1e3 * batch * steps / ms);

  FILE* out = std::fopen(argv[9], "wb");
  if (!out) return 1;
  const int32_t hdr[4] = {batch, plh_frontend_parts(fe), 0, 0};
  std::fwrite(hdr, 4, 4, out);
  for (int part = 0; part < plh_frontend_parts(fe); part++) {
    plh_frontend_records r;
    CHECK(plh_frontend_records_of(fe, part, &r));
    const int32_t ph[4] = {r.first, r.frames, r.orb_capacity, r.line_capacity};
    std::fwrite(ph, 4, 4, out);
    const size_t B = (size_t)r.frames, oc = (size_t)r.orb_capacity, lc = (size_t)r.line_capacity;
    bool ok = dump(out, r.n, B) && dump(out, r.kps, B * oc) && dump(out, r.desc, B * oc * 32) && dump(out, r.nid, B * oc) &&
              dump(out, r.word, B * oc) && dump(out, r.bow_n, B) && dump(out, r.bow_word, B * oc) && dump(out, r.bow_value, B * oc) &&
              dump(out, r.nl, B) && dump(out, r.kl, B * lc) && dump(out, r.ldesc, B * lc * 32) && dump(out, r.lfn, B * lc * 3) &&
              dump(out, r.nm_orb, B) && dump(out, r.m_orb, B * oc) && dump(out, r.nm_line, B) && dump(out, r.m_line, B * lc);
    if (!ok) { std::fprintf(stderr, "cannot write the records of sub-batch %d\n", part); return 1; }
  }
  std::fclose(out);
  plh_frontend_destroy(fe);
  plh_vocab_destroy(voc);
  (void)hipFree(d_imgs);
  return 0;
}
