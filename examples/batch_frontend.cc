// A C++ host of the batch front end: nothing but include/plslam_hip.h and the HIP runtime.
//
//   batch_frontend <frames> <rows> <cols> <batch> <nsplit> <nfeatures> <nlines> <steps> <out.bin> <vocabulary.txt>
//                  [K0 K1 K2 K3 D0 D1 D2 D3 D4] [--refine std|adv]
//
// frames: a raw file of rows x cols 8-bit planes, one binary PGM (P5), or a directory of such files (sorted by name), cycled to
// fill the batch -- a recorded sequence converted once to PGM, as mono_tum.cc / mono_kitti.cc read theirs with cv::imread.
// Without K / D the camera of the frame size is taken: Examples/Monocular/TUM1.yaml for 640 x 480 (with its distortion),
// KITTI00-02.yaml (no distortion) for 1241 x 376.  vocabulary.txt: a DBoW2 text vocabulary.  --refine picks
// cv::LineSegmentDetector's refine level (default: the library's build-time default).  Runs `steps` un-joined steps over the resident batch, joins, and writes for every
// frame: n, keypoints, descriptors, FeatureVector nodes, words, BowVector, keylines, LBD, line equations and both match
// lists -- tests/test_frontend_example.py compares them with the oracle.  This is what replaces the reference's per-frame
//   Frame::Frame(...) { thread(ExtractORB); thread(ExtractLSD); ... ComputeBoW }   (Frame.cc:193-276, 906-913)
// for a host that has a batch of frames (a recorded sequence, a camera rig) instead of one.
#include <hip/hip_runtime.h>

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include <dirent.h>
#include <sys/stat.h>

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

// One file of frames appended to `seq`: a binary PGM (P5, maxval <= 255, rows x cols) or raw rows x cols planes.
static bool read_frames_file(const std::string& path, int rows, int cols, std::vector<uint8_t>& seq) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::vector<uint8_t> data;
  uint8_t buf[1 << 16];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) data.insert(data.end(), buf, buf + got);
  std::fclose(f);
  const size_t plane = (size_t)rows * cols;
  if (data.size() >= 2 && data[0] == 'P' && data[1] == '5') {
    size_t pos = 2;
    long tok[3];
    for (int k = 0; k < 3;) {   // width, height, maxval; '#' comments run to the end of the line
      while (pos < data.size() && std::isspace(data[pos])) pos++;
      if (pos < data.size() && data[pos] == '#') { while (pos < data.size() && data[pos] != '\n') pos++; continue; }
      long v = 0;
      size_t start = pos;
      while (pos < data.size() && std::isdigit(data[pos])) v = v * 10 + (data[pos++] - '0');
      if (pos == start) return false;
      tok[k++] = v;
    }
    pos++;   // the single whitespace byte behind maxval
    if (tok[0] != cols || tok[1] != rows || tok[2] <= 0 || tok[2] > 255 || data.size() - pos < plane) {
      std::fprintf(stderr, "%s: a %ldx%ld PGM (maxval %ld), the plan is %dx%d 8-bit\n", path.c_str(), tok[0], tok[1], tok[2], cols, rows);
      return false;
    }
    seq.insert(seq.end(), data.begin() + (long)pos, data.begin() + (long)(pos + plane));
    return true;
  }
  if (data.empty() || data.size() % plane) {
    std::fprintf(stderr, "%s: %zu bytes is not a whole number of %dx%d frames\n", path.c_str(), data.size(), cols, rows);
    return false;
  }
  seq.insert(seq.end(), data.begin(), data.end());
  return true;
}

int main(int argc, char** argv) {
  int refine = -1;   // the library's default
  std::vector<char*> av;
  for (int i = 0; i < argc; i++) {
    if (!std::strcmp(argv[i], "--refine") && i + 1 < argc) { refine = !std::strcmp(argv[i + 1], "adv") ? PLH_LSD_REFINE_ADV : PLH_LSD_REFINE_STD; i++; }
    else av.push_back(argv[i]);
  }
  argc = (int)av.size();
  argv = av.data();
  if (argc < 11) {
    std::fprintf(stderr, "usage: %s frames rows cols batch nsplit nfeatures nlines steps out.bin vocabulary.txt [K x4 D x5] [--refine std|adv]\n", argv[0]);
    return 2;
  }
  const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]), batch = std::atoi(argv[4]), nsplit = std::atoi(argv[5]);
  const int nfeatures = std::atoi(argv[6]), nlines = std::atoi(argv[7]), steps = std::atoi(argv[8]);
  std::vector<uint8_t> frames((size_t)batch * rows * cols);
  {
    std::vector<std::string> files;
    struct stat st;
    if (stat(argv[1], &st) == 0 && S_ISDIR(st.st_mode)) {
      if (DIR* d = opendir(argv[1])) {
        while (dirent* e = readdir(d)) {
          const std::string n = e->d_name;
          const size_t dot = n.rfind('.');
          const std::string ext = dot == std::string::npos ? "" : n.substr(dot);
          if (ext == ".pgm" || ext == ".bin" || ext == ".raw" || ext == ".gray") files.push_back(std::string(argv[1]) + "/" + n);
        }
        closedir(d);
      }
      std::sort(files.begin(), files.end());
    } else {
      files.push_back(argv[1]);
    }
    std::vector<uint8_t> seq;
    for (const std::string& fn : files)
      if (!read_frames_file(fn, rows, cols, seq)) { std::fprintf(stderr, "cannot read %s\n", fn.c_str()); return 1; }
    const size_t plane = (size_t)rows * cols, nseq = seq.size() / plane;
    if (!nseq) { std::fprintf(stderr, "%s: no frames\n", argv[1]); return 1; }
    for (int b = 0; b < batch; b++) std::memcpy(&frames[(size_t)b * plane], &seq[((size_t)b % nseq) * plane], plane);   // the sequence, cycled
    std::printf("%zu frames read from %s\n", nseq, argv[1]);
  }
  // the vocabulary: a DBoW2 text file (ORBvoc.txt's format, TemplatedVocabulary.h:1350-1438), as System.cc:66-84 loads it
  plh_vocab* voc = nullptr;
  CHECK(plh_vocab_load_text(argv[10], 0, &voc));

  plh_frontend_params p;
  std::memset(&p, 0, sizeof(p));   // (lsd_refine = 0 = PLH_FRONTEND_REFINE_LIBRARY: the library's default)
  p.struct_size = (uint32_t)sizeof(p);
  p.rows = rows; p.cols = cols;
  p.orb.nfeatures = nfeatures; p.orb.scale_factor = 1.2f; p.orb.nlevels = 8; p.orb.ini_th_fast = 20; p.orb.min_th_fast = 7;
  p.line.num_octaves = 1; p.line.scale = 1.2f; p.line.n_lsd_feature = (uint32_t)nlines; p.line.min_line_length = 0.0;
  if (argc >= 20) {
    p.undistort = 1;
    for (int i = 0; i < 4; i++) p.K[i] = (float)std::atof(argv[11 + i]);
    for (int i = 0; i < 5; i++) p.D[i] = (float)std::atof(argv[15 + i]);
  } else if (rows == 480 && cols == 640) {   // Examples/Monocular/TUM1.yaml:8-17
    const float K[4] = {517.306408f, 516.469215f, 318.643040f, 255.313989f}, D[5] = {0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f};
    p.undistort = 1;
    std::memcpy(p.K, K, sizeof(K));
    std::memcpy(p.D, D, sizeof(D));
    std::printf("camera: TUM1.yaml\n");
  } else if (rows == 376 && cols == 1241) {  // Examples/Monocular/KITTI00-02.yaml: zero distortion, no remap (Frame.cc:917-921)
    std::printf("camera: KITTI00-02.yaml\n");
  }
  p.lsd_refine = refine < 0 ? PLH_FRONTEND_REFINE_LIBRARY : (refine ? PLH_FRONTEND_REFINE_ADV : PLH_FRONTEND_REFINE_STD);
  std::printf("cv::LineSegmentDetector refine level: %s\n",
              (refine < 0 ? plh_lsd_refine_default() : refine) == PLH_LSD_REFINE_ADV ? "LSD_REFINE_ADV" : "LSD_REFINE_STD");
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
