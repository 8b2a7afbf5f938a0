// oracle/_ref: the reference's own ORB extractor (src/ORBextractor.cc + include/ORBextractor.h, compiled from the sources
// where they lie under /root/reference by oracle/ref/build_ref.sh) behind a small C interface.  The OpenCV types come
// from oracle/ref/stub; cv::resize / cv::GaussianBlur / cv::FAST / cv::fastAtan2 are the ORACLE's restatements, so what
// this pins is the reference's control logic on top of them (ORBextractor.cc:410-470 constructor tables, :765-853
// ComputeKeyPointsOctTree, :481-763 quad tree, :77-147 IC_Angle / computeOrbDescriptor, :1043-1132 operator() and
// ComputePyramid).  TEST INFRASTRUCTURE ONLY.
#include <cstdint>
#include <cstring>
#include <vector>

#include <new>
#include <sys/mman.h>

#include "ORBextractor.h"

// ---------------------------------------------------------------------------------------------------------------
// DistributeOctTree orders nodes of equal size by the ADDRESS of their std::list node (sort of pair<int, ExtractorNode*>,
// ORBextractor.cc:684): under glibc malloc that order changes from run to run (observed: 1006 / 1007 keypoints on the same
// image).  The oracle pins the order "most recently created node first" = what a bump allocator gives.  To compare like
// with like, allocations made INSIDE ref_orb_extract come from a monotonic arena (addresses grow with creation time,
// nothing is reused), private to this library (hidden visibility + -Bsymbolic: the rest of the process keeps malloc).
// ---------------------------------------------------------------------------------------------------------------
namespace {
const size_t kArenaBytes = (size_t)2 << 30;
char* g_arena = nullptr;
size_t g_top = 0;
bool g_active = false;
inline bool in_arena(const void* p) { return g_arena && (const char*)p >= g_arena && (const char*)p < g_arena + kArenaBytes; }
void* arena_alloc(size_t n) {
  if (!g_arena) {
    g_arena = (char*)mmap(nullptr, kArenaBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g_arena == (char*)MAP_FAILED) { g_arena = nullptr; return nullptr; }
  }
  const size_t a = (g_top + 15) & ~(size_t)15;
  if (a + n > kArenaBytes) return nullptr;
  g_top = a + n;
  return g_arena + a;
}
}  // namespace
void* operator new(size_t n) {
  void* p = g_active ? arena_alloc(n) : nullptr;
  if (!p) p = std::malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void* operator new[](size_t n) { return operator new(n); }
void operator delete(void* p) noexcept { if (p && !in_arena(p)) std::free(p); }
void operator delete[](void* p) noexcept { operator delete(p); }
void operator delete(void* p, size_t) noexcept { operator delete(p); }
void operator delete[](void* p, size_t) noexcept { operator delete(p); }

extern "C" {

__attribute__((visibility("default"))) void* ref_orb_create(int nfeatures, float scale, int nlevels, int ini_th, int min_th) {
  return new ORB_SLAM2::ORBextractor(nfeatures, scale, nlevels, ini_th, min_th);
}
__attribute__((visibility("default"))) void ref_orb_destroy(void* h) { delete static_cast<ORB_SLAM2::ORBextractor*>(h); }

// returns the number of keypoints (or -1 if cap is too small); keypoints in plo_keypoint layout (== cv::KeyPoint)
__attribute__((visibility("default"))) int ref_orb_extract(void* h, const uint8_t* img, int rows, int cols, size_t step, plo_keypoint* kps, uint8_t* desc, int cap) {
  ORB_SLAM2::ORBextractor* ex = static_cast<ORB_SLAM2::ORBextractor*>(h);
  // everything the call allocates (pyramid, candidate vectors, list nodes) comes from the arena, restarted per call: what
  // survives in the extractor (mvImagePyramid) is overwritten by the next call before it is read, and deleting arena
  // memory is a no-op
  g_top = 0;
  g_active = true;
  int n;
  {
    cv::Mat image(rows, cols, CV_8UC1, const_cast<uint8_t*>(img), step), mask, d;
    std::vector<cv::KeyPoint> k;
    (*ex)(image, mask, k, d);
    n = (int)k.size();
    if (n > cap) n = -1;
    for (int i = 0; i < n; i++) {
      kps[i].x = k[i].pt.x; kps[i].y = k[i].pt.y; kps[i].size = k[i].size; kps[i].angle = k[i].angle;
      kps[i].response = k[i].response; kps[i].octave = k[i].octave; kps[i].class_id = k[i].class_id;
      std::memcpy(desc + (size_t)i * 32, d.ptr<uint8_t>(i), 32);
    }
  }
  g_active = false;
  return n;
}

__attribute__((visibility("default"))) void ref_orb_tables(void* h, float* scale, float* sigma2, int* nlevels) {
  ORB_SLAM2::ORBextractor* ex = static_cast<ORB_SLAM2::ORBextractor*>(h);
  const std::vector<float> s = ex->GetScaleFactors(), g = ex->GetScaleSigmaSquares();
  *nlevels = ex->GetLevels();
  for (size_t i = 0; i < s.size(); i++) { scale[i] = s[i]; sigma2[i] = g[i]; }
}

}  // extern "C"
