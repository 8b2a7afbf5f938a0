// Host side of the line extractor: plan (LSD constants computed in double exactly as flsd() does),
// workspace, launch sequence and the C ABI (include/plslam_hip.h, plh_line_*).
#include <algorithm>
#include <cmath>
#include <mutex>
#include <new>
#include <vector>

#include "line_plan.h"
#include "plh_common.h"

namespace plh {
void launch_remap(const LineDeviceArgs& a, hipStream_t s);
void launch_blur7(const uint8_t* src, long long sStride, int sPitch, uint8_t* dst, long long dStride, int dPitch, int w, int h,
                  int batch, const int taps[7], hipStream_t s);
void launch_resize(const uint8_t* src, long long sStride, int sPitch, int sw, int sh, uint8_t* dst, long long dStride, int dPitch, int dw,
                   int dh, int batch, const ResizeTap* xtab, const ResizeTap* ytab, int tileTP, int tileTR, hipStream_t s);
void launch_lsd_grad(const LineDeviceArgs& a, hipStream_t s);
void launch_lsd_angle_table(LsdAngleEntry* tab, hipStream_t s);
void launch_lsd_order(const LineDeviceArgs& a, hipStream_t s);
size_t lsd_order_work_u32();
void launch_lsd_grow(const LineDeviceArgs& a, hipStream_t s);
plh_status lsd_grow_request_lds();
void launch_lsd_rects(const LineDeviceArgs& a, hipStream_t s);
void launch_lsd_lgamma_table(double* t, int n, hipStream_t s);
size_t lsd_adv_rec_bytes();
void launch_pyr_down5(const uint8_t* src, long long sStride, int sw, int sh, uint8_t* dst, long long dStride, int dw, int dh, int batch,
                      hipStream_t s);
void launch_keylines(const LineDeviceArgs& a, plh_keyline* kl, double* fn, int* n, hipStream_t s);
void launch_sobel(const LineDeviceArgs& a, hipStream_t s);
void launch_lbd(const LineDeviceArgs& a, const plh_keyline* kl, const int* n, const float* coef, uint8_t* desc, hipStream_t s);
size_t lsd_grow_lds_bytes(int spitch, int sh);
}  // namespace plh

using namespace plh;

// The ll_angle() table (line_plan.h) is a property of the device, not of a handle: one copy per device, filled by the
// first plh_line_create on it and released with the last handle.
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_angleMu;
LsdAngleEntry* g_angleTab[kMaxDevices] = {};
int g_angleRefs[kMaxDevices] = {};
const LsdAngleEntry* angle_table_acquire(int device) {   // current device == `device`
  if (device < 0 || device >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lk(g_angleMu);
  if (!g_angleTab[device]) {
    LsdAngleEntry* t = nullptr;
    if (hipMalloc((void**)&t, ((size_t)LSD_ANGLE_ROWS << LSD_ANGLE_PITCH_LOG2) * sizeof(LsdAngleEntry)) != hipSuccess) return nullptr;
    launch_lsd_angle_table(t, nullptr);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(t); return nullptr; }
    g_angleTab[device] = t;
  }
  g_angleRefs[device]++;
  return g_angleTab[device];
}
void angle_table_release(int device) {
  if (device < 0 || device >= kMaxDevices) return;
  std::lock_guard<std::mutex> lk(g_angleMu);
  if (g_angleRefs[device] > 0 && --g_angleRefs[device] == 0) {
    (void)hipFree(g_angleTab[device]);
    g_angleTab[device] = nullptr;
  }
}
}  // namespace

struct plh_line {
  plh_line_params p;
  plh_line* oct1 = nullptr;      // LINEextractor(numOctaves = 2): the plan of the second octave (half size, one octave), owned
  uint8_t* dOctImg = nullptr;    // ... and its detection image, pyrDown of the (undistorted) frame
  int device, rows, cols, maxBatch;
  LineDeviceArgs a;   // template (pointers filled at create)
  int taps075[7], taps1[7];
  int rszTP = 0, rszTR = 0;   // k_resize_u8 source tile of a 256 x 16 output block (pitch in bytes, rows)
  // device buffers
  uint8_t *dUndist = nullptr, *dTmpA = nullptr, *dScaled = nullptr, *dMask = nullptr;
  void* dAdv = nullptr;   // LSD_REFINE_ADV: maxBatch x segCap LsdAdvRec + the angle planes, allocated by the first extract call at that level
  double* dLgamma = nullptr;   // ... and lsd_log_gamma(i), i = 1 .. sw sh + 1 (nfa()'s arguments are integers: lsd_rect_dev.h)
  uint32_t *dArena = nullptr, *dDxdy = nullptr;   // dArena: per-frame blocks (line_plan.h, arenaStride)
  // multi-wavefront region growing (small batches): transaction logs and private mark planes, allocated on first use
  uint32_t* dMwReg = nullptr;
  uint8_t* dMwMark = nullptr;
  uint16_t* dMwHint = nullptr;
  long long mwHintFrames = 0;
  long long mwWaveSlots = 0;   // (frame, wavefront) pairs the two buffers hold
  int growWaves = -1;          // plh_line_set_grow_waves
  int mwLag = 448, mwDrainGap = 8;   // plh_line_set_grow_tuning
  unsigned int* dQmax = nullptr;
  int *dNOrdered = nullptr, *dNSegs = nullptr, *dStatus = nullptr;
  hipEvent_t growGate = nullptr, growDone = nullptr;   // caller's events around the region-growing launch (plh_line_set_grow_events)
  hipEvent_t doneEv = nullptr;   // recorded behind the last kernel of every extract call: plh_line_status (and a workspace
                                 // re-allocation) waits on it -- the caller's stream may be gone by then
  bool doneValid = false;
  float* dCoef = nullptr;
  RemapTap* dMap = nullptr;
  ResizeTap *dXtab = nullptr, *dYtab = nullptr;
  // staging (host-buffer entry points)
  uint8_t *dImgs = nullptr, *dDesc = nullptr;
  plh_keyline* dKl = nullptr;
  double* dFn = nullptr;
  int* dN = nullptr;
  hipStream_t stream = nullptr;
  bool hasUndistort = false;
  // optional per-stage timing with HIP events on the caller's stream (bench.py roofline leg)
  bool profiling = false;
  std::vector<hipEvent_t> evPool;
  std::vector<int> evKind;
  size_t evUsed = 0;
  double kernelMs[4] = {0, 0, 0, 0};
  int kernelLaunches[4] = {0, 0, 0, 0};
};

static void line_prof_mark(plh_line* h, int kind, hipStream_t s) {
  if (!h->profiling) return;
  if (h->evUsed == h->evPool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    h->evPool.push_back(e);
    h->evKind.push_back(kind);
  }
  h->evKind[h->evUsed] = kind;
  (void)hipEventRecord(h->evPool[h->evUsed++], s);
}

namespace {

inline int cv_round_h(float v) { return (int)lrintf(v); }
inline int cv_floor_h(float v) { int i = (int)v; return i - (i > v); }

// getGaussianKernel(n, sigma, CV_32F) -> cvRound(k*256), centred in a 7-tap array (classic 8-bit path).
void gaussian_q8_7(int n, double sigma, int out[7]) {
  float cf[7];
  const double scale2X = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    const double x = i - (n - 1) * 0.5;
    cf[i] = (float)std::exp(scale2X * x * x);
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < 7; i++) out[i] = 0;
  const int off = (7 - n) / 2;
  for (int i = 0; i < n; i++) {
    cf[i] = (float)(cf[i] * sum);
    out[off + i] = cv_round_h(cf[i] * 256.f);
  }
}

// cv::resize(..., Size(), fx, fy, INTER_LINEAR): scale = 1/fx exactly.
void resize_axis_scale(int ssize, int dsize, double inv_scale, bool isX, std::vector<ResizeTap>& out) {
  const double scale = 1.0 / inv_scale;
  int xmax = dsize;
  for (int d = 0; d < dsize; d++) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = cv_floor_h(f);
    f -= s;
    if (isX) {
      if (s < 0) { f = 0; s = 0; }
      if (s + 1 >= ssize) {
        xmax = std::min(xmax, d);
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
      }
    }
    ResizeTap t;
    t.ofs = (short)s;
    t.a0 = (short)cv_round_h((1.f - f) * 2048.f);
    t.a1 = (short)cv_round_h(f * 2048.f);
    t.pad = 0;
    if (isX && d >= xmax) { t.a0 = 2048; t.a1 = 0; }
    out.push_back(t);
  }
}

}  // namespace

extern "C" {

plh_status plh_line_destroy(plh_line* h) {
  if (!h) return PLH_OK;
  (void)hipSetDevice(h->device);
  if (h->oct1) {
    h->oct1->doneEv = nullptr;   // (the parent's event, shared)
    plh_line_destroy(h->oct1);
  }
  void* ptrs[] = {h->dOctImg, h->dAdv, h->dLgamma, h->dUndist, h->dTmpA, h->dScaled, h->dMask, h->dArena, h->dDxdy, h->dQmax, h->dMwReg, h->dMwMark, h->dMwHint,
                  h->dNOrdered, h->dNSegs, h->dStatus, h->dMap, h->dCoef, h->dXtab, h->dYtab, h->dImgs, h->dDesc,
                  h->dKl, h->dFn, h->dN};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  for (hipEvent_t e : h->evPool) (void)hipEventDestroy(e);
  if (h->doneEv) (void)hipEventDestroy(h->doneEv);
  if (h->a.angleTab) angle_table_release(h->device);
  delete h;
  return PLH_OK;
}

plh_status plh_line_create(const plh_line_params* p, int device, int rows, int cols, int max_batch, plh_line** out) {
  if (!p || !out || rows < 16 || cols < 16 || max_batch <= 0 || rows > 8192 || cols > 8192) {
    set_error("plh_line_create: invalid argument");
    return PLH_ERR_INVALID;
  }
  // LINEextractor(numOctaves, scale, ...): what the reference does with more than one octave (oracle/line.cc plo_line_extract_oct):
  // lsd->detect() takes `int scale` (1.2 -> 1, 2.0 -> 2) and pyrDown()s to Size(cols / scale, rows / scale), which cv::pyrDown
  // accepts only for (int)scale == 2 (it throws otherwise); with three or more octaves BinaryDescriptor::computeImpl runs into
  // undefined behaviour (it erases from the per-line vectors it iterates over).  The one defined multi-octave configuration --
  // two octaves, scale in [2, 3) -- is supported; the others are refused here, where the reference throws or is undefined.
  if (p->num_octaves < 1 || p->num_octaves > 2) {
    set_error("plh_line_create: numOctaves %d: the reference's behaviour is undefined for three or more octaves "
              "(binary_descriptor_custom.cpp:617-626); 1 and 2 are supported", p->num_octaves);
    return PLH_ERR_INVALID;
  }
  if (p->num_octaves == 2 && (int)p->scale != 2) {
    set_error("plh_line_create: numOctaves 2 with scale %g: the reference throws (cv::pyrDown asserts |2 dst - src| <= 2: LSDDetector::detect "
              "takes an int scale, only (int)scale == 2 passes)", (double)p->scale);
    return PLH_ERR_INVALID;
  }
  if (p->num_octaves == 2 && (rows < 32 || cols < 32)) {
    set_error("plh_line_create: two octaves need an image of at least 32 x 32");
    return PLH_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    set_error("plh_line_create: no HIP device %d", device);
    return PLH_ERR_NO_DEVICE;
  }
  PLH_HIP(hipSetDevice(device));
  plh_line* h = new (std::nothrow) plh_line();
  if (!h) return PLH_ERR_ALLOC;
  h->p = *p; h->device = device; h->rows = rows; h->cols = cols; h->maxBatch = max_batch;
  LineDeviceArgs& a = h->a;
  memset(&a, 0, sizeof(a));
  a.w = cols; a.h = rows;
  a.sw = (int)lrint(cols * 0.8); a.sh = (int)lrint(rows * 0.8);
  a.spitch = align_up(a.sw, 64);
  a.fullStride = align_up<long long>((long long)cols * rows, 256);
  a.scaledStride = align_up<long long>((long long)a.spitch * lsd_rec_rows(a.sh), 256);   // (the record plane is made of 4 x 4 blocks)
  // flsd() constants
  const double ANG_TH = 22.5, QUANT = 2.0;
  a.prec = M_PI * ANG_TH / 180;
  a.p = ANG_TH / 180;
  a.densityTh = 0.7;
  const double rho = QUANT / std::sin(a.prec);
  unsigned q = 0;
  while (std::sqrt((double)(int)(q + 1) / 4.0) <= rho) q++;   // largest q with norm <= rho  (NOTDEF)
  a.qThresh = q;
  {   // margins of k_lsd_grow's direction pre-test (LSD_ALIGN_MARGIN_DEG in lsd_grow.hip: 0.05 degrees)
    const double m = 0.05 * 3.14159265358979323846 / 180.0;
    a.alignFast = a.prec + m < 89.0 * 3.14159265358979323846 / 180.0 ? 1 : 0;
    const double ci = a.prec > m ? std::cos(a.prec - m) : 2.0, co = std::cos(a.prec + m);
    a.alignCin2 = (float)(ci * ci);
    a.alignCout2 = (float)(co * co);
  }
  const double LOG_NT = 5 * (std::log10(double(a.sw)) + std::log10(double(a.sh))) / 2 + std::log10(11.0);
  a.minRegSize = (int)(size_t)(-LOG_NT / std::log10(a.p));
  a.logNT = LOG_NT;
  a.refineAdv = PLH_LSD_REFINE_DEFAULT == PLH_LSD_REFINE_ADV ? 1 : 0;
  a.screen = 1;
  a.screenLo = (float)(a.densityTh * (1.0 - 2e-5));
  a.screenHi = (float)(a.densityTh * (1.0 + 2e-5));
  a.segCap = (a.sw * a.sh) / std::max(a.minRegSize, 1) + 16;
  a.nFeature = (int)p->n_lsd_feature;
  a.minLineLength = p->min_line_length;
  a.outCap = a.nFeature + 1;
  gaussian_q8_7(7, 0.6 / 0.8, h->taps075);
  gaussian_q8_7(5, 1.0, h->taps1);
  for (const int* tp : {h->taps075, h->taps1}) {   // k_blur7_u8 multiplies bytes by taps in dot4s and keeps 16-bit row sums
    int sum = 0;
    bool ok = true;
    for (int i = 0; i < 7; i++) { ok = ok && tp[i] >= 0 && tp[i] <= 255; sum += tp[i]; }
    if (!ok || sum > 257) {
      set_error("plh_line_create: Gaussian taps outside the range of the packed blur kernel");
      delete h;
      return PLH_ERR_INVALID;
    }
  }
  if (a.sw >= 65536 || a.sh >= 32768) {   // packed queue coordinates x:16 | y:16, mark in bit 31 of q
    set_error("plh_line_create: image too large (%d x %d scaled)", a.sw, a.sh);
    delete h;
    return PLH_ERR_INVALID;
  }
  std::vector<ResizeTap> xt, yt;
  resize_axis_scale(cols, a.sw, 0.8, true, xt);
  resize_axis_scale(rows, a.sh, 0.8, false, yt);
  {   // exact extent of the source tile behind any 256 x 16 output block (k_resize_u8 stages it in LDS)
    int maxW = 0, maxH = 0;
    for (int x0 = 0; x0 < a.sw; x0 += 256) {
      const int lo = xt[x0].ofs & ~3, hi = std::min((int)xt[std::min(x0 + 255, a.sw - 1)].ofs + 1, cols - 1);
      maxW = std::max(maxW, hi - lo + 1);
    }
    for (int y0 = 0; y0 < a.sh; y0 += 16) {
      const int lo = std::min(std::max((int)yt[y0].ofs, 0), rows - 1);
      const int hi = std::min(std::max((int)yt[std::min(y0 + 15, a.sh - 1)].ofs + 1, 0), rows - 1);
      maxH = std::max(maxH, hi - lo + 1);
    }
    h->rszTP = (maxW + 3) & ~3;
    h->rszTR = maxH;
  }
  // LBD weights: BinaryDescriptor ctor, binary_descriptor_custom.cpp:217-259 (integer divisions as written there)
  std::vector<float> coef(21 + 63);
  {
    double u = (LBD_BAND_WIDTH * 3 - 1) / 2;
    double sigma = (LBD_BAND_WIDTH * 2 + 1) / 2;
    double invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < 21; i++) { double dis = i - u; coef[i] = (float)std::exp(dis * dis * invsigma2); }
    u = (LBD_NUM_BANDS * LBD_BAND_WIDTH - 1) / 2;
    sigma = u;
    invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < 63; i++) { double dis = i - u; coef[21 + i] = (float)std::exp(dis * dis * invsigma2); }
  }
  const size_t B = (size_t)max_batch;
#define TRYHIP(x) do { if ((x) != hipSuccess) { set_error("plh_line_create: %s failed (batch %d)", #x, max_batch); plh_line_destroy(h); return PLH_ERR_ALLOC; } } while (0)
  TRYHIP(hipMalloc((void**)&h->dTmpA, B * a.fullStride));
  TRYHIP(hipMalloc((void**)&h->dScaled, B * a.scaledStride));
  // per-frame block, hot parts first: segments | region queue / log | gradient norms of the log (+ reduce_region_radius scratch) |
  // level-line records | seed list | ordering counters | LSD_REFINE_ADV work lists
  const long long offSegs = 0, offReg = align_up<long long>((long long)a.segCap * 4, 64), offRegq = offReg + a.scaledStride,
                  offPix = offRegq + a.scaledStride, offOrd = offPix + a.scaledStride, offWork = offOrd + a.scaledStride,
                  offPark = offWork + align_up<long long>((long long)lsd_order_work_u32(), 64),
                  blockWords = offPark + align_up<long long>(2LL * a.segCap + 2, 64);
  // one contiguous block per frame, 2 MiB aligned (64 KiB for small frames): a wavefront's scattered accesses stay inside one or
  // two translation fragments.  (Measured again in round 4: with 256 KiB alignment the seed scatter k_lsd_bin_scatter took 2.82
  // instead of 2.05 ms per 1536 frames and k_lsd_grow 69.4 instead of 68.0, profiles/r04_screen_ab_v4_logq_256k_alignment.txt.)
  a.arenaStride = align_up<long long>(blockWords, blockWords >= (1 << 18) ? (1 << 19) : (1 << 14));   // 2 MiB (64 KiB for small frames)
  TRYHIP(hipMalloc((void**)&h->dArena, B * (size_t)a.arenaStride * 4));
  TRYHIP(hipMalloc((void**)&h->dDxdy, B * a.fullStride * 4));
  TRYHIP(hipMalloc((void**)&h->dQmax, B * 4));
  TRYHIP(hipMalloc((void**)&h->dNOrdered, B * 4));
  TRYHIP(hipMalloc((void**)&h->dNSegs, B * 4));
  TRYHIP(hipMalloc((void**)&h->dStatus, 64));
  TRYHIP(hipMemset(h->dStatus, 0, 64));
  TRYHIP(hipMalloc((void**)&h->dXtab, xt.size() * sizeof(ResizeTap)));
  TRYHIP(hipMalloc((void**)&h->dYtab, yt.size() * sizeof(ResizeTap)));
  TRYHIP(hipMemcpy(h->dXtab, xt.data(), xt.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
  TRYHIP(hipMemcpy(h->dYtab, yt.data(), yt.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
  TRYHIP(hipMalloc((void**)&h->dCoef, coef.size() * 4));
  TRYHIP(hipMemcpy(h->dCoef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
  TRYHIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
#undef TRYHIP
  if (lsd_grow_request_lds() != PLH_OK) {   // per kernel and device (hipSetDevice(device) above)
    plh_line_destroy(h);
    return PLH_ERR_INVALID;
  }
  h->a.angleTab = angle_table_acquire(device);
  if (!h->a.angleTab) {
    set_error("plh_line_create: the ll_angle table could not be built on device %d", device);
    plh_line_destroy(h);
    return PLH_ERR_ALLOC;
  }
  a.angleTab = h->a.angleTab;
  if (p->num_octaves == 2) {   // the second octave: a one-octave plan of its own for the half-size image
    plh_line_params p1 = *p;
    p1.num_octaves = 1;
    const plh_status st1 = plh_line_create(&p1, device, rows / 2, cols / 2, max_batch, &h->oct1);
    if (st1 != PLH_OK) { plh_line_destroy(h); return st1; }
    if (hipMalloc((void**)&h->dOctImg, B * (size_t)h->oct1->a.fullStride) != hipSuccess) {
      (void)hipGetLastError();
      set_error("plh_line_create: cannot allocate the second octave's images");
      plh_line_destroy(h);
      return PLH_ERR_ALLOC;
    }
    const LineDeviceArgs& a1 = h->oct1->a;
    a.segs1 = a1.segs; a.nSegs1 = a1.nSegs; a.arena1Stride = a1.arenaStride; a.segCap1 = a1.segCap; a.w1 = a1.w; a.h1 = a1.h;
    a.octScale1 = 2.0f;   // pow((float)(int)scale, 1)
    a.dxdy1 = a1.dxdy; a.full1Stride = a1.fullStride;
  }
  a.tmpA = h->dTmpA; a.scaled = h->dScaled; a.pix = h->dArena + offPix; a.ordered = h->dArena + offOrd; a.reg = h->dArena + offReg; a.regq = h->dArena + offRegq; a.orderWork = h->dArena + offWork; a.park = h->dArena + offPark;
  a.qmax = h->dQmax; a.nOrdered = h->dNOrdered; a.segs = reinterpret_cast<float*>(h->dArena + offSegs); a.nSegs = h->dNSegs; a.dxdy = h->dDxdy;
  a.xtab = h->dXtab; a.ytab = h->dYtab; a.status = h->dStatus;
  *out = h;
  return PLH_OK;
}

int plh_line_capacity(const plh_line* h) { return h ? h->a.outCap : 0; }


plh_status plh_line_set_undistort(plh_line* h, const float K[4], const float D[5]) {
  if (!h || !K) return PLH_ERR_INVALID;
  PLH_HIP(hipSetDevice(h->device));
  bool any = false;
  if (D)
    for (int i = 0; i < 5; i++) any |= D[i] != 0.f;
  if (!any) { h->hasUndistort = false; return PLH_OK; }
  const int w = h->cols, hh = h->rows;
  if (w < 2 || hh < 2) { set_error("plh_line_set_undistort: image smaller than 2 x 2"); return PLH_ERR_INVALID; }
  std::vector<RemapTap> map((size_t)w * hh);
  // cv::initUndistortRectifyMap(K, D, I, K, size, CV_32F) (Frame.cc:221), evaluated per pixel in double (pinned), then
  // cv::remap's fixed-point split of the float coordinates (1/32 pixel, cvRound) and its constant-0 border, folded into
  // one RemapTap per pixel (line_plan.h)
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
  auto axis = [](float m, int n, int& i0, unsigned& w0, unsigned& w1) {
    // the taps of this axis are at i and i + 1 with weights 32 - f and f; outside [0, n) a tap reads 0
    // cvRound saturates: beyond the int range the taps are far outside either way; a NaN converts to 0
    long s = std::isnan(m) ? 0L : (std::fabs(m) < 6.0e7f ? std::lrintf(m * 32.f) : (m > 0 ? (long)n * 64 : -64L));
    const long i = s >> 5;
    const unsigned f = (unsigned)(s & 31);
    i0 = (int)std::min<long>(std::max<long>(i, 0), n - 2);
    auto wt = [&](long pos) -> unsigned { return pos == i ? 32u - f : (pos == i + 1 ? f : 0u); };
    w0 = wt(i0);
    w1 = wt(i0 + 1);
  };
  for (int v = 0; v < hh; v++)
    for (int u = 0; u < w; u++) {
      const double x = (u - cx) / fx, y = (v - cy) / fy;
      const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
      const double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
      const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
      const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
      int x0, y0;
      unsigned wx0, wx1, wy0, wy1;
      axis((float)(fx * xd + cx), w, x0, wx0, wx1);
      axis((float)(fy * yd + cy), hh, y0, wy0, wy1);
      RemapTap t;
      t.off = (uint32_t)y0 * (uint32_t)w + (uint32_t)x0;
      t.wts = wx0 | (wx1 << 8) | (wy0 << 16) | (wy1 << 24);
      map[(size_t)v * w + u] = t;
    }
  if (!h->dMap) PLH_HIP(hipMalloc((void**)&h->dMap, map.size() * sizeof(RemapTap)));
  if (!h->dUndist) PLH_HIP(hipMalloc((void**)&h->dUndist, (size_t)h->maxBatch * h->a.fullStride));
  PLH_HIP(hipMemcpy(h->dMap, map.data(), map.size() * sizeof(RemapTap), hipMemcpyHostToDevice));
  h->hasUndistort = true;
  return PLH_OK;
}

// Wavefronts per frame for k_lsd_grow's launch: small batches leave most of the 1024 SIMDs idle with one wavefront per
// frame, so a frame gets several (k_lsd_grow_mw: optimistic transactions, in-order commit -- same segments); large batches
// fill the GPU with frames and keep one wavefront each.  plh_line_set_grow_waves overrides (0 / 1 = never, n = always n).
static int mw_waves_for(const plh_line* h, int batch) {
  int forced = h->growWaves;
#if defined(HIPEMU) || defined(PLH_GROW_PROF)
  // the CPU emulator build (tests/hipemu: a frame with eight fiber wavefronts costs four times the wall clock of two) and the
  // counter build of tools/ take the count from the environment; the product build has no such knob
  if (forced < 0) {
    const char* e = getenv("PLH_GROW_MW_WAVES");
    if (e) forced = atoi(e);
  }
#endif
  if (forced >= 0) return forced == 1 ? 0 : std::min(forced, 16);
  // measured on MI355X, 640 x 480 (tools/mw_sweep.py, profiles/r03_mw_sweep.txt), region growing per launch: 1 frame 44.6 ms with
  // one wavefront, 13.5 with 8, 12.9 with 16; 512 frames 58.0 -> 26.8 with 8 (16: 39.7); 1024 frames 64.5 -> 49.4 with 8 (4: 65.7);
  // from 2048 frames on one wavefront per frame is as fast
  // (16 wavefronts: 12.9 ms of region growing for a lone frame, but 14.2 against 13.9 ms in the tracker's call, where the ORB
  // extractor's kernels run beside it: profiles/r03_bench_full.json history)
  if (batch <= 1024) return 8;
  return 0;
}

static plh_status mw_reserve(plh_line* h, LineDeviceArgs& a, int batch, int waves) {
  a.mwWaves = waves;
  a.mwRegStride = 5 * a.scaledStride;   // posted logs | three regions of a running transaction | reduce_region_radius scratch
  a.mwMarkStride = a.scaledStride;
  a.mwLag = h->mwLag;   // (at most 448: below the ring of posted transactions, 512, by more than the wavefronts' own)
  a.mwDrainGap = h->mwDrainGap;
  const long long slots = (long long)batch * (waves + 1);
  if (slots > h->mwWaveSlots) {
    if (h->doneValid) PLH_HIP(hipEventSynchronize(h->doneEv));   // earlier launches may still use the old buffers
    if (h->dMwReg) (void)hipFree(h->dMwReg);
    if (h->dMwMark) (void)hipFree(h->dMwMark);
    h->dMwReg = nullptr; h->dMwMark = nullptr; h->mwWaveSlots = 0;
    if (hipMalloc((void**)&h->dMwReg, (size_t)slots * a.mwRegStride * 4) != hipSuccess ||
        hipMalloc((void**)&h->dMwMark, (size_t)slots * a.mwMarkStride) != hipSuccess ||
        hipMemset(h->dMwMark, 0, (size_t)slots * a.mwMarkStride) != hipSuccess) {   // planes are zero between transactions
      (void)hipGetLastError();
      set_error("plh_line_extract: cannot allocate the multi-wavefront region-growing workspace (%lld wavefront slots)", slots);
      return PLH_ERR_ALLOC;
    }
    h->mwWaveSlots = slots;
  }
  if (batch > h->mwHintFrames) {
    if (h->doneValid) PLH_HIP(hipEventSynchronize(h->doneEv));
    if (h->dMwHint) (void)hipFree(h->dMwHint);
    h->dMwHint = nullptr; h->mwHintFrames = 0;
    if (hipMalloc((void**)&h->dMwHint, (size_t)batch * a.mwMarkStride * 2) != hipSuccess) {
      (void)hipGetLastError();
      set_error("plh_line_extract: cannot allocate the claim-hint planes (%d frames)", batch);
      return PLH_ERR_ALLOC;
    }
    h->mwHintFrames = batch;
  }
  a.mwReg = h->dMwReg; a.mwMark = h->dMwMark; a.mwHint = h->dMwHint;
  return PLH_OK;
}

// A scheduler's hooks around the region-growing launch of the next plh_line_extract_batch_dev calls (events owned by the caller,
// NULL = none): the launch waits for `wait_before`, `record_after` is recorded behind it.  plh_frontend uses them for small resident
// batches, where the multi-wavefront kernel fills every SIMD's register file and nothing can run beside it: the ORB chain is
// placed around region growing instead of underneath it.
plh_status plh_line_set_grow_events(plh_line* h, void* wait_before, void* record_after) {
  if (!h) return PLH_ERR_INVALID;
  h->growGate = (hipEvent_t)wait_before;
  h->growDone = (hipEvent_t)record_after;
  return PLH_OK;
}

// What a launch sequence needs besides the plan: the multi-wavefront workspace for this batch size, the LSD_REFINE_ADV buffers
// (the log_gamma table is filled on `s`, in front of the kernels that read it).
static plh_status line_prepare(plh_line* h, LineDeviceArgs& a, int batch, hipStream_t s) {
  const int waves = mw_waves_for(h, batch);
  if (waves > 0) {
    const plh_status st = mw_reserve(h, a, batch, waves);
    if (st != PLH_OK) return st;
  }
  if (a.refineAdv) {
    const size_t recBytes = align_up<size_t>((size_t)h->maxBatch * a.segCap * lsd_adv_rec_bytes(), 256);
    if (!h->dAdv && hipMalloc(&h->dAdv, recBytes + (size_t)h->maxBatch * a.scaledStride * 4) != hipSuccess) {
      (void)hipGetLastError();
      set_error("plh_line_extract: cannot allocate the LSD_REFINE_ADV rectangle records (%d frames x %d)", h->maxBatch, a.segCap);
      return PLH_ERR_ALLOC;
    }
    if (!h->dLgamma) {
      const int n = a.sw * a.sh + 2;   // a rectangle counts at most every pixel of the scaled image: arguments 1 .. sw sh + 1
      if (hipMalloc((void**)&h->dLgamma, (size_t)n * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        set_error("plh_line_extract: cannot allocate the log_gamma table (%d entries)", n);
        return PLH_ERR_ALLOC;
      }
      // filled ONCE and synchronously (a later extract on another stream must not find a table that is still being written, and a
      // failed fill must not leave a pointer behind that every later call would take for a filled table -- ADVICE r5)
      launch_lsd_lgamma_table(h->dLgamma, n, s);
      hipError_t le = hipGetLastError();
      if (le == hipSuccess) le = hipStreamSynchronize(s);
      if (le != hipSuccess) {
        (void)hipFree(h->dLgamma);
        h->dLgamma = nullptr;
        set_error("plh_line_extract: filling the log_gamma table failed: %s", hipGetErrorString(le));
        return PLH_ERR_HIP;
      }
    }
    a.adv = reinterpret_cast<LsdAdvRec*>(h->dAdv);
    a.advAng = reinterpret_cast<float*>(static_cast<uint8_t*>(h->dAdv) + recBytes);
    a.lgamma = h->dLgamma;
  } else {
    a.adv = nullptr; a.advAng = nullptr; a.lgamma = nullptr;
  }
  return PLH_OK;
}

// An abandoned multi-wavefront launch (status bit 4) leaves private marks behind; the planes must be zero between transactions.
// Both octaves: the second one shares the parent's status word but has mark planes of its own.
static plh_status line_reset_mw_marks(plh_line* h) {
  PLH_HIP(hipDeviceSynchronize());   // nothing of this device may still be running on the planes (error path: cost is no concern)
  for (plh_line* q = h; q; q = q->oct1)
    if (q->dMwMark) PLH_HIP(hipMemset(q->dMwMark, 0, (size_t)q->mwWaveSlots * q->a.scaledStride));
  return PLH_OK;
}

// cv::LineSegmentDetector::detect on one octave's images: 7x7 sigma 0.75 blur -> 0.8x resize -> level-line field -> seed order ->
// region growing -> the kept regions' rectangles (and LSD_REFINE_ADV's rect_improve).  Leaves the segments in the plan's arena.
// `top`: the caller's octave (profiling marks and the scheduler's events around region growing belong to it).
static plh_status line_lsd_stages(plh_line* h, const LineDeviceArgs& a, const uint8_t* src, long long srcStride, int batch, hipStream_t s,
                                  bool top) {
  if (top) line_prof_mark(h, 0, s);
  launch_blur7(src, srcStride, a.w, h->dTmpA, a.fullStride, a.w, a.w, a.h, batch, h->taps075, s);
  PLH_LAUNCH_CHECK();
  launch_resize(h->dTmpA, a.fullStride, a.w, a.w, a.h, h->dScaled, a.scaledStride, a.spitch, a.sw, a.sh, batch, h->dXtab, h->dYtab,
                h->rszTP, h->rszTR, s);
  PLH_LAUNCH_CHECK();
  PLH_HIP(hipMemsetAsync(h->dQmax, 0, (size_t)batch * 4, s));
  launch_lsd_grad(a, s);
  PLH_LAUNCH_CHECK();
  launch_lsd_order(a, s);
  PLH_LAUNCH_CHECK();
  if (top) { line_prof_mark(h, 0, s); line_prof_mark(h, 1, s); }
  if (top && h->growGate) PLH_HIP(hipStreamWaitEvent(s, h->growGate, 0));
#if defined(PLH_GROW_PROF) && !defined(HIPEMU)
  if (getenv("PLH_PROF_LOG_ALLOC")) {   // counter build only: where the buffers of this launch lie (to place a GPU fault address)
    const struct { const char* name; const void* p; size_t n; } bufs[] = {
        {"arena", h->dArena, (size_t)h->maxBatch * (size_t)a.arenaStride * 4}, {"mwReg", h->dMwReg, (size_t)h->mwWaveSlots * (size_t)a.mwRegStride * 4},
        {"mwMark", h->dMwMark, (size_t)h->mwWaveSlots * (size_t)a.mwMarkStride}, {"mwHint", h->dMwHint, (size_t)h->mwHintFrames * (size_t)a.mwMarkStride * 2},
        {"scaled", h->dScaled, (size_t)h->maxBatch * (size_t)a.scaledStride}, {"tmpA", h->dTmpA, (size_t)h->maxBatch * (size_t)a.fullStride},
        {"dxdy", h->dDxdy, (size_t)h->maxBatch * (size_t)a.fullStride * 4}, {"angleTab", a.angleTab, ((size_t)LSD_ANGLE_ROWS << LSD_ANGLE_PITCH_LOG2) * sizeof(LsdAngleEntry)},
        {"adv", h->dAdv, 0}, {"lgamma", h->dLgamma, 0}, {"status", h->dStatus, 64}};
    for (const auto& b : bufs) fprintf(stderr, "PLHBUF %-8s %p .. %p (%zu bytes)\n", b.name, b.p, (const void*)((const char*)b.p + b.n), b.n);
    fprintf(stderr, "PLHBUF batch %d waves %d arenaStride %lld mwRegStride %lld mwMarkStride %lld\n", batch, a.mwWaves, a.arenaStride, a.mwRegStride, a.mwMarkStride);
  }
#endif
  launch_lsd_grow(a, s);
  PLH_LAUNCH_CHECK();
  if (top && h->growDone) PLH_HIP(hipEventRecord(h->growDone, s));
  if (top) { line_prof_mark(h, 1, s); line_prof_mark(h, 2, s); }   // stage 2 = rectangles (+ a second octave's LSD) + KeyLine selection
  launch_lsd_rects(a, s);   // one lane per region
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_line_extract_batch_dev(plh_line* h, const uint8_t* d_imgs, int batch, size_t frame_stride, const uint8_t* d_mask,
                                      plh_keyline* d_keylines, uint8_t* d_desc, double* d_linefn, int32_t* d_n, void* stream) {
  if (!h || !d_imgs || !d_keylines || !d_desc || !d_linefn || !d_n || batch <= 0 || batch > h->maxBatch ||
      frame_stride < (size_t)h->rows * h->cols) {
    set_error("plh_line_extract_batch_dev: invalid argument (batch %d, plan max %d)", batch, h ? h->maxBatch : 0);
    return PLH_ERR_INVALID;
  }
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  hipStream_t s = (hipStream_t)stream;
  LineDeviceArgs a = h->a;
  a.batch = batch;
  a.img = d_imgs; a.imgStride = (long long)frame_stride;
  a.mask = d_mask;
  const uint8_t* src = d_imgs;
  long long srcStride = (long long)frame_stride;
  plh_status st = line_prepare(h, a, batch, s);
  if (st != PLH_OK) return st;
  PLH_HIP(hipMemsetAsync(h->dStatus, 0, 4, s));   // capacity flags of this call only (plh_line_status)
  if (h->hasUndistort) {
    a.remap = h->dMap; a.undist = h->dUndist;
    launch_remap(a, s);
    PLH_LAUNCH_CHECK();
    src = h->dUndist; srcStride = a.fullStride;
  }
  st = line_lsd_stages(h, a, src, srcStride, batch, s, true);
  if (st != PLH_OK) return st;
  plh_line* h1 = h->oct1;
  LineDeviceArgs a1;
  if (h1) {   // second octave: pyrDown of the frame, the same LSD stages on its own plan (same refine level, wavefronts, flags word)
    a1 = h1->a;
    a1.batch = batch;
    a1.refineAdv = a.refineAdv; a1.screen = a.screen;
    a1.status = h->dStatus;
    h1->growWaves = h->growWaves; h1->mwLag = h->mwLag; h1->mwDrainGap = h->mwDrainGap;
    st = line_prepare(h1, a1, batch, s);
    if (st != PLH_OK) return st;
    launch_pyr_down5(src, srcStride, a.w, a.h, h->dOctImg, a1.fullStride, a1.w, a1.h, batch, s);
    PLH_LAUNCH_CHECK();
    st = line_lsd_stages(h1, a1, h->dOctImg, a1.fullStride, batch, s, false);
    if (st != PLH_OK) return st;
  }
  launch_keylines(a, d_keylines, d_linefn, d_n, s);
  PLH_LAUNCH_CHECK();
  line_prof_mark(h, 2, s);
  line_prof_mark(h, 3, s);
  // LBD: 5x5 sigma 1 blur -> Sobel -> band descriptor (per further octave: pyrDown of the blurred image, Sobel)
  launch_blur7(src, srcStride, a.w, h->dTmpA, a.fullStride, a.w, a.w, a.h, batch, h->taps1, s);
  PLH_LAUNCH_CHECK();
  launch_sobel(a, s);
  PLH_LAUNCH_CHECK();
  if (h1) {
    launch_pyr_down5(h->dTmpA, a.fullStride, a.w, a.h, h1->dTmpA, a1.fullStride, a1.w, a1.h, batch, s);
    PLH_LAUNCH_CHECK();
    launch_sobel(a1, s);
    PLH_LAUNCH_CHECK();
  }
  launch_lbd(a, d_keylines, d_n, h->dCoef, d_desc, s);
  PLH_LAUNCH_CHECK();
  line_prof_mark(h, 3, s);
  if (!h->doneEv) PLH_HIP(hipEventCreateWithFlags(&h->doneEv, hipEventDisableTiming));
  PLH_HIP(hipEventRecord(h->doneEv, s));
  h->doneValid = true;
  if (h1) { h1->doneEv = h->doneEv; h1->doneValid = true; }   // (a workspace re-allocation of the second octave waits on it too)
  return PLH_OK;
}

plh_status plh_line_extract(plh_line* h, const uint8_t* img, int rows, int cols, size_t step, const uint8_t* mask,
                            plh_keyline* keylines, uint8_t* desc, double* linefn, int cap, int* n_out) {
  if (!h || !n_out) return PLH_ERR_INVALID;
  if (rows == 0 || cols == 0 || !img) {   // reference: empty image -> silent return (LineExtractor.cpp:29-30)
    *n_out = 0;
    return PLH_OK;
  }
  if (rows != h->rows || cols != h->cols || step < (size_t)cols || !keylines || !desc || !linefn) {
    set_error("plh_line_extract: image %dx%d does not match the plan %dx%d", rows, cols, h->rows, h->cols);
    return PLH_ERR_INVALID;
  }
  PLH_HIP(hipSetDevice(h->device));
  const size_t B = (size_t)h->maxBatch, ocap = h->a.outCap;
  if (!h->dImgs) {
    PLH_HIP(hipMalloc((void**)&h->dImgs, B * rows * cols));
    PLH_HIP(hipMalloc((void**)&h->dKl, B * ocap * sizeof(plh_keyline)));
    PLH_HIP(hipMalloc((void**)&h->dDesc, B * ocap * 32));
    PLH_HIP(hipMalloc((void**)&h->dFn, B * ocap * 24));
    PLH_HIP(hipMalloc((void**)&h->dN, B * 4));
  }
  PLH_HIP(hipMemcpy2DAsync(h->dImgs, cols, img, step, cols, rows, hipMemcpyHostToDevice, h->stream));
  const uint8_t* dmask = nullptr;
  if (mask) {
    if (!h->dMask) PLH_HIP(hipMalloc((void**)&h->dMask, (size_t)rows * cols));
    PLH_HIP(hipMemcpyAsync(h->dMask, mask, (size_t)rows * cols, hipMemcpyHostToDevice, h->stream));
    dmask = h->dMask;
  }
  plh_status st = plh_line_extract_batch_dev(h, h->dImgs, 1, (size_t)rows * cols, dmask, h->dKl, h->dDesc, h->dFn, h->dN, h->stream);
  if (st != PLH_OK) return st;
  int n = 0;
  PLH_HIP(hipMemcpyAsync(&n, h->dN, 4, hipMemcpyDeviceToHost, h->stream));
  PLH_HIP(hipStreamSynchronize(h->stream));
  if (n > cap) {
    set_error("plh_line_extract: %d keylines exceed the caller's capacity %d", n, cap);
    return PLH_ERR_CAPACITY;
  }
  PLH_HIP(hipMemcpy(keylines, h->dKl, (size_t)n * sizeof(plh_keyline), hipMemcpyDeviceToHost));
  PLH_HIP(hipMemcpy(desc, h->dDesc, (size_t)n * 32, hipMemcpyDeviceToHost));
  PLH_HIP(hipMemcpy(linefn, h->dFn, (size_t)n * 24, hipMemcpyDeviceToHost));
  *n_out = n;
  int stf = 0;
  PLH_HIP(hipMemcpy(&stf, h->dStatus, 4, hipMemcpyDeviceToHost));
  if (stf) {
    if (stf & 16) (void)line_reset_mw_marks(h);
    set_error("line kernels reported a capacity overflow or an abandoned launch (flags 0x%x)", stf);
    return PLH_ERR_CAPACITY;
  }
  return PLH_OK;
}

plh_status plh_line_status(plh_line* h, int* flags) {
  if (!h || !flags) return PLH_ERR_INVALID;
  PLH_HIP(hipSetDevice(h->device));
  if (h->doneValid) PLH_HIP(hipEventSynchronize(h->doneEv));
  PLH_HIP(hipMemcpy(flags, h->dStatus, 4, hipMemcpyDeviceToHost));
  if (*flags & 16) return line_reset_mw_marks(h);   // an abandoned launch: the workspace of both octaves is re-initialised
  return PLH_OK;
}

// Allocate now what plh_line_extract_batch_dev would allocate on its first call with `batch` frames at the current settings (refine
// level, wavefronts per frame): the multi-wavefront workspace is tens of GB for a thousand KITTI frames, and a host that builds its
// pipeline up front wants an out-of-memory condition at create time, not at the first step.
plh_status plh_line_reserve(plh_line* h, int batch) {
  if (!h || batch <= 0 || batch > h->maxBatch) {
    set_error("plh_line_reserve: invalid argument (batch %d, plan max %d)", batch, h ? h->maxBatch : 0);
    return PLH_ERR_INVALID;
  }
  PLH_HIP(hipSetDevice(h->device));
  for (plh_line* q = h; q; q = q->oct1) {
    if (q != h) { q->growWaves = h->growWaves; q->a.refineAdv = h->a.refineAdv; }
    LineDeviceArgs a = q->a;
    const plh_status st = line_prepare(q, a, batch, h->stream);
    if (st != PLH_OK) return st;
  }
  PLH_HIP(hipStreamSynchronize(h->stream));   // (the log_gamma table, if it was filled just now)
  return PLH_OK;
}

plh_status plh_line_set_refine(plh_line* h, int level) {
  if (!h || (level != PLH_LSD_REFINE_STD && level != PLH_LSD_REFINE_ADV)) {
    set_error("plh_line_set_refine: level must be PLH_LSD_REFINE_STD (0) or PLH_LSD_REFINE_ADV (1)");
    return PLH_ERR_INVALID;
  }
  h->a.refineAdv = level == PLH_LSD_REFINE_ADV ? 1 : 0;
  return PLH_OK;
}

int plh_lsd_refine_default(void) { return PLH_LSD_REFINE_DEFAULT; }

plh_status plh_line_set_screen(plh_line* h, int on) {
  if (!h) return PLH_ERR_INVALID;
  h->a.screen = on ? 1 : 0;
  return PLH_OK;
}

plh_status plh_line_set_grow_tuning(plh_line* h, int run_ahead, int drain_gap) {
  if (!h) return PLH_ERR_INVALID;
  h->mwLag = run_ahead <= 0 ? 448 : std::min(run_ahead, 448);
  h->mwDrainGap = drain_gap <= 0 ? 8 : drain_gap;
  return PLH_OK;
}

plh_status plh_line_set_grow_waves(plh_line* h, int waves) {
  if (!h || waves < -1 || waves > 16) {
    set_error("plh_line_set_grow_waves: waves must be -1 (automatic), 0 or 1 (one wavefront per frame) or 2..16");
    return PLH_ERR_INVALID;
  }
  h->growWaves = waves;
  return PLH_OK;
}

plh_status plh_line_set_profiling(plh_line* h, int on) {
  if (!h) return PLH_ERR_INVALID;
  h->profiling = on != 0;
  h->evUsed = 0;
  for (int k = 0; k < 4; k++) { h->kernelMs[k] = 0; h->kernelLaunches[k] = 0; }
  return PLH_OK;
}

plh_status plh_line_kernel_ms(plh_line* h, int stage, double* total_ms, int* intervals) {
  if (!h || stage < 0 || stage > 3 || !total_ms || !intervals) return PLH_ERR_INVALID;
  for (size_t i = 0; i + 1 < h->evUsed; i += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->evPool[i], h->evPool[i + 1]) == hipSuccess) {
      h->kernelMs[h->evKind[i]] += ms;
      h->kernelLaunches[h->evKind[i]]++;
    }
  }
  h->evUsed = 0;
  *total_ms = h->kernelMs[stage];
  *intervals = h->kernelLaunches[stage];
  return PLH_OK;
}

plh_status plh_line_read_segments(plh_line* h, int b, float* out_xyxy, int cap, int* n_out) {
  if (!h || b < 0 || b >= h->maxBatch || !n_out) return PLH_ERR_INVALID;
  PLH_HIP(hipSetDevice(h->device));
  PLH_HIP(hipDeviceSynchronize());
  int n = 0;
  PLH_HIP(hipMemcpy(&n, h->dNSegs + b, 4, hipMemcpyDeviceToHost));
  *n_out = n;
  const int m = std::min(n, std::min(cap, h->a.segCap));
  if (out_xyxy && m > 0)
    PLH_HIP(hipMemcpy(out_xyxy, h->a.segs + (size_t)b * h->a.arenaStride, (size_t)m * 16, hipMemcpyDeviceToHost));
  return PLH_OK;
}

}  // extern "C"
