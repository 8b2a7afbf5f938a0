// Host side of the ORB extractor: plan construction (geometry identical to the reference's
// constructor and per-level setup), workspace allocation sized for the whole batch, launch
// sequence, and the C ABI entry points declared in include/plslam_hip.h.
#include <cmath>
#include <new>
#include <vector>

#include "orb_plan.h"
#include "plh_common.h"

namespace plh {

// launchers implemented in orb_kernels.hip
void launch_pyr_down(const OrbDeviceArgs& a, int l, int pitch, int h, size_t lds, const PyrLaunch* fast, hipStream_t s);
void launch_fast_strips(const OrbDeviceArgs& a, size_t lds, hipStream_t s);
size_t fast_strip_lds_bytes(int width, int ch);
constexpr int FAST_STRIP_MAX_W = 200;   // measured on MI355X (1024 frames): 100 -> 3.13 ms, 140 -> 2.82, 200 -> 2.64, 270 -> 3.15, 330 -> 3.24, 660 -> 3.44
size_t octree_lds_bytes(int nodeCap);
void launch_octree(const OrbDeviceArgs& a, int nodeCapMax, hipStream_t s);
void launch_orient_brief(const OrbDeviceArgs& a, plh_keypoint* kps, uint8_t* desc, int* nOut, int cap, hipStream_t s);

static thread_local char g_err[512];
char* tls_error() { return g_err; }
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

plh_status ensure_runtime() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device visible (the product path has no CPU fallback)");
    return PLH_ERR_NO_DEVICE;
  }
  (void)hipGetLastError();
  return PLH_OK;
}

static inline int cv_round_host(float v) { return (int)lrintf(v); }
static inline int cv_floor_host(float v) { int i = (int)v; return i - (i > v); }

}  // namespace plh

using namespace plh;

struct plh_orb {
  plh_orb_params p;
  int device, rows, cols, maxBatch;
  int nlevels;
  double scaleFactorD;               // the reference stores scaleFactor as double (ORBextractor.h:95)
  std::vector<float> sf, isf, sig2, isig2;
  std::vector<int> perLevel;
  std::vector<OrbLevel> levels;
  std::vector<OrbCell> cells;
  std::vector<OrbStrip> strips;
  size_t fastLds = 0;
  std::vector<ResizeTap> xtab, ytab;
  int nodeCapMax = 0, selPerFrame = 0;
  long long pyrFrameBytes = 0, slotsPerFrame = 0;
  // device
  OrbLevel* dLevels = nullptr;
  OrbCell* dCells = nullptr;
  OrbStrip* dStrips = nullptr;
  ResizeTap *dXtab = nullptr, *dYtab = nullptr;
  uint8_t* dPyr = nullptr;
  uint32_t *dSlots = nullptr, *dCellCount = nullptr, *dKeys = nullptr, *dSel = nullptr;
  int *dSelCount = nullptr, *dStatus = nullptr;
  hipEvent_t doneEv = nullptr;   // recorded behind the last kernel of every extract call: plh_orb_status waits on it (the
                                 // caller's stream may be gone by then; the event is the handle's own)
  bool doneValid = false;
  // staging for the host-buffer entry points
  uint8_t* dImgs = nullptr;
  plh_keypoint* dKps = nullptr;
  uint8_t* dDesc = nullptr;
  int* dN = nullptr;
  hipStream_t stream = nullptr;
  // optional per-kernel timing with HIP events on the caller's stream (bench.py roofline leg)
  bool profiling = false;
  std::vector<hipEvent_t> evPool;
  std::vector<int> evKind;           // kernel id of the interval [2i, 2i+1]
  size_t evUsed = 0;
  double kernelMs[4] = {0, 0, 0, 0};
  int kernelLaunches[4] = {0, 0, 0, 0};
  // last call (for the taps)
  const uint8_t* lastImgs = nullptr;
  long long lastStride = 0;
  int lastBatch = 0;
};

namespace {

// ORBextractor::ORBextractor, reference src/ORBextractor.cc:410-446 (scale tables + feature split)
void build_tables(plh_orb* h) {
  const int nl = h->nlevels;
  h->sf.resize(nl); h->isf.resize(nl); h->sig2.resize(nl); h->isig2.resize(nl); h->perLevel.resize(nl);
  h->sf[0] = 1.0f; h->sig2[0] = 1.0f;
  for (int i = 1; i < nl; i++) {
    h->sf[i] = (float)(h->sf[i - 1] * h->scaleFactorD);
    h->sig2[i] = h->sf[i] * h->sf[i];
  }
  for (int i = 0; i < nl; i++) { h->isf[i] = 1.0f / h->sf[i]; h->isig2[i] = 1.0f / h->sig2[i]; }
  float factor = (float)(1.0f / h->scaleFactorD);
  float nDesired = h->p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) {
    h->perLevel[l] = cv_round_host(nDesired);
    sum += h->perLevel[l];
    nDesired *= factor;
  }
  h->perLevel[nl - 1] = std::max(h->p.nfeatures - sum, 0);
}

// cv::resize(INTER_LINEAR) coefficient tables for one axis (11-bit fixed point).
void resize_axis(int ssize, int dsize, bool isX, std::vector<ResizeTap>& out, int* xmaxOut) {
  const double scale = 1.0 / ((double)dsize / ssize);
  int xmax = dsize;
  for (int d = 0; d < dsize; d++) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = cv_floor_host(f);
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
    t.a0 = (short)cv_round_host((1.f - f) * 2048.f);
    t.a1 = (short)cv_round_host(f * 2048.f);
    t.pad = 0;
    if (isX && d >= xmax) { t.a0 = 2048; t.a1 = 0; }   // single-tap tail: S[xofs]*ONE
    out.push_back(t);
  }
  if (xmaxOut) *xmaxOut = xmax;
}

plh_status build_plan(plh_orb* h) {
  const int nl = h->nlevels;
  h->levels.assign(nl, OrbLevel());
  long long off = 0;
  int slotOff = 0, selOff = 0;
  for (int l = 0; l < nl; l++) {
    OrbLevel& L = h->levels[l];
    // ORBextractor::ComputePyramid, ORBextractor.cc:1111-1112
    L.w = cv_round_host((float)h->cols * h->isf[l]);
    L.h = cv_round_host((float)h->rows * h->isf[l]);
    if (L.w < 2 * ORB_EDGE_THRESHOLD || L.h < 2 * ORB_EDGE_THRESHOLD || L.w >= (1 << ORB_KEY_XY_BITS) || L.h >= (1 << ORB_KEY_XY_BITS)) {
      set_error("level %d size %dx%d unsupported (needs 38 <= side < 4096)", l, L.w, L.h);
      return PLH_ERR_INVALID;
    }
    if (l == 0) {
      L.pitch = h->cols;
      L.off = 0;
    } else {
      L.pitch = align_up(L.w, 64);
      L.off = off;
      off += (long long)L.pitch * L.h;
      off = align_up<long long>(off, 256);
      L.xtabOff = (int)h->xtab.size();
      resize_axis(h->levels[l - 1].w, L.w, true, h->xtab, &L.xmax);
      L.ytabOff = (int)h->ytab.size();
      resize_axis(h->levels[l - 1].h, L.h, false, h->ytab, nullptr);
      // exact extent of the source tile behind any 256 x 16 output block (k_pyr_down stages it in LDS)
      const int sw = h->levels[l - 1].w, sh = h->levels[l - 1].h;
      int maxW = 0, maxH = 0;
      for (int x0 = 0; x0 < L.w; x0 += 256) {
        const int lo = h->xtab[L.xtabOff + x0].ofs & ~3;
        const int hi = std::min((int)h->xtab[L.xtabOff + std::min(x0 + 255, L.w - 1)].ofs + 1, sw - 1);
        maxW = std::max(maxW, hi - lo + 1);
      }
      for (int y0 = 0; y0 < L.h; y0 += 16) {
        const int lo = std::min(std::max((int)h->ytab[L.ytabOff + y0].ofs, 0), sh - 1);
        const int hi = std::min(std::max((int)h->ytab[L.ytabOff + std::min(y0 + 15, L.h - 1)].ofs + 1, 0), sh - 1);
        maxH = std::max(maxH, hi - lo + 1);
      }
      // the fast kernel wants the 8 taps of every 4-pixel group inside one 8-byte window (true up to scale 2) and reads three
      // aligned dwords from the window's dword on: 12 bytes of slack at the end of a tile row
      bool fast = true;
      for (int x4 = 0; x4 < L.w; x4 += 4)
        if (h->xtab[L.xtabOff + std::min(x4 + 3, L.w - 1)].ofs + 1 - h->xtab[L.xtabOff + x4].ofs > 7) fast = false;
      L.pyrFast = fast ? 1 : 0;
      L.pyrTP = align_up(maxW, 4) + (fast ? 12 : 0);
      L.pyrTR = maxH;
      // Both tables are padded to a multiple of 4 entries (a thread fetches the taps of its 4 columns / 4 rows with two
      // 16-byte loads; padding taps have zero weights), followed by one entry per 256-column / 16-row output block that holds
      // the origin and the extent of the block's source tile (ofs = first source column / row, a0 = dwords per tile row /
      // rows): k_pyr_down then starts with ONE dependent table fetch instead of a chain of them.
      auto pad4 = [](std::vector<ResizeTap>& t, int base) {
        while (((int)t.size() - base) & 3) { ResizeTap z = t.back(); z.a0 = 0; z.a1 = 0; t.push_back(z); }
      };
      pad4(h->xtab, L.xtabOff);
      pad4(h->ytab, L.ytabOff);
      L.xtileOff = (int)h->xtab.size();
      for (int x0 = 0; x0 < L.pitch; x0 += 256) {
        ResizeTap e = {0, 0, 0, 0};
        if (x0 < L.w) {
          const int lo = h->xtab[L.xtabOff + x0].ofs & ~3;
          const int hi = std::min((int)h->xtab[L.xtabOff + std::min(x0 + 255, L.w - 1)].ofs + 1, sw - 1);
          e.ofs = (short)lo; e.a0 = (short)((hi - lo + 4) >> 2);
        }
        h->xtab.push_back(e);
      }
      pad4(h->xtab, L.xtabOff);
      L.ytileOff = (int)h->ytab.size();
      for (int y0 = 0; y0 < L.h; y0 += 16) {
        const int lo = std::min(std::max((int)h->ytab[L.ytabOff + y0].ofs, 0), sh - 1);
        const int hi = std::min(std::max((int)h->ytab[L.ytabOff + std::min(y0 + 15, L.h - 1)].ofs + 1, 0), sh - 1);
        ResizeTap e = {(short)lo, (short)(hi - lo + 1), 0, 0};
        h->ytab.push_back(e);
      }
      pad4(h->ytab, L.ytabOff);
    }
    // ComputeKeyPointsOctTree, ORBextractor.cc:771-787
    L.minBX = ORB_EDGE_THRESHOLD - 3; L.minBY = L.minBX;
    L.maxBX = L.w - ORB_EDGE_THRESHOLD + 3; L.maxBY = L.h - ORB_EDGE_THRESHOLD + 3;
    const float W = 30;
    const float width = (float)(L.maxBX - L.minBX), height = (float)(L.maxBY - L.minBY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    L.cellBase = (int)h->cells.size();
    L.slotOff = slotOff;
    if (nCols > 0 && nRows > 0) {
      const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(L.minBY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= L.maxBY - 3) continue;
        if (maxY > L.maxBY) maxY = (float)L.maxBY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(L.minBX + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= L.maxBX - 6) continue;
          if (maxX > L.maxBX) maxX = (float)L.maxBX;
          OrbCell c;
          c.level = (short)l;
          c.x0 = (short)iniX; c.y0 = (short)iniY;
          c.cw = (short)((int)maxX - (int)iniX); c.ch = (short)((int)maxY - (int)iniY);
          c.pad = 0;
          if (c.cw > ORB_CELL_MAX || c.ch > ORB_CELL_MAX) {
            set_error("cell %dx%d exceeds the LDS tile", c.cw, c.ch);
            return PLH_ERR_INVALID;
          }
          const int ew = std::max(c.cw - 6, 0), eh = std::max(c.ch - 6, 0);
          c.slotOff = slotOff;
          c.slotCap = ((ew + 1) / 2) * ((eh + 1) / 2);   // 3x3 strict NMS: survivors are never 8-adjacent
          slotOff += c.slotCap;
          h->cells.push_back(c);
        }
      }
    }
    L.nCells = (int)h->cells.size() - L.cellBase;
    L.slotCap = slotOff - L.slotOff;
    // strips: consecutive cells of one cell row (same y0) -> one k_fast_strips block
    for (int ci = L.cellBase; ci < (int)h->cells.size();) {
      int cj = ci;   // at most ~FAST_STRIP_MAX_W columns per strip: small LDS tiles keep >= 5 blocks per CU resident
#if defined(PLH_GROW_PROF)
      static const int stripW = [] {   // tuning knob of the counter build (tools/, profiles/r02_fast_strip_width_sweep.txt)
        const char* e = getenv("PLH_FAST_STRIP_W");
        const int v = e ? atoi(e) : 0;
        return v >= 64 && v <= 1000 ? v : FAST_STRIP_MAX_W;
      }();
#else
      const int stripW = FAST_STRIP_MAX_W;
#endif
      while (cj < (int)h->cells.size() && h->cells[cj].y0 == h->cells[ci].y0 &&
             (cj == ci || h->cells[cj].x0 + h->cells[cj].cw - h->cells[ci].x0 <= stripW)) cj++;
      OrbStrip st;
      st.level = (short)l; st.nCells = (short)(cj - ci); st.cellFirst = ci;
      st.x0 = h->cells[ci].x0; st.y0 = h->cells[ci].y0;
      st.xEnd = (short)(h->cells[cj - 1].x0 + h->cells[cj - 1].cw); st.ch = h->cells[ci].ch;
      st.wCell = (short)(cj - ci > 1 ? h->cells[ci + 1].x0 - h->cells[ci].x0 : std::max<int>(h->cells[ci].cw - 6, 1));
      st.pad = 0;
      h->strips.push_back(st);
      h->fastLds = std::max(h->fastLds, fast_strip_lds_bytes(st.xEnd - st.x0, st.ch));
      ci = cj;
    }
    // DistributeOctTree, ORBextractor.cc:543-545
    L.nFeat = h->perLevel[l];
    L.nIni = (int)std::round(static_cast<float>(L.maxBX - L.minBX) / (L.maxBY - L.minBY));
    if (L.nIni < 1) L.nIni = 1;   // the reference indexes vpIniNodes[..] out of bounds here (UB); guarded
    L.hX = static_cast<float>(L.maxBX - L.minBX) / L.nIni;
    L.selOff = selOff;
    L.selCap = std::max(L.nFeat + 3, 4 * L.nIni);
    selOff += L.selCap;
    L.nodeCap = L.selCap + 8;
    h->nodeCapMax = std::max(h->nodeCapMax, L.nodeCap);
    L.scale = h->sf[l];
    L.kpSize = (float)(int)(ORB_PATCH_SIZE * h->sf[l]);
  }
  h->pyrFrameBytes = align_up<long long>(off, 256);
  h->slotsPerFrame = align_up<long long>(slotOff, 64);
  h->selPerFrame = selOff;
  if (h->fastLds > 150 * 1024) {
    set_error("image too wide for the FAST strip tile (%zu bytes of LDS)", h->fastLds);
    return PLH_ERR_INVALID;
  }
  if (h->nodeCapMax > 8000 || octree_lds_bytes(h->nodeCapMax) > 150 * 1024) {
    set_error("nfeatures too large for the quad-tree LDS plan (node cap %d)", h->nodeCapMax);
    return PLH_ERR_INVALID;
  }
  return PLH_OK;
}

void fill_args(const plh_orb* h, const uint8_t* dImgs, long long stride, int batch, OrbDeviceArgs* a) {
  a->img0 = dImgs; a->stride0 = stride;
  a->pyr = h->dPyr; a->pyrFrameBytes = h->pyrFrameBytes;
  a->levels = h->dLevels; a->cells = h->dCells; a->strips = h->dStrips; a->nStrips = (int)h->strips.size(); a->xtab = h->dXtab; a->ytab = h->dYtab;
  a->slots = h->dSlots; a->slotsPerFrame = h->slotsPerFrame; a->cellCount = h->dCellCount;
  a->keys = h->dKeys; a->sel = h->dSel; a->selCount = h->dSelCount; a->selPerFrame = h->selPerFrame;
  a->nlevels = h->nlevels; a->nCellsTotal = (int)h->cells.size(); a->batch = batch;
  a->iniTh = std::min(std::max(h->p.ini_th_fast, 0), 255);
  a->minTh = std::min(std::max(h->p.min_th_fast, 0), 255);
  a->status = h->dStatus;
}

// Record one end of a timed interval (two marks per kernel id) on the caller's stream.
void prof_mark(plh_orb* h, int kind, hipStream_t s) {
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

template <typename T>
plh_status upload(const std::vector<T>& v, T** d) {
  const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  PLH_HIP(hipMalloc((void**)d, bytes));
  if (!v.empty()) PLH_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return PLH_OK;
}

}  // namespace

extern "C" {

const char* plh_last_error(void) { return plh::tls_error(); }
// PLH_BUILD_ID: hash of the sources this library was compiled from (set by __graft_entry__.build_hip); the GPU tests compare
// it with the hash of the sources next to them, so a stale shipped binary cannot pass for the current code.
#ifndef PLH_BUILD_ID
#define PLH_BUILD_ID "unknown"
#endif
const char* plh_version(void) {
#if defined(HIPEMU)
  return "plslam_hip 0.2 (hipemu CPU emulation -- test infrastructure, not the product) build " PLH_BUILD_ID;
#else
  return "plslam_hip 0.2 (gfx950) build " PLH_BUILD_ID;
#endif
}
int plh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// The kernels carry the patch disc and the blur taps as literals (orb_plan.h); recompute both the way the reference does --
// the quarter circle of ORBextractor.cc:454-469 and cvRound(getGaussianKernel(7, 2) * 256) of the 8-bit GaussianBlur -- once
// per process, so that an edit of one side cannot go unnoticed.
static bool orb_literals_match_reference() {
  int um[ORB_HALF_PATCH + 2] = {0};
  const int vmax = (int)std::floor(ORB_HALF_PATCH * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil(ORB_HALF_PATCH * std::sqrt(2.f) / 2);
  const double hp2 = (double)ORB_HALF_PATCH * ORB_HALF_PATCH;
  for (int v = 0; v <= vmax; v++) um[v] = (int)std::lrint(std::sqrt(hp2 - (double)v * v));
  for (int v = ORB_HALF_PATCH, v0 = 0; v >= vmin; v--) {
    while (um[v0] == um[v0 + 1]) v0++;
    um[v] = v0++;
  }
  for (int v = 0; v <= ORB_HALF_PATCH; v++)
    if (um[v] != ORB_UMAX[v]) return false;
  float cf[7];
  double sum = 0;
  for (int i = 0; i < 7; i++) { cf[i] = (float)std::exp(-0.5 / 4.0 * (i - 3.0) * (i - 3.0)); sum += cf[i]; }
  for (int i = 0; i < 7; i++) {
    const float w = (float)(cf[i] * (1.0 / sum));
    if ((unsigned)std::lrint(w * 256.f) != ORB_GAUSS7_Q8[i < 4 ? i : 6 - i]) return false;
  }
  return true;
}

plh_status plh_orb_create(const plh_orb_params* p, int device, int rows, int cols, int max_batch, plh_orb** out) {
  static const bool literalsOk = orb_literals_match_reference();
  if (!literalsOk) {
    set_error("plh_orb_create: the patch disc / Gaussian literals of orb_plan.h do not match the reference's construction");
    return PLH_ERR_INVALID;
  }
  if (!p || !out || rows <= 0 || cols <= 0 || max_batch <= 0 || p->nlevels < 1 || p->nlevels > ORB_MAX_LEVELS ||
      p->nfeatures < 0 || !(p->scale_factor > 1.0f)) {
    set_error("plh_orb_create: invalid argument");
    return PLH_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    set_error("plh_orb_create: no HIP device %d (count %d)", device, ndev);
    return PLH_ERR_NO_DEVICE;
  }
  PLH_HIP(hipSetDevice(device));
  plh_orb* h = new (std::nothrow) plh_orb();
  if (!h) return PLH_ERR_ALLOC;
  h->p = *p; h->device = device; h->rows = rows; h->cols = cols; h->maxBatch = max_batch;
  h->nlevels = p->nlevels;
  h->scaleFactorD = (double)p->scale_factor;
  build_tables(h);
  plh_status st = build_plan(h);
  if (st != PLH_OK) { delete h; return st; }
  const size_t B = (size_t)max_batch;
#define TRY(x) do { plh_status s__ = (x); if (s__ != PLH_OK) { plh_orb_destroy(h); return s__; } } while (0)
#define TRYHIP(x) do { if ((x) != hipSuccess) { set_error("plh_orb_create: %s failed (batch %d)", #x, max_batch); plh_orb_destroy(h); return PLH_ERR_ALLOC; } } while (0)
  TRY(upload(h->levels, &h->dLevels));
  TRY(upload(h->cells, &h->dCells));
  TRY(upload(h->strips, &h->dStrips));
  TRY(upload(h->xtab, &h->dXtab));
  TRY(upload(h->ytab, &h->dYtab));
  TRYHIP(hipMalloc((void**)&h->dPyr, std::max<size_t>(B * h->pyrFrameBytes, 256)));
  TRYHIP(hipMalloc((void**)&h->dSlots, B * h->slotsPerFrame * 4));
  TRYHIP(hipMalloc((void**)&h->dCellCount, B * std::max<size_t>(h->cells.size(), 1) * 4));
  TRYHIP(hipMalloc((void**)&h->dKeys, B * 2 * h->slotsPerFrame * 4));
  TRYHIP(hipMalloc((void**)&h->dSel, B * h->selPerFrame * 4));
  TRYHIP(hipMalloc((void**)&h->dSelCount, B * h->nlevels * 4));
  TRYHIP(hipMalloc((void**)&h->dStatus, 64));
  TRYHIP(hipMemset(h->dStatus, 0, 64));
  TRYHIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
#undef TRY
#undef TRYHIP
  *out = h;
  return PLH_OK;
}

plh_status plh_orb_destroy(plh_orb* h) {
  if (!h) return PLH_OK;
  (void)hipSetDevice(h->device);
  void* ptrs[] = {h->dLevels, h->dCells, h->dStrips, h->dXtab, h->dYtab, h->dPyr, h->dSlots, h->dCellCount, h->dKeys,
                  h->dSel, h->dSelCount, h->dStatus, h->dImgs, h->dKps, h->dDesc, h->dN};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  for (hipEvent_t e : h->evPool) (void)hipEventDestroy(e);
  if (h->doneEv) (void)hipEventDestroy(h->doneEv);
  delete h;
  return PLH_OK;
}

int plh_orb_levels(const plh_orb* h) { return h ? h->nlevels : 0; }
int plh_orb_capacity(const plh_orb* h) { return h ? h->selPerFrame : 0; }

plh_status plh_orb_scale_table(const plh_orb* h, int which, float* out) {
  if (!h || !out || which < 0 || which > 3) return PLH_ERR_INVALID;
  const std::vector<float>& v = which == 0 ? h->sf : which == 1 ? h->isf : which == 2 ? h->sig2 : h->isig2;
  std::copy(v.begin(), v.end(), out);
  return PLH_OK;
}

plh_status plh_orb_features_per_level(const plh_orb* h, int32_t* out) {
  if (!h || !out) return PLH_ERR_INVALID;
  std::copy(h->perLevel.begin(), h->perLevel.end(), out);
  return PLH_OK;
}

plh_status plh_orb_extract_batch_dev(plh_orb* h, const uint8_t* d_imgs, int batch, size_t frame_stride,
                                     plh_keypoint* d_kps, uint8_t* d_desc, int32_t* d_n, void* stream) {
  if (!h || !d_imgs || !d_kps || !d_desc || !d_n || batch <= 0 || batch > h->maxBatch ||
      frame_stride < (size_t)h->rows * h->cols) {
    set_error("plh_orb_extract_batch_dev: invalid argument (batch %d, plan max %d)", batch, h ? h->maxBatch : 0);
    return PLH_ERR_INVALID;
  }
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  hipStream_t s = (hipStream_t)stream;
  OrbDeviceArgs a;
  fill_args(h, d_imgs, (long long)frame_stride, batch, &a);
  // the capacity flags describe THIS call only (plh_orb_status): cleared in stream order in front of the kernels
  PLH_HIP(hipMemsetAsync(h->dStatus, 0, sizeof(int), s));
  prof_mark(h, 0, s);
  for (int l = 1; l < h->nlevels; l++) {
    const OrbLevel &S = h->levels[l - 1], &D = h->levels[l];
    PyrLaunch pl;
    pl.src = l == 1 ? d_imgs : h->dPyr + S.off;
    pl.srcStride = l == 1 ? (long long)frame_stride : h->pyrFrameBytes;
    pl.dst = h->dPyr + D.off; pl.dstStride = h->pyrFrameBytes;
    pl.sW = S.w; pl.sH = S.h; pl.sPitch = S.pitch; pl.dW = D.w; pl.dH = D.h; pl.dPitch = D.pitch; pl.TP = D.pyrTP;
    pl.xt = h->dXtab + D.xtabOff; pl.yt = h->dYtab + D.ytabOff;
    pl.xtile = D.xtileOff - D.xtabOff; pl.ytile = D.ytileOff - D.ytabOff;
    launch_pyr_down(a, l, D.pitch, D.h, (size_t)D.pyrTP * D.pyrTR, D.pyrFast ? &pl : nullptr, s);
    PLH_LAUNCH_CHECK();
  }
  prof_mark(h, 0, s);
  prof_mark(h, 1, s);
  launch_fast_strips(a, h->fastLds, s);
  PLH_LAUNCH_CHECK();
  prof_mark(h, 1, s);
  prof_mark(h, 2, s);
  launch_octree(a, h->nodeCapMax, s);
  PLH_LAUNCH_CHECK();
  prof_mark(h, 2, s);
  prof_mark(h, 3, s);
  launch_orient_brief(a, d_kps, d_desc, d_n, h->selPerFrame, s);
  PLH_LAUNCH_CHECK();
  prof_mark(h, 3, s);
  h->lastImgs = d_imgs; h->lastStride = (long long)frame_stride; h->lastBatch = batch;
  if (!h->doneEv) PLH_HIP(hipEventCreateWithFlags(&h->doneEv, hipEventDisableTiming));
  PLH_HIP(hipEventRecord(h->doneEv, s));
  h->doneValid = true;
  return PLH_OK;
}

static plh_status ensure_staging(plh_orb* h) {
  if (h->dImgs) return PLH_OK;
  const size_t B = (size_t)h->maxBatch;
  PLH_HIP(hipMalloc((void**)&h->dImgs, B * h->rows * h->cols));
  PLH_HIP(hipMalloc((void**)&h->dKps, B * h->selPerFrame * sizeof(plh_keypoint)));
  PLH_HIP(hipMalloc((void**)&h->dDesc, B * h->selPerFrame * 32));
  PLH_HIP(hipMalloc((void**)&h->dN, B * sizeof(int)));
  return PLH_OK;
}

static plh_status check_status(plh_orb* h) {
  int st = 0;
  PLH_HIP(hipMemcpy(&st, h->dStatus, sizeof(int), hipMemcpyDeviceToHost));
  if (st != 0) {
    set_error("ORB kernels reported a capacity overflow (flags 0x%x)", st);
    return PLH_ERR_CAPACITY;
  }
  return PLH_OK;
}

plh_status plh_orb_status(plh_orb* h, int* flags) {
  if (!h || !flags) return PLH_ERR_INVALID;
  PLH_HIP(hipSetDevice(h->device));
  if (h->doneValid) PLH_HIP(hipEventSynchronize(h->doneEv));
  PLH_HIP(hipMemcpy(flags, h->dStatus, sizeof(int), hipMemcpyDeviceToHost));
  return PLH_OK;
}

plh_status plh_orb_extract_batch(plh_orb* h, const uint8_t* imgs, int batch, size_t frame_stride, plh_keypoint* kps,
                                 uint8_t* desc, int32_t* n_out) {
  if (!h || !imgs || !kps || !desc || !n_out || batch <= 0 || batch > h->maxBatch) {
    set_error("plh_orb_extract_batch: invalid argument");
    return PLH_ERR_INVALID;
  }
  PLH_HIP(hipSetDevice(h->device));
  plh_status st = ensure_staging(h);
  if (st != PLH_OK) return st;
  const size_t fb = (size_t)h->rows * h->cols;
  PLH_HIP(hipMemcpy2DAsync(h->dImgs, fb, imgs, frame_stride, fb, batch, hipMemcpyHostToDevice, h->stream));
  st = plh_orb_extract_batch_dev(h, h->dImgs, batch, fb, h->dKps, h->dDesc, h->dN, h->stream);
  if (st != PLH_OK) return st;
  const size_t cap = h->selPerFrame;
  PLH_HIP(hipMemcpyAsync(n_out, h->dN, batch * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  PLH_HIP(hipMemcpyAsync(kps, h->dKps, batch * cap * sizeof(plh_keypoint), hipMemcpyDeviceToHost, h->stream));
  PLH_HIP(hipMemcpyAsync(desc, h->dDesc, batch * cap * 32, hipMemcpyDeviceToHost, h->stream));
  PLH_HIP(hipStreamSynchronize(h->stream));
  return check_status(h);
}

plh_status plh_orb_extract(plh_orb* h, const uint8_t* img, int rows, int cols, size_t step, plh_keypoint* kps,
                           uint8_t* desc, int cap, int* n_out) {
  if (!h || !n_out) return PLH_ERR_INVALID;
  if (rows == 0 || cols == 0 || !img) {   // reference: empty image -> silent return (ORBextractor.cc:1046-1047)
    *n_out = 0;
    return PLH_OK;
  }
  if (rows != h->rows || cols != h->cols || step < (size_t)cols || !kps || !desc) {
    set_error("plh_orb_extract: image %dx%d does not match the plan %dx%d", rows, cols, h->rows, h->cols);
    return PLH_ERR_INVALID;
  }
  PLH_HIP(hipSetDevice(h->device));
  plh_status st = ensure_staging(h);
  if (st != PLH_OK) return st;
  PLH_HIP(hipMemcpy2DAsync(h->dImgs, cols, img, step, cols, rows, hipMemcpyHostToDevice, h->stream));
  st = plh_orb_extract_batch_dev(h, h->dImgs, 1, (size_t)rows * cols, h->dKps, h->dDesc, h->dN, h->stream);
  if (st != PLH_OK) return st;
  int n = 0;
  PLH_HIP(hipMemcpyAsync(&n, h->dN, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  PLH_HIP(hipStreamSynchronize(h->stream));
  if (n > cap) {
    set_error("plh_orb_extract: %d keypoints exceed the caller's capacity %d", n, cap);
    return PLH_ERR_CAPACITY;
  }
  PLH_HIP(hipMemcpy(kps, h->dKps, (size_t)n * sizeof(plh_keypoint), hipMemcpyDeviceToHost));
  PLH_HIP(hipMemcpy(desc, h->dDesc, (size_t)n * 32, hipMemcpyDeviceToHost));
  *n_out = n;
  return check_status(h);
}

plh_status plh_orb_set_profiling(plh_orb* h, int on) {
  if (!h) return PLH_ERR_INVALID;
  h->profiling = on != 0;
  h->evUsed = 0;
  for (int k = 0; k < 4; k++) { h->kernelMs[k] = 0; h->kernelLaunches[k] = 0; }
  return PLH_OK;
}

plh_status plh_orb_kernel_ms(plh_orb* h, int kernel, double* total_ms, int* intervals) {
  if (!h || kernel < 0 || kernel > 3 || !total_ms || !intervals) return PLH_ERR_INVALID;
  // fold the recorded event pairs (caller must have synchronised the stream)
  for (size_t i = 0; i + 1 < h->evUsed; i += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->evPool[i], h->evPool[i + 1]) == hipSuccess) {
      h->kernelMs[h->evKind[i]] += ms;
      h->kernelLaunches[h->evKind[i]]++;
    }
  }
  h->evUsed = 0;
  *total_ms = h->kernelMs[kernel];
  *intervals = h->kernelLaunches[kernel];
  return PLH_OK;
}

plh_status plh_orb_pyramid_dev(const plh_orb* h, int b, int level, const uint8_t** d_ptr, int* rows, int* cols,
                               size_t* pitch) {
  if (!h || level < 0 || level >= h->nlevels || b < 0 || b >= h->maxBatch || !d_ptr) return PLH_ERR_INVALID;
  const OrbLevel& L = h->levels[level];
  if (level == 0) {
    if (!h->lastImgs) return PLH_ERR_INVALID;
    *d_ptr = h->lastImgs + (long long)b * h->lastStride;
  } else {
    *d_ptr = h->dPyr + (long long)b * h->pyrFrameBytes + L.off;
  }
  if (rows) *rows = L.h;
  if (cols) *cols = L.w;
  if (pitch) *pitch = (size_t)L.pitch;
  return PLH_OK;
}

plh_status plh_orb_read_level(plh_orb* h, int b, int level, uint8_t* out, size_t out_bytes) {
  const uint8_t* d = nullptr;
  int rows = 0, cols = 0;
  size_t pitch = 0;
  plh_status st = plh_orb_pyramid_dev(h, b, level, &d, &rows, &cols, &pitch);
  if (st != PLH_OK) return st;
  if (!out || out_bytes < (size_t)rows * cols) return PLH_ERR_INVALID;
  PLH_HIP(hipSetDevice(h->device));
  PLH_HIP(hipDeviceSynchronize());
  PLH_HIP(hipMemcpy2DAsync(out, cols, d, pitch, cols, rows, hipMemcpyDeviceToHost, h->stream));
  PLH_HIP(hipStreamSynchronize(h->stream));
  return PLH_OK;
}

plh_status plh_orb_read_candidates(plh_orb* h, int b, int level, plh_keypoint* out, int cap, int* n_out) {
  if (!h || level < 0 || level >= h->nlevels || b < 0 || b >= h->maxBatch || !n_out) return PLH_ERR_INVALID;
  PLH_HIP(hipSetDevice(h->device));
  PLH_HIP(hipDeviceSynchronize());
  const OrbLevel& L = h->levels[level];
  std::vector<uint32_t> counts(std::max(L.nCells, 1)), slots(std::max(L.slotCap, 1));
  const size_t nc = h->cells.size();
  if (L.nCells)
    PLH_HIP(hipMemcpy(counts.data(), h->dCellCount + (size_t)b * nc + L.cellBase, (size_t)L.nCells * 4, hipMemcpyDeviceToHost));
  if (L.slotCap)
    PLH_HIP(hipMemcpy(slots.data(), h->dSlots + (size_t)b * h->slotsPerFrame + L.slotOff, (size_t)L.slotCap * 4, hipMemcpyDeviceToHost));
  int n = 0;
  for (int c = 0; c < L.nCells; c++) {
    const OrbCell& cell = h->cells[L.cellBase + c];
    for (uint32_t k = 0; k < counts[c]; k++) {
      const uint32_t key = slots[cell.slotOff - L.slotOff + k];
      if (out && n < cap) {
        out[n].x = (float)(key >> 20); out[n].y = (float)((key >> 8) & 0xfff);
        out[n].size = 7.f; out[n].angle = -1.f; out[n].response = (float)(key & 0xff);
        out[n].octave = level; out[n].class_id = -1;
      }
      n++;
    }
  }
  *n_out = n;
  return PLH_OK;
}

}  // extern "C"
