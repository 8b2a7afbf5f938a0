// Line extractor plan: geometry and workspace layout shared by host code and the HIP kernels.
// Constants follow cv::LineSegmentDetector's defaults (LSD_REFINE_STD) and cv::line_descriptor::BinaryDescriptor's
// parameters as used by LINEextractor::operator() (reference src/LineExtractor.cpp:39-40,77-78).
#pragma once
#include <cstdint>

#include "orb_plan.h"   // ResizeTap

namespace plh {

constexpr int LBD_NUM_BANDS = 9;      // NUM_OF_BANDS, binary_descriptor_custom.cpp:57
constexpr int LBD_BAND_WIDTH = 7;     // Params::widthOfBand_, binary_descriptor_custom.cpp:112
constexpr int LBD_ROWS = LBD_NUM_BANDS * LBD_BAND_WIDTH;   // 63 rows of the line support region
constexpr int LSD_NBINS = 1024;

// ll_angle() as a table.  The 2x2 gradient of 8-bit pixels is a pair of integers in [-510, 510], and everything
// ll_angle() and its consumers derive from it per pixel -- fastAtan2 in degrees, the float cos / sin of the double angle (seed
// terms) and of the float angle (region increments), each a correctly rounded double sincos, and the gradient norm
// sqrt((gx^2 + gy^2) / 4) -- depends on that pair alone.  The table is filled once per device; a pixel of the level-line
// field is then nothing but its 20-bit table index (gy + 510) << 10 | (gx + 510) plus two flag bits -- a 4-byte record
// instead of 16 + 8 bytes -- and the kernels gather what they need from the table, whose hot part (a frame touches
// ~19 k of the 1.04 M entries) lives in L2.  That cut k_lsd_grow's HBM traffic, which bounded it at full residency:
// region growing reads records at random, and 64-byte sectors that held 4 records now hold 16.
constexpr int LSD_GRAD_MAX = 510, LSD_ANGLE_ROWS = 2 * LSD_GRAD_MAX + 1, LSD_ANGLE_PITCH_LOG2 = 10;
struct LsdAngleEntry {
  float angf, cs, sn, seedx;   // first 16 bytes: what a region-growing candidate needs (+ cos of the double angle)
  float seedy, pad;
  double modgrad;              // sqrt((gx^2 + gy^2) / 4.0): region2rect()'s weights
};
// level-line record of a pixel of the 0.8x image (u32): table index | DEF (gradient above the threshold) | USED (region growing's mark)
constexpr uint32_t LSD_REC_IDX = 0x000fffffu, LSD_REC_DEF = 0x40000000u, LSD_REC_USED = 0x80000000u;
// Where the record of pixel (x, y) lies in a frame's record plane: 4 x 4-pixel blocks of 64 bytes, blocks in row-major order
// (round 4).  Region growing reads 3 x 3 neighbourhoods along segments of every direction; in a row-major plane a region that
// advances by a row touches a new 128-byte line for every row, and at full residency (768 frames per 4 MiB L2) that line comes
// from HBM: the step waits for it.  Two side-by-side blocks are one such line (8 x 4 pixels), a block is one 64-byte sector.
// Four pixels x .. x+3 of a row (x a multiple of 4) are still 16 contiguous bytes.  The pitch is a multiple of 64; the plane
// holds lsd_rec_rows(sh) rows.
#if defined(__HIPCC__) || defined(__HIP__)
#define PLH_REC_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define PLH_REC_HD inline
#endif
PLH_REC_HD uint32_t lsd_rec_index(uint32_t x, uint32_t y, uint32_t spitch) {
  return (y >> 2) * (spitch << 2) + ((((x & ~3u) + (y & 3u)) << 2) | (x & 3u));
}
PLH_REC_HD int lsd_rec_rows(int sh) { return (sh + 3) & ~3; }
// cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) with a fixed map (Frame.cc:220-222) reduced to what depends on the map alone:
// per output pixel the offset of the top-left byte of a 2 x 2 source block that lies inside the image, and the four axis
// weights (0..32, one byte each: column 0, column 1, row 0, row 1) of that block.  A tap of the reference that falls
// outside the image contributes 0, and it does so per axis, so it becomes a zero weight; the block is moved inside the
// image where needed and the weights move with it.  pixel = (sum_rc wy[r] wx[c] P[r][c] + 512) >> 10, saturated.
struct RemapTap {
  uint32_t off;   // y0 * w + x0
  uint32_t wts;   // wx0 | wx1 << 8 | wy0 << 16 | wy1 << 24
};

struct LsdAdvRec;   // lsd_rect_dev.h
struct LineDeviceArgs {
  // geometry
  int w, h;                 // full-resolution image
  int sw, sh, spitch;       // LSD's 0.8x image
  int batch;
  // per-frame strides (elements)
  long long fullStride;     // w*h rounded up (u8 planes at full resolution)
  long long scaledStride;   // spitch*sh rounded up
  // The per-frame working set of the sequential stages (segment list, region queue, level-line records, seed list, scratch,
  // ordering counters) is ONE contiguous block per frame, blocks `arenaStride` 4-byte words apart and 2 MiB aligned: a
  // region-growing wavefront then touches one or two translation fragments instead of six (at full residency 4.4 % of
  // k_lsd_grow's L1 requests missed the L1 TLB with one array per buffer).  pix / reg / ordered / scr / orderWork / segs
  // below point at frame 0's part of the block; frame b's is + b * arenaStride words.
  long long arenaStride;
  // inputs / intermediates (frame-major)
  const uint8_t* img;       // caller's frames (pitch = w)
  long long imgStride;
  const RemapTap* remap;    // undistortion taps per output pixel, or null
  uint8_t* undist;          // remapped frames (== img when no undistortion), pitch w
  uint8_t* tmpA;            // full-res scratch plane (blur output), pitch w
  uint8_t* scaled;          // 0.8x image, pitch spitch
  uint32_t* pix;            // level-line record per scaled pixel (LSD_REC_*), at lsd_rec_index(x, y, spitch)
  uint32_t* ordered;        // seed list (packed coordinates x | y << 16), bins descending / raster inside a bin
  uint32_t* reg;            // region point queue; behind region growing: the frame's log of kept regions (packed coordinates)
  uint32_t* regq;           // beside every log entry: gx^2 + gy^2 of the pixel (what its region2rect() weight is the root of)
  uint32_t* orderWork;      // seed ordering: per frame counts / offsets [16 chunks][1024 bins] + the 1024 bin thresholds (line_kernels.hip)
  unsigned int* qmax;       // per frame max(gx^2+gy^2) over defined pixels
  int* nOrdered;            // per frame
  float* segs;              // [frame][segCap][4]
  int* nSegs;               // per frame
  int segCap;
  uint32_t* dxdy;           // Sobel (dx:int16 | dy:int16 << 16) of the 5x5-blurred full-res frame, pitch w
  const ResizeTap* xtab;
  const ResizeTap* ytab;
  const uint8_t* mask;      // optional w*h mask shared by all frames, or null
  const LsdAngleEntry* angleTab;   // per-device ll_angle() table (line_plan.h)
  // LSD constants (computed on the host in double exactly as flsd() does)
  double prec, p, densityTh;
  unsigned int qThresh;     // pixel is NOTDEF  <=>  gx^2+gy^2 <= qThresh  (<=> sqrt(q/4) <= rho)
  float alignCin2, alignCout2;   // cos^2(prec -+ 0.05 degrees): the direction pre-test of region growing (lsd_grow.hip, lsd_classify)
  int alignFast;
  int minRegSize;
  // the density screen of region growing (lsd_density_screen, lsd_grow.hip): 0.7 (1 -+ 2e-5) as floats; screen = 0 evaluates the
  // exact rectangle for every decision (the path every undecided region takes anyway: same segments; an A/B and test switch)
  float screenLo, screenHi;
  int screen;
  uint32_t* park;           // per frame 2 + 2 segCap words: [0] = length of LSD_REFINE_ADV's work list; from word 2: k_lsd_rects' size-class
                            // order of the kept regions (segCap words), then the work list (slots of the rectangles in rect_improve())
  const double* lgamma;     // LSD_REFINE_ADV: lsd_log_gamma(i) for i = 1 .. sw sh + 1 (every argument nfa() can have), per handle
  LsdAdvRec* adv;           // LSD_REFINE_ADV: segCap records per frame (lsd_rect_dev.h), allocated when the level is first used
  float* advAng;            // LSD_REFINE_ADV: level-line angle per scaled pixel in float degrees (-1024 = NOTDEF), written by k_lsd_grad:
                            // rect_nfa()'s scan reads one float per pixel instead of record -> table; scaledStride floats per frame
  int refineAdv;            // 1: LSD_REFINE_ADV (rect_improve / NFA on the kept regions' rectangles, k_lsd_rects), plh_line_set_refine
  double logNT;             // 5 (log10 sw + log10 sh) / 2 + log10 11, flsd()'s LOG_NT
  // selection
  int nFeature;             // nLSDFeature
  double minLineLength;
  int outCap;               // nFeature + 1
  int* status;
  // multi-wavefront region growing (k_lsd_grow_mw, small batches): per frame mwWaves + 1 slots (one per wavefront, one for
  // the committing wavefront's re-runs), each a log arena of mwRegStride = 5 x scaledStride words (posted logs below 1 x, a
  // running transaction's three region queues, reduce_region_radius scratch: no phase can overflow it) and a private mark
  // plane of mwMarkStride bytes (zero between transactions); null when the batch runs one wavefront per frame
  uint32_t* mwReg;
  uint8_t* mwMark;
  uint16_t* mwHint;         // per frame: claim hints shared by its wavefronts (mwMarkStride 16-bit tags, cleared by the kernel)
  long long mwRegStride, mwMarkStride;
  // LINEextractor(numOctaves = 2, scale in [2, 3)): the second octave -- pyrDown of the frame (detection) and of the 5x5-blurred
  // frame (description), half size; its LSD runs on a plan of its own (plh_line::oct1), k_keylines / k_lbd read both.  All zero /
  // null with one octave.
  const float* segs1;       // octave 1's segments [frame][segCap1][4], frames arena1Stride words apart
  const int* nSegs1;
  long long arena1Stride;
  int segCap1, w1, h1;      // octave 1's segment capacity and image size
  float octScale1;          // pow((float)(int)scale, 1) = 2
  const uint32_t* dxdy1;    // Sobel of octave 1's blurred image, pitch w1, frames full1Stride apart
  long long full1Stride;
  int mwWaves;              // wavefronts per frame of this launch (0: k_lsd_grow / k_lsd_grow_lone)
  int mwLag;                // a wavefront starts no transaction further than this ahead of the commits
  int mwDrainGap;           // a wavefront that posted tries to commit when the commits are this far behind it (or it is the oldest)
};

}  // namespace plh
