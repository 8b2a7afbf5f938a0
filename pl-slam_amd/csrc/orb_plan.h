// ORB extractor plan: geometry tables shared by host planning code and the HIP kernels.
// Everything here is derived from the reference's constructor / per-level setup
// (reference src/ORBextractor.cc:410-470, 765-806, 1107-1113).
#pragma once
#include <cstdint>

#include "plh_xcd.h"

namespace plh {

constexpr int ORB_MAX_LEVELS = 16;
constexpr int ORB_PATCH_SIZE = 31;        // ORBextractor.cc:72
constexpr int ORB_HALF_PATCH = 15;        // ORBextractor.cc:73
constexpr int ORB_EDGE_THRESHOLD = 19;    // ORBextractor.cc:74
constexpr int ORB_CELL_MAX = 66;          // max cell sub-image side (wCell <= 60, +6)
constexpr int ORB_KEY_XY_BITS = 12;       // packed candidate key: x:12 | y:12 | response:8
// u_max of the circular 31-px patch (ORBextractor.cc:454-469) and the non-negative half of cvRound(getGaussianKernel(7, 2) * 256)
// (GaussianBlur(7x7, 2, 2) on 8-bit data, ORBextractor.cc:1090): literals so that the kernels fold them into dot-product
// operands; plh_orb_create recomputes both the reference's way and refuses to run if they differ.
constexpr int ORB_UMAX[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
constexpr unsigned ORB_GAUSS7_Q8[4] = {18, 34, 49, 55};

struct OrbLevel {
  int w, h, pitch;             // level image; pitch in bytes (level 0 uses the caller's pitch = cols)
  long long off;               // byte offset inside one frame's pyramid block (levels >= 1)
  int minBX, minBY, maxBX, maxBY;   // FAST window  [16, w-16) x [16, h-16)
  int cellBase, nCells;        // this level's cells in the cell table (reference loop order: row-major)
  int slotOff, slotCap;        // candidate slots of the level inside one frame's slot array
  int nFeat;                   // mnFeaturesPerLevel[l]
  int nIni;                    // initial quad-tree nodes = round(W/H)
  float hX;                    // W / nIni
  int selOff, selCap;          // selected-keypoint slots of the level inside one frame's record array
  int nodeCap;                 // quad-tree node slots
  int xtabOff, ytabOff;        // resize tables (levels >= 1)
  int xmax;                    // first dx using the single-tap path (cv::resize)
  int pyrTP, pyrTR;            // k_pyr_down source tile of a 256 x 16 output block: pitch (bytes, multiple of 4) and rows
  int xtileOff, ytileOff;      // per output block column / row: source tile origin and extent (entries of xtab / ytab)
  int pyrFast;                 // 1: the 8 taps of every 4-pixel group lie in one 8-byte window (k_pyr_down), 0: k_pyr_down_gather
  float scale;                 // mvScaleFactor[l]
  float kpSize;                // (int)(31 * mvScaleFactor[l])
};

struct OrbCell {               // one FAST cell = one cv::FAST call on a sub-image (ORBextractor.cc:789-829)
  short level;
  short x0, y0;                // sub-image origin in level-image coordinates
  short cw, ch;                // sub-image size (evaluated window is [3,cw-3) x [3,ch-3))
  short pad;
  int slotOff;                 // first candidate slot of this cell (inside the frame's slot array)
  int slotCap;                 // ceil(ew/2)*ceil(eh/2): hard upper bound of 3x3-NMS survivors
};

struct OrbStrip {              // one row of FAST cells of a level = the work of one k_fast_strips block
  short level;
  short nCells;                // cells in the row (consecutive entries of the cell table)
  int cellFirst;               // index of the row's first cell
  short x0, y0;                // origin of the row's sub-images (level-image coordinates): x0 = first cell's x0
  short xEnd, ch;              // last cell's x0 + cw ; sub-image height (evaluated rows are [y0+3, y0+ch-3))
  short wCell, pad;            // nominal cell width: cell j's evaluated window starts at x0 + 3 + j*wCell
};

struct ResizeTap {             // one entry of the cv::resize coefficient tables
  short ofs, a0, a1, pad;
};

struct PyrLaunch {             // everything one k_pyr_down launch needs, by value (no dependent table-of-levels fetches)
  const uint8_t* src;          // level l-1 of frame 0
  long long srcStride;         // bytes between frames of the source level
  uint8_t* dst;                // level l of frame 0
  long long dstStride;
  int sW, sH, sPitch, dW, dH, dPitch;
  int TP;                      // tile pitch (bytes)
  const ResizeTap* xt;         // the level's x taps (padded), then its per-block-column tile entries at xt + xtile
  const ResizeTap* yt;
  int xtile, ytile;
  PlhXcdGrid xg;               // the launch's tiles per frame and frames (set by launch_pyr_down: plh_xcd.h)
};

struct OrbDeviceArgs {         // kernel argument block (passed by value)
  const uint8_t* img0;         // batch input (level 0)
  long long stride0;           // bytes between frames of the input
  uint8_t* pyr;                // levels >= 1, frame-major
  long long pyrFrameBytes;
  const OrbLevel* levels;
  const OrbCell* cells;
  const OrbStrip* strips;
  int nStrips;
  const ResizeTap* xtab;
  const ResizeTap* ytab;
  uint32_t* slots;             // candidate slots [frame][slot]
  long long slotsPerFrame;
  uint32_t* cellCount;         // [frame][cell]
  uint32_t* keys;              // quad-tree key scratch: [frame][2][slotsPerFrame]
  uint32_t* sel;               // selected keys [frame][selPerFrame]
  int* selCount;               // [frame][nlevels]
  int selPerFrame;
  int nlevels, nCellsTotal, batch;
  int iniTh, minTh;
  int* status;                 // device-side error flag (capacity overflow etc.)
};

}  // namespace plh
