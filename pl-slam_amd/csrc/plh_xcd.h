// XCD-aware block order (MI355X: eight XCDs with an L2 each, workgroups go to them round robin by linear id).
//
// A (blocks of a frame) x (frames) grid spreads the blocks of ONE frame over all eight L2s, and whatever those blocks share (the
// level images behind overlapping keypoint patches, the planes behind a frame's rectangles and line bands, the halo rows of a
// stencil's tiles) is fetched from HBM once per XCD that touches it.  With a one-dimensional grid decoded here block L runs on XCD
// (L % 8) and takes frame 8 (L / 8 / perFrame) + L % 8: all blocks of a frame sit behind one L2 and are dispatched side by side
// (DESIGN.md 3.5).  Batches below PLH_XCD_MIN_BATCH frames keep the plain order (a lone frame wants all 256 CUs, and nine frames
// would put two on one XCD and one on each of the others).
//
// The two divisions of the decode are by launch constants: the host passes floor(2^32 / d) beside d (PlhXcdGrid, by value in the
// kernel arguments) and the block divides with one s_mul_hi_u32 and a fix-up -- a division by a run-time value costs every
// wavefront ~ 25 instructions, ten of them on the vector unit (the compiler's float-reciprocal sequence), which for a kernel of
// one-row blocks was a tenth of its instruction stream (profiles/r06b_xcd_decode_magic_ab.txt).
#pragma once
#include <cstdint>

namespace plh {

#if defined(HIPEMU)
constexpr int PLH_XCD_MIN_BATCH = 8;    // (the CPU emulator's small batches walk the decode too)
#else
constexpr int PLH_XCD_MIN_BATCH = 64;
#endif

struct PlhXcdGrid {
  int perFrame, nx, batch;   // blocks of a frame (= nx * ny for a tiled kernel, nx = perFrame otherwise), frames of the launch
  uint32_t mPer, mNx;        // floor(2^32 / perFrame), floor(2^32 / nx)
};

inline uint32_t plh_xcd_magic(int d) { return d <= 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint64_t)d); }
inline PlhXcdGrid plh_xcd_make(int nx, int ny, int batch) {
  PlhXcdGrid g;
  g.perFrame = nx * ny; g.nx = nx; g.batch = batch;
  g.mPer = plh_xcd_magic(g.perFrame); g.mNx = plh_xcd_magic(nx);
  return g;
}
inline PlhXcdGrid plh_xcd_make(int perFrame, int batch) { return plh_xcd_make(perFrame, 1, batch); }
// the grid of a launch decoded with g
inline unsigned plh_xcd_grid(const PlhXcdGrid& g) {
  return g.batch < PLH_XCD_MIN_BATCH ? (unsigned)(g.perFrame * g.batch) : (unsigned)((long long)g.perFrame * ((g.batch + 7) / 8) * 8);
}

}  // namespace plh
