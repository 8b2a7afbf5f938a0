// HIP kernels of the ORB extractor hot path (gfx950 / CDNA4, wave64).  Batch-first: every kernel
// takes the frame index from the grid, because one 640x480 frame is ~1 us of HBM time and can
// never fill 256 CUs by itself (DESIGN.md "batch-first").
//
//   k_pyr_down      cv::resize(INTER_LINEAR) level l-1 -> l      ORBextractor::ComputePyramid   (ORBextractor.cc:1107-1132)
//   k_fast_strips   per-cell cv::FAST 9/16 + 3x3 NMS + fallback   ComputeKeyPointsOctTree        (ORBextractor.cc:789-829)
//   k_octree        quad-tree keypoint distribution               DistributeOctTree / DivideNode (ORBextractor.cc:481-763)
//   k_orient_brief  IC_Angle + 7x7 blur + steered rBRIEF          IC_Angle / GaussianBlur / computeOrbDescriptor
//                                                                 (ORBextractor.cc:77-147, 1085-1101)
//
// All integer work is exact; the only floating point on the path is float32 with no FMA
// contraction (this file is compiled with -ffp-contract=off) so results are bit-identical to
// the CPU oracle's pinned definition.
#include "orb_plan.h"
#include "plh_common.h"

namespace plh {

// rBRIEF sampling pattern (data): 256 pairs, row i = descriptor byte i, as floats: the steering multiplies them, and a
// signed-byte load + sign extension + int -> float conversion per coordinate would be 16 x 3 VALU instructions per lane.
__device__ const float c_orb_pattern_f[1024] = {
#include "../../include/plh_orb_pattern.inc"
};

// IC_Angle weights.  The integer moments of the circular 31-px patch (u_max from ORB_UMAX, orb_plan.h; the host checks that
// table against the reference's construction, ORBextractor.cc:454-469, at create time) are linear in the pixels, so they
// are taken four pixels at a time with v_dot4_u32_u8: entry (row r = v + 15, dword q) holds the byte weights (u + 15) and
// the 0 / 1 disc mask of columns u = 4q - 15 .. 4q - 12.  m10 = sum (u + 15) I - 15 sum I, m01 = sum v (row sum).
struct IcWeights { unsigned w[32 * 8 * 2]; };
constexpr IcWeights make_ic_weights() {
  IcWeights t{};
  for (int r = 0; r < 31; r++) {
    const int v = r - 15, av = v < 0 ? -v : v;
    for (int q = 0; q < 8; q++) {
      unsigned wu = 0, w1 = 0;
      for (int k = 0; k < 4; k++) {
        const int j = 4 * q + k, u = j - 15, au = u < 0 ? -u : u;
        if (j < 31 && au <= ORB_UMAX[av]) { wu |= (unsigned)(u + 15) << (8 * k); w1 |= 1u << (8 * k); }
      }
      t.w[(r * 8 + q) * 2] = wu;
      t.w[(r * 8 + q) * 2 + 1] = w1;
    }
  }
  return t;   // row 31: zero weights (the lanes past the last row run the same code)
}
__device__ const IcWeights c_icw = make_ic_weights();

__device__ __forceinline__ const uint8_t* level_ptr(const OrbDeviceArgs& a, const OrbLevel& lv, int level, int b) {
  return level == 0 ? a.img0 + (long long)b * a.stride0 : a.pyr + (long long)b * a.pyrFrameBytes + lv.off;
}

__device__ __forceinline__ unsigned align_bytes_u(unsigned hi, unsigned lo, int sh) {   // bytes sh..sh+3 of {hi:lo}, sh in 0..3
  return plh_alignbyte(hi, lo, (unsigned)sh);   // one v_alignbyte_b32 instead of a 64-bit shift
}

// ---------------------------------------------------------------------------------------------
// Pyramid: one level-to-level bilinear downscale, OpenCV fixed-point semantics.
// grid (ceil(pitch/256), ceil(h/16), batch), block (64,4) = a 256 x 16 output tile.  The source pixels behind the tile
// (exact extent precomputed on the host: pyrTP x pyrTR, 7 KiB at scale 1.2) are staged in LDS with aligned dword loads;
// every thread then produces 4 pixels (one aligned 32-bit store) of 4 consecutive rows, reading its taps from LDS.
// (Gathering the taps straight from memory -- 4 byte loads per pixel -- ran at 350 Gpixel/s whatever the number of
// loads in flight: bound by the vector-memory address path.)  Coefficient tables are built on the host in float
// exactly as cv::resize does, so the device does integer work only.
// ---------------------------------------------------------------------------------------------
constexpr int PYR_ROWS = 4;

// (v_perm_b32 / v_dot2_u32_u16 / v_dot4_u32_u8 / v_pk_min_u16: plh_perm, plh_udot2, plh_udot4, plh_pk_min_u16 of plh_shims.h)
// dot4 / dot2 operands from tap weights (byte / half 0 = lowest address)
constexpr unsigned w4(unsigned a, unsigned b, unsigned c, unsigned d) { return a | (b << 8) | (c << 16) | (d << 24); }
constexpr unsigned w2(unsigned lo, unsigned hi) { return lo | (hi << 16); }

// Source tile of a 256 x 16 output block -> LDS (aligned dword loads, byte funnel for odd row addresses); shared by both
// pyramid kernels.  Returns the tile origin (xBase, syBase).
__device__ __forceinline__ void pyr_stage_tile(const OrbLevel& S, const OrbLevel& D, const ResizeTap* xt, const ResizeTap* yt,
                                               const uint8_t* src, unsigned char* smem, int xb, int yb, int tid, int& xBase, int& syBase) {
  const int TP = D.pyrTP;
  xBase = xb < D.w ? (xt[xb].ofs & ~3) : 0;
  const int xHi = xb < D.w ? min((int)xt[min(xb + 255, D.w - 1)].ofs + 1, S.w - 1) : -1;
  syBase = min(max((int)yt[min(yb, D.h - 1)].ofs, 0), S.h - 1);
  const int syHi = min(max((int)yt[min(yb + 15, D.h - 1)].ofs + 1, 0), S.h - 1);
  const int nd = (xHi - xBase + 4) >> 2, nrows = syHi - syBase + 1;   // dwords per tile row (0 for an all-padding block)
  // i / nd by multiplication (exact while i * nd < 2^18; a division by a run-time value costs ~30 VALU instructions per
  // element -- more than the rest of the staging loop)
  const unsigned rowMul = (1u << 18) / (unsigned)max(nd, 1) + 1u;
  const bool mulOk = (unsigned)(nrows * nd) * (unsigned)nd < (1u << 18);   // uniform; false only for scale factors near 2
  for (int i = tid; i < nrows * nd; i += 256) {
    const int r = mulOk ? (int)(__umul24((unsigned)i, rowMul) >> 18) : i / nd, d = i - __mul24(r, nd);
    const int xs = xBase + 4 * d;
    const uint8_t* rowp = src + (long long)(syBase + r) * S.pitch + xs;
    const int m = (int)((size_t)rowp & 3);
    unsigned v;
    if (xs + 4 + (m ? 4 : 0) <= S.pitch) {   // the aligned dword (pair) stays inside the source row
      const unsigned* ap = reinterpret_cast<const unsigned*>(rowp - m);
      const unsigned lo = ap[0];
      v = m ? (unsigned)(((((unsigned long long)ap[1]) << 32) | lo) >> (8 * m)) : lo;
    } else {
      v = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) v |= (unsigned)rowp[min(k, S.w - 1 - xs)] << (8 * k);
    }
    reinterpret_cast<unsigned*>(smem)[r * (TP >> 2) + d] = v;
  }
}

// General form: every tap is a byte read from the LDS tile (any scale factor).
__global__ void __launch_bounds__(256) k_pyr_down_gather(OrbDeviceArgs a, int l) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  const OrbLevel S = a.levels[l - 1];
  const OrbLevel D = a.levels[l];
  const int b = blockIdx.z, tid = (int)threadIdx.y * 64 + (int)threadIdx.x;
  const int xb = (int)blockIdx.x * 256, yb = (int)blockIdx.y * 16;
  const uint8_t* src = level_ptr(a, S, l - 1, b);
  uint8_t* dst = a.pyr + (long long)b * a.pyrFrameBytes + D.off;
  const ResizeTap* xt = a.xtab + D.xtabOff;
  const ResizeTap* yt = a.ytab + D.ytabOff;
  const int TP = D.pyrTP;
  int xBase, syBase;
  pyr_stage_tile(S, D, xt, yt, src, smem, xb, yb, tid, xBase, syBase);
  __syncthreads();
  const int x4 = xb + (int)threadIdx.x * 4;
  const int y0 = yb + (int)threadIdx.y * PYR_ROWS;
  if (y0 >= D.h || x4 >= D.pitch) return;
  ResizeTap tx[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    tx[k].ofs = (short)xBase; tx[k].a0 = 0; tx[k].a1 = 0;
    if (x4 + k < D.w) tx[k] = xt[x4 + k];
  }
#pragma unroll
  for (int r = 0; r < PYR_ROWS; r++) {
    if (y0 + r >= D.h) break;
    const ResizeTap ty = yt[y0 + r];
    const int sy0 = min(max((int)ty.ofs, 0), S.h - 1);
    const int sy1 = min(max((int)ty.ofs + 1, 0), S.h - 1);
    const uint8_t* r0 = smem + (sy0 - syBase) * TP - xBase;
    const uint8_t* r1 = smem + (sy1 - syBase) * TP - xBase;
    const int b0 = ty.a0, b1 = ty.a1;
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (x4 + k < D.w) {
        int s0 = __mul24((int)r0[tx[k].ofs], (int)tx[k].a0), s1 = __mul24((int)r1[tx[k].ofs], (int)tx[k].a0);
        if (tx[k].a1) {
          s0 += __mul24((int)r0[tx[k].ofs + 1], (int)tx[k].a1);
          s1 += __mul24((int)r1[tx[k].ofs + 1], (int)tx[k].a1);
        }
        const int v = ((__mul24(b0, s0 >> 4) >> 16) + (__mul24(b1, s1 >> 4) >> 16) + 2) >> 2;
        o |= (uint32_t)(v & 255) << (8 * k);
      }
    }
    *reinterpret_cast<uint32_t*>(dst + (long long)(y0 + r) * D.pitch + x4) = o;
  }
}

// Fast form (every group of 4 output pixels draws its 8 taps from one 8-byte source window -- any scale factor up to 2, in
// particular the reference's 1.2): per source row a thread reads three aligned dwords of the tile, funnels them to the
// window (2 x v_alignbyte), and every output pixel costs one v_perm_b32 (its two taps as a u16 pair, selector precomputed
// per column) and one v_dot2_u32_u16 against the packed (a0, a1) coefficients per source row; the vertical pass and the
// rounding are OpenCV's fixed-point expressions unchanged.  The kernel is latency-bound (a block is a handful of dependent
// memory round trips), so the table fetches are few and wide: the block's tile origin / extent comes precomputed from the
// host, the taps of a thread's 4 columns and 4 rows are four 16-byte loads issued before the staging loop.
struct alignas(16) Tap4 {
  uint4 lo, hi;   // 4 ResizeTap entries
};
__device__ __forceinline__ void tap_unpack(const Tap4& t, int k, int& ofs, int& a0, int& a1) {
  const unsigned w0 = k == 0 ? t.lo.x : k == 1 ? t.lo.z : k == 2 ? t.hi.x : t.hi.z;
  const unsigned w1 = k == 0 ? t.lo.y : k == 1 ? t.lo.w : k == 2 ? t.hi.y : t.hi.w;
  ofs = (int)(short)(w0 & 0xffffu); a0 = (int)(short)(w0 >> 16); a1 = (int)(short)(w1 & 0xffffu);
}
__global__ void __launch_bounds__(256) k_pyr_down(PyrLaunch p) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  struct { int w, h, pitch; } S = {p.sW, p.sH, p.sPitch}, D = {p.dW, p.dH, p.dPitch};
  int bx, by, b;   // (plh_xcd_decode_tiles: the source tiles of neighbouring blocks overlap -- a frame's blocks behind one L2)
  if (!plh_xcd_decode_tiles(p.xg, bx, by, b)) return;
  const int tid = (int)threadIdx.y * 64 + (int)threadIdx.x;
  const int xb = bx * 256, yb = by * 16;
  const uint8_t* src = p.src + (long long)b * p.srcStride;
  uint8_t* dst = p.dst + (long long)b * p.dstStride;
  const int TP = p.TP;
  const int x4 = xb + (int)threadIdx.x * 4;
  const int y0 = yb + (int)threadIdx.y * PYR_ROWS;
  // every table fetch of the block, independent of each other (the tables are padded, see the host plan)
  const ResizeTap tileX = p.xt[p.xtile + bx], tileY = p.yt[p.ytile + by];
  const bool work = y0 < D.h && x4 < D.pitch;
  Tap4 tx4, ty4;
  tx4.lo = tx4.hi = ty4.lo = ty4.hi = uint4{0u, 0u, 0u, 0u};
  if (work) {
    const uint4* px = reinterpret_cast<const uint4*>(p.xt + min(x4, (D.w - 1) & ~3));
    const uint4* py = reinterpret_cast<const uint4*>(p.yt + y0);
    tx4.lo = px[0]; tx4.hi = px[1];
    ty4.lo = py[0]; ty4.hi = py[1];
  }
  // stage the source tile
  const int xBase = tileX.ofs, nd = tileX.a0, syBase = tileY.ofs, nrows = tileY.a0;
  // i / nd by multiplication (exact while i * nd < 2^18; a division by a run-time value costs ~30 VALU instructions per
  // element -- more than the rest of the staging loop)
  const unsigned rowMul = (1u << 18) / (unsigned)max(nd, 1) + 1u;
  const bool mulOk = (unsigned)(nrows * nd) * (unsigned)nd < (1u << 18);   // uniform; false only for scale factors near 2
  for (int i = tid; i < nrows * nd; i += 256) {
    const int r = mulOk ? (int)(__umul24((unsigned)i, rowMul) >> 18) : i / nd, d = i - __mul24(r, nd);
    const int xs = xBase + 4 * d;
    const uint8_t* rowp = src + (__mul24(syBase + r, S.pitch) + xs);
    const int m = (int)((size_t)rowp & 3);
    unsigned v;
    if (xs + 4 + (m ? 4 : 0) <= S.pitch) {   // the aligned dword (pair) stays inside the source row
      const unsigned* ap = reinterpret_cast<const unsigned*>(rowp - m);
      const unsigned lo = ap[0];
      v = m ? align_bytes_u(ap[1], lo, m) : lo;
    } else {
      v = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) v |= (unsigned)rowp[min(k, S.w - 1 - xs)] << (8 * k);
    }
    reinterpret_cast<unsigned*>(smem)[__mul24(r, TP >> 2) + d] = v;
  }
  __syncthreads();
  if (!work) return;
  // per column constants: window origin, byte selectors and packed coefficients of the 4 pixels
  unsigned sel[4], coef[4];
  int wx, a0, a1;
  tap_unpack(tx4, 0, wx, a0, a1);                           // source column of the window's first byte
  if (x4 >= D.w) wx = xBase;                                // an all-padding group (pitch > width): reads the tile origin, writes zeros
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int ofs;
    tap_unpack(tx4, k, ofs, a0, a1);
    const bool live = x4 + k < D.w;
    const unsigned o = live ? (unsigned)(ofs - wx) : 0u;    // 0 .. 6 (host-checked)
    sel[k] = 0x0c000c00u | o | ((o + 1u) << 16);            // bytes: tap, 0, tap + 1, 0
    coef[k] = live ? ((unsigned)(unsigned short)a0 | ((unsigned)(unsigned short)a1 << 16)) : 0u;
  }
  const int wofs = wx - xBase;                              // byte offset of the window in a tile row
  const int wd = wofs >> 2, wsh = wofs & 3;
#pragma unroll
  for (int r = 0; r < PYR_ROWS; r++) {
    if (y0 + r >= D.h) break;
    int yofs, yb0, yb1;
    tap_unpack(ty4, r, yofs, yb0, yb1);
    const int sy0 = min(max(yofs, 0), S.h - 1);
    const int sy1 = min(max(yofs + 1, 0), S.h - 1);
    const unsigned* r0 = reinterpret_cast<const unsigned*>(smem + __mul24(sy0 - syBase, TP)) + wd;
    const unsigned* r1 = reinterpret_cast<const unsigned*>(smem + __mul24(sy1 - syBase, TP)) + wd;
    const unsigned p0 = r0[0], p1 = r0[1], p2 = r0[2], q0 = r1[0], q1 = r1[1], q2 = r1[2];
    const unsigned A0 = align_bytes_u(p1, p0, wsh), B0 = align_bytes_u(p2, p1, wsh);   // window bytes 0..3 / 4..7, upper source row
    const unsigned A1 = align_bytes_u(q1, q0, wsh), B1 = align_bytes_u(q2, q1, wsh);
    const unsigned b0 = (unsigned)yb0, b1 = (unsigned)yb1;
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned s0 = plh_udot2(plh_perm(B0, A0, sel[k]), coef[k], 0u);   // S[sx] * a0 + S[sx + 1] * a1, < 2^19
      const unsigned s1 = plh_udot2(plh_perm(B1, A1, sel[k]), coef[k], 0u);
      // 11-bit coefficient x 15-bit sum: v_mul_u32_u24 (a plain 32-bit multiply is a quarter-rate instruction)
      const unsigned v = ((__umul24(b0, s0 >> 4) >> 16) + (__umul24(b1, s1 >> 4) >> 16) + 2u) >> 2;
      o |= (v & 255u) << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(dst + (__mul24(y0 + r, D.pitch) + x4)) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// FAST-9/16.  For ring differences d[i] = v - p[i]:
//   dark = max_i min(d[i..i+8]),  bright = max_i min(-d[i..i+8]),  M = max(dark, bright)
// pixel is a corner at threshold t  <=>  M > t ; cornerScore<16>() = M - 1 for any corner (the `threshold` floor inside
// cornerScore only matters for non-corners).  A pixel cannot be a dark and a bright corner at once (9 + 9 > 16), so only
// the side a pixel can be a corner on is ever measured: fast_arc_side(v, bright, p) = max_i min_j (sv + ns * p[i+j]) with
// (sv, ns) = (v, -1) for the dark side and (-v, +1) for the bright side.  The sliding 9-window minimum over the circular
// 16-ring is built from windows of 3 (v_min3), the maximum over the 16 arcs is a v_max3 tree: 16 + 16 + 16 + 8 instructions.
// ---------------------------------------------------------------------------------------------
// Round 6: the ring enters as p ^ m with m = 0 (bright side) or 255 (dark side: 255 - p), and v ^ m leaves at the end -- min / max commute
// with the shift by v, so max_i min_j (p[j] - v) = (max_i min_j p[j]) - v and max_i min_j (v - p[j]) = (max_i min_j (255 - p[j])) - (255 - v):
// sixteen v_xor_b32 (2.6 cycles per wave64 instruction on gfx950, profiles/r01_valu_issue_rate_gfx950.txt) instead of sixteen v_mad_i32_i24 (4.2).
__device__ __forceinline__ int fast_arc_side(int v, bool bright, const int p[16]) {
  int x[16], lo3[16], lo9[16];
  const int m = bright ? 0 : 255;
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = p[i] ^ m;
#pragma unroll
  for (int i = 0; i < 16; i++) lo3[i] = min(min(x[i], x[(i + 1) & 15]), x[(i + 2) & 15]);
#pragma unroll
  for (int i = 0; i < 16; i++) lo9[i] = min(min(lo3[i], lo3[(i + 3) & 15]), lo3[(i + 6) & 15]);
  int a[6];
#pragma unroll
  for (int i = 0; i < 5; i++) a[i] = max(max(lo9[3 * i], lo9[3 * i + 1]), lo9[3 * i + 2]);
  a[5] = lo9[15];
  return max(max(max(a[0], a[1]), a[2]), max(max(a[3], a[4]), a[5])) - (v ^ m);
}

// One block per (row of FAST cells, frame).  The rows of the level that the cell row covers are staged in LDS with
// aligned dword copies; then
//   1. compass quick test per pixel, class consistent (a dark corner needs a darker pixel in each of the antipodal pairs
//      (0, 8) and (4, 12), a bright corner a brighter one): byte compares straight out of the packed dwords (SDWA) whose
//      results are wave masks, combined on the scalar unit.  Passing (pixel, side) pairs are queued per wave and the arc
//      measure of that side runs on dense batches of 64; a corner stores its score byte and sets its bit in a corner bitmap;
//   2. the corner bitmap is compacted into a list (in the dead image tile) and the 3x3 NMS runs on dense batches of corners,
//      everything outside the pixel's own cell window counting as score 0 exactly like a per-cell cv::FAST call; survivors
//      set a bit in the "any" (>= minThFAST) / "hi" (>= iniThFAST) bitmaps;
//   3. one wavefront per cell, one lane per row of the cell window, emits the survivors in raster order from the bitmap
//      the cell uses: "hi" if the cell has a survivor at iniThFAST, else the minThFAST fallback (ORBextractor.cc:808-816).
// Tile column c holds level column xa + c with xa = (x0 & ~3) - 4, so pixel groups are dword aligned in the tile.
constexpr int FAST_QCAP = 320;   // per-wave queue of (pixel, side) pairs that passed the quick test (63 + 2 * 128 + slack)
#define ORB_WAVE_SYNC() PLH_WAVE_SYNC()

__device__ __forceinline__ unsigned ld_u32(const uint8_t* p) { return *reinterpret_cast<const unsigned*>(p); }
__device__ __forceinline__ unsigned align_bytes(unsigned hi, unsigned lo, int sh) {   // bytes sh..sh+3 of {hi:lo}
  return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * sh));
}

// queue entry: row (tile) : 8 | bright side : 1 | tile column : 16
__device__ __forceinline__ void fast_score_entry(const uint8_t* tile, int TP, uint8_t* sc, unsigned* cmask, int W32, unsigned e,
                                                 int tlo) {
  const int r = (int)(e >> 24), cx = (int)(e & 0xffffu);
  const bool bright = (e >> 16) & 1u;
  const uint8_t* t = tile + r * TP + cx;
  const int v = t[0];
  int p[16];
  p[0] = t[3 * TP];       p[1] = t[3 * TP + 1];   p[2] = t[2 * TP + 2];   p[3] = t[TP + 3];
  p[4] = t[3];            p[5] = t[-TP + 3];      p[6] = t[-2 * TP + 2];  p[7] = t[-3 * TP + 1];
  p[8] = t[-3 * TP];      p[9] = t[-3 * TP - 1];  p[10] = t[-2 * TP - 2]; p[11] = t[-TP - 3];
  p[12] = t[-3];          p[13] = t[TP - 3];      p[14] = t[2 * TP - 2];  p[15] = t[3 * TP - 1];
  const int M = fast_arc_side(v, bright, p);
  if (M > tlo) {
    sc[__mul24(r - 2, TP) + cx] = (uint8_t)(M - 1);   // score row rr = r - 3 is stored at sc row rr + 1
    atomicOr(&cmask[__mul24(r - 3, W32) + (cx >> 5)], 1u << (cx & 31));
  }
}

__global__ void __launch_bounds__(256) k_fast_strips(OrbDeviceArgs a, PlhXcdGrid xg) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  __shared__ unsigned s_queue[4][FAST_QCAP];
  __shared__ int s_hi[64];
  __shared__ int s_total, s_listN;

  // XCD-aware decode (plh_xcd.h): keep all strips of a frame on one XCD so the overlapping halos and the level rows are served from
  // that XCD's L2.
  int strip, b;
  if (!plh_xcd_decode(xg, strip, b)) return;

  const OrbStrip st = a.strips[strip];
  const OrbLevel lv = a.levels[st.level];
  const uint8_t* src = level_ptr(a, lv, st.level, b);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int xa = (st.x0 & ~3) - 4;                       // level column of tile column 0
  const int TP = ((st.xEnd + 4 - xa + 3) & ~3) + 4;      // tile pitch (bytes), multiple of 4
  const int W32 = (TP + 31) >> 5;                        // bitmap words per row
  const int ch = st.ch, eh = ch - 6;
  const int ex0 = st.x0 + 3, ex1 = st.xEnd - 3;          // evaluated columns [ex0, ex1), level coordinates
  const int tileBytes = (ch * TP + 15) & ~15;
  uint8_t* tile = smem;                                  // [ch][TP]
  uint8_t* sc = smem + tileBytes;                        // [eh + 2][TP], rows 0 and eh + 1 stay zero
  unsigned* cmask = reinterpret_cast<unsigned*>(sc + (((max(eh, 0) + 2) * TP + 15) & ~15));   // [eh][W32] corner bitmap
  const int tlo = min(a.iniTh, a.minTh);

  // ---- stage the rows (dword copies when the source rows are dword aligned), clear the score tile and the bitmap
  {
    const int nd = TP >> 2;
    const int xmaxd = (lv.pitch - xa - 4) >> 2;          // dwords whose aligned pair stays inside the source row
    for (int r = wv; r < ch; r += 4)                     // aligned dword loads + byte funnel for odd row addresses
     for (int d = lane; d < nd; d += 64) {
      unsigned v = 0;
      if (d < xmaxd) {
        const uint8_t* rowp = src + (long long)(st.y0 + r) * lv.pitch + xa + 4 * d;
        const int m = (int)((size_t)rowp & 3);
        const unsigned* ap = reinterpret_cast<const unsigned*>(rowp - m);
        const unsigned lo = ap[0];
        v = m ? align_bytes(ap[1], lo, m) : lo;
      }
      reinterpret_cast<unsigned*>(tile)[r * nd + d] = v;
    }
    const int ns = (((max(eh, 0) + 2) * TP + 15) & ~15) >> 2;
    for (int i = tid; i < ns + max(eh, 0) * W32; i += 256) reinterpret_cast<unsigned*>(sc)[i] = 0u;   // sc and cmask are adjacent
    if (tid < 64) s_hi[tid] = 0;
    if (tid == 0) { s_total = 0; s_listN = 0; }
  }
  __syncthreads();

  // ---- scores.  A wave owns rows wv, wv + 4, ..; its (row, pixel group) items are laid out densely over the lanes, so a
  // strip whose width is not a multiple of 256 pixels still fills the wavefront.
  const int gx0 = (ex0 - xa) >> 2, gx1 = (ex1 - 1 - xa) >> 2;   // dword columns that contain evaluated pixels
  const int ngx = (eh > 0 && ex1 > ex0) ? gx1 - gx0 + 1 : 0;
  const int nrw = eh > wv ? (eh - wv + 3) >> 2 : 0;             // rows of this wave
  const int nitems = nrw * ngx;
  const unsigned rowMul = 65536u / (unsigned)max(ngx, 1) + 1u;  // it / ngx == (it * rowMul) >> 16 for it < 16384
  unsigned* myq = s_queue[wv];
  int qn = 0;
  for (int ib = 0; ib < nitems; ib += 64) {
    const int it = ib + lane;
    const int j = (int)(__umul24((unsigned)it, rowMul) >> 16);
    const int g = it - __mul24(j, ngx);
    const bool live = it < nitems;
    const int r = live ? wv + 4 * j + 3 : 3;
    const int cx = live ? (gx0 + g) << 2 : gx0 << 2;
    const int lo = live ? ex0 - (xa + cx) : 4, hi = ex1 - (xa + cx);   // evaluated pixels of the group: lo <= k < hi
    const uint8_t* t = tile + __mul24(r, TP) + cx;
    const unsigned C = ld_u32(t), Lw = ld_u32(t - 4), R = ld_u32(t + 4), U = ld_u32(t - 3 * TP), D = ld_u32(t + 3 * TP);
    const unsigned P12 = align_bytes(C, Lw, 1), P4 = align_bytes(R, C, 3);
    const unsigned ebase = ((unsigned)r << 24) | (unsigned)cx;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int v = (int)((C >> (8 * k)) & 255);
      const int vlo = v - tlo, vhi = v + tlo;
      const int n0 = (int)((D >> (8 * k)) & 255), n8 = (int)((U >> (8 * k)) & 255);
      const int n4 = (int)((P4 >> (8 * k)) & 255), n12 = (int)((P12 >> (8 * k)) & 255);
      // every 9-arc holds one pixel of each antipodal pair: the compares are wave masks, the logic is scalar
      const unsigned long long in = wballot(k >= lo) & wballot(k < hi);
      const unsigned long long mD = (wballot(n0 < vlo) | wballot(n8 < vlo)) & (wballot(n4 < vlo) | wballot(n12 < vlo)) & in;
      const unsigned long long mB = (wballot(n0 > vhi) | wballot(n8 > vhi)) & (wballot(n4 > vhi) | wballot(n12 > vhi)) & in;
      const unsigned long long m = mD | mB;
      if (PLH_INV_BALLOT(m)) myq[qn + mbcnt64(m)] = (ebase + (unsigned)k) | (PLH_INV_BALLOT(mD) ? 0u : 0x10000u);
      qn += __popcll(m);
      const unsigned long long m2 = mD & mB;          // both sides possible (rare): the bright side as a second entry
      if (m2) {
        if (PLH_INV_BALLOT(m2)) myq[qn + mbcnt64(m2)] = (ebase + (unsigned)k) | 0x10000u;
        qn += __popcll(m2);
      }
      if (k & 1) {                                    // at most 63 + 2 * 128 entries are queued at this point
        while (qn >= 64) {
          ORB_WAVE_SYNC();
          const unsigned e = myq[qn - 64 + lane];
          ORB_WAVE_SYNC();
          fast_score_entry(tile, TP, sc, cmask, W32, e, tlo);
          qn -= 64;
        }
      }
    }
  }
  ORB_WAVE_SYNC();
  if (lane < qn) fast_score_entry(tile, TP, sc, cmask, W32, myq[lane], tlo);
  __syncthreads();

  // ---- corner bitmap -> list -> 3x3 NMS inside each cell window + threshold class (hi: >= iniThFAST, any: >= minThFAST).
  // The image tile is dead: it now holds the two survivor bitmaps and the corner list (u16: row : 6 | tile column : 10).
  unsigned* anyMask = reinterpret_cast<unsigned*>(tile);             // [eh][W32]
  unsigned* hiMask = anyMask + max(eh, 0) * W32;
  unsigned short* list = reinterpret_cast<unsigned short*>(hiMask + max(eh, 0) * W32);
  const int LCAP = (tileBytes - 8 * max(eh, 0) * W32) >> 1;          // >= one bitmap row (host-checked plan invariant)
  const int nmw = max(eh, 0) * W32;
  {
    int c = 0;
    for (int i = tid; i < nmw; i += 256) { c += __popc(cmask[i]); anyMask[i] = 0u; hiMask[i] = 0u; }
    c = wave_sum(c);
    if (lane == 0 && c) atomicAdd(&s_total, c);
  }
  __syncthreads();
  // one chunk when every corner fits the list, else chunks of as many rows as fit in the worst case
#if defined(HIPEMU)   // the emulator tests force the multi-chunk path, which real frames only take when > 40 % of the pixels are corners
  const int forcedRows = getenv("PLH_EMU_FAST_ROWS") ? atoi(getenv("PLH_EMU_FAST_ROWS")) : 0;
  const int rowsPer = forcedRows > 0 ? forcedRows : (s_total <= LCAP ? max(eh, 1) : max(LCAP / (32 * W32), 1));
#else
  const int rowsPer = s_total <= LCAP ? max(eh, 1) : max(LCAP / (32 * W32), 1);
#endif
  const int lastCell = st.nCells - 1;
  const int wCell = st.wCell;
  const unsigned cellMul = 65536u / (unsigned)wCell + 1u;            // ex / wCell == (ex * cellMul) >> 16 for ex < 4096
  for (int r0 = 0; r0 < eh; r0 += rowsPer) {
    const int r1 = min(r0 + rowsPer, eh);
    for (int i = r0 * W32 + tid; i < r1 * W32; i += 256) {
      unsigned w = cmask[i];
      if (w) {
        const int rr = i / W32, c0 = (i - rr * W32) << 5;
        int pos = atomicAdd(&s_listN, __popc(w));
        while (w) {
          const int bit = __ffs((int)w) - 1;
          w &= w - 1;
          list[pos++] = (unsigned short)((rr << 10) | (c0 + bit));
        }
      }
    }
    __syncthreads();
    const int nlist = s_listN;
    for (int i = tid; i < nlist; i += 256) {
      const unsigned e = list[i];
      const int rr = (int)(e >> 10), cxk = (int)(e & 1023u);
      const uint8_t* s1 = sc + __mul24(rr + 1, TP) + cxk;
      const int s = s1[0];
      const int ex = xa + cxk - ex0;                      // column inside the strip's evaluated window
      const int cj = min((int)(__umul24((unsigned)ex, cellMul) >> 16), lastCell);
      const int cxs = ex - __mul24(cj, wCell);            // column inside the cell's evaluated window
      const int cew = (cj == lastCell ? (ex1 - ex0) - __mul24(cj, wCell) : wCell);
      const int mL = max(max((int)s1[-1], (int)s1[-TP - 1]), (int)s1[TP - 1]);
      const int mR = max(max((int)s1[1], (int)s1[-TP + 1]), (int)s1[TP + 1]);
      int m = max((int)s1[-TP], (int)s1[TP]);             // rows outside the strip are zero
      m = max(m, max(cxs > 0 ? mL : 0, cxs < cew - 1 ? mR : 0));
      if (s > m && s >= a.minTh) {
        const unsigned bit = 1u << (cxk & 31);
        atomicOr(&anyMask[__mul24(rr, W32) + (cxk >> 5)], bit);
        if (s >= a.iniTh) { atomicOr(&hiMask[__mul24(rr, W32) + (cxk >> 5)], bit); s_hi[cj] = 1; }
      }
    }
    __syncthreads();
    if (tid == 0) s_listN = 0;
    __syncthreads();
  }

  // ---- emission: one wavefront per cell, one lane per row of the cell window, raster order
  for (int cj = wv; cj < st.nCells; cj += 4) {
    const OrbCell c = a.cells[st.cellFirst + cj];
    const int ew = c.cw - 6;
    const unsigned* mk = s_hi[cj] ? hiMask : anyMask;
    const int colBase = c.x0 + 3 - xa;
    uint32_t* out = a.slots + (long long)b * a.slotsPerFrame + c.slotOff;
    int total = 0;
    for (int row0 = 0; row0 < ((ew > 0 && eh > 0) ? eh : 0); row0 += 64) {
      const int ey = row0 + lane;
      unsigned long long bits = 0;
      if (ey < eh) {                                       // the row's window [colBase, colBase + ew), ew <= 60: three words
        const unsigned* mr = mk + ey * W32;
        const int w0 = colBase >> 5, sh = colBase & 31;
        const unsigned long long lo = (unsigned long long)mr[w0] | ((unsigned long long)(w0 + 1 < W32 ? mr[w0 + 1] : 0u) << 32);
        const unsigned long long hi = w0 + 2 < W32 ? mr[w0 + 2] : 0u;
        bits = (lo >> sh) | (sh ? hi << (64 - sh) : 0ull);
        bits &= (1ull << ew) - 1ull;
      }
      const int cnt = __popcll(bits);
      int incl = cnt;
      for (int d = 1; d < 64; d <<= 1) {
        const int tv = __shfl_up(incl, d);
        if (lane >= d) incl += tv;
      }
      int pos = total + incl - cnt;
      while (bits) {
        const int exb = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        if (pos < c.slotCap)
          out[pos] = ((uint32_t)(c.x0 + 3 + exb) << 20) | ((uint32_t)(c.y0 + 3 + ey) << 8) | sc[(ey + 1) * TP + colBase + exb];
        pos++;
      }
      total += __shfl(incl, 63);
    }
    if (lane == 0) {
      if (total > c.slotCap) { atomicOr(a.status, 1); total = c.slotCap; }
      a.cellCount[(long long)b * a.nCellsTotal + st.cellFirst + cj] = (uint32_t)total;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Quad-tree distribution.  One wavefront per (level, frame).  The reference algorithm is a
// sequential std::list manipulation whose OUTPUT ORDER is the final list order, so it is kept
// sequential: lane 0 runs the list logic on node arrays in LDS; the whole wave services the two
// data-parallel steps it asks for -- the stable 4-way key partition of DivideNode and the
// (size, creation-id) sort of the "largest first" phase.  Parallelism comes from the batch:
// batch x levels independent wavefronts.
//   list order  == descending creation id (children are always push_front'ed)
//   tie-break   == creation order (PINNED; the reference's is a heap address)
// ---------------------------------------------------------------------------------------------
enum { OCT_DONE = 0, OCT_SPLIT = 1, OCT_SORT = 2 };
enum { S_P1_BEGIN, S_P1_SCAN, S_P1_AFTER, S_P2_BEGIN, S_P2_SORTED, S_P2_NEXT, S_P2_AFTER, S_P2_END };

struct OctLds {
  short *x0, *x1, *y0, *y1, *next, *prev, *freeStack, *order;
  int *start, *cnt, *id;
  uint8_t* buf;
  int *evSize0, *evId0;   // two event lists each; list k starts at + k * evStride / + k * cap (a pointer ARRAY indexed at run time
  short* evSlot0;         // would push this whole struct into scratch memory)
  int evStride, cap;
  int* req;   // [0] op [1] slot/prevIdx [2] n
  __device__ __forceinline__ int* evSize(int k) const { return evSize0 + k * evStride; }
  __device__ __forceinline__ int* evId(int k) const { return evId0 + k * evStride; }
  __device__ __forceinline__ short* evSlot(int k) const { return evSlot0 + k * cap; }
};

__device__ __forceinline__ int key_x(uint32_t k) { return (int)(k >> 20); }
__device__ __forceinline__ int key_y(uint32_t k) { return (int)((k >> 8) & 0xfff); }
__device__ __forceinline__ int key_resp(uint32_t k) { return (int)(k & 0xff); }

__global__ void __launch_bounds__(64) k_octree(OrbDeviceArgs a, int nodeCapMax) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  const int level = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x;
  const OrbLevel lv = a.levels[level];
  const int cap = nodeCapMax;

  OctLds n;
  {
    unsigned char* p = smem;
    n.start = (int*)p;      p += 4 * cap;
    n.cnt = (int*)p;        p += 4 * cap;
    n.id = (int*)p;         p += 4 * cap;
    n.evSize0 = (int*)p; n.evId0 = (int*)p + cap; n.evStride = 2 * cap; n.cap = cap;   // [size 0][id 0][size 1][id 1]
    p += 16 * cap;
    n.req = (int*)p;        p += 4 * 16;
    n.x0 = (short*)p;       p += 2 * cap;
    n.x1 = (short*)p;       p += 2 * cap;
    n.y0 = (short*)p;       p += 2 * cap;
    n.y1 = (short*)p;       p += 2 * cap;
    n.next = (short*)p;     p += 2 * cap;
    n.prev = (short*)p;     p += 2 * cap;
    n.freeStack = (short*)p; p += 2 * cap;
    n.order = (short*)p;    p += 2 * cap;
    n.evSlot0 = (short*)p; p += 4 * cap;
    n.buf = (uint8_t*)p;
  }

  uint32_t* kbuf[2];
  kbuf[0] = a.keys + ((long long)b * 2 + 0) * a.slotsPerFrame + lv.slotOff;
  kbuf[1] = a.keys + ((long long)b * 2 + 1) * a.slotsPerFrame + lv.slotOff;
  const uint32_t* slots = a.slots + (long long)b * a.slotsPerFrame;
  const uint32_t* ccount = a.cellCount + (long long)b * a.nCellsTotal + lv.cellBase;

  // ---- gather the level's candidates in cell order (== vToDistributeKeys order) into kbuf[0] ----
  int K = 0;
  for (int cb = 0; cb < lv.nCells; cb += 64) {
    const int ci = cb + lane;
    const int cnt = ci < lv.nCells ? (int)ccount[ci] : 0;
    // inclusive scan over the wave
    int incl = cnt;
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    const int excl = incl - cnt;
    if (cnt > 0) {
      const uint32_t* s = slots + a.cells[lv.cellBase + ci].slotOff;
      for (int k = 0; k < cnt; k++) kbuf[0][K + excl + k] = s[k];
    }
    K += __shfl(incl, 63);
  }
  __syncthreads();

  const int N = lv.nFeat;
  int* selCount = a.selCount + (long long)b * a.nlevels + level;
  uint32_t* sel = a.sel + (long long)b * a.selPerFrame + lv.selOff;

  // ---- initial nodes: stable compaction of kbuf[0] by vpIniNodes[kp.pt.x / hX] into kbuf[1] ----
  // lane-0 list state (registers; only lane 0's copies are meaningful)
  int head = -1, tail = -1, lsize = 0, nextId = 0, freeTop = 0;
  if (lane == 0) {
    for (int s = 0; s < cap; s++) n.freeStack[s] = (short)(cap - 1 - s);
    freeTop = cap;
  }
  {
    int off = 0;
    for (int j = 0; j < lv.nIni; j++) {
      int cj = 0;
      for (int base = 0; base < K; base += 64) {
        const int i = base + lane;
        uint32_t key = 0;
        bool mine = false;
        if (i < K) {
          key = kbuf[0][i];
          const float xw = (float)(key_x(key) - lv.minBX);
          mine = (int)(xw / lv.hX) == j;
        }
        const unsigned long long mask = __ballot(mine);
        if (mine) kbuf[1][off + cj + __popcll(mask & lanemask_lt())] = key;
        cj += __popcll(mask);
      }
      if (lane == 0 && cj > 0) {
        const int s = n.freeStack[--freeTop];
        n.x0[s] = (short)(int)(lv.hX * (float)j);
        n.x1[s] = (short)(int)(lv.hX * (float)(j + 1));
        n.y0[s] = 0;
        n.y1[s] = (short)(lv.maxBY - lv.minBY);
        n.start[s] = off;
        n.cnt[s] = cj;
        n.buf[s] = 1;
        n.id[s] = nextId++;
        // push_back
        n.next[s] = -1;
        n.prev[s] = (short)tail;
        if (tail >= 0) n.next[tail] = (short)s; else head = s;
        tail = s;
        lsize++;
      }
      off += cj;
    }
  }
  __syncthreads();

  // ---- lane-0 state machine + wave services ----
  int state = S_P1_BEGIN, cursor = -1, savedNext = -1, prevSize = 0, nToExpand = 0, jdx = 0, evCur = 0, prevIdx = 0;
  int evN[2] = {0, 0};
  int c4[4] = {0, 0, 0, 0};   // child counts of the last split (uniform)
  int splitSlot = -1;

  for (;;) {
    if (lane == 0) {
      int op = -1;
      while (op < 0) {
        switch (state) {
          case S_P1_BEGIN:
            prevSize = lsize; nToExpand = 0; evN[evCur] = 0; cursor = head;
            state = S_P1_SCAN;
            break;
          case S_P1_SCAN:
            while (cursor >= 0 && n.cnt[cursor] == 1) cursor = n.next[cursor];
            if (cursor >= 0) {
              savedNext = n.next[cursor];
              splitSlot = cursor;
              op = OCT_SPLIT;
              state = S_P1_AFTER;
            } else if (lsize >= N || lsize == prevSize) {
              op = OCT_DONE;
            } else if (lsize + nToExpand * 3 > N) {
              state = S_P2_BEGIN;
            } else {
              state = S_P1_BEGIN;
            }
            break;
          case S_P1_AFTER:
          case S_P2_AFTER: {
            // create the non-empty children n1..n4 (push_front each), then erase the parent
            const int sp = splitSlot;
            const int px0 = n.x0[sp], px1 = n.x1[sp], py0 = n.y0[sp], py1 = n.y1[sp];
            const int hx = (px1 - px0 + 1) >> 1, hy = (py1 - py0 + 1) >> 1;   // ceil(d/2.f)
            const int db = n.buf[sp] ^ 1;
            int st = n.start[sp];
            for (int k = 0; k < 4; k++) {
              const int ck = c4[k];
              if (ck > 0) {
                const int s = n.freeStack[--freeTop];
                n.x0[s] = (short)((k & 1) ? px0 + hx : px0);
                n.x1[s] = (short)((k & 1) ? px1 : px0 + hx);
                n.y0[s] = (short)((k & 2) ? py0 + hy : py0);
                n.y1[s] = (short)((k & 2) ? py1 : py0 + hy);
                n.start[s] = st;
                n.cnt[s] = ck;
                n.buf[s] = (uint8_t)db;
                n.id[s] = nextId++;
                n.prev[s] = -1;
                n.next[s] = (short)head;
                if (head >= 0) n.prev[head] = (short)s; else tail = s;
                head = s;
                lsize++;
                if (ck > 1) {
                  if (state == S_P1_AFTER) nToExpand++;
                  const int e = evN[evCur]++;
                  n.evSize(evCur)[e] = ck;
                  n.evId(evCur)[e] = n.id[s];
                  n.evSlot(evCur)[e] = (short)s;
                }
              }
              st += ck;
            }
            {  // erase parent
              const int p = n.prev[sp], q = n.next[sp];
              if (p >= 0) n.next[p] = (short)q; else head = q;
              if (q >= 0) n.prev[q] = (short)p; else tail = p;
              lsize--;
              n.freeStack[freeTop++] = (short)sp;
            }
            if (state == S_P1_AFTER) {
              cursor = savedNext;
              state = S_P1_SCAN;
            } else if (lsize >= N) {
              state = S_P2_END;
            } else {
              jdx--;
              state = S_P2_NEXT;
            }
            break;
          }
          case S_P2_BEGIN:
            prevSize = lsize;
            prevIdx = evCur;
            evCur ^= 1;
            evN[evCur] = 0;
            op = OCT_SORT;
            state = S_P2_SORTED;
            break;
          case S_P2_SORTED:
            jdx = evN[prevIdx] - 1;
            state = S_P2_NEXT;
            break;
          case S_P2_NEXT:
            if (jdx < 0) {
              state = S_P2_END;
            } else {
              splitSlot = n.evSlot(prevIdx)[n.order[jdx]];
              op = OCT_SPLIT;
              state = S_P2_AFTER;
            }
            break;
          case S_P2_END:
            if (lsize >= N || lsize == prevSize) op = OCT_DONE;
            else state = S_P2_BEGIN;
            break;
        }
      }
      n.req[0] = op;
      if (op == OCT_SPLIT) {
        const int sp = splitSlot;
        n.req[1] = n.buf[sp];
        n.req[2] = n.start[sp];
        n.req[3] = n.cnt[sp];
        n.req[4] = n.x0[sp] + ((n.x1[sp] - n.x0[sp] + 1) >> 1);   // n1.UR.x
        n.req[5] = n.y0[sp] + ((n.y1[sp] - n.y0[sp] + 1) >> 1);   // n1.BR.y
      } else if (op == OCT_SORT) {
        n.req[1] = prevIdx;
        n.req[2] = evN[prevIdx];
      }
    }
    __syncthreads();
    const int op = n.req[0];
    if (op == OCT_DONE) break;
    if (op == OCT_SPLIT) {
      // stable 4-way partition of the node's keys into the other key buffer (DivideNode)
      const int sb = n.req[1], st = n.req[2], cnt = n.req[3], bx = n.req[4], by = n.req[5];
      const uint32_t* src = kbuf[sb] + st;
      uint32_t* dst = kbuf[sb ^ 1] + st;
      int tot[4] = {0, 0, 0, 0};
      if (cnt <= 64) {   // most splits: one load serves the count and the scatter
        int cls = -1;
        uint32_t key = 0;
        if (lane < cnt) {
          key = src[lane];
          const int xw = key_x(key) - lv.minBX, yw = key_y(key) - lv.minBY;
          cls = (xw < bx) ? (yw < by ? 0 : 2) : (yw < by ? 1 : 3);
        }
        unsigned long long m[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { m[k] = wballot(cls == k); tot[k] = __popcll(m[k]); }
        const int r1 = tot[0], r2 = r1 + tot[1], r3 = r2 + tot[2];
        if (cls >= 0) {
          const unsigned long long mm = cls == 0 ? m[0] : (cls == 1 ? m[1] : (cls == 2 ? m[2] : m[3]));
          const int rb = cls == 0 ? 0 : (cls == 1 ? r1 : (cls == 2 ? r2 : r3));
          dst[rb + __popcll(mm & lanemask_lt())] = key;
        }
      } else {
      for (int base = 0; base < cnt; base += 64) {
        const int i = base + lane;
        int cls = -1;
        if (i < cnt) {
          const uint32_t key = src[i];
          const int xw = key_x(key) - lv.minBX, yw = key_y(key) - lv.minBY;
          cls = (xw < bx) ? (yw < by ? 0 : 2) : (yw < by ? 1 : 3);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) tot[k] += __popcll(__ballot(cls == k));
      }
      int run[4];
      run[0] = 0; run[1] = tot[0]; run[2] = tot[0] + tot[1]; run[3] = tot[0] + tot[1] + tot[2];
      for (int base = 0; base < cnt; base += 64) {
        const int i = base + lane;
        int cls = -1;
        uint32_t key = 0;
        if (i < cnt) {
          key = src[i];
          const int xw = key_x(key) - lv.minBX, yw = key_y(key) - lv.minBY;
          cls = (xw < bx) ? (yw < by ? 0 : 2) : (yw < by ? 1 : 3);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned long long m = __ballot(cls == k);
          if (cls == k) dst[run[k] + __popcll(m & lanemask_lt())] = key;
          run[k] += __popcll(m);
        }
      }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) c4[k] = tot[k];
    } else {
      // rank sort of the expand vector by (size, creation id) ascending -> order[rank] = entry
      const int pi = n.req[1], cntE = n.req[2];
      for (int e = lane; e < cntE; e += 64) {
        const int sz = n.evSize(pi)[e], id = n.evId(pi)[e];
        int rank = 0;
        for (int f = 0; f < cntE; f++) {
          const int sf = n.evSize(pi)[f], idf = n.evId(pi)[f];
          rank += (sf < sz) || (sf == sz && idf < id);
        }
        n.order[rank] = (short)e;
      }
    }
    __syncthreads();
  }

  // ---- retain the best point of each leaf, in list order (ORBextractor.cc:744-760) ----
  if (lane == 0) {
    int i = 0;
    for (int s = head; s >= 0; s = n.next[s]) n.order[i++] = (short)s;
    n.req[1] = lsize;
    *selCount = min(lsize, lv.selCap);
    if (lsize > lv.selCap) atomicOr(a.status, 2);
  }
  __syncthreads();
  const int nleaf = min(n.req[1], lv.selCap);
  for (int i = lane; i < nleaf; i += 64) {
    const int s = n.order[i];
    const uint32_t* kk = kbuf[n.buf[s]] + n.start[s];
    uint32_t best = kk[0];
    const int cnt = n.cnt[s];
    for (int k = 1; k < cnt; k++) {
      const uint32_t key = kk[k];
      if (key_resp(key) > key_resp(best)) best = key;
    }
    sel[i] = best;
  }
}

// ---------------------------------------------------------------------------------------------
// Orientation + descriptor.  One wavefront per selected keypoint.  The 43x43 neighbourhood of the
// (un-blurred) level image is staged in LDS once; IC_Angle runs on its central 31x31 disc, the
// separable 7x7 sigma=2 Gaussian (Q8 integer, REFLECT_101 at the level's own edges as the
// reference blurs a border-less clone) is evaluated for the central 37x37 region only, and the
// 512 steered rBRIEF samples are gathered from that LDS tile.  No blurred pyramid is ever
// written to HBM.
//
// Everything between the staging and the gather is linear integer arithmetic on bytes / 16-bit sums, so it runs on the
// packed dot products: IC_Angle and the horizontal blur pass take four pixels per v_dot4_u32_u8 (the tap weights 18 34 49
// 55 fit a byte), the vertical pass two 16-bit row sums per v_dot2_u32_u16.  For that the row sums are stored TRANSPOSED
// (hT[column][row]) so that vertically adjacent sums share a dword, and the blurred tile comes out transposed as well
// (the gather does not care).  Work items are laid out two-dimensionally over 60 of the 64 lanes (5 rows x 12 dwords,
// 6 x 10 groups of four) so that no pass divides or multiplies for its addresses: round 2 measured 1 240 VALU
// wave-instructions per keypoint for the scalar formulation of the same sums, of which 240 were address arithmetic of the
// staging loop alone.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int nn) {
  if (p < 0) p = -p;
  if (p >= nn) p = 2 * nn - 2 - p;
  return p;
}

__global__ void __launch_bounds__(64) k_orient_brief(OrbDeviceArgs a, plh_keypoint* kps, uint8_t* desc, int* nOut,
                                                     int cap, PlhXcdGrid xg) {
  constexpr int PR = 21, PW = 43, PP = 48;   // patch radius / width / pitch (pitch 48 = 12 dwords)
  constexpr int BR = 18, BW = 37, BP = 40;   // blurred radius / width / pitch of the transposed blurred tile
  constexpr int HP = 50;                     // pitch (in u16) of the transposed row sums: rows 0..42 + the over-read of the last group (up to row 45);
                                             // 25 dwords: the ten column groups of a horizontal-pass store then fall into different LDS banks (24: one)
  // 7x7 sigma = 2 Gaussian in Q8 (ORB_GAUSS7_Q8, which the host checks against its own float evaluation at create time)
  constexpr unsigned G0 = ORB_GAUSS7_Q8[0], G1 = ORB_GAUSS7_Q8[1], G2 = ORB_GAUSS7_Q8[2], G3 = ORB_GAUSS7_Q8[3];
  __shared__ unsigned patchW[(PW * PP + 16) / 4];
  __shared__ __attribute__((aligned(16))) unsigned short hT[40 * HP];
  __shared__ unsigned blurW[BW * BP / 4];
  uint8_t* const patchBuf = reinterpret_cast<uint8_t*>(patchW);
  const uint8_t* const blurT = reinterpret_cast<const uint8_t*>(blurW);

  // XCD-aware decode (plh_xcd_decode, plh_common.h): all keypoints of a frame go to one XCD, one after the other.  The 43 x 43 patches
  // of a frame's ~1000 keypoints cover its pyramid about twice over; spread over the eight L2s as a (slot, frame) grid spreads them,
  // every keypoint fetched its 43 rows x 1 - 2 sectors from HBM (4.9 MB per frame for 1.9 MB of patches, profiles/hbm_traffic.json
  // of build b6c019a1); behind one L2 the frame's level images are fetched once (1.0 MB).
  int slot, b;
  if (!plh_xcd_decode(xg, slot, b)) return;
  const int lane = threadIdx.x;
  const int* selCount = a.selCount + (long long)b * a.nlevels;

  int level = 0, outBase = 0;
  {
    int l = 0, base = 0;
    for (; l < a.nlevels; l++) {
      const int so = a.levels[l].selOff, sc = a.levels[l].selCap;
      if (slot >= so && slot < so + sc) break;
      base += selCount[l];
    }
    level = l;
    outBase = base;
  }
  if (slot == 0 && lane == 0) {
    int tot = 0;
    for (int l = 0; l < a.nlevels; l++) tot += selCount[l];
    nOut[b] = tot;
  }
  if (level >= a.nlevels) return;
  const OrbLevel lv = a.levels[level];
  const int idx = slot - lv.selOff;
  if (idx >= selCount[level]) return;
  const int outIdx = outBase + idx;
  if (outIdx >= cap) return;

  const uint32_t key = a.sel[(long long)b * a.selPerFrame + slot];
  const int kx = key_x(key), ky = key_y(key);
  const uint8_t* img = level_ptr(a, lv, level, b);

  // ---- stage the 43x43 patch.  Interior keypoints with dword-aligned rows: 12 aligned dwords per row, the patch then
  // starts `sh` bytes into the buffer (the passes below shift by it with v_alignbyte); otherwise (reflection needed / odd
  // pitch) byte by byte.  Fast path: lane = 12 * (row in a group of five) + dword, so LDS dword index = lane + 60 * pass.
  const int xl = kx - PR;
  const int x0a = xl & ~3;
  const bool fastPath = xl >= 0 && kx + PR < lv.w && ky - PR >= 0 && ky + PR < lv.h && x0a + PP <= lv.pitch &&
                        (((size_t)img | (size_t)lv.pitch) & 3) == 0;
  const int sh = fastPath ? xl - x0a : 0;   // patch byte (r, c) = patchBuf[r * PP + sh + c] = level pixel (kx - 21 + c, ky - 21 + r)
  if (fastPath) {
    if (lane < 60) {
      const int lr = (lane * 43) >> 9, d = lane - lr * 12;   // lane / 12, lane % 12
      int off = __mul24(ky - PR + lr, lv.pitch) + x0a + 4 * d;   // 32-bit offset, 24-bit multiply
      const int step = 5 * lv.pitch;
#pragma unroll
      for (int k = 0; k < 9; k++) {
        if (k < 8 || lane < 36) patchW[lane + 60 * k] = *reinterpret_cast<const unsigned*>(img + off);   // rows 40..42 in the last pass
        off += step;
      }
    }
  } else {
    for (int i = lane; i < PW * PW; i += 64) {
      const int r = i / PW, c = i - r * PW;
      const int yy = reflect101(ky - PR + r, lv.h), xx = reflect101(kx - PR + c, lv.w);
      patchBuf[r * PP + c] = img[(long long)yy * lv.pitch + xx];
    }
  }
  __syncthreads();

  // ---- horizontal pass: rows 0..42, output columns 0..39 (37 used), lane = 10 * (row in a group of six) + column group.
  // Ten bytes q0..q9 of the row in three shifted dwords A B C; output k = sum_t G[t] q[k + t] as dot4s with the weights
  // slid along the bytes.
  if (lane < 60) {
    const int lr = (lane * 26) >> 8, g = lane - lr * 10;   // lane / 10, lane % 10
    const unsigned* pw = patchW + lr * 12 + g;
    unsigned short* hp = hT + 4 * g * HP + lr;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (k < 7 || lr == 0) {   // row lr + 6 k < 43
        const unsigned w0 = pw[72 * k], w1 = pw[72 * k + 1], w2 = pw[72 * k + 2], w3 = pw[72 * k + 3];
        const unsigned A = align_bytes_u(w1, w0, sh), B = align_bytes_u(w2, w1, sh), C = align_bytes_u(w3, w2, sh);
        const unsigned h0 = plh_udot4(A, w4(G0, G1, G2, G3), plh_udot4(B, w4(G2, G1, G0, 0), 0u));   // <= 257 * 255 < 2^16
        const unsigned h1 = plh_udot4(A, w4(0, G0, G1, G2), plh_udot4(B, w4(G3, G2, G1, G0), 0u));
        const unsigned h2 = plh_udot4(A, w4(0, 0, G0, G1), plh_udot4(B, w4(G2, G3, G2, G1), plh_udot4(C, w4(G0, 0, 0, 0), 0u)));
        const unsigned h3 = plh_udot4(A, w4(0, 0, 0, G0), plh_udot4(B, w4(G1, G2, G3, G2), plh_udot4(C, w4(G1, G0, 0, 0), 0u)));
        hp[6 * k] = (unsigned short)h0;
        hp[6 * k + HP] = (unsigned short)h1;
        hp[6 * k + 2 * HP] = (unsigned short)h2;
        hp[6 * k + 3 * HP] = (unsigned short)h3;
      }
    }
  }

  // ---- IC_Angle (reads the patch only, so it runs before the barrier the vertical pass needs): lane = 8 * (row in a
  // group of eight) + dword of the 31-byte row
  int m10, m01;
  {
    const int o = sh + (PR - 15);   // byte offset of column u = -15 inside a patch row
    const int d0 = o >> 2, s2 = o & 3;
    const unsigned* pw = patchW + ((lane >> 3) + (PR - 15)) * 12 + d0 + (lane & 7);
    const uint2* wt = reinterpret_cast<const uint2*>(c_icw.w) + lane;
    unsigned S1 = 0;
    int S0 = 0, Sv = 0;
    const int v0 = (lane >> 3) - 15;
#pragma unroll
    for (int k = 0; k < 4; k++) {   // rows 8 k + lane / 8; row 31 has zero weights
      const unsigned A = align_bytes_u(pw[96 * k + 1], pw[96 * k], s2);
      const uint2 w = wt[64 * k];
      S1 = plh_udot4(A, w.x, S1);
      const int rs = (int)plh_udot4(A, w.y, 0u);   // pixels of the row inside the disc, <= 4 * 255
      S0 += rs;
      Sv += __mul24(v0 + 8 * k, rs);
    }
    m10 = wave_sum((int)S1 - 15 * S0);
    m01 = wave_sum(Sv);
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  __syncthreads();

  // ---- vertical pass: lane = 10 * (column in a group of six) + group of four rows; five dwords = ten consecutive row sums
  // of the column; an even output row takes its taps as (t, t+1) pairs from its own dword on, an odd one from the dword
  // below with the weights slid by one.  Rounding constant in the accumulator, saturation on packed halves.
  if (lane < 60) {
    const int lc = (lane * 26) >> 8, m = lane - lc * 10;
    const unsigned* hp = reinterpret_cast<const unsigned*>(hT + lc * HP + 4 * m);
    unsigned* bo = blurW + lc * (BP / 4) + m;
#pragma unroll
    for (int k = 0; k < 7; k++) {
      if (k < 6 || lc == 0) {   // column lc + 6 k < 37
        const unsigned* p = hp + k * (6 * HP / 2);
        const unsigned P0 = p[0], P1 = p[1], P2 = p[2], P3 = p[3], P4 = p[4];
        unsigned a0 = plh_udot2(P0, w2(G0, G1), 1u << 15), a1 = plh_udot2(P0, w2(0, G0), 1u << 15);
        a0 = plh_udot2(P1, w2(G2, G3), a0); a1 = plh_udot2(P1, w2(G1, G2), a1);
        a0 = plh_udot2(P2, w2(G2, G1), a0); a1 = plh_udot2(P2, w2(G3, G2), a1);
        a0 = plh_udot2(P3, w2(G0, 0), a0);  a1 = plh_udot2(P3, w2(G1, G0), a1);
        unsigned a2 = plh_udot2(P1, w2(G0, G1), 1u << 15), a3 = plh_udot2(P1, w2(0, G0), 1u << 15);
        a2 = plh_udot2(P2, w2(G2, G3), a2); a3 = plh_udot2(P2, w2(G1, G2), a3);
        a2 = plh_udot2(P3, w2(G2, G1), a2); a3 = plh_udot2(P3, w2(G3, G2), a3);
        a2 = plh_udot2(P4, w2(G0, 0), a2);  a3 = plh_udot2(P4, w2(G1, G0), a3);
        // (a >> 16) of two results as a pair of halves, min(., 255) on both, then the four low bytes
        const unsigned p01 = plh_pk_min_u16(plh_perm(a1, a0, 0x07060302u), 0x00ff00ffu);
        const unsigned p23 = plh_pk_min_u16(plh_perm(a3, a2, 0x07060302u), 0x00ff00ffu);
        bo[k * (6 * BP / 4)] = plh_perm(p23, p01, 0x06040200u);
      }
    }
  }
  __syncthreads();

  // steered rBRIEF: lane -> 4 pairs (one nibble)
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  const float ang = angle * factorPI;
  // correctly rounded cosf / sinf of the pinned definition: double evaluation, one rounding (the short evaluation unless
  // a result sits next to a float rounding boundary, see plh_common.h)
  double sd, cd;
  sincos_0_2pi((double)ang, sd, cd);
  if (!(float_round_is_safe(cd) && float_round_is_safe(sd))) sincos((double)ang, &sd, &cd);
  const float ca = (float)cd, sa = (float)sd;
  const float* pat = c_orb_pattern_f + lane * 16;
  int nib = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float x0 = pat[k * 4 + 0], y0 = pat[k * 4 + 1];
    const float x1 = pat[k * 4 + 2], y1 = pat[k * 4 + 3];
    const int t0 = blurT[(BR + cv_round(x0 * ca - y0 * sa)) * BP + BR + cv_round(x0 * sa + y0 * ca)];   // blurT[column][row]
    const int t1 = blurT[(BR + cv_round(x1 * ca - y1 * sa)) * BP + BR + cv_round(x1 * sa + y1 * ca)];
    nib |= (t0 < t1) << k;
  }
  const int hiNib = __shfl_down(nib, 1);
  uint8_t* dptr = desc + ((long long)b * cap + outIdx) * 32;
  if ((lane & 1) == 0) dptr[lane >> 1] = (uint8_t)(nib | (hiNib << 4));

  if (lane == 0) {
    plh_keypoint o;
    o.x = (float)kx;
    o.y = (float)ky;
    if (level != 0) { o.x *= lv.scale; o.y *= lv.scale; }
    o.size = lv.kpSize;
    o.angle = angle;
    o.response = (float)key_resp(key);
    o.octave = level;
    o.class_id = -1;
    kps[(long long)b * cap + outIdx] = o;
  }
}

// ---------------------------------------------------------------------------------------------
// host-callable launchers (kept in this translation unit so the kernels stay file-local)
// ---------------------------------------------------------------------------------------------
void launch_pyr_down(const OrbDeviceArgs& a, int l, int pitch, int h, size_t lds, const PyrLaunch* fast, hipStream_t s) {
  dim3 grid((pitch / 4 + 63) / 64, (h + 4 * PYR_ROWS - 1) / (4 * PYR_ROWS), a.batch), block(64, 4);
  if (fast) {
    PyrLaunch pl = *fast;
    pl.xg = plh_xcd_make((int)grid.x, (int)grid.y, a.batch);
    hipLaunchKernelGGL(k_pyr_down, dim3(plh_xcd_grid(pl.xg)), block, lds, s, pl);
  } else {
    hipLaunchKernelGGL(k_pyr_down_gather, grid, block, lds, s, a, l);
  }
}
size_t fast_strip_lds_bytes(int width, int ch) {   // image tile + score tile + corner bitmap of k_fast_strips (width = xEnd - x0)
  const size_t TP = (size_t)((width + 8 + 3 + 3) & ~3) + 4;
  const size_t eh = ch > 6 ? ch - 6 : 0, W32 = (TP + 31) / 32;
  return (((size_t)ch * TP + 15) & ~(size_t)15) + (((eh + 2) * TP + 15) & ~(size_t)15) + eh * W32 * 4 + 64;
}
void launch_fast_strips(const OrbDeviceArgs& a, size_t lds, hipStream_t s) {
  const PlhXcdGrid xg = plh_xcd_make(a.nStrips, a.batch);
  dim3 grid(plh_xcd_grid(xg)), block(256);
  hipLaunchKernelGGL(k_fast_strips, grid, block, lds, s, a, xg);
}
size_t octree_lds_bytes(int nodeCap) { return (size_t)nodeCap * (4 * 7 + 2 * 10 + 1) + 64 + 64; }
void launch_octree(const OrbDeviceArgs& a, int nodeCapMax, hipStream_t s) {
  dim3 grid(a.nlevels, a.batch), block(64);
  hipLaunchKernelGGL(k_octree, grid, block, octree_lds_bytes(nodeCapMax), s, a, nodeCapMax);
}
void launch_orient_brief(const OrbDeviceArgs& a, plh_keypoint* kps, uint8_t* desc, int* nOut, int cap, hipStream_t s) {
  const PlhXcdGrid xg = plh_xcd_make(a.selPerFrame, a.batch);
  dim3 grid(plh_xcd_grid(xg)), block(64);
  hipLaunchKernelGGL(k_orient_brief, grid, block, 0, s, a, kps, desc, nOut, cap, xg);
}

}  // namespace plh
