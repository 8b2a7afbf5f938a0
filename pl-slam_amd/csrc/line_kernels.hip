// HIP kernels of the line half of the front end (gfx950 / CDNA4, wave64), batch-first.
//
//   k_remap_u8      cv::remap(INTER_LINEAR) with prebuilt undistortion maps     Frame.cc:220-222
//   k_blur7_u8      8-bit separable Gaussian (Q8, REFLECT_101), 5 or 7 taps     LSD internal 7x7 s=0.75; LBD 5x5 s=1
//   k_resize_u8     cv::resize(INTER_LINEAR) fixed point                        LSD internal 0.8x
//   k_lsd_grad      ll_angle(): 2x2 gradient, NOTDEF threshold, max gradient    SURVEY.md B.7
//   k_lsd_bin_*     ll_angle(): 1024-bin pseudo-ordering of seeds (stable)      SURVEY.md B.7 / 8c pin (6)
//   k_lsd_grow      flsd(): region_grow / region2rect / refine / reduce_region_radius
//   k_keylines      LSDDetector::detectImpl KeyLine fill + mask; LINEextractor sort/keep/class_id/line equation
//                                                                              LSDDetector_custom.cpp:162-213, LineExtractor.cpp:43-90
//   k_sobel_pack    cv::Sobel 3x3 dx,dy -> packed int16x2                       binary_descriptor_custom.cpp:395-396
//   k_lbd           BinaryDescriptor::computeLBD + binaryConversion             binary_descriptor_custom.cpp:1026-1372, 401-412
//
// LSD's region growing is inherently sequential per frame (global `used` map, seed order, running mean
// angle): it runs as ONE WAVEFRONT PER FRAME with the sequential semantics kept (lsd_grow.hip): 8 queue points x
// 8 neighbours per step, speculative in-order resolution, the `used` map in bit 31 of the level-line records and
// the region queue mirrored in an LDS ring.
// Throughput comes from the batch (thousands of frames = thousands of independent wavefronts).
// Float32 / float64 arithmetic follows the oracle's operation order exactly (-ffp-contract=off).
#include "line_dev.h"

namespace plh {

// ---------------------------------------------------------------------------------------------
// Block (64,4): 4 rows x 256 columns, 4 adjacent pixels per thread: the four RemapTap entries are one 32-byte run, the 16
// byte gathers are in flight together and the result leaves as one dword when the address allows.  The map is fixed per
// camera, so everything that depends on it alone -- the fixed-point split of the coordinates, the border tests -- was
// done once on the host (plh_line_set_undistort).
__global__ void __launch_bounds__(256) k_remap_u8(LineDeviceArgs a, PlhXcdGrid xg) {
  int bx, by, b;   // (plh_xcd_decode_tiles: a frame's tiles behind one L2 -- the taps of neighbouring tiles share sectors)
  if (!plh_xcd_decode_tiles(xg, bx, by, b)) return;
  const int x4 = (bx * 64 + threadIdx.x) * 4, y = by * 4 + threadIdx.y;
  if (x4 >= a.w || y >= a.h) return;
  const uint8_t* src = a.img + (long long)b * a.imgStride;
  const int pix = __mul24(y, a.w) + x4;
  const uint2* tp = reinterpret_cast<const uint2*>(a.remap) + pix;
  const int nk = min(4, a.w - x4);
  uint2 t[4];
  if (nk == 4 && (((size_t)tp) & 15) == 0) {
    const uint4 lo = reinterpret_cast<const uint4*>(tp)[0], hi = reinterpret_cast<const uint4*>(tp)[1];
    t[0].x = lo.x; t[0].y = lo.y; t[1].x = lo.z; t[1].y = lo.w;
    t[2].x = hi.x; t[2].y = hi.y; t[3].x = hi.z; t[3].y = hi.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) { t[k].x = 0; t[k].y = 0; if (k < nk) t[k] = tp[k]; }
  }
  int p[4][4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint8_t* q = src + t[k].x;
    p[k][0] = q[0]; p[k][1] = q[1]; p[k][2] = q[a.w]; p[k][3] = q[a.w + 1];
  }
  unsigned out = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int wx0 = t[k].y & 255, wx1 = (t[k].y >> 8) & 255, wy0 = (t[k].y >> 16) & 255, wy1 = t[k].y >> 24;
    const int top = __mul24(wx0, p[k][0]) + __mul24(wx1, p[k][1]), bot = __mul24(wx0, p[k][2]) + __mul24(wx1, p[k][3]);
    const int v = (__mul24(wy0, top) + __mul24(wy1, bot) + (1 << 9)) >> 10;
    out |= (unsigned)(v > 255 ? 255 : v) << (8 * k);
  }
  uint8_t* o = a.undist + (long long)b * a.fullStride + pix;
  if (nk == 4 && (((size_t)o) & 3) == 0) {
    *reinterpret_cast<unsigned*>(o) = out;
  } else {
    for (int k = 0; k < nk; k++) o[k] = (uint8_t)(out >> (8 * k));
  }
}

// Separable Q8 blur with 2R+1 taps (R = 3: general 7 taps; R = 2 when the outer taps are zero, which is the case for
// both users: LSD's sigma 0.75 and LBD's 5x5); 64x16 output tile per block, input tile (+R halo rows, REFLECT_101)
// staged in LDS with aligned dword loads (byte funnel for odd row addresses; byte-wise with reflection only in the
// image's edge columns).  Tile column j holds image column x0 - 4 + j.
//
// Both passes are integer dot products (taps < 256, tap sum <= 257, checked at create time): the horizontal pass takes
// four pixels per v_dot4_u32_u8 with the tap weights slid along three dwords of the row, the vertical pass two 16-bit
// row sums per v_dot2_u32_u16.  For the latter the row sums are stored transposed (hT[column][row], pitch 13 dwords:
// conflict-free for the 64 columns of a wavefront) so that vertically adjacent sums share a dword; a thread then owns
// one column and four rows and writes its four pixels as bytes (64 lanes = 64 consecutive bytes of a row per store).
__device__ __forceinline__ unsigned plh_udot4_l(unsigned a, unsigned b, unsigned c) { return plh_udot4(a, b, c); }
__device__ __forceinline__ unsigned plh_udot2_l(unsigned a, unsigned b, unsigned c) { return plh_udot2(a, b, c); }
// (hi_halves_sat255: bytes 2, 3 of two accumulators as two halves, saturated at 255 -- plh_shims.h)

template <int R>
__global__ void __launch_bounds__(256) k_blur7_u8(const uint8_t* src, long long sStride, int sPitch, uint8_t* dst,
                                                  long long dStride, int dPitch, int w, int h, PlhXcdGrid xg, BlurWeights bw) {
  constexpr int TW = 64, TH = 16, IH = TH + 2 * R, IP = TW + 8, IPD = IP / 4;   // 72-byte tile rows = 18 dwords
  constexpr int HP = 26;                                                        // u16 pitch of a transposed column of row sums (IH <= 22)
  __shared__ unsigned tin[IH * IPD + 1];
  __shared__ unsigned short hT[TW * HP];
  int bx, by, b;   // (plh_xcd_decode_tiles: the halo rows and columns of a tile are its neighbours' sectors -- one frame, one L2)
  if (!plh_xcd_decode_tiles(xg, bx, by, b)) return;
  const int x0 = bx * TW, y0 = by * TH, tid = threadIdx.x;
  const uint8_t* S = src + (long long)b * sStride;
  const bool interior = x0 >= 4 && x0 + TW + 8 <= w;   // the aligned dword pairs stay inside the row
  {   // thread = 18 * (row in a group of 14) + dword
    const int lr = (tid * 57) >> 10, d = tid - lr * IPD;   // tid / 18, tid % 18 for tid < 256
#pragma unroll
    for (int k = 0; k < (IH + 13) / 14; k++) {
      const int r = lr + 14 * k;
      if (tid < 14 * IPD && r < IH) {
        const uint8_t* row = S + __mul24(refl101(y0 - R + r, h), sPitch);
        unsigned v;
        if (interior) {
          const uint8_t* p = row + x0 - 4 + 4 * d;
          const int m = (int)((size_t)p & 3);
          const unsigned* ap = reinterpret_cast<const unsigned*>(p - m);
          const unsigned lo = ap[0];
          v = m ? (unsigned)(((((unsigned long long)ap[1]) << 32) | lo) >> (8 * m)) : lo;
        } else {
          v = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) v |= (unsigned)row[refl101(x0 - 4 + 4 * d + q, w)] << (8 * q);
        }
        tin[r * IPD + d] = v;
      }
    }
  }
  __syncthreads();
  {   // horizontal pass: thread = 16 * (row in a group of 16) + group of four output columns.  Output column 4g + k is tile
      // column 4g + k + 4: tap j (-R..R) multiplies byte 4 + k + j of the twelve bytes A B C
    const int lr = tid >> 4, g = tid & 15;
#pragma unroll
    for (int p = 0; p < (IH + 15) / 16; p++) {
      const int r = lr + 16 * p;
      if (r < IH) {
        const unsigned A = tin[r * IPD + g], B = tin[r * IPD + g + 1], C = tin[r * IPD + g + 2];
        unsigned short* hp = hT + 4 * g * HP + r;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          unsigned acc = plh_udot4_l(B, bw.h[k][1], 0u);
          if (4 * 0 - k - 1 >= -R) acc = plh_udot4_l(A, bw.h[k][0], acc);   // some tap of output k falls into A
          if (4 * 2 - k - 4 <= R) acc = plh_udot4_l(C, bw.h[k][2], acc);    // ... into C
          hp[k * HP] = (unsigned short)acc;   // <= 257 * 255
        }
      }
    }
  }
  __syncthreads();
  {   // vertical pass: thread = column + 64 * (group of four rows); taps as (t, t+1) pairs from the dword of the output row
      // (even rows) or the dword below with the weights slid by one (odd rows)
    const int c = tid & 63, m = tid >> 6;
    const unsigned* hp = reinterpret_cast<const unsigned*>(hT + c * HP + 4 * m);
    unsigned P[R + 2];
#pragma unroll
    for (int i = 0; i < R + 2; i++) P[i] = hp[i];
    unsigned acc[4];
#pragma unroll
    for (int o = 0; o < 4; o++) {   // output row 4 m + o: first dword P[o / 2], weights shifted by o % 2
      unsigned s = 1u << 15;
#pragma unroll
      for (int i = 0; i <= R; i++) {
        s = plh_udot2_l(P[(o >> 1) + i], bw.v[o & 1][i], s);
      }
      acc[o] = s;
    }
    const unsigned p01 = hi_halves_sat255(acc[1], acc[0]), p23 = hi_halves_sat255(acc[3], acc[2]);
    const int x = x0 + c, y = y0 + 4 * m;
    if (x < w) {
      uint8_t* o = dst + (long long)b * dStride + (__mul24(y, dPitch) + x);
      if (y < h) o[0] = (uint8_t)p01;
      if (y + 1 < h) o[dPitch] = (uint8_t)(p01 >> 16);
      if (y + 2 < h) o[2 * dPitch] = (uint8_t)p23;
      if (y + 3 < h) o[3 * dPitch] = (uint8_t)(p23 >> 16);
    }
  }
}

// cv::resize(INTER_LINEAR) fixed point, same scheme as k_pyr_down: block (64,4) = a 256 x 16 output tile whose source
// pixels (exact extent from the host tables: TP x TR) are staged in LDS with aligned dword loads (byte funnel for odd
// row addresses); every thread produces 4 pixels of 4 consecutive rows from LDS taps.
constexpr int RESIZE_ROWS = 4;
__global__ void __launch_bounds__(256) k_resize_u8(const uint8_t* src, long long sStride, int sPitch, int sw, int sh, uint8_t* dst,
                                                   long long dStride, int dPitch, int dw, int dh, const ResizeTap* xtab,
                                                   const ResizeTap* ytab, int TP) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  const int b = blockIdx.z, tid = (int)threadIdx.y * 64 + (int)threadIdx.x;
  const int xb = (int)blockIdx.x * 256, yb = (int)blockIdx.y * 16;
  const uint8_t* S = src + (long long)b * sStride;
  const int xBase = xb < dw ? (xtab[xb].ofs & ~3) : 0;
  const int xHi = xb < dw ? min((int)xtab[min(xb + 255, dw - 1)].ofs + 1, sw - 1) : -1;
  const int syBase = min(max((int)ytab[min(yb, dh - 1)].ofs, 0), sh - 1);
  const int syHi = min(max((int)ytab[min(yb + 15, dh - 1)].ofs + 1, 0), sh - 1);
  const int nd = (xHi - xBase + 4) >> 2, nrows = syHi - syBase + 1;
  const unsigned rowMul = (1u << 18) / (unsigned)max(nd, 1) + 1u;
  const bool mulOk = (unsigned)(nrows * nd) * (unsigned)nd < (1u << 18);   // uniform; false only for scale factors near 2   // i / nd by multiplication, exact while i * nd < 2^18
  for (int i = tid; i < nrows * nd; i += 256) {
    const int r = mulOk ? (int)(__umul24((unsigned)i, rowMul) >> 18) : i / nd, d = i - __mul24(r, nd);
    const int xs = xBase + 4 * d;
    const uint8_t* rowp = S + (long long)(syBase + r) * sPitch + xs;
    const int m = (int)((size_t)rowp & 3);
    unsigned v;
    if (xs + 4 + (m ? 4 : 0) <= sPitch) {   // the aligned dword (pair) stays inside the source row
      const unsigned* ap = reinterpret_cast<const unsigned*>(rowp - m);
      const unsigned lo = ap[0];
      v = m ? (unsigned)(((((unsigned long long)ap[1]) << 32) | lo) >> (8 * m)) : lo;
    } else {
      v = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) v |= (unsigned)rowp[min(k, sw - 1 - xs)] << (8 * k);
    }
    reinterpret_cast<unsigned*>(smem)[r * (TP >> 2) + d] = v;
  }
  __syncthreads();
  const int x4 = xb + (int)threadIdx.x * 4;
  const int y0 = yb + (int)threadIdx.y * RESIZE_ROWS;
  if (y0 >= dh || x4 >= dPitch) return;
  ResizeTap tx[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    tx[k].ofs = (short)xBase; tx[k].a0 = 0; tx[k].a1 = 0;
    if (x4 + k < dw) tx[k] = xtab[x4 + k];
  }
#pragma unroll
  for (int r = 0; r < RESIZE_ROWS; r++) {
    if (y0 + r >= dh) break;
    const ResizeTap ty = ytab[y0 + r];
    const int sy0 = min(max((int)ty.ofs, 0), sh - 1), sy1 = min(max((int)ty.ofs + 1, 0), sh - 1);
    const uint8_t* r0 = smem + (sy0 - syBase) * TP - xBase;
    const uint8_t* r1 = smem + (sy1 - syBase) * TP - xBase;
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (x4 + k < dw) {
        // every product is (<= 12 bits) x (<= 15 bits): 24-bit multiplies, a plain 32-bit one is a quarter-rate instruction
        int s0 = __mul24((int)r0[tx[k].ofs], (int)tx[k].a0), s1 = __mul24((int)r1[tx[k].ofs], (int)tx[k].a0);
        if (tx[k].a1) { s0 += __mul24((int)r0[tx[k].ofs + 1], (int)tx[k].a1); s1 += __mul24((int)r1[tx[k].ofs + 1], (int)tx[k].a1); }
        const int v = ((__mul24((int)ty.a0, s0 >> 4) >> 16) + (__mul24((int)ty.a1, s1 >> 4) >> 16) + 2) >> 2;
        o |= (uint32_t)(v & 255) << (8 * k);
      }
    }
    *reinterpret_cast<uint32_t*>(dst + (long long)b * dStride + (long long)(y0 + r) * dPitch + x4) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// ll_angle(): level-line field (4-byte records, see line_plan.h), padding columns cleared, per-frame max gradient.
// ---------------------------------------------------------------------------------------------
// What ll_angle() derives from one gradient (gx, gy); evaluated once per possible pair into the per-device table
// (line_plan.h), never per pixel.
__device__ __forceinline__ LsdAngleEntry lsd_angle_entry(int gx, int gy) {
  LsdAngleEntry e;
  e.angf = fast_atan2_deg((float)gx, (float)(-gy));
  // seed terms: float(cos(ad)), float(sin(ad)) of the double angle ad; region increments: float(cos(af)),
  // float(sin(af)) of af = float(ad).  One double sincos serves both: af = ad - d with |d| <= 2^-22, and
  // cos(ad - d) = c (1 - d^2/2) + s d, sin(ad - d) = s (1 - d^2/2) - c d hold to O(d^3) < 1e-19, far below the
  // double ulp the direct evaluation carries itself.
  const double ad = (double)e.angf * kDegToRads;
  const float af = (float)ad;
  const double d = ad - (double)af, h = 1.0 - 0.5 * d * d;
  double sd, cd;
  sincos_0_2pi(ad, sd, cd);
  double cf = cd * h + sd * d, sf = sd * h - cd * d;
  if (!(float_round_is_safe(cd) && float_round_is_safe(sd) && float_round_is_safe(cf) && float_round_is_safe(sf))) {
    sincos(ad, &sd, &cd);
    cf = cd * h + sd * d;
    sf = sd * h - cd * d;
  }
  e.seedx = (float)cd;
  e.seedy = (float)sd;
  e.cs = (float)cf;
  e.sn = (float)sf;
  e.pad = 0.f;
  e.modgrad = q_modgrad((unsigned)(gx * gx + gy * gy));
  return e;
}
__global__ void __launch_bounds__(256) k_lsd_angle_table(LsdAngleEntry* tab) {
  const int ix = blockIdx.x * 256 + threadIdx.x, iy = blockIdx.y;
  if (ix < LSD_ANGLE_ROWS) tab[((size_t)iy << LSD_ANGLE_PITCH_LOG2) + ix] = lsd_angle_entry(ix - LSD_GRAD_MAX, iy - LSD_GRAD_MAX);
}

// Block (64,4): 4 rows x 256 columns, 4 horizontally adjacent pixels per thread (two aligned dword loads per row, one
// 16-byte store of four records).  A record is the gradient's table index and the DEF flag (line_plan.h); nothing else
// is written per pixel.
__global__ void __launch_bounds__(256) k_lsd_grad(LineDeviceArgs a) {
  __shared__ unsigned s_max;
  const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
  if (threadIdx.x == 0 && threadIdx.y == 0) s_max = 0;
  __syncthreads();
  if (x4 < a.spitch && y < lsd_rec_rows(a.sh)) {   // (the rows that only pad the last band of blocks get NOTDEF records)
    const uint8_t* I = a.scaled + (long long)b * a.scaledStride;
    // pixels x4 .. x4+4 of rows y and y+1 (the row below the last one and the dword beyond the pitch are never used)
    unsigned long long r0 = 0, r1 = 0;
    {
      const unsigned* p0 = reinterpret_cast<const unsigned*>(I + (__mul24(y, a.spitch) + x4));
      const bool more = x4 + 4 < a.spitch;
      r0 = p0[0] | ((unsigned long long)(more ? p0[1] : 0u) << 32);
      if (y < a.sh - 1) {
        const unsigned* p1 = reinterpret_cast<const unsigned*>(I + (__mul24(y + 1, a.spitch) + x4));
        r1 = p1[0] | ((unsigned long long)(more ? p1[1] : 0u) << 32);
      }
    }
    unsigned qmax = 0;
    uint32_t rec[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int x = x4 + k;
      const int p00 = (int)((r0 >> (8 * k)) & 255), p01 = (int)((r0 >> (8 * k + 8)) & 255);
      const int p10 = (int)((r1 >> (8 * k)) & 255), p11 = (int)((r1 >> (8 * k + 8)) & 255);
      const int DA = p11 - p00, BC = p01 - p10;
      const int gx = DA + BC, gy = DA - BC;
      const unsigned q = (unsigned)(__mul24(gx, gx) + __mul24(gy, gy));
      const bool def = x < a.sw - 1 && y < a.sh - 1 && q > a.qThresh;
      rec[k] = def ? ((uint32_t)(((gy + LSD_GRAD_MAX) << LSD_ANGLE_PITCH_LOG2) + gx + LSD_GRAD_MAX) | LSD_REC_DEF) : 0u;
      if (def) qmax = max(qmax, q);
    }
    uint32_t* o = a.pix + (long long)b * a.arenaStride + lsd_rec_index((unsigned)x4, (unsigned)y, (unsigned)a.spitch);   // four pixels of a block row: 16-byte aligned
    uint4 o4;
    o4.x = rec[0]; o4.y = rec[1]; o4.z = rec[2]; o4.w = rec[3];
    *reinterpret_cast<uint4*>(o) = o4;
    if (a.advAng) {   // LSD_REFINE_ADV: the level-line angle itself (what the table holds for the pixel's gradient; NOTDEF = -1024)
      float an[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int gx = (int)(rec[k] & 1023u) - LSD_GRAD_MAX, gy = (int)((rec[k] >> LSD_ANGLE_PITCH_LOG2) & 1023u) - LSD_GRAD_MAX;
        an[k] = (rec[k] & LSD_REC_DEF) ? fast_atan2_deg((float)gx, (float)(-gy)) : -1024.f;
      }
#if defined(PLH_ANG_TILED)   // (lsd_rect_dev.h rc_row / rc_at: the plane is made of the record plane's 4 x 4 blocks)
      float* ao = a.advAng + (long long)b * a.scaledStride + lsd_rec_index((unsigned)x4, (unsigned)y, (unsigned)a.spitch);
#else
      float* ao = a.advAng + (long long)b * a.scaledStride + (__mul24(y, a.spitch) + x4);
#endif
      *reinterpret_cast<uint4*>(ao) = uint4{__float_as_uint(an[0]), __float_as_uint(an[1]), __float_as_uint(an[2]), __float_as_uint(an[3])};
    }
    if (qmax) atomicMax(&s_max, qmax);
  }
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && s_max > 0) atomicMax(&a.qmax[b], s_max);
}

// cv::pyrDown(8U): 5 x 5 kernel [1 4 6 4 1] x [1 4 6 4 1] centred on source pixel (2x, 2y), REFLECT_101, (v + 128) >> 8
// (LSDDetector::computeGaussianPyramid / BinaryDescriptor::computeGaussianPyramid with numOctaves = 2; oracle/img_ops.cc
// plo_pyr_down_u8).  One thread per output pixel: the second octave is a configuration no shipped YAML uses.
__global__ void __launch_bounds__(256) k_pyr_down5_u8(const uint8_t* src, long long sStride, int sw, int sh, uint8_t* dst, long long dStride,
                                                      int dw, int dh) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (x >= dw || y >= dh) return;
  const uint8_t* S = src + (long long)b * sStride;
  const int k5[5] = {1, 4, 6, 4, 1};
  int xs[5];
#pragma unroll
  for (int k = 0; k < 5; k++) xs[k] = refl101(2 * x + k - 2, sw);
  int v = 0;
#pragma unroll
  for (int ky = 0; ky < 5; ky++) {
    const uint8_t* row = S + (long long)refl101(2 * y + ky - 2, sh) * sw;
    int r = 0;
#pragma unroll
    for (int kx = 0; kx < 5; kx++) r += k5[kx] * (int)row[xs[kx]];
    v += k5[ky] * r;
  }
  dst[(long long)b * dStride + (long long)y * dw + x] = (uint8_t)((v + 128) >> 8);
}
void launch_pyr_down5(const uint8_t* src, long long sStride, int sw, int sh, uint8_t* dst, long long dStride, int dw, int dh, int batch,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_pyr_down5_u8, dim3((dw + 63) / 64, (dh + 3) / 4, batch), dim3(256), 0, s, src, sStride, sw, sh, dst, dStride, dw, dh);
}

// (plh_sbfe1, plh_sqrt_approx: plh_shims.h)

// Seed ordering: stable counting sort of the DEFINED pixels of a frame by bin (descending), raster order inside a bin.
// A frame is cut into LSD_ORDER_CHUNKS contiguous raster chunks; the sort is four launches of small blocks
//   k_lsd_bin_thresholds   256 threads: the frame's 1024 exact bin thresholds
//   k_lsd_bin_hist         one wavefront per (chunk, frame): bins of the chunk's pixels + the chunk's histogram
//   k_lsd_bin_scan         256 threads per frame: offsets of every (bin, chunk) in the output list
//   k_lsd_bin_scatter      one wavefront per (chunk, frame): ranks inside the chunk, scatter
// with the per-(frame, chunk, bin) counts handed over in global memory (64 KB per frame).  Round 2 started with ONE
// 1024-thread block per frame holding all 16 histograms in 72 KB of LDS: 2.8 ms per 1536 frames alone, but 40 - 90 ms inside
// the pipeline (`profiles/r02_pipeline_timeline_before_order_split.txt`) -- a block that needs four wave slots on every SIMD
// of a CU and half its LDS at once waits for the region-growing wavefronts of the other sub-batches to drain, and the
// kernel sits on the line chain's critical path in front of k_lsd_grow, so only two of the four sub-batches were ever
// growing at the same time.  Single-wavefront blocks with 4 - 8 KB of LDS start wherever a wave slot frees up.
//
// The bin of a pixel is (int)(sqrt(q / 4.0) * bin_coef) in double -- a correctly rounded f64 square root per pixel (about
// 20 instructions).  It is monotone in q, so it is fixed by the 1024 thresholds binLo[k] = smallest q whose bin is >= k,
// found once per frame with the exact expression (analytic guess 4 (k / coef)^2, then stepped until exact).  Per pixel
// a float estimate (within 2e-4 of the real-valued product) names the bin up to one, and two compares against the
// thresholds decide it.
constexpr int LSD_ORDER_CHUNKS = 16;
constexpr int LSD_ORDER_WORK = LSD_ORDER_CHUNKS * LSD_NBINS + LSD_NBINS + 8;   // u32 per frame: counts / offsets, then thresholds (+ pad)
// pixels per chunk: whole bands of four rows (the record plane is made of 4 x 4 blocks: k_lsd_bin_hist reads whole sectors), a
// multiple of 64 because the pitch is
__device__ __forceinline__ int lsd_order_chunk(int spitch, int sh) { return ((((sh + LSD_ORDER_CHUNKS - 1) / LSD_ORDER_CHUNKS) + 3) & ~3) * spitch; }
__device__ __forceinline__ double lsd_bin_coef(unsigned qmax) {
  return qmax > 0 ? (double)(LSD_NBINS - 1) / sqrt((double)(int)qmax / 4.0) : 0.0;
}

__global__ void __launch_bounds__(256) k_lsd_bin_thresholds(LineDeviceArgs a) {
  const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;   // bin
  uint32_t* binLo = a.orderWork + (long long)b * a.arenaStride + LSD_ORDER_CHUNKS * LSD_NBINS;
  const double bin_coef = lsd_bin_coef(a.qmax[b]);
  auto exact_bin = [&](unsigned q) -> int { return (int)(q_modgrad(q) * bin_coef); };
  unsigned t = k > 0 ? 0xffffffffu : 0u;   // no defined pixel (bin_coef == 0): everything is bin 0
  if (k > 0 && bin_coef > 0.0) {
    const double r = (double)k / bin_coef;
    const double g = 4.0 * r * r;
    t = g < 4.0e9 ? (unsigned)g : 4000000000u;
    while (t > 0u && exact_bin(t - 1u) >= k) t--;
    while (t < 4000000000u && exact_bin(t) < k) t++;
  }
  binLo[k] = t;
  if (k == 0) binLo[LSD_NBINS] = 0xffffffffu;
}

__global__ void __launch_bounds__(64) k_lsd_bin_hist(LineDeviceArgs a) {
  __shared__ unsigned binLo[LSD_NBINS + 8];
  __shared__ int hist[LSD_NBINS];
  const int b = blockIdx.y, lane = threadIdx.x;
  // (a block takes the chunks blockIdx.x, blockIdx.x + gridDim.x, ...: launch_lsd_order picks the blocks per frame by batch size)
  for (int wv = blockIdx.x; wv < LSD_ORDER_CHUNKS; wv += gridDim.x) {
  const uint32_t* Q = a.pix + (long long)b * a.arenaStride;      // level-line records (k_lsd_grad)
  uint16_t* BIN = reinterpret_cast<uint16_t*>(a.reg + (long long)b * a.arenaStride);   // bin + 1 per pixel, 0 = NOTDEF (scratch; 16 bits:
                                                                                       // half the bytes between this kernel and the scatter)
  uint32_t* work = a.orderWork + (long long)b * a.arenaStride;
  const int chunk = lsd_order_chunk(a.spitch, a.sh);
  const int c0 = wv * chunk, c1 = min(a.spitch * lsd_rec_rows(a.sh), c0 + chunk);   // whole bands (rows beyond sh are skipped below)
  for (int i = lane; i < LSD_NBINS + 1; i += 64) binLo[i] = work[LSD_ORDER_CHUNKS * LSD_NBINS + i];
  for (int i = lane; i < LSD_NBINS; i += 64) hist[i] = 0;
  const float coefF = (float)(lsd_bin_coef(a.qmax[b]) * 0.5);   // sqrt(q / 4) = sqrt(q) / 2
  __syncthreads();
  // order is irrelevant here: a wavefront takes 64 columns of a band of four rows at a time -- 16 whole blocks of the record
  // plane, 1 KiB of consecutive sectors; lane = block * 4 + row reads the 16 bytes of its block row and writes the four bins in
  // raster order (the scatter's order)
  for (int base = c0; base < c1; base += 256) {   // 256 pixels = 64 columns x 4 rows of the band
    const int band = base / (4 * a.spitch), x0 = (base - band * 4 * a.spitch) >> 2;   // (the chunk starts on a band; spitch % 64 == 0)
    const int iy = band * 4 + (lane & 3), ix = x0 + (lane >> 2) * 4;
    const int i = iy * a.spitch + ix;
    if (iy < a.sh) {
      const uint4 q4 = *reinterpret_cast<const uint4*>(Q + lsd_rec_index((unsigned)ix, (unsigned)iy, (unsigned)a.spitch));
      const unsigned qq[4] = {q4.x, q4.y, q4.z, q4.w};
      unsigned bb[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        bb[k] = 0;
        if (qq[k] & LSD_REC_DEF) {
          const unsigned q = lsd_rec_q(qq[k]);
          const int est = min((int)(plh_sqrt_approx((float)q) * coefF), LSD_NBINS - 1);
          const int bin = est - (q < binLo[est] ? 1 : 0) + (q >= binLo[est + 1] ? 1 : 0);
          atomicAdd(&hist[bin], 1);
          bb[k] = (unsigned)bin + 1u;
        }
      }
      uint2 o2;
      o2.x = bb[0] | (bb[1] << 16); o2.y = bb[2] | (bb[3] << 16);
      *reinterpret_cast<uint2*>(BIN + i) = o2;
    }
  }
  __syncthreads();
  for (int i = lane; i < LSD_NBINS; i += 64) work[wv * LSD_NBINS + i] = (uint32_t)hist[i];
  __syncthreads();
  }
}

// counts[chunk][bin] -> offsets[chunk][bin] in place: bins descending, chunks ascending inside a bin.  Thread t owns the
// list positions of bins 1023 - 4t .. 1020 - 4t.
__global__ void __launch_bounds__(256) k_lsd_bin_scan(LineDeviceArgs a) {
  __shared__ int part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  uint32_t* work = a.orderWork + (long long)b * a.arenaStride;
  const int bin0 = LSD_NBINS - 4 - 4 * tid;   // lowest of the thread's four bins
  uint4 cnt[LSD_ORDER_CHUNKS];
  int tot = 0;
#pragma unroll
  for (int w = 0; w < LSD_ORDER_CHUNKS; w++) {
    cnt[w] = *reinterpret_cast<const uint4*>(work + w * LSD_NBINS + bin0);
    tot += (int)(cnt[w].x + cnt[w].y + cnt[w].z + cnt[w].w);
  }
  part[tid] = tot;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - tot;
  uint4 off[LSD_ORDER_CHUNKS];
  // descending bins: .w (bin0 + 3) first
#pragma unroll
  for (int w = 0; w < LSD_ORDER_CHUNKS; w++) { off[w].w = (unsigned)run; run += (int)cnt[w].w; }
#pragma unroll
  for (int w = 0; w < LSD_ORDER_CHUNKS; w++) { off[w].z = (unsigned)run; run += (int)cnt[w].z; }
#pragma unroll
  for (int w = 0; w < LSD_ORDER_CHUNKS; w++) { off[w].y = (unsigned)run; run += (int)cnt[w].y; }
#pragma unroll
  for (int w = 0; w < LSD_ORDER_CHUNKS; w++) { off[w].x = (unsigned)run; run += (int)cnt[w].x; }
#pragma unroll
  for (int w = 0; w < LSD_ORDER_CHUNKS; w++) *reinterpret_cast<uint4*>(work + w * LSD_NBINS + bin0) = off[w];
  if (tid == 255) a.nOrdered[b] = part[255];
}

__global__ void __launch_bounds__(64) k_lsd_bin_scatter(LineDeviceArgs a) {
  __shared__ int cur[LSD_NBINS];   // next list position of every bin for this chunk
  const int b = blockIdx.y, lane = threadIdx.x;
  for (int wv = blockIdx.x; wv < LSD_ORDER_CHUNKS; wv += gridDim.x) {
  const uint16_t* BIN = reinterpret_cast<const uint16_t*>(a.reg + (long long)b * a.arenaStride);
  uint32_t* ord = a.ordered + (long long)b * a.arenaStride;
  const uint32_t* work = a.orderWork + (long long)b * a.arenaStride;
  const int npix = a.spitch * a.sh;
  const int chunk = lsd_order_chunk(a.spitch, a.sh);
  const int c0 = wv * chunk, c1 = min(npix, c0 + chunk);
  if (c0 >= c1) break;   // (chunks are in raster order: nothing behind an empty one)
  for (int i = lane; i < LSD_NBINS; i += 64) cur[i] = (int)work[wv * LSD_NBINS + i];
  __syncthreads();
  const unsigned long long lt = lanemask_lt();
  // lane order = raster order inside a 64-pixel group; the next group's bins are requested before this one is ranked.
  // A 64-pixel group never straddles a row (pitch and chunk bounds are multiples of 64): its row and first column are
  // uniform and advance without divisions
  int gy = c0 / a.spitch, gx = c0 - gy * a.spitch;
  auto rank_group = [&](unsigned bp1) {
    const uint32_t coord = (uint32_t)(gx + lane) | ((uint32_t)gy << 16);
    gx += 64;
    if (gx >= a.spitch) { gx = 0; gy++; }
    const unsigned bin = bp1 - 1u;
    const unsigned long long act = wballot(bp1 != 0u);
    if (!act) return;
    // lanes that differ from this lane in some bin bit: (ballot of bit k) xor (own bit k, spread over the word), or-ed
    // over the ten bits -- compare results are used as the masks they are, the rest is 32-bit logic
    unsigned dlo = 0, dhi = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
      const int mineI = plh_sbfe1(bin, k);            // bit k spread over the word: one v_bfe_i32
      const unsigned long long m = wballot(mineI < 0);
      const unsigned mine = (unsigned)mineI;
      dlo |= (unsigned)m ^ mine;
      dhi |= (unsigned)(m >> 32) ^ mine;
    }
    const unsigned long long same = ~(((unsigned long long)dhi << 32) | dlo) & act;
    const bool active = bp1 != 0u;
    PLH_WAVE_SYNC();
    int basep = 0;
    if (active) basep = cur[bin];
    PLH_WAVE_SYNC();
    if (active) {
      const int rank = __popcll(same & lt);
      if (rank == 0) cur[bin] = basep + __popcll(same);
      ord[basep + rank] = coord;
    }
  };
  // Bins are fetched two groups ahead into three named registers that take turns (no copies): memory operations
  // retire in order, so a load can only be waited for without also waiting for the scattered stores of the groups in
  // between if younger operations have been issued behind it.  Loads are unconditional (index clamped, value masked
  // afterwards) so that they sit in straight-line code.
  const int last = npix - 1;
  auto fetch = [&](int g0) { return (unsigned)BIN[min(g0 + lane, last)]; };
  auto mask = [&](unsigned v, int g0) { return g0 + lane < c1 ? v : 0u; };
  unsigned b0 = fetch(c0), b1 = fetch(c0 + 64), b2;
  for (int base = c0; base < c1; base += 192) {
    b2 = fetch(base + 128);
    rank_group(mask(b0, base));
    if (base + 64 >= c1) break;
    b0 = fetch(base + 192);
    rank_group(mask(b1, base + 64));
    if (base + 128 >= c1) break;
    b1 = fetch(base + 256);
    rank_group(mask(b2, base + 128));
  }
  __syncthreads();   // (the next chunk reloads `cur`)
  }
}

// ---------------------------------------------------------------------------------------------
// host-callable launchers (stage 1: image preparation + level-line field + seed ordering)
// ---------------------------------------------------------------------------------------------
void launch_remap(const LineDeviceArgs& a, hipStream_t s) {
  const PlhXcdGrid xg = plh_xcd_make((a.w + 255) / 256, (a.h + 3) / 4, a.batch);
  hipLaunchKernelGGL(k_remap_u8, dim3(plh_xcd_grid(xg)), dim3(64, 4), 0, s, a, xg);
}
void launch_blur7(const uint8_t* src, long long sStride, int sPitch, uint8_t* dst, long long dStride, int dPitch, int w, int h,
                  int batch, const int taps[7], hipStream_t s) {
  const int R = (taps[0] == 0 && taps[6] == 0) ? 2 : 3;   // integer sums: dropping zero taps changes nothing
  BlurWeights bw;
  for (int k = 0; k < 4; k++)
    for (int part = 0; part < 3; part++) {
      unsigned v = 0;
      for (int i = 0; i < 4; i++) {
        const int j = i + 4 * part - k - 4;
        if (j >= -3 && j <= 3) v |= (unsigned)taps[3 + j] << (8 * i);
      }
      bw.h[k][part] = v;
    }
  auto tap = [&](int i) -> unsigned { return (i >= 0 && i <= 2 * R) ? (unsigned)taps[3 - R + i] : 0u; };
  for (int odd = 0; odd < 2; odd++)
    for (int i = 0; i < 4; i++) bw.v[odd][i] = tap(2 * i - odd) | (tap(2 * i - odd + 1) << 16);
  const PlhXcdGrid xg = plh_xcd_make((w + 63) / 64, (h + 15) / 16, batch);   // (64 x 16 = the kernel's TW x TH)
  const dim3 grid(plh_xcd_grid(xg));
  if (R == 2)
    hipLaunchKernelGGL(k_blur7_u8<2>, grid, dim3(256), 0, s, src, sStride, sPitch, dst, dStride, dPitch, w, h, xg, bw);
  else
    hipLaunchKernelGGL(k_blur7_u8<3>, grid, dim3(256), 0, s, src, sStride, sPitch, dst, dStride, dPitch, w, h, xg, bw);
}
void launch_resize(const uint8_t* src, long long sStride, int sPitch, int sw, int sh, uint8_t* dst, long long dStride, int dPitch,
                   int dw, int dh, int batch, const ResizeTap* xtab, const ResizeTap* ytab, int tileTP, int tileTR, hipStream_t s) {
  hipLaunchKernelGGL(k_resize_u8, dim3((dPitch / 4 + 63) / 64, (dh + 4 * RESIZE_ROWS - 1) / (4 * RESIZE_ROWS), batch), dim3(64, 4),
                     (size_t)tileTP * tileTR, s, src, sStride, sPitch, sw, sh, dst, dStride, dPitch, dw, dh, xtab, ytab, tileTP);
}
void launch_lsd_angle_table(LsdAngleEntry* tab, hipStream_t s) {
#if defined(HIPEMU)
  // the emulator runs a fiber per thread; the table is a plain loop over the same function there
  for (int iy = 0; iy < LSD_ANGLE_ROWS; iy++)
    for (int ix = 0; ix < LSD_ANGLE_ROWS; ix++)
      tab[((size_t)iy << LSD_ANGLE_PITCH_LOG2) + ix] = lsd_angle_entry(ix - LSD_GRAD_MAX, iy - LSD_GRAD_MAX);
  (void)s;
#else
  hipLaunchKernelGGL(k_lsd_angle_table, dim3((LSD_ANGLE_ROWS + 255) / 256, LSD_ANGLE_ROWS), dim3(256), 0, s, tab);
#endif
}
void launch_lsd_grad(const LineDeviceArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_lsd_grad, dim3((a.spitch + 255) / 256, (a.sh + 3) / 4, a.batch), dim3(64, 4), 0, s, a);
}
size_t lsd_order_work_u32() { return (size_t)LSD_ORDER_WORK; }
int lsd_blocks_per_frame(int batch, int lo, int hi);   // lsd_rects.hip
void launch_lsd_order(const LineDeviceArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_lsd_bin_thresholds, dim3(LSD_NBINS / 256, a.batch), dim3(256), 0, s, a);
  const int groups = lsd_blocks_per_frame(a.batch, 4, LSD_ORDER_CHUNKS);   // (lsd_rects.hip: few fat blocks per frame at large batches)
  hipLaunchKernelGGL(k_lsd_bin_hist, dim3(groups, a.batch), dim3(64), 0, s, a);
  hipLaunchKernelGGL(k_lsd_bin_scan, dim3(a.batch), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_lsd_bin_scatter, dim3(groups, a.batch), dim3(64), 0, s, a);
}

}  // namespace plh

