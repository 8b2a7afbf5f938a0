// Frame / map post-processing either side of the matching path, batch-first:
//   k_undistort_kps   Frame::UndistortKeyPoints = cv::undistortPoints(pts, pts, K, D, noArray(), K)   reference src/Frame.cc:915-945
//   k_distinctive     MapPoint::ComputeDistinctiveDescriptors / MapLine twin                          reference src/MapPoint.cc:249-314,
//                                                                                                    src/MapLine.cpp:256-330
// Both are exact: the iteration runs in double with the reference's operation order (no FMA contraction), the median
// selection is integer.
#include "plh_common.h"
#include "plh_stage.h"

namespace plh {

struct UndistArgs {
  double fx, fy, cx, cy, k1, k2, p1, p2, k3;
  int enabled;
};

// One thread per keypoint.  cvUndistortPoints (OpenCV 3.2): normalise with the reciprocal focal lengths, 5 iterations
// of the inverse Brown model, re-project with P = K.
__global__ void __launch_bounds__(256) k_undistort_kps(const plh_keypoint* in, const int* nArr, int cap, UndistArgs u,
                                                       plh_keypoint* out) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= min(nArr[b], cap)) return;
  plh_keypoint kp = in[(long long)b * cap + i];
  if (u.enabled) {
    const double ifx = 1. / u.fx, ify = 1. / u.fy;
    double x = kp.x, y = kp.y;
    const double x0 = x = (x - u.cx) * ifx, y0 = y = (y - u.cy) * ify;
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((u.k3 * r2 + u.k2) * r2 + u.k1) * r2);
      const double deltaX = 2 * u.p1 * x * y + u.p2 * (r2 + 2 * x * x);
      const double deltaY = u.p1 * (r2 + 2 * y * y) + 2 * u.p2 * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = u.fx * x + 0 * y + u.cx, yy = 0 * x + u.fy * y + u.cy, ww = 1. / (0 * x + 0 * y + 1);
    kp.x = (float)(xx * ww);
    kp.y = (float)(yy * ww);
  }
  out[(long long)b * cap + i] = kp;
}

// One wavefront per set (map point / map line).  Row i's distances live one per lane (R registers for N <= 64 R);
// the median (the element of rank floor(0.5 (N-1))) is found by bisection on the distance value with ballot counts.
constexpr int DIST_R = 16;   // up to 1024 observations per set

__global__ void __launch_bounds__(64) k_distinctive(const uint8_t* desc, const int* offsets, int nsets, int* best) {
  const int s = blockIdx.x, lane = threadIdx.x;
  if (s >= nsets) return;
  const int o = offsets[s], n = min(offsets[s + 1] - o, 64 * DIST_R);
  if (n <= 0) { if (lane == 0) best[s] = -1; return; }
  const unsigned long long* D = reinterpret_cast<const unsigned long long*>(desc + (long long)o * 32);
  const int k = (int)(0.5 * (double)(n - 1));   // vDists[0.5*(N-1)]
  const int R = (n + 63) >> 6;
  int bestMedian = 0x7fffffff, bestIdx = 0;
  for (int i = 0; i < n; i++) {
    const unsigned long long a0 = D[i * 4], a1 = D[i * 4 + 1], a2 = D[i * 4 + 2], a3 = D[i * 4 + 3];
    int d[DIST_R];
#pragma unroll
    for (int r = 0; r < DIST_R; r++) {
      const int j = r * 64 + lane;
      d[r] = 0x7fff;   // beyond the set: never counted
      if (r < R && j < n)
        d[r] = __popcll(a0 ^ D[j * 4]) + __popcll(a1 ^ D[j * 4 + 1]) + __popcll(a2 ^ D[j * 4 + 2]) + __popcll(a3 ^ D[j * 4 + 3]);
    }
    int lo = 0, hi = 256;   // smallest v with #(d <= v) >= k + 1
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      int c = 0;
#pragma unroll
      for (int r = 0; r < DIST_R; r++)
        if (r < R) c += __popcll(__ballot(d[r] <= mid));
      if (c >= k + 1) hi = mid; else lo = mid + 1;
    }
    if (lo < bestMedian) { bestMedian = lo; bestIdx = i; }
  }
  if (lane == 0) best[s] = bestIdx;
}

// Frame::isInFrustum for map points / map lines (reference src/Frame.cc:560-623, 625-711) with PredictScale
// (src/MapPoint.cc:413-428, src/MapLine.cpp:395-404).  One thread per map element, one pose per frame.  The cv::Mat
// arithmetic follows the pinned definition of oracle/plo.h: `mRcw*P+mtcw` = one gemm with double accumulation and a single
// rounding, cv::norm / Mat::dot accumulate in double, the rest is the float expression as written (no FMA contraction).
__device__ __forceinline__ void to_camera(const plh_frame_view& v, const float* P, float* Pc) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double s = (double)v.Rcw[i * 3] * (double)P[0];
    s += (double)v.Rcw[i * 3 + 1] * (double)P[1];
    s += (double)v.Rcw[i * 3 + 2] * (double)P[2];
    Pc[i] = (float)(s + (double)v.tcw[i]);
  }
}
__device__ __forceinline__ float norm3(const float* a) {
  double s = (double)a[0] * (double)a[0];
  s += (double)a[1] * (double)a[1];
  s += (double)a[2] * (double)a[2];
  return (float)sqrt(s);
}
__device__ __forceinline__ double dot3(const float* a, const float* b) {
  double s = (double)a[0] * (double)b[0];
  s += (double)a[1] * (double)b[1];
  s += (double)a[2] * (double)b[2];
  return s;
}
// ceil(log(ratio) / logScaleFactor) with log = the float overload; the double logarithm rounded to float is logf
// wherever logf is correctly rounded
__device__ __forceinline__ int predict_scale(float maxDist, float dist, float logScale) {
  const float ratio = maxDist / dist;
  return (int)ceilf((float)log((double)ratio) / logScale);
}

// The projection in front of the pose-driven searches, three spellings (see plh_frame_project_points_batch_dev).
__global__ void __launch_bounds__(256) k_project_points(const plh_frame_view* views, const int* nArr, int qcap, const float* pos, int form,
                                                        uint8_t* front, float* uv) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= qcap) return;
  const long long o = (long long)b * qcap + i;
  uint8_t fr = 0;
  float u = 0.f, w = 0.f;
  if (i < nArr[b]) {
    const plh_frame_view v = views[b];
    const float P[3] = {pos[o * 3], pos[o * 3 + 1], pos[o * 3 + 2]};
    float Pc[3];
    to_camera(v, P, Pc);
    if (form == 0) {
      const float invzc = (float)(1.0 / (double)Pc[2]);
      fr = invzc < 0 ? 0 : 1;
      u = v.fx * Pc[0] * invzc + v.cx;
      w = v.fy * Pc[1] * invzc + v.cy;
    } else {
      fr = Pc[2] < 0.0f ? 0 : 1;
      const float invz = form == 2 ? (float)(1.0 / (double)Pc[2]) : 1.0f / Pc[2];
      const float x = Pc[0] * invz;
      const float y = Pc[1] * invz;
      u = v.fx * x + v.cx;
      w = v.fy * y + v.cy;
    }
  }
  front[o] = fr; uv[o * 2] = u; uv[o * 2 + 1] = w;
}

// The gates the back end's pose-driven searches apply to every map point between the transform and the window lookup
// (plh_map_point_gates): the loops of ORBmatcher.cc:1591-1640 (relocalisation), :337-395 (loop closing), :945-975 / :1096-1128 (both
// Fuse overloads), :1206-1290 / :1313-1365 (SearchBySim3, both directions), one thread per map point.
__global__ void __launch_bounds__(256) k_map_point_gates(plh_point_gates g, int n, const float* pos, const float* normal,
                                                         const float* minInv, const float* maxInv, const float* maxRaw, uint8_t* valid,
                                                         float* uv, float* distOut, int* level) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint8_t ok = 0;
  float u = 0.f, w = 0.f, dOut = 0.f;
  int lvl = 0;
  if (valid[i]) {
    const plh_frame_view& v = g.view;
    const float P[3] = {pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]};
    float Pc[3];
    to_camera(v, P, Pc);
    if (g.flags & PLH_GATE_SECOND) {   // pTarget = sR21 * p3Dc1 + t21: a second gemm on the rounded camera point
      float Q[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        double s = (double)g.R2[r * 3] * (double)Pc[0];
        s += (double)g.R2[r * 3 + 1] * (double)Pc[1];
        s += (double)g.R2[r * 3 + 2] * (double)Pc[2];
        Q[r] = (float)(s + (double)g.t2[r]);
      }
      Pc[0] = Q[0]; Pc[1] = Q[1]; Pc[2] = Q[2];
    }
    bool in = !((g.flags & PLH_GATE_Z) && Pc[2] < 0.0f);
    if (in) {
      const float invz = (g.flags & PLH_GATE_INVZ_DOUBLE) ? (float)(1.0 / (double)Pc[2]) : 1.0f / Pc[2];
      if (g.flags & PLH_GATE_UV_NORMALISED) {
        const float x = Pc[0] * invz;
        const float y = Pc[1] * invz;
        u = v.fx * x + v.cx;
        w = v.fy * y + v.cy;
      } else {
        u = v.fx * Pc[0] * invz + v.cx;
        w = v.fy * Pc[1] * invz + v.cy;
      }
      if (g.flags & PLH_GATE_KEYFRAME_BOUNDS) in = u >= v.min_x && u < v.max_x && w >= v.min_y && w < v.max_y;   // KeyFrame::IsInImage
      else in = !(u < v.min_x || u > v.max_x) && !(w < v.min_y || w > v.max_y);
    }
    if (in) {
      const float PO[3] = {P[0] - v.Ow[0], P[1] - v.Ow[1], P[2] - v.Ow[2]};
      const float dist = (g.flags & PLH_GATE_DIST_OF_TARGET) ? norm3(Pc) : norm3(PO);
      in = !(dist < minInv[i] || dist > maxInv[i]);
      if (in && (g.flags & PLH_GATE_NORMAL)) {
        const float N[3] = {normal[i * 3], normal[i * 3 + 1], normal[i * 3 + 2]};
        in = !(dot3(PO, N) < 0.5 * (double)dist);
      }
      if (in) {
        if (maxRaw) {
          int nScale = predict_scale(maxRaw[i], dist, v.log_scale_factor);
          if (nScale < 0) nScale = 0;
          else if (nScale >= v.n_scale_levels) nScale = v.n_scale_levels - 1;
          lvl = nScale;
        }
        dOut = dist;
        ok = 1;
      }
    }
  }
  if (!ok) { u = 0.f; w = 0.f; }
  valid[i] = ok; uv[i * 2] = u; uv[i * 2 + 1] = w; distOut[i] = dOut; level[i] = lvl;
}

__global__ void __launch_bounds__(256) k_frustum_points(const plh_frame_view* views, const int* nArr, int qcap, const float* pos,
                                                        const float* normal, const float* minDist, const float* maxDist,
                                                        float cosLimit, uint8_t* valid, float* uv, int* level, float* viewcos) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= qcap) return;
  const long long o = (long long)b * qcap + i;
  uint8_t ok = 0;
  float u = 0.f, w = 0.f, vc = 0.f;
  int lvl = 0;
  if (i < nArr[b]) {
    const plh_frame_view v = views[b];
    const float P[3] = {pos[o * 3], pos[o * 3 + 1], pos[o * 3 + 2]};
    float Pc[3];
    to_camera(v, P, Pc);
    if (!(Pc[2] < 0.0f)) {
      const float invz = 1.0f / Pc[2];
      const float uu = v.fx * Pc[0] * invz + v.cx;
      const float ww = v.fy * Pc[1] * invz + v.cy;
      if (!(uu < v.min_x || uu > v.max_x) && !(ww < v.min_y || ww > v.max_y)) {
        const float maxD = 1.2f * maxDist[o], minD = 0.8f * minDist[o];
        const float PO[3] = {P[0] - v.Ow[0], P[1] - v.Ow[1], P[2] - v.Ow[2]};
        const float dist = norm3(PO);
        if (!(dist < minD || dist > maxD)) {
          const float N[3] = {normal[o * 3], normal[o * 3 + 1], normal[o * 3 + 2]};
          const float c = (float)(dot3(PO, N) / dist);
          if (!(c < cosLimit)) {
            int nScale = predict_scale(maxDist[o], dist, v.log_scale_factor);
            if (nScale < 0) nScale = 0;
            else if (nScale >= v.n_scale_levels) nScale = v.n_scale_levels - 1;
            ok = 1; u = uu; w = ww; vc = c; lvl = nScale;
          }
        }
      }
    }
  }
  valid[o] = ok; uv[o * 2] = u; uv[o * 2 + 1] = w; level[o] = lvl; viewcos[o] = vc;
}

__global__ void __launch_bounds__(256) k_frustum_lines(const plh_frame_view* views, const int* nArr, int qcap, const float* pos6,
                                                       const float* normal, const float* minDist, const float* maxDist,
                                                       float cosLimit, uint8_t* valid, float* seg, int* level, float* viewcos) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= qcap) return;
  const long long o = (long long)b * qcap + i;
  uint8_t ok = 0;
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, vc = 0.f;
  int lvl = 0;
  if (i < nArr[b]) {
    const plh_frame_view v = views[b];
    float SP[3], EP[3], SPc[3], EPc[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { SP[k] = pos6[o * 6 + k]; EP[k] = pos6[o * 6 + 3 + k]; }
    to_camera(v, SP, SPc);
    to_camera(v, EP, EPc);
    if (!(SPc[2] < 0.0f || EPc[2] < 0.0f)) {
      const float invz1 = 1.0f / SPc[2];
      const float u1 = v.fx * SPc[0] * invz1 + v.cx;
      const float v1 = v.fy * SPc[1] * invz1 + v.cy;
      const float invz2 = 1.0f / EPc[2];
      const float u2 = v.fx * EPc[0] * invz2 + v.cx;
      const float v2 = v.fy * EPc[1] * invz2 + v.cy;
      const bool in1 = !(u1 < v.min_x || u1 > v.max_x) && !(v1 < v.min_y || v1 > v.max_y);
      const bool in2 = !(u2 < v.min_x || u2 > v.max_x) && !(v2 < v.min_y || v2 > v.max_y);
      if (in1 && in2) {
        const float maxD = 1.2f * maxDist[o], minD = 0.8f * minDist[o];
        float OM[3];
#pragma unroll
        for (int k = 0; k < 3; k++) OM[k] = 0.5f * (SP[k] + EP[k]) - v.Ow[k];
        const float dist = norm3(OM);
        if (!(dist < minD || dist > maxD)) {
          const float N[3] = {normal[o * 3], normal[o * 3 + 1], normal[o * 3 + 2]};
          const float c = (float)(dot3(OM, N) / dist);
          if (!(c < cosLimit)) {
            ok = 1; s4[0] = u1; s4[1] = v1; s4[2] = u2; s4[3] = v2; vc = c;
            lvl = predict_scale(maxDist[o], dist, v.log_scale_factor);   // MapLine::PredictScale does not clamp
          }
        }
      }
    }
  }
  valid[o] = ok; level[o] = lvl; viewcos[o] = vc;
#pragma unroll
  for (int k = 0; k < 4; k++) seg[o * 4 + k] = s4[k];
}

}  // namespace plh

using namespace plh;

extern "C" {

plh_status plh_undistort_keypoints_batch_dev(const plh_keypoint* d_kps, const int32_t* d_n, int cap, int batch, const float K[4],
                                             const float D[5], plh_keypoint* d_kps_un, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_kps || !d_n || !K || !d_kps_un || cap <= 0 || batch <= 0) {
    set_error("plh_undistort_keypoints_batch_dev: invalid argument");
    return PLH_ERR_INVALID;
  }
  UndistArgs u;
  u.fx = K[0]; u.fy = K[1]; u.cx = K[2]; u.cy = K[3];
  u.enabled = D && D[0] != 0.0f;   // Frame.cc:917: mDistCoef.at<float>(0) == 0.0 -> mvKeysUn = mvKeys
  u.k1 = D ? D[0] : 0; u.k2 = D ? D[1] : 0; u.p1 = D ? D[2] : 0; u.p2 = D ? D[3] : 0; u.k3 = D ? D[4] : 0;
  hipLaunchKernelGGL(k_undistort_kps, dim3((cap + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, d_kps, (const int*)d_n, cap,
                     u, d_kps_un);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_distinctive_descriptor_batch_dev(const uint8_t* d_desc, const int32_t* d_offsets, int nsets, int32_t* d_best,
                                                void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_desc || !d_offsets || !d_best || nsets <= 0) {
    set_error("plh_distinctive_descriptor_batch_dev: invalid argument");
    return PLH_ERR_INVALID;
  }
  hipLaunchKernelGGL(k_distinctive, dim3(nsets), dim3(64), 0, (hipStream_t)stream, d_desc, (const int*)d_offsets, nsets, d_best);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}


static plh_status launch_frustum(int lines, const plh_frame_view* d_views, int frames, const int32_t* d_nq, int qcap, const float* d_pos,
                                 const float* d_normal, const float* d_min_dist, const float* d_max_dist, float cos_limit,
                                 uint8_t* d_valid, float* d_proj, int32_t* d_level, float* d_viewcos, void* stream, const char* who) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_views || !d_nq || !d_pos || !d_normal || !d_min_dist || !d_max_dist || !d_valid || !d_proj || !d_level || !d_viewcos ||
      frames <= 0 || qcap <= 0) {
    set_error("%s: invalid argument", who);
    return PLH_ERR_INVALID;
  }
  const dim3 grid((qcap + 255) / 256, frames);
  if (lines)
    hipLaunchKernelGGL(k_frustum_lines, grid, dim3(256), 0, (hipStream_t)stream, d_views, (const int*)d_nq, qcap, d_pos, d_normal,
                       d_min_dist, d_max_dist, cos_limit, d_valid, d_proj, (int*)d_level, d_viewcos);
  else
    hipLaunchKernelGGL(k_frustum_points, grid, dim3(256), 0, (hipStream_t)stream, d_views, (const int*)d_nq, qcap, d_pos, d_normal,
                       d_min_dist, d_max_dist, cos_limit, d_valid, d_proj, (int*)d_level, d_viewcos);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_frame_is_in_frustum_points_batch_dev(const plh_frame_view* d_views, int frames, const int32_t* d_nq, int qcap,
                                                    const float* d_pos, const float* d_normal, const float* d_min_dist,
                                                    const float* d_max_dist, float viewing_cos_limit, uint8_t* d_valid, float* d_uv,
                                                    int32_t* d_level, float* d_viewcos, void* stream) {
  return launch_frustum(0, d_views, frames, d_nq, qcap, d_pos, d_normal, d_min_dist, d_max_dist, viewing_cos_limit, d_valid, d_uv, d_level,
                        d_viewcos, stream, "plh_frame_is_in_frustum_points_batch_dev");
}

plh_status plh_frame_is_in_frustum_lines_batch_dev(const plh_frame_view* d_views, int frames, const int32_t* d_nq, int qcap,
                                                   const float* d_pos6, const float* d_normal, const float* d_min_dist,
                                                   const float* d_max_dist, float viewing_cos_limit, uint8_t* d_valid, float* d_seg,
                                                   int32_t* d_level, float* d_viewcos, void* stream) {
  return launch_frustum(1, d_views, frames, d_nq, qcap, d_pos6, d_normal, d_min_dist, d_max_dist, viewing_cos_limit, d_valid, d_seg,
                        d_level, d_viewcos, stream, "plh_frame_is_in_frustum_lines_batch_dev");
}


plh_status plh_frame_project_points_batch_dev(const plh_frame_view* d_views, int frames, const int32_t* d_nq, int qcap, const float* d_pos,
                                              int form, uint8_t* d_front, float* d_uv, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_views || !d_nq || !d_pos || !d_front || !d_uv || frames <= 0 || qcap <= 0 || form < 0 || form > 2) {
    set_error("plh_frame_project_points_batch_dev: invalid argument");
    return PLH_ERR_INVALID;
  }
  hipLaunchKernelGGL(k_project_points, dim3((qcap + 255) / 256, frames), dim3(256), 0, (hipStream_t)stream, d_views, (const int*)d_nq, qcap,
                     d_pos, form, d_front, d_uv);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}


plh_status plh_map_point_gates_dev(const plh_point_gates* gates, int n, const float* d_pos, const float* d_normal,
                                   const float* d_min_dist_inv, const float* d_max_dist_inv, const float* d_max_dist, uint8_t* d_valid,
                                   float* d_uv, float* d_dist, int32_t* d_level, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!gates || n <= 0 || !d_pos || !d_min_dist_inv || !d_max_dist_inv || !d_valid || !d_uv || !d_dist || !d_level ||
      ((gates->flags & PLH_GATE_NORMAL) && !d_normal)) {
    set_error("plh_map_point_gates_dev: invalid argument");
    return PLH_ERR_INVALID;
  }
  hipLaunchKernelGGL(k_map_point_gates, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *gates, n, d_pos, d_normal,
                     d_min_dist_inv, d_max_dist_inv, d_max_dist, d_valid, d_uv, d_dist, (int*)d_level);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_map_point_gates(const plh_point_gates* gates, int n, const float* pos, const float* normal, const float* min_dist_inv,
                               const float* max_dist_inv, const float* max_dist, uint8_t* valid, float* uv, float* dist, int32_t* level,
                               int device) {
  if (!gates || n < 0 || (n > 0 && (!pos || !min_dist_inv || !max_dist_inv || !valid || !uv || !dist || !level)) ||
      (n > 0 && (gates->flags & PLH_GATE_NORMAL) && !normal)) {
    set_error("plh_map_point_gates: invalid argument");
    return PLH_ERR_INVALID;
  }
  if (n == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stager st;
  const size_t N = (size_t)n;
  plh_status rc = st.begin(device, 2 * Stager::padded(N * 12) + 5 * Stager::padded(N * 4) + Stager::padded(N) + Stager::padded(N * 8));
  if (rc != PLH_OK) return rc;
  const float* dPos = st.in(pos, N * 3);
  const float* dNormal = (gates->flags & PLH_GATE_NORMAL) ? st.in(normal, N * 3) : nullptr;
  const float* dMin = st.in(min_dist_inv, N);
  const float* dMax = st.in(max_dist_inv, N);
  const float* dRaw = max_dist ? st.in(max_dist, N) : nullptr;
  uint8_t* dValid = st.inout(valid, N);
  float* dUv = st.out(uv, N * 2);
  float* dDist = st.out(dist, N);
  int32_t* dLevel = st.out(level, N);
  if ((rc = st.upload()) != PLH_OK) return rc;
  if ((rc = plh_map_point_gates_dev(gates, n, dPos, dNormal, dMin, dMax, dRaw, dValid, dUv, dDist, dLevel, st.stream())) != PLH_OK) return rc;
  return st.download();
}

}  // extern "C"
