// Frame / map post-processing either side of the matching path, batch-first:
//   k_undistort_kps   Frame::UndistortKeyPoints = cv::undistortPoints(pts, pts, K, D, noArray(), K)   reference src/Frame.cc:915-945
//   k_distinctive     MapPoint::ComputeDistinctiveDescriptors / MapLine twin                          reference src/MapPoint.cc:249-314,
//                                                                                                    src/MapLine.cpp:256-330
// Both are exact: the iteration runs in double with the reference's operation order (no FMA contraction), the median
// selection is integer.
#include "plh_common.h"

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

}  // extern "C"
