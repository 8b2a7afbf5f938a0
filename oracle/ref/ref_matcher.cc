// oracle/_ref: the reference's own ORBmatcher (src/ORBmatcher.cc, every function, compiled from the source where it lies
// by oracle/ref/build_ref.sh) driven from flat arrays.  Frame / KeyFrame / MapPoint are the stand-ins of slam_stub.h
// (the real classes pull in the whole SLAM system); the grid lookup behind Frame::GetFeaturesInArea is the oracle's.
// Entry points:
//   SearchByBoW(KeyFrame*, Frame&)            src/ORBmatcher.cc:187-327   (+ ComputeThreeMaxima :1718-1759, DescriptorDistance)
//   SearchByBoW(KeyFrame*, KeyFrame*)         :574-709
//   SearchForInitialization                   :455-572
//   SearchByProjection(Frame&, MapPoints, th) :56-144
//   SearchByProjection(Cur, Last, th, bMono)  :1441-1585   } driven with the current pose = identity; the projection the
//   SearchByProjection(Cur, pKF, found, th, ORBdist) :1587-1716 } reference computes inline is handed back to the caller
// TEST INFRASTRUCTURE ONLY.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "ORBmatcher.h"

namespace ORB_SLAM2 {
float Frame::mnMinX = 0, Frame::mnMaxX = 0, Frame::mnMinY = 0, Frame::mnMaxY = 0;

void GridLookup::build() {
  cellStart.assign(64 * 48 + 1, 0);
  cellItems.assign(kps.size() + 1, 0);
  plo_frame_assign_grid(kps.data(), (int)kps.size(), gp, cellStart.data(), cellItems.data());
}
std::vector<size_t> GridLookup::query(float x, float y, float r, int minLevel, int maxLevel) const {
  std::vector<int32_t> out(kps.size() + 1);
  const int n = plo_features_in_area(kps.data(), gp, cellStart.data(), cellItems.data(), x, y, r, minLevel, maxLevel, out.data(),
                                     (int)out.size());
  return std::vector<size_t>(out.begin(), out.begin() + n);
}
}  // namespace ORB_SLAM2

using namespace ORB_SLAM2;

namespace {
cv::Mat desc_mat(const uint8_t* d, int n) {
  cv::Mat m(n > 0 ? n : 1, 32, CV_8U);
  if (n > 0) std::memcpy(m.data, d, (size_t)n * 32);
  if (n == 0) m = m.rowRange(0, 0);
  return m;
}
std::vector<cv::KeyPoint> keypoints(const plo_keypoint* k, int n) {
  std::vector<cv::KeyPoint> v(n);
  for (int i = 0; i < n; i++) v[i] = cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id);
  return v;
}
void feat_vec(DBoW2::FeatureVector& fv, const int32_t* node, int n) {   // Frame::ComputeBoW order: ascending feature index
  for (int i = 0; i < n; i++)
    if (node[i] >= 0) fv.addFeature((DBoW2::NodeId)node[i], (unsigned)i);
}
struct Points {   // owns the MapPoints of one harness call
  std::vector<std::unique_ptr<MapPoint> > all;
  MapPoint* make(long id) {
    all.emplace_back(new MapPoint());
    all.back()->mnId = (unsigned long)id;
    return all.back().get();
  }
};
}  // namespace

extern "C" {

// valid1[i] = KeyFrame feature i has a MapPoint that is not bad.  matches21[f] = KeyFrame feature whose MapPoint went to
// Frame feature f, or -1.  Returns nmatches.
int ref_orb_search_by_bow(const uint8_t* desc1, const float* angle1, const int32_t* node1, const uint8_t* valid1, int n1,
                          const uint8_t* desc2, const float* angle2, const int32_t* node2, int n2, float nnratio, int check_ori,
                          int32_t* matches21) {
  Points pts;
  KeyFrame kf;
  Frame f;
  kf.N = n1; f.N = n2;
  kf.mvKeysUn.resize(n1); kf.mvpMapPoints.assign(n1, nullptr);
  for (int i = 0; i < n1; i++) {
    kf.mvKeysUn[i].angle = angle1[i];
    if (valid1[i]) kf.mvpMapPoints[i] = pts.make(i);
  }
  kf.mvKeys = kf.mvKeysUn;
  f.mvKeys.resize(n2);
  for (int i = 0; i < n2; i++) f.mvKeys[i].angle = angle2[i];
  f.mvKeysUn = f.mvKeys;
  f.mvpMapPoints.assign(n2, nullptr);
  kf.mDescriptors = desc_mat(desc1, n1);
  f.mDescriptors = desc_mat(desc2, n2);
  feat_vec(kf.mFeatVec, node1, n1);
  feat_vec(f.mFeatVec, node2, n2);
  ORBmatcher m(nnratio, check_ori != 0);
  std::vector<MapPoint*> out;
  const int n = m.SearchByBoW(&kf, f, out);
  for (int i = 0; i < n2; i++) matches21[i] = out[i] ? (int32_t)out[i]->mnId : -1;
  return n;
}

// matches12[i1] = KeyFrame-2 feature whose MapPoint was matched to KeyFrame-1 feature i1, or -1
int ref_orb_search_by_bow_kfkf(const uint8_t* desc1, const float* angle1, const int32_t* node1, const uint8_t* valid1, int n1,
                               const uint8_t* desc2, const float* angle2, const int32_t* node2, const uint8_t* valid2, int n2,
                               float nnratio, int check_ori, int32_t* matches12) {
  Points pts;
  KeyFrame k1, k2;
  k1.N = n1; k2.N = n2;
  k1.mvKeysUn.resize(n1); k1.mvpMapPoints.assign(n1, nullptr);
  k2.mvKeysUn.resize(n2); k2.mvpMapPoints.assign(n2, nullptr);
  for (int i = 0; i < n1; i++) { k1.mvKeysUn[i].angle = angle1[i]; if (valid1[i]) k1.mvpMapPoints[i] = pts.make(i); }
  for (int i = 0; i < n2; i++) { k2.mvKeysUn[i].angle = angle2[i]; if (valid2[i]) k2.mvpMapPoints[i] = pts.make(i); }
  k1.mvKeys = k1.mvKeysUn; k2.mvKeys = k2.mvKeysUn;
  k1.mDescriptors = desc_mat(desc1, n1);
  k2.mDescriptors = desc_mat(desc2, n2);
  feat_vec(k1.mFeatVec, node1, n1);
  feat_vec(k2.mFeatVec, node2, n2);
  ORBmatcher m(nnratio, check_ori != 0);
  std::vector<MapPoint*> out;
  const int n = m.SearchByBoW(&k1, &k2, out);
  for (int i = 0; i < n1; i++) matches12[i] = out[i] ? (int32_t)out[i]->mnId : -1;
  return n;
}

// prev_matched [n1][2] in/out, matches12 [n1] out
int ref_orb_search_for_initialization(const plo_keypoint* kps1, const uint8_t* desc1, int n1, const plo_keypoint* kps2,
                                      const uint8_t* desc2, int n2, const float gp[6], float* prev_matched, int window_size,
                                      float nnratio, int check_ori, int32_t* matches12) {
  Frame f1, f2;
  f1.N = n1; f2.N = n2;
  f1.mvKeysUn = keypoints(kps1, n1); f1.mvKeys = f1.mvKeysUn;
  f2.mvKeysUn = keypoints(kps2, n2); f2.mvKeys = f2.mvKeysUn;
  f1.mDescriptors = desc_mat(desc1, n1);
  f2.mDescriptors = desc_mat(desc2, n2);
  f2.grid.kps.assign(kps2, kps2 + n2);
  std::memcpy(f2.grid.gp, gp, sizeof(f2.grid.gp));
  f2.grid.build();
  std::vector<cv::Point2f> prev(n1);
  for (int i = 0; i < n1; i++) prev[i] = cv::Point2f(prev_matched[2 * i], prev_matched[2 * i + 1]);
  std::vector<int> m12;
  ORBmatcher m(nnratio, check_ori != 0);
  const int n = m.SearchForInitialization(f1, f2, prev, m12, window_size);
  for (int i = 0; i < n1; i++) { matches12[i] = m12[i]; prev_matched[2 * i] = prev[i].x; prev_matched[2 * i + 1] = prev[i].y; }
  return n;
}

// SearchByProjection(F, vpMapPoints, th).  Query q: valid = mbTrackInView && !isBad(); xy = mTrackProjX / Y; level =
// mnTrackScaleLevel; viewcos; desc; hasobs = Observations() > 0.  occupied[idx] (in/out) = F.mvpMapPoints[idx] is set and
// has observations.  assigned[idx] = query stored into F.mvpMapPoints[idx] by this call, or -1.  Monocular (mvuRight = -1).
int ref_orb_search_by_projection_mp(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6],
                                    const float* scale_factors, int nlevels, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                    const float* q_xy, const int32_t* q_level, const float* q_viewcos, const uint8_t* q_desc,
                                    const uint8_t* q_hasobs, float th, float nnratio, int32_t* assigned) {
  Points pts;
  Frame f;
  f.N = n;
  f.mvKeysUn = keypoints(kps_un, n); f.mvKeys = f.mvKeysUn;
  f.mDescriptors = desc_mat(desc, n);
  f.mvuRight.assign(n, -1.f);
  f.mvScaleFactors.assign(scale_factors, scale_factors + nlevels);
  f.mvpMapPoints.assign(n, nullptr);
  for (int i = 0; i < n; i++)
    if (occupied[i]) { f.mvpMapPoints[i] = pts.make(-1); f.mvpMapPoints[i]->nobs = 1; }
  f.grid.kps.assign(kps_un, kps_un + n);
  std::memcpy(f.grid.gp, gp, sizeof(f.grid.gp));
  f.grid.build();
  std::vector<MapPoint*> q(nq);
  for (int i = 0; i < nq; i++) {
    MapPoint* p = pts.make(i);
    p->mbTrackInView = q_valid[i] != 0;
    p->mTrackProjX = q_xy[2 * i]; p->mTrackProjY = q_xy[2 * i + 1];
    p->mnTrackScaleLevel = q_level[i];
    p->mTrackViewCos = q_viewcos[i];
    p->desc = desc_mat(q_desc + (size_t)i * 32, 1);
    p->nobs = q_hasobs[i] ? 1 : 0;
    q[i] = p;
  }
  ORBmatcher m(nnratio, true);
  const int nm = m.SearchByProjection(f, q, th);
  for (int i = 0; i < n; i++) {
    MapPoint* p = f.mvpMapPoints[i];
    assigned[i] = (p && (long)p->mnId >= 0) ? (int32_t)p->mnId : -1;
    occupied[i] = (p && p->Observations() > 0) ? 1 : 0;
  }
  return nm;
}

}  // extern "C"

// Shared by the two pose-driven searches below: the current frame with pose = identity (so that the reference's
// Rcw*x3Dw+tcw returns x3Dw exactly and nothing depends on how the stub's float algebra rounds).
static void fill_current(Frame& f, Points& pts, const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6],
                         const float* scale_factors, int nlevels, const uint8_t* occupied, const float K[4]) {
  f.N = n;
  f.mvKeysUn = keypoints(kps_un, n); f.mvKeys = f.mvKeysUn;
  f.mDescriptors = desc_mat(desc, n);
  f.mvuRight.assign(n, -1.f);
  f.mvScaleFactors.assign(scale_factors, scale_factors + nlevels);
  f.mvpMapPoints.assign(n, nullptr);
  for (int i = 0; i < n; i++)
    if (occupied[i]) { f.mvpMapPoints[i] = pts.make(-1); f.mvpMapPoints[i]->nobs = 1; }
  f.grid.kps.assign(kps_un, kps_un + n);
  std::memcpy(f.grid.gp, gp, sizeof(f.grid.gp));
  f.grid.build();
  f.fx = K[0]; f.fy = K[1]; f.cx = K[2]; f.cy = K[3];
  Frame::mnMinX = gp[0]; Frame::mnMinY = gp[1]; Frame::mnMaxX = gp[2]; Frame::mnMaxY = gp[3];
  f.mTcw = cv::Mat::zeros(4, 4, CV_32F);
  for (int i = 0; i < 4; i++) f.mTcw.at<float>(i, i) = 1.f;
}
static cv::Mat point3(const float* x) {
  cv::Mat m(3, 1, CV_32F);
  for (int i = 0; i < 3; i++) m.at<float>(i) = x[i];
  return m;
}
// The projection exactly as ORBmatcher.cc:1476-1484 / :1617-1622 writes it (same compiler, same flags): what the caller of the
// flat-array searches hands over as q_uv.
static void project(const float* x, const float K[4], float* uv, uint8_t* front) {
  const float xc = x[0], yc = x[1];
  const float invzc = 1.0 / x[2];
  uv[0] = K[0] * xc * invzc + K[2];
  uv[1] = K[1] * yc * invzc + K[3];
  if (front) *front = invzc < 0 ? 0 : 1;
}

extern "C" {

// SearchByProjection(CurrentFrame, LastFrame, th, bMono), src/ORBmatcher.cc:1441-1585.  Query i = LastFrame feature i with
// MapPoint (q_mp) at world position q_xyz; mode 0 = bMono, 1 = forward (LastFrame ahead along z by more than mb), 2 = backward.
// Writes the projections (uv_out) and invzc >= 0 (front_out) the flat-array searches take as inputs.
int ref_orb_search_by_projection_frame(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6],
                                       const float* scale_factors, int nlevels, uint8_t* occupied, int nq, const uint8_t* q_mp,
                                       const uint8_t* q_outlier, const float* q_xyz, const int32_t* q_octave, const float* q_angle,
                                       const uint8_t* q_desc, const uint8_t* q_hasobs, const float K[4], float th, int mode,
                                       int check_ori, float* uv_out, uint8_t* front_out, int32_t* assigned) {
  Points pts;
  Frame cur, last;
  fill_current(cur, pts, kps_un, desc, n, gp, scale_factors, nlevels, occupied, K);
  cur.mb = 0.1f;
  last.N = nq;
  last.mvKeys.resize(nq); last.mvKeysUn.resize(nq);
  last.mvpMapPoints.assign(nq, nullptr);
  last.mvbOutlier.assign(nq, false);
  last.mTcw = cur.mTcw.clone();
  last.mTcw.at<float>(2, 3) = mode == 1 ? 1.f : mode == 2 ? -1.f : 0.f;
  for (int i = 0; i < nq; i++) {
    last.mvKeys[i].octave = q_octave[i]; last.mvKeysUn[i].octave = q_octave[i];
    last.mvKeys[i].angle = q_angle[i]; last.mvKeysUn[i].angle = q_angle[i];
    last.mvbOutlier[i] = q_outlier[i] != 0;
    project(q_xyz + 3 * i, K, uv_out + 2 * i, front_out + i);
    if (!q_mp[i]) continue;
    MapPoint* p = pts.make(i);
    p->pos = point3(q_xyz + 3 * i);
    p->desc = desc_mat(q_desc + (size_t)i * 32, 1);
    p->nobs = q_hasobs[i] ? 1 : 0;
    last.mvpMapPoints[i] = p;
  }
  ORBmatcher m(0.9f, check_ori != 0);
  const int nm = m.SearchByProjection(cur, last, th, mode == 0);
  for (int i = 0; i < n; i++) {
    MapPoint* p = cur.mvpMapPoints[i];
    assigned[i] = (p && (long)p->mnId >= 0) ? (int32_t)p->mnId : -1;
    occupied[i] = (p && p->Observations() > 0) ? 1 : 0;
  }
  return nm;
}

// SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1587-1716 (relocalisation).
// Query i = pKF feature i: q_mp, q_bad, q_found (in sAlreadyFound), q_inrange (0 = the point's distance-invariance interval
// excludes it), q_level = PredictScale.  occupied = CurrentFrame.mvpMapPoints[i2] != NULL.
int ref_orb_search_by_projection_kf(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6],
                                    const float* scale_factors, int nlevels, uint8_t* occupied, int nq, const uint8_t* q_mp,
                                    const uint8_t* q_bad, const uint8_t* q_found, const uint8_t* q_inrange, const float* q_xyz,
                                    const int32_t* q_level, const float* q_angle, const uint8_t* q_desc, const float K[4], float th,
                                    int orb_dist, int check_ori, float* uv_out, int32_t* assigned) {
  Points pts;
  Frame cur;
  fill_current(cur, pts, kps_un, desc, n, gp, scale_factors, nlevels, occupied, K);
  KeyFrame kf;
  kf.N = nq;
  kf.mvKeysUn.resize(nq);
  kf.mvpMapPoints.assign(nq, nullptr);
  std::set<MapPoint*> found;
  for (int i = 0; i < nq; i++) {
    kf.mvKeysUn[i].angle = q_angle[i];
    project(q_xyz + 3 * i, K, uv_out + 2 * i, nullptr);
    if (!q_mp[i]) continue;
    MapPoint* p = pts.make(i);
    p->pos = point3(q_xyz + 3 * i);
    p->desc = desc_mat(q_desc + (size_t)i * 32, 1);
    p->bad = q_bad[i] != 0;
    p->predicted = q_level[i];
    if (!q_inrange[i]) { p->minDist = 1e29f; }
    if (q_found[i]) found.insert(p);
    kf.mvpMapPoints[i] = p;
  }
  ORBmatcher m(0.9f, check_ori != 0);
  const int nm = m.SearchByProjection(cur, &kf, found, th, orb_dist);
  for (int i = 0; i < n; i++) {
    MapPoint* p = cur.mvpMapPoints[i];
    assigned[i] = (p && (long)p->mnId >= 0) ? (int32_t)p->mnId : -1;
    occupied[i] = p ? 1 : 0;
  }
  return nm;
}

}  // extern "C"

// ---- the KeyFrame-side searches (Fuse x2, loop-closing SearchByProjection).  KeyFrame pose / Scw = identity. ----
static void fill_keyframe(KeyFrame& kf, const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6],
                          const float* scale_factors, const float* inv_sigma2, int nlevels, const float K[4]) {
  kf.N = n;
  kf.mvKeysUn = keypoints(kps_un, n); kf.mvKeys = kf.mvKeysUn;
  kf.mDescriptors = desc_mat(desc, n);
  kf.mvuRight.assign(n, -1.f);
  kf.mvScaleFactors.assign(scale_factors, scale_factors + nlevels);
  if (inv_sigma2) kf.mvInvLevelSigma2.assign(inv_sigma2, inv_sigma2 + nlevels);
  kf.mvpMapPoints.assign(n, nullptr);
  kf.grid.kps.assign(kps_un, kps_un + n);
  std::memcpy(kf.grid.gp, gp, sizeof(kf.grid.gp));
  kf.grid.build();
  kf.fx = K[0]; kf.fy = K[1]; kf.cx = K[2]; kf.cy = K[3];
  kf.mnMinX = gp[0]; kf.mnMinY = gp[1]; kf.mnMaxX = gp[2]; kf.mnMaxY = gp[3];   // float -> const int, as KeyFrame.cc:44 does
  kf.Rcw = cv::Mat::zeros(3, 3, CV_32F);
  for (int i = 0; i < 3; i++) kf.Rcw.at<float>(i, i) = 1.f;
  kf.tcw = cv::Mat::zeros(3, 1, CV_32F);
  kf.Ow = cv::Mat::zeros(3, 1, CV_32F);
}
// The three projections as written at ORBmatcher.cc:952-957 (Fuse), :1103-1108 (Fuse, Sim3) and :369-375 (SearchByProjection,
// Sim3): `1/z` in float for the first and third, `1.0/z` in double for the second.
static void project_kf(const float* X, const float K[4], bool double_inv, const KeyFrame& kf, float* uv, uint8_t* front, uint8_t* inimg) {
  const float invz = double_inv ? (float)(1.0 / X[2]) : 1 / X[2];
  const float x = X[0] * invz;
  const float y = X[1] * invz;
  uv[0] = K[0] * x + K[2];
  uv[1] = K[1] * y + K[3];
  *front = X[2] < 0.0f ? 0 : 1;
  *inimg = kf.IsInImage(uv[0], uv[1]) ? 1 : 0;
}
static MapPoint* query_point(Points& pts, int i, const float* xyz, const uint8_t* q_desc, const uint8_t* q_bad, const uint8_t* q_inrange,
                             const uint8_t* q_viewok, const int32_t* q_level) {
  MapPoint* p = pts.make(i);
  p->pos = point3(xyz + 3 * i);
  p->desc = desc_mat(q_desc + (size_t)i * 32, 1);
  p->bad = q_bad[i] != 0;
  p->predicted = q_level[i];
  if (!q_inrange[i]) p->minDist = 1e29f;
  const float nz[3] = {0.f, 0.f, q_viewok[i] ? 1.f : -1.f};   // PO.dot(Pn) = +-z against 0.5*|PO| (|PO| <= 1.3 z inside the image)
  p->normal = point3(nz);
  p->nobs = 1;
  return p;
}

extern "C" {

// ORBmatcher::Fuse(pKF, vpMapPoints, th), src/ORBmatcher.cc:914-1061.  kf_mp[idx]: 0 = empty slot, 1 = MapPoint with more
// observations than the queries, 2 = a bad MapPoint.  q_mp = 0: NULL entry; q_inkf: pMP->IsInKeyFrame(pKF).
// best_idx[q] = the keypoint the loop settled on (read back from the GetMapPoint(bestIdx) call of that query), or -1.
int ref_orb_fuse(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6], const float* scale_factors,
                 const float* inv_sigma2, int nlevels, const uint8_t* kf_mp, int nq, const uint8_t* q_mp, const uint8_t* q_bad,
                 const uint8_t* q_inkf, const uint8_t* q_inrange, const uint8_t* q_viewok, const float* q_xyz, const int32_t* q_level,
                 const uint8_t* q_desc, const float K[4], float th, float* uv_out, uint8_t* front_out, uint8_t* inimg_out,
                 int32_t* best_idx) {
  Points pts;
  KeyFrame kf;
  fill_keyframe(kf, kps_un, desc, n, gp, scale_factors, inv_sigma2, nlevels, K);
  for (int i = 0; i < n; i++)
    if (kf_mp[i]) { MapPoint* p = pts.make(-1); p->nobs = 2; p->bad = kf_mp[i] == 2; kf.mvpMapPoints[i] = p; }
  std::vector<MapPoint*> q(nq, nullptr);
  for (int i = 0; i < nq; i++) {
    project_kf(q_xyz + 3 * i, K, false, kf, uv_out + 2 * i, front_out + i, inimg_out + i);
    best_idx[i] = -1;
    if (!q_mp[i]) continue;
    q[i] = query_point(pts, i, q_xyz, q_desc, q_bad, q_inrange, q_viewok, q_level);
    if (q_inkf[i]) q[i]->obs[&kf] = 0;
  }
  ORBmatcher m(0.6f, true);
  const int nf = m.Fuse(&kf, q, th);
  for (const auto& e : kf.getLog)
    if (e.first && (long)e.first->mnId >= 0) best_idx[e.first->mnId] = e.second;
  return nf;
}

// ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint), :1063-1197.  q_slot[q] >= 0: the query already sits in that
// KeyFrame slot (so it is in spAlreadyFound).
int ref_orb_fuse_sim3(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6], const float* scale_factors,
                      int nlevels, const uint8_t* kf_mp, int nq, const uint8_t* q_bad, const int32_t* q_slot, const uint8_t* q_inrange,
                      const uint8_t* q_viewok, const float* q_xyz, const int32_t* q_level, const uint8_t* q_desc, const float K[4],
                      float th, float* uv_out, uint8_t* front_out, uint8_t* inimg_out, int32_t* best_idx) {
  Points pts;
  KeyFrame kf;
  fill_keyframe(kf, kps_un, desc, n, gp, scale_factors, nullptr, nlevels, K);
  for (int i = 0; i < n; i++)
    if (kf_mp[i]) { MapPoint* p = pts.make(-1); p->nobs = 2; p->bad = kf_mp[i] == 2; kf.mvpMapPoints[i] = p; }
  std::vector<MapPoint*> q(nq, nullptr), repl(nq, nullptr);
  for (int i = 0; i < nq; i++) {
    project_kf(q_xyz + 3 * i, K, true, kf, uv_out + 2 * i, front_out + i, inimg_out + i);
    best_idx[i] = -1;
    q[i] = query_point(pts, i, q_xyz, q_desc, q_bad, q_inrange, q_viewok, q_level);
    if (q_slot[i] >= 0) kf.mvpMapPoints[q_slot[i]] = q[i];
  }
  cv::Mat Scw = cv::Mat::zeros(4, 4, CV_32F);
  for (int i = 0; i < 4; i++) Scw.at<float>(i, i) = 1.f;
  ORBmatcher m(0.8f, true);
  const int nf = m.Fuse(&kf, Scw, q, th, repl);
  for (const auto& e : kf.getLog)
    if (e.first && (long)e.first->mnId >= 0) best_idx[e.first->mnId] = e.second;
  return nf;
}

// ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th), :329-453.  q_slot[q] >= 0: the query is vpMatched[slot]
// on entry; occupied[idx] = other entries of vpMatched that are not NULL (in), vpMatched[idx] != NULL (out).
int ref_orb_search_by_projection_sim3(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6],
                                      const float* scale_factors, int nlevels, uint8_t* occupied, int nq, const uint8_t* q_bad,
                                      const int32_t* q_slot, const uint8_t* q_inrange, const uint8_t* q_viewok, const float* q_xyz,
                                      const int32_t* q_level, const uint8_t* q_desc, const float K[4], int th, float* uv_out,
                                      uint8_t* front_out, uint8_t* inimg_out, int32_t* assigned) {
  Points pts;
  KeyFrame kf;
  fill_keyframe(kf, kps_un, desc, n, gp, scale_factors, nullptr, nlevels, K);
  std::vector<MapPoint*> matched(n, nullptr), q(nq, nullptr);
  for (int i = 0; i < n; i++)
    if (occupied[i]) matched[i] = pts.make(-1);
  for (int i = 0; i < nq; i++) {
    project_kf(q_xyz + 3 * i, K, false, kf, uv_out + 2 * i, front_out + i, inimg_out + i);
    q[i] = query_point(pts, i, q_xyz, q_desc, q_bad, q_inrange, q_viewok, q_level);
    if (q_slot[i] >= 0) matched[q_slot[i]] = q[i];
  }
  std::vector<MapPoint*> before = matched;
  cv::Mat Scw = cv::Mat::zeros(4, 4, CV_32F);
  for (int i = 0; i < 4; i++) Scw.at<float>(i, i) = 1.f;
  ORBmatcher m(0.75f, true);
  const int nm = m.SearchByProjection(&kf, Scw, q, matched, th);
  for (int i = 0; i < n; i++) {
    assigned[i] = (matched[i] && matched[i] != before[i]) ? (int32_t)matched[i]->mnId : -1;
    occupied[i] = matched[i] ? 1 : 0;
  }
  return nm;
}

}  // extern "C"

extern "C" {

// ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), :1199-1439, with both keyframe poses and the Sim3
// = identity.  Side a (a = 1, 2): keypoints / descriptors / grid of KeyFrame a and, per keypoint slot, its MapPoint (mp_a,
// bad_a, world position xyz_a, PredictScale level_a, inrange_a, descriptor mpdesc_a).  already12[i1]: -1 = vpMatches12[i1]
// NULL on entry, otherwise the KeyFrame-2 index of the MapPoint stored there (>= n2: a point KeyFrame 2 does not see).
// uv12 / front12 / inimg12: projection of side 1's points into KeyFrame 2 (:1258-1267), uv21 ... the other way (:1339-1348).
// match12[i1] = KeyFrame-2 slot whose MapPoint was written to vpMatches12[i1] by the agreement pass, or -1.
int ref_orb_search_by_sim3(const plo_keypoint* kps1, const uint8_t* desc1, int n1, const uint8_t* mp1, const uint8_t* bad1,
                           const float* xyz1, const int32_t* level1, const uint8_t* inrange1, const uint8_t* mpdesc1,
                           const plo_keypoint* kps2, const uint8_t* desc2, int n2, const uint8_t* mp2, const uint8_t* bad2,
                           const float* xyz2, const int32_t* level2, const uint8_t* inrange2, const uint8_t* mpdesc2,
                           const float gp[6], const float* scale_factors, int nlevels, const int32_t* already12, const float K[4],
                           float th, float* uv12, uint8_t* front12, uint8_t* inimg12, float* uv21, uint8_t* front21,
                           uint8_t* inimg21, int32_t* match12) {
  Points pts;
  KeyFrame kf1, kf2;
  fill_keyframe(kf1, kps1, desc1, n1, gp, scale_factors, nullptr, nlevels, K);
  fill_keyframe(kf2, kps2, desc2, n2, gp, scale_factors, nullptr, nlevels, K);
  auto side = [&](KeyFrame& kf, const KeyFrame& other, int n, const uint8_t* mp, const uint8_t* bad, const float* xyz,
                  const int32_t* level, const uint8_t* inrange, const uint8_t* mpdesc, float* uv, uint8_t* front, uint8_t* inimg) {
    for (int i = 0; i < n; i++) {
      project_kf(xyz + 3 * i, K, true, other, uv + 2 * i, front + i, inimg + i);
      if (!mp[i]) continue;
      MapPoint* p = pts.make(i);
      p->pos = point3(xyz + 3 * i);
      p->desc = desc_mat(mpdesc + (size_t)i * 32, 1);
      p->bad = bad[i] != 0;
      p->predicted = level[i];
      if (!inrange[i]) p->minDist = 1e29f;
      kf.mvpMapPoints[i] = p;
    }
  };
  side(kf1, kf2, n1, mp1, bad1, xyz1, level1, inrange1, mpdesc1, uv12, front12, inimg12);
  side(kf2, kf1, n2, mp2, bad2, xyz2, level2, inrange2, mpdesc2, uv21, front21, inimg21);
  std::vector<MapPoint*> m12(n1, nullptr);
  for (int i = 0; i < n1; i++) {
    if (already12[i] < 0) continue;
    MapPoint* p = pts.make(-1);
    p->obs[&kf2] = (size_t)already12[i];
    m12[i] = p;
  }
  std::vector<MapPoint*> before = m12;
  cv::Mat R12 = cv::Mat::zeros(3, 3, CV_32F);
  for (int i = 0; i < 3; i++) R12.at<float>(i, i) = 1.f;
  cv::Mat t12 = cv::Mat::zeros(3, 1, CV_32F);
  const float s12 = 1.f;
  ORBmatcher m(0.75f, true);
  const int nf = m.SearchBySim3(&kf1, &kf2, m12, s12, R12, t12, th);
  for (int i = 0; i < n1; i++) match12[i] = (m12[i] && m12[i] != before[i]) ? (int32_t)m12[i]->mnId : -1;
  return nf;
}

}  // extern "C"

extern "C" {

// ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo = false), :720-912 (+ CheckDistEpipolarLine
// :153-170).  KeyFrame 2 at the origin, KeyFrame 1's camera centre at `cw` (the epipole the function derives from it comes
// back in epi_out).  matches12[i1] = i2 or -1.
int ref_orb_search_for_triangulation(const plo_keypoint* kps1, const uint8_t* desc1, const int32_t* node1, const uint8_t* has_mp1,
                                     int n1, const plo_keypoint* kps2, const uint8_t* desc2, const int32_t* node2,
                                     const uint8_t* has_mp2, int n2, const float F12[9], const float cw[3], const float K[4],
                                     const float* scale_factors2, const float* level_sigma2_2, int nlevels, int check_ori,
                                     float* epi_out, int32_t* matches12) {
  Points pts;
  KeyFrame kf1, kf2;
  const float gp[6] = {0, 0, 640, 480, 64.f / 640.f, 48.f / 480.f};   // the grid is not used by this search
  fill_keyframe(kf1, kps1, desc1, n1, gp, scale_factors2, nullptr, nlevels, K);
  fill_keyframe(kf2, kps2, desc2, n2, gp, scale_factors2, nullptr, nlevels, K);
  kf2.mvLevelSigma2.assign(level_sigma2_2, level_sigma2_2 + nlevels);
  feat_vec(kf1.mFeatVec, node1, n1);
  feat_vec(kf2.mFeatVec, node2, n2);
  for (int i = 0; i < n1; i++) if (has_mp1[i]) kf1.mvpMapPoints[i] = pts.make(-1);
  for (int i = 0; i < n2; i++) if (has_mp2[i]) kf2.mvpMapPoints[i] = pts.make(-1);
  kf1.Ow = point3(cw);
  {   // :732-737 with R2w = I, t2w = 0
    const float invz = 1.0f / cw[2];
    epi_out[0] = K[0] * cw[0] * invz + K[2];
    epi_out[1] = K[1] * cw[1] * invz + K[3];
  }
  cv::Mat F(3, 3, CV_32F);
  for (int i = 0; i < 9; i++) F.at<float>(i / 3, i % 3) = F12[i];
  std::vector<std::pair<size_t, size_t> > pairs;
  ORBmatcher m(0.6f, check_ori != 0);
  const int nm = m.SearchForTriangulation(&kf1, &kf2, F, pairs, false);
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  for (const auto& pr : pairs) matches12[pr.first] = (int32_t)pr.second;
  return nm;
}

}  // extern "C"
