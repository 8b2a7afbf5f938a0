// oracle/_ref: the reference's own LSDmatcher (src/LSDmatcher.cpp, every function, compiled from the source where it
// lies by oracle/ref/build_ref.sh) driven from flat arrays.  Frame / KeyFrame / MapLine are the stand-ins of slam_stub.h;
// cv::BFMatcher::knnMatch and the grid lookup behind Frame::GetFeaturesInAreaForLine are the oracle's restatements; the
// matcher's debugging pictures (cv::line / cv::imwrite) are no-ops.  Entry points:
//   FrameBFMatch + lineDescriptorMAD                 src/LSDmatcher.cpp:462-486, 627-652
//   SearchDouble(Frame&, Frame&, LineMatches)        :427-460
//   SearchByProjection(Cur, Last, th)                :72-176
//   SearchByProjection(F, vpMapLines, th)            :221-338
//   FrameBFMatchNew + mutualOverlap                  :488-625
//   SearchForTriangulationNew (+ ComputeF12)         :780-858
// TEST INFRASTRUCTURE ONLY.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "LSDmatcher.h"

namespace ORB_SLAM2 {
void LineGridLookup::build() {
  cellStart.assign(64 * 48 + 1, 0);
  cellItems.assign(kl.size() * 64 + 64, 0);
  plo_frame_assign_grid_lines(kl.data(), (int)kl.size(), gp, cellStart.data(), cellItems.data(), (int)cellItems.size());
}
std::vector<size_t> LineGridLookup::query(float x1, float y1, float x2, float y2, float r, float TH) const {
  std::vector<int32_t> out(kl.size() + 1);
  const int n = plo_features_in_area_for_line(kl.data(), fn.data(), (int)kl.size(), gp, cellStart.data(), cellItems.data(), x1, y1, x2,
                                              y2, r, TH, out.data(), (int)out.size());
  return std::vector<size_t>(out.begin(), out.begin() + n);
}
// KeyFrame::GetLinesInArea restated from src/KeyFrame.cc:647-683 (brute force over the KeyFrame's lines).
std::vector<size_t> KeyFrame::GetLinesInArea(const float& x1, const float& y1, const float& x2, const float& y2, const float& r,
                                             const float TH) const {
  std::vector<size_t> vIndices;
  float delta1x = x1 - x2, delta1y = y1 - y2;
  const float norm_delta1 = std::sqrt(delta1x * delta1x + delta1y * delta1y);
  delta1x /= norm_delta1;
  delta1y /= norm_delta1;
  for (size_t i = 0; i < mvKeyLines.size(); i++) {
    const KeyLine& k = mvKeyLines[i];
    const float distance = (0.5 * (x1 + x2) - k.pt.x) * (0.5 * (x1 + x2) - k.pt.x) + (0.5 * (y1 + y2) - k.pt.y) * (0.5 * (y1 + y2) - k.pt.y);
    if (distance > r * r) continue;
    float delta2x = k.startPointX - k.endPointX, delta2y = k.startPointY - k.endPointY;
    const float norm_delta2 = std::sqrt(delta2x * delta2x + delta2y * delta2y);
    delta2x /= norm_delta2;
    delta2y /= norm_delta2;
    const float CosSita = std::abs(delta1x * delta2x + delta1y * delta2y);
    if (CosSita < TH) continue;
    vIndices.push_back(i);
  }
  return vIndices;
}
}  // namespace ORB_SLAM2

using namespace ORB_SLAM2;

namespace {
struct Matcher : LSDmatcher {   // FrameBFMatch / lineDescriptorMAD are protected members
  Matcher(float r) : LSDmatcher(r, true) {}
  using LSDmatcher::FrameBFMatch;
  using LSDmatcher::FrameBFMatchNew;
  using LSDmatcher::ComputeF12;
  using LSDmatcher::lineDescriptorMAD;
};
cv::Mat desc_mat(const uint8_t* d, int n) {
  cv::Mat m(n > 0 ? n : 1, 32, CV_8U);
  if (n > 0) std::memcpy(m.data, d, (size_t)n * 32);
  if (n == 0) m = m.rowRange(0, 0);
  return m;
}
cv::Mat identity4() {
  cv::Mat m = cv::Mat::zeros(4, 4, CV_32F);
  for (int i = 0; i < 4; i++) m.at<float>(i, i) = 1.f;
  return m;
}
struct Lines {
  std::vector<std::unique_ptr<MapLine> > all;
  MapLine* make(long id) {
    all.emplace_back(new MapLine());
    all.back()->mnId = (unsigned long)id;
    return all.back().get();
  }
};
void fill_frame(Frame& f, const plo_keyline* kl, const uint8_t* ldesc, const double* fn, int nl, const float gp[6]) {
  f.NL = nl;
  f.mvKeylinesUn.resize(nl);
  f.mvKeyLineFunctions.resize(nl);
  for (int i = 0; i < nl; i++) {
    std::memcpy(&f.mvKeylinesUn[i], &kl[i], sizeof(plo_keyline));
    f.mvKeyLineFunctions[i](0) = fn[3 * i]; f.mvKeyLineFunctions[i](1) = fn[3 * i + 1]; f.mvKeyLineFunctions[i](2) = fn[3 * i + 2];
  }
  f.mLdesc = desc_mat(ldesc, nl);
  f.mvpMapLines.assign(nl, nullptr);
  f.mvbLineOutlier.assign(nl, false);
  f.mTcw = identity4();
  f.lineGrid.kl.assign(kl, kl + nl);
  f.lineGrid.fn.assign(fn, fn + (size_t)nl * 3);
  std::memcpy(f.lineGrid.gp, gp, sizeof(f.lineGrid.gp));
  f.lineGrid.build();
}
}  // namespace

extern "C" {

void ref_line_bfmatch(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float th, float nnratio, int32_t* matches) {
  Matcher m(nnratio);
  std::vector<int> out;
  m.FrameBFMatch(desc_mat(d1, n1), desc_mat(d2, n2), out, th);
  for (int i = 0; i < n1; i++) matches[i] = out[i];
}

int ref_line_search_double(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnratio, int32_t* matches12) {
  Frame a, b;
  a.NL = n1; b.NL = n2;
  a.mLdesc = desc_mat(d1, n1);
  b.mLdesc = desc_mat(d2, n2);
  LSDmatcher m(nnratio, true);
  std::vector<int> out;
  const int n = m.SearchDouble(a, b, out);
  for (int i = 0; i < n1; i++) matches12[i] = i < (int)out.size() ? out[i] : -1;
  return n;
}

// SearchByProjection(Cur, Last, th).  Query i = Last line i: valid = MapLine && !outlier && isInFrustum; seg = mTrackProj*;
// length = Last.mvKeylinesUn[i].lineLength; desc; hasobs.  occupied (in/out), assigned[i2] = query or -1.
int ref_line_search_by_projection_frame(const plo_keyline* kl, const uint8_t* ldesc, const double* fn, int nl, const float gp[6],
                                        uint8_t* occupied, int nq, const uint8_t* q_valid, const float* q_seg, const float* q_length,
                                        const uint8_t* q_desc, const uint8_t* q_hasobs, float th, int32_t* assigned) {
  Lines ls;
  Frame cur, last;
  fill_frame(cur, kl, ldesc, fn, nl, gp);
  for (int i = 0; i < nl; i++)
    if (occupied[i]) { cur.mvpMapLines[i] = ls.make(-1); cur.mvpMapLines[i]->nobs = 1; }
  last.NL = nq;
  last.mvKeylinesUn.resize(nq);
  last.mvpMapLines.assign(nq, nullptr);
  last.mvbLineOutlier.assign(nq, false);
  last.mTcw = identity4();
  for (int i = 0; i < nq; i++) {
    std::memset(&last.mvKeylinesUn[i], 0, sizeof(plo_keyline));
    last.mvKeylinesUn[i].lineLength = q_length[i];
    if (!q_valid[i]) continue;
    MapLine* p = ls.make(i);
    p->mbTrackInView = true;
    p->mTrackProjX1 = q_seg[4 * i]; p->mTrackProjY1 = q_seg[4 * i + 1]; p->mTrackProjX2 = q_seg[4 * i + 2]; p->mTrackProjY2 = q_seg[4 * i + 3];
    p->mLDescriptor = desc_mat(q_desc + (size_t)i * 32, 1);
    p->nobs = q_hasobs[i] ? 1 : 0;
    last.mvpMapLines[i] = p;
  }
  LSDmatcher m(0.9f, true);
  const int n = m.SearchByProjection(cur, last, th);
  for (int i = 0; i < nl; i++) {
    MapLine* p = cur.mvpMapLines[i];
    assigned[i] = (p && (long)p->mnId >= 0) ? (int32_t)p->mnId : -1;
    occupied[i] = (p && p->Observations() > 0) ? 1 : 0;
  }
  return n;
}

// SearchByProjection(F, vpMapLines, th).  Query: valid = mbTrackInView && !isBad(); seg; viewcos; desc; hasobs.
int ref_line_search_by_projection_ml(const plo_keyline* kl, const uint8_t* ldesc, const double* fn, int nl, const float gp[6],
                                     uint8_t* occupied, int nq, const uint8_t* q_valid, const float* q_seg, const float* q_viewcos,
                                     const uint8_t* q_desc, const uint8_t* q_hasobs, float th, float nnratio, int32_t* assigned) {
  Lines ls;
  Frame f;
  fill_frame(f, kl, ldesc, fn, nl, gp);
  for (int i = 0; i < nl; i++)
    if (occupied[i]) { f.mvpMapLines[i] = ls.make(-1); f.mvpMapLines[i]->nobs = 1; }
  std::vector<MapLine*> q(nq);
  for (int i = 0; i < nq; i++) {
    MapLine* p = ls.make(i);
    p->mbTrackInView = q_valid[i] != 0;
    p->mTrackProjX1 = q_seg[4 * i]; p->mTrackProjY1 = q_seg[4 * i + 1]; p->mTrackProjX2 = q_seg[4 * i + 2]; p->mTrackProjY2 = q_seg[4 * i + 3];
    p->mTrackViewCos = q_viewcos[i];
    p->mLDescriptor = desc_mat(q_desc + (size_t)i * 32, 1);
    p->nobs = q_hasobs[i] ? 1 : 0;
    q[i] = p;
  }
  LSDmatcher m(nnratio, true);
  const int n = m.SearchByProjection(f, q, th);
  for (int i = 0; i < nl; i++) {
    MapLine* p = f.mvpMapLines[i];
    assigned[i] = (p && (long)p->mnId >= 0) ? (int32_t)p->mnId : -1;
    occupied[i] = (p && p->Observations() > 0) ? 1 : 0;
  }
  return n;
}

}  // extern "C"

extern "C" {

// LSDmatcher::Fuse(pKF, vpMapLines, th), src/LSDmatcher.cpp:860-1002, KeyFrame pose = identity.  kf_ml[idx]: 0 empty, 1 a
// MapLine with more observations than the queries, 2 a bad one.  cand_desc = pKF->mDescriptors (the matrix :963 reads the
// candidate rows from, with the LINE index).  q_pos = the six world coordinates (start, end).  Outputs: seg_out = (u1, v1, u2,
// v2) as :898-906 computes them, front_out = both endpoints have z >= 0 (otherwise the reference leaves the whole function
// with `return false`, :893-894), inimg_out = both inside the image; best_idx[q] from the GetMapLine(bestIdx) call of query q.
int ref_line_fuse(const plo_keyline* kl, const uint8_t* cand_desc, int nl, const float* scale_factors_line, int nlevels,
                  const uint8_t* kf_ml, const float gp[6], int nq, const uint8_t* q_ml, const uint8_t* q_bad, const uint8_t* q_inkf,
                  const uint8_t* q_inrange, const uint8_t* q_viewok, const float* q_pos, const int32_t* q_level, const uint8_t* q_desc,
                  const float K[4], float th, float* seg_out, uint8_t* front_out, uint8_t* inimg_out, int32_t* best_idx) {
  Lines ls;
  KeyFrame kf;
  kf.NL = nl;
  kf.mvKeyLines.resize(nl);
  for (int i = 0; i < nl; i++) std::memcpy(&kf.mvKeyLines[i], &kl[i], sizeof(plo_keyline));
  kf.mvKeylinesUn = kf.mvKeyLines;
  kf.mDescriptors = desc_mat(cand_desc, nl);
  kf.mLineDescriptors = kf.mDescriptors;
  kf.mvScaleFactorsLine.assign(scale_factors_line, scale_factors_line + nlevels);
  kf.mvpMapLines.assign(nl, nullptr);
  for (int i = 0; i < nl; i++)
    if (kf_ml[i]) { MapLine* p = ls.make(-1); p->nobs = 2; p->bad = kf_ml[i] == 2; kf.mvpMapLines[i] = p; }
  kf.fx = K[0]; kf.fy = K[1]; kf.cx = K[2]; kf.cy = K[3];
  kf.mnMinX = gp[0]; kf.mnMinY = gp[1]; kf.mnMaxX = gp[2]; kf.mnMaxY = gp[3];
  kf.Rcw = cv::Mat::zeros(3, 3, CV_32F);
  for (int i = 0; i < 3; i++) kf.Rcw.at<float>(i, i) = 1.f;
  kf.tcw = cv::Mat::zeros(3, 1, CV_32F);
  kf.Ow = cv::Mat::zeros(3, 1, CV_32F);
  std::vector<MapLine*> q(nq, nullptr);
  for (int i = 0; i < nq; i++) {
    const float* X = q_pos + 6 * i;
    const float invz1 = 1.0f / X[2];
    seg_out[4 * i] = K[0] * X[0] * invz1 + K[2];
    seg_out[4 * i + 1] = K[1] * X[1] * invz1 + K[3];
    const float invz2 = 1.0f / X[5];
    seg_out[4 * i + 2] = K[0] * X[3] * invz2 + K[2];
    seg_out[4 * i + 3] = K[1] * X[4] * invz2 + K[3];
    front_out[i] = (X[2] < 0.0f || X[5] < 0.0f) ? 0 : 1;
    inimg_out[i] = (kf.IsInImage(seg_out[4 * i], seg_out[4 * i + 1]) && kf.IsInImage(seg_out[4 * i + 2], seg_out[4 * i + 3])) ? 1 : 0;
    best_idx[i] = -1;
    if (!q_ml[i]) continue;
    MapLine* p = ls.make(i);
    for (int k = 0; k < 6; k++) p->pos(k) = X[k];
    p->normal(0) = 0; p->normal(1) = 0; p->normal(2) = q_viewok[i] ? 1.0 : -1.0;
    p->mLDescriptor = desc_mat(q_desc + (size_t)i * 32, 1);
    p->bad = q_bad[i] != 0;
    p->predicted = q_level[i];
    if (!q_inrange[i]) p->minDist = 1e29f;
    p->nobs = 1;
    if (q_inkf[i]) p->obs[&kf] = 0;
    q[i] = p;
  }
  LSDmatcher m(0.6f, true);
  const int nf = m.Fuse(&kf, q, th);
  for (const auto& e : kf.lineLog)
    if (e.first && (long)e.first->mnId >= 0) best_idx[e.first->mnId] = e.second;
  return nf;
}

// FrameBFMatchNew(ldesc1, ldesc2, LineMatches, kls1, kls2, kls2func, F, TH), src/LSDmatcher.cpp:488-548.  seg = (startPointX,
// startPointY, endPointX, endPointY) per line, func2 = mvKeyLineFunctions of set 2 (3 doubles per line), F row-major CV_32F.
static std::vector<KeyLine> seg_lines(const float* seg, int n) {
  std::vector<KeyLine> k(n);
  for (int i = 0; i < n; i++) {
    std::memset(&k[i], 0, sizeof(KeyLine));
    k[i].startPointX = seg[4 * i]; k[i].startPointY = seg[4 * i + 1]; k[i].endPointX = seg[4 * i + 2]; k[i].endPointY = seg[4 * i + 3];
  }
  return k;
}
static std::vector<Eigen::Vector3d> line_funcs(const double* fn, int n) {
  std::vector<Eigen::Vector3d> f(n);
  for (int i = 0; i < n; i++) { f[i](0) = fn[3 * i]; f[i](1) = fn[3 * i + 1]; f[i](2) = fn[3 * i + 2]; }
  return f;
}
static cv::Mat mat_f32(const float* v, int r, int c) {
  cv::Mat m(r, c, CV_32F);
  for (int i = 0; i < r; i++)
    for (int j = 0; j < c; j++) m.at<float>(i, j) = v[i * c + j];
  return m;
}
void ref_line_bfmatch_new(const uint8_t* d1, int n1, const uint8_t* d2, int n2, const float* seg1, const float* seg2,
                          const double* func2, const float* F, float th, float nnratio, int32_t* matches) {
  Matcher m(nnratio);
  std::vector<int> out;
  m.FrameBFMatchNew(desc_mat(d1, n1), desc_mat(d2, n2), out, seg_lines(seg1, n1), seg_lines(seg2, n2), line_funcs(func2, n2),
                    mat_f32(F, 3, 3), th);
  for (int i = 0; i < n1; i++) matches[i] = out[i];
}

// SearchForTriangulationNew(pKF1, pKF2, vMatchedPairs, isDouble), src/LSDmatcher.cpp:780-832, on two stand-in KeyFrames: pose = (Rcw
// row-major 3 x 3, tcw 3), K 3 x 3; has_ml = the line carries a MapLine.  F21_out / F12_out = what the reference's ComputeF12 (:834-858;
// Mat::inv is the stand-in's 3 x 3 closed form) gave for (pKF2, pKF1) / (pKF1, pKF2): the matrices the test hands to the oracle and
// to the library, whose C ABI takes them as inputs.
int ref_line_search_for_triangulation_new(const uint8_t* d1, int n1, const uint8_t* d2, int n2, const float* seg1, const float* seg2,
                                          const double* func1, const double* func2, const float* pose1, const float* pose2,
                                          const float* K1, const float* K2, const uint8_t* has_ml1, const uint8_t* has_ml2,
                                          float nnratio, int is_double, int32_t* matches12, float* F21_out, float* F12_out) {
  Lines ls;
  KeyFrame a, b;
  struct { KeyFrame* kf; const uint8_t* d; int n; const float* seg; const double* fn; const float* pose; const float* K; const uint8_t* ml; } in[2] =
      {{&a, d1, n1, seg1, func1, pose1, K1, has_ml1}, {&b, d2, n2, seg2, func2, pose2, K2, has_ml2}};
  for (auto& e : in) {
    e.kf->NL = e.n;
    e.kf->mLineDescriptors = desc_mat(e.d, e.n);
    e.kf->mvKeyLines = seg_lines(e.seg, e.n);
    e.kf->mvKeyLineFunctions = line_funcs(e.fn, e.n);
    e.kf->mvpMapLines.assign(e.n, nullptr);
    for (int i = 0; i < e.n; i++)
      if (e.ml[i]) e.kf->mvpMapLines[i] = ls.make(-1);
    e.kf->Rcw = mat_f32(e.pose, 3, 3);
    e.kf->tcw = mat_f32(e.pose + 9, 3, 1);
    e.kf->mK = mat_f32(e.K, 3, 3);
  }
  Matcher m(nnratio);
  KeyFrame *pa = &a, *pb = &b;
  const cv::Mat F21 = m.ComputeF12(pb, pa), F12 = m.ComputeF12(pa, pb);
  for (int i = 0; i < 9; i++) { F21_out[i] = F21.at<float>(i / 3, i % 3); F12_out[i] = F12.at<float>(i / 3, i % 3); }
  std::vector<int> out;
  const int n = m.SearchForTriangulationNew(&a, &b, out, is_double != 0);
  for (int i = 0; i < n1; i++) matches12[i] = i < (int)out.size() ? out[i] : -1;
  return n;
}

}  // extern "C"
