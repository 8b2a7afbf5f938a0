// ORACLE (test infrastructure) -- CPU restatement of ORB_SLAM2::ORBextractor
// (reference src/ORBextractor.cc, include/ORBextractor.h).  PARITY UNPINNED, see oracle/plo.h.
//
// Pinned definitions where the reference is not a stable target:
//   * quad-tree "largest first" order (ORBextractor.cc:684 sorts pair<int, ExtractorNode*>, i.e. ties are
//     broken by HEAP ADDRESS): here ties are broken by node creation order -- among equal sizes the most
//     recently created node is expanded first.
//   * no FMA contraction in the rBRIEF rotation (build flag -ffp-contract=off), cvRound = round-half-even.
//   * cos/sin of the keypoint angle: correctly rounded float, computed as (float)cos((double)a).
#include "plo.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

namespace {

const int8_t kPattern[1024] = {
#include "../include/plh_orb_pattern.inc"
};

const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;   // ORBextractor.cc:72-74

struct Img {
  int w = 0, h = 0;
  std::vector<uint8_t> px;
  uint8_t* row(int y) { return px.data() + (size_t)y * w; }
  const uint8_t* row(int y) const { return px.data() + (size_t)y * w; }
};

struct KP {   // level-local keypoint
  float x, y, response, angle;
};

// ExtractorNode of ORBextractor.h:37-52 plus a creation id used for the pinned tie-break.
struct Node {
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  std::vector<KP> keys;
  std::list<Node>::iterator lit;
  bool noMore = false;
  long id = 0;
};

// ExtractorNode::DivideNode, ORBextractor.cc:481-537
void divide(const Node& p, Node& n1, Node& n2, Node& n3, Node& n4) {
  const int halfX = (int)std::ceil(static_cast<float>(p.URx - p.ULx) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(p.BRy - p.ULy) / 2);
  n1.ULx = p.ULx;          n1.ULy = p.ULy;
  n1.URx = p.ULx + halfX;  n1.URy = p.ULy;
  n1.BLx = p.ULx;          n1.BLy = p.ULy + halfY;
  n1.BRx = p.ULx + halfX;  n1.BRy = p.ULy + halfY;
  n2.ULx = n1.URx; n2.ULy = n1.URy;
  n2.URx = p.URx;  n2.URy = p.URy;
  n2.BLx = n1.BRx; n2.BLy = n1.BRy;
  n2.BRx = p.URx;  n2.BRy = p.ULy + halfY;
  n3.ULx = n1.BLx; n3.ULy = n1.BLy;
  n3.URx = n1.BRx; n3.URy = n1.BRy;
  n3.BLx = p.BLx;  n3.BLy = p.BLy;
  n3.BRx = n1.BRx; n3.BRy = p.BLy;
  n4.ULx = n3.URx; n4.ULy = n3.URy;
  n4.URx = n2.BRx; n4.URy = n2.BRy;
  n4.BLx = n3.BRx; n4.BLy = n3.BRy;
  n4.BRx = p.BRx;  n4.BRy = p.BRy;
  for (const KP& kp : p.keys) {
    if (kp.x < n1.URx) {
      if (kp.y < n1.BRy) n1.keys.push_back(kp);
      else n3.keys.push_back(kp);
    } else if (kp.y < n1.BRy)
      n2.keys.push_back(kp);
    else
      n4.keys.push_back(kp);
  }
  n1.noMore = n1.keys.size() == 1;
  n2.noMore = n2.keys.size() == 1;
  n3.noMore = n3.keys.size() == 1;
  n4.noMore = n4.keys.size() == 1;
}

}  // namespace

struct plo_orb {
  int nfeatures, nlevels, iniTh, minTh;
  double scaleFactor;   // ORBextractor.h:95 stores it as double
  std::vector<float> sf, isf, sig2, isig2;
  std::vector<int> perLevel, umax;
  std::vector<Img> pyr, blurred;
  std::vector<std::vector<KP>> cands;   // level-image coordinates, quad-tree input order
  long nextId = 0;

  // ORBextractor::ORBextractor, ORBextractor.cc:410-470
  plo_orb(int nf, float s, int nl, int ini, int mn) : nfeatures(nf), nlevels(nl), iniTh(ini), minTh(mn), scaleFactor(s) {
    sf.resize(nl); sig2.resize(nl); isf.resize(nl); isig2.resize(nl);
    sf[0] = 1.0f; sig2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
      sf[i] = (float)(sf[i - 1] * scaleFactor);   // float * double -> double -> float
      sig2[i] = sf[i] * sf[i];
    }
    for (int i = 0; i < nl; i++) { isf[i] = 1.0f / sf[i]; isig2[i] = 1.0f / sig2[i]; }
    perLevel.resize(nl);
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
      perLevel[l] = (int)lrintf(nDesired);
      sum += perLevel[l];
      nDesired *= factor;
    }
    perLevel[nl - 1] = std::max(nfeatures - sum, 0);
    umax.assign(HALF_PATCH_SIZE + 1, 0);
    int v, v0, vmax = (int)std::floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = (int)lrint(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
    pyr.resize(nl); blurred.resize(nl); cands.resize(nl);
  }

  // ORBextractor::ComputePyramid, ORBextractor.cc:1107-1132.  The 19-px reflect-101 frame the reference
  // adds around every level is never read on the monocular path (FAST starts 16 px in, patch radius 15,
  // rBRIEF runs on a border-less clone) and is therefore not materialised.
  void computePyramid(const uint8_t* img, int rows, int cols, size_t step) {
    for (int l = 0; l < nlevels; l++) {
      float scale = isf[l];
      int w = (int)lrintf((float)cols * scale), h = (int)lrintf((float)rows * scale);
      Img& L = pyr[l];
      L.w = w; L.h = h; L.px.assign((size_t)w * h, 0);
      if (l == 0) {
        for (int y = 0; y < rows; y++) memcpy(L.row(y), img + (size_t)y * step, cols);
      } else {
        plo_resize_linear_u8(pyr[l - 1].px.data(), pyr[l - 1].w, pyr[l - 1].h, pyr[l - 1].w, L.px.data(), w, h, w);
      }
    }
  }

  // ORBextractor::DistributeOctTree, ORBextractor.cc:539-763
  std::vector<KP> distribute(const std::vector<KP>& in, int minX, int maxX, int minY, int maxY, int N) {
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<Node> nodes;
    std::vector<Node*> ini(nIni);
    for (int i = 0; i < nIni; i++) {
      Node ni;
      ni.ULx = (int)(hX * static_cast<float>(i));      ni.ULy = 0;
      ni.URx = (int)(hX * static_cast<float>(i + 1));  ni.URy = 0;
      ni.BLx = ni.ULx; ni.BLy = maxY - minY;
      ni.BRx = ni.URx; ni.BRy = maxY - minY;
      ni.id = nextId++;
      nodes.push_back(ni);
      ini[i] = &nodes.back();
    }
    for (const KP& kp : in) ini[(int)(kp.x / hX)]->keys.push_back(kp);
    for (auto lit = nodes.begin(); lit != nodes.end();) {
      if (lit->keys.size() == 1) { lit->noMore = true; ++lit; }
      else if (lit->keys.empty()) lit = nodes.erase(lit);
      else ++lit;
    }
    bool finish = false;
    typedef std::pair<int, std::pair<long, Node*>> SizeNode;   // (size, (creation id, node))
    std::vector<SizeNode> toExpand;
    auto pushChild = [&](Node& c, int* nToExpand) {
      if (c.keys.empty()) return;
      c.id = nextId++;
      nodes.push_front(c);
      if (c.keys.size() > 1) {
        if (nToExpand) ++*nToExpand;
        toExpand.push_back(SizeNode((int)c.keys.size(), std::make_pair(nodes.front().id, &nodes.front())));
        nodes.front().lit = nodes.begin();
      }
    };
    while (!finish) {
      int prevSize = (int)nodes.size();
      int nToExpand = 0;
      toExpand.clear();
      for (auto lit = nodes.begin(); lit != nodes.end();) {
        if (lit->noMore) { ++lit; continue; }
        Node n1, n2, n3, n4;
        divide(*lit, n1, n2, n3, n4);
        pushChild(n1, &nToExpand); pushChild(n2, &nToExpand); pushChild(n3, &nToExpand); pushChild(n4, &nToExpand);
        lit = nodes.erase(lit);
      }
      if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
        finish = true;
      } else if ((int)nodes.size() + nToExpand * 3 > N) {
        while (!finish) {
          prevSize = (int)nodes.size();
          std::vector<SizeNode> prev = toExpand;
          toExpand.clear();
          std::sort(prev.begin(), prev.end(), [](const SizeNode& a, const SizeNode& b) {
            if (a.first != b.first) return a.first < b.first;
            return a.second.first < b.second.first;   // PINNED tie-break: creation order, not heap address
          });
          for (int j = (int)prev.size() - 1; j >= 0; j--) {
            Node* p = prev[j].second.second;
            Node n1, n2, n3, n4;
            divide(*p, n1, n2, n3, n4);
            pushChild(n1, nullptr); pushChild(n2, nullptr); pushChild(n3, nullptr); pushChild(n4, nullptr);
            nodes.erase(p->lit);
            if ((int)nodes.size() >= N) break;
          }
          if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
        }
      }
    }
    std::vector<KP> result;
    result.reserve(nodes.size());
    for (const Node& n : nodes) {
      const KP* best = &n.keys[0];
      float maxResponse = best->response;
      for (size_t k = 1; k < n.keys.size(); k++)
        if (n.keys[k].response > maxResponse) { best = &n.keys[k]; maxResponse = n.keys[k].response; }
      result.push_back(*best);
    }
    return result;
  }

  // ORBextractor::ComputeKeyPointsOctTree, ORBextractor.cc:765-853 (candidates only; the per-level
  // quad-tree and orientation follow in extract()).
  void fastCells(int level, std::vector<KP>& toDistribute, int& minBX, int& maxBX, int& minBY, int& maxBY) {
    const Img& im = pyr[level];
    const float W = 30;
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = im.w - EDGE_THRESHOLD + 3, maxBorderY = im.h - EDGE_THRESHOLD + 3;
    minBX = minBorderX; maxBX = maxBorderX; minBY = minBorderY; maxBY = maxBorderY;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols <= 0 || nRows <= 0) return;   // (reference divides by zero here; guard)
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    std::vector<plo_keypoint> cell((size_t)(wCell + 6) * (hCell + 6));
    for (int i = 0; i < nRows; i++) {
      const float iniY = (float)(minBorderY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBorderY - 3) continue;
      if (maxY > maxBorderY) maxY = (float)maxBorderY;
      for (int j = 0; j < nCols; j++) {
        const float iniX = (float)(minBorderX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBorderX - 6) continue;
        if (maxX > maxBorderX) maxX = (float)maxBorderX;
        const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
        int n = plo_fast9_16(im.row(y0) + x0, cw, ch, im.w, iniTh, 1, cell.data(), (int)cell.size());
        if (n == 0) n = plo_fast9_16(im.row(y0) + x0, cw, ch, im.w, minTh, 1, cell.data(), (int)cell.size());
        for (int k = 0; k < n; k++) {
          KP kp;
          kp.x = cell[k].x + j * wCell;
          kp.y = cell[k].y + i * hCell;
          kp.response = cell[k].response;
          kp.angle = -1.f;
          toDistribute.push_back(kp);
        }
      }
    }
  }

  // IC_Angle, ORBextractor.cc:77-104
  float icAngle(const Img& im, float px, float py) const {
    int m_01 = 0, m_10 = 0;
    const int cx = (int)lrintf(px), cy = (int)lrintf(py);
    const uint8_t* center = im.row(cy) + cx;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    const int step = im.w;
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
      int v_sum = 0, d = umax[v];
      for (int u = -d; u <= d; ++u) {
        int val_plus = center[u + v * step], val_minus = center[u - v * step];
        v_sum += (val_plus - val_minus);
        m_10 += u * (val_plus + val_minus);
      }
      m_01 += v * v_sum;
    }
    return plo_fast_atan2((float)m_01, (float)m_10);
  }

  // computeOrbDescriptor, ORBextractor.cc:108-147
  void descriptor(const Img& im, const KP& kp, uint8_t* desc) const {
    const float factorPI = (float)(M_PI / 180.f);
    float angle = kp.angle * factorPI;
    float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
    const uint8_t* center = im.row((int)lrintf(kp.y)) + (int)lrintf(kp.x);
    const int step = im.w;
    const int8_t* pat = kPattern;
    auto get = [&](int idx) -> int {
      float px = (float)pat[idx * 2], py = (float)pat[idx * 2 + 1];
      int yy = (int)lrintf(px * b + py * a);
      int xx = (int)lrintf(px * a - py * b);
      return center[yy * step + xx];
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
      int val = 0;
      for (int k = 0; k < 8; k++) {
        int t0 = get(2 * k), t1 = get(2 * k + 1);
        val |= (t0 < t1) << k;
      }
      desc[i] = (uint8_t)val;
    }
  }

  // ORBextractor::operator(), ORBextractor.cc:1043-1105
  int extract(const uint8_t* img, int rows, int cols, size_t step, plo_keypoint* kps, uint8_t* desc, int cap) {
    if (rows <= 0 || cols <= 0 || !img) return 0;
    computePyramid(img, rows, cols, step);
    std::vector<std::vector<KP>> all(nlevels);
    for (int l = 0; l < nlevels; l++) {
      std::vector<KP> toDist;
      int minBX = 0, maxBX = 0, minBY = 0, maxBY = 0;
      fastCells(l, toDist, minBX, maxBX, minBY, maxBY);
      cands[l].clear();
      for (const KP& k : toDist) { KP c = k; c.x += minBX; c.y += minBY; cands[l].push_back(c); }
      std::vector<KP>& out = all[l];
      if (!toDist.empty() || true) out = distribute(toDist, minBX, maxBX, minBY, maxBY, perLevel[l]);
      for (KP& k : out) { k.x += minBX; k.y += minBY; }
    }
    for (int l = 0; l < nlevels; l++)
      for (KP& k : all[l]) k.angle = icAngle(pyr[l], k.x, k.y);
    int n = 0;
    for (int l = 0; l < nlevels; l++) n += (int)all[l].size();
    if (n > cap) return -1;
    int off = 0;
    for (int l = 0; l < nlevels; l++) {
      blurred[l].w = pyr[l].w; blurred[l].h = pyr[l].h; blurred[l].px.clear();
      if (all[l].empty()) continue;
      Img& B = blurred[l];
      B.px.resize(pyr[l].px.size());
      plo_gaussian_blur_u8(pyr[l].px.data(), B.w, B.h, B.w, B.px.data(), B.w, 7, 2.0);
      const int scaledPatchSize = (int)(PATCH_SIZE * sf[l]);
      for (const KP& k : all[l]) {
        descriptor(B, k, desc + (size_t)off * 32);
        plo_keypoint o;
        o.x = k.x; o.y = k.y;
        if (l != 0) { o.x *= sf[l]; o.y *= sf[l]; }
        o.size = (float)scaledPatchSize;
        o.angle = k.angle;
        o.response = k.response;
        o.octave = l;
        o.class_id = -1;
        kps[off++] = o;
      }
    }
    return n;
  }
};

extern "C" {

plo_orb* plo_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
  if (nlevels < 1 || nfeatures < 0) return nullptr;
  return new plo_orb(nfeatures, scale_factor, nlevels, ini_th, min_th);
}
void plo_orb_destroy(plo_orb* h) { delete h; }
int plo_orb_levels(const plo_orb* h) { return h->nlevels; }
void plo_orb_scale_table(const plo_orb* h, int which, float* out) {
  const std::vector<float>& v = which == 0 ? h->sf : which == 1 ? h->isf : which == 2 ? h->sig2 : h->isig2;
  std::copy(v.begin(), v.end(), out);
}
void plo_orb_features_per_level(const plo_orb* h, int32_t* out) { std::copy(h->perLevel.begin(), h->perLevel.end(), out); }
void plo_orb_umax(const plo_orb* h, int32_t* out16) { std::copy(h->umax.begin(), h->umax.end(), out16); }
int plo_orb_extract(plo_orb* h, const uint8_t* img, int rows, int cols, size_t step, plo_keypoint* kps, uint8_t* desc,
                    int cap) {
  return h->extract(img, rows, cols, step, kps, desc, cap);
}
int plo_orb_level_size(const plo_orb* h, int level, int* rows, int* cols) {
  if (level < 0 || level >= h->nlevels) return -1;
  *rows = h->pyr[level].h; *cols = h->pyr[level].w;
  return 0;
}
int plo_orb_read_level(const plo_orb* h, int level, uint8_t* out) {
  if (level < 0 || level >= h->nlevels) return -1;
  memcpy(out, h->pyr[level].px.data(), h->pyr[level].px.size());
  return 0;
}
int plo_orb_read_blurred(const plo_orb* h, int level, uint8_t* out) {
  if (level < 0 || level >= h->nlevels || h->blurred[level].px.empty()) return -1;
  memcpy(out, h->blurred[level].px.data(), h->blurred[level].px.size());
  return 0;
}
int plo_orb_read_candidates(const plo_orb* h, int level, plo_keypoint* out, int cap) {
  if (level < 0 || level >= h->nlevels) return -1;
  const auto& c = h->cands[level];
  for (size_t i = 0; i < c.size() && (int)i < cap; i++) {
    out[i].x = c[i].x; out[i].y = c[i].y; out[i].size = 7.f; out[i].angle = -1.f;
    out[i].response = c[i].response; out[i].octave = level; out[i].class_id = -1;
  }
  return (int)c.size();
}

}  // extern "C"
