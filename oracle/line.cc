// ORACLE (test infrastructure) -- CPU restatement of the line half of the front end:
//   LINEextractor::operator()                     reference src/LineExtractor.cpp:26-93
//   LSDDetector::detectImpl (KeyLine fill, mask)   Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:76-215
//   BinaryDescriptor ctor / computeSobel / compute / computeLBD / binaryConversion
//                                                  Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp:74-116,
//                                                  217-259, 338-412, 524-687, 1026-1372
// (the vendored copy is the only in-tree source of the contrib module the reference links).
// PARITY UNPINNED, see oracle/plo.h.  Pinned: cos/sin/atan2 of float arguments are evaluated in double and
// rounded to float; no FMA; std::sort of lines by response replaced by a stable sort (ties keep detection order).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "plo.h"

namespace {

const int NUM_OF_BANDS = 9, WIDTH_OF_BAND = 7;

// static const int combinations[32][2], binary_descriptor_custom.cpp:74-107: all pairs (i<j) of the 9 bands
// except (0,7),(0,8),(1,7),(1,8) -- generated, then checked against the known first/last entries.
struct Comb {
  int c[32][2];
  Comb() {
    int n = 0;
    for (int i = 0; i < 9; i++)
      for (int j = i + 1; j < 9; j++) {
        if ((i == 0 || i == 1) && (j == 7 || j == 8)) continue;
        c[n][0] = i; c[n][1] = j; n++;
      }
  }
};
const Comb kComb;

inline int cv_round_f(float v) { return (int)lrintf(v); }

}  // namespace

extern "C" {

// LSDDetector::detectImpl's KeyLine fill for the segments of ONE octave (LSDDetector_custom.cpp:161-199): w x h = that octave's
// image, octaveScale = pow((float)scale, octaveIdx), class ids continue from class0; the mask (full resolution, mstep pitch) is
// looked up at the scaled end points (:202-213).  Returns the number of KeyLines written.
static int keylines_of_octave(const float* segs, int n, int w, int h, int octave, float octaveScale, int class0, const uint8_t* mask,
                              size_t mstep, plo_keyline* out) {
  int m = 0;
  for (int k = 0; k < n; k++) {
    float e[4] = {segs[k * 4], segs[k * 4 + 1], segs[k * 4 + 2], segs[k * 4 + 3]};
    // checkLineExtremes
    if (e[0] < 0) e[0] = 0;
    if (e[0] >= w) e[0] = (float)w - 1.0f;
    if (e[2] < 0) e[2] = 0;
    if (e[2] >= w) e[2] = (float)w - 1.0f;
    if (e[1] < 0) e[1] = 0;
    if (e[1] >= h) e[1] = (float)h - 1.0f;
    if (e[3] < 0) e[3] = 0;
    if (e[3] >= h) e[3] = (float)h - 1.0f;
    plo_keyline kl;
    kl.startPointX = e[0] * octaveScale; kl.startPointY = e[1] * octaveScale;
    kl.endPointX = e[2] * octaveScale;   kl.endPointY = e[3] * octaveScale;
    kl.sPointInOctaveX = e[0]; kl.sPointInOctaveY = e[1];
    kl.ePointInOctaveX = e[2]; kl.ePointInOctaveY = e[3];
    kl.lineLength = (float)std::sqrt(std::pow((double)(e[0] - e[2]), 2) + std::pow((double)(e[1] - e[3]), 2));
    // cv::LineIterator(img, Point2f -> Point (cvRound), ...).count for in-image endpoints
    const int x1 = cv_round_f(e[0]), y1 = cv_round_f(e[1]), x2 = cv_round_f(e[2]), y2 = cv_round_f(e[3]);
    kl.numOfPixels = std::max(std::abs(x2 - x1), std::abs(y2 - y1)) + 1;
    kl.angle = (float)std::atan2((double)(kl.endPointY - kl.startPointY), (double)(kl.endPointX - kl.startPointX));
    kl.class_id = class0 + k;
    kl.octave = octave;
    kl.size = (kl.endPointX - kl.startPointX) * (kl.endPointY - kl.startPointY);
    kl.response = kl.lineLength / (float)std::max(w, h);
    kl.pt_x = (kl.endPointX + kl.startPointX) / 2;
    kl.pt_y = (kl.endPointY + kl.startPointY) / 2;
    // mask: drop only if BOTH endpoints sit on mask == 0 (LSDDetector_custom.cpp:202-213)
    if (mask) {
      if (mask[(size_t)(int)kl.startPointY * mstep + (int)kl.startPointX] == 0 &&
          mask[(size_t)(int)kl.endPointY * mstep + (int)kl.endPointX] == 0)
        continue;
    }
    out[m++] = kl;
  }
  return m;
}

// LSDDetector::detectImpl, single octave (scale int -> 1, numOctaves 1): Vec4f -> KeyLine + mask filter.
int plo_keylines_from_segments(const float* segs, int n, int w, int h, const uint8_t* mask, size_t mstep,
                               plo_keyline* out) {
  return keylines_of_octave(segs, n, w, h, 0, 1.0f, 0, mask, mstep, out);
}

struct OctaveGrad {   // dxImg_vector[o] / dyImg_vector[o] and images_sizes[o] of BinaryDescriptor::computeSobel
  std::vector<int16_t> dx, dy;
  int w, h;
};
static void lbd_compute_octaves(const std::vector<OctaveGrad>& oct, const plo_keyline* kls, int n, uint8_t* desc32, float* desc_float72);

// BinaryDescriptor::compute on one octave: GaussianBlur 5x5 sigma 1 -> Sobel dx/dy (int16) -> computeLBD ->
// 32-byte binary conversion.  desc_float72 (optional) receives the 72-float LBD.
void plo_lbd_compute(const uint8_t* img, int w, int h, size_t step, const plo_keyline* kls, int n, uint8_t* desc32,
                     float* desc_float72) {
  std::vector<uint8_t> blur((size_t)w * h);
  plo_gaussian_blur_u8(img, w, h, step, blur.data(), w, 5, 1.0);
  std::vector<OctaveGrad> oct(1);
  oct[0].w = w; oct[0].h = h;
  oct[0].dx.resize((size_t)w * h); oct[0].dy.resize((size_t)w * h);
  plo_sobel3_s16(blur.data(), w, h, w, oct[0].dx.data(), oct[0].dy.data());
  lbd_compute_octaves(oct, kls, n, desc32, desc_float72);
}

// computeLBD (binary_descriptor_custom.cpp:1026-1372): every line on the gradient images of ITS octave (:1080-1104)
static void lbd_compute_octaves(const std::vector<OctaveGrad>& oct, const plo_keyline* kls, int n, uint8_t* desc32, float* desc_float72) {
  // constructor weights (note the integer divisions in the originals)
  double gaussCoefL[WIDTH_OF_BAND * 3], gaussCoefG[NUM_OF_BANDS * WIDTH_OF_BAND];
  {
    double u = (WIDTH_OF_BAND * 3 - 1) / 2;
    double sigma = (WIDTH_OF_BAND * 2 + 1) / 2;
    double invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < WIDTH_OF_BAND * 3; i++) { double dis = i - u; gaussCoefL[i] = std::exp(dis * dis * invsigma2); }
    u = (NUM_OF_BANDS * WIDTH_OF_BAND - 1) / 2;
    sigma = u;
    invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < NUM_OF_BANDS * WIDTH_OF_BAND; i++) { double dis = i - u; gaussCoefG[i] = std::exp(dis * dis * invsigma2); }
  }

  const short heightOfLSP = (short)(WIDTH_OF_BAND * NUM_OF_BANDS);
  const short descriptor_size = NUM_OF_BANDS * 8;
  const short halfHeight = (heightOfLSP - 1) / 2;

  for (int li = 0; li < n; li++) {
    const plo_keyline& L = kls[li];
    const OctaveGrad& G = oct[(size_t)L.octave];
    const int16_t *dxImg = G.dx.data(), *dyImg = G.dy.data();
    const short realWidth = (short)G.w;
    const short imageWidth = realWidth - 1;
    const short imageHeight = (short)(G.h - 1);
    float pgdLBandSum[NUM_OF_BANDS] = {0}, ngdLBandSum[NUM_OF_BANDS] = {0}, pgdL2BandSum[NUM_OF_BANDS] = {0},
          ngdL2BandSum[NUM_OF_BANDS] = {0}, pgdOBandSum[NUM_OF_BANDS] = {0}, ngdOBandSum[NUM_OF_BANDS] = {0},
          pgdO2BandSum[NUM_OF_BANDS] = {0}, ngdO2BandSum[NUM_OF_BANDS] = {0};
    const short lengthOfLSP = (short)L.numOfPixels;
    const short halfWidth = (lengthOfLSP - 1) / 2;
    const float lineMiddlePointX = (float)(0.5 * (L.sPointInOctaveX + L.ePointInOctaveX));
    const float lineMiddlePointY = (float)(0.5 * (L.sPointInOctaveY + L.ePointInOctaveY));
    float dL[2], dO[2];
    dL[0] = (float)std::cos((double)L.angle);
    dL[1] = (float)std::sin((double)L.angle);
    dO[0] = -dL[1];
    dO[1] = dL[0];
    float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
    float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
    for (short hID = 0; hID < heightOfLSP; hID++) {
      float sCorX = sCorX0, sCorY = sCorY0;
      float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
      for (short wID = 0; wID < lengthOfLSP; wID++) {
        short tempCor = (short)std::round(sCorX);
        const short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
        tempCor = (short)std::round(sCorY);
        const short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
        const short dx = dxImg[(size_t)yCor * realWidth + xCor];
        const short dy = dyImg[(size_t)yCor * realWidth + xCor];
        const float gDL = dx * dL[0] + dy * dL[1];
        const float gDO = dx * dO[0] + dy * dO[1];
        if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
        if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
        sCorX += dL[0];
        sCorY += dL[1];
      }
      sCorX0 -= dL[1];
      sCorY0 += dL[0];
      float coefInGaussion = (float)gaussCoefG[hID];
      pgdLRowSum = coefInGaussion * pgdLRowSum;
      ngdLRowSum = coefInGaussion * ngdLRowSum;
      const float pgdL2RowSum = pgdLRowSum * pgdLRowSum;
      const float ngdL2RowSum = ngdLRowSum * ngdLRowSum;
      pgdORowSum = coefInGaussion * pgdORowSum;
      ngdORowSum = coefInGaussion * ngdORowSum;
      const float pgdO2RowSum = pgdORowSum * pgdORowSum;
      const float ngdO2RowSum = ngdORowSum * ngdORowSum;
      auto addBand = [&](short bandID, float c) {
        pgdLBandSum[bandID] += c * pgdLRowSum;
        ngdLBandSum[bandID] += c * ngdLRowSum;
        pgdL2BandSum[bandID] += c * c * pgdL2RowSum;
        ngdL2BandSum[bandID] += c * c * ngdL2RowSum;
        pgdOBandSum[bandID] += c * pgdORowSum;
        ngdOBandSum[bandID] += c * ngdORowSum;
        pgdO2BandSum[bandID] += c * c * pgdO2RowSum;
        ngdO2BandSum[bandID] += c * c * ngdO2RowSum;
      };
      short bandID = (short)(hID / WIDTH_OF_BAND);
      addBand(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND + WIDTH_OF_BAND]);
      bandID--;
      if (bandID >= 0) addBand(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND + 2 * WIDTH_OF_BAND]);
      bandID = bandID + 2;
      if (bandID < NUM_OF_BANDS) addBand(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND]);
    }
    float desVec[NUM_OF_BANDS * 8];
    const float invN2 = (float)(1.0 / (WIDTH_OF_BAND * 2.0));
    const float invN3 = (float)(1.0 / (WIDTH_OF_BAND * 3.0));
    for (short bandID = 0; bandID < NUM_OF_BANDS; bandID++) {
      const float invN = (bandID == 0 || bandID == NUM_OF_BANDS - 1) ? invN2 : invN3;
      const short desID = bandID * 8;
      float temp = pgdLBandSum[bandID] * invN;
      desVec[desID] = temp;
      desVec[desID + 4] = std::sqrt(pgdL2BandSum[bandID] * invN - temp * temp);
      temp = ngdLBandSum[bandID] * invN;
      desVec[desID + 1] = temp;
      desVec[desID + 5] = std::sqrt(ngdL2BandSum[bandID] * invN - temp * temp);
      temp = pgdOBandSum[bandID] * invN;
      desVec[desID + 2] = temp;
      desVec[desID + 6] = std::sqrt(pgdO2BandSum[bandID] * invN - temp * temp);
      temp = ngdOBandSum[bandID] * invN;
      desVec[desID + 3] = temp;
      desVec[desID + 7] = std::sqrt(ngdO2BandSum[bandID] * invN - temp * temp);
    }
    float tempM = 0, tempS = 0;
    for (int b = 0; b < NUM_OF_BANDS; b++) {
      const int i = b * 8;
      tempM += desVec[i] * desVec[i];
      tempM += desVec[i + 1] * desVec[i + 1];
      tempM += desVec[i + 2] * desVec[i + 2];
      tempM += desVec[i + 3] * desVec[i + 3];
      tempS += desVec[i + 4] * desVec[i + 4];
      tempS += desVec[i + 5] * desVec[i + 5];
      tempS += desVec[i + 6] * desVec[i + 6];
      tempS += desVec[i + 7] * desVec[i + 7];
    }
    tempM = 1 / std::sqrt(tempM);
    tempS = 1 / std::sqrt(tempS);
    for (int b = 0; b < NUM_OF_BANDS; b++) {
      const int i = b * 8;
      for (int k = 0; k < 4; k++) desVec[i + k] = desVec[i + k] * tempM;
      for (int k = 4; k < 8; k++) desVec[i + k] = desVec[i + k] * tempS;
    }
    for (short i = 0; i < descriptor_size; i++)
      if (desVec[i] > 0.4) desVec[i] = (float)0.4;
    float temp = 0;
    for (short i = 0; i < descriptor_size; i++) temp += desVec[i] * desVec[i];
    temp = 1 / std::sqrt(temp);
    for (short i = 0; i < descriptor_size; i++) desVec[i] = desVec[i] * temp;
    if (desc_float72) memcpy(desc_float72 + (size_t)li * 72, desVec, sizeof(desVec));
    // binaryConversion over the 32 band pairs
    for (int comb = 0; comb < 32; comb++) {
      const float* f1 = &desVec[8 * kComb.c[comb][0]];
      const float* f2 = &desVec[8 * kComb.c[comb][1]];
      uint8_t result = 0;
      for (int i = 0; i < 8; i++)
        if (f1[i] > f2[i]) result += (uint8_t)(1 << i);
      desc32[(size_t)li * 32 + comb] = result;
    }
  }
}

// LINEextractor::operator().  Returns the number of keylines (<= cap) or -1 on a mask-size error
// (the reference throws std::runtime_error).  The caller passes mask == NULL for "empty Mat".
int plo_line_extract(const uint8_t* img, int rows, int cols, size_t step, const uint8_t* mask, unsigned nLSDFeature,
                     double min_line_length, plo_keyline* keylines, uint8_t* desc, double* linefn, int cap) {
  return plo_line_extract_ex(img, rows, cols, step, mask, nLSDFeature, min_line_length, keylines, desc, linefn, cap, 0);
}
// ... with the refine level of the LineSegmentDetector behind LSDDetector (0 = LSD_REFINE_STD, 1 = LSD_REFINE_ADV; lsd.cc)
int plo_line_extract_ex(const uint8_t* img, int rows, int cols, size_t step, const uint8_t* mask, unsigned nLSDFeature,
                        double min_line_length, plo_keyline* keylines, uint8_t* desc, double* linefn, int cap, int refine) {
  return plo_line_extract_oct(img, rows, cols, step, mask, 1, 1.2f, nLSDFeature, min_line_length, keylines, desc, linefn, cap, refine);
}

// ... and with LINEextractor's numOctaves / scale (LineExtractor.cpp:5-24, 39-40).  What the reference does with them:
//   * lsd->detect(image, keylines, scale, numOctaves, mask) takes `int scale`: the float truncates (1.2 -> 1, 2.0 -> 2);
//   * computeGaussianPyramid pyrDown()s to Size(cols / scale, rows / scale) per octave, and cv::pyrDown asserts
//     |2 dst - src| <= 2 per axis: with more than one octave only (int)scale == 2 gets past it (-3 here: the reference throws);
//   * BinaryDescriptor::computeImpl then sizes its per-line vectors by the highest octave and erases the unused slots while it
//     iterates over them (binary_descriptor_custom.cpp:617-626): with three or more octaves a slot is skipped, a default-constructed
//     line is described and the map lookup behind it runs off the end -- undefined behaviour (-4 here).
// So the one multi-octave configuration with defined behaviour is numOctaves == 2 with scale in [2, 3): octave 1 is pyrDown of the
// image (detection) resp. of the 5x5-blurred image (description), half size, octaveScale 2.
int plo_line_extract_oct(const uint8_t* img, int rows, int cols, size_t step, const uint8_t* mask, int numOctaves, float scale,
                         unsigned nLSDFeature, double min_line_length, plo_keyline* keylines, uint8_t* desc, double* linefn, int cap,
                         int refine) {
  if (!img || rows <= 0 || cols <= 0) return 0;
  if (numOctaves < 1) return -3;
  if (numOctaves > 2) return -4;
  const int iscale = (int)scale;
  std::vector<std::vector<uint8_t>> pyr((size_t)numOctaves);
  std::vector<int> pw((size_t)numOctaves), ph((size_t)numOctaves);
  pw[0] = cols; ph[0] = rows;
  pyr[0].resize((size_t)rows * cols);
  for (int y = 0; y < rows; y++) memcpy(&pyr[0][(size_t)y * cols], img + (size_t)y * step, (size_t)cols);
  for (int o = 1; o < numOctaves; o++) {
    if (iscale <= 0) return -3;
    pw[o] = pw[o - 1] / iscale; ph[o] = ph[o - 1] / iscale;
    pyr[o].resize((size_t)std::max(pw[o], 0) * std::max(ph[o], 0));
    if (plo_pyr_down_u8(pyr[o - 1].data(), pw[o - 1], ph[o - 1], (size_t)pw[o - 1], pyr[o].data(), pw[o], ph[o], (size_t)pw[o]) != 0) return -3;
  }
  std::vector<plo_keyline> kls;
  int class0 = 0;
  for (int o = 0; o < numOctaves; o++) {
    std::vector<float> segs((size_t)4 * 20000);
    int ns = plo_lsd_detect_ex(pyr[o].data(), pw[o], ph[o], (size_t)pw[o], segs.data(), 20000, refine);
    if (ns > 20000) ns = 20000;
    std::vector<plo_keyline> ko((size_t)std::max(ns, 1));
    const float octaveScale = (float)std::pow((float)iscale, o);
    const int m = keylines_of_octave(segs.data(), ns, pw[o], ph[o], o, octaveScale, class0, mask, (size_t)cols, ko.data());
    kls.insert(kls.end(), ko.begin(), ko.begin() + m);
    class0 += ns;   // (class_counter runs over every detected segment, masked or not)
  }
  int n = (int)kls.size();
  kls.resize(n);
  // sort(_keylines, sort_lines_by_response()) -- PINNED stable
  std::stable_sort(kls.begin(), kls.end(), [](const plo_keyline& a, const plo_keyline& b) { return a.response > b.response; });
  if (n == 0) return 0;   // (reference: _keylines[total-1] with total == 0 is UB; guarded)
  int total, index;
  if ((unsigned)n > nLSDFeature) { total = (int)nLSDFeature; index = (int)nLSDFeature; }
  else { total = n; index = n; }
  if (total >= 1 && kls[total - 1].lineLength < min_line_length) {
    for (int i = 0; i < total - 1; i++) {
      if (kls[i].lineLength >= min_line_length && kls[i + 1].lineLength < min_line_length) { index = i; break; }
    }
  }
  // _keylines.resize(index + 1): keeps nLSDFeature+1 lines when more are available; resize() beyond size()
  // would value-initialise a KeyLine in the reference -- here clamped to the detected lines.
  int keep = std::min(index + 1, n);
  if (total == 0) keep = std::min(1, n);
  if (keep > cap) return -2;
  kls.resize(keep);
  for (int i = 0; i < keep; i++) kls[i].class_id = i;
  {   // BinaryDescriptor::compute -> computeSobel(image, highest octave + 1): GaussianBlur 5x5 sigma 1, pyrDown by reductionRatio = 2
      // per further octave (binary_descriptor_custom.cpp:350-398), Sobel of every level
    int top = 0;
    for (int i = 0; i < keep; i++) top = std::max(top, (int)kls[i].octave);
    std::vector<OctaveGrad> oct((size_t)top + 1);
    std::vector<uint8_t> cur((size_t)rows * cols);
    plo_gaussian_blur_u8(img, cols, rows, step, cur.data(), cols, 5, 1.0);
    int cw = cols, ch = rows;
    for (int o = 0; o <= top; o++) {
      if (o > 0) {
        const int nw = cw / 2, nh = ch / 2;
        std::vector<uint8_t> nxt((size_t)nw * nh);
        if (plo_pyr_down_u8(cur.data(), cw, ch, (size_t)cw, nxt.data(), nw, nh, (size_t)nw) != 0) return -3;
        cur.swap(nxt); cw = nw; ch = nh;
      }
      oct[o].w = cw; oct[o].h = ch;
      oct[o].dx.resize((size_t)cw * ch); oct[o].dy.resize((size_t)cw * ch);
      plo_sobel3_s16(cur.data(), cw, ch, (size_t)cw, oct[o].dx.data(), oct[o].dy.data());
    }
    lbd_compute_octaves(oct, kls.data(), keep, desc, nullptr);
  }
  for (int i = 0; i < keep; i++) {
    const plo_keyline& k = kls[i];
    // sp x ep with homogeneous 1.0, normalised by sqrt(l0^2 + l1^2) (Eigen Vector3d)
    const double sx = k.startPointX, sy = k.startPointY, ex = k.endPointX, ey = k.endPointY;
    double l0 = sy * 1.0 - 1.0 * ey;
    double l1 = 1.0 * ex - sx * 1.0;
    double l2 = sx * ey - sy * ex;
    const double nrm = std::sqrt(l0 * l0 + l1 * l1);
    linefn[i * 3 + 0] = l0 / nrm; linefn[i * 3 + 1] = l1 / nrm; linefn[i * 3 + 2] = l2 / nrm;
    keylines[i] = k;
  }
  return keep;
}

}  // extern "C"
