/*
 * plo.h -- CPU ORACLE for the PL-SLAM front end.  TEST INFRASTRUCTURE ONLY.
 *
 * A dependency-free C++17 restatement of the reference's per-frame hot path, used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the CHECKER.  Nothing in the
 * product path (pl-slam_amd/) may include, link or call this.
 *
 * PARITY, what is pinned and what is not.  The reference (HarborC/PL-SLAM) ships no tests, golden vectors or fixtures for
 * this path and its build needs OpenCV 3.x (+ opencv_contrib line_descriptor) and Eigen3, none of which
 * exist in this image.  Two pieces of the reference DO compile from the sources where they lie once a stand-in for the
 * OpenCV *types* is supplied (oracle/ref/, output oracle/_ref/):
 *   - src/ORBextractor.cc (all of it) on top of this oracle's restated cv::resize / GaussianBlur / FAST / fastAtan2:
 *     the ORB restatement here (orb.cc) is bit-identical to it on every image and parameter set tried
 *     (tests/test_ref_orb.py, goldens tests/golden/ref_orb_*.npz);
 *   - Thirdparty/DBoW2 (FORB::distance, TemplatedVocabulary loader + transform): match.cc's BoW transform and
 *     descriptor distance reproduce it (tests/test_ref_dbow2.py, goldens ref_dbow2_*.npz);
 *   - src/LineExtractor.cpp + the vendored twin of opencv_contrib's line_descriptor (LSDDetector_custom.cpp,
 *     binary_descriptor_custom.cpp) on top of this oracle's LSD / GaussianBlur / Sobel: line.cc's KeyLines, LBD bytes and
 *     line equations are bit-identical to it (KeyLine.angle within one float ulp: an atan2 overload that depends on the
 *     toolchain, see tests/test_ref_line.py; goldens ref_line_*.npz);
 *   - src/ORBmatcher.cc (against stand-ins for Frame / KeyFrame / MapPoint): SearchByBoW (both), SearchForInitialization
 *     SearchByProjection(F, MapPoints), SearchByProjection(Cur, Last, th, bMono), the relocalisation
 *     SearchByProjection(Cur, pKF, found, th, ORBdist), both Fuse overloads, the loop-closing SearchByProjection(pKF, Scw),
 *     SearchBySim3 and SearchForTriangulation of match.cc / frame_search.cc return the same matches
 *     (tests/test_ref_orbmatcher.py, tests/test_ref_orbmatcher_kf.py);
 *   - src/LSDmatcher.cpp (same stand-ins + MapLine; knnMatch = this oracle's knn2): FrameBFMatch / lineDescriptorMAD,
 *     SearchDouble, both SearchByProjection forms and the search inside Fuse of match.cc / frame_search.cc return the same matches
 *     (tests/test_ref_lsdmatcher.py);
 *   - src/MapPoint.cc, src/MapLine.cpp with their own headers: ComputeDistinctiveDescriptors picks the same observation
 *     as plo_distinctive_descriptor (tests/test_ref_mapobj.py);
 *   - src/Frame.cc with its own header: AssignFeaturesToGrid[ForLine], GetFeaturesInArea, GetFeaturesInAreaForLine return
 *     the same cells and the same candidates in the same order as frame_search.cc; so do KeyFrame::GetFeaturesInArea /
 *     GetLinesInArea of a real KeyFrame (src/KeyFrame.cc) built from that Frame (tests/test_ref_frame.py);
 *     Frame::isInFrustum (points and lines, with the real MapPoint / MapLine::PredictScale) for poses without rotation
 *     (tests/test_frustum.py); with a rotation the gemm rounding is this oracle's definition (unpinned);
 *     and Tracking's local-map search on real Frame / MapPoint / MapLine objects (isInFrustum -> SearchByProjection) assigns
 *     the same map elements as the chain of this oracle's functions (tests/test_ref_track.py);
 *   - src/lineIterator.cpp: the line grid of frame_search.cc (tests/test_ref_linegrid.py).
 * PARITY UNPINNED for the rest: the OpenCV primitives themselves (resize, GaussianBlur, FAST, fastAtan2, Sobel, remap,
 * LineSegmentDetector, LineIterator, BFMatcher) are restated from the published OpenCV 3.2-3.4.0 algorithms and are THE
 * definition wherever the reference is ambiguous (SURVEY.md 8c "pinned definitions"); Frame::UndistortKeyPoints
 * (cv::undistortPoints underneath) is a restatement that nothing executable stands behind.
 *
 * Build: see oracle/Makefile  (g++ -O2 -ffp-contract=off: no FMA contraction, IEEE float32).
 */
#ifndef PLO_H
#define PLO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plo_keypoint {   /* == cv::KeyPoint, 28 bytes */
  float x, y, size, angle, response;
  int32_t octave, class_id;
} plo_keypoint;

typedef struct plo_keyline {    /* == cv::line_descriptor::KeyLine, 68 bytes */
  float angle;
  int32_t class_id, octave;
  float pt_x, pt_y;
  float response, size;
  float startPointX, startPointY, endPointX, endPointY;
  float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
  float lineLength;
  int32_t numOfPixels;
} plo_keyline;

/* ---- OpenCV primitives (restated; not under /root/reference) ---- */
int   plo_cv_round_f(float v);                                    /* cvRound: round-half-even */
float plo_fast_atan2(float y, float x);                           /* cv::fastAtan2, degrees */
void  plo_resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstep,
                           uint8_t* dst, int dw, int dh, size_t dstep);       /* cv::resize INTER_LINEAR 8UC1 */
void  plo_resize_linear_u8_scale(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh,
                                 size_t dstep, double inv_scale_x, double inv_scale_y);    /* fx/fy form (LSD) */
void  plo_gaussian_kernel_q8(int ksize, double sigma, int32_t* out);          /* cvRound(getGaussianKernel*256) */
void  plo_gaussian_blur_u8(const uint8_t* src, int w, int h, size_t sstep,
                           uint8_t* dst, size_t dstep, int ksize, double sigma); /* 8U classic path, REFLECT_101 */
int   plo_fast_score(const uint8_t* center, size_t step, int threshold);      /* cornerScore<16> */
int   plo_fast9_16(const uint8_t* img, int w, int h, size_t step, int threshold, int nonmax,
                   plo_keypoint* out, int cap);                               /* cv::FAST TYPE_9_16; returns count */
void  plo_sobel3_s16(const uint8_t* src, int w, int h, size_t sstep, int16_t* dx, int16_t* dy); /* cv::Sobel ksize 3 */
int   plo_pyr_down_u8(const uint8_t* src, int w, int h, size_t sstep, uint8_t* dst, int dw, int dh, size_t dstep); /* cv::pyrDown; -1: OpenCV's size assertion */
void  plo_undistort_maps(const float K[4], const float D[5], int w, int h, float* mapx, float* mapy);
void  plo_remap_linear_u8(const uint8_t* src, int w, int h, size_t sstep, const float* mapx, const float* mapy,
                          uint8_t* dst, size_t dstep);                        /* cv::remap INTER_LINEAR, BORDER_CONSTANT 0 */

/* ---- ORB extractor (reference src/ORBextractor.cc) ---- */
typedef struct plo_orb plo_orb;
plo_orb* plo_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
void     plo_orb_destroy(plo_orb*);
int      plo_orb_levels(const plo_orb*);
void     plo_orb_scale_table(const plo_orb*, int which, float* out);          /* 0 sf, 1 inv sf, 2 sigma2, 3 inv sigma2 */
void     plo_orb_features_per_level(const plo_orb*, int32_t* out);
void     plo_orb_umax(const plo_orb*, int32_t* out16);
/* operator(): returns number of keypoints (<= cap, -1 if cap too small) */
int      plo_orb_extract(plo_orb*, const uint8_t* img, int rows, int cols, size_t step,
                         plo_keypoint* kps, uint8_t* desc, int cap);
/* taps into the last extract call */
int      plo_orb_level_size(const plo_orb*, int level, int* rows, int* cols);
int      plo_orb_read_level(const plo_orb*, int level, uint8_t* out);         /* tightly packed */
int      plo_orb_read_blurred(const plo_orb*, int level, uint8_t* out);
int      plo_orb_read_candidates(const plo_orb*, int level, plo_keypoint* out, int cap); /* level-image coords */

/* ---- Hamming matching (reference src/ORBmatcher.cc, src/LSDmatcher.cpp) ---- */
int  plo_descriptor_distance(const uint8_t* a, const uint8_t* b);
void plo_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist);
void plo_line_mad(const int32_t* dist, int nq, double* nn_mad, double* nn12_mad);
void plo_line_bfmatch(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float th, float nnratio, int32_t* matches);
int  plo_line_search_double(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float th, float nnratio,
                            int32_t* matches12);
/* LSDmatcher::FrameBFMatchNew / SearchForTriangulationNew (src/LSDmatcher.cpp:488-548, 780-832; mutualOverlap :550-625) */
void plo_line_bfmatch_new(const uint8_t* d1, int n1, const uint8_t* d2, int n2, const float* seg1, const float* seg2,
                          const double* func2, const float* F, float th, float nnratio, int32_t* matches);
int  plo_line_search_for_triangulation_new(const uint8_t* d1, int n1, const uint8_t* d2, int n2, const float* seg1,
                                           const float* seg2, const double* func1, const double* func2, const float* F21,
                                           const float* F12, const uint8_t* has_ml1, const uint8_t* has_ml2, float th,
                                           float nnratio, int is_double, int32_t* matches12);
int  plo_orb_search_by_bow(const uint8_t* desc1, const float* angle1, const int32_t* node1, const uint8_t* valid1, int n1,
                           const uint8_t* desc2, const float* angle2, const int32_t* node2, int n2,
                           int th_low, float nnratio, int check_ori, int32_t* matches21);

void plo_bow_transform(const uint8_t* desc, int n, const uint8_t* node_desc, const int32_t* child_start,
                       const int32_t* child_count, const int32_t* word_id, const float* weight, int L, int levelsup,
                       int32_t* nid_out, int32_t* word_out);   /* DBoW2 TemplatedVocabulary::transform */
/* BowVector of TemplatedVocabulary::transform(features, v, fv, levelsup) (TemplatedVocabulary.h:1139-1205) and the two
 * vocabulary file formats (loadFromTextFile :1350-1438, loadFromBinaryFile :1465-1506) -- oracle/bow.cc */
int  plo_bow_vector(const int32_t* word, int n, const double* word_weight, int weighting, int scoring, int32_t* out_word,
                    double* out_value, int cap);
int  plo_vocab_parse_text(const char* path, int32_t header[4], int32_t* parent, uint8_t* is_leaf, uint8_t* desc, double* weight,
                          int cap);
int  plo_vocab_parse_bin(const char* path, int32_t header[4], int32_t* parent, uint8_t* is_leaf, uint8_t* desc, double* weight,
                         int cap);

/* ---- Line extractor (reference src/LineExtractor.cpp + contrib LSDDetector / BinaryDescriptor) ---- */
int  plo_lsd_detect(const uint8_t* img, int w, int h, size_t step, float* segs_xyxy, int cap);  /* cv::LineSegmentDetector (STD) */
int  plo_lsd_detect_ex(const uint8_t* img, int w, int h, size_t step, float* segs_xyxy, int cap, int refine);   /* 1: LSD_REFINE_ADV */
double plo_lsd_nfa(int w, int h, int n, int k, double p);   /* nfa() for an image of w x h scaled pixels */
double plo_lsd_log_gamma(double x);
int  plo_lsd_stage_taps(const uint8_t* img, int w, int h, size_t step, uint8_t* scaled, double* angles, double* modgrad,
                        int32_t* ordered, int* sw_out, int* sh_out);
int  plo_keylines_from_segments(const float* segs, int n, int w, int h, const uint8_t* mask, size_t mstep,
                                plo_keyline* out);                                              /* LSDDetector::detectImpl */
void plo_lbd_compute(const uint8_t* img, int w, int h, size_t step, const plo_keyline* kl, int n, uint8_t* desc32,
                     float* desc_float72);                                                      /* BinaryDescriptor::compute */
int  plo_line_extract(const uint8_t* img, int rows, int cols, size_t step, const uint8_t* mask,
                      unsigned n_lsd_feature, double min_line_length,
                      plo_keyline* keylines, uint8_t* desc, double* linefn, int cap);           /* LINEextractor::operator() */
int  plo_line_extract_ex(const uint8_t* img, int rows, int cols, size_t step, const uint8_t* mask, unsigned n_lsd_feature,
                         double min_line_length, plo_keyline* keylines, uint8_t* desc, double* linefn, int cap, int refine);

/* ---- the whole front end of a batch of frames on std::threads: bench.py's cpu_baseline leg (oracle/frontend.cc) ---- */
double plo_frontend_batch(const uint8_t* frames, int n, int rows, int cols, int nfeatures, int nlevels, int nlines, int refine,
                          const float* mapx, const float* mapy, const uint8_t* node_desc, const int32_t* child_start,
                          const int32_t* child_count, const int32_t* word_id, const float* weight, const double* word_weight, int L,
                          int nthreads, int per_thread, unsigned long long* checksum);

/* LINEextractor(numOctaves, scale, ...): -3 = the reference throws (cv::pyrDown's size assertion: more than one octave needs
 * (int)scale == 2), -4 = undefined behaviour in the reference (three or more octaves); oracle/line.cc */
int  plo_line_extract_oct(const uint8_t* img, int rows, int cols, size_t step, const uint8_t* mask, int num_octaves, float scale,
                          unsigned n_lsd_feature, double min_line_length, plo_keyline* keylines, uint8_t* desc, double* linefn, int cap,
                          int refine);

/* ---- Windowed (grid) searches of the tracking front end (reference src/Frame.cc, ORBmatcher.cc, LSDmatcher.cpp;
 *      oracle/frame_search.cc).  gp = {mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv};
 *      grid = CSR over 64 x 48 cells, cell (ix, iy) -> ix*48 + iy. ---- */
int  plo_frame_assign_grid(const plo_keypoint* kps_un, int n, const float gp[6], int32_t* cell_start, int32_t* cell_items);
int  plo_frame_assign_grid_lines(const plo_keyline* kl, int nl, const float gp[6], int32_t* cell_start, int32_t* cell_items,
                                 int cap);
int  plo_features_in_area(const plo_keypoint* kps_un, const float gp[6], const int32_t* cs, const int32_t* ci, float x, float y,
                          float r, int min_level, int max_level, int32_t* out, int cap);
int  plo_features_in_area_for_line(const plo_keyline* kl, const double* fn, int nl, const float gp[6], const int32_t* cs,
                                   const int32_t* ci, float x1, float y1, float x2, float y2, float r, float TH, int32_t* out,
                                   int cap);
int  plo_orb_search_for_initialization(const plo_keypoint* kps1, const uint8_t* desc1, int n1, const plo_keypoint* kps2,
                                       const uint8_t* desc2, int n2, const float gp2[6], const int32_t* cs2, const int32_t* ci2,
                                       float* prev_matched, int window_size, float nnratio, int check_ori, int32_t* matches12);
int  plo_orb_search_by_projection_mp(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6], const int32_t* cs,
                                     const int32_t* ci, const float* scale_factors, uint8_t* occupied, int nq,
                                     const uint8_t* q_valid, const float* q_xy, const int32_t* q_level, const float* q_viewcos,
                                     const uint8_t* q_desc, const uint8_t* q_hasobs, float th, float nnratio, int32_t* assigned);
int  plo_orb_search_by_projection_frame(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6],
                                        const int32_t* cs, const int32_t* ci, const float* scale_factors, uint8_t* occupied,
                                        int nq, const uint8_t* q_valid, const float* q_uv, const int32_t* q_octave,
                                        const float* q_angle, const uint8_t* q_desc, const uint8_t* q_hasobs, float th, int mode,
                                        int check_ori, int32_t* assigned);
int  plo_line_search_by_projection_frame(const plo_keyline* kl, const uint8_t* ldesc, const double* fn, int nl, const float gp[6],
                                         const int32_t* cs, const int32_t* ci, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                         const float* q_seg, const float* q_length, const uint8_t* q_desc,
                                         const uint8_t* q_hasobs, float th, int32_t* assigned);
int  plo_line_search_by_projection_ml(const plo_keyline* kl, const uint8_t* ldesc, const double* fn, int nl, const float gp[6],
                                      const int32_t* cs, const int32_t* ci, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                      const float* q_seg, const float* q_viewcos, const uint8_t* q_desc, const uint8_t* q_hasobs,
                                      float th, float nnratio, int32_t* assigned);

/* Frame::UndistortKeyPoints (src/Frame.cc:915-945) and MapPoint / MapLine::ComputeDistinctiveDescriptors
 * (src/MapPoint.cc:249-314, src/MapLine.cpp:256-330) */
void plo_undistort_keypoints(const plo_keypoint* kps, int n, const float K[4], const float D[5], plo_keypoint* out);
int  plo_distinctive_descriptor(const uint8_t* desc, int n);
/* ORBmatcher::SearchByBoW(KF, KF) (src/ORBmatcher.cc:574-709) and the relocalisation
 * SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1587-1716) */
int  plo_orb_search_by_bow_kfkf(const uint8_t* desc1, const float* angle1, const int32_t* node1, const uint8_t* valid1, int n1,
                                const uint8_t* desc2, const float* angle2, const int32_t* node2, const uint8_t* valid2, int n2,
                                int th_low, float nnratio, int check_ori, int32_t* matches12);
int  plo_keyframe_lines_in_area(const plo_keyline* kl, int nl, float x1, float y1, float x2, float y2, float r, float TH,
                                int32_t* out, int cap);                  /* KeyFrame::GetLinesInArea, src/KeyFrame.cc:647-683 */
int  plo_line_fuse_search(const plo_keyline* kl, const uint8_t* cand_desc, int nl, const float* scale_factors_line, int nq,
                          const uint8_t* q_valid, const float* q_seg, const int32_t* q_level, const uint8_t* q_desc, float th,
                          float TH, int th_low, int32_t* best_idx);           /* search inside LSDmatcher::Fuse, :860-1002 */
int  plo_orb_fuse_search(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6], const int32_t* cs,
                         const int32_t* ci, const float* scale_factors, const float* inv_level_sigma2, int nq,
                         const uint8_t* q_valid, const float* q_uv, const int32_t* q_level, const uint8_t* q_desc, float th,
                         int th_low, int32_t* best_idx);                      /* search inside ORBmatcher::Fuse, :914-1197 */
int  plo_orb_search_by_projection_sim3(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6],
                                       const int32_t* cs, const int32_t* ci, const float* scale_factors, uint8_t* occupied,
                                       int nq, const uint8_t* q_valid, const float* q_uv, const int32_t* q_level,
                                       const uint8_t* q_desc, float th, int th_low, int32_t* assigned);   /* :329-453 */
int  plo_orb_search_by_sim3(const plo_keypoint* kps1, const uint8_t* desc1, int n1, const int32_t* cs1, const int32_t* ci1,
                            const plo_keypoint* kps2, const uint8_t* desc2, int n2, const int32_t* cs2, const int32_t* ci2,
                            const float gp[6], const float* scale_factors, const uint8_t* q12_valid, const float* q12_uv,
                            const int32_t* q12_level, const uint8_t* q12_desc, const uint8_t* q21_valid, const float* q21_uv,
                            const int32_t* q21_level, const uint8_t* q21_desc, float th, int th_high, int32_t* match1,
                            int32_t* match2, int32_t* match12);            /* src/ORBmatcher.cc:1199-1439 */
int  plo_orb_search_for_triangulation(const plo_keypoint* kps1, const uint8_t* desc1, const int32_t* node1,
                                      const uint8_t* has_mp1, int n1, const plo_keypoint* kps2, const uint8_t* desc2,
                                      const int32_t* node2, const uint8_t* has_mp2, int n2, const float F12[9], float ex,
                                      float ey, const float* scale_factors2, const float* level_sigma2_2, int th_low,
                                      int check_ori, int32_t* matches12);   /* src/ORBmatcher.cc:720-912, monocular */
int  plo_orb_search_by_projection_kf(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6], const int32_t* cs,
                                     const int32_t* ci, const float* scale_factors, uint8_t* occupied, int nq,
                                     const uint8_t* q_valid, const float* q_uv, const int32_t* q_level, const float* q_angle,
                                     const uint8_t* q_desc, const uint8_t* q_hasobs, float th, int orb_dist, int check_ori,
                                     int32_t* assigned);


/* Frame::isInFrustum(MapPoint*, viewingCosLimit) / (MapLine*, ...) (src/Frame.cc:560-623, 625-711) with
 * MapPoint::PredictScale (src/MapPoint.cc:413-428) / MapLine::PredictScale (src/MapLine.cpp:395-404): what Tracking runs on
 * every local map element to produce the queries of SearchByProjection(F, MapPoints / MapLines).  view[24] = Rcw[9] row-major,
 * tcw[3], Ow[3], fx, fy, cx, cy, mnMinX, mnMinY, mnMaxX, mnMaxY, mfLogScaleFactor; nlevels = mnScaleLevels.
 * Pinned definition of the cv::Mat arithmetic inside (OpenCV is not in the tree): `mRcw*P+mtcw` is ONE gemm call with double
 * accumulation and a single rounding to float; cv::norm and Mat::dot accumulate in double; everything else is the float
 * expression as written.  log() of a float resolves to the float overload (as with g++ / libstdc++ here). */
/* The projection the pose-driven searches compute inline before they look anything up.  form 0: ORBmatcher::SearchByProjection
 * (Cur, Last, th, mono) :1474-1484 and the relocalisation form :1614-1622 -- invz = (float)(1.0 / z), u = fx*xc*invz + cx,
 * front = !(invz < 0); form 1: Fuse :945-957 and the loop-closing SearchByProjection :362-375 -- front = !(z < 0),
 * invz = 1 / z in float, x = X*invz, u = fx*x + cx; form 2: Fuse(Scw) :1096-1108, SearchBySim3 :1253-1267 -- as form 1 with
 * invz = (float)(1.0 / z).  Camera coordinates = `R*P + t` as one double-accumulated gemm (see plo_frame_is_in_frustum_*). */
void plo_frame_project_points(const float view[24], int form, int n, const float* pos, uint8_t* front, float* uv);
void plo_map_point_gates(const float view[24], int nlevels, const float R2t2[12], int flags, int n, const float* pos, const float* normal,
                         const float* min_dist_inv, const float* max_dist_inv, const float* max_dist, uint8_t* valid, float* uv,
                         float* dist_out, int32_t* level);
void plo_frame_is_in_frustum_points(const float view[24], int nlevels, int n, const float* pos, const float* normal,
                                    const float* min_dist, const float* max_dist, float viewing_cos_limit, uint8_t* valid,
                                    float* uv, int32_t* level, float* viewcos);
void plo_frame_is_in_frustum_lines(const float view[24], int n, const float* pos6, const float* normal, const float* min_dist,
                                   const float* max_dist, float viewing_cos_limit, uint8_t* valid, float* seg, int32_t* level,
                                   float* viewcos);

#ifdef __cplusplus
}
#endif
#endif
