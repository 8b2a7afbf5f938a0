/*
 * plslam_hip.h -- C ABI of the MI355X-native PL-SLAM front end (libplslam_hip.so).
 *
 * This is the drop-in boundary for the per-frame hot path of HarborC/PL-SLAM:
 *   ORBextractor::operator()      reference src/ORBextractor.cc:1043-1105
 *   LINEextractor::operator()     reference src/LineExtractor.cpp:26-93
 *   ORBmatcher  Hamming searches  reference src/ORBmatcher.cc:187-327, 455-572, 1764-1780
 *   LSDmatcher  brute-force 2-NN  reference src/LSDmatcher.cpp:375-486, 627-670
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch / OpenCV types cross this boundary.
 *   - every function returns a plh_status (0 = ok); nothing throws.
 *   - "_dev" entry points take DEVICE pointers and a HIP stream (void* = hipStream_t) and
 *     enqueue work asynchronously on that stream; the caller synchronises.  The non-"_dev"
 *     entry points take HOST buffers, stage them over PCIe and block until results are back:
 *     they are the direct replacement for one reference call (see pl-slam_amd/adaptor/).
 *   - batch layout is "frame-major": B frames, each a contiguous rows*cols u8 plane at
 *     stride `frame_stride` bytes; results are fixed-stride records (capacity per frame is
 *     queried from the handle) plus an int32 count per frame.
 *   - a handle is NOT re-entrant (same rule as the reference's ORBextractor, which keeps
 *     mvImagePyramid); different handles may be used concurrently from different host
 *     threads (Frame.cc:224-225 runs the ORB and line extractors on two threads).
 */
#ifndef PLSLAM_HIP_H
#define PLSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLH_API __attribute__((visibility("default")))

typedef enum plh_status {
  PLH_OK = 0,
  PLH_ERR_INVALID = 1,    /* bad argument (null, size out of the handle's plan, ...) */
  PLH_ERR_NO_DEVICE = 2,  /* no HIP device / hipSetDevice failed */
  PLH_ERR_HIP = 3,        /* a HIP runtime call failed; see plh_last_error() */
  PLH_ERR_CAPACITY = 4,   /* an internal fixed-capacity buffer overflowed (reported, never silent) */
  PLH_ERR_ALLOC = 5
} plh_status;

/* Layout-identical to cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id. */
typedef struct plh_keypoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} plh_keypoint;

/* Layout-identical to cv::line_descriptor::KeyLine (68 bytes),
 * field order per Thirdparty/line_descriptor/include/line_descriptor/descriptor_custom.hpp:105-175. */
typedef struct plh_keyline {
  float angle;
  int32_t class_id, octave;
  float pt_x, pt_y;
  float response, size;
  float startPointX, startPointY, endPointX, endPointY;
  float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
  float lineLength;
  int32_t numOfPixels;
} plh_keyline;

PLH_API const char* plh_last_error(void);     /* thread-local message of the last failing call */
PLH_API const char* plh_version(void);
PLH_API int plh_device_count(void);
/* Device-side self test: every gfx950 instruction the kernels use through a shim (wave votes and broadcasts, byte permutes,
 * packed dot products, the hand-scheduled walk of LSD's region growing, ...) is run next to its portable description over all
 * 64 lanes and compared (pl-slam_amd/csrc/plh_shims.h, selftest.hip).  *failing_checks = 0 on a device / toolchain where the
 * instructions do what the descriptions say; per_shim[plh_selftest_shims()] (optional) = mismatching lanes per shim. */
PLH_API plh_status plh_selftest(int device, int* failing_checks, int32_t* per_shim, int per_shim_cap);
PLH_API int plh_selftest_shims(void);
/* A fixed VALU-only launch (4096 x 256 threads x iters x 64 multiply-adds), timed twice with HIP events: ms[0] from whatever
 * power state the GPU was in, ms[1] at its running clocks.  bench.py reports it in front of the timed region so that rates
 * measured on different boxes can be normalised (no counterpart in the reference). */
PLH_API plh_status plh_box_probe(int device, int iters, float ms[2]);

/* ---------------------------------------------------------------------------------------------
 * ORB extractor  (replaces ORB_SLAM2::ORBextractor, include/ORBextractor.h:45-111)
 * ------------------------------------------------------------------------------------------- */
typedef struct plh_orb plh_orb;

typedef struct plh_orb_params {    /* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) */
  int32_t nfeatures;
  float scale_factor;
  int32_t nlevels;
  int32_t ini_th_fast;
  int32_t min_th_fast;
} plh_orb_params;

/* Plan a handle for images of exactly rows x cols and up to max_batch frames per call. */
PLH_API plh_status plh_orb_create(const plh_orb_params* p, int device, int rows, int cols, int max_batch, plh_orb** out);
PLH_API plh_status plh_orb_destroy(plh_orb* h);

/* Scale tables, as the reference getters return them (GetScaleFactors() etc., ORBextractor.h:62-84).
 * which: 0 = mvScaleFactor, 1 = mvInvScaleFactor, 2 = mvLevelSigma2, 3 = mvInvLevelSigma2. out[nlevels]. */
PLH_API plh_status plh_orb_scale_table(const plh_orb* h, int which, float* out);
PLH_API int plh_orb_levels(const plh_orb* h);
/* Features-per-level split (mnFeaturesPerLevel, ORBextractor.cc:435-446). out[nlevels]. */
PLH_API plh_status plh_orb_features_per_level(const plh_orb* h, int32_t* out);
/* Per-frame record capacity: sum over levels of max(N_l + 3, 4*nIni) (the quad-tree can overshoot N_l by <= 2). */
PLH_API int plh_orb_capacity(const plh_orb* h);

/* One frame, host buffers: the ORBextractor::operator() replacement (ORBextractor.cc:1043-1105).
 * img: rows x cols u8, `step` bytes per row. kps[cap], desc[cap*32]; *n_out = number of keypoints.
 * rows==0 || cols==0 is the reference's "empty image -> silent return": *n_out = 0, PLH_OK. */
PLH_API plh_status plh_orb_extract(plh_orb* h, const uint8_t* img, int rows, int cols, size_t step,
                                   plh_keypoint* kps, uint8_t* desc, int cap, int* n_out);

/* Batch, host buffers (frames contiguous, frame_stride bytes apart, rows tightly packed at `cols`). */
PLH_API plh_status plh_orb_extract_batch(plh_orb* h, const uint8_t* imgs, int batch, size_t frame_stride,
                                         plh_keypoint* kps, uint8_t* desc, int32_t* n_out);

/* Batch, device buffers, asynchronous on `stream`.
 *   d_imgs : batch planes of rows*cols u8 (row pitch = cols), frame_stride bytes apart
 *   d_kps  : batch * capacity plh_keypoint records
 *   d_desc : batch * capacity * 32 bytes
 *   d_n    : batch int32 counts
 * Records of frame b start at b*capacity; order inside a frame is the reference's (level-major,
 * quad-tree list order). */
PLH_API plh_status plh_orb_extract_batch_dev(plh_orb* h, const uint8_t* d_imgs, int batch, size_t frame_stride,
                                             plh_keypoint* d_kps, uint8_t* d_desc, int32_t* d_n, void* stream);

/* Capacity flags of the most recent extract call on this handle (any entry point; the call clears them in stream order
 * before its kernels run, so they never leak from one call into the next -- hence ONE call in flight per handle: a second
 * call enqueued on another stream before the first has run would clear or overwrite the first one's flags).  Waits for an
 * event the handle recorded behind that call's last kernel (the caller's stream may have been destroyed since), then reads
 * the device word.  bit 0: a FAST cell produced more candidates than its slot array; bit 1: a level's quad tree selected more
 * keypoints than the record capacity.  Both are impossible by construction of the plan; a non-zero value
 * means truncated lists.  The host-buffer entry points check it themselves and return PLH_ERR_CAPACITY. */
PLH_API plh_status plh_orb_status(plh_orb* h, int* flags);

/* Per-kernel device time, measured with HIP events recorded on the caller's stream around each launch
 * group.  kernel: 0 = pyramid (all levels), 1 = FAST cells, 2 = quad-tree, 3 = orientation+rBRIEF.
 * Call plh_orb_kernel_ms only after synchronising the stream; it folds and clears the pending events. */
PLH_API plh_status plh_orb_set_profiling(plh_orb* h, int on);
PLH_API plh_status plh_orb_kernel_ms(plh_orb* h, int kernel, double* total_ms, int* intervals);

/* After an extract call: geometry and DEVICE pointer of pyramid level `level` of frame `b`
 * (mvImagePyramid[level] of the reference; border-less). */
PLH_API plh_status plh_orb_pyramid_dev(const plh_orb* h, int b, int level, const uint8_t** d_ptr,
                                       int* rows, int* cols, size_t* pitch);
/* Debug/parity taps (device -> host copies, synchronous):
 *   pyramid level pixels (tightly packed rows*cols), and the FAST candidates of a level in
 *   the order they enter the quad-tree (x, y in level-image coordinates, response). */
PLH_API plh_status plh_orb_read_level(plh_orb* h, int b, int level, uint8_t* out, size_t out_bytes);
PLH_API plh_status plh_orb_read_candidates(plh_orb* h, int b, int level, plh_keypoint* out, int cap, int* n_out);

/* ---------------------------------------------------------------------------------------------
 * Hamming matching  (ORBmatcher / LSDmatcher inner loops)
 * ------------------------------------------------------------------------------------------- */

/* 256-bit Hamming distance of two 32-byte rows: ORBmatcher::DescriptorDistance (ORBmatcher.cc:1764-1780),
 * LSDmatcher::DescriptorDistance (LSDmatcher.cpp:654-670).  Host helper (scalar). */
PLH_API int plh_descriptor_distance(const uint8_t* a, const uint8_t* b);

/* Brute-force 2-NN, cv::BFMatcher(NORM_HAMMING,false).knnMatch(q,t,k=2) semantics (LSDmatcher.cpp:468-469):
 * for each query row the two smallest distances, ties -> lower train index first.
 * idx/dist are nq x 2 int32 (idx = -1, dist = 256+ when nt < 2 leaves a slot empty). */
PLH_API plh_status plh_hamming_knn2_dev(const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                                        int32_t* d_idx, int32_t* d_dist, void* stream);
PLH_API plh_status plh_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt,
                                    int32_t* idx, int32_t* dist, int device);

/* Batched form: P independent (query set, train set) pairs at fixed strides (sets padded to
 * q_cap / t_cap rows; nq[p], nt[p] give the live row counts). */
PLH_API plh_status plh_hamming_knn2_batch_dev(const uint8_t* d_q, const int32_t* d_nq, int q_cap,
                                              const uint8_t* d_t, const int32_t* d_nt, int t_cap,
                                              int pairs, int32_t* d_idx, int32_t* d_dist, void* stream);

/* LSDmatcher::FrameBFMatch + lineDescriptorMAD (LSDmatcher.cpp:462-486, 627-652) on top of a knn2 table:
 * matches12[nq] = train index or -1.  th = TH_LOW (50), nnratio = mfNNratio. Batched over pairs. */
PLH_API plh_status plh_line_bfmatch_batch_dev(const int32_t* d_idx, const int32_t* d_dist, const int32_t* d_nq,
                                              const int32_t* d_nt, int q_cap, int pairs, float th, float nnratio,
                                              int32_t* d_matches, void* stream);
/* LSDmatcher::SearchDouble(Frame&,Frame&,vector<int>&) (LSDmatcher.cpp:427-460): both directions + mutual check.
 * d_desc1/d_desc2: pairs x cap x 32; matches12: pairs x cap (train index in set 2 or -1); nmatches[pairs]. */
PLH_API plh_status plh_line_search_double_batch_dev(const uint8_t* d_desc1, const int32_t* d_n1,
                                                    const uint8_t* d_desc2, const int32_t* d_n2, int cap, int pairs,
                                                    float th, float nnratio, int32_t* d_matches12,
                                                    int32_t* d_nmatches, void* d_workspace, size_t workspace_bytes,
                                                    void* stream);
PLH_API size_t plh_line_search_double_workspace(int cap, int pairs);

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:187-327) on flat arrays.
 * Per pair p: set 1 (the KeyFrame) and set 2 (the Frame), each `cap` rows of 32-byte descriptors with
 * angle[] (degrees) and node[] (DBoW2 FeatureVector node id of the row, ascending groups are NOT required:
 * rows are grouped by the kernel exactly like std::map<node, vector<idx>> iteration: node ascending, row
 * index ascending inside a node).  valid1[] != 0 marks KeyFrame rows that carry a live MapPoint.
 * matches21[cap]: for Frame row j the matched KeyFrame row or -1 (the reference stores the MapPoint*).
 * th_low = TH_LOW (50), nnratio = mfNNratio, check_ori = mbCheckOrientation. */
PLH_API plh_status plh_orb_search_by_bow_batch_dev(const uint8_t* d_desc1, const float* d_angle1, const int32_t* d_node1,
                                                   const uint8_t* d_valid1, const int32_t* d_n1,
                                                   const uint8_t* d_desc2, const float* d_angle2, const int32_t* d_node2,
                                                   const int32_t* d_n2, int cap, int pairs, int th_low, float nnratio,
                                                   int check_ori, int32_t* d_matches21, int32_t* d_nmatches,
                                                   void* stream);

/* Host-buffer forms of the two matchers (one call = one reference call; they stage over PCIe and block):
 *   plh_line_search_double : LSDmatcher::SearchDouble(Frame&, Frame&, vector<int>&) on two mLdesc matrices
 *                            (LSDmatcher.cpp:427-460); matches12[n1] = row of ldesc2 or -1; th = TH_LOW (50).
 *   plh_orb_search_by_bow  : ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:187-327) on flat arrays;
 *                            matches21[n2] = KeyFrame feature whose MapPoint goes to Frame feature j, or -1. */
PLH_API plh_status plh_line_search_double(const uint8_t* ldesc1, int n1, const uint8_t* ldesc2, int n2, float th,
                                          float nnratio, int32_t* matches12, int* nmatches, int device);
PLH_API plh_status plh_orb_search_by_bow(const uint8_t* desc1, const float* angle1, const int32_t* node1,
                                         const uint8_t* valid1, int n1, const uint8_t* desc2, const float* angle2,
                                         const int32_t* node2, int n2, int th_low, float nnratio, int check_ori,
                                         int32_t* matches21, int* nmatches, int device);

/* Same search, angles taken from plh_keypoint records (kp.angle of mvKeysUn / mvKeys, ORBmatcher.cc:268-276). */
PLH_API plh_status plh_orb_search_by_bow_kp_batch_dev(const uint8_t* d_desc1, const plh_keypoint* d_kps1,
                                                      const int32_t* d_node1, const uint8_t* d_valid1, const int32_t* d_n1,
                                                      const uint8_t* d_desc2, const plh_keypoint* d_kps2,
                                                      const int32_t* d_node2, const int32_t* d_n2, int cap, int pairs,
                                                      int th_low, float nnratio, int check_ori, int32_t* d_matches21,
                                                      int32_t* d_nmatches, void* stream);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (ORBmatcher.cc:574-709; SURVEY 8f
 * row 2).  valid1 / valid2: the feature carries a non-bad MapPoint.  d_matches12[pairs][cap]: for feature idx1 of the first
 * KeyFrame the matched feature of the second one (the reference stores vpMapPoints2[idx2]) or -1. */
PLH_API plh_status plh_orb_search_by_bow_kfkf_batch_dev(const uint8_t* d_desc1, const plh_keypoint* d_kps1,
                                                        const int32_t* d_node1, const uint8_t* d_valid1, const int32_t* d_n1,
                                                        const uint8_t* d_desc2, const plh_keypoint* d_kps2,
                                                        const int32_t* d_node2, const uint8_t* d_valid2, const int32_t* d_n2,
                                                        int cap, int pairs, int th_low, float nnratio, int check_ori,
                                                        int32_t* d_matches12, int32_t* d_nmatches, void* stream);

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo = false) (ORBmatcher.cc:720-912 +
 * CheckDistEpipolarLine :154-173), monocular.  has_mp1 / has_mp2: the feature already carries a MapPoint
 * (skipped).  F12: row-major 3x3 float (HOST); (ex, ey): epipole of KF1's centre in KF2 (:731-735, from the caller's poses);
 * scale_factors2 / level_sigma2_2: HOST arrays pKF2->mvScaleFactors / mvLevelSigma2.  d_matches12[pairs][cap] = feature of
 * KF2 paired with feature idx1 of KF1, or -1 (vMatchedPairs = the (i, matches12[i]) pairs in index order). */
PLH_API plh_status plh_orb_search_for_triangulation_batch_dev(
    const plh_keypoint* d_kps1, const uint8_t* d_desc1, const int32_t* d_node1, const uint8_t* d_has_mp1, const int32_t* d_n1,
    const plh_keypoint* d_kps2, const uint8_t* d_desc2, const int32_t* d_node2, const uint8_t* d_has_mp2, const int32_t* d_n2,
    int cap, int pairs, const float F12[9], float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
    int nlevels, int th_low, int check_ori, int32_t* d_matches12, int32_t* d_nmatches, void* stream);

/* DBoW2 TemplatedVocabulary::transform(feature, word, weight, &nid, levelsup) for every descriptor of a batch
 * (Frame::ComputeBoW, Frame.cc:906-913 -> Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1255).
 * The tree is given as flat arrays over node ids (root = 0): 32-byte node descriptors, contiguous children
 * [child_start, child_start + child_count), word id and weight of the leaves.  L = tree depth, levelsup as in
 * ComputeBoW (4).  Outputs per descriptor: FeatureVector node id (or -1 for a stopped word / unused row) and word id. */
PLH_API plh_status plh_bow_transform_batch_dev(const uint8_t* d_desc, const int32_t* d_n, int cap, int batch,
                                               const uint8_t* d_node_desc, const int32_t* d_child_start,
                                               const int32_t* d_child_count, const int32_t* d_word_id,
                                               const float* d_weight, int L, int levelsup, int32_t* d_nid,
                                               int32_t* d_word, void* stream);

/* The BowVector half of TemplatedVocabulary::transform(features, v, fv, levelsup) (TemplatedVocabulary.h:1139-1205) on the
 * word ids plh_bow_transform_batch_dev produced: per frame the distinct words of the live features in ascending order
 * (std::map order) with their weights accumulated and normalised exactly as BowVector::addWeight / addIfNotExist /
 * normalize do (BowVector.cpp:34-84; WordValue = double, sums in map order -> bit-identical values).
 * d_word_weight[word id] = weight of the word's node (double); weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; scoring:
 * 0 L1_NORM, 1 L2_NORM, 2 CHI_SQUARE, 3 KL, 4 BHATTACHARYYA, 5 DOT_PRODUCT (BowVector.h:36-53; ORBvoc is TF_IDF / L1_NORM).
 * Outputs: d_bow_word / d_bow_value [batch][cap] (the first d_bow_n[b] entries of frame b), cap <= 8192. */
PLH_API plh_status plh_bow_vector_batch_dev(const int32_t* d_word, const int32_t* d_n, int cap, int batch,
                                            const double* d_word_weight, int weighting, int scoring, int32_t* d_bow_word,
                                            double* d_bow_value, int32_t* d_bow_n, void* stream);

/* DBoW2 vocabulary handle (ORBVocabulary = TemplatedVocabulary<FORB::TDescriptor, FORB>, include/ORBVocabulary.h:30-31), so that
 * a C++ host gets ORBvoc.txt / ORBvoc.bin onto the device without any Python (System.cc:66-84 loads one of the two by suffix).
 *   plh_vocab_load_text    TemplatedVocabulary::loadFromTextFile   (TemplatedVocabulary.h:1350-1438)
 *   plh_vocab_load_binary  TemplatedVocabulary::loadFromBinaryFile (:1465-1506; file layout of saveToBinaryFile :1511-1536)
 *   plh_vocab_save_binary  TemplatedVocabulary::saveToBinaryFile
 *   plh_vocab_create       from arrays in the reference's node numbering: node 0 = root, arrays indexed by node id (entry 0
 *                          ignored), parent[i] < i, weight in double.
 * Node and word ids are the reference's (word ids count the leaves in node order). */
typedef struct plh_vocab plh_vocab;
typedef struct plh_vocab_info {
  int32_t k, L, scoring, weighting;   /* m_k, m_L, m_scoring, m_weighting */
  int32_t n_nodes, n_words;           /* m_nodes.size(), m_words.size() */
  int32_t identity_ids;               /* 1: flat index == NodeId (children were contiguous in the file); 0: renumbered */
} plh_vocab_info;
PLH_API plh_status plh_vocab_load_text(const char* path, int device, plh_vocab** out);
PLH_API plh_status plh_vocab_load_binary(const char* path, int device, plh_vocab** out);
PLH_API plh_status plh_vocab_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                                    const uint8_t* is_leaf, const uint8_t* node_desc, const double* weight, int device,
                                    plh_vocab** out);
PLH_API plh_status plh_vocab_save_binary(const plh_vocab* v, const char* path);
PLH_API plh_status plh_vocab_destroy(plh_vocab* v);
PLH_API plh_status plh_vocab_get_info(const plh_vocab* v, plh_vocab_info* info);
/* The handle's device arrays in the form plh_bow_transform_batch_dev takes (any pointer may be NULL); *d_node_id is NULL when
 * the flat index is the NodeId, else the map flat index -> NodeId. */
PLH_API plh_status plh_vocab_device_arrays(const plh_vocab* v, const uint8_t** d_node_desc, const int32_t** d_child_start,
                                           const int32_t** d_child_count, const int32_t** d_word_id, const float** d_weight,
                                           const int32_t** d_node_id);
/* Host copies of the flat tree, n_nodes entries each (any pointer may be NULL). */
PLH_API plh_status plh_vocab_read(const plh_vocab* v, uint8_t* node_desc, int32_t* child_start, int32_t* child_count,
                                  int32_t* word_id, double* weight, int32_t* node_id);
/* Frame::ComputeBoW for a batch (Frame.cc:906-913: transform(vCurrentDesc, mBowVec, mFeatVec, 4)): d_nid / d_word per
 * descriptor as plh_bow_transform_batch_dev (FeatureVector = rows grouped by d_nid), BowVector as plh_bow_vector_batch_dev
 * (pass NULL for the three bow outputs to skip it). */
PLH_API plh_status plh_vocab_transform_batch_dev(const plh_vocab* v, const uint8_t* d_desc, const int32_t* d_n, int cap, int batch,
                                                 int levelsup, int32_t* d_nid, int32_t* d_word, int32_t* d_bow_word,
                                                 double* d_bow_value, int32_t* d_bow_n, void* stream);
/* The same for ONE frame on host buffers: the body of the drop-in ORBVocabulary::transform(features, v, fv, levelsup)
 * (pl-slam_amd/adaptor/ORBVocabulary.h) behind Frame::ComputeBoW (Frame.cc:906-913) and KeyFrame::ComputeBoW (KeyFrame.cc:76-83).
 * desc: n x 32 bytes (mDescriptors, n <= 8192).  nid[i] / word[i]: FeatureVector node (-1 = stopped word) and word id of feature i;
 * bow_word / bow_value (room for n entries): the *bow_n distinct words in ascending order with their normalised weights (the
 * BowVector's std::map order and doubles).  Stages through the calling thread's own arena and stream: the tracking and the
 * local-mapping thread may call it concurrently on one handle. */
PLH_API plh_status plh_vocab_transform(const plh_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* nid, int32_t* word,
                                       int32_t* bow_word, double* bow_value, int* bow_n);

/* ---------------------------------------------------------------------------------------------
 * Windowed (grid) searches  (Frame::AssignFeaturesToGrid*, GetFeaturesInArea*, ORBmatcher::SearchForInitialization /
 * SearchByProjection, LSDmatcher::SearchByProjection)
 *
 * The boundary sits after the projection: a query carries what the reference reads from the MapPoint / MapLine
 * (mTrackProjX/Y, mnTrackScaleLevel, mTrackViewCos, descriptor, "Observations() > 0") or computes from the pose
 * (u, v); the cv::Mat pose algebra stays with the caller (pl-slam_amd/adaptor/HipMatchers.h).
 * Batched over `pairs` independent frames at fixed strides: `cap` rows per frame, `qcap` queries per frame.
 * Limits: cap <= 6000 (per-frame state lives in LDS); qcap is unlimited -- queries are streamed -- except for the two
 * forms with a rotation histogram (`check_ori` != 0 in *_projection_frame / *_projection_kf: qcap <= 12000).
 * A query whose predicted level lies outside [0, nlevels) is skipped (the reference would index mvScaleFactors out of
 * bounds there; MapPoint::PredictScale clamps, MapLine::PredictScale does not).
 * ------------------------------------------------------------------------------------------- */
#define PLH_GRID_COLS 64   /* FRAME_GRID_COLS, include/Frame.h:45 */
#define PLH_GRID_ROWS 48   /* FRAME_GRID_ROWS, include/Frame.h:44 */
#define PLH_GRID_CELLS (PLH_GRID_COLS * PLH_GRID_ROWS)

typedef struct plh_grid_params {   /* Frame::mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv */
  float min_x, min_y, max_x, max_y, inv_w, inv_h;
} plh_grid_params;

/* Frame::AssignFeaturesToGrid (Frame.cc:278-293; PosInGrid :893-905 rounds, it does not floor).  CSR per frame:
 * d_cell_start[batch][64*48+1], d_cell_items[batch][cap]; cell (ix, iy) -> ix*48 + iy; items in insertion order. */
PLH_API plh_status plh_frame_assign_grid_batch_dev(const plh_keypoint* d_kps_un, const int32_t* d_n, int cap, int batch,
                                                   const plh_grid_params* gp, int32_t* d_cell_start, int32_t* d_cell_items,
                                                   void* stream);
/* Frame::AssignFeaturesToGridForLine (Frame.cc:295-320) with the reference's LineIterator (src/lineIterator.cpp:34-77):
 * a line is listed in every cell it crosses.  d_cell_items[batch][item_cap], item_cap >= cap * 64. */
PLH_API plh_status plh_frame_assign_grid_lines_batch_dev(const plh_keyline* d_kl, const int32_t* d_nl, int cap, int batch,
                                                         const plh_grid_params* gp, int32_t* d_cell_start,
                                                         int32_t* d_cell_items, int item_cap, void* stream);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:455-572).
 * kps1 / kps2 = mvKeysUn of the two frames, grid = F2's; d_prev_matched[pairs][cap][2] is updated in place;
 * d_matches12[pairs][cap] = index in F2 or -1. */
PLH_API plh_status plh_orb_search_for_initialization_batch_dev(
    const plh_keypoint* d_kps1, const uint8_t* d_desc1, const int32_t* d_n1, const plh_keypoint* d_kps2, const uint8_t* d_desc2,
    const int32_t* d_n2, int cap, int pairs, const plh_grid_params* gp2, const int32_t* d_cell_start2,
    const int32_t* d_cell_items2, float* d_prev_matched, int window_size, float nnratio, int check_ori, int32_t* d_matches12,
    int32_t* d_nmatches, void* stream);

/* ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>&, th) (ORBmatcher.cc:56-144), monocular.
 * Query: valid = mbTrackInView && !isBad(); xy = (mTrackProjX, mTrackProjY); level = mnTrackScaleLevel;
 * viewcos = mTrackViewCos; hasobs = Observations() > 0.  d_occupied[pairs][cap] (in/out): F.mvpMapPoints[idx] set and
 * Observations() > 0.  d_assigned[pairs][cap]: query whose MapPoint now sits at F feature idx, or -1.
 * scale_factors: HOST array F.mvScaleFactors[nlevels <= 16]. */
PLH_API plh_status plh_orb_search_by_projection_mp_batch_dev(
    const plh_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_n, int cap, int pairs, const plh_grid_params* gp,
    const int32_t* d_cell_start, const int32_t* d_cell_items, const float* scale_factors, int nlevels, uint8_t* d_occupied,
    const int32_t* d_nq, int qcap, const uint8_t* d_q_valid, const float* d_q_xy, const int32_t* d_q_level,
    const float* d_q_viewcos, const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th, float nnratio, int32_t* d_assigned,
    int32_t* d_nmatches, void* stream);

/* ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) (ORBmatcher.cc:1441-1585).
 * Query i = Last feature i: valid = MapPoint && !mvbOutlier[i] && invzc >= 0; uv = projection into Cur (the image
 * bounds test is done here); octave = Last.mvKeys[i].octave; angle = Last.mvKeysUn[i].angle.
 * mode 0 = monocular / lateral (octave-1..octave+1), 1 = forward (>= octave), 2 = backward (<= octave). */
PLH_API plh_status plh_orb_search_by_projection_frame_batch_dev(
    const plh_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_n, int cap, int pairs, const plh_grid_params* gp,
    const int32_t* d_cell_start, const int32_t* d_cell_items, const float* scale_factors, int nlevels, uint8_t* d_occupied,
    const int32_t* d_nq, int qcap, const uint8_t* d_q_valid, const float* d_q_uv, const int32_t* d_q_octave,
    const float* d_q_angle, const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th, int mode, int check_ori,
    int32_t* d_assigned, int32_t* d_nmatches, void* stream);

/* ORBmatcher::SearchByProjection(Frame& Cur, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
 * (ORBmatcher.cc:1587-1716, relocalisation): the last-frame form with the caller's distance threshold.
 * Query i = KeyFrame map point i: valid = pMP && !isBad() && !sAlreadyFound.count(pMP) && depth inside the scale pyramid;
 * level = pMP->PredictScale(dist3D, &Cur); angle = pKF->mvKeysUn[i].angle; occupied = Cur.mvpMapPoints[i2] != NULL; hasobs = 1. */
PLH_API plh_status plh_orb_search_by_projection_kf_batch_dev(
    const plh_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_n, int cap, int pairs, const plh_grid_params* gp,
    const int32_t* d_cell_start, const int32_t* d_cell_items, const float* scale_factors, int nlevels, uint8_t* d_occupied,
    const int32_t* d_nq, int qcap, const uint8_t* d_q_valid, const float* d_q_uv, const int32_t* d_q_level,
    const float* d_q_angle, const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th, int orb_dist, int check_ori,
    int32_t* d_assigned, int32_t* d_nmatches, void* stream);

/* The search inside ORBmatcher::Fuse(pKF, vpMapPoints, th) / Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (ORBmatcher.cc:914-1061,
 * 1063-1197): per query (valid = every pre-check of the loop passed; uv = projection; level = PredictScale)
 * the best keypoint of level l-1..l in the window th*scale[l] whose reprojection error passes e2*invLevelSigma2 <= 5.99 and
 * whose Hamming distance is <= th_low (TH_LOW).  d_best_idx[pairs][qcap] = keypoint index or -1.  The replace / add logic on
 * the map stays with the caller.  scale_factors / inv_level_sigma2: HOST arrays of the KeyFrame.  The Sim3 overload (:1063-1197)
 * has no reprojection-error gate: pass inv_level_sigma2 = 0 for it. */
PLH_API plh_status plh_orb_fuse_search_batch_dev(const plh_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_n, int cap,
                                                 int pairs, const plh_grid_params* gp, const int32_t* d_cell_start,
                                                 const int32_t* d_cell_items, const float* scale_factors,
                                                 const float* inv_level_sigma2, int nlevels, const int32_t* d_nq, int qcap,
                                                 const uint8_t* d_q_valid, const float* d_q_uv, const int32_t* d_q_level,
                                                 const uint8_t* d_q_desc, float th, int th_low, int32_t* d_best_idx,
                                                 int32_t* d_nfound, void* stream);
/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:329-453, loop closing).
 * occupied = vpMatched[idx] != NULL (in/out); d_assigned[pairs][cap] = query whose MapPoint goes to vpMatched[idx], or -1. */
PLH_API plh_status plh_orb_search_by_projection_sim3_batch_dev(
    const plh_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_n, int cap, int pairs, const plh_grid_params* gp,
    const int32_t* d_cell_start, const int32_t* d_cell_items, const float* scale_factors, int nlevels, uint8_t* d_occupied,
    const int32_t* d_nq, int qcap, const uint8_t* d_q_valid, const float* d_q_uv, const int32_t* d_q_level,
    const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th, int th_low, int32_t* d_assigned, int32_t* d_nmatches,
    void* stream);

/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (ORBmatcher.cc:1199-1439, loop closing).
 * Both KeyFrames share `cap`, the grid geometry and the scale table; query i of a direction belongs to keypoint slot i.
 * q12_* = KeyFrame 1's map points transformed into KeyFrame 2 (valid = pMP && !vbAlreadyMatched1[i] && !isBad() && depth >= 0 &&
 * IsInImage && distance inside the invariance region; uv; level = PredictScale; desc = pMP->GetDescriptor()), q21_* the reverse.
 * Per direction the best keypoint of level l-1..l in the window th*scale[l] with Hamming <= th_high (TH_HIGH); outputs
 * d_match1 / d_match2 [pairs][cap] = vnMatch1 / vnMatch2, d_match12[pairs][cap] = agreed index in KeyFrame 2 or -1
 * (vpMatches12[i1] = vpMapPoints2[that index]), d_nfound[pairs] = nFound. */
PLH_API plh_status plh_orb_search_by_sim3_batch_dev(
    const plh_keypoint* d_kps1_un, const uint8_t* d_desc1, const int32_t* d_n1, const int32_t* d_cell_start1,
    const int32_t* d_cell_items1, const plh_keypoint* d_kps2_un, const uint8_t* d_desc2, const int32_t* d_n2,
    const int32_t* d_cell_start2, const int32_t* d_cell_items2, int cap, int pairs, const plh_grid_params* gp,
    const float* scale_factors, int nlevels, const uint8_t* d_q12_valid, const float* d_q12_uv, const int32_t* d_q12_level,
    const uint8_t* d_q12_desc, const uint8_t* d_q21_valid, const float* d_q21_uv, const int32_t* d_q21_level,
    const uint8_t* d_q21_desc, float th, int th_high, int32_t* d_match1, int32_t* d_match2, int32_t* d_match12,
    int32_t* d_nfound, void* stream);

/* LSDmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th) (LSDmatcher.cpp:72-176).
 * Query i = Last line i: valid = MapLine && !mvbLineOutlier[i] && Cur.isInFrustum(pML, 0.5);
 * seg = (mTrackProjX1, Y1, X2, Y2); length = Last.mvKeylinesUn[i].lineLength.  d_linefn = mvKeyLineFunctions. */
PLH_API plh_status plh_line_search_by_projection_frame_batch_dev(
    const plh_keyline* d_kl, const uint8_t* d_ldesc, const double* d_linefn, const int32_t* d_nl, int cap, int pairs,
    const plh_grid_params* gp, const int32_t* d_cell_start, const int32_t* d_cell_items, int item_cap, uint8_t* d_occupied,
    const int32_t* d_nq, int qcap, const uint8_t* d_q_valid, const float* d_q_seg, const float* d_q_length,
    const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th, int32_t* d_assigned, int32_t* d_nmatches, void* stream);

/* The search inside LSDmatcher::Fuse(pKF, vpMapLines, th) (LSDmatcher.cpp:860-1002) with
 * KeyFrame::GetLinesInArea (KeyFrame.cc:647-683, cos_th = 0.998): per query the best line of level l-1..l near the projected
 * segment, Hamming <= th_low.  d_cand_desc is the matrix the reference reads the candidates from (it reads pKF->mDescriptors
 * with the line index, :963 -- pass mLineDescriptors for the intended behaviour).  d_best_idx[pairs][qcap] = line or -1.
 * Reference quirk for the caller: a line with an endpoint behind the camera makes Fuse `return false` (:893-894), i.e. no
 * later query is searched -- set valid = 0 from that query on to reproduce it. */
PLH_API plh_status plh_line_fuse_search_batch_dev(const plh_keyline* d_kl, const uint8_t* d_cand_desc, const int32_t* d_nl, int cap,
                                                  int pairs, const float* scale_factors_line, int nlevels, const int32_t* d_nq,
                                                  int qcap, const uint8_t* d_q_valid, const float* d_q_seg,
                                                  const int32_t* d_q_level, const uint8_t* d_q_desc, float th, float cos_th,
                                                  int th_low, int32_t* d_best_idx, int32_t* d_nfound, void* stream);

/* LSDmatcher::SearchByProjection(Frame& F, const vector<MapLine*>&, th) (LSDmatcher.cpp:221-338). */
PLH_API plh_status plh_line_search_by_projection_ml_batch_dev(
    const plh_keyline* d_kl, const uint8_t* d_ldesc, const double* d_linefn, const int32_t* d_nl, int cap, int pairs,
    const plh_grid_params* gp, const int32_t* d_cell_start, const int32_t* d_cell_items, int item_cap, uint8_t* d_occupied,
    const int32_t* d_nq, int qcap, const uint8_t* d_q_valid, const float* d_q_seg, const float* d_q_viewcos,
    const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th, float nnratio, int32_t* d_assigned, int32_t* d_nmatches,
    void* stream);

/* The projection searches above run as a prepass (one lane per query: candidates, Hamming distances, the few best per query -- every
 * query of every frame in parallel) and an ordered resolve (one wavefront per frame replays the reference's order on those short
 * lists, 64 queries at a time).  on = 1 selects the one-wavefront-per-frame kernels of rounds 1-5 instead, on = 2 the prepass + resolve
 * with the candidate lists cut to two entries (contended queries then run out of list and take the slow path); identical results; an
 * A/B and test switch, process-wide, not for concurrent use. */
PLH_API plh_status plh_debug_set_proj_serial(int on);

/* Host-buffer forms: one call = one reference call on one frame (they stage over PCIe, rebuild the frame's grid on the
 * device and block).  Arrays as in the *_batch_dev forms, without padding. */
PLH_API plh_status plh_orb_search_for_initialization(const plh_keypoint* kps1, const uint8_t* desc1, int n1,
                                                     const plh_keypoint* kps2, const uint8_t* desc2, int n2,
                                                     const plh_grid_params* gp2, float* prev_matched, int window_size,
                                                     float nnratio, int check_ori, int32_t* matches12, int* nmatches, int device);
PLH_API plh_status plh_orb_search_by_projection_mp(const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp,
                                                   const float* scale_factors, int nlevels, uint8_t* occupied, int nq,
                                                   const uint8_t* q_valid, const float* q_xy, const int32_t* q_level,
                                                   const float* q_viewcos, const uint8_t* q_desc, const uint8_t* q_hasobs, float th,
                                                   float nnratio, int32_t* assigned, int* nmatches, int device);
PLH_API plh_status plh_orb_search_by_projection_frame(const plh_keypoint* kps_un, const uint8_t* desc, int n,
                                                      const plh_grid_params* gp, const float* scale_factors, int nlevels,
                                                      uint8_t* occupied, int nq, const uint8_t* q_valid, const float* q_uv,
                                                      const int32_t* q_octave, const float* q_angle, const uint8_t* q_desc,
                                                      const uint8_t* q_hasobs, float th, int mode, int check_ori,
                                                      int32_t* assigned, int* nmatches, int device);
PLH_API plh_status plh_line_search_by_projection_frame(const plh_keyline* kl, const uint8_t* ldesc, const double* linefn, int nl,
                                                       const plh_grid_params* gp, uint8_t* occupied, int nq,
                                                       const uint8_t* q_valid, const float* q_seg, const float* q_length,
                                                       const uint8_t* q_desc, const uint8_t* q_hasobs, float th,
                                                       int32_t* assigned, int* nmatches, int device);
PLH_API plh_status plh_line_search_by_projection_ml(const plh_keyline* kl, const uint8_t* ldesc, const double* linefn, int nl,
                                                    const plh_grid_params* gp, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                                    const float* q_seg, const float* q_viewcos, const uint8_t* q_desc,
                                                    const uint8_t* q_hasobs, float th, float nnratio, int32_t* assigned,
                                                    int* nmatches, int device);

/* Back-end searches, host-buffer forms: one call = the search of one reference call (arrays as in the *_batch_dev forms above;
 * stage over PCIe, block until done).  What each replaces, and who calls it in the reference:
 *   plh_orb_search_by_bow_kfkf         ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)      src/ORBmatcher.cc:574-709    LoopClosing.cc:282
 *   plh_orb_search_for_triangulation   ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, pairs, false) :720-912 (+ :154-173)      LocalMapping.cc:385
 *   plh_orb_fuse_search                the search inside ORBmatcher::Fuse(pKF, vpMapPoints, th)          :914-1061                  LocalMapping.cc:1545, 1573
 *                                      and Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (NULL sigma2)   :1063-1197                 LoopClosing.cc:595
 *   plh_orb_search_by_projection_sim3  ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) :329-453                   LoopClosing.cc:360
 *   plh_orb_search_by_sim3             ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) :1199-1439              LoopClosing.cc:327
 *   plh_orb_search_by_projection_kf    ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) :1587-1716    Tracking::Relocalization
 *   plh_line_frame_bfmatch             LSDmatcher::FrameBFMatch(ldesc1, ldesc2, LineMatches, TH)   src/LSDmatcher.cpp:462-486       (SearchForTriangulation, isDouble = false)
 *   plh_line_fuse_search               the search inside LSDmatcher::Fuse(pKF, vpMapLines, th)           :860-1002                  LocalMapping.cc:1600, 1627
 *   plh_line_frame_bfmatch_new         LSDmatcher::FrameBFMatchNew(ldesc1, ldesc2, LineMatches, kls1, kls2, kls2func, F, TH)   :488-625
 *   plh_line_search_for_triangulation_new   LSDmatcher::SearchForTriangulationNew(pKF1, pKF2, vMatchedPairs, isDouble)         :780-832      (LocalMapping.cc:960, commented out)
 * (LSDmatcher::SearchForTriangulation's mutual forms, :672-778, are plh_line_search_double at TH_LOW / TH_HIGH.)
 * The adaptor overloads that keep the reference's signatures around them: pl-slam_amd/adaptor/HipORBmatcher.h, HipLSDmatcher.h. */
PLH_API plh_status plh_orb_search_by_bow_kfkf(const plh_keypoint* kps1, const uint8_t* desc1, const int32_t* node1,
                                              const uint8_t* valid1, int n1, const plh_keypoint* kps2, const uint8_t* desc2,
                                              const int32_t* node2, const uint8_t* valid2, int n2, int th_low, float nnratio,
                                              int check_ori, int32_t* matches12, int* nmatches, int device);
PLH_API plh_status plh_orb_search_for_triangulation(const plh_keypoint* kps1, const uint8_t* desc1, const int32_t* node1,
                                                    const uint8_t* has_mp1, int n1, const plh_keypoint* kps2, const uint8_t* desc2,
                                                    const int32_t* node2, const uint8_t* has_mp2, int n2, const float F12[9],
                                                    float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                                    int nlevels, int th_low, int check_ori, int32_t* matches12, int* nmatches,
                                                    int device);
PLH_API plh_status plh_line_frame_bfmatch(const uint8_t* ldesc1, int n1, const uint8_t* ldesc2, int n2, float th, float nnratio,
                                          int32_t* matches12, int device);
/* LSDmatcher::FrameBFMatchNew(ldesc1, ldesc2, LineMatches, kls1, kls2, kls2func, F, TH) (src/LSDmatcher.cpp:488-548, with
 * mutualOverlap :550-625): the nearest LBD neighbour of every line of set 1, accepted if the segment carried over the fundamental
 * matrix F (row-major 3 x 3; end point p of line 1 -> epipolar line F p, intersected with line 2's equation) overlaps line 2's own
 * segment by more than 0.8, the distance is below th and the ratio test holds.  seg = (startPointX, startPointY, endPointX, endPointY)
 * of every KeyLine, func2 = mvKeyLineFunctions of set 2 (3 doubles per line).  matches12[n1] = index in set 2 or -1. */
PLH_API plh_status plh_line_frame_bfmatch_new(const uint8_t* ldesc1, int n1, const uint8_t* ldesc2, int n2, const float* seg1,
                                              const float* seg2, const double* func2, const float F[9], float th, float nnratio,
                                              int32_t* matches12, int device);
/* LSDmatcher::SearchForTriangulationNew(pKF1, pKF2, vMatchedPairs, isDouble) (src/LSDmatcher.cpp:780-832; the reference's only call
 * site, LocalMapping.cc:960, is commented out): FrameBFMatchNew both ways at th = TH_LOW -- F21 = ComputeF12(pKF2, pKF1) for 1 -> 2,
 * F12 = ComputeF12(pKF1, pKF2) for 2 -> 1 (:834-858, the caller's) -- the mutual check if is_double, and only pairs of lines neither
 * of which has a MapLine (has_ml = pKF->GetMapLine(i) != NULL).  matches12[n1], *nmatches = the method's return value. */
PLH_API plh_status plh_line_search_for_triangulation_new(const uint8_t* ldesc1, int n1, const uint8_t* ldesc2, int n2, const float* seg1,
                                                         const float* seg2, const double* func1, const double* func2,
                                                         const float F21[9], const float F12[9], const uint8_t* has_ml1,
                                                         const uint8_t* has_ml2, float th, float nnratio, int is_double,
                                                         int32_t* matches12, int* nmatches, int device);
PLH_API plh_status plh_line_fuse_search(const plh_keyline* kl, const uint8_t* cand_desc, int nl, const float* scale_factors_line,
                                        int nlevels, int nq, const uint8_t* q_valid, const float* q_seg, const int32_t* q_level,
                                        const uint8_t* q_desc, float th, float cos_th, int th_low, int32_t* best_idx, int* nfound,
                                        int device);
PLH_API plh_status plh_orb_search_by_sim3(const plh_keypoint* kps1_un, const uint8_t* desc1, int n1, const plh_keypoint* kps2_un,
                                          const uint8_t* desc2, int n2, const plh_grid_params* gp, const float* scale_factors,
                                          int nlevels, const uint8_t* q12_valid, const float* q12_uv, const int32_t* q12_level,
                                          const uint8_t* q12_desc, const uint8_t* q21_valid, const float* q21_uv,
                                          const int32_t* q21_level, const uint8_t* q21_desc, float th, int th_high, int32_t* match12,
                                          int* nfound, int device);
PLH_API plh_status plh_orb_search_by_projection_kf(const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp,
                                                   const float* scale_factors, int nlevels, uint8_t* occupied, int nq,
                                                   const uint8_t* q_valid, const float* q_uv, const int32_t* q_level,
                                                   const float* q_angle, const uint8_t* q_desc, float th, int orb_dist,
                                                   int check_ori, int32_t* assigned, int* nmatches, int device);
PLH_API plh_status plh_orb_search_by_projection_sim3(const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp,
                                                     const float* scale_factors, int nlevels, uint8_t* occupied, int nq,
                                                     const uint8_t* q_valid, const float* q_uv, const int32_t* q_level,
                                                     const uint8_t* q_desc, const uint8_t* q_hasobs, float th, int th_low,
                                                     int32_t* assigned, int* nmatches, int device);
PLH_API plh_status plh_orb_fuse_search(const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp,
                                       const float* scale_factors, const float* inv_level_sigma2, int nlevels, int nq,
                                       const uint8_t* q_valid, const float* q_uv, const int32_t* q_level, const uint8_t* q_desc,
                                       float th, int th_low, int32_t* best_idx, int* nfound, int device);

/* ---------------------------------------------------------------------------------------------
 * Resident frames (round 6): what the searches read from one Frame / KeyFrame, uploaded ONCE.
 *
 * A tracked frame is searched two to four times (Tracking.cc:1321-1357 TrackWithMotionModel or :1151-1159 TrackReferenceKeyFrame,
 * :1792-1855 SearchLocalPoints / SearchLocalLines, and again as the last frame of the next one); the host-buffer forms above
 * upload mvKeysUn + mDescriptors (60 KB) and rebuild the grid of Frame::AssignFeaturesToGrid inside every call.  A resident handle
 * holds them on the device -- plh_frame_points: mvKeysUn, mDescriptors, the 64 x 48 grid (Frame.cc:278-293), optionally the
 * FeatureVector node of every feature (after ComputeBoW); plh_frame_lines: mvKeylinesUn, mLdesc, mvKeyLineFunctions, the line grid
 * (Frame.cc:295-320) -- and the *_resident forms stage the queries only (the calling thread's arena and stream).  Handles are
 * immutable after creation (set_nodes once, before the first SearchByBoW): any number of threads may search one concurrently.
 * Arguments and results are those of the host-buffer forms of the same name.  The adaptor classes keep an LRU of handles keyed by
 * Frame::mnId (pl-slam_amd/adaptor/HipMatchers.h: hip::FrameResidency).
 * ------------------------------------------------------------------------------------------- */
typedef struct plh_frame_points plh_frame_points;
typedef struct plh_frame_lines plh_frame_lines;
PLH_API plh_status plh_frame_points_create(const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp, int device,
                                           plh_frame_points** out);
PLH_API plh_status plh_frame_points_set_nodes(plh_frame_points* f, const int32_t* node);   /* node[i] = FeatureVector node of feature i, -1 = none */
PLH_API plh_status plh_frame_points_destroy(plh_frame_points* f);
PLH_API int plh_frame_points_count(const plh_frame_points* f);
PLH_API plh_status plh_frame_lines_create(const plh_keyline* kl, const uint8_t* ldesc, const double* linefn, int nl,
                                          const plh_grid_params* gp, int device, plh_frame_lines** out);
PLH_API plh_status plh_frame_lines_destroy(plh_frame_lines* f);
PLH_API int plh_frame_lines_count(const plh_frame_lines* f);
PLH_API plh_status plh_orb_search_by_projection_mp_resident(const plh_frame_points* f, const float* scale_factors, int nlevels,
                                                            uint8_t* occupied, int nq, const uint8_t* q_valid, const float* q_xy,
                                                            const int32_t* q_level, const float* q_viewcos, const uint8_t* q_desc,
                                                            const uint8_t* q_hasobs, float th, float nnratio, int32_t* assigned,
                                                            int* nmatches);
PLH_API plh_status plh_orb_search_by_projection_frame_resident(const plh_frame_points* f, const float* scale_factors, int nlevels,
                                                               uint8_t* occupied, int nq, const uint8_t* q_valid, const float* q_uv,
                                                               const int32_t* q_octave, const float* q_angle, const uint8_t* q_desc,
                                                               const uint8_t* q_hasobs, float th, int mode, int check_ori,
                                                               int32_t* assigned, int* nmatches);
PLH_API plh_status plh_orb_search_for_initialization_resident(const plh_frame_points* f1, const plh_frame_points* f2, float* prev_matched,
                                                              int window_size, float nnratio, int check_ori, int32_t* matches12,
                                                              int* nmatches);
PLH_API plh_status plh_orb_search_by_bow_resident(const plh_frame_points* kf, const uint8_t* valid1, const plh_frame_points* f, int th_low,
                                                  float nnratio, int check_ori, int32_t* matches21, int* nmatches);
PLH_API plh_status plh_line_search_by_projection_frame_resident(const plh_frame_lines* f, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                                                const float* q_seg, const float* q_length, const uint8_t* q_desc,
                                                                const uint8_t* q_hasobs, float th, int32_t* assigned, int* nmatches);
PLH_API plh_status plh_line_search_by_projection_ml_resident(const plh_frame_lines* f, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                                             const float* q_seg, const float* q_viewcos, const uint8_t* q_desc,
                                                             const uint8_t* q_hasobs, float th, float nnratio, int32_t* assigned,
                                                             int* nmatches);
PLH_API plh_status plh_line_search_double_resident(const plh_frame_lines* l1, const plh_frame_lines* l2, float th, float nnratio,
                                                   int32_t* matches12, int* nmatches);

/* ---------------------------------------------------------------------------------------------
 * Frame / map post-processing either side of the matching path
 * ------------------------------------------------------------------------------------------- */
/* Frame::UndistortKeyPoints (Frame.cc:915-945): cv::undistortPoints(pts, pts, K, D, noArray(), K) on every keypoint,
 * all other KeyPoint fields copied; D[0] == 0 -> plain copy.  K = fx,fy,cx,cy ; D = k1,k2,p1,p2,k3 (host arrays).
 * d_kps / d_kps_un: batch x cap records (may alias). */
PLH_API plh_status plh_undistort_keypoints_batch_dev(const plh_keypoint* d_kps, const int32_t* d_n, int cap, int batch,
                                                     const float K[4], const float D[5], plh_keypoint* d_kps_un, void* stream);
/* Frame::isInFrustum(MapPoint*, viewingCosLimit) / (MapLine*, viewingCosLimit) (Frame.cc:560-623, 625-711) with
 * MapPoint::PredictScale(dist, Frame*) (MapPoint.cc:413-428) / MapLine::PredictScale (MapLine.cpp:395-404): the per-element
 * visibility test Tracking::SearchLocalPoints / SearchLocalLines run in front of SearchByProjection(F, MapPoints / MapLines);
 * its outputs are that search's query arrays.  One plh_frame_view per frame (device array), d_nq[frames] elements per frame
 * at stride qcap.  Points: d_pos[..][3] = GetWorldPos, d_normal[..][3] = GetNormal, d_min_dist / d_max_dist = mfMinDistance /
 * mfMaxDistance (the 0.8 / 1.2 invariance factors are applied inside); outputs valid = mbTrackInView, uv = (mTrackProjX,
 * mTrackProjY), level = mnTrackScaleLevel (clamped to the frame's levels), viewcos = mTrackViewCos.  Lines: d_pos6[..][6] =
 * (float) of the six world coordinates, d_normal = (float) GetNormal; outputs seg = (mTrackProjX1, Y1, X2, Y2), level
 * unclamped as the reference leaves it (it passes the ORB log scale factor, Frame.cc:701).  cv::Mat arithmetic: `mRcw*P+mtcw`
 * as one gemm with double accumulation, cv::norm / Mat::dot in double (the definition pinned in oracle/plo.h). */
typedef struct plh_frame_view {
  float Rcw[9];            /* mRcw, row-major */
  float tcw[3];            /* mtcw */
  float Ow[3];             /* mOw */
  float fx, fy, cx, cy;
  float min_x, min_y, max_x, max_y;   /* mnMinX .. mnMaxY */
  float log_scale_factor;  /* mfLogScaleFactor */
  int32_t n_scale_levels;  /* mnScaleLevels */
} plh_frame_view;
/* The projection the pose-driven searches compute inline before the window lookup, for the caller that keeps its map on the
 * device: camera coordinates by `R*P + t` (one double-accumulated gemm, as above), then
 *   form 0  ORBmatcher::SearchByProjection(Cur, Last, th, mono) :1474-1484 and (Cur, pKF, found, th, ORBdist) :1614-1622:
 *           invz = (float)(1.0 / z); u = fx*xc*invz + cx; front = !(invz < 0)
 *   form 1  Fuse(pKF, vpMapPoints, th) :945-957, SearchByProjection(pKF, Scw, ..) :362-375: front = !(z < 0); invz = 1 / z in
 *           float; x = X*invz; u = fx*x + cx
 *   form 2  Fuse(pKF, Scw, ..) :1096-1108, SearchBySim3 :1253-1267: as form 1 with invz = (float)(1.0 / z).
 * d_front / d_uv are what goes into q_valid (together with the caller's map-side flags) and q_uv of the searches. */
PLH_API plh_status plh_frame_project_points_batch_dev(const plh_frame_view* d_views, int frames, const int32_t* d_nq, int qcap,
                                                      const float* d_pos, int form, uint8_t* d_front, float* d_uv, void* stream);
/* plh_orb_search_by_projection_frame_resident with the projection of ORBmatcher.cc:1474-1484 on the device as well (form 0 above): the
 * queries are the WORLD positions of the last frame's map points (q_world: 3 floats each) and `view` holds the current pose and
 * intrinsics; q_valid = the caller's map-side gates (pMP && !mvbOutlier[i] && a descriptor), `!(invzc < 0)` is added here.  One
 * staging round trip per TrackWithMotionModel search; no cv::Mat arithmetic per map point on the host. */
PLH_API plh_status plh_orb_search_by_projection_frame_resident_world(const plh_frame_points* f, const float* scale_factors, int nlevels,
                                                                     uint8_t* occupied, int nq, const plh_frame_view* view,
                                                                     const uint8_t* q_valid, const float* q_world, const int32_t* q_octave,
                                                                     const float* q_angle, const uint8_t* q_desc, const uint8_t* q_hasobs,
                                                                     float th, int mode, int check_ori, int32_t* assigned, int* nmatches);
PLH_API plh_status plh_frame_is_in_frustum_points_batch_dev(const plh_frame_view* d_views, int frames, const int32_t* d_nq, int qcap,
                                                            const float* d_pos, const float* d_normal, const float* d_min_dist,
                                                            const float* d_max_dist, float viewing_cos_limit, uint8_t* d_valid,
                                                            float* d_uv, int32_t* d_level, float* d_viewcos, void* stream);
PLH_API plh_status plh_frame_is_in_frustum_lines_batch_dev(const plh_frame_view* d_views, int frames, const int32_t* d_nq, int qcap,
                                                           const float* d_pos6, const float* d_normal, const float* d_min_dist,
                                                           const float* d_max_dist, float viewing_cos_limit, uint8_t* d_valid,
                                                           float* d_seg, int32_t* d_level, float* d_viewcos, void* stream);
/* The gates the back end's pose-driven searches apply to every map point between the pose transform and the window lookup -- the loops
 * in front of the searches of ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) ORBmatcher.cc:1591-1640,
 * SearchByProjection(KeyFrame*, Scw, ...) :337-395, Fuse(KeyFrame*, vpMapPoints, th) :945-975, Fuse(KeyFrame*, Scw, ...) :1096-1128 and
 * SearchBySim3 :1206-1290, :1313-1365 -- for all map points of a call at once:
 *     Pc = Rcw*P + tcw                       (one gemm, double accumulation; PLH_GATE_SECOND: then Pc = R2*Pc + t2, SearchBySim3's sR21 / t21)
 *     PLH_GATE_Z: skip when Pc.z < 0
 *     invz = 1 / Pc.z in float, or (float)(1.0 / z) with PLH_GATE_INVZ_DOUBLE
 *     u = fx*Pc.x*invz + cx, or x = Pc.x*invz; u = fx*x + cx with PLH_GATE_UV_NORMALISED
 *     inside [min, max] (Frame: u < min || u > max skips) or, PLH_GATE_KEYFRAME_BOUNDS, KeyFrame::IsInImage (min <= u < max)
 *     dist = |P - Ow| (cv::norm, double accumulation), or |Pc| with PLH_GATE_DIST_OF_TARGET;  min_dist_inv <= dist <= max_dist_inv
 *           (what MapPoint::GetMinDistanceInvariance / GetMaxDistanceInvariance return: 0.8f*mfMinDistance, 1.2f*mfMaxDistance)
 *     PLH_GATE_NORMAL: (P - Ow) . normal >= 0.5*dist
 *     max_dist (mfMaxDistance itself, may be NULL): level = MapPoint::PredictScale(dist, pKF / pF) = ceil(logf(max_dist / dist) /
 *           log_scale_factor) clamped to [0, n_scale_levels).  mfMaxDistance is a protected member: a host that reaches the map through
 *           the reference's classes passes NULL and calls pMP->PredictScale(dist[i], ..) on the points that pass (dist is returned).
 * valid[i] (in: the caller's map-side gates -- isBad(), already found, an empty descriptor; out: those && the gates above), uv, level
 * are q_valid / q_uv / q_level of plh_orb_search_by_projection_kf / _sim3, plh_orb_fuse_search and plh_orb_search_by_sim3. */
#define PLH_GATE_Z 1
#define PLH_GATE_INVZ_DOUBLE 2
#define PLH_GATE_UV_NORMALISED 4
#define PLH_GATE_KEYFRAME_BOUNDS 8
#define PLH_GATE_DIST_OF_TARGET 16
#define PLH_GATE_NORMAL 32
#define PLH_GATE_SECOND 64
typedef struct plh_point_gates {
  plh_frame_view view;     /* pose, intrinsics, image bounds, mfLogScaleFactor / mnScaleLevels of the camera searched in */
  float R2[9], t2[3];      /* PLH_GATE_SECOND */
  int32_t flags;           /* PLH_GATE_* */
} plh_point_gates;
PLH_API plh_status plh_map_point_gates(const plh_point_gates* gates, int n, const float* pos, const float* normal, const float* min_dist_inv,
                                       const float* max_dist_inv, const float* max_dist, uint8_t* valid, float* uv, float* dist,
                                       int32_t* level, int device);
PLH_API plh_status plh_map_point_gates_dev(const plh_point_gates* gates, int n, const float* d_pos, const float* d_normal,
                                           const float* d_min_dist_inv, const float* d_max_dist_inv, const float* d_max_dist,
                                           uint8_t* d_valid, float* d_uv, float* d_dist, int32_t* d_level, void* stream);
/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:249-314) / MapLine twin (MapLine.cpp:256-330) for many map
 * elements at once: set s owns the descriptor rows [d_offsets[s], d_offsets[s+1]) of d_desc (32 bytes each, <= 1024 rows);
 * d_best[s] = row (relative to the set) with the least median Hamming distance to the others, -1 for an empty set. */
PLH_API plh_status plh_distinctive_descriptor_batch_dev(const uint8_t* d_desc, const int32_t* d_offsets, int nsets,
                                                        int32_t* d_best, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: collecting the records of a frame batch that is sharded over the GPUs of a node
 *
 * Frames are independent, so every rank (one process per GPU) runs the front end on its own frames with no collective
 * inside; the one exchange is the gather of the fixed-stride result records -- counts, plh_keypoint records, ORB
 * descriptors, plh_keyline records, LBD descriptors, line equations, ... -- over RCCL (xGMI inside a node).  RCCL is bound
 * at run time (the copy already loaded in the process, else librccl.so.1); a host that never calls these never loads it.
 * ------------------------------------------------------------------------------------------- */
typedef struct plh_comm plh_comm;
#define PLH_COMM_ID_BYTES 128   /* = NCCL_UNIQUE_ID_BYTES */
/* ncclGetUniqueId: called by ONE rank, whose host hands the 128 bytes to the others (MPI, a socket, a file, torch.distributed). */
PLH_API plh_status plh_comm_unique_id(uint8_t id[PLH_COMM_ID_BYTES]);
/* ncclCommInitRank on `device`: collective over the `world` ranks that share `id`. */
PLH_API plh_status plh_comm_create(const uint8_t id[PLH_COMM_ID_BYTES], int rank, int world, int device, plh_comm** out);
/* Use an ncclComm_t the host already owns (not destroyed by plh_comm_destroy). */
PLH_API plh_status plh_comm_wrap(void* nccl_comm, int rank, int world, plh_comm** out);
PLH_API plh_status plh_comm_destroy(plh_comm* c);
PLH_API plh_status plh_comm_info(const plh_comm* c, int* rank, int* world, int* rccl_version);
/* One block = `bytes` bytes of records per rank (the same on every rank: fixed strides, frames per rank equal).
 * recv holds world * bytes, rank r's records at recv + r * bytes; it may be NULL on ranks that do not receive. */
typedef struct plh_gather_block {
  const void* send;
  void* recv;
  size_t bytes;
} plh_gather_block;
/* All blocks of a sub-batch in ONE grouped RCCL launch, asynchronous on `stream` (device pointers).
 * root = -1: every rank receives everything (ncclAllGather); root >= 0: only that rank does (ncclGather), which is what a
 * tracker that consumes the records on one GPU needs and moves 1/world of the all-gather's bytes per link. */
PLH_API plh_status plh_gather_records(plh_comm* c, const plh_gather_block* blocks, int nblocks, int root, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Line extractor  (replaces ORB_SLAM2::LINEextractor, include/LineExtractor.h:20-62)
 * ------------------------------------------------------------------------------------------- */
typedef struct plh_line plh_line;

typedef struct plh_line_params {   /* LINEextractor(numOctaves, scale, nLSDFeature, min_line_length) */
  int32_t num_octaves;             /* 1, or 2 with scale in [2, 3): what the reference itself can run -- LSDDetector::detect takes an
                                      int scale (1.2 -> 1), cv::pyrDown then asserts unless (int)scale == 2, and three or more
                                      octaves are undefined behaviour in BinaryDescriptor::computeImpl (refused: PLH_ERR_INVALID) */
  float scale;
  uint32_t n_lsd_feature;
  double min_line_length;
} plh_line_params;

PLH_API plh_status plh_line_create(const plh_line_params* p, int device, int rows, int cols, int max_batch, plh_line** out);
PLH_API plh_status plh_line_destroy(plh_line* h);
PLH_API int plh_line_capacity(const plh_line* h);   /* n_lsd_feature + 1 (LineExtractor.cpp:64 keeps index+1) */

/* Optional per-frame undistortion in front of LSD (Frame.cc:220-222): K = fx,fy,cx,cy ; D = k1,k2,p1,p2,k3.
 * Maps are built once here (the reference rebuilds them every frame). Pass NULL D or all-zero D to disable. */
PLH_API plh_status plh_line_set_undistort(plh_line* h, const float K[4], const float D[5]);

/* One frame, host buffers: LINEextractor::operator() (LineExtractor.cpp:26-93).
 * mask may be NULL (no filtering); a non-NULL mask must be rows x cols u8 (size mismatch is the
 * reference's std::runtime_error -> PLH_ERR_INVALID). keylines[cap], desc[cap*32], linefn[cap*3] (double). */
PLH_API plh_status plh_line_extract(plh_line* h, const uint8_t* img, int rows, int cols, size_t step,
                                    const uint8_t* mask, plh_keyline* keylines, uint8_t* desc, double* linefn,
                                    int cap, int* n_out);
PLH_API plh_status plh_line_extract_batch_dev(plh_line* h, const uint8_t* d_imgs, int batch, size_t frame_stride,
                                              const uint8_t* d_mask, plh_keyline* d_keylines, uint8_t* d_desc,
                                              double* d_linefn, int32_t* d_n, void* stream);
/* The refine level of the cv::LineSegmentDetector behind LSDDetector::detect (LineExtractor.cpp:39-40).
 * PLH_LSD_REFINE_ADV (the library's default since round 5): region2rect + refine() + rect_improve() -- a rectangle is kept only
 * if its NFA says it is meaningful, after up to five kinds of adjustment.  This is what the SYSTEM opencv_contrib LSDDetector
 * passes as published for 3.x (createLineSegmentDetector(LSD_REFINE_ADV)), and the system module is the one LineExtractor.cpp
 * links (include/auxiliar.h:11-16 includes <opencv2/line_descriptor/descriptor.hpp>; the twin in the tree is commented out).
 * PLH_LSD_REFINE_STD: without rect_improve(): what the un-linked line_descriptor twin in the reference's tree creates
 * (Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:149, createLineSegmentDetector() with its default).  A maintainer
 * whose OpenCV build differs picks the other one. */
#define PLH_LSD_REFINE_STD 0
#define PLH_LSD_REFINE_ADV 1
/* What a new handle starts with.  Chosen when the library is BUILT (-DPLH_LSD_REFINE_DEFAULT=PLH_LSD_REFINE_STD for the other
 * one), so that a drop-in does not inherit a level unknowingly; plh_lsd_refine_default() reports the
 * choice of the loaded library. */
#ifndef PLH_LSD_REFINE_DEFAULT
#define PLH_LSD_REFINE_DEFAULT PLH_LSD_REFINE_ADV
#endif
PLH_API int plh_lsd_refine_default(void);
PLH_API plh_status plh_line_set_refine(plh_line* h, int level);
/* The density screen of region growing (on by default): the density decisions of refine() / reduce_region_radius() are taken
 * from a float bracket of the rectangle's density whenever the bracket is clear, the exact rectangle is evaluated only when it
 * straddles the threshold or is itself needed.  on = 0 evaluates the exact rectangle for every decision: identical segments
 * (asserted by the tests), the instruction stream of rounds 1-3 -- an A/B and test switch, no counterpart in the reference. */
PLH_API plh_status plh_line_set_screen(plh_line* h, int on);
/* Scheduling hooks around the region-growing launch of the following plh_line_extract_batch_dev calls (hipEvent_t, owned by the
 * caller, NULL = none): the launch waits for wait_before; record_after is recorded behind it.  No counterpart in the reference
 * (Frame.cc:224-227 runs its two extractors on two threads); used by plh_frontend_* for small resident batches. */
PLH_API plh_status plh_line_set_grow_events(plh_line* h, void* wait_before, void* record_after);

/* Wavefronts per frame of LSD's region growing (cv::LineSegmentDetector's region_grow / refine loop, the sequential core
 * of LINEextractor::operator(), LineExtractor.cpp:40).  -1 (default): by batch size -- small batches (the per-frame call of
 * Frame.cc:224-227) run several wavefronts per frame as optimistic transactions with in-order commit, large batches one
 * wavefront per frame; 0: always one; n in 2..16: always n.  The segments are identical in every setting. */
PLH_API plh_status plh_line_set_grow_waves(plh_line* h, int waves);
/* Allocates now what the first plh_line_extract_batch_dev call with `batch` frames would allocate at the current settings (the
 * several-wavefronts-per-frame workspace: ~36 MB per 640x480 frame at 8 wavefronts; LSD_REFINE_ADV's records), so that a host
 * that builds its pipeline up front sees PLH_ERR_ALLOC here and not at its first step.  plh_frontend_create calls it. */
PLH_API plh_status plh_line_reserve(plh_line* h, int batch);
/* Tuning of the several-wavefronts-per-frame schedule (same segments for any value): run_ahead = how many seeds a wavefront may
 * start ahead of the commits (1..448, default 448), drain_gap = posted transactions that make a wavefront commit (>= 1, default
 * 8).  Values <= 0 restore the defaults. */
PLH_API plh_status plh_line_set_grow_tuning(plh_line* h, int run_ahead, int drain_gap);
/* Capacity flags of the most recent extract call (see plh_orb_status; a handle carries ONE call in flight: the flags are
 * cleared in stream order by the next call on it).  bit 2: LSD produced more segments than the
 * segment list holds (|scaled pixels| / min_reg_size + 16 -- a hard bound, so never expected).  bit 4 (16): the
 * multi-wavefront region growing gave up on a wait that lasted seconds (a lost wake-up would otherwise hang the GPU): the
 * call's lines are void; the handle's workspace is re-initialised by this query. */
PLH_API plh_status plh_line_status(plh_line* h, int* flags);
/* Per-stage device time (HIP events on the caller's stream): 0 = image prep + level-line field + seed order,
 * 1 = LSD region growing (k_lsd_grow), 2 = the kept regions' rectangles (k_lsd_rects; LSD_REFINE_ADV's rect_improve; a second
 * octave's whole LSD) + KeyLine selection, 3 = LBD (blur + Sobel + descriptor). */
PLH_API plh_status plh_line_set_profiling(plh_line* h, int on);
PLH_API plh_status plh_line_kernel_ms(plh_line* h, int stage, double* total_ms, int* intervals);
/* parity taps */
PLH_API plh_status plh_line_read_segments(plh_line* h, int b, float* out_xyxy, int cap, int* n_out);

/* ---------------------------------------------------------------------------------------------
 * Batch front end: the host side of the throughput path (what Frame::Frame(), Frame.cc:193-276, and Tracking's
 * frame-to-frame matching do per frame, for a batch of independent frames resident in device memory).
 * One step = ORBextractor + undistort / LINEextractor + Frame::ComputeBoW per frame, ORBmatcher::SearchByBoW and
 * LSDmatcher::SearchDouble between frame b and frame b + 1 (the last frame of a sub-batch is matched against its first).
 * The handle owns `nsplit` sub-batches (extractor handles, record buffers, a high-priority stream for the line chain and a
 * second one for the ORB chain, events); a sub-batch depends on its own previous step only, so un-joined steps overlap.
 * ------------------------------------------------------------------------------------------- */
typedef struct plh_frontend plh_frontend;

/* plh_frontend_params::lsd_refine.  0 -- what a zero-initialised struct holds -- is the LIBRARY's default, so that the usual C
 * idiom does not silently force a level.  The two explicit levels carry a tag in bits 8.. so that they overlap NEITHER
 * PLH_LSD_REFINE_STD / _ADV (0 / 1: the plh_line_set_refine values, which round 4 also took here) NOR round 5's 1 / 2: a caller that
 * assigns PLH_LSD_REFINE_ADV (1) to this field out of habit got LSD_REFINE_STD from round 5's encoding without a word;
 * now 1 and 2 are refused by plh_frontend_create (PLH_ERR_INVALID, with the reason in plh_last_error()). */
#define PLH_FRONTEND_REFINE_LIBRARY 0
#define PLH_FRONTEND_REFINE_STD 0x100   /* 0x100 | PLH_LSD_REFINE_STD */
#define PLH_FRONTEND_REFINE_ADV 0x101   /* 0x100 | PLH_LSD_REFINE_ADV */

typedef struct plh_frontend_params {
  uint32_t struct_size;            /* sizeof(plh_frontend_params) of the header the caller was compiled against: plh_frontend_create
                                      refuses any other value instead of reading past an older caller's struct (ABI break of
                                      round 5) */
  int32_t rows, cols;
  plh_orb_params orb;              /* ORBextractor(nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST), Tracking.cc:96-131 */
  plh_line_params line;            /* LINEextractor(1, scale, nLSDFeature, min_line_length) */
  int32_t undistort;               /* 1: remap with K / D in front of LSD (Frame.cc:220-222) */
  float K[4], D[5];
  int32_t bow_levelsup;            /* Frame::ComputeBoW: 4 (Frame.cc:911) */
  int32_t orb_th_low;              /* ORBmatcher::TH_LOW = 50 */
  float orb_nnratio;               /* ORBmatcher(0.7, true) of TrackReferenceKeyFrame (Tracking.cc:1151) */
  int32_t orb_check_orientation;
  float line_th, line_nnratio;     /* LSDmatcher::SearchDouble: TH_LOW = 50, mfNNratio */
  int32_t external_records;        /* 1: the caller supplies the record buffers (plh_frontend_bind_records) */
  int32_t lsd_refine;              /* PLH_FRONTEND_REFINE_LIBRARY (0: the library's default, plh_lsd_refine_default()),
                                      PLH_FRONTEND_REFINE_STD, PLH_FRONTEND_REFINE_ADV: every sub-batch's LINEextractor */
} plh_frontend_params;

/* Device pointers of one sub-batch's records, frames [first, first + frames) of the batch.  Per-frame arrays have
 * frames + 1 slots where a successor is matched (slot `frames` = copy of slot 0): kps, desc, n, nid, word, bow_*, kl, ldesc,
 * lfn (frames + 1 slots as well), nl.  m_orb[frames][orb_capacity] = SearchByBoW's vpMapPointMatches index per feature of
 * the successor (-1: none), m_line[frames][line_capacity] = SearchDouble's match per line of the frame. */
typedef struct plh_frontend_records {
  int32_t first, frames, orb_capacity, line_capacity;
  plh_keypoint* kps; uint8_t* desc; int32_t* n;
  int32_t *nid, *word, *bow_word; double* bow_value; int32_t* bow_n;
  plh_keyline* kl; uint8_t* ldesc; double* lfn; int32_t* nl;
  int32_t *m_orb, *nm_orb, *m_line, *nm_line;
} plh_frontend_records;

PLH_API plh_status plh_frontend_create(const plh_frontend_params* p, const plh_vocab* voc, int batch, int nsplit, int device,
                                       plh_frontend** out);
PLH_API plh_status plh_frontend_destroy(plh_frontend* fe);
PLH_API int plh_frontend_parts(const plh_frontend* fe);
/* the sub-batch's extractor handles (profiling, plh_line_set_grow_waves, ...) and its record buffers */
PLH_API plh_status plh_frontend_handles(plh_frontend* fe, int part, plh_orb** orb, plh_line** line);
PLH_API plh_status plh_frontend_records_of(plh_frontend* fe, int part, plh_frontend_records* out);
/* external_records = 1: the buffers of a sub-batch (every pointer set, sized by first / frames / capacities of
 * plh_frontend_records_of, which reports the capacities with null pointers until they are bound) */
PLH_API plh_status plh_frontend_bind_records(plh_frontend* fe, int part, const plh_frontend_records* r);
/* Enqueue one pass over d_imgs[batch] (u8, frame_stride bytes apart) behind `stream`; join = 1 makes `stream` wait for it
 * (plh_frontend_join does that later).  overlap = 0 runs both halves of every sub-batch on `stream` itself. */
PLH_API plh_status plh_frontend_step(plh_frontend* fe, const uint8_t* d_imgs, size_t frame_stride, void* stream, int join);
PLH_API plh_status plh_frontend_join(plh_frontend* fe, void* stream);
PLH_API plh_status plh_frontend_set_overlap(plh_frontend* fe, int on);
/* RCCL gather of every sub-batch's records (n, kps, desc, nl, kl, ldesc, lfn: PLH_FRONTEND_GATHERED blocks each, sizes from
 * plh_frontend_gather_bytes) on `comm_stream`, as soon as that sub-batch is done; recv[parts * PLH_FRONTEND_GATHERED] device
 * buffers of world x bytes each on receiving ranks (root = -1: all ranks), NULL elsewhere.  A sub-batch's next step waits
 * for its own gather only. */
#define PLH_FRONTEND_GATHERED 7
PLH_API plh_status plh_frontend_gather(plh_frontend* fe, plh_comm* comm, int root, void* const* recv, void* comm_stream);
PLH_API plh_status plh_frontend_gather_bytes(const plh_frontend* fe, int part, size_t bytes[PLH_FRONTEND_GATHERED]);
/* Waits for everything enqueued; flags = OR of the extractors' capacity flags (ORB in bits 0-7, lines in bits 8-15). */
PLH_API plh_status plh_frontend_status(plh_frontend* fe, int* flags);

#ifdef __cplusplus
}
#endif
#endif /* PLSLAM_HIP_H */
