// DBoW2 vocabulary-tree descent on the GPU ("next" row 8f-1: Frame::ComputeBoW feeds ORBmatcher::SearchByBoW).
//   k_bow_transform   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
//                     reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1255 (+ FORB::distance, FORB.cpp:81-101)
// The vocabulary is passed as flat arrays (children of a node are contiguous: child_start/child_count), which is
// how DBoW2 numbers nodes when it builds or loads a tree.  One thread per descriptor; the tree's upper levels
// stay resident in L2.  Integer Hamming arithmetic: bit-exact.
//   k_bow_vector      the BowVector of TemplatedVocabulary::transform(features, v, fv, levelsup) (:1139-1205): per frame the
//                     words of the live features in ascending order with their accumulated, normalised weights
//                     (BowVector::addWeight / addIfNotExist / normalize, BowVector.cpp:34-84).  WordValue is double and the
//                     reference adds / sums in map order, so the kernel sorts the word ids (bitonic, LDS), accumulates each
//                     run by repeated addition and sums the norm sequentially in word order: bit-identical doubles.
#include "plh_common.h"
#include "vocab.h"
#include "plh_stage.h"

namespace plh {

__device__ __forceinline__ void load256(const uint8_t* p, unsigned long long w[4]) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  w[0] = (unsigned long long)a.x | ((unsigned long long)a.y << 32);
  w[1] = (unsigned long long)a.z | ((unsigned long long)a.w << 32);
  w[2] = (unsigned long long)b.x | ((unsigned long long)b.y << 32);
  w[3] = (unsigned long long)b.z | ((unsigned long long)b.w << 32);
}

__global__ void __launch_bounds__(256) k_bow_transform(const uint8_t* desc, const int* nArr, int cap, const uint8_t* nodeDesc,
                                                       const int* childStart, const int* childCount, const int* wordId,
                                                       const float* weight, int nidLevel, const int* nodeId, int32_t* nidOut,
                                                       int32_t* wordOut) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  const long long o = (long long)b * cap + i;
  if (i >= nArr[b]) { nidOut[o] = -1; wordOut[o] = -1; return; }
  unsigned long long f[4], d[4];
  load256(desc + o * 32, f);
  int final_id = 0, level = 0, nid = nidLevel <= 0 ? 0 : -1;
  do {
    ++level;
    const int cs = childStart[final_id], cc = childCount[final_id];
    int best = cs, best_d = 1 << 30;
    for (int k = 0; k < cc; k++) {
      load256(nodeDesc + (long long)(cs + k) * 32, d);
      const int dist = hamming256(f, d);
      if (dist < best_d) { best_d = dist; best = cs + k; }   // first minimum wins (strict '<')
    }
    final_id = best;
    if (level == nidLevel) nid = final_id;
  } while (childCount[final_id] > 0);
  const bool stopped = !(weight[final_id] > 0.f);   // transform(): `if (w > 0)` -- stopped words carry no feature
  if (nodeId && nid >= 0) nid = nodeId[nid];        // renumbered tree: back to the reference's NodeId
  nidOut[o] = stopped ? -1 : nid;
  wordOut[o] = stopped ? -1 : wordId[final_id];
}

// One block per frame.  LDS: keys[cap2] (sorted word ids, INT_MAX = no word), pos[cap2 + 1] (start of every run of equal
// keys), val[cap2] (double).  weighting / scoring as in DBoW2's enums (BowVector.h:36-53).
__global__ void __launch_bounds__(256) k_bow_vector(const int32_t* word, const int* nArr, int cap, int cap2, const double* wordWeight,
                                                    int weighting, int scoring, int32_t* bowWord, double* bowValue, int32_t* bowN) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  __shared__ int s_part[256];
  __shared__ int s_m, s_nw;
  __shared__ double s_norm;
  double* val = (double*)smem;
  int* keys = (int*)(val + cap2);
  int* pos = keys + cap2;
  const int b = blockIdx.x, tid = threadIdx.x;
  const long long o = (long long)b * cap;
  const int n = max(0, min(nArr[b], cap));
  for (int i = tid; i < cap2; i += 256) {
    const int w = i < n ? word[o + i] : -1;
    keys[i] = w >= 0 ? w : 0x7fffffff;
  }
  if (tid == 0) { s_m = 0; s_nw = 0; }
  __syncthreads();
  for (int size = 2; size <= cap2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (cap2 >> 1); t += 256) {
        const int i = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)), j = i | stride;
        const bool up = (i & size) == 0;
        const int a = keys[i], c = keys[j];
        if ((a > c) == up) { keys[i] = c; keys[j] = a; }
      }
      __syncthreads();
    }
  // m = number of live keys; run starts
  for (int i = tid; i < cap2; i += 256)
    if (keys[i] != 0x7fffffff && (i + 1 == cap2 || keys[i + 1] == 0x7fffffff)) s_m = i + 1;
  __syncthreads();
  const int m = s_m;
  const int per = cap2 >> 8 ? cap2 >> 8 : 1;              // consecutive elements per thread (cap2 >= 256 or a single element)
  const int i0 = tid * per;
  int cnt = 0;
  for (int i = i0; i < min(i0 + per, m); i++) cnt += (i == 0 || keys[i] != keys[i - 1]);
  s_part[tid] = cnt;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int t = 0; t < 256; t++) { const int c = s_part[t]; s_part[t] = acc; acc += c; }
    s_nw = acc;
    pos[acc] = m;
  }
  __syncthreads();
  {
    int slot = s_part[tid];
    for (int i = i0; i < min(i0 + per, m); i++)
      if (i == 0 || keys[i] != keys[i - 1]) pos[slot++] = i;
  }
  __syncthreads();
  const int nw = s_nw;
  const bool tf = weighting == 0 || weighting == 1;       // TF_IDF || TF: addWeight per feature; IDF || BINARY: addIfNotExist
  const bool must = scoring != 5;                         // DotProductScoring does not normalise
  for (int s = tid; s < nw; s += 256) {
    const int id = keys[pos[s]], c = pos[s + 1] - pos[s];
    const double w = wordWeight[id];
    double v = w;
    if (tf)
      for (int k = 1; k < c; k++) v += w;                 // `vit->second += v` once per feature of the word
    if (tf && !must) v /= (double)nw;                     // :1175-1181
    val[s] = v;
  }
  __syncthreads();
  if (tid == 0) {
    double norm = 0.0;
    if (must) {
      if (scoring == 1) {                                 // L2
        for (int s = 0; s < nw; s++) norm += val[s] * val[s];
        norm = sqrt(norm);
      } else {
        for (int s = 0; s < nw; s++) norm += fabs(val[s]);
      }
    }
    s_norm = norm;
    bowN[b] = nw;
  }
  __syncthreads();
  const double norm = s_norm;
  for (int s = tid; s < nw; s += 256) {
    bowWord[o + s] = keys[pos[s]];
    bowValue[o + s] = (must && norm > 0.0) ? val[s] / norm : val[s];
  }
}

}  // namespace plh

using namespace plh;

extern "C" plh_status plh_bow_transform_batch_dev(const uint8_t* d_desc, const int32_t* d_n, int cap, int batch,
                                                  const uint8_t* d_node_desc, const int32_t* d_child_start,
                                                  const int32_t* d_child_count, const int32_t* d_word_id,
                                                  const float* d_weight, int L, int levelsup, int32_t* d_nid,
                                                  int32_t* d_word, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_desc || !d_n || !d_node_desc || !d_child_start || !d_child_count || !d_word_id || !d_weight || !d_nid || !d_word ||
      cap <= 0 || batch <= 0 || L < 1) {
    set_error("plh_bow_transform_batch_dev: invalid argument");
    return PLH_ERR_INVALID;
  }
  hipLaunchKernelGGL(k_bow_transform, dim3((cap + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, d_desc, (const int*)d_n,
                     cap, d_node_desc, (const int*)d_child_start, (const int*)d_child_count, (const int*)d_word_id, d_weight,
                     L - levelsup, (const int*)nullptr, d_nid, d_word);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

extern "C" plh_status plh_bow_vector_batch_dev(const int32_t* d_word, const int32_t* d_n, int cap, int batch,
                                               const double* d_word_weight, int weighting, int scoring, int32_t* d_bow_word,
                                               double* d_bow_value, int32_t* d_bow_n, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_word || !d_n || !d_word_weight || !d_bow_word || !d_bow_value || !d_bow_n || cap <= 0 || cap > 8192 || batch <= 0 ||
      weighting < 0 || weighting > 3 || scoring < 0 || scoring > 5) {
    set_error("plh_bow_vector_batch_dev: invalid argument (cap in 1..8192)");
    return PLH_ERR_INVALID;
  }
  int cap2 = 256;
  while (cap2 < cap) cap2 <<= 1;
  const size_t lds = (size_t)cap2 * 16 + 8;
  if (lds_request(k_bow_vector, lds, "plh_bow_vector_batch_dev") != PLH_OK) return PLH_ERR_INVALID;
  hipLaunchKernelGGL(k_bow_vector, dim3(batch), dim3(256), lds, (hipStream_t)stream, d_word, (const int*)d_n, cap, cap2, d_word_weight,
                     weighting, scoring, d_bow_word, d_bow_value, d_bow_n);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

// Frame::ComputeBoW (reference src/Frame.cc:906-913): mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4) for a batch.
extern "C" plh_status plh_vocab_transform_batch_dev(const plh_vocab* v, const uint8_t* d_desc, const int32_t* d_n, int cap, int batch,
                                                    int levelsup, int32_t* d_nid, int32_t* d_word, int32_t* d_bow_word,
                                                    double* d_bow_value, int32_t* d_bow_n, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!v || !d_desc || !d_n || !d_nid || !d_word || cap <= 0 || batch <= 0) {
    set_error("plh_vocab_transform_batch_dev: invalid argument");
    return PLH_ERR_INVALID;
  }
  hipLaunchKernelGGL(k_bow_transform, dim3((cap + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, d_desc, (const int*)d_n,
                     cap, v->dDesc, (const int*)v->dChildStart, (const int*)v->dChildCount, (const int*)v->dWordId, v->dWeightF,
                     v->L - levelsup, (const int*)v->dNodeId, d_nid, d_word);
  PLH_LAUNCH_CHECK();
  if (!d_bow_word && !d_bow_value && !d_bow_n) return PLH_OK;   // FeatureVector only
  return plh_bow_vector_batch_dev(d_word, d_n, cap, batch, v->dWordWeight, v->weighting, v->scoring, d_bow_word, d_bow_value,
                                  d_bow_n, stream);
}

// Frame::ComputeBoW / KeyFrame::ComputeBoW for ONE frame on host buffers (Frame.cc:906-913, KeyFrame.cc:76-83): what the drop-in
// ORBVocabulary::transform(features, v, fv, levelsup) calls (pl-slam_amd/adaptor/ORBVocabulary.h).  desc = n rows of 32 bytes
// (mDescriptors); nid[i] = FeatureVector node of feature i or -1 (stopped word), word[i] its word id; bow_word / bow_value (n
// entries of room) receive the *bow_n distinct words in ascending order with their normalised weights.  Thread-safe for concurrent
// callers on one handle (the tracking and the local-mapping thread both call ComputeBoW): the staging is the calling thread's own.
extern "C" plh_status plh_vocab_transform(const plh_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* nid, int32_t* word,
                                          int32_t* bow_word, double* bow_value, int* bow_n) {
  if (!v || n < 0 || !bow_n || (n > 0 && (!desc || !nid || !word || !bow_word || !bow_value))) {
    set_error("plh_vocab_transform: invalid argument");
    return PLH_ERR_INVALID;
  }
  *bow_n = 0;
  if (n == 0) return PLH_OK;
  if (n > 8192) { set_error("plh_vocab_transform: %d features (at most 8192 per frame)", n); return PLH_ERR_CAPACITY; }
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stager st;
  const size_t N = (size_t)n;
  plh_status rc = st.begin(v->device, Stager::padded(N * 32) + 4 * Stager::padded(N * 4) + Stager::padded(N * 8) + 2 * 256);
  if (rc != PLH_OK) return rc;
  const int32_t n32 = n;
  const uint8_t* dDesc = st.in(desc, N * 32);
  const int32_t* dN = st.in(&n32, 1);
  int32_t* dNid = st.out(nid, N);
  int32_t* dWord = st.out(word, N);
  int32_t* dBw = st.out(bow_word, N);
  double* dBv = st.out(bow_value, N);
  int32_t bn = 0;
  int32_t* dBn = st.out(&bn, 1);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = plh_vocab_transform_batch_dev(v, dDesc, dN, n, 1, levelsup, dNid, dWord, dBw, dBv, dBn, st.stream());
  if (rc != PLH_OK) return rc;
  if ((rc = st.download()) != PLH_OK) return rc;
  *bow_n = bn;
  return PLH_OK;
}
