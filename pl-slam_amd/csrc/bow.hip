// DBoW2 vocabulary-tree descent on the GPU ("next" row 8f-1: Frame::ComputeBoW feeds ORBmatcher::SearchByBoW).
//   k_bow_transform   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
//                     reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1255 (+ FORB::distance, FORB.cpp:81-101)
// The vocabulary is passed as flat arrays (children of a node are contiguous: child_start/child_count), which is
// how DBoW2 numbers nodes when it builds or loads a tree.  One thread per descriptor; the tree's upper levels
// stay resident in L2.  Integer Hamming arithmetic: bit-exact.
#include "plh_common.h"

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
                                                       const float* weight, int nidLevel, int32_t* nidOut, int32_t* wordOut) {
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
  nidOut[o] = stopped ? -1 : nid;
  wordOut[o] = stopped ? -1 : wordId[final_id];
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
                     L - levelsup, d_nid, d_word);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}
