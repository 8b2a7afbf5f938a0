// oracle/_ref: the reference's own DBoW2 (Thirdparty/DBoW2, compiled from the sources where they lie under
// /root/reference by oracle/ref/build_ref.sh) behind a small C interface, used to PIN the oracle's restatement of
//   DBoW2::FORB::distance                       Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101
//   TemplatedVocabulary::loadFromTextFile       Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1350-1438
//   TemplatedVocabulary::transform (both forms) Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1139-1271
// against the real code.  TEST INFRASTRUCTURE ONLY: nothing under pl-slam_amd/ links or loads this.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "DBoW2/FORB.h"
#include "DBoW2/TemplatedVocabulary.h"

typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> RefVocabulary;   // include/ORBVocabulary.h:30-31
struct ORBVocabulary : RefVocabulary {   // the per-feature transform (word, weight, node) is a protected member
  using RefVocabulary::transform;
};

static cv::Mat row_of(const uint8_t* d) {
  cv::Mat m(1, 32, CV_8U);
  std::memcpy(m.data, d, 32);
  return m;
}

extern "C" {

int ref_forb_distance(const uint8_t* a, const uint8_t* b) { return DBoW2::FORB::distance(row_of(a), row_of(b)); }

void* ref_voc_load_text(const char* path) {
  ORBVocabulary* v = new ORBVocabulary();
  if (!v->loadFromTextFile(path)) { delete v; return nullptr; }
  return v;
}
// TemplatedVocabulary::loadFromBinaryFile / saveToBinaryFile / saveToTextFile (TemplatedVocabulary.h:1465-1536, 1442-1462):
// the file formats plh_vocab_load_binary / plh_vocab_load_text read.
void* ref_voc_load_binary(const char* path) {
  ORBVocabulary* v = new ORBVocabulary();
  if (!v->loadFromBinaryFile(path)) { delete v; return nullptr; }
  return v;
}
void ref_voc_save_binary(void* v, const char* path) { static_cast<ORBVocabulary*>(v)->saveToBinaryFile(path); }
void ref_voc_save_text(void* v, const char* path) { static_cast<ORBVocabulary*>(v)->saveToTextFile(path); }
void ref_voc_free(void* v) { delete static_cast<ORBVocabulary*>(v); }
int ref_voc_size(void* v) { return (int)static_cast<ORBVocabulary*>(v)->size(); }

// per feature: word id, weight and the node `levelsup` levels above the leaf (the single-feature transform)
void ref_voc_transform_each(void* vp, const uint8_t* desc, int n, int levelsup, int32_t* word, double* weight, int32_t* node) {
  const ORBVocabulary* v = static_cast<ORBVocabulary*>(vp);
  for (int i = 0; i < n; i++) {
    DBoW2::WordId w;
    DBoW2::WordValue wt;
    DBoW2::NodeId nid;
    v->transform(row_of(desc + (size_t)i * 32), w, wt, &nid, levelsup);
    word[i] = (int32_t)w; weight[i] = wt; node[i] = (int32_t)nid;
  }
}

// Frame::ComputeBoW (src/Frame.cc:906-913): mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4).
// Outputs the BowVector as (word, value) pairs in map order and, per feature, the FeatureVector node it was filed
// under (-1 if the word is stopped).  Returns the number of BowVector entries.
int ref_voc_transform(void* vp, const uint8_t* desc, int n, int levelsup, int32_t* bow_word, double* bow_value, int cap,
                      int32_t* feat_node) {
  const ORBVocabulary* v = static_cast<ORBVocabulary*>(vp);
  std::vector<cv::Mat> feats;
  feats.reserve(n);
  for (int i = 0; i < n; i++) feats.push_back(row_of(desc + (size_t)i * 32));
  DBoW2::BowVector bv;
  DBoW2::FeatureVector fv;
  v->transform(feats, bv, fv, levelsup);
  for (int i = 0; i < n; i++) feat_node[i] = -1;
  for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it)
    for (size_t j = 0; j < it->second.size(); j++) feat_node[it->second[j]] = (int32_t)it->first;
  int k = 0;
  for (DBoW2::BowVector::const_iterator it = bv.begin(); it != bv.end() && k < cap; ++it, ++k) {
    bow_word[k] = (int32_t)it->first;
    bow_value[k] = it->second;
  }
  return (int)bv.size();
}

}  // extern "C"
