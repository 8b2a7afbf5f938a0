// Drop-in replacement for the reference's include/ORBVocabulary.h (ORB_SLAM2::ORBVocabulary, :27-31 -- there a typedef of
// DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>): a class of that name DERIVED from the reference's own template, so that every
// member the rest of the system uses keeps working unchanged on the host tree --
//     System.cc:69-75        new ORBVocabulary(); loadFromTextFile / loadFromBinaryFile
//     KeyFrameDatabase.cc    mpVoc->size(), mpVoc->score(v1, v2)          LoopClosing.cc:134   mpORBVocabulary->score(...)
// -- while the one call on the per-frame path,
//     Frame::ComputeBoW     (src/Frame.cc:906-913)     mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4);
//     KeyFrame::ComputeBoW  (src/KeyFrame.cc:76-83)    the same on the keyframe's descriptors
// (TemplatedVocabulary.h:1139-1205 -- a VIRTUAL member, :145) runs on the GPU: the tree descent of every descriptor
// (k_bow_transform) and the BowVector accumulation / normalisation (k_bow_vector) through plh_vocab_transform.  The device copy of
// the tree is made from the host tree the reference's own loader has just built (m_nodes is a protected member), so a vocabulary
// file is parsed once, by the reference's code, and node / word ids are the reference's by construction.  BowVector values are
// the reference's doubles (sums in std::map order), FeatureVector lists are in feature order: tests/test_adaptor_exec.py runs the
// reference's own Frame::ComputeBoW through this class against the same method over the reference's DBoW2.
// There is no CPU fallback: without a device transform() throws.
#ifndef PLSLAM_HIP_ADAPTOR_ORBVOCABULARY_H
#define PLSLAM_HIP_ADAPTOR_ORBVOCABULARY_H
#define ORBVOCABULARY_H   // the include guard of the reference's own header (Frame.h, KeyFrame.h, KeyFrameDatabase.h, LoopClosing.h,
// System.h and Tracking.h include it as a sibling file)

#include <cstdint>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "Thirdparty/DBoW2/DBoW2/FORB.h"
#include "Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h"

#include "plslam_hip.h"

namespace ORB_SLAM2 {

class ORBVocabulary : public DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> {
  typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> Base;

 public:
  ORBVocabulary(int k = 10, int L = 5, DBoW2::WeightingType weighting = DBoW2::TF_IDF, DBoW2::ScoringType scoring = DBoW2::L1_NORM,
                int device = 0)
      : Base(k, L, weighting, scoring), mDevice(device), mDev(nullptr) {}
  ORBVocabulary(const std::string& filename, int device = 0) : Base(filename), mDevice(device), mDev(nullptr) {}
  ORBVocabulary(const ORBVocabulary& o) : Base(o), mDevice(o.mDevice), mDev(nullptr) {}
  ORBVocabulary& operator=(const ORBVocabulary& o) {
    Base::operator=(o);
    Invalidate();
    mDevice = o.mDevice;
    return *this;
  }
  virtual ~ORBVocabulary() { plh_vocab_destroy(mDev); }

  // the loaders System.cc:69-75 calls (non-virtual in the reference: hidden by name, the host tree is the reference's own)
  bool loadFromTextFile(const std::string& filename) {
    Invalidate();
    return Base::loadFromTextFile(filename);
  }
  bool loadFromBinaryFile(const std::string& filename) {
    Invalidate();
    return Base::loadFromBinaryFile(filename);
  }
  // everything else that rebuilds the tree is virtual in the reference
  virtual void create(const std::vector<std::vector<DBoW2::FORB::TDescriptor> >& training_features) {
    Invalidate();
    Base::create(training_features);
  }
  virtual void create(const std::vector<std::vector<DBoW2::FORB::TDescriptor> >& training_features, int k, int L) {
    Invalidate();
    Base::create(training_features, k, L);
  }
  virtual void create(const std::vector<std::vector<DBoW2::FORB::TDescriptor> >& training_features, int k, int L,
                      DBoW2::WeightingType weighting, DBoW2::ScoringType scoring) {
    Invalidate();
    Base::create(training_features, k, L, weighting, scoring);
  }
  virtual void load(const cv::FileStorage& fs, const std::string& name = "vocabulary") {
    Invalidate();
    Base::load(fs, name);
  }
  virtual int stopWords(double minWeight) {
    Invalidate();
    return Base::stopWords(minWeight);
  }

  using Base::transform;   // the single-feature and BowVector-only forms stay the reference's
  // Frame::ComputeBoW / KeyFrame::ComputeBoW (TemplatedVocabulary.h:1139-1205)
  virtual void transform(const std::vector<DBoW2::FORB::TDescriptor>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv,
                         int levelsup) const {
    v.clear();
    fv.clear();
    if (empty()) return;   // :1146-1149
    const int n = (int)features.size();
    if (n == 0) return;
    const plh_vocab* dev = Device();
    std::vector<uint8_t> desc((size_t)n * 32);
    for (int i = 0; i < n; i++) {
      const cv::Mat& d = features[i];
      if (d.type() != CV_8U || d.total() != 32 || !d.isContinuous()) throw std::runtime_error("ORBVocabulary::transform: a descriptor is not 1 x 32 CV_8U");
      std::memcpy(&desc[(size_t)i * 32], d.data, 32);
    }
    std::vector<int32_t> nid(n), word(n), bw(n);
    std::vector<double> bv(n);
    int nw = 0;
    if (plh_vocab_transform(dev, desc.data(), n, levelsup, nid.data(), word.data(), bw.data(), bv.data(), &nw) != PLH_OK)
      throw std::runtime_error(std::string("ORBVocabulary::transform: ") + plh_last_error());
    // BowVector / FeatureVector are std::maps: hinted inserts in ascending key order are O(1) each
    for (int s = 0; s < nw; s++) v.insert(v.end(), std::make_pair((DBoW2::WordId)bw[s], (DBoW2::WordValue)bv[s]));
    for (int i = 0; i < n; i++)
      if (nid[i] >= 0) fv.addFeature((DBoW2::NodeId)nid[i], (unsigned int)i);   // feature order, as the reference's loop adds them
  }

  // the device handle (NULL until the first transform): for callers that keep descriptors on the device and use
  // plh_vocab_transform_batch_dev themselves
  const plh_vocab* DeviceHandle() const { return Device(); }

 private:
  void Invalidate() {
    std::lock_guard<std::mutex> lock(mMutex);
    plh_vocab_destroy(mDev);
    mDev = nullptr;
  }
  // the flat device tree from m_nodes: arrays in the reference's node numbering (plh_vocab_create)
  const plh_vocab* Device() const {
    std::lock_guard<std::mutex> lock(mMutex);   // Tracking and LocalMapping both call ComputeBoW
    if (mDev) return mDev;
    const int nn = (int)m_nodes.size();
    std::vector<int32_t> parent(nn, 0);
    std::vector<uint8_t> leaf(nn, 0), desc((size_t)nn * 32, 0);
    std::vector<double> weight(nn, 0.0);
    for (int i = 1; i < nn; i++) {
      const Node& nd = m_nodes[i];
      parent[i] = (int32_t)nd.parent;
      leaf[i] = nd.isLeaf() ? 1 : 0;
      weight[i] = nd.weight;
      if (nd.descriptor.total() == 32) std::memcpy(&desc[(size_t)i * 32], nd.descriptor.data, 32);
    }
    plh_vocab* h = nullptr;
    if (plh_vocab_create(m_k, m_L, (int)m_scoring, (int)m_weighting, nn, parent.data(), leaf.data(), desc.data(), weight.data(), mDevice, &h) !=
        PLH_OK)
      throw std::runtime_error(std::string("ORBVocabulary: cannot put the vocabulary on the device: ") + plh_last_error());
    mDev = h;
    return mDev;
  }

  int mDevice;
  mutable plh_vocab* mDev;
  mutable std::mutex mMutex;
};

}  // namespace ORB_SLAM2

#endif
