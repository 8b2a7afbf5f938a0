// plh_vocab: a DBoW2 vocabulary tree in the flat form of the descent kernel (vocab_host.hip builds it, bow.hip uses it).
#pragma once
#include <cstdint>
#include <vector>

struct plh_vocab {
  int device = 0;
  int k = 0, L = 0, scoring = 0, weighting = 0, nNodes = 0, nWords = 0;
  bool identity = true;                 // flat index == reference NodeId
  // host copies (flat order)
  std::vector<uint8_t> hDesc;
  std::vector<int32_t> hChildStart, hChildCount, hWordId, hNodeId;
  std::vector<double> hWeight, hWordWeight;   // per node / per word (WordValue is double)
  std::vector<float> hWeightF;                // sign-preserving float copy: the descent only tests `w > 0`
  // device copies
  uint8_t* dDesc = nullptr;
  int32_t *dChildStart = nullptr, *dChildCount = nullptr, *dWordId = nullptr, *dNodeId = nullptr;
  float* dWeightF = nullptr;
  double* dWordWeight = nullptr;
};
