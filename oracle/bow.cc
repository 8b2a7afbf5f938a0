// ORACLE (test infrastructure only -- see plo.h): the BowVector half of DBoW2's TemplatedVocabulary::transform and the
// vocabulary file formats, restated from the reference's vendored DBoW2.
//
//   plo_bow_vector        TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
//                         reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1139-1205 with
//                         BowVector::addWeight / addIfNotExist / normalize  (BowVector.cpp:34-84) and the normalisation
//                         rule of the scoring classes (ScoringObject.h:72-89: every scoring normalises with L1 except
//                         L2_NORM (L2) and DOT_PRODUCT (none)).
//   plo_vocab_parse_text  TemplatedVocabulary::loadFromTextFile   :1350-1438
//   plo_vocab_parse_bin   TemplatedVocabulary::loadFromBinaryFile :1465-1506 (layout written by saveToBinaryFile :1511-1536)
//
// WordValue is double (BowVector.h:25); the BowVector is a std::map<WordId, WordValue>, so the values below come out in
// ascending word order and every sum runs in that order.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "plo.h"

extern "C" {

// word[i]: word id of feature i as plo_bow_transform returns it (-1: stopped word, `w > 0` failed at :1168 / :1191).
// word_weight[id]: the weight transform(feature, id, w, ...) hands back for that word (m_words[id]->weight, :1251).
// weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (BowVector.h:36-42); scoring: 0 L1_NORM .. 5 DOT_PRODUCT (BowVector.h:45-53).
// Returns the number of (word, value) pairs written in map order.
int plo_bow_vector(const int32_t* word, int n, const double* word_weight, int weighting, int scoring, int32_t* out_word,
                   double* out_value, int cap) {
  std::map<int32_t, double> v;
  const bool must = scoring != 5;                // DotProductScoring: mustNormalize() == false
  const bool l2 = scoring == 1;
  if (weighting == 0 || weighting == 1) {        // TF_IDF || TF : v.addWeight(id, w) per feature, in feature order
    for (int i = 0; i < n; i++) {
      if (word[i] < 0) continue;
      const double w = word_weight[word[i]];
      std::map<int32_t, double>::iterator it = v.lower_bound(word[i]);
      if (it != v.end() && it->first == word[i]) it->second += w;
      else v.insert(it, std::make_pair(word[i], w));
    }
    if (!v.empty() && !must) {                   // :1175-1181 "unnecessary when normalizing"
      const double nd = (double)v.size();
      for (std::map<int32_t, double>::iterator it = v.begin(); it != v.end(); ++it) it->second /= nd;
    }
  } else {                                       // IDF || BINARY : v.addIfNotExist(id, w)
    for (int i = 0; i < n; i++) {
      if (word[i] < 0) continue;
      if (v.find(word[i]) == v.end()) v[word[i]] = word_weight[word[i]];
    }
  }
  if (must) {                                    // BowVector::normalize(norm), BowVector.cpp:62-84
    double norm = 0.0;
    if (!l2) {
      for (std::map<int32_t, double>::iterator it = v.begin(); it != v.end(); ++it) norm += std::fabs(it->second);
    } else {
      for (std::map<int32_t, double>::iterator it = v.begin(); it != v.end(); ++it) norm += it->second * it->second;
      norm = std::sqrt(norm);
    }
    if (norm > 0.0)
      for (std::map<int32_t, double>::iterator it = v.begin(); it != v.end(); ++it) it->second /= norm;
  }
  int k = 0;
  for (std::map<int32_t, double>::iterator it = v.begin(); it != v.end() && k < cap; ++it, ++k) {
    out_word[k] = it->first;
    out_value[k] = it->second;
  }
  return (int)v.size();
}

// ---- vocabulary files.  Both parsers fill per-node arrays in the reference's node numbering (node 0 = root, node i = the
// i-th record of the file): parent, is_leaf, 32-byte descriptor, weight (double).  Return the number of nodes including the
// root, or -1.  header[4] = k, L, scoring, weighting.  `cap` = capacity of the arrays in nodes.
static bool read_all(const char* path, std::vector<char>& buf) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  const long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  buf.resize((size_t)sz + 1);
  const size_t got = std::fread(buf.data(), 1, (size_t)sz, f);
  std::fclose(f);
  buf[got] = 0;
  buf.resize(got + 1);
  return true;
}

int plo_vocab_parse_text(const char* path, int32_t header[4], int32_t* parent, uint8_t* is_leaf, uint8_t* desc, double* weight,
                         int cap) {
  std::vector<char> buf;
  if (!read_all(path, buf)) return -1;
  char* p = buf.data();
  for (int i = 0; i < 4; i++) header[i] = (int32_t)std::strtol(p, &p, 10);     // "k L scoring weighting" (:1363-1370)
  if (header[0] < 0 || header[0] > 20 || header[1] < 1 || header[1] > 10 || header[2] < 0 || header[2] > 5 || header[3] < 0 ||
      header[3] > 3)
    return -1;                                                                 // :1372-1376
  while (*p && *p != '\n') p++;
  int n = 1;
  for (;;) {
    while (*p == '\n' || *p == '\r' || *p == ' ') p++;
    if (!*p) break;   // (the reference's eof loop would parse a trailing empty line into a node with an uninitialised parent)
    if (n >= cap) return -1;
    parent[n] = (int32_t)std::strtol(p, &p, 10);                               // :1404-1406
    is_leaf[n] = std::strtol(p, &p, 10) > 0;                                   // :1408-1409
    for (int k = 0; k < 32; k++) desc[(size_t)n * 32 + k] = (uint8_t)std::strtol(p, &p, 10);   // F::fromString (FORB.cpp:128-146)
    weight[n] = std::strtod(p, &p);                                            // `ssnode >> m_nodes[nid].weight` (double)
    n++;
  }
  parent[0] = -1; is_leaf[0] = 0; weight[0] = 0.0;
  std::memset(desc, 0, 32);
  return n;
}

int plo_vocab_parse_bin(const char* path, int32_t header[4], int32_t* parent, uint8_t* is_leaf, uint8_t* desc, double* weight,
                        int cap) {
  std::vector<char> buf;
  if (!read_all(path, buf)) return -1;
  const size_t sz = buf.size() - 1;
  if (sz < 24) return -1;
  uint32_t nb_nodes, size_node;
  std::memcpy(&nb_nodes, buf.data(), 4);
  std::memcpy(&size_node, buf.data() + 4, 4);
  std::memcpy(header, buf.data() + 8, 16);          // m_k, m_L, m_scoring, m_weighting (4-byte ints / enums)
  if (size_node != 4 + 32 + 4 + 1) return -1;       // parent, descriptor, float weight, bool is_leaf (:1517)
  // saveToBinaryFile writes nb_nodes = m_nodes.size() and then the nb_nodes - 1 non-root records.  The reference's reader
  // loops on !eof, so after the last record it processes its buffer once more: a phantom copy of the last node is appended
  // as one more child of that node's parent (and one more word).  It sits behind its original among the children with the
  // same descriptor and transform() takes the FIRST minimum, so it can never be chosen: it is not materialised here.
  const size_t nrec = (sz - 24) / size_node;
  if (nrec + 1 > (size_t)cap) return -1;
  const char* r = buf.data() + 24;
  for (size_t i = 0; i < nrec; i++, r += size_node) {
    const int n = (int)i + 1;
    std::memcpy(&parent[n], r, 4);
    std::memcpy(desc + (size_t)n * 32, r + 4, 32);
    float w;
    std::memcpy(&w, r + 36, 4);
    weight[n] = w;
    is_leaf[n] = r[40] != 0;
  }
  parent[0] = -1; is_leaf[0] = 0; weight[0] = 0.0;
  std::memset(desc, 0, 32);
  (void)nb_nodes;
  return (int)nrec + 1;
}

}  // extern "C"
