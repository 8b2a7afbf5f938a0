// DBoW2 vocabulary handle of libplslam_hip: the C++ host needs no Python to get ORBvoc onto the device.
//   plh_vocab_load_text    TemplatedVocabulary::loadFromTextFile    reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1350-1438
//   plh_vocab_load_binary  TemplatedVocabulary::loadFromBinaryFile  :1465-1506 (System.cc:69-75 picks one by file suffix)
//   plh_vocab_save_binary  TemplatedVocabulary::saveToBinaryFile    :1511-1536
//   plh_vocab_create       the same tree from arrays
// The tree is kept in the flat form the descent kernel wants -- children of a node contiguous, in the reference's child
// order -- and in the reference's node numbering whenever the file already has contiguous children (every vocabulary DBoW2
// itself built has: HKmeansStep appends the k children of a node back to back, :600-640).  Otherwise nodes are renumbered
// breadth first and `nodeId` maps a flat index back to the reference's NodeId, which is what FeatureVector keys must be.
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <string>
#include <charconv>
#include <vector>

#include "plh_common.h"
#include "vocab.h"

using namespace plh;

namespace {

struct RawTree {               // nodes in the reference's numbering, node 0 = root
  int k = 0, L = 0, scoring = 0, weighting = 0;
  std::vector<int32_t> parent;
  std::vector<uint8_t> leaf;
  std::vector<uint8_t> desc;   // 32 bytes per node
  std::vector<double> weight;
};

bool read_file(const char* path, std::vector<char>& buf) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  const long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (sz < 0) { std::fclose(f); return false; }
  buf.resize((size_t)sz + 1);
  const size_t got = std::fread(buf.data(), 1, (size_t)sz, f);
  std::fclose(f);
  buf.resize(got + 1);
  buf[got] = 0;
  return true;
}

void push_root(RawTree& t) {
  t.parent.assign(1, -1);
  t.leaf.assign(1, 0);
  t.desc.assign(32, 0);
  t.weight.assign(1, 0.0);
}

// Decimal integer / floating point fields separated by blanks; hand-rolled because ORBvoc.txt holds 37 million of them.
inline void skip_blanks(const char*& p) { while (*p == ' ' || *p == '\t' || *p == '\r') p++; }
inline bool parse_int(const char*& p, long& v) {
  skip_blanks(p);
  bool neg = false;
  if (*p == '-') { neg = true; p++; }
  if (*p < '0' || *p > '9') return false;
  long r = 0;
  int digits = 0;
  while (*p >= '0' && *p <= '9') {
    if (++digits > 18) return false;   // no field of a vocabulary file comes near: a run of digits this long is a corrupt file
    r = r * 10 + (*p++ - '0');
  }
  v = neg ? -r : r;
  return true;
}

plh_status parse_text(const char* path, RawTree& t) {
  std::vector<char> buf;
  if (!read_file(path, buf)) { set_error("plh_vocab_load_text: cannot read %s", path); return PLH_ERR_INVALID; }
  const char* p = buf.data();
  long h[4];
  for (int i = 0; i < 4; i++)
    if (!parse_int(p, h[i])) { set_error("plh_vocab_load_text: %s is not a DBoW2 text vocabulary", path); return PLH_ERR_INVALID; }
  // the reference's own sanity check (:1372-1376)
  if (h[0] < 0 || h[0] > 20 || h[1] < 1 || h[1] > 10 || h[2] < 0 || h[2] > 5 || h[3] < 0 || h[3] > 3) {
    set_error("plh_vocab_load_text: %s: header k=%ld L=%ld scoring=%ld weighting=%ld is out of range", path, h[0], h[1], h[2], h[3]);
    return PLH_ERR_INVALID;
  }
  t.k = (int)h[0]; t.L = (int)h[1]; t.scoring = (int)h[2]; t.weighting = (int)h[3];
  while (*p && *p != '\n') p++;
  push_root(t);
  for (;;) {
    while (*p == '\n' || *p == ' ' || *p == '\t' || *p == '\r') p++;
    if (!*p) break;   // blank lines (also a trailing one) are skipped; the reference would turn it into a stray node
    long pid, isleaf, b;
    if (!parse_int(p, pid) || !parse_int(p, isleaf)) { set_error("plh_vocab_load_text: %s: malformed node line %zu", path, t.parent.size()); return PLH_ERR_INVALID; }
    const size_t nid = t.parent.size();
    if (pid < 0 || (size_t)pid >= nid) { set_error("plh_vocab_load_text: %s: node %zu names parent %ld", path, nid, pid); return PLH_ERR_INVALID; }
    t.parent.push_back((int32_t)pid);
    t.leaf.push_back(isleaf > 0);
    for (int k = 0; k < 32; k++) {
      if (!parse_int(p, b)) { set_error("plh_vocab_load_text: %s: node %zu has a short descriptor", path, nid); return PLH_ERR_INVALID; }
      t.desc.push_back((uint8_t)b);
    }
    skip_blanks(p);
    // `ssnode >> weight` reads a double (WordValue) from THIS line: the field must not be taken from the next line (strtod
    // skips line feeds) nor depend on the process locale (from_chars never does)
    const char* le = p;
    while (*le && *le != '\n') le++;
    double w = 0.0;
    if (*p == '+') p++;
    const std::from_chars_result fr = std::from_chars(p, le, w);
    if (fr.ec != std::errc() || fr.ptr == p) { set_error("plh_vocab_load_text: %s: node %zu has no weight", path, nid); return PLH_ERR_INVALID; }
    t.weight.push_back(w);
    p = le;
  }
  return PLH_OK;
}

plh_status parse_binary(const char* path, RawTree& t) {
  std::vector<char> buf;
  if (!read_file(path, buf)) { set_error("plh_vocab_load_binary: cannot read %s", path); return PLH_ERR_INVALID; }
  const size_t sz = buf.size() - 1;
  uint32_t nb_nodes = 0, size_node = 0;
  int32_t h[4];
  if (sz < 24) { set_error("plh_vocab_load_binary: %s is too short", path); return PLH_ERR_INVALID; }
  memcpy(&nb_nodes, buf.data(), 4);
  memcpy(&size_node, buf.data() + 4, 4);
  memcpy(h, buf.data() + 8, 16);
  if (size_node != 41 || h[0] < 0 || h[0] > 20 || h[1] < 1 || h[1] > 10 || h[2] < 0 || h[2] > 5 || h[3] < 0 || h[3] > 3) {
    set_error("plh_vocab_load_binary: %s: not a DBoW2 ORB vocabulary (node size %u, k %d, L %d)", path, size_node, h[0], h[1]);
    return PLH_ERR_INVALID;
  }
  t.k = h[0]; t.L = h[1]; t.scoring = h[2]; t.weighting = h[3];
  // records: u32 parent, 32 descriptor bytes, float weight, bool is_leaf.  (The reference's !eof loop parses its last buffer
  // twice and hangs a copy of the last node behind the original under the same parent; equal descriptors and a first-minimum
  // descent mean that copy is never reached, so it is not created here.)
  const size_t nrec = (sz - 24) / 41;
  push_root(t);
  t.parent.reserve(nrec + 1);
  const char* r = buf.data() + 24;
  for (size_t i = 0; i < nrec; i++, r += 41) {
    uint32_t pid;
    float w;
    memcpy(&pid, r, 4);
    memcpy(&w, r + 36, 4);
    if (pid > i) { set_error("plh_vocab_load_binary: %s: node %zu names parent %u", path, i + 1, pid); return PLH_ERR_INVALID; }
    t.parent.push_back((int32_t)pid);
    t.desc.insert(t.desc.end(), (const uint8_t*)r + 4, (const uint8_t*)r + 36);
    t.weight.push_back((double)w);
    t.leaf.push_back(r[40] != 0);
  }
  (void)nb_nodes;
  return PLH_OK;
}

template <typename T>
plh_status upload(const std::vector<T>& v, T** d) {
  PLH_HIP(hipMalloc((void**)d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) PLH_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return PLH_OK;
}

plh_status build(const RawTree& t, int device, plh_vocab** out) {
  const int n = (int)t.parent.size();
  if (n < 2) { set_error("plh_vocab: empty vocabulary"); return PLH_ERR_INVALID; }
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  PLH_HIP(hipSetDevice(device));
  plh_vocab* v = new plh_vocab();
  v->device = device;
  v->k = t.k; v->L = t.L; v->scoring = t.scoring; v->weighting = t.weighting; v->nNodes = n;
  // children in the reference's order (node ids ascending == file order), word ids count the leaves in id order
  std::vector<std::vector<int32_t>> kids(n);
  for (int i = 1; i < n; i++) kids[t.parent[i]].push_back(i);
  std::vector<int32_t> refWord(n, -1);
  int nWords = 0;
  for (int i = 1; i < n; i++)
    if (t.leaf[i]) refWord[i] = nWords++;
  v->nWords = nWords;
  bool contiguous = true;
  for (int i = 0; i < n && contiguous; i++)
    for (size_t c = 1; c < kids[i].size(); c++)
      if (kids[i][c] != kids[i][c - 1] + 1) { contiguous = false; break; }
  // flat order: identity when the file is already contiguous, else breadth first
  std::vector<int32_t> order(n);
  if (contiguous) {
    for (int i = 0; i < n; i++) order[i] = i;
  } else {
    int w = 0;
    order[w++] = 0;
    for (int r = 0; r < w; r++)
      for (int32_t c : kids[order[r]]) order[w++] = c;
    if (w != n) { delete v; set_error("plh_vocab: %d nodes are not reachable from the root", n - w); return PLH_ERR_INVALID; }
  }
  std::vector<int32_t> flatOf(n);
  for (int i = 0; i < n; i++) flatOf[order[i]] = i;
  v->hDesc.resize((size_t)n * 32);
  v->hChildStart.assign(n, 0); v->hChildCount.assign(n, 0); v->hWordId.assign(n, -1); v->hNodeId.resize(n);
  v->hWeight.resize(n); v->hWeightF.resize(n);
  v->hWordWeight.assign(std::max(nWords, 1), 0.0);
  for (int i = 0; i < n; i++) {
    const int r = order[i];
    memcpy(&v->hDesc[(size_t)i * 32], &t.desc[(size_t)r * 32], 32);
    v->hChildCount[i] = (int32_t)kids[r].size();
    v->hChildStart[i] = kids[r].empty() ? 0 : flatOf[kids[r][0]];
    // a leaf flag on a node with children cannot be descended into consistently: the reference would stop on isLeaf()
    // == children.empty(), so the children decide
    v->hWordId[i] = kids[r].empty() ? refWord[r] : -1;
    v->hNodeId[i] = r;
    v->hWeight[i] = t.weight[r];
    const float wf = (float)t.weight[r];
    v->hWeightF[i] = (t.weight[r] > 0 && !(wf > 0.f)) ? FLT_MIN : wf;   // the kernels only test `w > 0`
    if (kids[r].empty() && refWord[r] >= 0) v->hWordWeight[refWord[r]] = t.weight[r];
    if (kids[r].empty() && refWord[r] < 0) {   // childless node the file does not call a leaf: descent ends there without a word
      v->hWeightF[i] = 0.f;
    }
  }
  v->identity = contiguous;
  plh_status st = upload(v->hDesc, &v->dDesc);
  if (st == PLH_OK) st = upload(v->hChildStart, &v->dChildStart);
  if (st == PLH_OK) st = upload(v->hChildCount, &v->dChildCount);
  if (st == PLH_OK) st = upload(v->hWordId, &v->dWordId);
  if (st == PLH_OK) st = upload(v->hWeightF, &v->dWeightF);
  if (st == PLH_OK) st = upload(v->hWordWeight, &v->dWordWeight);
  if (st == PLH_OK && !contiguous) st = upload(v->hNodeId, &v->dNodeId);
  if (st != PLH_OK) { plh_vocab_destroy(v); return st; }
  *out = v;
  return PLH_OK;
}

}  // namespace

extern "C" {

plh_status plh_vocab_load_text(const char* path, int device, plh_vocab** out) {
  if (!path || !out) return PLH_ERR_INVALID;
  RawTree t;
  plh_status st = parse_text(path, t);
  return st != PLH_OK ? st : build(t, device, out);
}

plh_status plh_vocab_load_binary(const char* path, int device, plh_vocab** out) {
  if (!path || !out) return PLH_ERR_INVALID;
  RawTree t;
  plh_status st = parse_binary(path, t);
  return st != PLH_OK ? st : build(t, device, out);
}

plh_status plh_vocab_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent, const uint8_t* is_leaf,
                            const uint8_t* node_desc, const double* weight, int device, plh_vocab** out) {
  if (!parent || !is_leaf || !node_desc || !weight || !out || n_nodes < 2 || scoring < 0 || scoring > 5 || weighting < 0 ||
      weighting > 3 || L < 1) {
    set_error("plh_vocab_create: invalid argument");
    return PLH_ERR_INVALID;
  }
  RawTree t;
  t.k = k; t.L = L; t.scoring = scoring; t.weighting = weighting;
  push_root(t);
  for (int i = 1; i < n_nodes; i++) {
    if (parent[i] < 0 || parent[i] >= i) { set_error("plh_vocab_create: node %d names parent %d", i, parent[i]); return PLH_ERR_INVALID; }
    t.parent.push_back(parent[i]);
    t.leaf.push_back(is_leaf[i] != 0);
    t.desc.insert(t.desc.end(), node_desc + (size_t)i * 32, node_desc + (size_t)i * 32 + 32);
    t.weight.push_back(weight[i]);
  }
  return build(t, device, out);
}

plh_status plh_vocab_save_binary(const plh_vocab* v, const char* path) {
  if (!v || !path) return PLH_ERR_INVALID;
  FILE* f = std::fopen(path, "wb");
  if (!f) { set_error("plh_vocab_save_binary: cannot write %s", path); return PLH_ERR_INVALID; }
  const int n = v->nNodes;
  // back to the reference's numbering
  std::vector<int32_t> flatOf(n);
  for (int i = 0; i < n; i++) flatOf[v->hNodeId[i]] = i;
  std::vector<int32_t> parentRef(n, 0);
  for (int i = 0; i < n; i++)
    for (int c = 0; c < v->hChildCount[i]; c++) parentRef[v->hNodeId[v->hChildStart[i] + c]] = v->hNodeId[i];
  const uint32_t nb = (uint32_t)n, sizeNode = 41;
  const int32_t h[4] = {v->k, v->L, v->scoring, v->weighting};
  std::fwrite(&nb, 4, 1, f);
  std::fwrite(&sizeNode, 4, 1, f);
  std::fwrite(h, 4, 4, f);
  for (int r = 1; r < n; r++) {
    const int i = flatOf[r];
    const uint32_t pid = (uint32_t)parentRef[r];
    const float w = (float)v->hWeight[i];
    const uint8_t leaf = v->hChildCount[i] == 0;
    std::fwrite(&pid, 4, 1, f);
    std::fwrite(&v->hDesc[(size_t)i * 32], 1, 32, f);
    std::fwrite(&w, 4, 1, f);
    std::fwrite(&leaf, 1, 1, f);
  }
  const bool wrote = std::ferror(f) == 0;   // a full disk shows up in the stream's error flag, not necessarily in fclose()
  const bool ok = (std::fclose(f) == 0) && wrote;
  if (!ok) { set_error("plh_vocab_save_binary: write to %s failed", path); return PLH_ERR_INVALID; }
  return PLH_OK;
}

plh_status plh_vocab_destroy(plh_vocab* v) {
  if (!v) return PLH_OK;
  (void)hipSetDevice(v->device);
  (void)hipFree(v->dDesc); (void)hipFree(v->dChildStart); (void)hipFree(v->dChildCount); (void)hipFree(v->dWordId);
  (void)hipFree(v->dWeightF); (void)hipFree(v->dWordWeight); (void)hipFree(v->dNodeId);
  delete v;
  return PLH_OK;
}

plh_status plh_vocab_get_info(const plh_vocab* v, plh_vocab_info* info) {
  if (!v || !info) return PLH_ERR_INVALID;
  info->k = v->k; info->L = v->L; info->scoring = v->scoring; info->weighting = v->weighting;
  info->n_nodes = v->nNodes; info->n_words = v->nWords; info->identity_ids = v->identity ? 1 : 0;
  return PLH_OK;
}

plh_status plh_vocab_device_arrays(const plh_vocab* v, const uint8_t** d_node_desc, const int32_t** d_child_start,
                                   const int32_t** d_child_count, const int32_t** d_word_id, const float** d_weight,
                                   const int32_t** d_node_id) {
  if (!v) return PLH_ERR_INVALID;
  if (d_node_desc) *d_node_desc = v->dDesc;
  if (d_child_start) *d_child_start = v->dChildStart;
  if (d_child_count) *d_child_count = v->dChildCount;
  if (d_word_id) *d_word_id = v->dWordId;
  if (d_weight) *d_weight = v->dWeightF;
  if (d_node_id) *d_node_id = v->dNodeId;   // NULL when flat index == reference NodeId
  return PLH_OK;
}

plh_status plh_vocab_read(const plh_vocab* v, uint8_t* node_desc, int32_t* child_start, int32_t* child_count, int32_t* word_id,
                          double* weight, int32_t* node_id) {
  if (!v) return PLH_ERR_INVALID;
  const size_t n = (size_t)v->nNodes;
  if (node_desc) memcpy(node_desc, v->hDesc.data(), n * 32);
  if (child_start) memcpy(child_start, v->hChildStart.data(), n * 4);
  if (child_count) memcpy(child_count, v->hChildCount.data(), n * 4);
  if (word_id) memcpy(word_id, v->hWordId.data(), n * 4);
  if (weight) memcpy(weight, v->hWeight.data(), n * 8);
  if (node_id) memcpy(node_id, v->hNodeId.data(), n * 4);
  return PLH_OK;
}

}  // extern "C"
