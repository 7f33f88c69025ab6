// lm.hip -- host side of the external scorer: ARPA reader -> device hash table (lm.h).
// Replaces `Scorer(alpha, beta, model_path, vocabulary)` of paddlespeech_ctcdecoders (PPASR call site
// decoders/swig_wrapper.py:18-33, decoders/beam_search_decoder.py:28-29) for CHARACTER-BASED models, i.e. models whose
// words are all single UTF-8 characters (scorer.cpp `load_lm`: is_character_based_), which is what PPASR's Mandarin
// models are.  Word-based models need the OpenFST dictionary constraint of the trie and are refused.
// The model file is read in the ARPA text format; KenLM's binary formats (.klm) are not parsed.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "capi_internal.h"
#include "lm.h"

struct ppasr_lm_s {
  LmDev dev{};
  int order = 0;
  int n_words = 0;
  bool character_based = true;
  size_t n_grams = 0;
  std::vector<void*> allocs;
  ~ppasr_lm_s() {
    for (void* p : allocs) (void)hipFree(p);
  }
};

namespace {

int utf8_len(const std::string& s) {
  int n = 0;
  for (unsigned char c : s) n += (c & 0xc0) != 0x80;
  return n;
}

struct Gram {
  std::vector<int32_t> w;
  float prob, backoff;
};

}  // namespace

extern "C" {

ppasr_status ppasr_lm_create_arpa(const char* arpa_path, const char* const* vocab_utf8, int V, ppasr_lm_handle* out) {
  if (!arpa_path || !vocab_utf8 || V <= 0 || !out) return fail(PPASR_EINVAL, "lm: null argument");
  std::ifstream in(arpa_path, std::ios::binary);
  if (!in) return fail(PPASR_EINVAL, std::string("lm: cannot open ") + arpa_path);
  {
    char magic[8] = {0};
    in.read(magic, 8);
    if (std::memcmp(magic, "mmap lm ", 8) == 0)
      return fail(PPASR_EUNSUPPORTED, "lm: KenLM binary (.klm) files are not parsed; provide the ARPA text model");
    in.clear();
    in.seekg(0);
  }
  std::unordered_map<std::string, int32_t> words;
  words["<unk>"] = 0;  // KenLM: index 0 is <unk>, what Vocabulary::Index returns for unknown strings
  std::vector<Gram> grams;
  std::vector<size_t> counts;
  std::string line;
  int section = 0;  // 0 = before \data\, -1 = in \data\, n = in \n-grams:
  auto word_id = [&](const std::string& s) {
    auto it = words.find(s);
    if (it != words.end()) return it->second;
    int32_t id = (int32_t)words.size();
    words.emplace(s, id);
    return id;
  };
  while (std::getline(in, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
    if (line.empty()) continue;
    if (line[0] == '\\') {
      if (line == "\\data\\") section = -1;
      else if (line == "\\end\\") break;
      else {
        int n = 0;
        if (std::sscanf(line.c_str(), "\\%d-grams:", &n) == 1 && n >= 1) section = n;
        else return fail(PPASR_EINVAL, "lm: unrecognised ARPA section header: " + line);
      }
      continue;
    }
    if (section == -1) {
      int n = 0;
      unsigned long long c = 0;
      if (std::sscanf(line.c_str(), "ngram %d=%llu", &n, &c) == 2) {
        if ((int)counts.size() < n) counts.resize(n, 0);
        counts[n - 1] = (size_t)c;
      }
      continue;
    }
    if (section < 1) continue;
    if (section > kLmMaxOrder) return fail(PPASR_EUNSUPPORTED, "lm: model order above 6");
    // "<log10 prob>\t<w1> ... <wn>[\t<log10 backoff>]" ; fields may be separated by tabs or spaces
    std::istringstream ss(line);
    std::vector<std::string> tok;
    std::string t;
    while (ss >> t) tok.push_back(t);
    if ((int)tok.size() != section + 1 && (int)tok.size() != section + 2)
      return fail(PPASR_EINVAL, "lm: malformed n-gram line: " + line);
    Gram g;
    g.prob = std::strtof(tok[0].c_str(), nullptr);
    g.backoff = (int)tok.size() == section + 2 ? std::strtof(tok[section + 1].c_str(), nullptr) : 0.f;
    g.w.resize(section);
    for (int i = 0; i < section; ++i) g.w[i] = word_id(tok[1 + i]);
    grams.push_back(std::move(g));
  }
  if (grams.empty()) return fail(PPASR_EINVAL, "lm: no n-grams found (not an ARPA file?)");
  auto lm = std::make_unique<ppasr_lm_s>();
  for (const Gram& g : grams) lm->order = std::max(lm->order, (int)g.w.size());
  lm->n_words = (int)words.size();
  lm->n_grams = grams.size();
  if (!words.count("<s>") || !words.count("</s>")) return fail(PPASR_EINVAL, "lm: the model has no <s> / </s>");
  for (const auto& kv : words)
    if (kv.first != "<unk>" && kv.first != "<s>" && kv.first != "</s>" && utf8_len(kv.first) > 1) lm->character_based = false;
  if (!lm->character_based)
    return fail(PPASR_EUNSUPPORTED, "lm: word-based language model (needs the dictionary-constrained trie); only "
                                    "character-based models are built");
  // ---- hash table ----
  size_t cap = 16;
  while (cap < 2 * grams.size()) cap <<= 1;
  std::vector<uint64_t> keys(cap, 0);
  std::vector<float> prob(cap, 0.f), backoff(cap, 0.f);
  for (const Gram& g : grams) {
    const uint64_t key = lm_key(g.w.data(), (int)g.w.size());
    size_t slot = (size_t)(key >> 17) & (cap - 1);
    while (keys[slot] != 0) {
      if (keys[slot] == key) return fail(PPASR_EINVAL, "lm: duplicate n-gram (or a 64-bit hash collision) in the model");
      slot = (slot + 1) & (cap - 1);
    }
    keys[slot] = key;
    prob[slot] = g.prob;
    backoff[slot] = g.backoff;
  }
  std::vector<int32_t> tok2lm(V, 0);
  for (int v = 0; v < V; ++v) {
    if (!vocab_utf8[v]) continue;
    const std::string s(vocab_utf8[v]);
    // a literal space is SPACE_ID_: Scorer::make_ngram stops on it with an empty word, i.e. OOV (scorer.cpp)
    if (s == " ") continue;
    auto it = words.find(s);
    if (it != words.end()) tok2lm[v] = it->second;
  }
  auto up = [&](const void* src, size_t bytes, const void** dst) -> ppasr_status {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    lm->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
    *dst = d;
    return PPASR_OK;
  };
  const void* p = nullptr;
  ppasr_status s;
  if ((s = up(keys.data(), keys.size() * 8, &p)) != PPASR_OK) return s;
  lm->dev.keys = static_cast<const uint64_t*>(p);
  if ((s = up(prob.data(), prob.size() * 4, &p)) != PPASR_OK) return s;
  lm->dev.prob = static_cast<const float*>(p);
  if ((s = up(backoff.data(), backoff.size() * 4, &p)) != PPASR_OK) return s;
  lm->dev.backoff = static_cast<const float*>(p);
  if ((s = up(tok2lm.data(), tok2lm.size() * 4, &p)) != PPASR_OK) return s;
  lm->dev.tok2lm = static_cast<const int32_t*>(p);
  lm->dev.order = lm->order;
  lm->dev.bos = words["<s>"];
  lm->dev.eos = words["</s>"];
  lm->dev.mask = (uint32_t)(cap - 1);
  *out = lm.release();
  return PPASR_OK;
}

ppasr_status ppasr_lm_destroy(ppasr_lm_handle lm) {
  delete lm;
  return PPASR_OK;
}

int ppasr_lm_order(ppasr_lm_handle lm) { return lm ? lm->order : 0; }
int ppasr_lm_is_character_based(ppasr_lm_handle lm) { return lm ? (lm->character_based ? 1 : 0) : 0; }
long long ppasr_lm_ngram_count(ppasr_lm_handle lm) { return lm ? (long long)lm->n_grams : 0; }

// internal (capi.hip): the device view handed to the beam-search kernel
const ppasr::LmDev* ppasr_lm_device_view(ppasr_lm_handle lm) { return lm ? &lm->dev : nullptr; }

}  // extern "C"
