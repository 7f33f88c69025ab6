// lm.hip -- host side of the external scorer: ARPA reader -> device hash table (lm.h).
// Replaces `Scorer(alpha, beta, model_path, vocabulary)` of paddlespeech_ctcdecoders (PPASR call site
// decoders/swig_wrapper.py:18-33, decoders/beam_search_decoder.py:28-29) for CHARACTER-BASED models, i.e. models whose
// words are all single UTF-8 characters (scorer.cpp `load_lm`: is_character_based_), which is what PPASR's Mandarin
// models are, and for WORD-BASED models (any vocabulary word longer than one character; the English configs): those are
// consulted when a space is appended and constrain the prefixes to spellings of their vocabulary through a dictionary
// (upstream: an OpenFST acceptor built by Scorer::fill_dictionary; here the same language as a character trie, lm.h).
// This file reads the ARPA text format; KenLM's binary formats (.klm) are read by klm.hip.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "lm_host.h"

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

ppasr_status arpa_load(const char* arpa_path, const char* const* vocab_utf8, int V, bool host_only, ppasr_lm_handle* out) {
  if (!arpa_path || !vocab_utf8 || V <= 0 || !out) return fail(PPASR_EINVAL, "lm: null argument");
  std::ifstream in(arpa_path, std::ios::binary);
  if (!in) return fail(PPASR_EINVAL, std::string("lm: cannot open ") + arpa_path);
  {
    char magic[8] = {0};
    in.read(magic, 8);
    if (std::memcmp(magic, "mmap lm ", 8) == 0)
      return fail(PPASR_EINVAL, "lm: this is a KenLM binary, not an ARPA file (use ppasr_lm_create / ppasr_lm_create_klm)");
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
  lm->format = "arpa";
  for (const Gram& g : grams) lm->order = std::max(lm->order, (int)g.w.size());
  std::string err = lm_bind_vocabulary(*lm, words, vocab_utf8, V);
  if (!err.empty()) return fail(PPASR_EINVAL, err);
  std::vector<LmEntry> entries;
  entries.reserve(grams.size());
  for (const Gram& g : grams) entries.push_back(LmEntry{lm_key(g.w.data(), (int)g.w.size()), g.prob, g.backoff});
  err = lm_build_table(*lm, entries);
  if (!err.empty()) return fail(PPASR_EINVAL, err);
  if (host_only) {
    *out = lm.release();
    return PPASR_OK;
  }
  ppasr_status s = lm_upload(*lm);
  if (s != PPASR_OK) return s;
  *out = lm.release();
  return PPASR_OK;
}

}  // namespace

namespace ppasr {

std::string lm_bind_vocabulary(ppasr_lm_s& lm, const std::unordered_map<std::string, int32_t>& words,
                               const char* const* vocab_utf8, int V) {
  lm.n_words = (int)words.size();
  auto bos = words.find("<s>"), eos = words.find("</s>");
  if (bos == words.end() || eos == words.end()) return "lm: the model has no <s> / </s>";
  lm.bos = bos->second;
  lm.eos = eos->second;
  lm.character_based = true;
  for (const auto& kv : words)
    if (kv.first != "<unk>" && kv.first != "<s>" && kv.first != "</s>" && utf8_len(kv.first) > 1) lm.character_based = false;
  std::unordered_map<std::string, int> tok_of;  // acoustic character -> token id
  for (int v = 0; v < V; ++v) {
    if (!vocab_utf8[v]) continue;
    const std::string s(vocab_utf8[v]);
    // the space token: " " (upstream DeepSpeech / early PaddleSpeech) or "<space>" (PaddleSpeech kSPACE; PPASR's
    // vocabularies spell it that way, data_utils/featurizer/text_featurizer.py:23)
    if (s == " " || s == "<space>") {
      lm.space_id = v;
      continue;
    }
    tok_of.emplace(s, v);
  }
  lm.tok2lm.assign(V, 0);
  if (lm.character_based) {
    for (int v = 0; v < V; ++v) {
      if (!vocab_utf8[v] || v == lm.space_id) continue;  // a space: Scorer::make_ngram stops on it with an empty word = OOV
      auto it = words.find(std::string(vocab_utf8[v]));
      if (it != words.end()) lm.tok2lm[v] = it->second;
    }
    return "";
  }
  // ---- word-based: Scorer::fill_dictionary(add_space = true) -- every vocabulary word that can be spelt with the
  // acoustic characters, followed by the space, goes into the dictionary ----
  if (lm.space_id < 0) return "lm: word-based language model, but the acoustic vocabulary has no space token (\" \" or \"<space>\")";
  struct Node {
    std::vector<std::pair<int, int>> arcs;  // (char, target)
    int word = 0;
  };
  std::vector<Node> nodes(1);
  auto child = [&](int s, int c) {
    for (auto& a : nodes[s].arcs)
      if (a.first == c) return a.second;
    const int t = (int)nodes.size();
    nodes[s].arcs.emplace_back(c, t);
    nodes.emplace_back();
    return t;
  };
  for (const auto& kv : words) {
    const std::string& w = kv.first;
    if (w == "<unk>" || w == "<s>" || w == "</s>" || w.empty()) continue;
    std::vector<int> spelt;
    bool ok = true;
    for (size_t i = 0; i < w.size() && ok;) {
      size_t j = i + 1;
      while (j < w.size() && ((unsigned char)w[j] & 0xc0) == 0x80) ++j;
      auto it = tok_of.find(w.substr(i, j - i));
      if (it == tok_of.end()) ok = false;
      else spelt.push_back(it->second);
      i = j;
    }
    if (!ok) continue;  // add_word_to_dictionary: a word with a character outside the acoustic vocabulary is skipped
    int s = 0;
    for (int c : spelt) s = child(s, c);
    s = child(s, lm.space_id);
    nodes[s].word = kv.second;
    ++lm.dict_words;
  }
  if (lm.dict_words == 0) return "lm: word-based language model, but none of its words can be spelt with the acoustic vocabulary";
  lm.dict_first.assign(nodes.size() + 1, 0);
  lm.dict_word.assign(nodes.size(), 0);
  for (size_t n = 0; n < nodes.size(); ++n) {
    std::sort(nodes[n].arcs.begin(), nodes[n].arcs.end());
    lm.dict_first[n + 1] = lm.dict_first[n] + (int32_t)nodes[n].arcs.size();
    lm.dict_word[n] = nodes[n].word;
    for (auto& a : nodes[n].arcs) {
      lm.dict_arc_char.push_back(a.first);
      lm.dict_arc_next.push_back(a.second);
    }
  }
  return "";
}

std::string lm_build_table(ppasr_lm_s& lm, const std::vector<LmEntry>& entries) {
  size_t cap = 16;
  while (cap < 3 * entries.size()) cap <<= 1;  // load factor <= 1/3: lm_probe_many reads two adjacent slots per key
  lm.slots.assign(cap + 1, LmSlot{0, 0.f, 0.f});
  for (const LmEntry& e : entries) {
    size_t slot = (size_t)lm_slot_of(e.key, (uint32_t)(cap - 1));
    while (lm.slots[slot].key != 0) {
      if (lm.slots[slot].key == e.key) return "lm: duplicate n-gram (or a 64-bit hash collision) in the model";
      slot = (slot + 1) & (cap - 1);
    }
    lm.slots[slot] = LmSlot{e.key, e.prob, e.backoff};
  }
  lm.slots[cap] = lm.slots[0];  // (slot + 1 of the last slot)
  lm.n_grams = entries.size();
  // unigram table: the level every look-up ends at, indexed by the LM word (the candidate side of the factorised form)
  lm.uni_prob.assign((size_t)(lm.n_words > 0 ? lm.n_words : 1), std::nanf(""));
  for (int32_t w = 0; w < lm.n_words; ++w) {
    float p, b;
    if (lm_find_key(lm.slots.data(), (uint32_t)(cap - 1), lm_key_any(lm.kenlm_keys ? 1 : 0, &w, 1), p, b)) lm.uni_prob[w] = p;
  }
  return "";
}

ppasr_status lm_upload(ppasr_lm_s& lm) {
  auto up = [&](const void* src, size_t bytes, const void** dst) -> ppasr_status {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    lm.allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
    *dst = d;
    return PPASR_OK;
  };
  const void* p = nullptr;
  ppasr_status s;
  if ((s = up(lm.slots.data(), lm.slots.size() * sizeof(LmSlot), &p)) != PPASR_OK) return s;
  lm.dev.slots = static_cast<const LmSlot*>(p);
  if ((s = up(lm.uni_prob.data(), lm.uni_prob.size() * 4, &p)) != PPASR_OK) return s;
  lm.dev.uni_prob = static_cast<const float*>(p);
  lm.dev.n_words = lm.n_words;
  if ((s = up(lm.tok2lm.data(), lm.tok2lm.size() * 4, &p)) != PPASR_OK) return s;
  lm.dev.tok2lm = static_cast<const int32_t*>(p);
  lm.dev.order = lm.order;
  lm.dev.bos = lm.bos;
  lm.dev.eos = lm.eos;
  lm.dev.mask = (uint32_t)(lm.slots.size() - 2);
  lm.dev.kenlm_keys = lm.kenlm_keys ? 1 : 0;
  lm.dev.word_based = lm.character_based ? 0 : 1;
  lm.dev.space_id = lm.space_id;
  if (!lm.character_based) {
    if ((s = up(lm.dict_first.data(), lm.dict_first.size() * 4, &p)) != PPASR_OK) return s;
    lm.dev.dict_first = static_cast<const int32_t*>(p);
    if ((s = up(lm.dict_arc_char.data(), lm.dict_arc_char.size() * 4, &p)) != PPASR_OK) return s;
    lm.dev.dict_arc_char = static_cast<const int32_t*>(p);
    if ((s = up(lm.dict_arc_next.data(), lm.dict_arc_next.size() * 4, &p)) != PPASR_OK) return s;
    lm.dev.dict_arc_next = static_cast<const int32_t*>(p);
    if ((s = up(lm.dict_word.data(), lm.dict_word.size() * 4, &p)) != PPASR_OK) return s;
    lm.dev.dict_word = static_cast<const int32_t*>(p);
  }
  return PPASR_OK;
}

}  // namespace ppasr

// klm.hip
ppasr_status klm_load(const char* path, const char* const* vocab_utf8, int V, bool host_only, ppasr_lm_handle* out);

static bool is_klm(const char* path) {
  std::ifstream in(path, std::ios::binary);
  char magic[8] = {0};
  in.read(magic, 8);
  return in.gcount() == 8 && std::memcmp(magic, "mmap lm ", 8) == 0;
}

extern "C" {

ppasr_status ppasr_lm_create_arpa(const char* arpa_path, const char* const* vocab_utf8, int V, ppasr_lm_handle* out) {
  return arpa_load(arpa_path, vocab_utf8, V, false, out);
}

ppasr_status ppasr_lm_create_klm(const char* klm_path, const char* const* vocab_utf8, int V, ppasr_lm_handle* out) {
  return klm_load(klm_path, vocab_utf8, V, false, out);
}

// Scorer(alpha, beta, model_path, vocabulary) accepts both formats like KenLM does (lm::ngram::LoadVirtual sniffs the magic)
ppasr_status ppasr_lm_create(const char* model_path, const char* const* vocab_utf8, int V, ppasr_lm_handle* out) {
  if (!model_path) return fail(PPASR_EINVAL, "lm: null argument");
  return is_klm(model_path) ? klm_load(model_path, vocab_utf8, V, false, out) : arpa_load(model_path, vocab_utf8, V, false, out);
}

// Verification hook for the model-file readers (tests/test_klm_cpu.py): parses the file into the host table only (no
// device is touched) so that `ppasr_lm_debug_host_score` can be compared between formats.  The decoder never uses it.
ppasr_status ppasr_lm_debug_load_host(const char* model_path, const char* const* vocab_utf8, int V, ppasr_lm_handle* out) {
  if (!model_path) return fail(PPASR_EINVAL, "lm: null argument");
  return is_klm(model_path) ? klm_load(model_path, vocab_utf8, V, true, out) : arpa_load(model_path, vocab_utf8, V, true, out);
}

// Scorer::get_log_cond_prob of an `order`-word window of LM word indices (oldest first), evaluated on the HOST copy of
// the table with the arithmetic of lm_log_cond_prob (lm.h).
double ppasr_lm_debug_host_score(ppasr_lm_handle lm, const int32_t* win) {
  if (!lm || !win) return 0.0;
  const int order = lm->order;
  const uint32_t mask = (uint32_t)(lm->slots.size() - 2);
  auto find = [&](const int32_t* w, int n, float& p, float& b) {
    return lm_find_key(lm->slots.data(), mask, lm_key_any(lm->kenlm_keys ? 1 : 0, w, n), p, b);
  };
  for (int i = 0; i < order; ++i)
    if (win[i] == 0) return kLmOovScore;
  float acc = 0.f;
  for (int n = order; n >= 1; --n) {
    float p, b;
    if (find(win + order - n, n, p, b)) return (double)(acc + p) / (double)kLmLog10E;
    if (n > 1 && find(win + order - n, n - 1, p, b)) acc += b;
  }
  return kLmOovScore;
}

int ppasr_lm_word_index(ppasr_lm_handle lm, int token) {
  return (lm && token >= 0 && token < (int)lm->tok2lm.size()) ? lm->tok2lm[token] : -1;
}
int ppasr_lm_bos(ppasr_lm_handle lm) { return lm ? lm->bos : -1; }
int ppasr_lm_eos(ppasr_lm_handle lm) { return lm ? lm->eos : -1; }
const char* ppasr_lm_format(ppasr_lm_handle lm) { return lm ? lm->format.c_str() : ""; }

ppasr_status ppasr_lm_destroy(ppasr_lm_handle lm) {
  delete lm;
  return PPASR_OK;
}

int ppasr_lm_order(ppasr_lm_handle lm) { return lm ? lm->order : 0; }
int ppasr_lm_is_character_based(ppasr_lm_handle lm) { return lm ? (lm->character_based ? 1 : 0) : 0; }
long long ppasr_lm_dict_size(ppasr_lm_handle lm) { return lm ? (long long)lm->dict_words : 0; }
int ppasr_lm_space_id(ppasr_lm_handle lm) { return lm ? lm->space_id : -1; }
long long ppasr_lm_ngram_count(ppasr_lm_handle lm) { return lm ? (long long)lm->n_grams : 0; }

}  // extern "C"

// internal (capi.hip): the device view handed to the beam-search kernel
namespace ppasr {
const LmDev* lm_device_view(ppasr_lm_handle lm) { return lm ? &lm->dev : nullptr; }
}  // namespace ppasr
