// lm_host.h -- host side of the language-model object shared by the model-file readers (lm.hip: ARPA text,
// klm.hip: KenLM binaries): the flattened n-gram table is built on the host, kept there for the verification hook
// ppasr_lm_debug_host_score, and uploaded to the device for the beam-search kernel.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "capi_internal.h"
#include "lm.h"

struct ppasr_lm_s {
  ppasr::LmDev dev{};
  int order = 0;
  int n_words = 0;
  bool character_based = true;
  bool kenlm_keys = false;
  size_t n_grams = 0;
  std::string format;  // "arpa", "klm-probing", "klm-trie", ...
  // host copy of the table (what `dev` points at on the device)
  std::vector<ppasr::LmSlot> slots;  // power-of-two slots + one wrap-around copy of slot 0
  std::vector<float> uni_prob;       // [n_words] log10 P of every unigram (NaN: absent)
  std::vector<int32_t> tok2lm;
  int bos = 0, eos = 0;
  // word-based models: the dictionary (lm.h) and the acoustic token of the space
  int space_id = -1;
  size_t dict_words = 0;  // Scorer::get_dict_size(): vocabulary words that could be spelt in acoustic characters
  std::vector<int32_t> dict_first, dict_arc_char, dict_arc_next, dict_word;
  std::vector<void*> allocs;
  ~ppasr_lm_s() {
    for (void* p : allocs) (void)hipFree(p);
  }
};

namespace ppasr {

struct LmEntry {
  uint64_t key;  // slot key: lm_key(word ids, n) or lm_key_from_kenlm(chain, n)
  float prob, backoff;
};

// words: LM word string -> index (0 = <unk>); fills character_based / bos / eos / tok2lm; the error string is empty on success
std::string lm_bind_vocabulary(ppasr_lm_s& lm, const std::unordered_map<std::string, int32_t>& words,
                               const char* const* vocab_utf8, int V);
// open-addressing table from the entries (host); returns an error string or ""
std::string lm_build_table(ppasr_lm_s& lm, const std::vector<LmEntry>& entries);
// copies the host table to the device and fills lm.dev
ppasr_status lm_upload(ppasr_lm_s& lm);

}  // namespace ppasr
