// lm.h -- back-off n-gram language model on the device, for the external scorer of the CTC prefix beam search
// (PaddleSpeech third_party/ctc_decoders scorer.cpp: `Scorer::get_log_cond_prob` walks KenLM's `BaseScore` over the
// n-gram window; PPASR builds it in decoders/beam_search_decoder.py:28-29 via decoders/swig_wrapper.py:18-33).
//
// All n-grams of all orders live in ONE open-addressing hash table keyed by a 64-bit hash of (order, word ids) --
// the layout KenLM's "probing" model uses, flattened: slot = (key, log10 prob, log10 back-off).  A conditional
// probability costs at most 2*order - 1 probes; the table is read-only and L2-resident.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ppasr {

constexpr int kLmMaxOrder = 6;
constexpr double kLmOovScore = -1000.0;              // OOV_SCORE (scorer.h)
constexpr float kLmLog10E = 0.4342944819f;           // NUM_FLT_LOGE (decoder_utils.h)

struct LmDev {
  int order;                 // 0 = no language model
  int bos, eos;              // word index of <s>, </s>
  uint32_t mask;             // table size - 1 (power of two)
  const uint64_t* keys;      // 0 = empty slot
  const float* prob;         // log10 P
  const float* backoff;      // log10 back-off weight (0 when absent)
  const int32_t* tok2lm;     // [V] acoustic-vocabulary id -> LM word index, 0 = OOV (<unk>)
  int kenlm_keys;            // 1: n-grams are keyed by KenLM's own word-hash chain (tables taken over from a "probing"
                             //    .klm binary, whose entries carry only that hash); 0: by lm_key over the word ids
  // ---- word-based models (scorer.cpp: is_character_based_ == false): the LM is consulted when a SPACE is appended, and
  // the prefix trie is constrained to the model's vocabulary by a dictionary (upstream: an OpenFST acceptor of every
  // vocabulary word spelt in acoustic characters + the space; here the same language as a plain character trie in CSR
  // form: node s has the arcs [dict_first[s], dict_first[s + 1]), sorted by character; node 0 = start) ----
  int word_based;            // 0: character-based
  int space_id;              // acoustic token id of the space
  const int32_t* dict_first;     // [n_nodes + 1]
  const int32_t* dict_arc_char;  // [n_arcs] acoustic token id
  const int32_t* dict_arc_next;  // [n_arcs] target node
  const int32_t* dict_word;      // [n_nodes] LM word index of the word that ENDS at this node (nodes entered by a space
                                 // arc: the dictionary's final states), 0 elsewhere
};

// arc of dictionary node `s` labelled `c`: target node, or -1 (binary search over the node's sorted arcs)
__host__ __device__ inline int lm_dict_arc(const LmDev& lm, int s, int c) {
  int lo = lm.dict_first[s], hi = lm.dict_first[s + 1];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int a = lm.dict_arc_char[mid];
    if (a == c) return lm.dict_arc_next[mid];
    if (a < c) lo = mid + 1;
    else hi = mid;
  }
  return -1;
}
__host__ __device__ inline bool lm_dict_final(const LmDev& lm, int s) { return lm.dict_word[s] != 0; }

__host__ __device__ inline uint64_t lm_mix(uint64_t h, uint64_t v) {
  h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  return h;
}
// hash of the n-gram w[0..n-1] (never 0)
__host__ __device__ inline uint64_t lm_key(const int32_t* w, int n) {
  uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)n;
  for (int i = 0; i < n; ++i) h = lm_mix(h, (uint64_t)(uint32_t)w[i]);
  return h | 1ull;
}

// KenLM's n-gram hash (lm/search_hashed.hh detail::CombineWordHash): starts from the LAST word's index and folds the
// preceding words in one by one, newest first -- the key under which a probing-model entry for w[0..n-1] is stored.
__host__ __device__ inline uint64_t kenlm_combine(uint64_t current, uint32_t next) {
  return (current * 8978948897894561157ULL) ^ ((uint64_t)(1 + next) * 17894857484156487943ULL);
}
__host__ __device__ inline uint64_t kenlm_chain(const int32_t* w, int n) {
  uint64_t h = (uint64_t)(uint32_t)w[n - 1];
  for (int i = n - 2; i >= 0; --i) h = kenlm_combine(h, (uint32_t)w[i]);
  return h;
}
// slot key of a KenLM-keyed entry of order n (the order is mixed in: KenLM keeps one table per order, we keep one)
__host__ __device__ inline uint64_t lm_key_from_kenlm(uint64_t chain, int n) {
  return lm_mix(0x2545f4914f6cdd1dull ^ (uint64_t)n, chain) | 1ull;
}
__host__ __device__ inline uint64_t lm_key_any(int kenlm_keys, const int32_t* w, int n) {
  return kenlm_keys ? lm_key_from_kenlm(kenlm_chain(w, n), n) : lm_key(w, n);
}

__device__ inline bool lm_find(const LmDev& lm, const int32_t* w, int n, float& prob, float& backoff) {
  const uint64_t key = lm_key_any(lm.kenlm_keys, w, n);
  uint32_t slot = (uint32_t)(key >> 17) & lm.mask;
  for (;;) {
    const uint64_t k = lm.keys[slot];
    if (k == key) {
      prob = lm.prob[slot];
      backoff = lm.backoff[slot];
      return true;
    }
    if (k == 0) return false;
    slot = (slot + 1) & lm.mask;
  }
}

// Scorer::get_log_cond_prob(ngram) for an n-gram window of `order` words (oldest first, <s>-padded): natural-log
// probability of the LAST word given the others, or OOV_SCORE when ANY word of the window is out of vocabulary
// (scorer.cpp returns OOV_SCORE from inside the loop over the window).  Back-off recursion of ARPA models:
//   p(w | ctx) = p(ctx w)                      if the n-gram exists
//              = bo(ctx) * p(w | ctx[1:])      otherwise (bo = 1 when the context is not in the model)
// KenLM accumulates prob + back-offs in float; the division by log10(e) is done in double (scorer.cpp).
__device__ inline double lm_log_cond_prob(const LmDev& lm, const int32_t* win /*[order]*/) {
  const int order = lm.order;
  for (int i = 0; i < order; ++i)
    if (win[i] == 0) return kLmOovScore;
  float acc = 0.f;
  for (int n = order; n >= 1; --n) {
    float p, b;
    if (lm_find(lm, win + order - n, n, p, b)) return (double)(acc + p) / (double)kLmLog10E;
    if (n > 1 && lm_find(lm, win + order - n, n - 1, p, b)) acc += b;
  }
  return kLmOovScore;  // the unigram of an in-vocabulary word always exists; defensive
}

}  // namespace ppasr
