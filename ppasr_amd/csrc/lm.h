// lm.h -- back-off n-gram language model on the device, for the external scorer of the CTC prefix beam search
// (PaddleSpeech third_party/ctc_decoders scorer.cpp: `Scorer::get_log_cond_prob` walks KenLM's `BaseScore` over the
// n-gram window; PPASR builds it in decoders/beam_search_decoder.py:28-29 via decoders/swig_wrapper.py:18-33).
//
// All n-grams of all orders live in ONE open-addressing hash table keyed by a 64-bit hash of (order, word ids) --
// the layout KenLM's "probing" model uses, flattened: 16-byte slots (key, log10 prob, log10 back-off), load factor
// <= 1/3, so that one probe is one (or two adjacent) 16-byte loads.  The hash folds the words from the NEWEST backwards
// (like KenLM's own chain), so the keys of all suffixes of a window share their prefix computation.
// Two ways to evaluate Scorer::get_log_cond_prob:
//   * lm_log_cond_prob: the plain back-off walk, up to 2*order - 1 dependent probes (sentence scores at the end);
//   * the factorised form the search uses per (hypothesis, candidate) pair: the back-off weights depend on the
//     hypothesis' context only (lm_context_acc, once per hypothesis and frame), the unigram on the candidate only (a
//     table), so a pair costs the order-1 probes of the n-grams that END in the candidate, all issued at once
//     (lm_pair_log_cond_prob).  Same float sums in the same order: identical values.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ppasr {

constexpr int kLmMaxOrder = 6;
constexpr double kLmOovScore = -1000.0;              // OOV_SCORE (scorer.h)
constexpr float kLmLog10E = 0.4342944819f;           // NUM_FLT_LOGE (decoder_utils.h)

struct LmSlot {
  uint64_t key;              // 0 = empty slot
  float prob;                // log10 P
  float backoff;             // log10 back-off weight (0 when absent)
};

struct LmDev {
  int order;                 // 0 = no language model
  int bos, eos;              // word index of <s>, </s>
  uint32_t mask;             // table size - 1 (power of two); the array holds one more slot, a copy of slot 0
  const LmSlot* slots;
  const float* uni_prob;     // [n_words] log10 P of the unigram of LM word w (NaN: no such unigram)
  int n_words;
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
// ---- suffix-incremental keys: fold the words of an n-gram from the NEWEST backwards; the key of the n-gram made of the
// last n words folded so far is lm_fold_key(h, n) ----
__host__ __device__ inline uint64_t lm_fold_init(int kenlm_keys, uint32_t newest) {
  return kenlm_keys ? (uint64_t)newest : lm_mix(0xcbf29ce484222325ull, (uint64_t)newest);
}
__host__ __device__ inline uint64_t lm_fold(int kenlm_keys, uint64_t h, uint32_t older) {
  return kenlm_keys ? kenlm_combine(h, older) : lm_mix(h, (uint64_t)older);
}
__host__ __device__ inline uint64_t lm_fold_key(int kenlm_keys, uint64_t h, int n) {
  return kenlm_keys ? lm_key_from_kenlm(h, n) : (lm_mix(h ^ 0x9ae16a3b2f90404full, (uint64_t)n) | 1ull);
}
// key of the n-gram w[0..n-1] (oldest first; never 0)
__host__ __device__ inline uint64_t lm_key(const int32_t* w, int n) {
  uint64_t h = lm_fold_init(0, (uint32_t)w[n - 1]);
  for (int i = n - 2; i >= 0; --i) h = lm_fold(0, h, (uint32_t)w[i]);
  return lm_fold_key(0, h, n);
}
__host__ __device__ inline uint64_t lm_key_any(int kenlm_keys, const int32_t* w, int n) {
  return kenlm_keys ? lm_key_from_kenlm(kenlm_chain(w, n), n) : lm_key(w, n);
}
__host__ __device__ inline uint32_t lm_slot_of(uint64_t key, uint32_t mask) { return (uint32_t)(key >> 17) & mask; }

// look-up of one key; `t`: the slot table (device or host copy)
__host__ __device__ inline bool lm_find_key(const LmSlot* t, uint32_t mask, uint64_t key, float& prob, float& backoff) {
  uint32_t slot = lm_slot_of(key, mask);
  for (;;) {
    const LmSlot s = t[slot];
    if (s.key == key) {
      prob = s.prob;
      backoff = s.backoff;
      return true;
    }
    if (s.key == 0) return false;
    slot = (slot + 1) & mask;
  }
}
__device__ inline bool lm_find(const LmDev& lm, const int32_t* w, int n, float& prob, float& backoff) {
  return lm_find_key(lm.slots, lm.mask, lm_key_any(lm.kenlm_keys, w, n), prob, backoff);
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

// ---- the factorised form (see the header comment) ----
// Probes of up to kLmPar keys issued TOGETHER: both candidate slots of every key are requested before the first is looked
// at (load factor <= 1/3: a key or the end of its cluster is in the first two slots in ~95 % of the probes; the rest
// continue slot by slot).  want[j] selects the keys; found[j] / prob[j] / backoff[j] are the results.  (Three at a time:
// the slots in flight are registers, and a 1 024-thread search workgroup has 128 per lane; orders above 4 take a second
// round.)
constexpr int kLmPar = 3;
__device__ inline void lm_probe_many(const LmDev& lm, const uint64_t* key, const bool* want, bool* found, float* prob,
                                     float* backoff) {
  LmSlot s0[kLmPar], s1[kLmPar];
#pragma unroll
  for (int j = 0; j < kLmPar; ++j) {
    if (want[j]) {
      const uint32_t slot = lm_slot_of(key[j], lm.mask);
      s0[j] = lm.slots[slot];
      s1[j] = lm.slots[slot + 1];  // (the array carries a copy of slot 0 behind its last slot)
    }
  }
#pragma unroll
  for (int j = 0; j < kLmPar; ++j) {
    found[j] = false;
    prob[j] = 0.f;
    backoff[j] = 0.f;
    if (want[j]) {
      if (s0[j].key == key[j]) {
        found[j] = true; prob[j] = s0[j].prob; backoff[j] = s0[j].backoff;
      } else if (s0[j].key != 0) {
        if (s1[j].key == key[j]) {
          found[j] = true; prob[j] = s1[j].prob; backoff[j] = s1[j].backoff;
        } else if (s1[j].key != 0) {  // a cluster longer than two slots: continue the walk
          uint32_t slot = (lm_slot_of(key[j], lm.mask) + 2) & lm.mask;
          for (;;) {
            const LmSlot s = lm.slots[slot];
            if (s.key == key[j]) { found[j] = true; prob[j] = s.prob; backoff[j] = s.backoff; break; }
            if (s.key == 0) break;
            slot = (slot + 1) & lm.mask;
          }
        }
      }
    }
  }
}

// Context summary of a hypothesis: ctx = its last order-1 words (oldest first).  acc[n], n = order .. 1: the float sum of
// back-off weights lm_log_cond_prob has accumulated when it consults level n (acc[order] = 0; acc[n-1] = acc[n] + bo of
// the (n-1)-gram made of the last n-1 context words, when that n-gram exists).  acc[0] != 0: a context word is OOV.
__device__ inline void lm_context_acc(const LmDev& lm, const int32_t* ctx, float* acc /*[kLmMaxOrder + 1]*/) {
  const int order = lm.order;
  bool oov = false;
  for (int i = 0; i < order - 1; ++i) oov |= ctx[i] == 0;
  // back-off of the m-gram = last m context words, m = 1 .. order - 1, in rounds of kLmPar (the fold runs newest -> oldest)
  float bo[kLmMaxOrder];
  bool has[kLmMaxOrder];
#pragma unroll
  for (int m = 0; m < kLmMaxOrder; ++m) { bo[m] = 0.f; has[m] = false; }
  uint64_t h = 0;
#pragma unroll
  for (int r0 = 1; r0 < kLmMaxOrder; r0 += kLmPar) {
    if (r0 <= order - 1) {
      uint64_t key[kLmPar];
      bool want[kLmPar], found[kLmPar];
      float p[kLmPar], b[kLmPar];
#pragma unroll
      for (int j = 0; j < kLmPar; ++j) {
        const int m = r0 + j;
        want[j] = !oov && m <= order - 1;
        key[j] = 0;
        if (m <= order - 1) {
          const uint32_t w = (uint32_t)ctx[order - 1 - m];
          h = m == 1 ? lm_fold_init(lm.kenlm_keys, w) : lm_fold(lm.kenlm_keys, h, w);
          key[j] = lm_fold_key(lm.kenlm_keys, h, m);
        }
      }
      lm_probe_many(lm, key, want, found, p, b);
#pragma unroll
      for (int j = 0; j < kLmPar; ++j)
        if (r0 + j < kLmMaxOrder) { bo[r0 + j] = b[j]; has[r0 + j] = found[j]; }
    }
  }
  acc[0] = oov ? 1.f : 0.f;
  float a = 0.f;
#pragma unroll
  for (int n = kLmMaxOrder; n >= 1; --n) {
    if (n <= order) {
      acc[n] = a;
      if (n >= 2 && has[n - 1]) a += bo[n - 1];
    }
  }
}

// Scorer::get_log_cond_prob of the window (ctx, word) with the context summary `acc` of lm_context_acc(ctx): the n-grams
// that end in `word` are probed together (highest orders first, kLmPar at a time), the highest order that exists decides.
// Equal to lm_log_cond_prob bit for bit.  `uni`: log10 P of the unigram of `word` (lm.uni_prob[word]; NaN: none)
__device__ inline double lm_pair_log_cond_prob(const LmDev& lm, const int32_t* ctx, const float* acc, int32_t word, float uni) {
  if (acc[0] != 0.f || word == 0) return kLmOovScore;
  const int order = lm.order;
  // keys of the n-grams ending in `word`, n = 2 .. order (the fold runs newest -> oldest: all of them are computed first)
  uint64_t keyn[kLmMaxOrder + 1];
  uint64_t h = lm_fold_init(lm.kenlm_keys, (uint32_t)word);
#pragma unroll
  for (int n = 2; n <= kLmMaxOrder; ++n) {
    keyn[n] = 0;
    if (n <= order) {
      h = lm_fold(lm.kenlm_keys, h, (uint32_t)ctx[order - n]);
      keyn[n] = lm_fold_key(lm.kenlm_keys, h, n);
    }
  }
#pragma unroll
  for (int r = 0; r < (kLmMaxOrder - 1 + kLmPar - 1) / kLmPar; ++r) {
    const int n_hi = order - kLmPar * r;  // this round: n_hi, n_hi - 1, ... (>= 2)
    if (n_hi >= 2) {
      uint64_t key[kLmPar];
      bool want[kLmPar], found[kLmPar];
      float p[kLmPar], b[kLmPar];
#pragma unroll
      for (int j = 0; j < kLmPar; ++j) {
        const int n = n_hi - j;
        want[j] = n >= 2;
        key[j] = 0;
#pragma unroll
        for (int q = 2; q <= kLmMaxOrder; ++q)
          if (q == n) key[j] = keyn[q];
      }
      lm_probe_many(lm, key, want, found, p, b);
#pragma unroll
      for (int j = 0; j < kLmPar; ++j) {
        if (found[j]) {
          const int n = n_hi - j;
          float a = 0.f;
#pragma unroll
          for (int q = 1; q <= kLmMaxOrder; ++q)
            if (q == n) a = acc[q];
          return (double)(a + p[j]) / (double)kLmLog10E;
        }
      }
    }
  }
  if (uni != uni) return kLmOovScore;  // (the unigram of an in-vocabulary word always exists; defensive)
  return (double)(acc[1] + uni) / (double)kLmLog10E;
}
__device__ inline double lm_pair_log_cond_prob(const LmDev& lm, const int32_t* ctx, const float* acc, int32_t word) {
  const float u = (word > 0 && word < lm.n_words) ? lm.uni_prob[word] : __builtin_nanf("");
  return lm_pair_log_cond_prob(lm, ctx, acc, word, u);
}

}  // namespace ppasr
