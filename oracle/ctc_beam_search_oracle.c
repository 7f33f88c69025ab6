/*
 * ORACLE (test infrastructure, not product code) -- plain-C restatement of the CTC prefix beam
 * search that PPASR calls through `paddlespeech_ctcdecoders` (call sites:
 * ppasr/decoders/swig_wrapper.py:61-62,98-100,119-121; ppasr/decoders/beam_search_decoder.py:49,64,86-91).
 *
 * PARITY UNPINNED: that module is a third-party SWIG/C++ dependency that is NOT in the reference
 * tree, has no version pin (docs/beam_search.md:3-6 installs `-U` from a private index) and cannot
 * be installed offline; the reference has no tests or golden vectors at this boundary.  This file
 * restates the published algorithm of PaddleSpeech `third_party/ctc_decoders`
 * (ctc_beam_search_decoder.cpp / path_trie.cpp / decoder_utils.cpp, itself DeepSpeech's decoder):
 * float-valued log probabilities on a prefix trie, per-frame vocabulary pruning
 * (cutoff_prob / cutoff_top_n), top-`beam_size` selection with prefix_compare (score desc, then
 * last character asc), result score = -log P(prefix).  The `ext_scorer` branch (scorer.cpp, character-based
 * back-off n-gram model: min_cutoff pruning, alpha * ln P_lm + beta per extension, approx_ctc result score) is restated
 * over n-gram arrays handed in by the tests (which parse the ARPA file themselves); KenLM itself is not available.
 *
 * Build: make -C oracle   ->  oracle/_build/libctc_beam_oracle.so   (loaded with ctypes by tests/)
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NUM_FLT_INF FLT_MAX
#define NUM_FLT_MIN FLT_MIN

typedef struct PathTrie {
  float log_prob_b_prev, log_prob_nb_prev, log_prob_b_cur, log_prob_nb_cur, score;
  int character;
  int exists;
  int dict_state; /* word-based scorer: state of the dictionary acceptor this prefix has reached (path_trie.h dictionary_state_) */
  struct PathTrie* parent;
  struct PathTrie** children;
  int n_children, cap_children;
  int* child_slot; /* test-speed only: character -> index into children[] + 1 (0 = none), allocated once a node has more
                    * than CHILD_INDEX_MIN children (unpruned searches give every prefix V children per frame; upstream's
                    * linear scan over them is O(V^2) per prefix).  Same lookups, same container order. */
} PathTrie;
#define CHILD_INDEX_MIN 32
static int g_vocab_size = 0; /* size of child_slot[] (set by ctc_beam_oracle_create; the tests use one V at a time) */

static PathTrie* trie_new(int ch, PathTrie* parent) {
  PathTrie* t = (PathTrie*)calloc(1, sizeof(PathTrie));
  t->log_prob_b_prev = t->log_prob_nb_prev = t->log_prob_b_cur = t->log_prob_nb_cur = t->score = -NUM_FLT_INF;
  t->character = ch;
  t->exists = 1;
  t->parent = parent;
  return t;
}

static void trie_free(PathTrie* t) {
  for (int i = 0; i < t->n_children; ++i) trie_free(t->children[i]);
  free(t->children);
  free(t->child_slot);
  free(t);
}

static float log_sum_exp(float x, float y) {
  if (x <= -NUM_FLT_INF) return y;
  if (y <= -NUM_FLT_INF) return x;
  float m = x > y ? x : y;
  return logf(expf(x - m) + expf(y - m)) + m;
}

/* the dictionary of a word-based scorer: upstream an OpenFST acceptor of every vocabulary word spelt in characters + the
 * space (Scorer::fill_dictionary), here handed in by the tests as a character trie in CSR form (node 0 = start; a node is
 * final iff a word ends there, word[node] = its LM word index) */
typedef struct {
  int n_nodes;
  const int *first, *arc_char, *arc_next, *word;
} Dict;
static int dict_arc(const Dict* d, int s, int c) {
  for (int a = d->first[s]; a < d->first[s + 1]; ++a)
    if (d->arc_char[a] == c) return d->arc_next[a];
  return -1;
}

/* PathTrie::get_path_trie(new_char, reset=true); dict == NULL: no dictionary */
static PathTrie* get_path_trie(PathTrie* t, int c, const Dict* dict) {
  int lo = 0, hi = t->n_children;
  if (t->child_slot) { /* indexed: look at the one slot (or none) instead of all children */
    const int s1 = t->child_slot[c];
    lo = s1 ? s1 - 1 : 0;
    hi = s1 ? s1 : 0;
  }
  for (int i = lo; i < hi; ++i) {
    PathTrie* ch = t->children[i];
    if (ch->character == c) {
      if (!ch->exists) {
        ch->exists = 1;
        ch->log_prob_b_prev = ch->log_prob_nb_prev = ch->log_prob_b_cur = ch->log_prob_nb_cur = -NUM_FLT_INF;
      }
      return ch;
    }
  }
  int to = 0;
  if (dict) {
    /* matcher_->Find(new_char + 1) from dictionary_state_: not found -> no child; and if the state is final (a word has
     * just ended) the state is reset to the start state as a side effect (path_trie.cpp, `if (is_final && reset)`) */
    to = dict_arc(dict, t->dict_state, c);
    if (to < 0) {
      if (dict->word[t->dict_state] != 0) t->dict_state = 0;
      return NULL;
    }
  }
  if (t->n_children == t->cap_children) {
    t->cap_children = t->cap_children ? 2 * t->cap_children : 4;
    t->children = (PathTrie**)realloc(t->children, sizeof(PathTrie*) * t->cap_children);
  }
  PathTrie* n = trie_new(c, t);
  n->dict_state = to;
  t->children[t->n_children++] = n;
  if (t->child_slot) {
    t->child_slot[c] = t->n_children;
  } else if (t->n_children > CHILD_INDEX_MIN && g_vocab_size > 0) {
    t->child_slot = (int*)calloc(g_vocab_size, sizeof(int));
    for (int i = 0; i < t->n_children; ++i) t->child_slot[t->children[i]->character] = i + 1;
  }
  return n;
}

typedef struct {
  PathTrie** v;
  int n, cap;
} Vec;
static void vec_push(Vec* a, PathTrie* p) {
  if (a->n == a->cap) {
    a->cap = a->cap ? 2 * a->cap : 64;
    a->v = (PathTrie**)realloc(a->v, sizeof(PathTrie*) * a->cap);
  }
  a->v[a->n++] = p;
}

/* PathTrie::iterate_to_vec */
static void iterate_to_vec(PathTrie* t, Vec* out) {
  if (t->exists) {
    t->log_prob_b_prev = t->log_prob_b_cur;
    t->log_prob_nb_prev = t->log_prob_nb_cur;
    t->log_prob_b_cur = -NUM_FLT_INF;
    t->log_prob_nb_cur = -NUM_FLT_INF;
    t->score = log_sum_exp(t->log_prob_b_prev, t->log_prob_nb_prev);
    vec_push(out, t);
  }
  for (int i = 0; i < t->n_children; ++i) iterate_to_vec(t->children[i], out);
}

/* PathTrie::remove */
static void trie_remove(PathTrie* t) {
  t->exists = 0;
  if (t->n_children == 0 && t->parent) {
    PathTrie* p = t->parent;
    int at = -1;
    if (p->child_slot) at = p->child_slot[t->character] - 1;
    else
      for (int i = 0; i < p->n_children; ++i)
        if (p->children[i] == t) { at = i; break; }
    if (at >= 0) { /* swap-with-last removal, as before */
      p->children[at] = p->children[--p->n_children];
      if (p->child_slot) {
        p->child_slot[t->character] = 0;
        if (at < p->n_children) p->child_slot[p->children[at]->character] = at + 1;
      }
    }
    free(t->children);
    free(t->child_slot);
    free(t);
    if (p->n_children == 0 && !p->exists) trie_remove(p);
  }
}

/* prefix_compare: true if x ranks before y */
static int prefix_before(const PathTrie* x, const PathTrie* y) {
  if (x->score == y->score) {
    if (x->character == y->character) return 0;
    return x->character < y->character;
  }
  return x->score > y->score;
}
static int cmp_prefix(const void* a, const void* b) {
  const PathTrie* x = *(PathTrie* const*)a;
  const PathTrie* y = *(PathTrie* const*)b;
  if (prefix_before(x, y)) return -1;
  if (prefix_before(y, x)) return 1;
  return 0;
}

typedef struct {
  int idx;
  double p;
} ProbIdx;
static int cmp_prob_desc(const void* a, const void* b) {
  double x = ((const ProbIdx*)a)->p, y = ((const ProbIdx*)b)->p;
  if (x > y) return -1;
  if (x < y) return 1;
  int i = ((const ProbIdx*)a)->idx, j = ((const ProbIdx*)b)->idx; /* stable tie-break for reproducibility */
  return (i > j) - (i < j);
}

/* get_pruned_log_probs (decoder_utils.cpp); NB: with cutoff_prob >= 1 the list is sorted but NOT
 * truncated to cutoff_top_n (upstream behaviour, kept). returns count, fills idx/logp (caller: size V) */
static int pruned_log_probs(const float* prob, int V, double cutoff_prob, int cutoff_top_n, int* idx, float* logp,
                            ProbIdx* tmp) {
  for (int i = 0; i < V; ++i) {
    tmp[i].idx = i;
    tmp[i].p = (double)prob[i];
  }
  int cutoff_len = V;
  if (cutoff_prob < 1.0 || cutoff_top_n < cutoff_len) {
    qsort(tmp, V, sizeof(ProbIdx), cmp_prob_desc);
    if (cutoff_prob < 1.0) {
      double cum = 0.0;
      cutoff_len = 0;
      for (int i = 0; i < V; ++i) {
        cum += tmp[i].p;
        cutoff_len += 1;
        if (cum >= cutoff_prob || cutoff_len >= cutoff_top_n) break;
      }
    }
  }
  for (int i = 0; i < cutoff_len; ++i) {
    idx[i] = tmp[i].idx;
    logp[i] = (float)log(tmp[i].p + NUM_FLT_MIN);
  }
  return cutoff_len;
}

/* ---- external scorer (scorer.cpp), character-based ---- */
#define LM_MAX_ORDER 6
#define OOV_SCORE (-1000.0)
static const float NUM_FLT_LOGE = 0.4342944819f;
typedef struct {
  int n;                 /* order of this n-gram */
  int w[LM_MAX_ORDER];   /* LM word ids */
  float prob, backoff;   /* log10 */
} NGram;
static int cmp_ngram(const void* a, const void* b) {
  const NGram* x = (const NGram*)a;
  const NGram* y = (const NGram*)b;
  if (x->n != y->n) return x->n - y->n;
  for (int i = 0; i < x->n; ++i)
    if (x->w[i] != y->w[i]) return x->w[i] < y->w[i] ? -1 : 1;
  return 0;
}
typedef struct {
  int order, n_grams, bos, eos;
  NGram* grams; /* sorted by (n, words) */
  int* tok2lm;  /* [V], 0 = OOV */
  double alpha, beta;
  int word_based, space_id; /* scorer.cpp is_character_based_ == false: scored at spaces, dictionary-constrained */
  Dict dict;
} Lm;
static const NGram* lm_find(const Lm* lm, const int* w, int n) {
  NGram key;
  key.n = n;
  for (int i = 0; i < n; ++i) key.w[i] = w[i];
  return (const NGram*)bsearch(&key, lm->grams, lm->n_grams, sizeof(NGram), cmp_ngram);
}
/* Scorer::get_log_cond_prob: ln P(last word | the others), OOV_SCORE if any word of the window is unknown;
 * ARPA back-off recursion, float accumulation like KenLM, / log10(e) in double */
static double lm_log_cond_prob(const Lm* lm, const int* win) {
  for (int i = 0; i < lm->order; ++i)
    if (win[i] == 0) return OOV_SCORE;
  float acc = 0.f;
  for (int n = lm->order; n >= 1; --n) {
    const NGram* g = lm_find(lm, win + lm->order - n, n);
    if (g) return (double)(acc + g->prob) / (double)NUM_FLT_LOGE;
    if (n > 1) {
      const NGram* c = lm_find(lm, win + lm->order - n, n - 1);
      if (c) acc += c->backoff;
    }
  }
  return OOV_SCORE;
}
/* Scorer::make_ngram for a character-based model: the last `order` characters of the prefix, <s>-padded */
static void make_ngram(const Lm* lm, const PathTrie* node, int* win) {
  for (int j = lm->order - 1; j >= 0; --j) {
    if (node && node->parent) {
      win[j] = lm->tok2lm[node->character];
      node = node->parent;
    } else {
      win[j] = lm->bos;
    }
  }
}

/* LM word index of the characters chars[0..n-1] (a word of the hypothesis): spelt through the dictionary; 0 = OOV */
static int word_index_of(const Lm* lm, const int* chars, int n) {
  int s = 0;
  for (int i = 0; i < n && s >= 0; ++i) s = dict_arc(&lm->dict, s, chars[i]);
  if (s >= 0) s = dict_arc(&lm->dict, s, lm->space_id);
  return s >= 0 ? lm->dict.word[s] : 0;
}
/* Scorer::make_ngram for a word-based model: the last `order` WORDS of the prefix (split at the space token), <s>-padded */
static void make_ngram_words(const Lm* lm, const PathTrie* node, int* win) {
  int j = lm->order - 1;
  const PathTrie* cur = node;
  for (; j >= 0; --j) {
    int chars[4096], n = 0;
    const PathTrie* p = cur;
    while (p->parent && p->character != lm->space_id) { /* get_path_vec(stop = SPACE_ID_) */
      if (n < 4096) chars[n++] = p->character;
      p = p->parent;
    }
    for (int a = 0, b = n - 1; a < b; ++a, --b) { int t = chars[a]; chars[a] = chars[b]; chars[b] = t; }
    win[j] = word_index_of(lm, chars, n);
    if (!p->parent) { /* reached the root: pad with <s> */
      for (int q = j - 1; q >= 0; --q) win[q] = lm->bos;
      break;
    }
    cur = p->parent; /* skipping the space */
  }
}

typedef struct {
  PathTrie* root;
  Vec prefixes;
  int V, beam_size, cutoff_top_n, blank_id;
  double cutoff_prob;
  int* idx;
  float* logp;
  ProbIdx* tmp;
  Lm* lm;
} Decoder;

void* ctc_beam_oracle_create(int V, int beam_size, double cutoff_prob, int cutoff_top_n, int blank_id) {
  Decoder* d = (Decoder*)calloc(1, sizeof(Decoder));
  g_vocab_size = V;
  d->V = V;
  d->beam_size = beam_size;
  d->cutoff_prob = cutoff_prob;
  d->cutoff_top_n = cutoff_top_n;
  d->blank_id = blank_id;
  d->idx = (int*)malloc(sizeof(int) * V);
  d->logp = (float*)malloc(sizeof(float) * V);
  d->tmp = (ProbIdx*)malloc(sizeof(ProbIdx) * V);
  d->root = trie_new(-1, NULL);
  d->root->score = d->root->log_prob_b_prev = 0.0f;
  vec_push(&d->prefixes, d->root);
  return d;
}

/* attach a character-based n-gram model: gram_n[i] words of gram_w[i*6 ..], log10 prob / backoff */
void ctc_beam_oracle_set_lm(void* h, int order, int n_grams, const int* gram_n, const int* gram_w, const float* prob,
                            const float* backoff, const int* tok2lm, int bos, int eos, double alpha, double beta) {
  Decoder* d = (Decoder*)h;
  Lm* lm = (Lm*)calloc(1, sizeof(Lm));
  lm->order = order;
  lm->n_grams = n_grams;
  lm->bos = bos;
  lm->eos = eos;
  lm->alpha = alpha;
  lm->beta = beta;
  lm->grams = (NGram*)calloc(n_grams, sizeof(NGram));
  for (int i = 0; i < n_grams; ++i) {
    lm->grams[i].n = gram_n[i];
    for (int j = 0; j < gram_n[i]; ++j) lm->grams[i].w[j] = gram_w[(size_t)i * LM_MAX_ORDER + j];
    lm->grams[i].prob = prob[i];
    lm->grams[i].backoff = backoff[i];
  }
  qsort(lm->grams, n_grams, sizeof(NGram), cmp_ngram);
  lm->tok2lm = (int*)malloc(sizeof(int) * d->V);
  memcpy(lm->tok2lm, tok2lm, sizeof(int) * d->V);
  d->lm = lm;
}

/* make the attached scorer word-based: the space token and the dictionary (arrays owned by the caller, must outlive h) */
void ctc_beam_oracle_set_dictionary(void* h, int space_id, int n_nodes, const int* first, const int* arc_char,
                                    const int* arc_next, const int* word) {
  Decoder* d = (Decoder*)h;
  d->lm->word_based = 1;
  d->lm->space_id = space_id;
  d->lm->dict.n_nodes = n_nodes;
  d->lm->dict.first = first;
  d->lm->dict.arc_char = arc_char;
  d->lm->dict.arc_next = arc_next;
  d->lm->dict.word = word;
}

void ctc_beam_oracle_free(void* h) {
  Decoder* d = (Decoder*)h;
  if (d->lm) {
    free(d->lm->grams);
    free(d->lm->tok2lm);
    free(d->lm);
  }
  trie_free(d->root);
  free(d->prefixes.v);
  free(d->idx);
  free(d->logp);
  free(d->tmp);
  free(d);
}

/* the per-frame body of ctc_beam_search_decoding / CtcBeamSearchDecoderStorage::next */
void ctc_beam_oracle_next(void* h, const float* probs, int T) {
  Decoder* d = (Decoder*)h;
  for (int t = 0; t < T; ++t) {
    const float* prob = probs + (size_t)t * d->V;
    float min_cutoff = -NUM_FLT_INF;
    int full_beam = 0;
    if (d->lm) {
      int num_prefixes = d->prefixes.n < d->beam_size ? d->prefixes.n : d->beam_size;
      qsort(d->prefixes.v, num_prefixes, sizeof(PathTrie*), cmp_prefix);
      min_cutoff = (float)((double)d->prefixes.v[num_prefixes - 1]->score + log((double)prob[d->blank_id]) -
                           (d->lm->beta > 0.0 ? d->lm->beta : 0.0));
      full_beam = (num_prefixes == d->beam_size);
    }
    int n = pruned_log_probs(prob, d->V, d->cutoff_prob, d->cutoff_top_n, d->idx, d->logp, d->tmp);
    for (int k = 0; k < n; ++k) {
      int c = d->idx[k];
      float log_prob_c = d->logp[k];
      for (int i = 0; i < d->prefixes.n && i < d->beam_size; ++i) {
        PathTrie* prefix = d->prefixes.v[i];
        if (full_beam && log_prob_c + prefix->score < min_cutoff) break;
        if (c == d->blank_id) {
          prefix->log_prob_b_cur = log_sum_exp(prefix->log_prob_b_cur, log_prob_c + prefix->score);
          continue;
        }
        if (c == prefix->character)
          prefix->log_prob_nb_cur = log_sum_exp(prefix->log_prob_nb_cur, log_prob_c + prefix->log_prob_nb_prev);
        PathTrie* pn = get_path_trie(prefix, c, (d->lm && d->lm->word_based) ? &d->lm->dict : NULL);
        if (!pn) continue; /* the dictionary has no such spelling */
        float log_p = -NUM_FLT_INF;
        if (c == prefix->character && prefix->log_prob_b_prev > -NUM_FLT_INF)
          log_p = log_prob_c + prefix->log_prob_b_prev;
        else if (c != prefix->character)
          log_p = log_prob_c + prefix->score;
        if (d->lm && (!d->lm->word_based || c == d->lm->space_id)) {
          /* character-based scorer: every extension is scored on the NEW prefix; word-based: a space scores the word it
           * completes, i.e. the OLD prefix (`prefix_to_score = prefix`) */
          int win[LM_MAX_ORDER];
          if (d->lm->word_based) make_ngram_words(d->lm, prefix, win);
          else make_ngram(d->lm, pn, win);
          float score = (float)(lm_log_cond_prob(d->lm, win) * d->lm->alpha);
          log_p += score;
          log_p = (float)((double)log_p + d->lm->beta);
        }
        pn->log_prob_nb_cur = log_sum_exp(pn->log_prob_nb_cur, log_p);
      }
    }
    d->prefixes.n = 0;
    iterate_to_vec(d->root, &d->prefixes);
    if (d->prefixes.n >= d->beam_size) {
      /* std::nth_element + remove the tail: a full sort selects the same set (ties at the cut excepted) */
      qsort(d->prefixes.v, d->prefixes.n, sizeof(PathTrie*), cmp_prefix);
      for (int i = d->beam_size; i < d->prefixes.n; ++i) trie_remove(d->prefixes.v[i]);
      d->prefixes.n = d->beam_size;
    }
  }
}

/* get_beam_search_result: top `nbest` prefixes, score = -log P, tokens padded with -1 to max_len.
 * returns the number of results written. */
int ctc_beam_oracle_result(void* h, int nbest, int max_len, int* tokens, int* lens, double* scores) {
  Decoder* d = (Decoder*)h;
  int n = d->prefixes.n < d->beam_size ? d->prefixes.n : d->beam_size;
  PathTrie** s = (PathTrie**)malloc(sizeof(PathTrie*) * (n ? n : 1));
  memcpy(s, d->prefixes.v, sizeof(PathTrie*) * n);
  /* word-based scorer: "score the last word of each prefix that doesn't end with space" -- added to the score the result
   * is ranked by.  (Upstream adds it to prefix->score in place; here the addition is undone after the ranking so that a
   * streaming caller may ask for the current result after every chunk.) */
  float* saved = (float*)malloc(sizeof(float) * (n ? n : 1));
  for (int i = 0; i < n; ++i) {
    saved[i] = s[i]->score;
    if (d->lm && d->lm->word_based && s[i]->parent && s[i]->character != d->lm->space_id) {
      int win[LM_MAX_ORDER];
      make_ngram_words(d->lm, s[i], win);
      float score = (float)(lm_log_cond_prob(d->lm, win) * d->lm->alpha);
      score = (float)((double)score + d->lm->beta);
      s[i]->score += score;
    }
  }
  qsort(s, n, sizeof(PathTrie*), cmp_prefix);
  int out = n < nbest ? n : nbest;
  for (int i = 0; i < out; ++i) {
    int len = 0;
    for (PathTrie* p = s[i]; p->parent; p = p->parent) ++len;
    lens[i] = len;
    int* row = tokens + (size_t)i * max_len;
    for (int j = 0; j < max_len; ++j) row[j] = -1;
    int j = len;
    for (PathTrie* p = s[i]; p->parent; p = p->parent) {
      --j;
      if (j < max_len) row[j] = p->character;
    }
    double approx_ctc = (double)s[i]->score;
    if (d->lm) {
      /* approx_ctc -= prefix_length * beta + alpha * get_sent_log_prob(words); sentence = <s>^(order-1) words </s>,
       * one window per position (scorer.cpp get_sent_log_prob / get_log_prob); words = characters (character-based) or
       * the pieces between spaces (word-based, split_labels) */
      const Lm* lm = d->lm;
      int* chars = (int*)malloc(sizeof(int) * (len ? len : 1));
      {
        int jj = len;
        for (PathTrie* p = s[i]; p->parent; p = p->parent) chars[--jj] = p->character;
      }
      int* words = (int*)malloc(sizeof(int) * (len + 1));
      int n_words = 0;
      if (lm->word_based) {
        int start = 0;
        for (int q = 0; q <= len; ++q) {
          if (q == len || chars[q] == lm->space_id) {
            if (q > start) words[n_words++] = word_index_of(lm, chars + start, q - start);
            start = q + 1;
          }
        }
      } else {
        for (int q = 0; q < len; ++q) words[n_words++] = lm->tok2lm[chars[q]];
      }
      int total = lm->order - 1 + n_words + 1;
      if (n_words == 0) total = lm->order + 1;
      int* sent = (int*)malloc(sizeof(int) * total);
      int pos = 0;
      for (int q = 0; q < (n_words == 0 ? lm->order : lm->order - 1); ++q) sent[pos++] = lm->bos;
      for (int q = 0; q < n_words; ++q) sent[pos++] = words[q];
      sent[pos++] = lm->eos;
      double lp = 0.0;
      for (int q = 0; q + lm->order <= total; ++q) lp += lm_log_cond_prob(lm, sent + q);
      free(sent);
      free(words);
      free(chars);
      approx_ctc = approx_ctc - (double)len * lm->beta - lp * lm->alpha;
    }
    scores[i] = -approx_ctc;
  }
  for (int i = 0; i < n; ++i) d->prefixes.v[i]->score = saved[i]; /* (saved[] is in the prefixes' own order) */
  free(saved);
  free(s);
  return out;
}

/* one-shot: ctc_beam_search_decoding(probs_seq, vocabulary, beam_size, cutoff_prob, cutoff_top_n, NULL, blank_id) */
int ctc_beam_oracle_decode(const float* probs, int T, int V, int beam_size, double cutoff_prob, int cutoff_top_n,
                           int blank_id, int nbest, int max_len, int* tokens, int* lens, double* scores) {
  void* d = ctc_beam_oracle_create(V, beam_size, cutoff_prob, cutoff_top_n, blank_id);
  ctc_beam_oracle_next(d, probs, T);
  int n = ctc_beam_oracle_result(d, nbest, max_len, tokens, lens, scores);
  ctc_beam_oracle_free(d);
  return n;
}
