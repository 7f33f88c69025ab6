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
 * last character asc), result score = -log P(prefix).  No external scorer (no KenLM file is
 * reachable offline): PPASR's `ext_scoring_func` branch is not restated.
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
  struct PathTrie* parent;
  struct PathTrie** children;
  int n_children, cap_children;
} PathTrie;

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
  free(t);
}

static float log_sum_exp(float x, float y) {
  if (x <= -NUM_FLT_INF) return y;
  if (y <= -NUM_FLT_INF) return x;
  float m = x > y ? x : y;
  return logf(expf(x - m) + expf(y - m)) + m;
}

/* PathTrie::get_path_trie(new_char, reset=true) without a dictionary */
static PathTrie* get_path_trie(PathTrie* t, int c) {
  for (int i = 0; i < t->n_children; ++i) {
    PathTrie* ch = t->children[i];
    if (ch->character == c) {
      if (!ch->exists) {
        ch->exists = 1;
        ch->log_prob_b_prev = ch->log_prob_nb_prev = ch->log_prob_b_cur = ch->log_prob_nb_cur = -NUM_FLT_INF;
      }
      return ch;
    }
  }
  if (t->n_children == t->cap_children) {
    t->cap_children = t->cap_children ? 2 * t->cap_children : 4;
    t->children = (PathTrie**)realloc(t->children, sizeof(PathTrie*) * t->cap_children);
  }
  PathTrie* n = trie_new(c, t);
  t->children[t->n_children++] = n;
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
    for (int i = 0; i < p->n_children; ++i)
      if (p->children[i] == t) {
        p->children[i] = p->children[--p->n_children];
        break;
      }
    free(t->children);
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

typedef struct {
  PathTrie* root;
  Vec prefixes;
  int V, beam_size, cutoff_top_n, blank_id;
  double cutoff_prob;
  int* idx;
  float* logp;
  ProbIdx* tmp;
} Decoder;

void* ctc_beam_oracle_create(int V, int beam_size, double cutoff_prob, int cutoff_top_n, int blank_id) {
  Decoder* d = (Decoder*)calloc(1, sizeof(Decoder));
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

void ctc_beam_oracle_free(void* h) {
  Decoder* d = (Decoder*)h;
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
    int n = pruned_log_probs(prob, d->V, d->cutoff_prob, d->cutoff_top_n, d->idx, d->logp, d->tmp);
    for (int k = 0; k < n; ++k) {
      int c = d->idx[k];
      float log_prob_c = d->logp[k];
      for (int i = 0; i < d->prefixes.n && i < d->beam_size; ++i) {
        PathTrie* prefix = d->prefixes.v[i];
        if (c == d->blank_id) {
          prefix->log_prob_b_cur = log_sum_exp(prefix->log_prob_b_cur, log_prob_c + prefix->score);
          continue;
        }
        if (c == prefix->character)
          prefix->log_prob_nb_cur = log_sum_exp(prefix->log_prob_nb_cur, log_prob_c + prefix->log_prob_nb_prev);
        PathTrie* pn = get_path_trie(prefix, c);
        float log_p = -NUM_FLT_INF;
        if (c == prefix->character && prefix->log_prob_b_prev > -NUM_FLT_INF)
          log_p = log_prob_c + prefix->log_prob_b_prev;
        else if (c != prefix->character)
          log_p = log_prob_c + prefix->score;
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
    scores[i] = -(double)s[i]->score;
  }
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
