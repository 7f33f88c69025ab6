// ctc_beam.hip -- CTC prefix beam search on the GPU: a frame-parallel pruning pre-pass, then one workgroup per
// utterance with the beam hypotheses and the frame's pruned character list resident in LDS.
//
// Replaces the third-party C++/SWIG module `paddlespeech_ctcdecoders` that PPASR calls from
// ppasr/decoders/swig_wrapper.py:61-62 (ctc_beam_search_decoding), :98-100 (..._batch) and
// :119-121 (CtcBeamSearchDecoderBatch, streaming), via decoders/beam_search_decoder.py:45-96.
// Algorithm = PaddleSpeech third_party/ctc_decoders (prefix trie, float log-probs, per-frame
// vocabulary pruning by cutoff_prob / cutoff_top_n, top-beam_size selection with prefix_compare:
// score desc, then last character asc), restated on flat arrays:
//   * a hypothesis = (node id, last char, parent node id, log P_blank, log P_nonblank, score); the
//     prefix strings live in a parent-pointer arena in HBM and are only walked at the end;
//   * "does child (prefix, c) already exist in the beam" (the trie lookup) = a flag per list entry set by
//     the hypothesis that IS that child (every hypothesis knows its parent's slot); every hypothesis
//     receives at most two non-blank contributions (its own repeated character, the extension from its
//     parent), so no floating-point atomics and the same log_sum_exp values as the serial trie walk;
//   * top-k = exact MSD radix select of the 32-bit score keys of the frame's element list (ties at the cut:
//     character, then list order), then an ordered compaction; the list holds ~ beam ln beam entries
//     instead of beam x candidates wherever a verified bound allows it (see k_ctc_beam), and lives in HBM
//     scratch when it outgrows LDS (unpruned searches: beam x (1 + V) entries).
// External scorer (`ext_scorer`, character-based n-gram LM, lm.h): the min_cutoff pruning of (character, prefix) pairs,
// alpha * log P_lm + beta on every extension, and the approximate-CTC result score with the LM weight removed.  The LM
// term of a (hypothesis, candidate) pair is computed on the fly inside its score key (the key is inverted back to the
// extension's log-probability when the pair survives); each hypothesis carries its last order-1 LM word ids, shifted on
// extension.
// The vocabulary pruning of ALL frames runs first, frame-parallel (k_ctc_prune); k_ctc_beam consumes its records.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "ctc_beam.h"
#include "launch.h"

namespace ppasr {

namespace {

constexpr int kSmallList = 128;  // element lists up to this many entries are ranked with ballots (two keys per lane of a wave)
constexpr int kTinyBeam = 16;    // beams up to this size take the static-layout form of the clipped list (k_ctc_beam (e'))
constexpr int kBT = 1024;  // threads per utterance: 16 waves = 4 per SIMD (a batch of 32 utterances occupies 32 CUs with one
                          // workgroup each, and every phase is a chain of dependent LDS reads: latency hidden by wave count)
constexpr int kBW = kBT / 64;  // waves
constexpr float kNegInf = -FLT_MAX;  // NUM_FLT_INF of decoder_utils.h
constexpr float kNotCand = FLT_MAX;  // marker in lp[]: character not in the pruned list

__device__ __forceinline__ float lse(float x, float y) {  // log_sum_exp (decoder_utils.h)
  if (x <= kNegInf) return y;
  if (y <= kNegInf) return x;
  float m = fmaxf(x, y);
  return logf(expf(x - m) + expf(y - m)) + m;
}

// ascending-sortable image of a float, inverted so that LARGER scores sort FIRST
__device__ __forceinline__ uint32_t desc_key(float s) {
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~u;
}
__device__ __forceinline__ float score_of_key(uint32_t key) {  // inverse of desc_key
  const uint32_t v = ~key;
  return __uint_as_float((v & 0x80000000u) ? (v & 0x7fffffffu) : ~v);
}
// unique total-order key: score desc | (char+1) asc | element id asc
__device__ __forceinline__ uint64_t make_key(float score, int ch, int id) {
  return ((uint64_t)desc_key(score) << 32) | ((uint64_t)(uint32_t)(ch + 1) << 18) | (uint64_t)(uint32_t)id;
}

struct Beam {  // one double-buffer half, all in LDS
  int* node;
  int* chr;
  int* par;
  float* b;
  float* nb;
  float* score;
  int* ctx;  // [cap][kLmCtx] last LM word ids, most recent last, <s>-padded
  int* dst;  // dictionary state (word-based LM: node of lm.h's character trie the prefix's current word has reached)
  int* pslot;  // slot of the parent hypothesis in the PREVIOUS frame's beam (-1: not there); see k_ctc_beam (d)
};

__device__ __forceinline__ Beam carve_beam(char*& p, int cap) {
  Beam b;
  b.node = reinterpret_cast<int*>(p); p += cap * 4;
  b.chr = reinterpret_cast<int*>(p); p += cap * 4;
  b.par = reinterpret_cast<int*>(p); p += cap * 4;
  b.b = reinterpret_cast<float*>(p); p += cap * 4;
  b.nb = reinterpret_cast<float*>(p); p += cap * 4;
  b.score = reinterpret_cast<float*>(p); p += cap * 4;
  b.ctx = reinterpret_cast<int*>(p); p += cap * 4 * kLmCtx;
  b.dst = reinterpret_cast<int*>(p); p += cap * 4;
  b.pslot = reinterpret_cast<int*>(p); p += cap * 4;
  return b;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load / store
// (vmcnt(0)), which would serialise the record prefetch of the next frame and the arena stores with each step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / row broadcasts (a few cycles per step; the
// __shfl_up formulation goes through ds_bpermute, i.e. one LDS round trip per step -- and these scans sit on the
// critical path of every frame).
__device__ __forceinline__ int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2, 3
  return x;
}

__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int mbcnt(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

// block-wide exclusive scan of one int per thread (256 threads = 4 waves); returns (exclusive, total)
template <int BW>
__device__ __forceinline__ int block_excl_scan(int v, int* wave_tot /*[BW] LDS, private to the call site*/, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int incl = wave_incl_scan(v);
  if (lane == 63) wave_tot[wave] = incl;
  lds_barrier();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < BW; ++w) {
    int t = wave_tot[w];
    if (w < wave) base += t;
    tot += t;
  }
  total = tot;  // no trailing barrier: every call site owns its wave_tot[] and is separated from its next use by others
  return base + incl - v;
}

// Histogram increments with run-length pre-aggregation per thread: consecutive elements of a thread that fall into
// the same bin cost one LDS atomic.  The leading bytes of scores / probabilities are concentrated in a handful of bins
// (nearly every element of the first pass hits the same address, which LDS atomics serialise); with diverse bins this
// degenerates to one atomic per element, as before.
struct RunHist {
  int* hist;
  int bin, cnt;
  __device__ __forceinline__ explicit RunHist(int* h) : hist(h), bin(-1), cnt(0) {}
  __device__ __forceinline__ void add(int b) {
    if (b == bin) { ++cnt; return; }
    if (cnt) atomicAdd(&hist[bin], cnt);
    bin = b;
    cnt = 1;
  }
  __device__ __forceinline__ void flush() {
    if (cnt) atomicAdd(&hist[bin], cnt);
    cnt = 0;
  }
  // flush called by ALL lanes of the wave: the pending runs that share the first pending lane's bin (in the leading-byte
  // passes: all of them) are summed across the wave and cost one atomic instead of one per lane on the same address
  __device__ __forceinline__ void flush_wave() {
    const unsigned long long pend = __ballot(cnt > 0);
    if (pend) {
      const int lane = threadIdx.x & 63;
      const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)pend) - 1);
      const int b0 = __builtin_amdgcn_readlane(bin, leader);
      const bool same = cnt > 0 && bin == b0;
      const int total = __builtin_amdgcn_readlane(wave_incl_scan(same ? cnt : 0), 63);
      if (lane == leader) atomicAdd(&hist[b0], total);
      if (cnt > 0 && !same) atomicAdd(&hist[bin], cnt);
    }
    cnt = 0;
  }
};

// Executed by wave 0 only: find the histogram bin holding the k_rem-th element (1-based) when bins are walked in
// ascending (desc == false) or descending (desc == true) order; writes {bin, rank inside the bin, bin count} to out[0..2].
__device__ __forceinline__ void select_bin(const int* hist, int k_rem, bool desc, int* out) {
  const int lane = threadIdx.x & 63;
  int c[4], sum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pos = 4 * lane + j;  // position in walk order
    c[j] = hist[desc ? 255 - pos : pos];
    sum += c[j];
  }
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  int before = incl - sum;
  const bool mine = (before < k_rem) && (incl >= k_rem);
  if (mine) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (before + c[j] >= k_rem) {
        const int pos = 4 * lane + j;
        out[0] = desc ? 255 - pos : pos;
        out[1] = k_rem - before;
        out[2] = c[j];
        break;
      }
      before += c[j];
    }
  }
}

// select_bin computed redundantly by EVERY wave from the same LDS histogram: results in registers of all lanes, no
// broadcast through LDS and no barrier
__device__ __forceinline__ void select_bin_reg(const int* hist, int k_rem, int& bin, int& rank, int& count) {
  const int lane = threadIdx.x & 63;
  int c[4], sum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    c[j] = hist[4 * lane + j];
    sum += c[j];
  }
  const int incl = wave_incl_scan(sum);
  int before = incl - sum;
  const bool mine = (before < k_rem) && (incl >= k_rem);
  int b = 0, r = 0, n = 0;
  if (mine) {
    bool found = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!found && before + c[j] >= k_rem) {
        b = 4 * lane + j;
        r = k_rem - before;
        n = c[j];
        found = true;
      }
      before += c[j];
    }
  }
  const unsigned long long m = __ballot(mine);
  const int src = __builtin_amdgcn_readfirstlane(m ? (__ffsll((long long)m) - 1) : 0);
  bin = __builtin_amdgcn_readlane(b, src);
  rank = __builtin_amdgcn_readlane(r, src);
  count = __builtin_amdgcn_readlane(n, src);
}

}  // namespace

// ---- LDS plan of k_ctc_beam, shared by the host (size) and the kernel (offsets) ----
struct BeamLdsPlan {
  uint32_t hist, wtot, red, sh, cand, beam0, beam1, newv, rows, surv, lmacc, cmask, frow, kidx, fkey, lkey, lex, total;
};
constexpr int kLmAccWords = kLmMaxOrder + 1;  // lm_context_acc's summary of a hypothesis' context (lm.h)
__host__ __device__ inline uint32_t al8(uint32_t x) { return (x + 7u) & ~7u; }
constexpr int kBeamWords = 8 + kLmCtx;  // node, chr, par, b, nb, score, pslot, dst + LM context words: one beam half per hypothesis
constexpr int kMaskWords = kSmallCand / 32;  // "child exists" bits of a hypothesis: one per candidate of a narrow list
__host__ __device__ inline BeamLdsPlan beam_lds_plan(int beam, int V, int list_cap, bool has_lm, bool wide) {
  BeamLdsPlan p;
  uint32_t o = 0;
  p.hist = o;  o += 8 * 256 * 4;
  p.wtot = o;  o += 4 * 16 * 4;                      // 4 scan call sites x up to 16 waves
  p.red = o;   o += 32 * 4;
  p.sh = o;    o += 64;                              // 16 shared ints
  p.cand = o;  o += 2 * (has_lm ? 4 : 2) * kSmallCand * 4;  // TWO frames of cand_c, cand_lp (records of <= kSmallCand candidates);
                                                           // scorer: + the candidates' LM word ids and unigram log10 probabilities
  p.beam0 = o; o = al8(o + (uint32_t)beam * 4 * kBeamWords);
  p.beam1 = o; o = al8(o + (uint32_t)beam * 4 * kBeamWords);
  p.newv = o;  o = al8(o + (uint32_t)beam * 20);     // new_b, new_nb, new_score, new_dst, k_reset
  p.rows = o;  o = al8(o + (uint32_t)beam * 12 + 4); // rank_of[beam], off[beam + 1], newpos[beam]
  p.surv = o;  o = al8(o + (uint32_t)beam * 8);      // surv[beam], surv_lp[beam]
  p.lmacc = o; o = al8(o + (has_lm ? (uint32_t)beam * 4 * kLmAccWords : 0u));
  p.cmask = o; o = al8(o + (wide ? 0u : (uint32_t)beam * 4 * kMaskWords));
  p.frow = o;  o += 1024 * 2;                        // first_row[thread] (int16)
  p.kidx = o;  o = al8(o + 2 * (uint32_t)((V + 3) & ~3) * 2);  // two frames
  p.fkey = o;  o += kSmallList * 8 + kTinyBeam * 8 + kTinyBeam * 4;  // + srank_key, hyp_of_rank of the tiny-beam path
  p.lkey = o;  o = al8(o + (uint32_t)list_cap * 4);
  p.lex = o;   o = al8(o + (wide ? (uint32_t)list_cap : 0u));  // wide lists: one "child exists" flag per entry
  p.total = (o + 15u) & ~15u;
  return p;
}

size_t beam_lds_bytes(const BeamConfig& c) {
  return beam_lds_plan(c.beam, c.V, c.list_cap, c.lm.order > 0, c.n_cand_max > kSmallCand).total;
}

// largest element list (entries) the LDS of one CU can hold beside the fixed arrays, capped at what a frame can need
int beam_list_cap(int beam, int V, int n_cand_max, bool has_lm) {
  const bool wide = n_cand_max > kSmallCand;
  const size_t fixed = beam_lds_plan(beam, V, 0, has_lm, wide).total, per_entry = wide ? 5 : 4;
  const size_t budget = 160 * 1024 - 64;  // (the hardware limit of a workgroup; small beams stay far below it)
  if (fixed + per_entry * (size_t)kSmallList > budget) return 0;
  size_t cap = (budget - fixed) / per_entry;
  cap &= ~(size_t)7;
  const size_t need = (size_t)beam * (1 + (size_t)n_cand_max);
  const size_t hard = 24 * 1024;  // 24 entries per thread of the largest workgroup: beyond that the list lives in HBM
  if (cap > hard) cap = hard;
  if (cap > need) cap = (need + 7) & ~(size_t)7;
  return (int)cap;
}

// (state layout per utterance: ctc_beam.h)
size_t beam_state_bytes(const BeamConfig& c) { return beam_state_words(c.beam, c.max_nodes) * 4; }

// ---- HBM scratch (ctc_beam.h): [pruning records of all B x T frames, when a record does not fit the state buffer's
// kSmallCand-wide slots] [per utterance: the element list of one frame (score keys, "child exists" flags), when
// beam x (1 + candidates) entries do not fit the LDS list] ----
static size_t scratch_list_bytes_per_utt(const BeamConfig& c) {
  const size_t n = (size_t)c.beam * (1 + (size_t)c.n_cand_max);
  if (n <= (size_t)c.list_cap) return 0;
  return 4 * n + ((n + 15) & ~(size_t)15);
}
static size_t scratch_rec_bytes(const BeamConfig& c, int B, int T) {
  return c.n_cand_max > kSmallCand ? (size_t)B * T * prune_rec_words(c.n_cand_max) * 4 : 0;
}
size_t beam_scratch_bytes(const BeamConfig& c, int B, int T) {
  return ((scratch_rec_bytes(c, B, T) + 255) & ~(size_t)255) + (size_t)B * scratch_list_bytes_per_utt(c);
}

constexpr int kPruneThreads = 256;
static size_t prune_lds_bytes(const BeamConfig& c) {
  const int Vp = (c.V + 3) & ~3;
  return (16 + 256 * 4 + 2 * (kPruneThreads / 64) * 4 + 32 + 4 * kSmallCand * 4 + (size_t)Vp * 4 + 15) & ~(size_t)15;
}


// ---- frame-parallel pre-pass: get_pruned_log_probs of every frame ----
// The pruned character list of a frame does not depend on the beam, so it is computed for all frames at once by a
// chip-wide launch (one workgroup per frame) instead of inside the sequential per-utterance loop.  Record of frame
// (u, t) in HBM (int32 words): [0] C  [1] p_blank (raw probability, float bits)  [2 .. 2+CM) characters in
// (prob desc, index asc) order  [2+CM .. 2+2CM) their log(p + FLT_MIN).
// This kernel: lists of at most kSmallCand (128) characters -- cutoff_prob < 1 with cutoff_top_n <= 128, the shipped
// configurations.  Wider lists (cutoff_prob >= 1: upstream then keeps the WHOLE vocabulary) take k_ctc_prune_wide.
template <int NT>
__global__ __launch_bounds__(NT) void k_ctc_prune(const float* __restrict__ probs, const int32_t* __restrict__ frame_lens,
                                                  int T, BeamConfig cfg, int32_t* __restrict__ recs) {
  constexpr int NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = blockIdx.x, u = blockIdx.y;
  const int n_frames = frame_lens ? min(max(frame_lens[u], 0), T) : T;
  if (t >= n_frames) return;
  const int V = cfg.V, CM = cfg.n_cand_max, blank = cfg.blank;
  const int Vp = (V + 3) & ~3;
  char* p = smem;
  double* sh_d = reinterpret_cast<double*>(p); p += 16;
  int* hist = reinterpret_cast<int*>(p); p += 256 * 4;
  float* red_p = reinterpret_cast<float*>(p); p += NW * 4;
  int* red_i = reinterpret_cast<int*>(p); p += NW * 4;
  int* sh_i = reinterpret_cast<int*>(p); p += 32;
  int* tmp_c = reinterpret_cast<int*>(p); p += kSmallCand * 4;
  float* tmp_p = reinterpret_cast<float*>(p); p += kSmallCand * 4;
  int* cand_c = reinterpret_cast<int*>(p); p += kSmallCand * 4;
  float* cand_lp = reinterpret_cast<float*>(p); p += kSmallCand * 4;
  float* lp = reinterpret_cast<float*>(p); p += (size_t)Vp * 4;
#ifdef PPASR_BEAM_POISON
  for (uint32_t i = tid; i < (uint32_t)(p - smem) / 4; i += NT) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)(PPASR_BEAM_POISON);
  __syncthreads();
#endif
  {
    const float* row = probs + ((size_t)u * T + t) * V;
    for (int v = tid; v < V; v += NT) lp[v] = row[v];
    __syncthreads();
    // ---- (b) get_pruned_log_probs (decoder_utils.cpp): the n_sel largest probabilities in (prob desc, index asc)
    // order, cut where the cumulative probability reaches cutoff_prob.  Exact 4-pass radix select of the n_sel-th
    // largest value, unordered gather, rank sort of the <= 128 survivors.  Excess ties at the threshold (more equal
    // values than slots) fall back to the successive-maxima loop below.
    int C = 0;
    bool slow_path = false;
    const int n_sel = (cfg.cutoff_prob < 1.0) ? min(cfg.cutoff_top_n, V) : V;  // (the host sends n_sel <= CM <= kSmallCand here)
    const bool prune = (cfg.cutoff_prob < 1.0) || (cfg.cutoff_top_n < V);
    if (prune) {
      uint32_t thr_u = 0;
      if (n_sel < V) {
        uint32_t prefix = 0;
        int k_rem = n_sel;
        for (int pass = 0; pass < 4; ++pass) {
          const int shift = 24 - 8 * pass;
          const uint32_t hi_mask = pass == 0 ? 0u : (~0u << (shift + 8));
          if (tid < 256) hist[tid] = 0;
          __syncthreads();
          {
            RunHist rh(hist);
            for (int v = tid; v < V; v += NT) {
              const uint32_t u = __float_as_uint(lp[v]);
              if ((u & hi_mask) == prefix) rh.add((int)((u >> shift) & 0xff));
            }
            rh.flush();
          }
          __syncthreads();
          if (wave == 0) select_bin(hist, k_rem, true, sh_i);
          __syncthreads();
          prefix |= (uint32_t)sh_i[0] << shift;
          k_rem = sh_i[1];
          const int in_class = sh_i[2];
          if (pass == 3 && in_class > k_rem) slow_path = true;  // more values equal to the threshold than slots left
          __syncthreads();
          if (in_class == k_rem) break;  // the whole class is wanted: everything >= prefix (low bits 0) is selected
        }
        thr_u = prefix;
      }
      if (!slow_path) {
        if (tid == 0) sh_i[4] = 0;
        __syncthreads();
        for (int v = tid; v < V; v += NT) {
          const float pv = lp[v];
          if (__float_as_uint(pv) >= thr_u) {
            const int pos = atomicAdd(&sh_i[4], 1);
            if (pos < kSmallCand) { tmp_c[pos] = v; tmp_p[pos] = pv; }
          }
        }
        __syncthreads();
        const int n_got = min(sh_i[4], kSmallCand);
        // rank sort (prob desc, index asc): 8 threads per element, each counting a slice of the list
        for (int idx = tid; idx < 8 * n_got; idx += NT) {  // (NT is a multiple of 8: the 8 parts of an element share a wave)
          const int t = idx >> 3, part = idx & 7;
          const float pv = tmp_p[t];
          const int iv = tmp_c[t];
          int rank = 0;
          for (int s2 = part; s2 < n_got; s2 += 8) rank += (tmp_p[s2] > pv || (tmp_p[s2] == pv && tmp_c[s2] < iv)) ? 1 : 0;
          rank += __shfl_xor(rank, 1);
          rank += __shfl_xor(rank, 2);
          rank += __shfl_xor(rank, 4);
          if (part == 0) {
            cand_c[rank] = iv;
            cand_lp[rank] = pv;  // probability for now
          }
        }
        __syncthreads();
        if (wave == 0) {
          // cumulative cut (sequential double additions in sorted order, like upstream); the sorted probabilities are
          // held by the lanes of wave 0 and read with readlane instead of one LDS round trip per step
          const float p0 = lane < n_got ? cand_lp[lane] : 0.f;
          const float p1 = lane + 64 < n_got ? cand_lp[lane + 64] : 0.f;
          int len = n_got;
          if (cfg.cutoff_prob < 1.0) {
            double cum = 0.0;
            len = 0;
            for (int i = 0; i < n_got; ++i) {
              const float pi = i < 64 ? __shfl(p0, i) : __shfl(p1, i - 64);
              cum += (double)pi;
              len += 1;
              if (cum >= cfg.cutoff_prob || len >= cfg.cutoff_top_n) break;
            }
          }
          if (lane == 0) sh_i[5] = len;
        }
        __syncthreads();
        C = sh_i[5];
        if (tid < C) cand_lp[tid] = (float)log((double)cand_lp[tid] + (double)FLT_MIN);
        __syncthreads();
      }
    }
    if (slow_path) {
      C = 0;
      float last_p = INFINITY;
      int last_i = -1;
      if (tid == 0) sh_d[0] = 0.0;
      for (;;) {
        float bp = -INFINITY;
        int bi = 0x7fffffff;
        for (int v = tid; v < V; v += NT) {
          float pv = lp[v];
          bool after = (pv < last_p) || (pv == last_p && v > last_i);
          if (after && (pv > bp || (pv == bp && v < bi))) { bp = pv; bi = v; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          float p2 = __shfl_xor(bp, o);
          int i2 = __shfl_xor(bi, o);
          if (p2 > bp || (p2 == bp && i2 < bi)) { bp = p2; bi = i2; }
        }
        if (lane == 0) { red_p[wave] = bp; red_i[wave] = bi; }
        __syncthreads();
        bp = red_p[0]; bi = red_i[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          float p2 = red_p[w]; int i2 = red_i[w];
          if (p2 > bp || (p2 == bp && i2 < bi)) { bp = p2; bi = i2; }
        }
        bool stop;
        if (bi == 0x7fffffff) {
          stop = true;  // vocabulary exhausted
        } else {
          if (tid == 0) {
            cand_c[C] = bi;
            cand_lp[C] = (float)log((double)bp + (double)FLT_MIN);
            sh_d[0] += (double)bp;
          }
          C += 1;
          last_p = bp; last_i = bi;
          __syncthreads();
          if (cfg.cutoff_prob < 1.0) stop = (sh_d[0] >= cfg.cutoff_prob) || (C >= cfg.cutoff_top_n);
          else stop = (C >= V);  // upstream sorts but does not truncate when cutoff_prob >= 1
        }
        __syncthreads();
        if (stop || C >= CM) break;
      }
    } else if (!prune) {
      C = V;  // no pruning: vocabulary order (V <= kSmallCand here)
      for (int v = tid; v < V; v += NT) { cand_c[v] = v; cand_lp[v] = (float)log((double)lp[v] + (double)FLT_MIN); }
      __syncthreads();
    }
    int32_t* rec = recs + ((size_t)u * T + t) * prune_rec_words(CM);
    if (tid == 0) { rec[0] = C; rec[1] = __float_as_int(lp[blank]); }
    for (int k = tid; k < C; k += NT) { rec[2 + k] = cand_c[k]; rec[2 + CM + k] = __float_as_int(cand_lp[k]); }
  }
}

// ---- wide candidate lists: more than kSmallCand characters of a frame may survive ----
// cutoff_prob >= 1 is the default of the reference's wrappers (swig_wrapper.py:38,71) and upstream then keeps EVERY
// character: sorted by (prob desc, index asc) when cutoff_top_n < V (get_pruned_log_probs sorts but does not truncate),
// in vocabulary order otherwise.  cutoff_prob < 1 with cutoff_top_n > 128: the sorted list cut at the cumulative
// probability / at top_n.  One workgroup per frame: bitonic sort of the whole row on 64-bit (inverted probability,
// index) keys in LDS, then the sequential double-precision cumulative cut of upstream; the record goes to HBM scratch.
constexpr int kWideThreads = 1024;
static int wide_pow2(int V) {
  int n = 256;
  while (n < V) n <<= 1;
  return n;
}
static size_t prune_wide_lds_bytes(const BeamConfig& c) { return (size_t)wide_pow2(c.V) * 8 + 64; }
__global__ __launch_bounds__(kWideThreads) void k_ctc_prune_wide(const float* __restrict__ probs,
                                                                 const int32_t* __restrict__ frame_lens, int T, BeamConfig cfg,
                                                                 int P2, int32_t* __restrict__ recs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int t = blockIdx.x, u = blockIdx.y;
  const int n_frames = frame_lens ? min(max(frame_lens[u], 0), T) : T;
  if (t >= n_frames) return;
  const int V = cfg.V, CM = cfg.n_cand_max;
  unsigned long long* key = reinterpret_cast<unsigned long long*>(smem);
  int* sh_len = reinterpret_cast<int*>(smem + (size_t)P2 * 8);
  const float* row = probs + ((size_t)u * T + t) * V;
  const bool sort = (cfg.cutoff_prob < 1.0) || (cfg.cutoff_top_n < V);
  for (int v = tid; v < P2; v += kWideThreads)
    key[v] = v < V ? (((unsigned long long)(sort ? ~__float_as_uint(row[v]) : 0u) << 32) | (unsigned long long)(uint32_t)v) : ~0ull;
  __syncthreads();
  if (sort) {  // ascending on (~probability bits, index) = probability descending, index ascending (probabilities are >= 0)
    for (int k = 2; k <= P2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < P2; i += kWideThreads) {
          const int l = i ^ j;
          if (l > i) {
            const unsigned long long a = key[i], b = key[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { key[i] = b; key[l] = a; }
          }
        }
        __syncthreads();
      }
    }
  }
  int len = V;
  if (cfg.cutoff_prob < 1.0) {
    if (tid == 0) {  // upstream's loop: sequential double additions in sorted order
      double cum = 0.0;
      int n = 0;
      for (int i = 0; i < V; ++i) {
        cum += (double)__uint_as_float(~(uint32_t)(key[i] >> 32));
        n += 1;
        if (cum >= cfg.cutoff_prob || n >= cfg.cutoff_top_n) break;
      }
      *sh_len = n;
    }
    __syncthreads();
    len = *sh_len;
  }
  if (len > CM) len = CM;  // (CM = what the rule can produce; defensive)
  int32_t* rec = recs + ((size_t)u * T + t) * prune_rec_words(CM);
  if (tid == 0) { rec[0] = len; rec[1] = __float_as_int(row[cfg.blank]); }
  for (int k = tid; k < len; k += kWideThreads) {
    const int v = (int)(uint32_t)key[k];
    rec[2 + k] = v;
    rec[2 + CM + k] = __float_as_int((float)log((double)row[v] + (double)FLT_MIN));
  }
}

// ---- the search: one workgroup per utterance --------------------------------------------------------------------
// Per frame the ELEMENTS that compete for the next beam are the nb hypotheses already in it and, per hypothesis i (a
// ROW), its children (i, candidate k).  They form one list in element order: [0, nb) the hypotheses, then row after
// row, k ascending; row i holds its first len(i) candidates.  The list's 32-bit score keys sit in LDS (capacity
// cfg.list_cap entries) or, when a frame's list is longer, in the utterance's HBM scratch; the exact top-`beam` of the
// list in prefix_compare order = (score key, character, element order) is taken by an MSD radix select (lists of up to
// 128 entries: by ranking every key against all others with ballots), and the survivors are compacted in list order.
//
// FULL rows (len = C) reproduce upstream's search exactly.  CLIPPED rows: without a scorer the score of child (i, k) is
// at most U(i, k) = lp[k] + score[i] (a repeated character uses log P_b <= score), and U falls along both axes when
// hypotheses are ranked by score and candidates sorted by probability: a child with (rank + 1)(k + 1) > beam has `beam`
// elements in front of it unless some of those are lowered or merged.  So row i is clipped to beam / (rank_i + 1) + margin
// candidates (~ beam ln beam entries instead of beam x C), the top-`beam` is taken on the clipped list and VERIFIED:
// the bound U(i, len(i)) of every clipped row must sort strictly behind the last key taken.  If not, the frame is redone
// with full rows.  The result is therefore the full-row result bit for bit: same survivors, same order, same node ids.
// LM: 0 no scorer, 1 character-based, 2 word-based (compile-time: the scorer's look-ups cost ~80 registers, which a
// 1 024-thread workgroup -- 128 per lane -- does not have to spare on the scorer-less path)
template <int BT, int LM, bool WIDE>
__global__ __launch_bounds__(BT, 1) void k_ctc_beam(const float* __restrict__ probs, const int32_t* __restrict__ frame_lens,
                                                  int T, BeamConfig cfg, const int32_t* __restrict__ recs,
                                                  int32_t* __restrict__ state, int init_state,
                                                  int finalize, int32_t* __restrict__ out_tokens,
                                                  int32_t* __restrict__ out_lens, double* __restrict__ out_scores,
                                                  int32_t* __restrict__ status, char* __restrict__ scratch_lists,
                                                  size_t scratch_list_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = BT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u = blockIdx.x;
  const int V = cfg.V, beam = cfg.beam, blank = cfg.blank, CM = cfg.n_cand_max;
  constexpr bool has_lm = LM != 0;
  const BeamLdsPlan plan = beam_lds_plan(beam, V, cfg.list_cap, has_lm, WIDE);
  int* hist = reinterpret_cast<int*>(smem + plan.hist);
  int* wave_tot = reinterpret_cast<int*>(smem + plan.wtot);
  float* red_p = reinterpret_cast<float*>(smem + plan.red);
  int* sh_i = reinterpret_cast<int*>(smem + plan.sh);
  // candidate arrays and kidx[] exist twice: frame t + 1 is staged (from the prefetched record) in the last phase of frame t
  constexpr int kCandWords = (has_lm ? 4 : 2) * kSmallCand;  // words of one frame's block
  int* cand_base = reinterpret_cast<int*>(smem + plan.cand);
  int* cand_c_s = cand_base;                                           // (re-pointed every frame)
  float* cand_lp_s = reinterpret_cast<float*>(cand_base) + kSmallCand;
  int* cand_word_s = cand_base + 2 * kSmallCand;                       // (scorer, narrow lists) tok2lm[character of candidate k]
  float* cand_uni_s = reinterpret_cast<float*>(cand_base + 3 * kSmallCand);  // ... uni_prob of that word (NaN: OOV / absent)
  char* pb = smem + plan.beam0;
  Beam cur = carve_beam(pb, beam);
  pb = smem + plan.beam1;
  Beam nxt = carve_beam(pb, beam);
  float* new_b = reinterpret_cast<float*>(smem + plan.newv);
  float* new_nb = new_b + beam;
  float* new_score = new_nb + beam;
  int* new_dst = reinterpret_cast<int*>(new_score + beam);  // word-based LM: dictionary state of a surviving hypothesis
  int* k_reset = new_dst + beam;                            // ... candidate whose lookup reset the state (no child), or -1
  int* rank_of = reinterpret_cast<int*>(smem + plan.rows);  // score rank of a hypothesis (accumulated with atomics)
  int* off = rank_of + beam;                                // [beam + 1] first list entry of a row (clipped frames)
  int* newpos = off + beam + 1;                             // slot of the previous frame's hypothesis e in this frame's beam, or -1
  int* surv = reinterpret_cast<int*>(smem + plan.surv);     // survivor codes in list order
  float* surv_lp = reinterpret_cast<float*>(surv + beam);   // log-probability of a surviving CHILD
  int16_t* kidx_base = reinterpret_cast<int16_t*>(smem + plan.kidx);
  const int Vp4 = (V + 3) & ~3;
  int16_t* kidx = kidx_base;  // (re-pointed every frame)
  unsigned long long* fkey = reinterpret_cast<unsigned long long*>(smem + plan.fkey);
  unsigned long long* srank_key = fkey + kSmallList;               // tiny beams: the `beam` smallest keys, by rank
  int* hyp_of_rank = reinterpret_cast<int*>(srank_key + kTinyBeam);  // ... hypothesis of score rank r
  uint32_t* lkey_s = reinterpret_cast<uint32_t*>(smem + plan.lkey);
  uint8_t* lex_s = reinterpret_cast<uint8_t*>(smem + plan.lex);            // (wide lists only)
  uint32_t* cmask = reinterpret_cast<uint32_t*>(smem + plan.cmask);        // [beam][kMaskWords] (narrow lists only)
  int16_t* first_row = reinterpret_cast<int16_t*>(smem + plan.frow);       // row of the first child entry of a thread's range
  uint32_t* lkey_g = nullptr;
  uint8_t* lex_g = nullptr;
  if (scratch_lists) {
    lkey_g = reinterpret_cast<uint32_t*>(scratch_lists + (size_t)u * scratch_list_stride);
    lex_g = reinterpret_cast<uint8_t*>(lkey_g + (size_t)beam * (1 + (size_t)CM));
  }
  constexpr bool word_lm = LM == 2;  // scorer consulted at spaces, prefixes constrained by the dictionary (lm.word_based)
  const int space_id = cfg.lm.space_id;
  float* lm_acc = reinterpret_cast<float*>(smem + plan.lmacc);  // [beam][kLmAccWords] context summaries (scorer only)
  constexpr int kChildBit = 0x40000000;  // survivor code of a child: kChildBit | row << 14 | candidate (rows < 512, candidates < 16384)

#ifdef PPASR_BEAM_POISON
  // debug builds (tools/build_variant.sh poisonX -DPPASR_BEAM_POISON=0x...): the LDS a workgroup starts with is whatever
  // the previous workgroup on that CU left -- the decode must not depend on it (tests/test_ctc_beam_gpu.py under
  // PPASR_HIP_LIB=tools/_ts/lib_poisonX.so must give the same results)
  for (uint32_t i = tid; i < plan.total / 4; i += BT) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)(PPASR_BEAM_POISON);
  __syncthreads();
#endif
  int32_t* st = state + (size_t)u * beam_state_words(beam, cfg.max_nodes);
  int32_t* g_arr = st + 2;
  int32_t* arena = st + beam_fixed_words(beam);
  const size_t tslots = beam_table_slots(cfg.max_nodes);
  unsigned long long* tkeys = reinterpret_cast<unsigned long long*>(arena + beam_arena_words(cfg.max_nodes));  // (8-byte aligned)
  int32_t* tids = reinterpret_cast<int32_t*>(tkeys + tslots);
  int nb, n_nodes;
  if (init_state) {
    nb = 1;
    n_nodes = 1;
    if (tid == 0) {
      cur.node[0] = 0; cur.chr[0] = -1; cur.par[0] = -1;
      cur.b[0] = 0.f; cur.nb[0] = kNegInf; cur.score[0] = 0.f;  // root.score = root.log_prob_b_prev = 0
      for (int j = 0; j < kLmCtx; ++j) cur.ctx[j] = cfg.lm.bos;  // Scorer::make_ngram pads with START_TOKEN
      cur.dst[0] = 0;                                             // dictionary start state
      arena[0] = -1; arena[1] = -1; arena[2] = 0;
    }
  } else {
    nb = st[0];
    n_nodes = st[1];
    for (int i = tid; i < nb; i += BT) {
      cur.node[i] = g_arr[i]; cur.chr[i] = g_arr[beam + i]; cur.par[i] = g_arr[2 * beam + i];
      cur.b[i] = __int_as_float(g_arr[3 * beam + i]); cur.nb[i] = __int_as_float(g_arr[4 * beam + i]);
      cur.score[i] = __int_as_float(g_arr[5 * beam + i]);
      for (int j = 0; j < kLmCtx; ++j) cur.ctx[i * kLmCtx + j] = g_arr[(6 + j) * beam + i];
      cur.dst[i] = g_arr[(6 + kLmCtx) * beam + i];
    }
  }
  if (WIDE) {
    for (int i = tid; i < cfg.list_cap; i += BT) lex_s[i] = 0;  // "child exists" flags: set and cleared by their setter
  } else {
    for (int i = tid; i < beam * kMaskWords; i += BT) cmask[i] = 0;  // ... bits: set in (d), cleared when the beam is rewritten
  }
  for (int i = tid; i < beam; i += BT) { rank_of[i] = 0; newpos[i] = i; }
  __syncthreads();
  // slot of every hypothesis' parent in the beam (-1: not there).  Inside the frame loop it is carried along: a hypothesis
  // written in frame t records its parent's slot in frame t's numbering, translated by newpos[] at the start of frame t + 1.
  for (int q = tid; q < nb; q += BT) {
    const int pn = cur.par[q];
    int pi = -1;
    for (int i = 0; i < nb; ++i)
      if (cur.node[i] == pn) pi = i;
    cur.pslot[q] = pi;
  }

  nb = __builtin_amdgcn_readfirstlane(nb);
  const int n_frames = frame_lens ? min(max(frame_lens[u], 0), T) : T;
  // kidx[] = index of a character in the frame's candidate list (-1: not a candidate): cleared once, then only the entries
  // of the previous frame's characters are reset.  A character's log-prob is cand_lp[kidx[c]] -- a V-wide table of its own
  // (17 KB of LDS at V = 4233) kept the workgroup from sharing a CU with the encoder's 133 KB row-block workgroups when
  // the search of step i runs beside the encoder of step i + 1 (bench.py --config cfg4 / cfg5, evaluate()).
  for (int v = tid; v < 2 * Vp4; v += BT) kidx_base[v] = -1;
  // per-frame records of the pruning pre-pass; narrow records (<= kSmallCand candidates) are fetched one frame ahead into
  // registers and staged in LDS, wide ones are read in place (HBM scratch, L2-resident while their frame is processed)
  const int RW = prune_rec_words(CM);
  const int32_t* rec_u = recs + (size_t)u * T * RW;
  constexpr int KPT = WIDE ? 1 : (kSmallCand + BT - 1) / BT;  // candidates per thread
  int pre_C = 0, pre_pb = 0, pre_c[KPT], pre_lp[KPT];
  auto fetch = [&](int t) {
    const int32_t* r = rec_u + (size_t)t * RW;
    pre_C = r[0];
    pre_pb = r[1];
    if (!WIDE) {
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const int k = tid + j * BT;
        pre_c[j] = 0;
        pre_lp[j] = 0;
        if (k < CM) { pre_c[j] = r[2 + k]; pre_lp[j] = r[2 + CM + k]; }
      }
    }
  };
  // stage the prefetched record as frame `tf` (into buffer tf & 1): candidate arrays, kidx[], the scorer's candidate side
  int C_st = 0;
  float pb_st = 0.f;
  auto stage = [&](int tf) {
    const int b = tf & 1;
    C_st = __builtin_amdgcn_readfirstlane(pre_C);  // (block-uniform: tell the compiler)
    pb_st = __int_as_float(pre_pb);
    int16_t* kx = kidx_base + b * Vp4;
    if (WIDE) {
      const int32_t* r = rec_u + (size_t)tf * RW;
      for (int k = tid; k < C_st; k += BT) kx[r[2 + k]] = (int16_t)k;
    } else {
      int* cc = cand_base + b * kCandWords;
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const int k = tid + j * BT;
        if (k < C_st) {
          cc[k] = pre_c[j];
          cc[kSmallCand + k] = pre_lp[j];
          kx[pre_c[j]] = (int16_t)k;
          if (has_lm && !word_lm) {  // the candidate side of the factorised scorer look-up: once per frame, not per pair
            const int w = cfg.lm.tok2lm[pre_c[j]];
            cc[2 * kSmallCand + k] = w;
            cc[3 * kSmallCand + k] = __float_as_int((w > 0 && w < cfg.lm.n_words) ? cfg.lm.uni_prob[w] : __builtin_nanf(""));
          }
        }
      }
    }
  };
  __syncthreads();  // (kidx cleared)
  if (n_frames > 0) {
    fetch(0);
    stage(0);
    if (n_frames > 1) fetch(1);
  }
  // sh_i: [0] merged children (tiny beams)  [1] tiny beams' flag (node-table form)  [4] clipped list length  [6 + parity]
  // tiny beams' flags  [8 + 4 parity + {1, 2, 3}] the selection's words of a frame: verification failed / kept count /
  // largest key kept.  They alternate with the frame's parity: the words of frame t are read behind the selection's last
  // barrier and the tail of the frame has no barrier before its reset -- a wave that is late behind that barrier (the CU is
  // shared with other kernels when the search overlaps the encoder) must not find them re-armed; frame t re-arms t + 1's.
  if (tid < 16) sh_i[tid] = 0;
  __syncthreads();
#ifdef PPASR_BEAM_TS
  long long ts_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ts_last = wall_clock64(), ts_n = 0, ts_c = 0, ts_nb = 0, ts_att = 0, ts_ok = 0,
            ts_small = 0, ts_hbm = 0;
#define TS(i) do { if (tid == 0 && u == 0) { long long now = wall_clock64(); ts_acc[i] += now - ts_last; ts_last = now; } } while (0)
  long long tw_last = 0, tw_acc[4] = {0, 0, 0, 0};
#define TW0() do { if (tid == BT - 64 && u == 0) tw_last = wall_clock64(); } while (0)
#define TW(i) do { if (tid == BT - 64 && u == 0) { long long now = wall_clock64(); tw_acc[i] += now - tw_last; tw_last = now; } } while (0)
#else
#define TS(i)
#define TW0()
#define TW(i)
#endif
  // rows may be clipped when there is no scorer (the bound needs score = acoustic only) and the candidate lists are sorted
  // by probability (k_ctc_prune* leave them in vocabulary order when nothing is pruned or sorted)
  const bool may_clip = !has_lm && cfg.fast_path != 0 && cfg.sorted != 0;
  const int margin = cfg.margin;
  // (e') tiny beams (<= 16, the beam-10 evaluation setting of BASELINE configs[3] / [4]): the clipped list has a STATIC
  // layout -- slot q < beam = existing hypothesis q, then row r (the hypothesis of score rank r) with its first
  // K_r = beam / (r + 1) + margin candidates -- so slot `tid` is this thread's (my_r < 0: none) in every frame: no
  // offsets, no scans.  Same verification, same result as the general form below (survivors re-ordered to list order).
  bool tiny_ok = may_clip && !WIDE && beam <= kTinyBeam && BT >= 128;
  int my_r = -1, my_k = 0, n_s0 = beam;
  if (tiny_ok) {
    int o = tid - beam;
    for (int r = 0; r < beam; ++r) {
      const int kr = beam / (r + 1) + margin;
      if (my_r < 0 && o >= 0 && o < kr) { my_r = r; my_k = o; }
      o -= kr;
      n_s0 += kr;
    }
    tiny_ok = n_s0 <= kSmallList;
  }
  const int my_row_k = beam / (lane + 1) + margin;  // K_r of row r = lane (the verification's lanes)
  for (int t = 0; t < n_frames; ++t) {
    // ---- (b, c) this frame's pruned characters (get_pruned_log_probs, done by the pre-pass; staged in the last phase of
    // the previous frame) ----
    const int C = C_st;
    const float p_blank = pb_st;
    const int32_t* rec_t = rec_u + (size_t)t * RW;
    int* fl = sh_i + 8 + 4 * (t & 1);        // this frame's selection words ([1] failed, [2] kept, [3] largest key kept)
    int* fl_next = sh_i + 8 + 4 * (~t & 1);  // the next frame's: re-armed by this frame's tail
    kidx = kidx_base + (t & 1) * Vp4;
    cand_c_s = cand_base + (t & 1) * kCandWords;
    cand_lp_s = reinterpret_cast<float*>(cand_c_s) + kSmallCand;
    cand_word_s = cand_c_s + 2 * kSmallCand;
    cand_uni_s = reinterpret_cast<float*>(cand_c_s + 3 * kSmallCand);
    // candidate k of the frame: character, log-probability
    auto cand_c = [&](int k) -> int { return WIDE ? rec_t[2 + k] : cand_c_s[k]; };
    auto cand_lp = [&](int k) -> float { return WIDE ? __int_as_float(rec_t[2 + CM + k]) : cand_lp_s[k]; };
    auto lp_of = [&](int c) -> float {
      const int k = kidx[c];
      return k >= 0 ? cand_lp(k) : kNotCand;
    };
    TS(0);
    // ---- external scorer: pruning threshold of this frame and the LM term of every possible extension ----
    // (ctc_beam_search_decoder.cpp: prefixes sorted, min_cutoff = worst score + log(p_blank) - max(0, beta), and the
    //  `break` on log_prob_c + prefix->score < min_cutoff once the beam is full: it skips the blank, repeat and
    //  extension updates of that (character, prefix) pair; sorted order makes `break` == "skip every failing pair")
    float min_cutoff = kNegInf;
    bool full_beam = false;
    if (has_lm) {
      float m = FLT_MAX;
      for (int q = tid; q < nb; q += BT) m = fminf(m, cur.score[q]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
      if (lane == 0) red_p[wave] = m;
      // the back-off side of every look-up of this frame depends on the hypothesis only (lm.h, the factorised form):
      // summarised once per hypothesis, its order - 1 probes in flight together
      if (tid < nb) {
        float acc[kLmAccWords];
        lm_context_acc(cfg.lm, &cur.ctx[tid * kLmCtx + (kLmCtx - (cfg.lm.order - 1))], acc);
#pragma unroll
        for (int j = 0; j < kLmAccWords; ++j) lm_acc[tid * kLmAccWords + j] = acc[j];
      }
      lds_barrier();
      m = red_p[0];
      for (int w = 1; w < NW; ++w) m = fminf(m, red_p[w]);
      min_cutoff = (float)((double)m + log((double)p_blank) - fmax(0.0, cfg.beta));
      full_beam = (nb == beam);
    }
    auto pruned = [&](float lp_c, int q) -> bool { return full_beam && (lp_c + cur.score[q] < min_cutoff); };
    // alpha * ln P_lm(c | last order-1 words of hypothesis i): the scorer term of the extension (i, c)
    // word-based scorer: alpha * ln P_lm(word | last order-1 WORDS of hypothesis i), `word` = LM index of the word a space
    // has just completed (ctc_beam_search_decoder.cpp scores `prefix`, not `prefix_new`, when c == space_id)
    auto lm_term_word = [&](int i, int word) -> float {
      int32_t ctx[kLmMaxOrder];
      float acc[kLmAccWords];
      const int order = cfg.lm.order;
#pragma unroll
      for (int j = 0; j < kLmCtx; ++j) ctx[j] = j < order - 1 ? cur.ctx[i * kLmCtx + (kLmCtx - (order - 1)) + j] : 0;
#pragma unroll
      for (int j = 0; j < kLmAccWords; ++j) acc[j] = lm_acc[i * kLmAccWords + j];
      return (float)(lm_pair_log_cond_prob(cfg.lm, ctx, acc, word) * cfg.alpha);
    };
    auto lm_term = [&](int i, int c) -> float { return lm_term_word(i, cfg.lm.tok2lm[c]); };
    // character-based scorer, candidate k of a narrow list: word id and unigram from the frame's LDS tables
    auto lm_term_k = [&](int i, int k) -> float {
      if (WIDE) return lm_term(i, cand_c(k));
      int32_t ctx[kLmMaxOrder];
      float acc[kLmAccWords];
      const int order = cfg.lm.order;
#pragma unroll
      for (int j = 0; j < kLmCtx; ++j) ctx[j] = j < order - 1 ? cur.ctx[i * kLmCtx + (kLmCtx - (order - 1)) + j] : 0;
#pragma unroll
      for (int j = 0; j < kLmAccWords; ++j) acc[j] = lm_acc[i * kLmAccWords + j];
      return (float)(lm_pair_log_cond_prob(cfg.lm, ctx, acc, cand_word_s[k], cand_uni_s[k]) * cfg.alpha);
    };
    // log-probability carried by the extension of hypothesis i with candidate k (its LM term included); `to_state`:
    // dictionary state the extension lands in (word-based scorer only)
    auto ext_logp = [&](int i, int k, int to_state) -> float {
      const int c = cand_c(k);
      const float lpk = cand_lp(k);
      float log_p = kNegInf;
      if (c == cur.chr[i]) { if (cur.b[i] > kNegInf) log_p = lpk + cur.b[i]; }
      else log_p = lpk + cur.score[i];
      if (word_lm) {
        if (c == space_id) {
          log_p += lm_term_word(i, cfg.lm.dict_word[to_state]);
          log_p = (float)((double)log_p + cfg.beta);
        }
      } else if (has_lm) {
        log_p += lm_term_k(i, k);
        log_p = (float)((double)log_p + cfg.beta);
      }
      return log_p;
    };
    TS(1);
    TW0();
    // rows are clipped this frame if the shortest staircase row is shorter than the candidate list (block-uniform)
    const bool tiny = tiny_ok && C > 0;
    bool clip = may_clip && !tiny && C > beam / nb + margin;
    // ---- (d) contributions received by the hypotheses already in the beam (thread q < nb <= BT) ----
    const float lpb = lp_of(blank);
    int mrg_pi = -1, mrg_k = 0;  // the child (parent slot, candidate) this hypothesis IS: it exists, no new element for it
    if (tid < nb) {
      const int q = tid;
      const int cq = cur.chr[q];
      float bc = (lpb != kNotCand && !pruned(lpb, q)) ? lpb + cur.score[q] : kNegInf;
      float nbc = kNegInf;
      const float lq = (cq >= 0) ? lp_of(cq) : kNotCand;
      // the parent's slot: recorded in the previous frame's numbering, translated once (and kept for the hypothesis' copy)
      const int raw = cur.pslot[q];
      const int pi_t = raw >= 0 ? newpos[raw] : -1;
      cur.pslot[q] = pi_t;
      if (lq != kNotCand && cq != blank) {
        if (!pruned(lq, q)) nbc = lq + cur.nb[q];  // repeated character
        int pi = pi_t;
        if (cfg.node_table) {  // (a revived prefix takes its old node id back: found by id, not by slot)
          const int pn = cur.par[q];
          for (int i = 0; i < nb; ++i)
            if (cur.node[i] == pn) pi = i;
        }
        if (pi >= 0) {  // extension of the parent hypothesis by cq lands on this existing prefix
          // (word-based scorer: the word a space completes is read off the PARENT's dictionary state -- this prefix's own
          //  state may already have been reset to the start state by a failed look-up, see below)
          const int kq = kidx[cq];
          if (!pruned(lq, pi))
            nbc = lse(nbc, ext_logp(pi, kq, (word_lm && cq == space_id) ? lm_dict_arc(cfg.lm, cur.dst[pi], cq) : 0));
          mrg_pi = pi;
          mrg_k = kq;
          if (!WIDE) atomicOr(&cmask[pi * kMaskWords + (kq >> 5)], 1u << (kq & 31));
        }
      }
      new_b[q] = bc;
      new_nb[q] = nbc;
      new_score[q] = lse(bc, nbc);
    }
    if (tiny && wave == 0) {  // (nb <= 16: (d) ran in lanes of wave 0) merged children of the frame
      const unsigned long long m = __ballot(tid < nb && mrg_pi >= 0);
      if (lane == 0) sh_i[0] = __popcll(m);
    }
    if (tiny && wave == 1 && lane < nb) {  // an otherwise idle wave: rank of every hypothesis by its CURRENT score
      const float sq = cur.score[lane];
      int r = 0;
      for (int i = 0; i < nb; ++i) {
        const float si = cur.score[i];
        r += (si > sq || (si == sq && i < lane)) ? 1 : 0;
      }
      hyp_of_rank[r] = lane;
    }
    TS(8);
    TW(0);
    // ---- clipped rows: rank of every hypothesis by its CURRENT score (the rows of the staircase), row lengths, offsets ----
    int NLc_clip = 0;  // child entries of the clipped list
    // a / per for 0 <= a < 2^22 without the integer-division sequence (two of them sat on every row thread's critical path)
    auto div_per = [&](int a, int per) -> int {
      int q = (int)((float)a * __builtin_amdgcn_rcpf((float)per));
      if (q * per > a) --q;
      if ((q + 1) * per <= a) ++q;
      return q;
    };
    auto write_first_rows = [&](int i, int my_off, int my_len, int per) {
      // first_row[t] = row holding the first child entry of thread t's range [t*per, (t+1)*per) of the list (threads whose
      // range ends inside the hypotheses need none): row i covers child entries [my_off, my_off + my_len)
      if (my_len <= 0) return;
      const int t0 = div_per(nb, per);  // first thread whose range reaches the children; its first child entry is 0
      int tt = div_per(nb + my_off + per - 1, per);
      if (tt < t0 || my_off == 0) tt = t0;
      for (; tt < BT && max(0, tt * per - nb) < my_off + my_len; ++tt) first_row[tt] = (int16_t)i;
    };
    // entries per thread of the radix path (an ODD range length: thread t starts at word t*per of the list, and an even
    // stride would put the 64 lanes of a wave on 16 or fewer of the 64 LDS banks); 1 on the short-list path
    auto per_of = [&](int NL) -> int {
      return (NL <= kSmallList && cfg.fast_path != 0 && NL <= cfg.list_cap) ? 1 : (((NL + BT - 1) / BT) | 1);
    };
    if (clip && nb <= 64) {
      // up to 64 hypotheses: ONE wave (the last: (d) runs in the first) does the whole computation in registers -- no
      // atomics, no scan across waves, no barrier of its own (the results are read behind the barrier that closes (d))
      if (wave == NW - 1) {
      const float sq = lane < nb ? cur.score[lane] : 0.f;
      int r = 0;
      for (int i = 0; i < nb; ++i) {
        const float si = rl_f(sq, i);
        r += (si > sq || (si == sq && i < lane)) ? 1 : 0;
      }
      TW(1);
      const int my_len = lane < nb ? min(C, beam / (r + 1) + margin) : 0;
      const int incl = wave_incl_scan(my_len);
      NLc_clip = __builtin_amdgcn_readlane(incl, 63);
      const int my_off = incl - my_len;
      if (lane < nb) off[lane] = my_off;
      if (lane == 0) { off[nb] = NLc_clip; sh_i[4] = NLc_clip; }
      write_first_rows(lane, my_off, my_len, per_of(nb + NLc_clip));
      }
    } else if (clip) {
      // more than 64 hypotheses: exact ranks cost nb^2 comparisons (2.6 us of VALU time at nb = 300); the staircase only
      // needs a rank that is NOT ABOVE the true one (a longer row is still verified), so the scores are bucketed -- 256 equal
      // buckets between the best and the worst score of the beam (the top-`beam` of a frame sit within a fraction of a nat
      // of one another on flat posteriors, tens of nats apart on peaked ones) -- and a hypothesis takes the number of
      // hypotheses in strictly better buckets as its rank.  This phase: best and worst score per wave.
      const float sc = tid < nb ? cur.score[tid] : kNegInf;
      float m = sc, mn = (tid < nb && sc > -1e30f) ? sc : FLT_MAX;  // (a prefix of probability zero is no lower end)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        m = fmaxf(m, __shfl_xor(m, o));
        mn = fminf(mn, __shfl_xor(mn, o));
      }
      if (lane == 0) { red_p[wave] = m; red_p[16 + wave] = mn; }
      if (tid < 256) hist[tid] = 0;
    }
    TS(9);
    TW(2);
    lds_barrier();
    TS(2);
    TW(3);
    if (tid < beam) newpos[tid] = -1;  // consumed in (d); takes the new slots of the hypotheses that stay (end of the frame)
    if (clip && nb <= 64) NLc_clip = sh_i[4];
    // one slot of the next beam: survivor `code` (an existing hypothesis e, or kChildBit | row << 14 | candidate) at slot `pos`;
    // log_p: the extension's log-probability (children); id / fresh: node id of a child and whether the node is new
    auto mat_slot = [&](int pos, int code, float log_p, int id, bool fresh) {
      if (!(code & kChildBit)) {
        const int e = code;
        nxt.node[pos] = cur.node[e]; nxt.chr[pos] = cur.chr[e]; nxt.par[pos] = cur.par[e];
        nxt.b[pos] = new_b[e]; nxt.nb[pos] = new_nb[e]; nxt.score[pos] = new_score[e];
        for (int j = 0; j < kLmCtx; ++j) nxt.ctx[pos * kLmCtx + j] = cur.ctx[e * kLmCtx + j];
        nxt.dst[pos] = word_lm ? new_dst[e] : 0;
        nxt.pslot[pos] = cur.pslot[e];  // the parent's slot in THIS frame's beam (translated in (d))
        newpos[e] = pos;
      } else {
        const int i = (code >> 14) & 0xFFFF, kk = code & 0x3FFF;
        const int c = cand_c(kk);
        if (!cfg.node_table && id < cfg.max_nodes) { arena[kArenaWords * (size_t)id] = cur.node[i]; arena[kArenaWords * (size_t)id + 1] = c; }
        if (word_lm) {
          // the LM context holds WORDS: it moves on when a space completes one.  The dictionary state belongs to the trie
          // node: a new node starts where the arc leads, a revived one is where it was left (a final state may have been
          // reset to the start state by a failed look-up while the prefix was alive)
          const int to = lm_dict_arc(cfg.lm, new_dst[i], c);
          if (fresh) {
            nxt.dst[pos] = to;
            if (id < cfg.max_nodes) arena[kArenaWords * (size_t)id + 2] = to;
          } else {
            nxt.dst[pos] = arena[kArenaWords * (size_t)id + 2];
          }
          if (c == space_id) {
            for (int j = 0; j + 1 < kLmCtx; ++j) nxt.ctx[pos * kLmCtx + j] = cur.ctx[i * kLmCtx + j + 1];
            nxt.ctx[pos * kLmCtx + kLmCtx - 1] = cfg.lm.dict_word[to];
          } else {
            for (int j = 0; j < kLmCtx; ++j) nxt.ctx[pos * kLmCtx + j] = cur.ctx[i * kLmCtx + j];
          }
        } else {
          nxt.dst[pos] = 0;
          for (int j = 0; j + 1 < kLmCtx; ++j) nxt.ctx[pos * kLmCtx + j] = cur.ctx[i * kLmCtx + j + 1];
          nxt.ctx[pos * kLmCtx + kLmCtx - 1] = has_lm ? cfg.lm.tok2lm[c] : 0;
        }
        nxt.node[pos] = id; nxt.chr[pos] = c; nxt.par[pos] = cur.node[i];
        nxt.b[pos] = kNegInf; nxt.nb[pos] = log_p; nxt.score[pos] = log_p;
        nxt.pslot[pos] = i;  // the parent's slot in THIS frame's beam
      }
    };
    // last phase of a frame: this frame's "child exists" bits and kidx[] entries are cleared (the buffers are clean again for
    // frame t + 2), the next frame's candidates staged into the other buffer, its successor's record requested
    bool staged_next = false;
    auto end_of_frame = [&]() {
      if (!WIDE && tid < beam) {
#pragma unroll
        for (int j = 0; j < kMaskWords; ++j) cmask[tid * kMaskWords + j] = 0;
      }
      for (int k = tid; k < C; k += BT) kidx[cand_c(k)] = -1;
      if (!staged_next && t + 1 < n_frames) {
        stage(t + 1);
        if (t + 2 < n_frames) fetch(t + 2);
      }
      staged_next = true;
      if (tid == 0) { fl_next[1] = 0; fl_next[2] = 0; fl_next[3] = 0; }
    };
    int k_sel = 0;
    bool tiny_done = false;
    if (tiny) {
      const int has_blank = kidx[blank] >= 0 ? 1 : 0;
      const int n_valid_f = nb + nb * (C - has_blank) - sh_i[0];  // existing + non-blank, non-merged children
      if (n_valid_f > beam) {  // (block-uniform)
#ifdef PPASR_BEAM_TS
        ++ts_att;
#endif
        // keys: score key | character + 1 | id, id = e for hypothesis e, 1 << 17 | i << 7 | k for child (i, k): the order of
        // the general list (hypotheses, then children by (row, candidate))
        if (tid < n_s0) {
          unsigned long long key = ~0ull;
          if (tid < beam) {
            if (tid < nb) key = make_key(new_score[tid], cur.chr[tid], tid);
          } else if (my_r < nb && my_k < C) {
            const int i = hyp_of_rank[my_r], k = my_k;
            const int c = cand_c_s[k];
            if (c != blank && !((cmask[i * kMaskWords + (k >> 5)] >> (k & 31)) & 1u)) {
              const float lpk = cand_lp_s[k];
              float log_p = kNegInf;
              if (c == cur.chr[i]) { if (cur.b[i] > kNegInf) log_p = lpk + cur.b[i]; }
              else log_p = lpk + cur.score[i];
              key = make_key(log_p, c, (1 << 17) | (i << 7) | k);
            }
          }
          fkey[tid] = key;
        }
        if (tid < kTinyBeam) srank_key[tid] = ~0ull;
        lds_barrier();
        TS(3);
        {  // rank of every key = number of smaller keys (keys are unique: the id is part of them): wave w ranks the keys
           // w, w + NW, ...
          const unsigned long long k0 = lane < n_s0 ? fkey[lane] : ~0ull, k1 = lane + 64 < n_s0 ? fkey[lane + 64] : ~0ull;
          if (n_s0 <= 64) {  // (beam 10: 57 slots -- one key per lane, one ballot per ranked key)
            for (int i = wave; i < n_s0; i += NW) {
              const unsigned long long ki = fkey[i];
              const int rank = __popcll(__ballot(k0 < ki));
              if (lane == 0 && ki != ~0ull && rank < beam) srank_key[rank] = ki;
            }
          } else {
            for (int i = wave; i < n_s0; i += NW) {
              const unsigned long long ki = fkey[i];
              const int rank = __popcll(__ballot(k0 < ki)) + __popcll(__ballot(k1 < ki));
              if (lane == 0 && ki != ~0ull && rank < beam) srank_key[rank] = ki;
            }
          }
        }
        lds_barrier();
        TS(5);
        // survivors in LIST order (the order the general compaction leaves them in), written straight into the next beam, +
        // the verification, + the end-of-frame work: ONE phase.  (If the verification fails -- rare -- the slots written here
        // are overwritten by the general form below, which first rebuilds what the end-of-frame work cleared.)
        const bool merged_tail = !cfg.node_table;
        // (merged tail: the flag word alternates with the frame's parity -- this frame's is read behind the phase's only
        //  barrier while the other one is re-armed for the next frame)
        int* vflag = sh_i + (merged_tail ? 6 + (t & 1) : 1);
        if (merged_tail && tid == 0) { sh_i[6 + ((t + 1) & 1)] = 0; fl_next[1] = 0; fl_next[2] = 0; fl_next[3] = 0; }
        if (wave == 0) {  // lane = (survivor p = lane & 15, quarter g = lane >> 4 of the others it is compared with)
          const int pidx = lane & 15, g = lane >> 4;
          const unsigned long long kp = pidx < beam ? srank_key[pidx] : ~0ull;
          const int ep = (int)(kp & 0x3FFFFull);  // (make_key: the id is the low 18 bits)
          int pos = 0;
#pragma unroll
          for (int j = 0; j < kTinyBeam / 4; ++j) {
            const int jj = g + 4 * j;
            const unsigned long long kj = jj < beam ? srank_key[jj] : ~0ull;
            pos += (kj != ~0ull && (int)(kj & 0x3FFFFull) < ep) ? 1 : 0;
          }
          pos += __shfl_xor(pos, 16);
          pos += __shfl_xor(pos, 32);
          if (lane < beam) {
            if (kp == ~0ull) {
              *vflag = 1;  // fewer than `beam` valid elements in the list
            } else {
              const int code = (ep >> 17) ? (kChildBit | (((ep >> 7) & 0xF) << 14) | (ep & 0x7F)) : ep;
              const float lp = score_of_key((uint32_t)(kp >> 32));
              if (merged_tail) {
                mat_slot(pos, code, lp, n_nodes + pos, true);
              } else {
                surv[pos] = code;
                surv_lp[pos] = lp;
              }
            }
          }
        } else if (wave == 1 && lane < nb) {  // (another wave: row r = lane)
          const unsigned long long kl = srank_key[beam - 1];
          if (my_row_k < C && kl != ~0ull) {
            const uint32_t thr = (uint32_t)(kl >> 32);  // score key of the last element taken
            const float ub = cand_lp_s[my_row_k] + cur.score[hyp_of_rank[lane]];
            if (desc_key(ub) <= thr) *vflag = 1;  // an excluded child could score >= the last one taken
          }
        }
        if (merged_tail && wave >= 2) {  // (waves 0 and 1 are busy above; kidx / cmask are not read in this phase)
          if (!WIDE && tid - 128 < beam) {
#pragma unroll
            for (int j = 0; j < kMaskWords; ++j) cmask[(tid - 128) * kMaskWords + j] = 0;
          }
          for (int k = tid - 128; k < C; k += BT - 128) kidx[cand_c(k)] = -1;
        }
        int fail;
        if (merged_tail) {
          // staging needs every thread's prefetched registers: all waves, after their part above
          if (t + 1 < n_frames) {
            stage(t + 1);
            if (t + 2 < n_frames) fetch(t + 2);
          }
          staged_next = true;
          lds_barrier();
          fail = *vflag;
        } else {
          lds_barrier();
          fail = sh_i[1];
        }
        if (fail == 0) {
          tiny_done = true;
          k_sel = beam;
#ifdef PPASR_BEAM_TS
          ++ts_ok;
          ++ts_small;
#endif
        } else {
          if (merged_tail) {
            // undo what the merged phase cleared / wrote: the frame's kidx[] and "child exists" bits, the slots map
            for (int k = tid; k < C; k += BT) kidx[cand_c(k)] = (int16_t)k;
            if (tid < beam) newpos[tid] = -1;
            lds_barrier();  // (cmask was cleared by other threads than those that set its bits)
            if (tid < nb && mrg_pi >= 0) atomicOr(&cmask[mrg_pi * kMaskWords + (mrg_k >> 5)], 1u << (mrg_k & 31));
          } else {
            lds_barrier();  // (everyone has read the flag)
            if (tid == 0) sh_i[1] = 0;
          }
          lds_barrier();
        }
        TS(6);
      }
    }
    if (!tiny_done)
    for (;;) {  // (at most two rounds: clipped rows, then -- if the verification fails -- full rows)
      if (clip && nb > 64) {
        float m = red_p[0], mn = red_p[16];
        for (int w = 1; w < NW; ++w) {
          m = fmaxf(m, red_p[w]);
          mn = fminf(mn, red_p[16 + w]);
        }
        const float bscale = (m > mn) ? 255.f / (m - mn) : 0.f;
        int bkt = 0;
        if (tid < nb) {
          const float d = (m - cur.score[tid]) * bscale;
          bkt = !(d < 255.f) ? 255 : (int)d;
          atomicAdd(&hist[bkt], 1);
        }
        lds_barrier();
        TS(10);
        int my_len = 0;
        {
          // exclusive prefix of the 256 buckets, computed by every wave for itself (lane l: buckets 4 l .. 4 l + 3)
          int c[4], sum = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            c[j] = hist[4 * lane + j];
            sum += c[j];
          }
          const int excl = wave_incl_scan(sum) - sum;
          const int src = (bkt >> 2) * 4;  // (byte address of the lane for ds_bpermute)
          int r = __builtin_amdgcn_ds_bpermute(src, excl);
          const int c0 = __builtin_amdgcn_ds_bpermute(src, c[0]), c1 = __builtin_amdgcn_ds_bpermute(src, c[1]),
                    c2 = __builtin_amdgcn_ds_bpermute(src, c[2]);
          const int j = bkt & 3;
          r += (j > 0 ? c0 : 0) + (j > 1 ? c1 : 0) + (j > 2 ? c2 : 0);
          if (tid < nb) my_len = min(C, beam / (r + 1) + margin);
        }
        const int my_off = block_excl_scan<NW>(my_len, wave_tot, NLc_clip);
        TS(11);
        if (tid < nb) {
          off[tid] = my_off;
          write_first_rows(tid, my_off, my_len, per_of(nb + NLc_clip));
        }
        if (tid == 0) off[nb] = NLc_clip;
        lds_barrier();
        TS(12);
      }
      const int NLc = clip ? NLc_clip : nb * C;
      const int NL = nb + NLc;
      const bool in_lds = NL <= cfg.list_cap;
#ifdef PPASR_BEAM_TS
      if (clip) ++ts_att;
      if (!in_lds) ++ts_hbm;
      ts_n += NL; ts_c += C; ts_nb += nb;
#endif
      if (!in_lds && !lkey_g) {  // (the host refuses configurations that can get here without scratch; defensive)
        if (tid == 0 && status) status[u] = 2;
        clip = false;
        k_sel = -1;
        break;
      }
      // -------- the selection on one list; instantiated for the LDS list and for the HBM list --------
      auto select = [&](auto* lkey, uint8_t* lex, auto lds_tag) __attribute__((always_inline)) -> bool {
        constexpr bool kLds = decltype(lds_tag)::value;
        auto list_barrier = [&]() { if (kLds) lds_barrier(); else __syncthreads(); };
        auto row_off = [&](int i) -> int { return clip ? off[i] : i * C; };
        auto row_len = [&](int i) -> int { return clip ? off[i + 1] - off[i] : C; };
        auto row_of = [&](int s) -> int {  // row holding child entry s (any s: binary search over the offsets)
          if (!clip) return s / C;
          int lo = 0, hi = nb - 1;  // largest i with off[i] <= s
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (off[mid] <= s) lo = mid;
            else hi = mid - 1;
          }
          return lo;
        };
        // "the child (i, k) is already a hypothesis of the beam": narrow lists keep one bit per (hypothesis, candidate), set
        // in (d); wide lists one flag per list entry, set here by the hypothesis that IS the child and cleared by it below
        int my_flag = -1;
        if (WIDE) {
          if (tid < nb && mrg_pi >= 0 && mrg_k < row_len(mrg_pi)) {
            my_flag = row_off(mrg_pi) + mrg_k;
            lex[my_flag] = 1;
          }
          list_barrier();
        }
        auto child_exists = [&](int i, int k, int s) -> bool {
          if (WIDE) return lex[s] != 0;
          return ((cmask[i * kMaskWords + (k >> 5)] >> (k & 31)) & 1u) != 0;
        };
        if (word_lm) {
          // PathTrie::get_path_trie with a dictionary: a character that has no arc from the prefix's dictionary state yields
          // no child -- and when that state is FINAL (a word has just ended) the lookup resets the prefix's state to the
          // start state as a side effect.  Upstream walks the candidates in list order for every prefix, so for a prefix in
          // a final state the FIRST candidate that is looked up (not blank, not cut by min_cutoff, not an existing child)
          // finds nothing and resets the state; every later candidate of the frame is looked up from the start state.
          // (scorer present: rows are never clipped, this runs once per frame)
          if (tid < nb) {
            const int q = tid;
            int kr = -1, nd = cur.dst[q];
            if (lm_dict_final(cfg.lm, nd)) {
              const int ro = row_off(q);
              for (int k = 0; k < C; ++k) {
                if (cand_c(k) == blank || child_exists(q, k, ro + k) || pruned(cand_lp(k), q)) continue;
                kr = k;
                break;
              }
              if (kr >= 0) {
                nd = 0;
                if (cur.node[q] < cfg.max_nodes) arena[kArenaWords * (size_t)cur.node[q] + 2] = 0;  // (the node's own state)
              }
            }
            k_reset[q] = kr;
            new_dst[q] = nd;
          }
          lds_barrier();
        }
        // key of child (i, k) at list entry nb + s; kNoKey: not an element (blank, existing child, cut by the scorer's
        // min_cutoff, no dictionary arc)
        constexpr uint32_t kNoKey = 0xFFFFFFFFu;
        struct RowCtx { int ci; float bi, si; int di, kri; bool dead; };
        auto load_row = [&](int i) -> RowCtx {
          RowCtx r;
          r.ci = cur.chr[i]; r.bi = cur.b[i]; r.si = cur.score[i];
          // word-based scorer: dictionary state the children of i are looked up from (after the reset above), and the one
          // candidate that triggered the reset
          r.di = word_lm ? new_dst[i] : 0;
          r.kri = word_lm ? k_reset[i] : -1;
          r.dead = word_lm && lm_dict_final(cfg.lm, r.di);  // still final: no candidate was looked up this frame
          return r;
        };
        auto child_key = [&](int i, const RowCtx& r, int k, int s) -> uint32_t {
          const int c = cand_c(k);
          const float lpk = cand_lp(k);
          if (c == blank || child_exists(i, k, s) || (full_beam && (lpk + r.si < min_cutoff))) return kNoKey;
          int to = 0;
          if (word_lm) {
            to = (r.dead || k == r.kri) ? -1 : lm_dict_arc(cfg.lm, r.di, c);
            if (to < 0) return kNoKey;
          }
          float log_p = kNegInf;
          if (c == r.ci) { if (r.bi > kNegInf) log_p = lpk + r.bi; }
          else log_p = lpk + r.si;
          if (word_lm) {
            if (c == space_id) {
              log_p += lm_term_word(i, cfg.lm.dict_word[to]);
              log_p = (float)((double)log_p + cfg.beta);
            }
          } else if (has_lm) {
            log_p += lm_term_k(i, k);
            log_p = (float)((double)log_p + cfg.beta);
          }
          return desc_key(log_p);
        };
        bool small_done = false;
        if (kLds && NL <= kSmallList && cfg.fast_path != 0) {
          // ---- (e, f, g) short lists: one entry per thread, 64-bit keys (score key | character | entry) ranked against
          // all others with ballots -- no histograms, no scans ----
          small_done = true;
#ifdef PPASR_BEAM_TS
          ++ts_small;
#endif
          if (tid < NL) {
            unsigned long long key = ~0ull;
            if (tid < nb) {
              key = make_key(new_score[tid], cur.chr[tid], tid);
            } else {
              const int s = tid - nb, i = clip ? (int)first_row[tid] : s / C, k = s - row_off(i);
              const RowCtx r = load_row(i);
              const uint32_t k32 = child_key(i, r, k, s);
              if (k32 != kNoKey) key = ((unsigned long long)k32 << 32) | ((unsigned long long)(uint32_t)(cand_c(k) + 1) << 18) | (unsigned long long)(uint32_t)tid;
            }
            fkey[tid] = key;
          }
          lds_barrier();
          TS(3);
          {
            const unsigned long long k0 = lane < NL ? fkey[lane] : ~0ull, k1 = lane + 64 < NL ? fkey[lane + 64] : ~0ull;
            for (int i = wave; i < NL; i += NW) {
              const unsigned long long ki = fkey[i];
              const int rank = __popcll(__ballot(k0 < ki)) + __popcll(__ballot(k1 < ki));
              if (lane == 0) {
                const bool kept = ki != ~0ull && rank < beam;
                lkey[i] = kept ? 1u : 0u;
                if (kept) atomicMax(reinterpret_cast<unsigned int*>(&fl[3]), (unsigned int)(ki >> 32));
              }
            }
          }
          lds_barrier();
          TS(5);
          if (wave == 0) {  // survivors in list order
            const bool f0 = lane < NL && lkey[lane] != 0, f1 = lane + 64 < NL && lkey[lane + 64] != 0;
            const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
            auto put = [&](int e, int pos) {
              const unsigned long long ke = fkey[e];
              if (e < nb) {
                surv[pos] = e;
                surv_lp[pos] = 0.f;
              } else {
                const int s = e - nb, i = clip ? (int)first_row[e] : s / C, k = s - row_off(i);
                surv[pos] = kChildBit | (i << 14) | k;
                surv_lp[pos] = score_of_key((uint32_t)(ke >> 32));
              }
            };
            if (f0) put(lane, mbcnt(m0));
            if (f1) put(lane + 64, __popcll(m0) + mbcnt(m1));
            const int n_kept = __popcll(m0) + __popcll(m1);
            if (lane == 0) fl[2] = n_kept;
            if (clip && lane < nb) {  // verification (clipped rows: nb <= 64): the bound of the best excluded child of row `lane`
              const int K = row_len(lane);
              if (K < C) {
                const float ub = cand_lp(K) + cur.score[lane];
                // fewer than `beam` elements in a clipped list, or a bound that reaches the last key taken: redo with full rows
                if (n_kept < beam || desc_key(ub) <= (uint32_t)fl[3]) fl[1] = 1;
              }
            }
          }
        }
        if (!small_done) {
          // ---- (e) the 32-bit score key of every entry, computed ONCE into the list (kNoKey = not an element); thread t
          // owns the contiguous range [t*per, (t+1)*per) so that the compaction below keeps list order with a single
          // block scan.  (An ODD range length: thread t starts at word t*per, and an even stride would put the 64 lanes
          // of a wave on 16 or fewer of the 64 LDS banks.) ----
          const int per = per_of(NL);
          const int e_lo = min(tid * per, NL), e_hi = min(e_lo + per, NL);
          for (int i = tid; i < 8 * 256; i += BT) hist[i] = 0;  // 4 + 4 per-pass histograms of the selects below (complete
                                                                // behind the barrier of the scan that closes (e))
          const int row0 = e_hi > nb ? (clip ? (int)first_row[tid] : max(0, e_lo - nb) / C) : 0;  // row of my first child entry
          int my_valid = 0;
          {
            int e = e_lo;
            for (; e < e_hi && e < nb; ++e) {  // hypotheses already in the beam
              lkey[e] = desc_key(new_score[e]);
              ++my_valid;
            }
            if (e < e_hi) {  // children: one row at a time (the hypothesis' fields stay in registers)
              int s = e - nb, i = row0;
              while (e < e_hi) {
                const int ro = row_off(i), k0 = s - ro;
                const int k1 = min(row_len(i), k0 + (e_hi - e));
                const RowCtx r = load_row(i);
                for (int k = k0; k < k1; ++k, ++e, ++s) {
                  const uint32_t key = child_key(i, r, k, s);
                  my_valid += key != kNoKey ? 1 : 0;
                  lkey[e] = key;
                }
                ++i;
              }
            }
          }
          TS(13);
          int n_valid;
          (void)block_excl_scan<NW>(my_valid, wave_tot + NW, n_valid);
          if (!kLds) __syncthreads();  // (the scan's barrier orders LDS only: the list lives in global memory here)
          const int ksel = n_valid >= beam ? beam : n_valid;
          TS(3);
          // ---- (f) exact top-ksel in prefix_compare order = ascending (score key, char, entry): MSD radix select of
          // the ksel-th smallest 32-bit score key over the list; if the threshold class has more members than slots
          // left (ties), further selects over the character and the entry index inside that class ----
          // 4-pass radix select of the k-th smallest value of f(e) over elements with pred(e); returns the value and how
          // many members of its class are needed (k_need) / exist (k_have).  `fresh`: the 4 histograms are zero (frame start)
          auto radix_select32 = [&](auto&& value_of, int k, int* hists, bool fresh, uint32_t& out, int& k_need, int& k_have) {
            if (!fresh) {
              lds_barrier();
              for (int i = tid; i < 4 * 256; i += BT) hists[i] = 0;
              lds_barrier();
            }
            uint32_t prefix = 0;
            int k_rem = k;
            k_have = 0;
            for (int pass = 0; pass < 4; ++pass) {
              const int shift = 24 - 8 * pass;
              const uint32_t hi_mask = pass == 0 ? 0u : (~0u << (shift + 8));
              int* h = hists + pass * 256;  // one histogram per pass = one barrier per pass
              {
                RunHist rh(h);
                for (int e = tid; e < NL; e += BT) {
                  uint32_t v;
                  if (value_of(e, v) && (v & hi_mask) == prefix) rh.add((int)((v >> shift) & 0xff));
                }
                rh.flush_wave();
              }
              lds_barrier();
              int bin;
              select_bin_reg(h, k_rem, bin, k_rem, k_have);
              prefix |= (uint32_t)bin << shift;
              if (k_have == k_rem && shift > 0) {  // the searched value is the last of its class: take the whole class
                prefix |= (1u << shift) - 1u;
                break;
              }
            }
            out = prefix;
            k_need = k_rem;
          };
          auto entry_char1 = [&](int e) -> uint32_t {  // character + 1 of a list entry
            if (e < nb) return (uint32_t)(cur.chr[e] + 1);
            const int s = e - nb, i = row_of(s);
            return (uint32_t)(cand_c(s - row_off(i)) + 1);
          };
          // keep: key < thr1, or key == thr1 (exact_class) and (char + 1 < thr2, or char + 1 == thr2 and entry <= thr3)
          uint32_t thr1 = 0xFFFFFFFEu, thr2 = 0xFFFFFFFFu, thr3 = 0xFFFFFFFFu;
          bool exact_class = false, exact_char = false;
          if (ksel < n_valid) {
            int need, have;
            radix_select32([&](int e, uint32_t& v) { v = lkey[e]; return v != kNoKey; }, ksel, hist, true, thr1, need, have);
            // after an early exit thr1 is an upper bound of a wholly taken class; after 4 passes it is an exact key value
            if (need < have) {
              exact_class = true;
              const uint32_t eq = thr1;
              int n2, h2;
              radix_select32([&](int e, uint32_t& v) {
                if (lkey[e] != eq) return false;
                v = entry_char1(e);
                return true;
              }, need, hist + 4 * 256, true, thr2, n2, h2);
              if (n2 < h2) {  // several entries with the threshold score AND the threshold character: entry order decides
                exact_char = true;
                const uint32_t ceq = thr2;
                int n3, h3;
                radix_select32([&](int e, uint32_t& v) {
                  if (lkey[e] != eq || entry_char1(e) != ceq) return false;
                  v = (uint32_t)e;
                  return true;
                }, n2, hist, false, thr3, n3, h3);
              }
            }
          }
          TS(5);
          auto keeps = [&](int e) -> bool {
            const uint32_t v = lkey[e];
            if (v == kNoKey) return false;
            if (!exact_class) return v <= thr1;
            if (v != thr1) return v < thr1;
            const uint32_t c1 = entry_char1(e);
            if (!exact_char) return c1 <= thr2;
            if (c1 != thr2) return c1 < thr2;
            return (uint32_t)e <= thr3;
          };
          // ---- (g) ordered compaction: survivors in list order (one block scan over per-thread counts) ----
          int my_keep = 0;
          unsigned long long keep_bits = 0;  // verdicts of the first 64 entries of this thread's range, evaluated once
          for (int e = e_lo; e < e_hi; ++e) {
            const bool k = keeps(e);
            my_keep += k ? 1 : 0;
            if (e - e_lo < 64) keep_bits |= (unsigned long long)(k ? 1 : 0) << (e - e_lo);
          }
          int tot_keep;
          int wpos = block_excl_scan<NW>(my_keep, wave_tot + 2 * NW, tot_keep);
          if (my_keep) {
            int i = row0, ro = row_off(row0), rl = row_len(row0);  // row of the entry being visited (children are walked in order)
            for (int e = e_lo; e < e_hi; ++e) {
              const bool k = (e - e_lo < 64) ? (((keep_bits >> (e - e_lo)) & 1ull) != 0) : keeps(e);
              if (!k || wpos >= beam) continue;
              if (e < nb) {
                surv_lp[wpos] = 0.f;
                surv[wpos++] = e;
              } else {
                const int s = e - nb;
                while (s >= ro + rl) { ++i; ro += rl; rl = row_len(i); }
                surv_lp[wpos] = score_of_key(lkey[e]);  // the extension's log-probability, computed once in (e)
                surv[wpos++] = kChildBit | (i << 14) | (s - ro);
              }
            }
          }
          if (tid == 0) fl[2] = ksel;
          if (clip && tid < nb) {  // verification: the bound of the best excluded child of my row
            const int K = row_len(tid);
            if (K < C) {
              const float ub = cand_lp(K) + cur.score[tid];
              if (desc_key(ub) <= thr1) fl[1] = 1;  // (thr1 = 0xFFFFFFFE when the list held fewer than `beam` elements)
            }
          }
        }
        if (WIDE && my_flag >= 0) lex[my_flag] = 0;
        list_barrier();
        TS(6);
        return fl[1] == 0;
      };
      const bool ok = in_lds ? select(lkey_s, lex_s, std::true_type{}) : select(lkey_g, lex_g, std::false_type{});
      k_sel = fl[2];
      if (ok) {
#ifdef PPASR_BEAM_TS
        if (clip) ++ts_ok;
#endif
        break;
      }
      clip = false;  // the bound of a clipped row reaches into the selection: full rows
      lds_barrier();  // (every wave has read the words)
      if (tid == 0) { fl[1] = 0; fl[2] = 0; fl[3] = 0; }
      lds_barrier();
    }
    if (k_sel < 0) break;  // no scratch for a list that needs it (status set)
    int n_nodes_next;
    const bool tail_done = tiny_done && !cfg.node_table;  // the tiny-beam path wrote the beam and did the end-of-frame work
    // node ids of the new prefixes: looked up in the node table first (a prefix that was in the beam before keeps its
    // identity, ctc_beam.h), misses get fresh ids in slot order (one block scan) and are entered into the table
    int my_parent = -1, my_char = 1, my_id = -1;  // (k_sel <= beam <= BT: one slot per thread)
    if (!cfg.node_table) {
      n_nodes_next = n_nodes + k_sel;  // every new prefix takes the id of its slot
    } else {
      const int pos = tid;
      int miss = 0;
      if (pos < k_sel && (surv[pos] & kChildBit)) {
        const int code = surv[pos], i = (code >> 14) & 0xFFFF;
        my_parent = cur.node[i];
        my_char = cand_c(code & 0x3FFF);
        const unsigned long long key = beam_node_key(my_parent, my_char);
        size_t slot = beam_node_slot(key, tslots);
        for (;;) {
          const unsigned long long kx = tkeys[slot];
          if (kx == key) { my_id = tids[slot]; break; }
          if (kx == 0ull) break;
          slot = slot + 1 == tslots ? 0 : slot + 1;
        }
        miss = my_id < 0 ? 1 : 0;
      }
      int n_miss;
      const int before = block_excl_scan<NW>(miss, wave_tot + 3 * NW, n_miss);
      if (miss) {
        my_id = n_nodes + before;
        if (my_id < cfg.max_nodes) {
          arena[kArenaWords * (size_t)my_id] = my_parent;
          arena[kArenaWords * (size_t)my_id + 1] = my_char;
          const unsigned long long key = beam_node_key(my_parent, my_char);
          size_t slot = beam_node_slot(key, tslots);
          while (atomicCAS(&tkeys[slot], 0ull, key) != 0ull) slot = slot + 1 == tslots ? 0 : slot + 1;
          tids[slot] = my_id;
        }
      }
      my_char = miss;  // (re-used below: 1 = a new node, 0 = a revived one)
      n_nodes_next = n_nodes + n_miss;
    }
    if (!tail_done) {
      if (tid < k_sel) {
        const int code = surv[tid];
        mat_slot(tid, code, surv_lp[tid], cfg.node_table ? my_id : n_nodes + tid, my_char != 0);
      }
      end_of_frame();
      lds_barrier();
    }
    TS(7);
    nb = __builtin_amdgcn_readfirstlane(k_sel);
    n_nodes = n_nodes_next;
    if (cfg.node_table) __threadfence_block();  // its entries are looked up by other threads of this block in later frames
    if (n_nodes + beam > cfg.max_nodes) {  // arena exhausted: report, stop consuming frames
      if (tid == 0 && status) status[u] = 1;
      Beam tmp = cur; cur = nxt; nxt = tmp;
      break;
    }
    Beam tmp = cur; cur = nxt; nxt = tmp;
  }
  __syncthreads();
#ifdef PPASR_BEAM_TS
  if (tid == BT - 64 && u == 0 && n_frames > 0)
    printf("last wave (x10ns/frame): to-d %lld rank-loop %lld scan+rows %lld barrier %lld\n", tw_acc[0] / n_frames, tw_acc[1] / n_frames,
           tw_acc[2] / n_frames, tw_acc[3] / n_frames);
  if (tid == 0 && u == 0 && n_frames > 0)
    printf("beam ts (x10ns/frame): [bucket %lld prefix+scan %lld rows %lld gen %lld] [d %lld rank %lld] inst %lld lm %lld contrib+rank %lld keys %lld sel %lld keep %lld mat %lld | list %lld C %lld nb %lld frames %d | "
           "rounds: clipped %lld verified %lld short-list %lld hbm %lld\n",
           ts_acc[10] / n_frames, ts_acc[11] / n_frames, ts_acc[12] / n_frames, ts_acc[13] / n_frames,
           ts_acc[8] / n_frames, ts_acc[9] / n_frames, ts_acc[0] / n_frames, ts_acc[1] / n_frames, ts_acc[2] / n_frames, ts_acc[3] / n_frames, ts_acc[5] / n_frames,
           ts_acc[6] / n_frames, ts_acc[7] / n_frames, ts_n / n_frames, ts_c / n_frames, ts_nb / n_frames, n_frames, ts_att, ts_ok,
           ts_small, ts_hbm);
#endif
  // ---- persist the state (streaming: CtcBeamSearchDecoderBatch keeps its trie between next() calls) ----
  if (tid == 0) { st[0] = nb; st[1] = n_nodes; }
  for (int i = tid; i < nb; i += BT) {
    g_arr[i] = cur.node[i]; g_arr[beam + i] = cur.chr[i]; g_arr[2 * beam + i] = cur.par[i];
    g_arr[3 * beam + i] = __float_as_int(cur.b[i]); g_arr[4 * beam + i] = __float_as_int(cur.nb[i]);
    g_arr[5 * beam + i] = __float_as_int(cur.score[i]);
    for (int j = 0; j < kLmCtx; ++j) g_arr[(6 + j) * beam + i] = cur.ctx[i * kLmCtx + j];
    g_arr[(6 + kLmCtx) * beam + i] = cur.dst[i];
  }
  if (!finalize) return;
  __threadfence_block();
  __syncthreads();
  // word-based scorer (ctc_beam_search_decoder.cpp, after the last frame): "score the last word of each prefix that
  // doesn't end with space" -- alpha * ln P(word | context) + beta is added to the score the prefixes are RANKED by; a
  // partial word that is no vocabulary word scores OOV.  Done on a copy (new_score): the persisted beam keeps the
  // running scores, so a streaming decoder that asks for the current best after every chunk does not accumulate it.
  for (int q = tid; q < nb; q += BT) {
    float sc = cur.score[q];
    if (word_lm && cur.node[q] > 0 && cur.chr[q] != space_id) {
      const int to = lm_dict_arc(cfg.lm, cur.dst[q], space_id);
      int32_t win[kLmMaxOrder];
      const int order = cfg.lm.order;
      for (int j = 0; j < order - 1; ++j) win[j] = cur.ctx[q * kLmCtx + (kLmCtx - (order - 1)) + j];
      win[order - 1] = to >= 0 ? cfg.lm.dict_word[to] : 0;
      float add = (float)(lm_log_cond_prob(cfg.lm, win) * cfg.alpha);
      add = (float)((double)add + cfg.beta);
      sc += add;
    }
    new_score[q] = sc;
  }
  __syncthreads();
  // ---- get_beam_search_result: rank the beam by prefix_compare, emit the n-best paths ----
  // rank of slot q = number of slots that sort before it (beam <= a few hundred: O(beam^2 / 256))
  for (int q = tid; q < nb; q += BT) {
    const uint64_t kq = make_key(new_score[q], cur.chr[q], q);
    int rank = 0;
    for (int i = 0; i < nb; ++i) rank += (make_key(new_score[i], cur.chr[i], i) < kq) ? 1 : 0;
    if (rank < cfg.nbest) {
      int len = 0;
      for (int n = cur.node[q]; n > 0; n = arena[kArenaWords * (size_t)n]) ++len;
      int32_t* dst = out_tokens + ((size_t)u * cfg.nbest + rank) * cfg.max_tokens;
      for (int j = 0; j < cfg.max_tokens; ++j) dst[j] = -1;
      int j = len;
      for (int n = cur.node[q]; n > 0; n = arena[kArenaWords * (size_t)n]) {
        --j;
        if (j < cfg.max_tokens) dst[j] = arena[kArenaWords * (size_t)n + 1];
      }
      out_lens[(size_t)u * cfg.nbest + rank] = len;
      double approx_ctc = (double)new_score[q];
      if (word_lm) {
        // approx_ctc = score - prefix_length * beta - alpha * get_sent_log_prob(split_labels(prefix)): the words between
        // the spaces (a trailing partial word included), spelt back through the dictionary; prefix_length counts
        // CHARACTERS (upstream subtracts one beta per character although the search added one per word)
        const int order = cfg.lm.order;
        int32_t win[kLmMaxOrder];
        for (int j2 = 0; j2 < order; ++j2) win[j2] = cfg.lm.bos;
        double sent = 0.0;
        int n_words = 0, state = 0, run = 0;
        auto push = [&](int word) {
          for (int j2 = 0; j2 + 1 < order; ++j2) win[j2] = win[j2 + 1];
          win[order - 1] = word;
          sent += lm_log_cond_prob(cfg.lm, win);
          ++n_words;
        };
        const int lim = min(len, cfg.max_tokens);
        for (int t2 = 0; t2 < lim; ++t2) {
          const int c = dst[t2];
          if (c == space_id) {
            if (run > 0) {
              const int to = state >= 0 ? lm_dict_arc(cfg.lm, state, c) : -1;
              push(to >= 0 ? cfg.lm.dict_word[to] : 0);
            }
            state = 0;
            run = 0;
          } else {
            state = state >= 0 ? lm_dict_arc(cfg.lm, state, c) : -1;
            ++run;
          }
        }
        if (run > 0) {
          const int to = state >= 0 ? lm_dict_arc(cfg.lm, state, space_id) : -1;
          push(to >= 0 ? cfg.lm.dict_word[to] : 0);
        }
        if (n_words == 0) sent += lm_log_cond_prob(cfg.lm, win);  // no words: the sentence is order x <s>, then </s>
        push(cfg.lm.eos);
        approx_ctc = approx_ctc - (double)len * cfg.beta - sent * cfg.alpha;
      } else if (has_lm) {
        // approx_ctc = score - prefix_length * beta - alpha * Scorer::get_sent_log_prob(words): the sentence is
        // <s> x (order-1) + words + </s>, scored window by window (scorer.cpp get_log_prob)
        const int order = cfg.lm.order;
        double sent = 0.0;
        int32_t win[kLmMaxOrder];
        auto window_of = [&](int node, int last_word) {  // words before `node` (exclusive of last_word's own slot)
          win[order - 1] = last_word;
          int n = node;
          for (int j = order - 2; j >= 0; --j) {
            if (n > 0) { win[j] = cfg.lm.tok2lm[arena[kArenaWords * (size_t)n + 1]]; n = arena[kArenaWords * (size_t)n]; }
            else win[j] = cfg.lm.bos;
          }
        };
        if (len == 0) {
          for (int j = 0; j < order; ++j) win[j] = cfg.lm.bos;
          sent += lm_log_cond_prob(cfg.lm, win);
        }
        window_of(cur.node[q], cfg.lm.eos);
        sent += lm_log_cond_prob(cfg.lm, win);
        for (int n = cur.node[q]; n > 0; n = arena[kArenaWords * (size_t)n]) {
          window_of(arena[kArenaWords * (size_t)n], cfg.lm.tok2lm[arena[kArenaWords * (size_t)n + 1]]);
          sent += lm_log_cond_prob(cfg.lm, win);
        }
        approx_ctc = approx_ctc - (double)len * cfg.beta - sent * cfg.alpha;
      }
      out_scores[(size_t)u * cfg.nbest + rank] = -approx_ctc;
    }
  }
  // ranks >= nb (beam smaller than nbest): mark empty
  for (int r = nb + tid; r < cfg.nbest; r += BT) {
    out_lens[(size_t)u * cfg.nbest + r] = -1;
    out_scores[(size_t)u * cfg.nbest + r] = 0.0;
  }
}


// ---- small beams: ONE WAVE per utterance, everything in registers ------------------------------------------------
// beam_size <= BM (16) and <= 64 pruned characters per frame (PPASR's beam 10 / top-40 evaluation setting, BASELINE
// configs[3], [4]).  The block-wide kernel above spends its 5-6 us per frame in a dozen workgroup barriers and LDS round
// trips over 410 (hypothesis, candidate) elements; here lane k owns CANDIDATE k of the frame and lanes 0..nb-1 also own
// HYPOTHESIS q, so that
//   * a hypothesis' fields are wave-uniform when its children are formed (v_readlane -> SGPR), no LDS tables;
//   * "is character c in the pruned list / which slot is my parent" are ballots;
//   * the element space is nb + 1 registers per lane: slot i = child (hypothesis i, candidate = lane), plus the lane's
//     own hypothesis; exact top-beam = repeated extraction of the minimum key (DPP reduction), ties at the cut resolved
//     in the block kernel's (character, element id) order;
//   * survivors are placed in element order with ballot prefix counts; only the new beam passes through LDS (scatter).
// Same arithmetic, same keys, same order as k_ctc_beam: the two kernels return identical beams (tests run both).
namespace {
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x111, 0xf, 0xf, false));
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x112, 0xf, 0xf, false));
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x114, 0xf, 0xf, false));
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x118, 0xf, 0xf, false));
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x142, 0xa, 0xf, false));
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x143, 0xc, 0xf, false));
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ float wave_min_f32(float x) {
  for (int o = 32; o > 0; o >>= 1) x = fminf(x, __shfl_xor(x, o));
  return x;
}
}  // namespace

template <int BM, bool HAS_LM>
__global__ __launch_bounds__(64) void k_ctc_beam_wave(const int32_t* __restrict__ frame_lens, int T, BeamConfig cfg,
                                                       const int32_t* __restrict__ recs, int32_t* __restrict__ state,
                                                       int init_state, int finalize, int32_t* __restrict__ out_tokens,
                                                       int32_t* __restrict__ out_lens, double* __restrict__ out_scores,
                                                       int32_t* __restrict__ status) {
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  __shared__ int nx_i[3][BM];
  __shared__ float nx_f[3][BM];
  __shared__ int nx_ctx[BM][kLmCtx];
  const int lane = threadIdx.x;
  const int u = blockIdx.x;
  const int beam = cfg.beam, blank = cfg.blank, CM = cfg.n_cand_max;
  int32_t* st = state + (size_t)u * beam_state_words(beam, cfg.max_nodes);
  int32_t* g_arr = st + 2;
  int32_t* arena = st + beam_fixed_words(beam);
  const size_t tslots = beam_table_slots(cfg.max_nodes);
  unsigned long long* tkeys = reinterpret_cast<unsigned long long*>(arena + beam_arena_words(cfg.max_nodes));
  int32_t* tids = reinterpret_cast<int32_t*>(tkeys + tslots);
  // hypothesis q lives in lane q
  int h_node = -2, h_chr = -1, h_par = -1;
  float h_b = kNegInf, h_nb = kNegInf, h_score = kNegInf;
  int h_ctx[kLmCtx];
#pragma unroll
  for (int j = 0; j < kLmCtx; ++j) h_ctx[j] = cfg.lm.bos;
  int nb, n_nodes;
  if (init_state) {
    nb = 1;
    n_nodes = 1;
    if (lane == 0) {
      h_node = 0; h_chr = -1; h_par = -1;
      h_b = 0.f; h_nb = kNegInf; h_score = 0.f;
      arena[0] = -1; arena[1] = -1;
    }
  } else {
    nb = st[0];
    n_nodes = st[1];
    if (lane < nb) {
      h_node = g_arr[lane]; h_chr = g_arr[beam + lane]; h_par = g_arr[2 * beam + lane];
      h_b = __int_as_float(g_arr[3 * beam + lane]); h_nb = __int_as_float(g_arr[4 * beam + lane]);
      h_score = __int_as_float(g_arr[5 * beam + lane]);
#pragma unroll
      for (int j = 0; j < kLmCtx; ++j) h_ctx[j] = g_arr[(6 + j) * beam + lane];
    }
  }
  nb = __builtin_amdgcn_readfirstlane(nb);
  n_nodes = __builtin_amdgcn_readfirstlane(n_nodes);
  const int n_frames = frame_lens ? min(max(frame_lens[u], 0), T) : T;
  const int RW = prune_rec_words(CM);
  const int32_t* rec_u = recs + (size_t)u * T * RW;
  int pre_C = 0, pre_pb = 0, pre_c = 0, pre_lp = 0;
  auto fetch = [&](int t) {
    const int32_t* r = rec_u + (size_t)t * RW;
    pre_C = r[0];
    pre_pb = r[1];
    pre_c = 0;
    pre_lp = 0;
    if (lane < CM) { pre_c = r[2 + lane]; pre_lp = r[2 + CM + lane]; }
  };
  if (n_frames > 0) fetch(0);
#ifdef PPASR_BEAM_TS
  long long ws_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ws_last = wall_clock64(), ws_c = 0, ws_nb = 0, ws_rounds = 0;
#define WTS(i) do { long long now = wall_clock64(); ws_acc[i] += now - ws_last; ws_last = now; } while (0)
#else
#define WTS(i)
#endif
  for (int t = 0; t < n_frames; ++t) {
    const int C = __builtin_amdgcn_readfirstlane(pre_C);
    const float p_blank = __int_as_float(__builtin_amdgcn_readfirstlane(pre_pb));
    const int cc = lane < C ? pre_c : -2;  // this lane's candidate character (-2: none)
    const float clp = __int_as_float(pre_lp);
    if (t + 1 < n_frames) fetch(t + 1);
    // ---- blank among the candidates? ----
    const unsigned long long mblank = __ballot(cc == blank);
    const bool has_blank = mblank != 0;
    const float lpb = has_blank ? rl_f(clp, __ffsll((long long)mblank) - 1) : 0.f;
    // ---- external scorer: pruning threshold of the frame (see k_ctc_beam) ----
    float min_cutoff = kNegInf;
    bool full_beam = false;
    if (HAS_LM) {
      const float m = wave_min_f32(lane < nb ? h_score : FLT_MAX);
      min_cutoff = (float)((double)m + log((double)p_blank) - fmax(0.0, cfg.beta));
      full_beam = (nb == beam);
    }
    auto pruned = [&](float lp_c, float score) -> bool { return HAS_LM && full_beam && (lp_c + score < min_cutoff); };
    auto lm_term = [&](const int* ctx, int c) -> float {  // alpha * ln P_lm(c | last order-1 words)
      int32_t win[kLmMaxOrder];
      const int order = cfg.lm.order;
      for (int j = 0; j < order - 1; ++j) win[j] = ctx[(kLmCtx - (order - 1)) + j];
      win[order - 1] = cfg.lm.tok2lm[c];
      return (float)(lm_log_cond_prob(cfg.lm, win) * cfg.alpha);
    };
    WTS(0);
    // ---- per hypothesis q (uniform loop): slot of its last character in the candidate list, slot of its parent in the
    // beam, and the "child (parent, character) already exists" bit of candidate lane kq ----
    int my_kq = -1, my_pi = -1;
    float my_lq = kNotCand;
    uint32_t exbits = 0;  // bit i: the child (hypothesis i, this lane's candidate) is already in the beam
#pragma unroll
    for (int q = 0; q < BM; ++q) {
      if (q < nb) {
        const int cq = rl_i(h_chr, q), pn = rl_i(h_par, q);
        const unsigned long long mc = __ballot(cc == cq);
        const unsigned long long mp = __ballot(lane < nb && h_node == pn);
        const int kq = mc ? __ffsll((long long)mc) - 1 : -1;
        const int pi = mp ? __ffsll((long long)mp) - 1 : -1;
        const float lq = mc ? rl_f(clp, kq < 0 ? 0 : kq) : kNotCand;
        if (lane == q) { my_kq = kq; my_pi = pi; my_lq = lq; }
        if (mc && mp && cq != blank && lane == kq) exbits |= 1u << pi;
      }
    }
    WTS(1);
    // ---- (d) contributions received by the hypotheses already in the beam (lanes q < nb) ----
    float new_b = kNegInf, new_nb = kNegInf, new_score = kNegInf;
    {
      const int pa = (my_pi < 0 ? 0 : my_pi) * 4;
      const float p_score = __int_as_float(__builtin_amdgcn_ds_bpermute(pa, __float_as_int(h_score)));
      const float p_b = __int_as_float(__builtin_amdgcn_ds_bpermute(pa, __float_as_int(h_b)));
      const int p_chr = __builtin_amdgcn_ds_bpermute(pa, h_chr);
      int p_ctx[kLmCtx];
      if (HAS_LM) {
#pragma unroll
        for (int j = 0; j < kLmCtx; ++j) p_ctx[j] = __builtin_amdgcn_ds_bpermute(pa, h_ctx[j]);
      }
      if (lane < nb) {
        float bc = (has_blank && !pruned(lpb, h_score)) ? lpb + h_score : kNegInf;
        float nbc = kNegInf;
        if (my_lq != kNotCand && h_chr != blank && h_chr >= 0) {
          if (!pruned(my_lq, h_score)) nbc = my_lq + h_nb;  // repeated character
          if (my_pi >= 0 && !pruned(my_lq, p_score)) {     // extension of the parent hypothesis lands on this prefix
            float log_p = kNegInf;
            if (h_chr == p_chr) { if (p_b > kNegInf) log_p = my_lq + p_b; }
            else log_p = my_lq + p_score;
            if (HAS_LM) {
              log_p += lm_term(p_ctx, h_chr);
              log_p = (float)((double)log_p + cfg.beta);
            }
            nbc = lse(nbc, log_p);
          }
        }
        new_b = bc;
        new_nb = nbc;
        new_score = lse(bc, nbc);
      }
    }
    WTS(2);
    // ---- (e) keys: kk[i] = child (hypothesis i, this lane's candidate), kh = this lane's own hypothesis ----
    uint32_t kk[BM], kk0[BM];
    uint32_t kh = lane < nb ? desc_key(new_score) : kNone;
    const uint32_t kh0 = kh;
    int n_valid = nb;
#pragma unroll
    for (int i = 0; i < BM; ++i) {
      kk[i] = kNone;
      if (i < nb) {
        const int ci = rl_i(h_chr, i);
        const float bi = rl_f(h_b, i), si = rl_f(h_score, i);
        const bool ok = cc >= 0 && cc != blank && !((exbits >> i) & 1u) && !pruned(clp, si);
        float log_p = kNegInf;
        if (cc == ci) { if (bi > kNegInf) log_p = clp + bi; }
        else log_p = clp + si;
        if (HAS_LM) {
          if (ok) {
            int ictx[kLmCtx];
#pragma unroll
            for (int j = 0; j < kLmCtx; ++j) ictx[j] = rl_i(h_ctx[j], i);
            log_p += lm_term(ictx, cc);
            log_p = (float)((double)log_p + cfg.beta);
          }
        }
        kk[i] = ok ? desc_key(log_p) : kNone;
        n_valid += __popcll(__ballot(ok));
      }
      kk0[i] = kk[i];
    }
    const int k_sel = n_valid >= beam ? beam : n_valid;
    WTS(3);
#ifdef PPASR_BEAM_TS
    ws_c += C; ws_nb += nb;
#endif
    // ---- (f) exact top-k_sel in (score key, character, element id) order: extract the minimum key until k_sel elements
    // are taken; a taken element's key becomes kNone (kk0 / kh0 keep the original).  k_sel == n_valid: everything stays.
    if (k_sel < n_valid) {
      int taken = 0;
      while (taken < k_sel) {
        uint32_t lm = kh;
#pragma unroll
        for (int i = 0; i < BM; ++i)
          if (i < nb) lm = min(lm, kk[i]);
        const uint32_t g = wave_min_u32(lm);
        int total = __popcll(__ballot(kh == g));
#pragma unroll
        for (int i = 0; i < BM; ++i)
          if (i < nb) total += __popcll(__ballot(kk[i] == g));
        if (taken + total <= k_sel) {
          if (kh == g) kh = kNone;
#pragma unroll
          for (int i = 0; i < BM; ++i)
            if (i < nb && kk[i] == g) kk[i] = kNone;
          taken += total;
        } else {
          // tie at the cut: of the elements with key g take the k_sel - taken smallest (character + 1, element id)
          for (; taken < k_sel; ++taken) {
            uint32_t tk = (kh == g) ? (((uint32_t)(h_chr + 1) << 18) | (uint32_t)lane) : kNone;
#pragma unroll
            for (int i = 0; i < BM; ++i)
              if (i < nb && kk[i] == g) tk = min(tk, ((uint32_t)(cc + 1) << 18) | (uint32_t)(nb + i * C + lane));
            const uint32_t w = wave_min_u32(tk);
            if (kh == g && ((((uint32_t)(h_chr + 1) << 18) | (uint32_t)lane) == w)) kh = kNone;
#pragma unroll
            for (int i = 0; i < BM; ++i)
              if (i < nb && kk[i] == g && ((((uint32_t)(cc + 1) << 18) | (uint32_t)(nb + i * C + lane)) == w)) kk[i] = kNone;
          }
        }
      }
    } else {
      kh = kNone;
#pragma unroll
      for (int i = 0; i < BM; ++i) kk[i] = kNone;
    }
    WTS(4);
    // ---- (g) survivors in element order -> slots of the next beam (scatter through LDS) ----
    {
      const bool keep_h = kh0 != kNone && kh == kNone;
      const unsigned long long mh = __ballot(keep_h);
      if (keep_h) {
        const int pos = mbcnt(mh);
        nx_i[0][pos] = h_node; nx_i[1][pos] = h_chr; nx_i[2][pos] = h_par;
        nx_f[0][pos] = new_b; nx_f[1][pos] = new_nb; nx_f[2][pos] = new_score;
#pragma unroll
        for (int j = 0; j < kLmCtx; ++j) nx_ctx[pos][j] = h_ctx[j];
      }
      int base = __popcll(mh);
      int n_new = n_nodes;
#pragma unroll
      for (int i = 0; i < BM; ++i) {
        if (i < nb) {
          const bool keep = kk0[i] != kNone && kk[i] == kNone;
          const unsigned long long mk = __ballot(keep);
          if (mk) {
            const int node_i = rl_i(h_node, i);
            // node id: the node table first (ctc_beam.h: a prefix keeps its identity), fresh ids for the misses
            int id = -1;
            if (keep && cfg.node_table) {
              const unsigned long long key = beam_node_key(node_i, cc);
              size_t slot = beam_node_slot(key, tslots);
              for (;;) {
                const unsigned long long kx = tkeys[slot];
                if (kx == key) { id = tids[slot]; break; }
                if (kx == 0ull) break;
                slot = slot + 1 == tslots ? 0 : slot + 1;
              }
            }
            const unsigned long long mm = __ballot(keep && id < 0);
            if (keep && id < 0) {
              id = n_new + mbcnt(mm);
              if (id < cfg.max_nodes) {
                arena[kArenaWords * (size_t)id] = node_i;
                arena[kArenaWords * (size_t)id + 1] = cc;
                arena[kArenaWords * (size_t)id + 2] = 0;
                if (cfg.node_table) {
                  const unsigned long long key = beam_node_key(node_i, cc);
                  size_t slot = beam_node_slot(key, tslots);
                  while (atomicCAS(&tkeys[slot], 0ull, key) != 0ull) slot = slot + 1 == tslots ? 0 : slot + 1;
                  tids[slot] = id;
                }
              }
            }
            n_new += __popcll(mm);
            if (keep) {
              const int pos = base + mbcnt(mk);
              const float log_p = score_of_key(kk0[i]);
              nx_i[0][pos] = id; nx_i[1][pos] = cc; nx_i[2][pos] = node_i;
              nx_f[0][pos] = kNegInf; nx_f[1][pos] = log_p; nx_f[2][pos] = log_p;
#pragma unroll
              for (int j = 0; j + 1 < kLmCtx; ++j) nx_ctx[pos][j] = rl_i(h_ctx[j + 1], i);
              nx_ctx[pos][kLmCtx - 1] = HAS_LM ? cfg.lm.tok2lm[cc] : 0;
            }
            base += __popcll(mk);
          }
        }
      }
      n_nodes = n_new;
      if (cfg.node_table) __threadfence_block();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      h_node = -2; h_chr = -1; h_par = -1;
      h_b = kNegInf; h_nb = kNegInf; h_score = kNegInf;
      if (lane < k_sel) {
        h_node = nx_i[0][lane]; h_chr = nx_i[1][lane]; h_par = nx_i[2][lane];
        h_b = nx_f[0][lane]; h_nb = nx_f[1][lane]; h_score = nx_f[2][lane];
#pragma unroll
        for (int j = 0; j < kLmCtx; ++j) h_ctx[j] = nx_ctx[lane][j];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    nb = k_sel;
    WTS(5);
    if (n_nodes + beam > cfg.max_nodes) {  // arena exhausted: report, stop consuming frames
      if (lane == 0 && status) status[u] = 1;
      break;
    }
  }
#ifdef PPASR_BEAM_TS
  if (lane == 0 && u == 0 && n_frames > 0)
    printf("wave beam ts (x10ns/frame): fetch %lld qloop %lld contrib %lld keys %lld select %lld place %lld | C %lld nb %lld frames %d\n",
           ws_acc[0] / n_frames, ws_acc[1] / n_frames, ws_acc[2] / n_frames, ws_acc[3] / n_frames, ws_acc[4] / n_frames,
           ws_acc[5] / n_frames, ws_c / n_frames, ws_nb / n_frames, n_frames);
#endif
  // ---- persist the state (same layout as k_ctc_beam) ----
  if (lane == 0) { st[0] = nb; st[1] = n_nodes; }
  if (lane < nb) {
    g_arr[lane] = h_node; g_arr[beam + lane] = h_chr; g_arr[2 * beam + lane] = h_par;
    g_arr[3 * beam + lane] = __float_as_int(h_b); g_arr[4 * beam + lane] = __float_as_int(h_nb);
    g_arr[5 * beam + lane] = __float_as_int(h_score);
#pragma unroll
    for (int j = 0; j < kLmCtx; ++j) g_arr[(6 + j) * beam + lane] = h_ctx[j];
    g_arr[(6 + kLmCtx) * beam + lane] = 0;  // (dictionary state: word-based scorers take the block-wide kernel)
  }
  if (!finalize) return;
  __threadfence_block();  // this wave's arena stores are read back below
  // ---- get_beam_search_result: rank the beam by prefix_compare, emit the n-best paths ----
  {
    const uint64_t kq = make_key(h_score, h_chr, lane);
    int rank = 0;
#pragma unroll
    for (int i = 0; i < BM; ++i) {
      if (i < nb) {
        const uint64_t ki = make_key(rl_f(h_score, i), rl_i(h_chr, i), i);
        rank += (ki < kq) ? 1 : 0;
      }
    }
    if (lane < nb && rank < cfg.nbest) {
      int len = 0;
      for (int n = h_node; n > 0; n = arena[kArenaWords * (size_t)n]) ++len;
      int32_t* dst = out_tokens + ((size_t)u * cfg.nbest + rank) * cfg.max_tokens;
      for (int j = 0; j < cfg.max_tokens; ++j) dst[j] = -1;
      int j = len;
      for (int n = h_node; n > 0; n = arena[kArenaWords * (size_t)n]) {
        --j;
        if (j < cfg.max_tokens) dst[j] = arena[kArenaWords * (size_t)n + 1];
      }
      out_lens[(size_t)u * cfg.nbest + rank] = len;
      double approx_ctc = (double)h_score;
      if (HAS_LM) {
        const int order = cfg.lm.order;
        double sent = 0.0;
        int32_t win[kLmMaxOrder];
        auto window_of = [&](int node, int last_word) {
          win[order - 1] = last_word;
          int n = node;
          for (int j2 = order - 2; j2 >= 0; --j2) {
            if (n > 0) { win[j2] = cfg.lm.tok2lm[arena[kArenaWords * (size_t)n + 1]]; n = arena[kArenaWords * (size_t)n]; }
            else win[j2] = cfg.lm.bos;
          }
        };
        if (len == 0) {
          for (int j2 = 0; j2 < order; ++j2) win[j2] = cfg.lm.bos;
          sent += lm_log_cond_prob(cfg.lm, win);
        }
        window_of(h_node, cfg.lm.eos);
        sent += lm_log_cond_prob(cfg.lm, win);
        for (int n = h_node; n > 0; n = arena[kArenaWords * (size_t)n]) {
          window_of(arena[kArenaWords * (size_t)n], cfg.lm.tok2lm[arena[kArenaWords * (size_t)n + 1]]);
          sent += lm_log_cond_prob(cfg.lm, win);
        }
        approx_ctc = approx_ctc - (double)len * cfg.beta - sent * cfg.alpha;
      }
      out_scores[(size_t)u * cfg.nbest + rank] = -approx_ctc;
    }
    for (int r = nb + lane; r < cfg.nbest; r += 64) {
      out_lens[(size_t)u * cfg.nbest + r] = -1;
      out_scores[(size_t)u * cfg.nbest + r] = 0.0;
    }
  }
}

constexpr int kWaveBeamMax = 16;  // k_ctc_beam_wave: beam_size <= 16 and <= 64 pruned characters per frame

__global__ __launch_bounds__(256) void k_beam_rehash(int32_t* __restrict__ state, int beam, int max_nodes) {
  int32_t* st = state + (size_t)blockIdx.x * beam_state_words(beam, max_nodes);
  const int32_t* arena = st + beam_fixed_words(beam);
  const size_t tslots = beam_table_slots(max_nodes);
  unsigned long long* tkeys = reinterpret_cast<unsigned long long*>(st + beam_fixed_words(beam) + beam_arena_words(max_nodes));
  int32_t* tids = reinterpret_cast<int32_t*>(tkeys + tslots);
  const int n_nodes = st[1];
  for (int id = 1 + threadIdx.x; id < n_nodes; id += blockDim.x) {
    const unsigned long long key = beam_node_key(arena[kArenaWords * (size_t)id], arena[kArenaWords * (size_t)id + 1]);
    size_t slot = beam_node_slot(key, tslots);
    while (atomicCAS(&tkeys[slot], 0ull, key) != 0ull) slot = slot + 1 == tslots ? 0 : slot + 1;
    tids[slot] = id;
  }
}
hipError_t launch_beam_rehash(int32_t* state, int B, int beam, int max_nodes, hipStream_t st) {
  PPASR_LAUNCH(k_beam_rehash, dim3(B), dim3(256), 0, st, state, beam, max_nodes);
  return hipGetLastError();
}

hipError_t launch_ctc_beam(const float* probs, const int32_t* frame_lens, int B, int T, const BeamConfig& cfg,
                           int32_t* prune_recs, int32_t* state, int init_state, int finalize, int32_t* out_tokens,
                           int32_t* out_lens, double* out_scores, int32_t* status, void* scratch, hipStream_t st) {
  const size_t lds = beam_lds_bytes(cfg);
  const bool wide = cfg.n_cand_max > kSmallCand;
  // threads per utterance by the number of (hypothesis, candidate) elements a frame can have: every phase is a chain of
  // block-wide steps, and a barrier over few waves is cheaper than one over 16
  const size_t n_elem = (size_t)cfg.beam * (1 + (size_t)cfg.n_cand_max);
  // 512 threads up to 1 024 elements and for beams up to 128, else 768 (12 waves: 170 registers per lane -- the 1 024-thread
  // form has 128, which the scorer's look-ups overflow into scratch -- and cheaper barriers; measured, flat posteriors, us per
  // frame on 512 / 768 / 1 024 threads: beam 100 10.8 / 11.3 / 11.5, beam 300 14.3 / 12.8 / 13.0)
  int sel = (n_elem <= 1024 || cfg.beam <= 128) ? 0 : 2;
  if (const char* e = getenv("PPASR_BEAM_BT")) sel = atoi(e) >= 768 ? 2 : 0;  // (tuning knob: 512 / 768 threads)
  const bool wl = cfg.lm.order > 0 && cfg.lm.word_based != 0;
  // scratch: [wide pruning records] [per-utterance element lists]
  const size_t rec_bytes = (scratch_rec_bytes(cfg, B, T) + 255) & ~(size_t)255, list_stride = scratch_list_bytes_per_utt(cfg);
  if ((rec_bytes || list_stride) && !scratch) return hipErrorInvalidValue;
  int32_t* recs = wide ? static_cast<int32_t*>(scratch) : prune_recs;
  char* lists = list_stride ? static_cast<char*>(scratch) + rec_bytes : nullptr;
  if (list_stride) {  // "child exists" flags of the HBM lists start clear (each is set and cleared by its setter)
    const size_t n = (size_t)cfg.beam * (1 + (size_t)cfg.n_cand_max);
    hipError_t e = hipMemset2DAsync(lists + 4 * n, list_stride, 0, list_stride - 4 * n, (size_t)B, st);
    if (e != hipSuccess) return e;
  }
  if (T > 0) {
    if (wide) {
      const int P2 = wide_pow2(cfg.V);
      const size_t plds = prune_wide_lds_bytes(cfg);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_prune_wide),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
      if (e != hipSuccess) return e;
      PPASR_LAUNCH(k_ctc_prune_wide, dim3(T, B), dim3(kWideThreads), plds, st, probs, frame_lens, T, cfg, P2, recs);
    } else {
      const size_t plds = prune_lds_bytes(cfg);
      // hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set it on every launch (a few host
      // microseconds) rather than caching "already configured" in process-wide statics, which left a second GPU used
      // from the same process unconfigured and was not thread-safe
      if (plds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_prune<kPruneThreads>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
        if (e != hipSuccess) return e;
      }
      PPASR_LAUNCH(k_ctc_prune<kPruneThreads>, dim3(T, B), dim3(kPruneThreads), plds, st, probs, frame_lens, T, cfg, recs);
    }
  }
  // small beams: one wave per utterance.  OPT-IN (PPASR_BEAM_WAVE=1): measured 11 us per frame at beam 10 against
  // 5.4 us for the block-wide kernel (a lone wave issues its readlane / ballot / branch sequences at ~9 cycles per
  // instruction); kept because it needs no LDS tables and co-resides with the encoder's workgroups.
  const char* wave_env = getenv("PPASR_BEAM_WAVE");
  if (cfg.beam <= kWaveBeamMax && cfg.n_cand_max <= 64 && wave_env && atoi(wave_env) == 1 && !cfg.lm.word_based) {
    if (cfg.lm.order > 0)
      PPASR_LAUNCH((k_ctc_beam_wave<kWaveBeamMax, true>), dim3(B), dim3(64), 0, st, frame_lens, T, cfg, recs, state,
                   init_state, finalize, out_tokens, out_lens, out_scores, status);
    else
      PPASR_LAUNCH((k_ctc_beam_wave<kWaveBeamMax, false>), dim3(B), dim3(64), 0, st, frame_lens, T, cfg, recs, state,
                   init_state, finalize, out_tokens, out_lens, out_scores, status);
    return hipGetLastError();
  }
#define PPASR_LAUNCH_BEAM(BT, LMK, WIDE)                                                                                \
  do {                                                                                                                  \
    const void* fn = reinterpret_cast<const void*>(k_ctc_beam<BT, LMK, WIDE>);                                          \
    if (lds > 48 * 1024) {                                                                                              \
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
      if (e != hipSuccess) return e;                                                                                    \
    }                                                                                                                   \
    PPASR_LAUNCH((k_ctc_beam<BT, LMK, WIDE>), dim3(B), dim3(BT), lds, st, probs, frame_lens, T, cfg, recs, state,       \
                 init_state, finalize, out_tokens, out_lens, out_scores, status, lists, list_stride);                   \
  } while (0)
#define PPASR_LAUNCH_BEAM_LM(BT, WIDE)                                                                                  \
  do {                                                                                                                  \
    if (wl) PPASR_LAUNCH_BEAM(BT, 2, WIDE);                                                                             \
    else if (cfg.lm.order > 0) PPASR_LAUNCH_BEAM(BT, 1, WIDE);                                                          \
    else PPASR_LAUNCH_BEAM(BT, 0, WIDE);                                                                                \
  } while (0)
  if (wide) PPASR_LAUNCH_BEAM_LM(1024, true);  // (wide records: always 1 024 threads)
  else if (sel == 0) PPASR_LAUNCH_BEAM_LM(512, false);
  else PPASR_LAUNCH_BEAM_LM(768, false);
#undef PPASR_LAUNCH_BEAM_LM
#undef PPASR_LAUNCH_BEAM
  return hipGetLastError();
}

}  // namespace ppasr
