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
//   * "does child (prefix, c) already exist in the beam" (the trie lookup) = a flag table built from
//     each hypothesis' parent slot; every hypothesis receives at most two non-blank contributions
//     (its own repeated character, the extension from its parent), so no floating-point atomics
//     and the same log_sum_exp values as the serial trie walk;
//   * top-k = exact MSD radix select on unique 64-bit keys (score | char | element id) recomputed on
//     the fly, then an ordered compaction -- nothing of size beam x candidates is ever stored.
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

#include "ctc_beam.h"
#include "launch.h"

namespace ppasr {

namespace {

constexpr int kFastBeam = 16;   // staircase fast path: beams up to this size ...
constexpr int kFastCap = 128;   // ... whose restricted element list fits this many entries (two per lane of a wave)
constexpr int kFastMargin = 2;  // extra candidates per hypothesis beyond the (rank + 1) (k + 1) <= beam staircase
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

size_t beam_lds_bytes(const BeamConfig& c) {
  const int Vp = (c.V + 3) & ~3;
  size_t n = 8 * 256 * 4 + 3 * (kBT / 64) * 4 + 32;  // per-pass histograms, scan / reduction scratch, scalars
  n += (size_t)kMaxBeamCand * 8;                   // cand_c, cand_lp
  n += (size_t)2 * c.beam * (28 + 4 * kLmCtx);     // two beam halves
  n += (size_t)c.beam * 20;                        // new_b, new_nb, new_score, new_dst, k_reset
  n += (size_t)Vp * 2;                             // kidx (int16)
  n += (((size_t)c.beam * c.n_cand_max) + 3) & ~(size_t)3;  // exists flags
  n = (n + 7) & ~(size_t)7;                        // (fkey: 8-byte entries)
  n += (size_t)kFastCap * 8 + (size_t)kFastBeam * 12;  // staircase fast path: slot keys, the selected keys, rank table
  n += (size_t)c.beam * 4;                         // surv_lp
  n += (size_t)c.beam * (1 + c.n_cand_max) * 4;    // score keys
  return (n + 15) & ~(size_t)15;
}

// (state layout per utterance: ctc_beam.h)
size_t beam_state_bytes(const BeamConfig& c) { return beam_state_words(c.beam, c.max_nodes) * 4; }

constexpr int kPruneThreads = 256;
static size_t prune_lds_bytes(const BeamConfig& c) {
  const int Vp = (c.V + 3) & ~3;
  return (16 + 256 * 4 + 2 * (kPruneThreads / 64) * 4 + 32 + 4 * kMaxBeamCand * 4 + (size_t)Vp * 4 + 15) & ~(size_t)15;
}


// ---- frame-parallel pre-pass: get_pruned_log_probs of every frame ----
// The pruned character list of a frame does not depend on the beam, so it is computed for all frames at once by a
// chip-wide launch (one workgroup per frame) instead of inside the sequential per-utterance loop.  Record of frame
// (u, t) in HBM (int32 words): [0] C  [1] p_blank (raw probability, float bits)  [2 .. 2+CM) characters in
// (prob desc, index asc) order  [2+CM .. 2+2CM) their log(p + FLT_MIN).
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
  int* tmp_c = reinterpret_cast<int*>(p); p += kMaxBeamCand * 4;
  float* tmp_p = reinterpret_cast<float*>(p); p += kMaxBeamCand * 4;
  int* cand_c = reinterpret_cast<int*>(p); p += kMaxBeamCand * 4;
  float* cand_lp = reinterpret_cast<float*>(p); p += kMaxBeamCand * 4;
  float* lp = reinterpret_cast<float*>(p); p += (size_t)Vp * 4;
  const bool prune = (cfg.cutoff_prob < 1.0) || (cfg.cutoff_top_n < V);
  {
    const float* row = probs + ((size_t)u * T + t) * V;
    for (int v = tid; v < V; v += NT) lp[v] = row[v];
    __syncthreads();
    // ---- (b) get_pruned_log_probs (decoder_utils.cpp): the n_sel largest probabilities in (prob desc, index asc)
    // order, cut where the cumulative probability reaches cutoff_prob.  Fast path: exact 4-pass radix select of the
    // n_sel-th largest value, unordered gather, rank sort of the <= 128 survivors.  Excess ties at the threshold
    // (more equal values than slots) fall back to the successive-maxima loop below.
    int C = 0;
    bool slow_path = false;
    const int n_sel = (cfg.cutoff_prob < 1.0) ? min(cfg.cutoff_top_n, V) : V;
    if (prune && n_sel <= CM) {
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
            if (pos < kMaxBeamCand) { tmp_c[pos] = v; tmp_p[pos] = pv; }
          }
        }
        __syncthreads();
        const int n_got = min(sh_i[4], kMaxBeamCand);
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
    } else if (prune) {
      slow_path = true;
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
      C = V;  // no pruning: vocabulary order (host guarantees V <= n_cand_max)
      for (int v = tid; v < V; v += NT) { cand_c[v] = v; cand_lp[v] = (float)log((double)lp[v] + (double)FLT_MIN); }
      __syncthreads();
    }
    int32_t* rec = recs + ((size_t)u * T + t) * prune_rec_words(CM);
    if (tid == 0) { rec[0] = C; rec[1] = __float_as_int(lp[blank]); }
    for (int k = tid; k < C; k += NT) { rec[2 + k] = cand_c[k]; rec[2 + CM + k] = __float_as_int(cand_lp[k]); }
  }
}

#ifndef PPASR_BEAM_WAVES_PER_SIMD
#define PPASR_BEAM_WAVES_PER_SIMD 1  // (launch-bounds hint = register budget 512 / n per lane; tuning knob)
#endif
template <int BT, bool WORD_LM>
__global__ __launch_bounds__(BT, PPASR_BEAM_WAVES_PER_SIMD) void k_ctc_beam(const float* __restrict__ probs, const int32_t* __restrict__ frame_lens,
                                                  int T, BeamConfig cfg, const int32_t* __restrict__ recs,
                                                  int32_t* __restrict__ state, int init_state,
                                                  int finalize, int32_t* __restrict__ out_tokens,
                                                  int32_t* __restrict__ out_lens, double* __restrict__ out_scores,
                                                  int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u = blockIdx.x;
  const int V = cfg.V, beam = cfg.beam, blank = cfg.blank, CM = cfg.n_cand_max;
  const int Vp = (V + 3) & ~3;
  char* p = smem;
  int* hist = reinterpret_cast<int*>(p); p += 8 * 256 * 4;
  int* wave_tot = reinterpret_cast<int*>(p); p += 2 * (BT / 64) * 4;
  float* red_p = reinterpret_cast<float*>(p); p += (BT / 64) * 4;
  int* sh_i = reinterpret_cast<int*>(p); p += 32;            // misc shared ints
  int* cand_c = reinterpret_cast<int*>(p); p += kMaxBeamCand * 4;
  float* cand_lp = reinterpret_cast<float*>(p); p += kMaxBeamCand * 4;
  Beam cur = carve_beam(p, beam);
  Beam nxt = carve_beam(p, beam);
  float* new_b = reinterpret_cast<float*>(p); p += beam * 4;
  float* new_nb = reinterpret_cast<float*>(p); p += beam * 4;
  float* new_score = reinterpret_cast<float*>(p); p += beam * 4;
  int* new_dst = reinterpret_cast<int*>(p); p += beam * 4;    // word-based LM: dictionary state of a surviving hypothesis
  int* k_reset = reinterpret_cast<int*>(p); p += beam * 4;    // ... candidate whose lookup reset the state (no child), or -1
  int16_t* kidx = reinterpret_cast<int16_t*>(p); p += (size_t)Vp * 2;
  const bool has_lm = cfg.lm.order > 0;
  constexpr bool word_lm = WORD_LM;  // scorer consulted at spaces, prefixes constrained by the dictionary (lm.word_based)
  const int space_id = cfg.lm.space_id;
  uint8_t* exists = reinterpret_cast<uint8_t*>(p); p += (((size_t)beam * CM) + 3) & ~(size_t)3;
  // staircase fast path (small beams without a scorer, see (e')): the restricted element list and its bookkeeping
  p = smem + (((size_t)(p - smem) + 7) & ~(size_t)7);
  unsigned long long* fkey = reinterpret_cast<unsigned long long*>(p); p += kFastCap * 8;          // keys of the list's slots
  unsigned long long* srank_key = reinterpret_cast<unsigned long long*>(p); p += kFastBeam * 8;  // the `beam` smallest, by rank
  int* hyp_of_rank = reinterpret_cast<int*>(p); p += kFastBeam * 4;
  float* surv_lp = reinterpret_cast<float*>(p); p += beam * 4;  // log-probability of a surviving CHILD, by slot of the next beam
  uint32_t* skey = reinterpret_cast<uint32_t*>(p);  // [beam * (1 + CM)] score keys of the frame's elements

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
  __syncthreads();

  const int n_frames = frame_lens ? min(max(frame_lens[u], 0), T) : T;
  // kidx[] = index of a character in the frame's candidate list (-1: not a candidate): cleared once, then only the entries
  // of the previous frame's characters are reset.  A character's log-prob is cand_lp[kidx[c]] -- a V-wide table of its own
  // (17 KB of LDS at V = 4233) kept the workgroup from sharing a CU with the encoder's 133 KB row-block workgroups when
  // the search of step i runs beside the encoder of step i + 1 (bench.py --config cfg4 / cfg5, evaluate()): the encoder's
  // launches then had 16 .. 64 fewer CUs and ran extra rounds.
  for (int v = tid; v < V; v += BT) kidx[v] = -1;
  auto lp_of = [&](int c) -> float {
    const int k = kidx[c];
    return k >= 0 ? cand_lp[k] : kNotCand;
  };
  // per-frame records of the pruning pre-pass (k_ctc_prune); the next frame's record is fetched into registers while
  // the current frame is processed
  const int RW = prune_rec_words(CM);
  const int32_t* rec_u = recs + (size_t)u * T * RW;
  constexpr int KPT = (kMaxBeamCand + BT - 1) / BT;  // candidates per thread
  int pre_C = 0, pre_pb = 0, pre_c[KPT], pre_lp[KPT];
  auto fetch = [&](int t) {
    const int32_t* r = rec_u + (size_t)t * RW;
    pre_C = r[0];
    pre_pb = r[1];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int k = tid + j * BT;
      pre_c[j] = 0;
      pre_lp[j] = 0;
      if (k < CM) { pre_c[j] = r[2 + k]; pre_lp[j] = r[2 + CM + k]; }
    }
  };
  if (n_frames > 0) fetch(0);
  __syncthreads();
#ifdef PPASR_BEAM_TS
  long long ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ts_last = wall_clock64(), ts_n = 0, ts_c = 0, ts_nb = 0, ts_att = 0, ts_ok = 0;
#define TS(i) do { if (tid == 0 && u == 0) { long long now = wall_clock64(); ts_acc[i] += now - ts_last; ts_last = now; } } while (0)
#define TSN() do { ts_n += N; ts_c += C; ts_nb += nb; } while (0)
#else
#define TS(i)
#define TSN()
#endif
  // (e') is taken by searches without an external scorer (its upper bound on a child's score needs score = acoustic only)
  // ... whose candidate lists are sorted by probability (k_ctc_prune leaves them in index order when nothing is pruned:
  // cutoff_prob >= 1 with cutoff_top_n >= V)
  bool fast_ok = !has_lm && beam <= kFastBeam && BT >= 128 && cfg.fast_path != 0 &&
                 (cfg.cutoff_prob < 1.0 || cfg.cutoff_top_n < V);
  // the list's slots are the same in every frame: slot q < beam = existing hypothesis q, then row r (the hypothesis of score
  // rank r) with its first K_r = beam / (r + 1) + margin candidates; slot `tid` is this thread's (my_r < 0: none)
  int my_r = -1, my_k = 0, n_s0 = beam;
  if (fast_ok) {
    int off = tid - beam;
    for (int r = 0; r < beam; ++r) {
      const int kr = beam / (r + 1) + kFastMargin;
      if (my_r < 0 && off >= 0 && off < kr) { my_r = r; my_k = off; }
      off -= kr;
      n_s0 += kr;
    }
    fast_ok = n_s0 <= kFastCap;
  }
  const int my_row_k = beam / (lane + 1) + kFastMargin;  // K_r of row r = lane (the verification's lanes)
  for (int t = 0; t < n_frames; ++t) {
    // ---- (b, c) this frame's pruned characters (get_pruned_log_probs, done by the pre-pass) ----
    const int C = pre_C;
    const float p_blank = __int_as_float(pre_pb);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int k = tid + j * BT;
      if (k < C) {
        cand_c[k] = pre_c[j];
        cand_lp[k] = __int_as_float(pre_lp[j]);
        kidx[pre_c[j]] = (int16_t)k;
      }
    }
    if (t + 1 < n_frames) fetch(t + 1);
    for (int e = tid; e < nb * C; e += BT) exists[e] = 0;
    if (tid == 0) sh_i[1] = 0;  // fast path: "verification failed"
    lds_barrier();
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
      lds_barrier();
      m = red_p[0];
      for (int w = 1; w < (BT / 64); ++w) m = fminf(m, red_p[w]);
      min_cutoff = (float)((double)m + log((double)p_blank) - fmax(0.0, cfg.beta));
      full_beam = (nb == beam);
    }
    auto pruned = [&](float lp_c, int q) -> bool { return full_beam && (lp_c + cur.score[q] < min_cutoff); };
    // alpha * ln P_lm(c | last order-1 words of hypothesis i): the scorer term of the extension (i, c)
    auto lm_term = [&](int i, int c) -> float {
      int32_t win[kLmMaxOrder];
      const int order = cfg.lm.order;
      for (int j = 0; j < order - 1; ++j) win[j] = cur.ctx[i * kLmCtx + (kLmCtx - (order - 1)) + j];
      win[order - 1] = cfg.lm.tok2lm[c];
      return (float)(lm_log_cond_prob(cfg.lm, win) * cfg.alpha);
    };
    // word-based scorer: alpha * ln P_lm(word | last order-1 WORDS of hypothesis i), `word` = LM index of the word a space
    // has just completed (ctc_beam_search_decoder.cpp scores `prefix`, not `prefix_new`, when c == space_id)
    auto lm_term_word = [&](int i, int word) -> float {
      int32_t win[kLmMaxOrder];
      const int order = cfg.lm.order;
      for (int j = 0; j < order - 1; ++j) win[j] = cur.ctx[i * kLmCtx + (kLmCtx - (order - 1)) + j];
      win[order - 1] = word;
      return (float)(lm_log_cond_prob(cfg.lm, win) * cfg.alpha);
    };
    // log-probability carried by the extension of hypothesis i with candidate k (its LM term included); `to_state`:
    // dictionary state the extension lands in (word-based scorer only)
    auto ext_logp = [&](int i, int k, int to_state) -> float {
      const int c = cand_c[k];
      float log_p = kNegInf;
      if (c == cur.chr[i]) { if (cur.b[i] > kNegInf) log_p = cand_lp[k] + cur.b[i]; }
      else log_p = cand_lp[k] + cur.score[i];
      if (word_lm) {
        if (c == space_id) {
          log_p += lm_term_word(i, cfg.lm.dict_word[to_state]);
          log_p = (float)((double)log_p + cfg.beta);
        }
      } else if (has_lm) {
        log_p += lm_term(i, c);
        log_p = (float)((double)log_p + cfg.beta);
      }
      return log_p;
    };
    TS(1);
    // ---- (d) contributions received by the hypotheses already in the beam ----
    const float lpb = lp_of(blank);
    bool merged = false;
    for (int q = tid; q < nb; q += BT) {
      const int cq = cur.chr[q];
      float bc = (lpb != kNotCand && !pruned(lpb, q)) ? lpb + cur.score[q] : kNegInf;
      float nbc = kNegInf;
      const float lq = (cq >= 0) ? lp_of(cq) : kNotCand;
      if (lq != kNotCand && cq != blank) {
        if (!pruned(lq, q)) nbc = lq + cur.nb[q];  // repeated character
        const int pn = cur.par[q];
        int pi = -1;
        for (int i = 0; i < nb; ++i)
          if (cur.node[i] == pn) pi = i;
        if (pi >= 0) {  // extension of the parent hypothesis by cq lands on this existing prefix
          // (word-based scorer: the word a space completes is read off the PARENT's dictionary state -- this prefix's own
          //  state may already have been reset to the start state by a failed look-up, see below)
          if (!pruned(lq, pi))
            nbc = lse(nbc, ext_logp(pi, kidx[cq], (word_lm && cq == space_id) ? lm_dict_arc(cfg.lm, cur.dst[pi], cq) : 0));
          exists[pi * C + kidx[cq]] = 1;
          merged = true;  // (distinct (parent, character) per hypothesis: one flag each)
        }
      }
      new_b[q] = bc;
      new_nb[q] = nbc;
      new_score[q] = lse(bc, nbc);
    }
    if (fast_ok && wave == 0) {  // (nb <= beam <= 16: the loop above ran in lanes of wave 0) merged children of the frame
      const unsigned long long m = __ballot(merged);
      if (lane == 0) sh_i[0] = __popcll(m);
    }
    if (fast_ok && wave == 1 && lane < nb) {  // an otherwise idle wave: rank of every hypothesis by its CURRENT score
      const float sq = cur.score[lane];       // (rows of the staircase, see (e'))
      int r = 0;
      for (int i = 0; i < nb; ++i) {
        const float si = cur.score[i];
        r += (si > sq || (si == sq && i < lane)) ? 1 : 0;
      }
      hyp_of_rank[r] = lane;
    }
    lds_barrier();
    if (word_lm) {
      // PathTrie::get_path_trie with a dictionary: a character that has no arc from the prefix's dictionary state yields no
      // child -- and when that state is FINAL (a word has just ended) the lookup resets the prefix's state to the start
      // state as a side effect.  Upstream walks the candidates in list order for every prefix, so for a prefix in a final
      // state the FIRST candidate that is looked up (not blank, not cut by min_cutoff, not an existing child) finds
      // nothing and resets the state; every later candidate of the frame is looked up from the start state.
      for (int q = tid; q < nb; q += BT) {
        int kr = -1, nd = cur.dst[q];
        if (lm_dict_final(cfg.lm, nd)) {
          for (int k = 0; k < C; ++k) {
            if (cand_c[k] == blank || exists[q * C + k] || pruned(cand_lp[k], q)) continue;
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
    TS(2);
    // ---- (e') staircase fast path: small beams, no scorer ----
    // Without a scorer the score of child (i, k) is at most U(i, k) = lp[k] + score[i], and U falls along both axes when
    // the hypotheses are taken in score order (rank r) and the candidates in list order (probability descending).  A child
    // with (r + 1)(k + 1) > beam has beam elements in front of it -- unless some of those are lowered (a repeated
    // character uses log P_b) or merged into an existing hypothesis.  So: rank only the existing hypotheses and the children
    // with k < K_r = beam / (r + 1) + margin (<= kFastCap slots instead of nb (1 + C) elements), each key against all others
    // with ballots (no histograms, no scans), and VERIFY afterwards that the best excluded child of every row, bounded by
    // U(i, K_r), is strictly below the last score taken.  If that fails -- or the list does not hold `beam` valid elements
    // -- the frame takes the general selection below; so the result is the general one bit for bit: same survivors, same
    // (element) order, same node ids.  (Tried and slower, tools/experiments/r05: the whole path in ONE wave without
    // barriers -- a lone wave issues a dependent instruction every ~9 cycles: 5.4 us per frame against 4.8.)
    int k_sel = 0;
    int* surv = nxt.par;  // temporary list in the next beam's `par` column: slot p is read, then overwritten, by thread p
    bool fast_done = false;
    if (fast_ok && C > 0) {
      const int has_blank = kidx[blank] >= 0 ? 1 : 0;
      const int n_valid_f = nb + nb * (C - has_blank) - sh_i[0];  // = n_valid of (e): existing + non-blank, non-merged children
      if (n_valid_f > beam) {  // (block-uniform)
#ifdef PPASR_BEAM_TS
        ++ts_att;
#endif
        if (tid < n_s0) {
          unsigned long long key = ~0ull;
          if (tid < beam) {
            if (tid < nb) key = make_key(new_score[tid], cur.chr[tid], tid);
          } else if (my_r < nb && my_k < C) {
            const int i = hyp_of_rank[my_r], k = my_k;
            const int c = cand_c[k];
            if (c != blank && !exists[i * C + k]) {
              const float lpk = cand_lp[k];
              float log_p = kNegInf;
              if (c == cur.chr[i]) { if (cur.b[i] > kNegInf) log_p = lpk + cur.b[i]; }
              else log_p = lpk + cur.score[i];
              key = make_key(log_p, c, nb + i * C + k);
            }
          }
          fkey[tid] = key;
        }
        if (tid < kFastBeam) srank_key[tid] = ~0ull;
        lds_barrier();
        TS(3);
        {  // rank of every key = number of smaller keys (keys are unique: the element id is part of them): wave w ranks the
           // keys w, w + NW, ...  (requesting all of a wave's keys before the first is used was measured slower: the
           //  unrolled form walks kFastCap / NW slots whatever the list holds)
          const unsigned long long k0 = lane < n_s0 ? fkey[lane] : ~0ull, k1 = lane + 64 < n_s0 ? fkey[lane + 64] : ~0ull;
          for (int i = wave; i < n_s0; i += BT / 64) {
            const unsigned long long ki = fkey[i];
            const int rank = __popcll(__ballot(k0 < ki)) + __popcll(__ballot(k1 < ki));
            if (lane == 0 && ki != ~0ull && rank < beam) srank_key[rank] = ki;
          }
        }
        lds_barrier();
        TS(5);
        // survivors in ELEMENT order (the order the general compaction leaves them in) + the verification
        if (wave == 0) {  // lane = (survivor p = lane & 15, quarter g = lane >> 4 of the others it is compared with)
          const int pidx = lane & 15, g = lane >> 4;
          const unsigned long long kp = pidx < beam ? srank_key[pidx] : ~0ull;
          const int ep = (int)(kp & 0x3FFFFull);  // (make_key: the element id is the low 18 bits)
          int pos = 0;
#pragma unroll
          for (int j = 0; j < kFastBeam / 4; ++j) {
            const int jj = g + 4 * j;
            const unsigned long long kj = jj < beam ? srank_key[jj] : ~0ull;
            pos += (kj != ~0ull && (int)(kj & 0x3FFFFull) < ep) ? 1 : 0;
          }
          pos += __shfl_xor(pos, 16);
          pos += __shfl_xor(pos, 32);
          if (lane < beam) {
            if (kp == ~0ull) {
              sh_i[1] = 1;  // fewer than `beam` valid elements in the list
            } else {
              surv[pos] = ep;
              surv_lp[pos] = score_of_key((uint32_t)(kp >> 32));
            }
          }
        } else if (wave == 1 && lane < nb) {  // (another wave: row r = lane)
          const unsigned long long kl = srank_key[beam - 1];
          if (my_row_k < C && kl != ~0ull) {
            const uint32_t thr = (uint32_t)(kl >> 32);  // score key of the last element taken
            const float ub = cand_lp[my_row_k] + cur.score[hyp_of_rank[lane]];
            if (desc_key(ub) <= thr) sh_i[1] = 1;  // an excluded child could score >= the last one taken
          }
        }
        lds_barrier();
        if (sh_i[1] == 0) {
          fast_done = true;
          k_sel = beam;
#ifdef PPASR_BEAM_TS
          ++ts_ok;
#endif
        }
        TS(6);
      }
    }
    if (!fast_done) {
      // ---- (e) element space: [0,nb) existing hypotheses, nb + i*C + k = child (i, cand k).  The 32-bit score key of
      // every element is computed ONCE into LDS (0xFFFFFFFF = not a candidate); thread t owns the contiguous range
      // [t*per, (t+1)*per) so that the compaction below keeps element order with a single block scan ----
      for (int i = tid; i < 8 * 256; i += BT) hist[i] = 0;  // 4 + 4 per-pass histograms of the two selects below (complete
                                                            // behind the barrier of the scan that closes (e))
      const int N = nb + nb * C;
      // (an ODD range length: thread t starts at word t*per of skey[], and an even stride would put the 64 lanes of a
      //  wave on 16 or fewer of the 64 LDS banks)
      const int per = ((N + BT - 1) / BT) | 1;
      const int e_lo = min(tid * per, N), e_hi = min(e_lo + per, N);
      auto elem_char = [&](int e) -> int { return e < nb ? cur.chr[e] : cand_c[(e - nb) % C]; };
      int my_valid = 0;
      {
        int e = e_lo;
        for (; e < e_hi && e < nb; ++e) {  // hypotheses already in the beam
          skey[e] = desc_key(new_score[e]);
          ++my_valid;
        }
        while (e < e_hi) {  // children: one hypothesis i at a time (its fields stay in registers), candidates k0..k1
          const int r = e - nb, i = r / C, k0 = r - i * C;
          const int k1 = min(C, k0 + (e_hi - e));
          const int ci = cur.chr[i];
          const float bi = cur.b[i], si = cur.score[i];
          // word-based scorer: dictionary state the children of i are looked up from (after the reset above), and the one
          // candidate that triggered the reset
          const int di = word_lm ? new_dst[i] : 0, kri = word_lm ? k_reset[i] : -1;
          const bool dead = word_lm && lm_dict_final(cfg.lm, di);  // still final: no candidate was looked up this frame
          for (int k = k0; k < k1; ++k, ++e) {
            const int c = cand_c[k];
            const float lpk = cand_lp[k];
            uint32_t key = 0xFFFFFFFFu;
            if (c != blank && !exists[e - nb] && !(full_beam && (lpk + si < min_cutoff))) {
              int to = 0;
              bool ok = true;
              if (word_lm) {
                to = (dead || k == kri) ? -1 : lm_dict_arc(cfg.lm, di, c);
                ok = to >= 0;
              }
              if (ok) {
                float log_p = kNegInf;
                if (c == ci) { if (bi > kNegInf) log_p = lpk + bi; }
                else log_p = lpk + si;
                if (word_lm) {
                  if (c == space_id) {
                    log_p += lm_term_word(i, cfg.lm.dict_word[to]);
                    log_p = (float)((double)log_p + cfg.beta);
                  }
                } else if (has_lm) {
                  log_p += lm_term(i, c);
                  log_p = (float)((double)log_p + cfg.beta);
                }
                key = desc_key(log_p);
                ++my_valid;
              }
            }
            skey[e] = key;
          }
        }
      }
      int n_valid;
      (void)block_excl_scan<BT / 64>(my_valid, wave_tot, n_valid);  // (barrier inside: skey[] is complete afterwards)
      k_sel = n_valid >= beam ? beam : n_valid;
      TS(3); TSN();
      // ---- (f) exact top-k_sel in prefix_compare order = ascending (score key, char, element id): MSD radix select of
      // the k_sel-th smallest 32-bit score key over the LDS array; if the threshold class has more members than slots
      // left (ties), a second select over (char, id) inside that class ----
      // 4-pass radix select of the k-th smallest value of f(e) over elements with pred(e); returns the value and how
      // many members of its class are needed (k_need) / exist (k_have)
      auto radix_select32 = [&](auto&& value_of, int k, int* hists, uint32_t& out, int& k_need, int& k_have) {
        uint32_t prefix = 0;
        int k_rem = k;
        k_have = 0;
        for (int pass = 0; pass < 4; ++pass) {
          const int shift = 24 - 8 * pass;
          const uint32_t hi_mask = pass == 0 ? 0u : (~0u << (shift + 8));
          int* h = hists + pass * 256;  // zeroed at the start of the frame; one histogram per pass = one barrier per pass
          {
            RunHist rh(h);
            for (int e = tid; e < N; e += BT) {
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
      uint32_t thr1 = 0xFFFFFFFEu, thr2 = 0xFFFFFFFFu;  // keep: key < thr1, or key == thr1 and (char, id) <= thr2
      bool exact_class = false;                        // thr1 names one exact key value whose class is only partly taken
      if (k_sel < n_valid) {
        int need, have;
        radix_select32([&](int e, uint32_t& v) { v = skey[e]; return v != 0xFFFFFFFFu; }, k_sel, hist, thr1, need, have);
        // after an early exit thr1 is an upper bound of a wholly taken class; after 4 passes it is an exact key value
        if (need < have) {
          exact_class = true;
          const uint32_t eq = thr1;
          int n2, h2;
          radix_select32([&](int e, uint32_t& v) {
            if (skey[e] != eq) return false;
            v = ((uint32_t)(elem_char(e) + 1) << 18) | (uint32_t)e;
            return true;
          }, need, hist + 4 * 256, thr2, n2, h2);
        }
      }
      TS(5);
      auto keeps = [&](int e) -> bool {
        const uint32_t v = skey[e];
        if (v == 0xFFFFFFFFu) return false;
        if (!exact_class) return v <= thr1;
        if (v != thr1) return v < thr1;
        return (((uint32_t)(elem_char(e) + 1) << 18) | (uint32_t)e) <= thr2;
      };
      // ---- (g) ordered compaction: survivors' element ids in element order (one block scan over per-thread counts), then
      // slot p of the next beam is materialised by thread p (the survivors of one thread's range can be many) ----
      int my_keep = 0;
      unsigned long long keep_bits = 0;  // verdicts of the first 64 elements of this thread's range, evaluated once
      for (int e = e_lo; e < e_hi; ++e) {
        const bool k = keeps(e);
        my_keep += k ? 1 : 0;
        if (e - e_lo < 64) keep_bits |= (unsigned long long)(k ? 1 : 0) << (e - e_lo);
      }
      int tot_keep;
      int wpos = block_excl_scan<BT / 64>(my_keep, wave_tot + BT / 64, tot_keep);
      for (int e = e_lo; e < e_hi; ++e) {
        const bool k = (e - e_lo < 64) ? (((keep_bits >> (e - e_lo)) & 1ull) != 0) : keeps(e);
        if (!k || wpos >= beam) continue;
        surv_lp[wpos] = e >= nb ? score_of_key(skey[e]) : 0.f;  // the extension's log-probability, computed once in (e)
        surv[wpos++] = e;
      }
      lds_barrier();
      TS(6);
    }
    int n_nodes_next;
    // node ids of the new prefixes: looked up in the node table first (a prefix that was in the beam before keeps its
    // identity, ctc_beam.h), misses get fresh ids in slot order (one block scan) and are entered into the table
    int my_parent = -1, my_char = 1, my_id = -1;  // (k_sel <= beam <= BT: one slot per thread)
    if (!cfg.node_table) {
      n_nodes_next = n_nodes + k_sel;  // every new prefix takes the id of its slot
    } else {
      const int pos = tid;
      int miss = 0;
      if (pos < k_sel && surv[pos] >= nb) {
        const int r = surv[pos] - nb, i = r / C;
        my_parent = cur.node[i];
        my_char = cand_c[r - i * C];
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
      const int before = block_excl_scan<BT / 64>(miss, wave_tot, n_miss);
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
    for (int pos = tid; pos < k_sel; pos += BT) {
      const int e = surv[pos];
      if (e < nb) {
        nxt.node[pos] = cur.node[e]; nxt.chr[pos] = cur.chr[e]; nxt.par[pos] = cur.par[e];
        nxt.b[pos] = new_b[e]; nxt.nb[pos] = new_nb[e]; nxt.score[pos] = new_score[e];
        for (int j = 0; j < kLmCtx; ++j) nxt.ctx[pos * kLmCtx + j] = cur.ctx[e * kLmCtx + j];
        nxt.dst[pos] = word_lm ? new_dst[e] : 0;
      } else {
        const int r = e - nb, i = r / C, kk = r - i * C;
        const int c = cand_c[kk];
        const float log_p = surv_lp[pos];  // the extension's log-probability, computed once in (e) / (e')
        const int id = cfg.node_table ? my_id : n_nodes + pos;  // (with the table: pos == tid)
        if (!cfg.node_table && id < cfg.max_nodes) { arena[kArenaWords * (size_t)id] = cur.node[i]; arena[kArenaWords * (size_t)id + 1] = c; }
        if (word_lm) {
          // the LM context holds WORDS: it moves on when a space completes one.  The dictionary state belongs to the trie
          // node: a new node starts where the arc leads, a revived one is where it was left (a final state may have been
          // reset to the start state by a failed look-up while the prefix was alive)
          const int to = lm_dict_arc(cfg.lm, new_dst[i], c);
          if (my_char) {
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
      }
    }
    for (int k = tid; k < C; k += BT) kidx[cand_c[k]] = -1;  // reset for the next frame
    lds_barrier();
    TS(7);
    nb = k_sel;
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
  if (tid == 0 && u == 0 && n_frames > 0)
    printf("beam ts (x10ns/frame): inst %lld lm %lld contrib %lld keys %lld sel %lld keep %lld mat %lld | N %lld C %lld nb %lld frames %d | fast path: attempted %lld taken %lld\n",
           ts_acc[0] / n_frames, ts_acc[1] / n_frames, ts_acc[2] / n_frames, ts_acc[3] / n_frames, ts_acc[5] / n_frames,
           ts_acc[6] / n_frames, ts_acc[7] / n_frames, ts_n / n_frames, ts_c / n_frames, ts_nb / n_frames, n_frames, ts_att, ts_ok);
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
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
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
__device__ __forceinline__ int mbcnt(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
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
                           int32_t* out_lens, double* out_scores, int32_t* status, hipStream_t st) {
  const size_t lds = beam_lds_bytes(cfg), plds = prune_lds_bytes(cfg);
  // threads per utterance by the number of (hypothesis, candidate) elements of a frame: every phase is a chain of
  // block-wide steps, and a barrier over few waves is cheaper than one over 16
  const int n_elem = cfg.beam * (1 + cfg.n_cand_max);
  // 512 threads up to 1 024 elements (and beams of at most 512: the new beam is materialised one slot per thread), else 1 024
  const int sel = (n_elem <= 1024 && cfg.beam <= 512) ? 0 : 1;
  const bool wl = cfg.lm.order > 0 && cfg.lm.word_based != 0;
  const void* fns[2][2] = {
      {reinterpret_cast<const void*>(k_ctc_beam<512, false>), reinterpret_cast<const void*>(k_ctc_beam<1024, false>)},
      {reinterpret_cast<const void*>(k_ctc_beam<512, true>), reinterpret_cast<const void*>(k_ctc_beam<1024, true>)}};
  const void* fn = fns[wl ? 1 : 0][sel];
  // hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set it on every launch (a few host
  // microseconds) rather than caching "already configured" in process-wide statics, which left a second GPU used from
  // the same process unconfigured and was not thread-safe
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (plds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ctc_prune<kPruneThreads>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
    if (e != hipSuccess) return e;
  }
  if (T > 0)
    PPASR_LAUNCH(k_ctc_prune<kPruneThreads>, dim3(T, B), dim3(kPruneThreads), plds, st, probs, frame_lens, T, cfg,
                       prune_recs);
  // small beams: one wave per utterance.  OPT-IN for now (PPASR_BEAM_WAVE=1): measured 11 us per frame at beam 10 against
  // 5.4 us for the block-wide kernel (a lone wave issues its readlane / ballot / branch sequences at ~9 cycles per
  // instruction); kept because it needs no LDS tables and co-resides with the encoder's workgroups.
  const char* wave_env = getenv("PPASR_BEAM_WAVE");
  if (cfg.beam <= kWaveBeamMax && cfg.n_cand_max <= 64 && wave_env && atoi(wave_env) == 1 && !cfg.lm.word_based) {
    if (cfg.lm.order > 0)
      PPASR_LAUNCH((k_ctc_beam_wave<kWaveBeamMax, true>), dim3(B), dim3(64), 0, st, frame_lens, T, cfg, prune_recs, state,
                   init_state, finalize, out_tokens, out_lens, out_scores, status);
    else
      PPASR_LAUNCH((k_ctc_beam_wave<kWaveBeamMax, false>), dim3(B), dim3(64), 0, st, frame_lens, T, cfg, prune_recs, state,
                   init_state, finalize, out_tokens, out_lens, out_scores, status);
    return hipGetLastError();
  }
#define PPASR_LAUNCH_BEAM(BT)                                                                                          \
  do {                                                                                                                 \
    if (wl)                                                                                                            \
      PPASR_LAUNCH((k_ctc_beam<BT, true>), dim3(B), dim3(BT), lds, st, probs, frame_lens, T, cfg, prune_recs, state,   \
                   init_state, finalize, out_tokens, out_lens, out_scores, status);                                    \
    else                                                                                                               \
      PPASR_LAUNCH((k_ctc_beam<BT, false>), dim3(B), dim3(BT), lds, st, probs, frame_lens, T, cfg, prune_recs, state,  \
                   init_state, finalize, out_tokens, out_lens, out_scores, status);                                    \
  } while (0)
  if (sel == 0) PPASR_LAUNCH_BEAM(512);
  else PPASR_LAUNCH_BEAM(1024);
#undef PPASR_LAUNCH_BEAM
  return hipGetLastError();
}

}  // namespace ppasr
