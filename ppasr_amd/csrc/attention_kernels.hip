// attention_kernels.hip -- relative-position multi-head attention as a stand-alone launch (the two-kernel route:
// ragged batches, Squeezeformer, grouped attention of the Efficient-Conformer, streaming chunks, the general layer route).
//
// Reference: RelPositionMultiHeadedAttention.forward + forward_attention (conformer/attention.py:198-262, 86-126;
// squeezeformer/attention.py:96-162) and GroupedRelPositionMultiHeadedAttention (efficient_conformer/attention.py:40-79,
// 128-193).  scores = ((q+u) k^T + (q+v) p^T) / sqrt(dk) -- rel_shift is disabled in the reference (:256-258) -- is one
// contraction over the concatenated operands Q' = [q+u | q+v], K' = [k | p]; key-padding mask from the lengths (token j
// masked iff mask_mul * j >= len[b], subsampling.py:115), softmax, masked probabilities -> 0, times V.
//
// Design: the barrier-free transposed flash attention of k_attn_out_glu (conformer_kernels.hip) with the keys of a
// (utterance, head, 32-query block) split over the NW waves of the workgroup (sub-blocks w, w + NW, ...: a 750-key
// utterance is not one 60 us chain).  Every wave is an independent worker on
// TRANSPOSED score tiles S^T = K' Q'^T: the K' fragment (one key row per lane, straight from L2 in whole-line bursts)
// is the MFMA A operand, the Q' fragment (LDS) the B operand, so a lane owns ITS query row's scores -- row max / sum are
// in-lane (v_max3, packed math, one lane^32 exchange each), the probabilities never leave the accumulator registers (they
// are the B operand of O^T += V^T P^T) and the running rescale is a per-lane multiply.  V comes row-major: a lane loads
// two adjacent value columns of one key (8 bytes; 256 contiguous bytes per key row and wave), so the output tiles hold
// the even / odd columns of a 64-column chunk.  The waves' partial (max, sum, O^T) are merged through LDS at the end
// (flash-decoding), the merged rows stored with whole-row coalesced writes.
//
// (The round-1 kernel k_attention it replaces: waves of a workgroup sharing one LDS score tile, three barriers per key
//  block, one dword of V per lane per MFMA: 0.13 - 0.28 of the fp32-MFMA rate on BASELINE configs[3] / [4].)
#include <cstdlib>

#include "conformer_kernels.h"
#include "launch.h"
#include "phases_t.h"

#include <math.h>

namespace ppasr {

// tuning knobs (tools/build_variant.sh NAME -DPPASR_ATTN_...): V k-groups in flight, waves per SIMD the register allocation
// leaves room for, 32-key tiles per sub-block of the plain heads.  Measured on BASELINE configs[4] (same box, ms per step of
// the 12 attention launches): 64-key sub-blocks / 2 waves per SIMD / 8 V k-groups in flight 0.645; 32-key / 3 / 4: 0.637;
// 32-key / 4 / 2 (the defaults): 0.591.  Splitting the keys of a query block over FEWER waves for short utterances
// (1 / 2 / 4 by key count, the spare waves taking other query blocks) was measured and is slower -- 0.985 (adaptive),
// 1.06 (never split), 0.74 (two), 0.66 (always four): a launch is bounded by its longest chains, not by its staging.
#ifndef PPASR_ATTN_PQ
#define PPASR_ATTN_PQ 2
#endif
#ifndef PPASR_ATTN_OCC
#define PPASR_ATTN_OCC 4
#endif
#ifndef PPASR_ATTN_NW
#define PPASR_ATTN_NW 4  // waves = key splits per workgroup of the plain heads
#endif
#ifndef PPASR_ATTN_NW192
#define PPASR_ATTN_NW192 2  // ... of the grouped heads (d_k = 192); cfg4, same box, the 4 launches of a step: 2 waves 0.166 ms, 3: 0.197, 4: 0.186
#endif
#ifndef PPASR_ATTN_QB192
#define PPASR_ATTN_QB192 3  // query blocks per workgroup of the grouped heads (1: the one-block form, two waves)
#endif
#ifndef PPASR_ATTN_NT
#define PPASR_ATTN_NT 1  // 32-key tiles per sub-block of the plain heads
#endif

template <int DK>
struct AttnT {
  static constexpr int G = DK / 64;            // frames per token: 1, or group_size = pad4group's re-cut of 2 / 3 / 4 frames into
                                               // heads of 128 / 192 / 256 (3: the shipped configuration, tuned below; 2 and 4: the
                                               // constructor's other values, the plain one-block form)
  static constexpr int NC2 = DK / 64;          // 64-column chunks of the context (two output tiles each: even / odd columns)
  static constexpr int NGK = 2 * DK / 8;       // 8-wide k-groups of the score contraction over K' = [k | p]
  static constexpr int NT = DK == 64 ? PPASR_ATTN_NT : 1;  // 32-key tiles per sub-block (accumulator budget)
  static constexpr int NW = DK == 64 ? PPASR_ATTN_NW : (DK == 192 ? PPASR_ATTN_NW192 : 2);  // waves per workgroup = key splits of its query block
  static constexpr int MAXQ = 1;               // (historic: query blocks per wave set)
  // Query blocks per workgroup.  Plain heads: one, its NW waves split the keys.  Grouped heads (d_k = 192): THREE on eight
  // waves -- at 222 registers a SIMD holds two waves, and a (query block, 32-key sub-block) unit is 288 MFMAs that keep a
  // SIMD's matrix pipe busy on their own, so what matters is how the units are dealt to the SIMDs: with one query block
  // per two-wave workgroup, BASELINE configs[3]'s 83 tokens (three query blocks x three sub-blocks per (utterance, head))
  // put up to four units on one SIMD of a CU and one on another (41 us per launch, per-phase stamps); one workgroup per
  // (utterance, head) with waves = (block 0: three key splits, block 1: three, block 2: two) puts 2 / 2 / 3 / 2 units
  // on the four SIMDs whatever the dispatcher does
  static constexpr int QB = DK == 192 ? PPASR_ATTN_QB192 : 1;
  static constexpr int WV = QB == 1 ? NW : 8;  // waves per workgroup
  static constexpr int NSMAX = QB == 1 ? NW : 3;
  static __device__ __forceinline__ int wave_qi(int w) { return QB == 1 ? 0 : (w < 3 ? 0 : (w < 6 ? 1 : 2)); }
  static __device__ __forceinline__ int wave_ks(int w) { return QB == 1 ? w : (w < 6 ? w % 3 : w - 6); }
  static __device__ __forceinline__ int qb_ns(int qi) { return QB == 1 ? NW : (qi < 2 ? 3 : 2); }  // key splits of block qi
  static __device__ __forceinline__ int qb_w0(int qi) { return QB == 1 ? 0 : 3 * qi; }             // its first wave
  static constexpr int PQ = DK == 64 ? PPASR_ATTN_PQ : 2;  // V k-groups in flight
  static constexpr int QLD = 2 * DK + 4;       // Q' row stride (floats)
  static constexpr int OLD = DK + 4;           // partial O row stride
  static constexpr int SB = 32 * NT;           // keys per sub-block
  // a wave parks its partial O^T in slot `wave`: its own Q' tile where every wave has one (NS == 1 then needs no barrier)
  static constexpr int PSTR = 32 * OLD;
  static constexpr int TILE_FLOATS = (QB * 32 * QLD > NSMAX * PSTR) ? QB * 32 * QLD : NSMAX * PSTR;
  static constexpr int LDS_FLOATS = TILE_FLOATS + WV * 64;
};

__device__ __forceinline__ f32x2 load_b64(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}

// (second launch bound = waves per SIMD the register allocation must leave room for: 226 / 224 registers, so that two
//  workgroups of plain heads / three of grouped heads share a CU)
template <int DK>
__global__ __launch_bounds__(64 * AttnT<DK>::WV, DK == 64 ? PPASR_ATTN_OCC : 2) void k_attention_t(AttnArgs a, int B, int H) {
  using C = AttnT<DK>;
  constexpr int G = C::G, NT = C::NT, NC2 = C::NC2, NGK = C::NGK, PQ = C::PQ, QB = C::QB, WV = C::WV;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                       // QB tiles [32][QLD] Q' = [q+u | q+v]; after the key loop: partial O tiles [32][OLD]
  float* Stat = smem + C::TILE_FLOATS;    // [WV][2][32]: running max, running sum of each wave's key range
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  const int hh = lane >> 5, l31 = lane & 31;
  // XCD-aware block -> (utterance, head, query block) map.  Workgroups go round-robin to the 8 XCDs (linear id % 8), each
  // with its own L2: all query blocks of the pair (b, h) run on XCD (b * H + h) % 8, so that pair's key / value / position
  // columns are fetched into ONE L2 instead of eight, and consecutive utterances (RaggedPlan sorts them by length)
  // alternate between the XCDs' pair lists, which keeps a ragged batch balanced.
  const int nq = ((a.T1 + 31) / 32 + QB - 1) / QB;  // workgroups per (utterance, head): groups of QB query blocks
  const int slot = blockIdx.x >> 3;
  // (boustrophedon over the groups of 8 pairs: with utterances sorted by length, XCD x gets pairs x, 15 - x, 16 + x, ...
  //  instead of always the longer half of every group)
  const int grp8 = slot / nq, x8 = blockIdx.x & 7;
  const int pair = __builtin_amdgcn_readfirstlane(grp8 * 8 + ((grp8 & 1) ? 7 - x8 : x8));
  const int b = pair / H, h = pair - b * H;
  const int li = slot % nq;  // this workgroup's index among the pair's workgroups
  if (b >= B) return;
  const int T1 = a.T1, F1 = a.q_frames;
  int T2 = a.T2, F2 = a.kv_frames;
  const int dm = a.dm;
  const float* __restrict__ qb = a.q + (size_t)b * F1 * a.q_stride;
  const float* __restrict__ kbp = a.k + (size_t)b * F2 * a.k_stride;
  const float* __restrict__ vbp = a.v + (size_t)b * F2 * a.v_stride;
  const float* __restrict__ ptab = a.ptab + (size_t)a.pos0 * dm;
  if (a.sess) {  // multi-session streaming: per-session cache slot, length and position
    const SessDesc d = a.sess[b];
    T2 = F2 = d.cache_t + T1;
    kbp = a.k + (size_t)d.sess * a.sess_stride;
    vbp = a.v + (size_t)d.sess * a.sess_stride;
    ptab = a.ptab + (size_t)d.pos0 * dm;
  }
  float* __restrict__ ctx = a.ctx + (size_t)b * F1 * dm;
  const int my_qi = C::wave_qi(wave);     // this wave's query block among the workgroup's QB
  const int ks = C::wave_ks(wave);        // its key split: sub-blocks ks, ks + NS, ks + 2 NS, ...
  const int NS = C::qb_ns(my_qi);         // key splits of that query block
  const int q0g = li * QB * 32;           // the workgroup's first query token
  const int q0 = q0g + my_qi * 32;        // this wave's query block
  // grouped heads: flat feature c (< G * dm) of token tok = (frame G * tok + c / dm, feature c % dm); any width
  auto split = [&](int tok, int c, int& frame, int& feat) {
    const int k = (c >= dm) + (c >= 2 * dm) + (c >= 3 * dm);
    frame = G * tok + k;
    feat = c - k * dm;
  };
  // Everything the first MFMA needs is requested BEFORE the utterance's length is looked at: the length load, the query
  // rows and the first K' burst are then one memory round trip instead of three dependent ones (the launches of a ragged
  // batch's reduced layers are a handful of sub-blocks per wave: their prologue was as long as their key loop).
  // ---- Q' = [q + pos_bias_u | q + pos_bias_v] of the block's 32 query tokens -> LDS ----
  float* Qt = Qs + my_qi * 32 * C::QLD;
  for (int idx = tid; idx < QB * 32 * (DK / 4); idx += 64 * WV) {
    const int row = idx / (DK / 4), f4 = idx - row * (DK / 4);  // row: query token q0g + row (tile row / 32)
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    if (q0g + row < T1) {
      int frame = q0g + row, feat = h * DK + 4 * f4;
      if (G != 1) split(q0g + row, h * DK + 4 * f4, frame, feat);
      if (frame < F1) q = *reinterpret_cast<const f32x4*>(qb + (size_t)frame * a.q_stride + feat);
    }
    const f32x4 u = *reinterpret_cast<const f32x4*>(a.pos_u + h * DK + 4 * f4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.pos_v + h * DK + 4 * f4);
    *reinterpret_cast<f32x4*>(Qs + row * C::QLD + 4 * f4) = q + u;
    *reinterpret_cast<f32x4*>(Qs + row * C::QLD + DK + 4 * f4) = q + v;
  }

  // ---- operand resources: one per 64-feature chunk (a chunk never straddles a frame of the grouped re-cut).  Chunk c3
  // of head h starts at flat feature c = h * DK + 64 c3 = (frame offset fo, feature feat0) of a token's G frames.  Keys /
  // positions: bounded by the key FRAMES of the call (rows past them -- the zero-padded tail group -- read zeros; keys
  // behind the utterance's valid ones hold whatever the batch holds there and are masked by index after the score MFMAs) ----
  const int krow_b = G * a.k_stride * 4, prow_b = G * a.pos_stride * dm * 4, vrow_b = G * a.v_stride * 4;
  __amdgpu_buffer_rsrc_t rs_k[NC2], rs_p[NC2], rs_v[NC2];
  int fo_c[NC2], feat_c[NC2];
#pragma unroll
  for (int c3 = 0; c3 < NC2; ++c3) {
    fo_c[c3] = 0;
    feat_c[c3] = h * DK + 64 * c3;
    if (G != 1) split(0, h * DK + 64 * c3, fo_c[c3], feat_c[c3]);
    const long long rows = (long long)F2 - fo_c[c3] - 1;  // last row relative to the chunk's base row
    rs_k[c3] = buf_rsrc(kbp + (size_t)fo_c[c3] * a.k_stride + feat_c[c3], rows < 0 ? 0 : (size_t)rows * a.k_stride * 4 + 256);
    rs_p[c3] = buf_rsrc(ptab + (size_t)fo_c[c3] * a.pos_stride * dm + feat_c[c3], rows < 0 ? 0 : (size_t)rows * a.pos_stride * dm * 4 + 256);
  }
  // byte offsets of this lane's key (tile t) of the sub-block at u0, in K and in the positional table
  int vk[NT], vp[NT];
  auto key_offsets = [&](int u0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int key = min(u0 + 32 * t + l31, T2 - 1);  // (keys >= kv_end are masked afterwards)
      vk[t] = key * krow_b + 16 * hh;
      vp[t] = key * prow_b + 16 * hh;
    }
  };
  // K' fragment of k-group gk (features 8 gk + 4 hh .. +3 of [k | p]) of this lane's key of tile t
  auto kfrag = [&](int t, int gk) -> f32x4 {
    return gk < NGK / 2 ? wstream_load(rs_k[gk >> 3], vk[t], (gk & 7) * 32)
                        : wstream_load(rs_p[(gk - NGK / 2) >> 3], vp[t], ((gk - NGK / 2) & 7) * 32);
  };
  // K' operands in bursts of 4 k-groups (one whole 128-byte line of each key row), double buffered
  f32x4 kq[2][4][NT];
  auto load_sg = [&](int buf, int sg) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) kq[buf][i][t] = kfrag(t, 4 * sg + i);
  };
  const float* qfrag_p = Qt + l31 * C::QLD + 4 * hh;
  auto qfrag = [&](int gk) -> f32x4 { return *reinterpret_cast<const f32x4*>(qfrag_p + 8 * gk); };
  key_offsets(ks * C::SB);
  load_sg(0, 0);  // (a wave whose first sub-block lies behind the valid keys requested 8 lines for nothing)

  // ---- now the length: valid keys, needed queries ----
  const int64_t len_b = a.lens ? a.lens[b] : (int64_t)a.mask_mul * T2;
  int q_need = T1;  // query tokens that are computed
  if (a.pad_skip > 0 && a.lens) {  // ragged batch: query blocks behind the needed frames are not computed
    const int64_t lb = len_b > 0 ? len_b : 0;
    const int fmul = a.mask_mul / G;  // frame f is valid iff fmul * f < len
    const int need_frames = (int)min((int64_t)F1, (lb + fmul - 1) / fmul + (a.pad_skip - 1));
    q_need = min(T1, (need_frames + G - 1) / G);
  }
  if (q0g >= q_need) return;  // (workgroup-uniform: before any barrier)
  const bool active = q0 < q_need;  // (QB > 1: a wave whose query block is not needed only takes part in the barriers)
  // keys >= kv_end are PAD (mask_mul * j >= len) or beyond the key tokens
  const int kv_end = (int)min((int64_t)T2, max((int64_t)0, (len_b + a.mask_mul - 1) / a.mask_mul));
  const int nfr = min(G * kv_end, F2);  // frames behind the valid tokens: the VALUES are bounded by them (p = 0 times an
                                        // uninitialised row of a ragged batch would be NaN; out of range reads 0)
  const int n_sb = active ? (kv_end + C::SB - 1) / C::SB : 0;
#pragma unroll
  for (int c3 = 0; c3 < NC2; ++c3) {
    const long long rows = (long long)nfr - fo_c[c3] - 1;
    rs_v[c3] = buf_rsrc(vbp + (size_t)fo_c[c3] * a.v_stride + feat_c[c3], rows < 0 ? 0 : (size_t)rows * a.v_stride * 4 + 256);
  }
  constexpr float kScale = (DK == 64 ? 0.125f : DK == 128 ? 0.08838834764831845f : DK == 192 ? 0.07216878364870322f : 0.0625f) *
                           1.4426950408889634f;  // 1/sqrt(d_k * group) * log2(e)
  f32x16 acc_o[2 * NC2];  // O^T: acc_o[2 c3 + e][r] = O[query l31][64 c3 + 2 ((r&3) + 8(r>>2) + 4hh) + e]
#pragma unroll
  for (int t = 0; t < 2 * NC2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // raw-score running max / running sum of query row l31 (same in both lane halves)
  bool first = true;
  const int vlane = 4 * hh * vrow_b + 8 * l31;
  __syncthreads();  // Q' is in LDS

  for (int sb = ks; sb < n_sb; sb += NS) {
    const int u0 = sb * C::SB;
    const bool edge = u0 + C::SB > kv_end;  // the sub-block holds masked keys (wave-uniform)
    // V^T operands: k-group q = (tile t = q >> 2, i = q & 3) covers keys u0 + 32t + 8i + 4hh + j, j = 0..3; a lane holds
    // value columns 2 l31, 2 l31 + 1 of each 64-column chunk for its key (requested before the score MFMAs of the
    // k-groups that hide them, consumed after the softmax)
    f32x2 ringv[PQ][4][NC2];
    auto vload = [&](int slot_, int q) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int vo = vlane + (u0 + 32 * (q >> 2) + 8 * (q & 3) + j) * vrow_b;
#pragma unroll
        for (int c3 = 0; c3 < NC2; ++c3) ringv[slot_][j][c3] = load_b64(rs_v[c3], vo, 0);
      }
    };
    // ---- S^T = K' Q'^T for the sub-block's keys ----
    f32x16 acc_s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_s[t][r] = 0.f;
    {
      constexpr int NSG = NGK / 4;
      f32x4 q_cur = qfrag(0), q_nxt = q_cur;
#pragma unroll
      for (int sg = 0; sg < NSG; ++sg) {
        if (sg + 1 < NSG) load_sg((sg + 1) & 1, sg + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int gk = 4 * sg + i;
          if (gk + 1 < NGK) q_nxt = qfrag(gk + 1);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc_s[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kq[sg & 1][i][t][j], q_cur[j], acc_s[t], 0, 0, 0);
          q_cur = q_nxt;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      static_assert(NSG % 2 == 0, "the K' double buffer keeps its parity across sub-blocks");
    }
#pragma unroll
    for (int q = 0; q < PQ; ++q) vload(q, q);
    // ---- online softmax in registers ----
    if (edge) {
      // element r of tile t is key u0 + 32t + (r&3) + 8(r>>2) + 4hh: masked iff that is >= kv_end
      const int hi = kv_end - u0 - 4 * hh;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * t + (r & 3) + 8 * (r >> 2) >= hi) acc_s[t][r] = -INFINITY;
    }
    float bm = max3f(acc_s[0][0], acc_s[0][1], acc_s[0][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) bm = max3f(bm, acc_s[0][r], acc_s[0][r + 1]);
    bm = fmaxf(bm, acc_s[0][15]);
#pragma unroll
    for (int t = 1; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) bm = max3f(bm, acc_s[t][r], acc_s[t][r + 1]);
    }
    bm = fmaxf(bm, __shfl_xor(bm, 32));
    const float m_new = fmaxf(m_run, bm);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;  // every key so far masked: p = 0, alpha irrelevant (O = 0)
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_safe) * kScale);
    const float mb = -m_safe * kScale;
    f32x2 ps2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 e = f32x2{acc_s[t][r], acc_s[t][r + 1]} * f32x2{kScale, kScale} + f32x2{mb, mb};  // v_pk_fma_f32
        acc_s[t][r] = __builtin_amdgcn_exp2f(e[0]);
        acc_s[t][r + 1] = __builtin_amdgcn_exp2f(e[1]);
        ps2 += f32x2{acc_s[t][r], acc_s[t][r + 1]};
      }
    float ps = ps2[0] + ps2[1];
    ps += __shfl_xor(ps, 32);
    m_run = m_new;
    l_run = l_run * alpha + ps;
    if (sb + NS < n_sb) {  // next sub-block's first K' super-group: in flight across the PV phase
      key_offsets(u0 + NS * C::SB);
      load_sg(0, 0);
    }
    // ---- O^T = O^T * alpha + V^T P^T ----
    if (!first) {
#pragma unroll
      for (int ct = 0; ct < 2 * NC2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 o = f32x2{acc_o[ct][r], acc_o[ct][r + 1]} * f32x2{alpha, alpha};  // v_pk_mul_f32
          acc_o[ct][r] = o[0];
          acc_o[ct][r + 1] = o[1];
        }
    }
    first = false;
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) {
      // (keys >= kv_end meet p = 0 exactly and read zeros -- out of the resources' range -- so no select is needed)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pj = acc_s[q >> 2][4 * (q & 3) + j];
#pragma unroll
        for (int c3 = 0; c3 < NC2; ++c3) {
          const f32x2 v2 = ringv[q % PQ][j][c3];
          acc_o[2 * c3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2[0], pj, acc_o[2 * c3], 0, 0, 0);
          acc_o[2 * c3 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2[1], pj, acc_o[2 * c3 + 1], 0, 0, 0);
        }
      }
      if (q + PQ < 4 * NT) vload(q % PQ, q + PQ);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- merge the key splits: O = sum_s O_s e^{m_s - m} / sum_s l_s e^{m_s - m}; rows stored coalesced.  One round per
  // query block of the workgroup: its waves park their partial O^T in tile slots 0 .. NS - 1 (over the Q' tiles, which every
  // wave is done with), then all threads merge and store that block's rows ----
  __syncthreads();  // every wave is done with the Q' tiles
#pragma unroll
  for (int qbi = 0; qbi < QB; ++qbi) {
    if (my_qi == qbi) {
      float* Pt = Qs + ks * C::PSTR + l31 * C::OLD;
#pragma unroll
      for (int c3 = 0; c3 < NC2; ++c3)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x16 &o0 = acc_o[2 * c3], &o1 = acc_o[2 * c3 + 1];
          float* p = Pt + 64 * c3 + 16 * q4 + 8 * hh;  // columns 64 c3 + 2 (8 q4 + 4hh + e) + {0, 1}, e = 0..3
          *reinterpret_cast<f32x4*>(p) = f32x4{o0[4 * q4], o1[4 * q4], o0[4 * q4 + 1], o1[4 * q4 + 1]};
          *reinterpret_cast<f32x4*>(p + 4) = f32x4{o0[4 * q4 + 2], o1[4 * q4 + 2], o0[4 * q4 + 3], o1[4 * q4 + 3]};
        }
      if (hh == 0) {
        Stat[wave * 64 + l31] = m_run;
        Stat[wave * 64 + 32 + l31] = l_run;
      }
    }
    __syncthreads();
    const int nsq = C::qb_ns(qbi), w0 = C::qb_w0(qbi), q0r = q0g + 32 * qbi;
    if (q0r < q_need) {
      for (int idx = tid; idx < 32 * (DK / 4); idx += 64 * WV) {
        const int row = idx / (DK / 4), f4 = idx - row * (DK / 4);
        float m = -INFINITY;
        for (int s2 = 0; s2 < nsq; ++s2) m = fmaxf(m, Stat[(w0 + s2) * 64 + row]);
        float l = 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < nsq; ++s2) {
          const float mw = Stat[(w0 + s2) * 64 + row];
          const float fw = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mw - m) * kScale);
          l += Stat[(w0 + s2) * 64 + 32 + row] * fw;
          o += *reinterpret_cast<const f32x4*>(Qs + s2 * C::PSTR + row * C::OLD + 4 * f4) * fw;
        }
        const float inv = (l > 0.f) ? 1.0f / l : 0.f;  // fully masked row -> 0 (attention.py:118)
        o *= inv;
        if (q0r + row < T1) {
          int frame = q0r + row, feat = h * DK + 4 * f4;
          if (G != 1) split(q0r + row, h * DK + 4 * f4, frame, feat);
          if (frame < F1) *reinterpret_cast<f32x4*>(ctx + (size_t)frame * dm + feat) = o;  // x[:, :T - padding_q] (attention.py:124-125)
        }
      }
    }
    if (qbi + 1 < QB) __syncthreads();  // the slots are reused by the next query block's waves
  }
}

constexpr size_t kLdsAttnT64 = AttnT<64>::LDS_FLOATS * sizeof(float), kLdsAttnT192 = AttnT<192>::LDS_FLOATS * sizeof(float);
constexpr size_t kLdsAttnT128 = AttnT<128>::LDS_FLOATS * sizeof(float), kLdsAttnT256 = AttnT<256>::LDS_FLOATS * sizeof(float);

// -> false: a configuration the kernels do not take (group sizes other than 1 / 3, rows not 16-byte aligned)
bool launch_attention_t(const AttnArgs& a, int B, int H, hipStream_t st) {
  if (a.group < 1 || a.group > 4) return false;
  if ((a.q_stride | a.k_stride | a.v_stride | a.dm) & 3) return false;
  const int nqb = (a.T1 + 31) / 32;
  const int pairs8 = (B * H + 7) / 8;
  if (a.group == 3) {
    const int nq = (nqb + AttnT<192>::QB - 1) / AttnT<192>::QB;  // workgroups per (utterance, head)
    PPASR_LAUNCH(k_attention_t<192>, dim3(nq * pairs8 * 8), dim3(64 * AttnT<192>::WV), kLdsAttnT192, st, a, B, H);
  } else if (a.group == 2) {
    PPASR_LAUNCH(k_attention_t<128>, dim3(nqb * pairs8 * 8), dim3(64 * AttnT<128>::WV), kLdsAttnT128, st, a, B, H);
  } else if (a.group == 4) {
    PPASR_LAUNCH(k_attention_t<256>, dim3(nqb * pairs8 * 8), dim3(64 * AttnT<256>::WV), kLdsAttnT256, st, a, B, H);
  } else {
    PPASR_LAUNCH(k_attention_t<64>, dim3(nqb * pairs8 * 8), dim3(64 * AttnT<64>::WV), kLdsAttnT64, st, a, B, H);
  }
  return true;
}

hipError_t configure_attention_kernels() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention_t<64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)kLdsAttnT64);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention_t<192>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kLdsAttnT192);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention_t<128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kLdsAttnT128);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention_t<256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)kLdsAttnT256);
}

}  // namespace ppasr
