// split_route_kernels.hip -- the layer tail cut at its feed-forward modules for UNDER-FILLED grids (small batches,
// half-rate layers, single streaming sessions).  (Split from conformer_kernels.hip in round 5.)
#include <algorithm>
#include <cstdlib>

#include "conformer_kernels.h"
#include "launch.h"
#include "phases.h"
#include "phases_t.h"
#include "h3.h"

#include <math.h>

namespace ppasr {

// -------------------------------------------------------------------------------------
// Split route for UNDER-FILLED grids (<= 128 row blocks: small batches, half-rate layers, single streaming sessions).
// A launch with fewer row blocks than CUs takes as long as a full one, and 89 % of the fused kernel's time is the two
// FFNs; here the layer tail is cut at the FFNs and each FFN's hidden dimension is split over S workgroups per row
// block (partial sums through HBM, joined by the next launch -- kernel boundaries are the only synchronisation, so
// nothing can deadlock).  Same arithmetic except for the order of the final sum over hidden chunks.
//   k_conv_pre : dwconv -> LN -> swish -> pw2 -> mask -> +res                        -> x3
//   k_ffn_part : LN(x) -> FFN over hidden chunks [s n/S, (s+1) n/S)                  -> partial[s]   (grid blocks x S)
//   k_ffn_join : x + scale (sum_s partial[s] + b2) [-> LN]                           -> out
//   k_ln_qkv   : LN_mha(x1) -> one 256-column third of [Wq|Wk|Wv]                    -> qkv          (grid blocks x 3)
// H3 (round 5): the GEMM units on the fp16 x3 route (ppasr_set_gemm_mode; w: the layer's h3 view) -- a streaming chunk is
// a chain of these launches, each a few units long, so the units' 7 -> 3 us show directly in the chunk latency.
// -------------------------------------------------------------------------------------
template <int KS, bool STREAM, bool H3>
__global__ __launch_bounds__(kThreads) void k_conv_pre(const float* __restrict__ g, const float* __restrict__ g_hist,
                                                       const float* __restrict__ x2, float* __restrict__ x3, LayerW w,
                                                       const int64_t* __restrict__ lens, int M, int Tp, int mask_mul,
                                                       int left_ctx, PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  float* bufX = smem;
  float* bufA = bufX + kRows * kLda;
  float* bufH = bufA + kRows * kLda;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  const int col = wave * 32 + (lane & 31);
  BRing<1> ring;
  const f32x4* seg_pw2 = w.pw2 + (size_t)wave * kTs256;
  ring_prime(ring, seg_pw2, 0);
  PadRows is_pad{lens, r0, Tp, M, mask_mul};
  float res[16];
  unsigned pad_bits = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane);
    res[r] = x2[(size_t)(r0 + min(row, valid - 1)) * kD + col];
    pad_bits |= (is_pad(row) ? 1u : 0u) << r;
  }
  dwconv_phase<KS, STREAM>(g, g_hist, bufA, bufH, bufX, w.dw_w, w.dw_b, w.glu_pad, r0, M, Tp, left_ctx);
  __syncthreads();
  rb_layernorm<true>(bufA, bufA, kLda, kRows, w.ln_cm_g, w.ln_cm_b, w.cm_eps);
  __syncthreads();
  f32x16 acc[1][1];
  acc_zero(acc);
  if constexpr (H3) unit_std_h3(bufA, seg_pw2, nullptr, ring, acc);  // (the planes run on into bufH, free since the conv)
  else rb_gemm<1, 1, kG256>(bufA, kLda, seg_pw2, 0, nullptr, 0, ring, acc);
  const float bv = w.pw2_b[col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane);
    const float c = ((pad_bits >> r) & 1u) ? 0.f : acc[0][0][r] + bv;
    if (row < valid) x3[(size_t)(r0 + row) * kD + col] = res[r] + c;
  }
}

template <bool H3>
__global__ __launch_bounds__(kThreads) void k_ffn_part(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                       const float* __restrict__ ln_b, const f32x4* __restrict__ w1,
                                                       const float* __restrict__ b1, const f32x4* __restrict__ w2,
                                                       float* __restrict__ partial, int M, int n_total, PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  float* bufA = smem;
  float* bufH = bufA + kRows * kLda;  // two hidden-chunk buffers
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  const int n_chunks = n_total / gridDim.y, c0 = blockIdx.y * n_chunks;
  BRing<1> ring;
  ring_prime(ring, w1 + (size_t)(c0 * 8 + wave) * kTs256, 0);
  rb_load_rows(bufA, kLda, x + (size_t)r0 * kD, kRows, valid);
  // (same wave -> row mapping as the load: no barrier between; ln_g == nullptr: the rows are used as they are)
  if (ln_g) rb_layernorm(bufA, bufA, kLda, kRows, ln_g, ln_b, 1e-5f);
  __syncthreads();
  f32x16 acc2[1][1];
  acc_zero(acc2);
  float* out = partial + (size_t)blockIdx.y * M * kD;
  if constexpr (H3) {  // (transposed tile: lane = row, register quad q = columns wave * 32 + 8 q + 4 (lane >> 5) .. +3)
    ffn_phase_h3(bufA, w1, b1, w2, n_chunks, nullptr, ring, acc2, c0, n_total);
    const int row = lane & 31, cq = wave * 32 + 4 * (lane >> 5);
    if (row < valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(out + (size_t)(r0 + row) * kD + cq + 8 * q) =
            f32x4{acc2[0][0][4 * q], acc2[0][0][4 * q + 1], acc2[0][0][4 * q + 2], acc2[0][0][4 * q + 3]};
    }
    return;
  }
  ffn_phase(bufA, bufH, w1, b1, w2, n_chunks, nullptr, ring, acc2, c0, n_total);
  const int col = wave * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane);
    if (row < valid) out[(size_t)(r0 + row) * kD + col] = acc2[0][0][r];
  }
}

__device__ __forceinline__ f32x4 ln_row(f32x4 y, const float* __restrict__ g, const float* __restrict__ b, int lane) {
  const float mean = wave_sum(y[0] + y[1] + y[2] + y[3]) * (1.0f / kD);
  const f32x4 c = y - mean;
  const float var = wave_sum(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]) * (1.0f / kD);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  return c * rstd * *reinterpret_cast<const f32x4*>(g + 4 * lane) + *reinterpret_cast<const f32x4*>(b + 4 * lane);
}
// one wave per row: out = LN_out?(LN_pre?(x) + scale * (sum_s partial[s] + b2))
__global__ __launch_bounds__(256) void k_ffn_join(const float* __restrict__ x, const float* __restrict__ partial, int S,
                                                  const float* __restrict__ b2, float scale, const float* __restrict__ ln_g,
                                                  const float* __restrict__ ln_b, float* __restrict__ out, int M,
                                                  PadSkip ps, const float* __restrict__ pre_g,
                                                  const float* __restrict__ pre_b) {
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= M) return;
  if (pad_block_skippable(ps, row & ~(kRows - 1), kRows, M)) return;
  const int lane = lane_id();
  // (the S tiles' loads in flight eight at a time, added in slice order: a chain of S dependent L2 round trips otherwise)
  const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + 4 * lane);
  f32x4 y = *reinterpret_cast<const f32x4*>(x + (size_t)row * kD + 4 * lane);
  f32x4 acc = *reinterpret_cast<const f32x4*>(partial + (size_t)row * kD + 4 * lane);
  int s = 1;
  for (; s + 8 <= S; s += 8) {
    f32x4 p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = *reinterpret_cast<const f32x4*>(partial + ((size_t)(s + j) * M + row) * kD + 4 * lane);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += p[j];
  }
  for (; s < S; ++s) acc += *reinterpret_cast<const f32x4*>(partial + ((size_t)s * M + row) * kD + 4 * lane);
  if (pre_g) y = ln_row(y, pre_g, pre_b, lane);
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] = y[e] + scale * (acc[e] + bv[e]);
  if (ln_g) y = ln_row(y, ln_g, ln_b, lane);
  *reinterpret_cast<f32x4*>(out + (size_t)row * kD + 4 * lane) = y;
}

// rows of a 16-row block <- the pending join `jn` (conformer_kernels.h JoinIn), with the wave -> row mapping of
// rbt_load_rows / rbt_layernorm (wave w: rows w, w + 8): LDS rows for this workgroup, global rows if `store`
__device__ __forceinline__ void join_rows16(float* buf, const JoinIn& jn, int M, bool store) {
  const int lane = lane_id(), wave = wave_id();
  const f32x4 bv = *reinterpret_cast<const f32x4*>(jn.b2 + 4 * lane);
  // the tiles were written by workgroups all over the chip (other XCDs' L2s): a load is a ~ 2 us round trip, so both rows'
  // tiles are requested eight slices at a time (16 loads in flight) and added in slice order (k_ffn_join's sums)
  f32x4 y[2], acc[2];
  const float* src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = min(wave + 8 * i, M - 1);  // (rows past M: a clamped address, the result is dropped)
    y[i] = *reinterpret_cast<const f32x4*>(jn.x + (size_t)row * kD + 4 * lane);
    src[i] = jn.partial + (size_t)row * kD + 4 * lane;
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const size_t tile = (size_t)M * kD;
  for (int s0 = 0; s0 < jn.S; s0 += 8) {
    f32x4 p[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        p[i][j] = s0 + j < jn.S ? *reinterpret_cast<const f32x4*>(src[i] + (size_t)(s0 + j) * tile) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (s0 + j < jn.S) acc[i] += p[i][j];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave + 8 * i;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[i][e] = y[i][e] + jn.scale * (acc[i][e] + bv[e]);
    if (jn.ln_g) y[i] = ln_row(y[i], jn.ln_g, jn.ln_b, lane);
    if (row >= M) y[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    else if (store) *reinterpret_cast<f32x4*>(jn.out + (size_t)row * kD + 4 * lane) = y[i];
    *reinterpret_cast<f32x4*>(buf + row * kLda + 4 * lane) = y[i];
  }
}

// kc / vc != nullptr (single-session streaming): the K and V thirds go straight to the session's cache rows (row m of
// the chunk -> kc + m*256) instead of qkv -- the separate append launch disappears
template <bool H3>
__global__ __launch_bounds__(kThreads) void k_ln_qkv(const float* __restrict__ x1, float* __restrict__ qkv, LayerW w, int M,
                                                     PadSkip ps, float* __restrict__ kc, float* __restrict__ vc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  float* bufA = smem;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  const int c = blockIdx.y;  // 0: q, 1: k, 2: v
  BRing<1> ring;
  const f32x4* seg = w.wqkv + (size_t)(c * 8 + wave) * kTs256;
  ring_prime(ring, seg, 0);
  rb_load_rows(bufA, kLda, x1 + (size_t)r0 * kD, kRows, valid);
  rb_layernorm(bufA, bufA, kLda, kRows, w.ln_mha_g, w.ln_mha_b, 1e-5f);
  __syncthreads();
  f32x16 acc[1][1];
  acc_zero(acc);
  if constexpr (H3) unit_std_h3(bufA, seg, nullptr, ring, acc);
  else rb_gemm<1, 1, kG256>(bufA, kLda, seg, 0, nullptr, 0, ring, acc);
  const int col = c * 256 + wave * 32 + (lane & 31);
  const float bv = w.bqkv[col];
  float* cache = (c == 1) ? kc : (c == 2 ? vc : nullptr);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane);
    if (row >= valid) continue;
    if (cache) cache[(size_t)(r0 + row) * kD + wave * 32 + (lane & 31)] = acc[0][0][r] + bv;
    else qkv[(size_t)(r0 + row) * 768 + col] = acc[0][0][r] + bv;
  }
}

// ---- the same two launches on 16-ROW blocks (rbt.h: v_mfma_f32_16x16x4_f32 on the same packed weights) ----
// A streaming chunk is 16 frames (decoding_chunk_size 16): on the 32-row forms above half of every tile is padding, and a
// GEMM unit is bound by one CU's matrix pipe (6.8 us for 32 rows x 256 x 256 in fp32) -- the 16-row unit takes half.
// Round 6: the feed-forward slices and the Q / K / V thirds, 36 of a chunk's ~110 units.  Results agree with the 32-row
// forms to the order of the sums inside a 16-wide k step.
// In-kernel join (jn.ticket != nullptr; single stream handles): the S workgroups of a row block take a ticket when their
// partial tile is in memory, and the LAST one to arrive runs k_ffn_join's row loop for the block -- no spinning (the others
// have left), so nothing can deadlock; the counter goes back to 0 for the next launch.  One launch per feed-forward module
// instead of two.
struct FfnJoin {
  const float* b2;
  float scale;
  const float *ln_g, *ln_b;  // LayerNorm behind the residual sum (or nullptr)
  float* out;
  const float *pre_g, *pre_b;  // the residual is LN_pre(x) (Squeezeformer's second module) or x
  int* ticket;               // [row blocks] arrival counters, zero between launches
};
template <int R>
__global__ __launch_bounds__(RBT<R>::THREADS) void k_ffn_part_t(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                                const float* __restrict__ ln_b, const f32x4* __restrict__ w1,
                                                                const float* __restrict__ b1, const f32x4* __restrict__ w2,
                                                                float* __restrict__ partial, int M, int n_total, PadSkip ps,
                                                                FfnJoin jn) {
  using T = RBT<R>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, T::ROWS, M);
  if (blk < 0) return;
  float* bufA = smem;
  float* bufH = bufA + T::ROWS * kLda;  // two hidden-chunk buffers
  const LaneT<R> L;
  const int r0 = blk * T::ROWS;
  const int valid = min(T::ROWS, M - r0);
  const int n_chunks = n_total / gridDim.y, c0 = blockIdx.y * n_chunks;
  typename T::Ring ring;
  rbt_prime(ring, w1 + (size_t)(c0 * 8 + L.tile()) * kTs256);
  rbt_load_rows<R>(bufA, x + (size_t)r0 * kD, valid);
  // (same wave -> row mapping as the load: no barrier between; ln_g == nullptr: the rows are used as they are)
  if (ln_g) rbt_layernorm<R>(bufA, bufA, ln_g, ln_b, 1e-5f);
  __syncthreads();
  typename T::Acc acc2;
  T::zero(acc2);
  ffn_phase_t<R>(bufA, bufH, w1, b1, w2, n_chunks, nullptr, ring, acc2, c0, n_total);
  float* out = partial + (size_t)blockIdx.y * M * kD;
#pragma unroll
  for (int q = 0; q < T::NQ; ++q)
    if (L.row(q) < valid) *reinterpret_cast<f32x4*>(out + (size_t)(r0 + L.row(q)) * kD + L.col(q)) = T::quad(acc2, q);
  if (!jn.ticket) return;
  __shared__ int s_last;
  __threadfence();  // this workgroup's partial tile is visible device-wide before its ticket is
  __syncthreads();
  if (threadIdx.x == 0) {
    const int S = (int)gridDim.y;
    const int t = atomicAdd(&jn.ticket[blk], 1);
    s_last = (t == S - 1) ? 1 : 0;
    if (t == S - 1) jn.ticket[blk] = 0;  // (everyone has arrived: re-armed for the next launch on this stream)
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();  // (the other workgroups' tiles: no stale lines of this CU's vector cache)
  const int S = (int)gridDim.y, lane = L.lane;
  for (int row = r0 + L.wave; row < r0 + valid; row += T::WAVES) {  // k_ffn_join, one wave per row
    f32x4 acc = *reinterpret_cast<const f32x4*>(partial + (size_t)row * kD + 4 * lane);
    for (int sidx = 1; sidx < S; ++sidx)
      acc += *reinterpret_cast<const f32x4*>(partial + ((size_t)sidx * M + row) * kD + 4 * lane);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(jn.b2 + 4 * lane);
    f32x4 y = *reinterpret_cast<const f32x4*>(x + (size_t)row * kD + 4 * lane);
    if (jn.pre_g) y = ln_row(y, jn.pre_g, jn.pre_b, lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = y[e] + jn.scale * (acc[e] + bv[e]);
    if (jn.ln_g) y = ln_row(y, jn.ln_g, jn.ln_b, lane);
    *reinterpret_cast<f32x4*>(jn.out + (size_t)row * kD + 4 * lane) = y;
  }
}

template <int R>
__global__ __launch_bounds__(RBT<R>::THREADS) void k_ln_qkv_t(const float* __restrict__ x1, float* __restrict__ qkv, LayerW w,
                                                              int M, PadSkip ps, float* __restrict__ kc,
                                                              float* __restrict__ vc, JoinIn jn) {
  using T = RBT<R>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, T::ROWS, M);
  if (blk < 0) return;
  float* bufA = smem;
  const LaneT<R> L;
  const int r0 = blk * T::ROWS;
  const int valid = min(T::ROWS, M - r0);
  const int c = blockIdx.y;  // 0: q, 1: k, 2: v
  typename T::Ring ring;
  const f32x4* seg = w.wqkv + (size_t)(c * 8 + L.tile()) * kTs256;
  rbt_prime(ring, seg);
  if (R == 16 && jn.partial) join_rows16(bufA, jn, M, c == 0);  // (one row block; x1 = jn.out is written here)
  else rbt_load_rows<R>(bufA, x1 + (size_t)r0 * kD, valid);
  // (w.ln_mha_g == nullptr: the rows are used as they are -- Squeezeformer's projection, whose scale is folded into wqkv)
  if (w.ln_mha_g) rbt_layernorm<R>(bufA, bufA, w.ln_mha_g, w.ln_mha_b, 1e-5f);
  __syncthreads();
  typename T::Acc acc;
  T::zero(acc);
  rbt_gemm<kG256>(bufA, kLda, seg, nullptr, ring, acc);
  float* cache = (c == 1) ? kc : (c == 2 ? vc : nullptr);
#pragma unroll
  for (int q = 0; q < T::NQ; ++q) {
    if (L.row(q) >= valid) continue;
    const f32x4 v = T::quad(acc, q) + *reinterpret_cast<const f32x4*>(w.bqkv + c * 256 + L.col(q));
    if (cache) *reinterpret_cast<f32x4*>(cache + (size_t)(r0 + L.row(q)) * kD + L.col(q)) = v;
    else *reinterpret_cast<f32x4*>(qkv + (size_t)(r0 + L.row(q)) * 768 + c * 256 + L.col(q)) = v;
  }
}
// ---- a feed-forward slice of HALF a hidden chunk (128 units) on 16 rows: gridDim.y = 2 x chunks ----
// One chunk per workgroup is two dependent 256 x 256 units on one CU's matrix pipes (3.4 us each on 16 rows: 8 waves, two
// per SIMD); with 128 hidden units per workgroup the first unit is 4 waves x 32 hidden columns (one wave per SIMD) and the
// second contracts K = 128 on all 8 waves -- 1.7 us each, twice as many CUs at work, twice as many partial tiles for the
// join.  Both weight streams are requested before the LayerNorm (two rings: 64 registers).
__global__ __launch_bounds__(RBT<16>::THREADS) void k_ffn_half16(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                                 const float* __restrict__ ln_b, const f32x4* __restrict__ w1,
                                                                 const float* __restrict__ b1, const f32x4* __restrict__ w2,
                                                                 float* __restrict__ partial, int M, int n_total, PadSkip ps,
                                                                 JoinIn jn) {
  using T = RBT<16>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, T::ROWS, M);
  if (blk < 0) return;
  float* bufA = smem;
  float* bufH = bufA + T::ROWS * kLda;  // [16][128 of kLda]
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * T::ROWS;
  const int valid = min(T::ROWS, M - r0);
  const int c = blockIdx.y >> 1, half = blockIdx.y & 1;
  const int ts2 = n_total * 32 * 64;  // W2: K = hidden
  const f32x4* seg1 = w1 + (size_t)(c * 8 + 4 * half + (wave & 3)) * kTs256;
  const f32x4* seg2 = w2 + (size_t)wave * ts2 + (size_t)(c * 32 + 16 * half) * 64;
  typename T::Ring ring1, ring2;
  if (wave < 4) rbt_prime(ring1, seg1);
  rbt_prime(ring2, seg2);
  if (jn.partial) join_rows16(bufA, jn, M, blockIdx.y == 0);  // (one row block: r0 = 0)
  else rbt_load_rows<16>(bufA, x + (size_t)r0 * kD, valid);
  if (ln_g) rbt_layernorm<16>(bufA, bufA, ln_g, ln_b, 1e-5f);
  __syncthreads();
  const int row = lane & 15;
  if (wave < 4) {
    typename T::Acc cur;
    T::zero(cur);
    rbt_gemm<kG256>(bufA, kLda, seg1, nullptr, ring1, cur);
#pragma unroll
    for (int q = 0; q < T::NQ; ++q) {
      const int c128 = 32 * wave + 16 * q + 4 * (lane >> 4);  // hidden unit inside the workgroup's 128
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b1 + c * 256 + 128 * half + c128);
      const f32x4 v = cur.s[q];
      const f32x2 lo = swish2(f32x2{v[0] + bv[0], v[1] + bv[1]});
      const f32x2 hi = swish2(f32x2{v[2] + bv[2], v[3] + bv[3]});
      *reinterpret_cast<f32x4*>(bufH + row * kLda + c128) = f32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  }
  __syncthreads();
  typename T::Acc acc2;
  T::zero(acc2);
  rbt_gemm<kG256 / 2>(bufH, kLda, seg2, nullptr, ring2, acc2);
  float* out = partial + (size_t)blockIdx.y * M * kD;
#pragma unroll
  for (int q = 0; q < T::NQ; ++q)
    if (row < valid) *reinterpret_cast<f32x4*>(out + (size_t)(r0 + row) * kD + wave * 32 + 16 * q + 4 * (lane >> 4)) = acc2.s[q];
}

// ---- the conv module on 16-row blocks with pointwise_conv2's columns over gridDim.y = 2 ----
// k_conv_pre<KS, ..> runs it on 32-row workgroups: 18.6 us per block of the encoder for one row block, half of it the 32-row
// unit on one CU.  Here both workgroups of a row block stage the window, run the depthwise conv + LayerNorm / folded BatchNorm
// + swish for all 256 channels (wave w: rows 2w, 2w + 1; taps in ascending order like dwconv_phase) and waves 0 - 3 contract
// their 128 output columns; the feed-forward slices that follow normalise their input themselves, so the halves need no join.
// BATCH = false: ONE streaming session's chunk (M <= 16 rows = the only row block; the window's first KS - 1 rows are the
// session's g_hist).  BATCH = true: rows of B utterances of Tp frames (single utterances, small batches): taps outside the
// row's utterance read glu_pad (causal) / 0, padded frames pass the residual through (k_conv_pre's semantics).
template <int KS, bool BATCH>
__global__ __launch_bounds__(RBT<16>::THREADS) void k_conv_pre_cols16(const float* __restrict__ g, const float* __restrict__ g_hist,
                                                                      const float* __restrict__ x2, float* __restrict__ x3,
                                                                      LayerW w, const int64_t* __restrict__ lens, int M, int Tp,
                                                                      int mask_mul, int left) {
  using T = RBT<16>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LO = KS - 1;
  float* win = smem;                     // [LO + 16][kLda]: rows r0 - left .. (stream: g_hist rows, then the chunk's)
  float* taps = win + (LO + 16) * kLda;  // [KS][kLda]
  float* bufA = taps + KS * kLda;        // [16][kLda]
  const int lane = lane_id(), wave = wave_id(), y = blockIdx.y;
  const int r0 = blockIdx.x * 16;
  const f32x4* seg = w.pw2 + (size_t)(4 * y + (wave & 3)) * kTs256;
  typename T::Ring ring;
  if (wave < 4) rbt_prime(ring, seg);
  for (int q = wave; q < LO + 16; q += T::WAVES) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (BATCH) {
      const int mq = r0 - left + q;
      if (mq >= 0 && mq < M) v = *reinterpret_cast<const f32x4*>(g + (size_t)mq * kD + 4 * lane);
    } else {
      if (q < LO) v = *reinterpret_cast<const f32x4*>(g_hist + (size_t)q * kD + 4 * lane);
      else if (q - LO < M) v = *reinterpret_cast<const f32x4*>(g + (size_t)(q - LO) * kD + 4 * lane);
    }
    *reinterpret_cast<f32x4*>(win + q * kLda + 4 * lane) = v;
  }
  for (int j = wave; j < KS; j += T::WAVES)
    *reinterpret_cast<f32x4*>(taps + j * kLda + 4 * lane) = *reinterpret_cast<const f32x4*>(w.dw_w + j * kD + 4 * lane);
  const f32x4 bias = *reinterpret_cast<const f32x4*>(w.dw_b + 4 * lane);
  const f32x4 gam = *reinterpret_cast<const f32x4*>(w.ln_cm_g + 4 * lane);
  const f32x4 bet = *reinterpret_cast<const f32x4*>(w.ln_cm_b + 4 * lane);
  f32x4 gp = {0.f, 0.f, 0.f, 0.f};
  if (BATCH && left == LO) gp = *reinterpret_cast<const f32x4*>(w.glu_pad + 4 * lane);
  __syncthreads();
  f32x4 out[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 2 * wave + i;
    const int m = r0 + row;
    const int t = BATCH ? m - (m / Tp) * Tp : 0;
    f32x4 acc = bias;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      f32x4 xv = *reinterpret_cast<const f32x4*>(win + (row + j) * kLda + 4 * lane);
      if (BATCH) {
        const int tt = t - left + j;  // frame this tap reads inside the row's utterance
        if (!(tt >= 0 && tt < Tp)) xv = gp;
      }
      acc += *reinterpret_cast<const f32x4*>(taps + j * kLda + 4 * lane) * xv;
    }
    out[i] = acc;
  }
  ln_rows_inreg<true, 2>(out, gam, bet, w.cm_eps);
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(bufA + (2 * wave + i) * kLda + 4 * lane) = out[i];
  __syncthreads();
  if (wave >= 4) return;
  typename T::Acc acc;
  T::zero(acc);
  rbt_gemm<kG256>(bufA, kLda, seg, nullptr, ring, acc);
  const int row = lane & 15;
  if (r0 + row >= M) return;
  const bool pad = BATCH && PadRows{lens, r0, Tp, M, mask_mul}(row);
#pragma unroll
  for (int q = 0; q < T::NQ; ++q) {
    const int col = 128 * y + 32 * wave + 16 * q + 4 * (lane >> 4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(w.pw2_b + col);
    const f32x4 r = *reinterpret_cast<const f32x4*>(x2 + (size_t)(r0 + row) * kD + col);
    const f32x4 a = acc.s[q];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = r[e] + (pad ? 0.f : a[e] + bv[e]);
    *reinterpret_cast<f32x4*>(x3 + (size_t)(r0 + row) * kD + col) = o;
  }
}
template <int KS>
constexpr size_t conv_cols16_lds() { return (size_t)(KS - 1 + 16 + KS + 16) * kLda * sizeof(float); }

// (LDS asked for: kLdsExclusive, so that a CU holds ONE of these workgroups -- co-resident row-block workgroups share the
//  matrix pipe and the weight stream's lead time no longer covers a unit: with 32 / 64 independent sessions on their own
//  HIP streams the own-size allocation, 3 workgroups per CU, was 13 % slower per chunk round, same box)
constexpr size_t kLdsFfnPart16 = kLdsExclusive, kLdsLnQkv16 = kLdsExclusive;
static_assert(3 * 16 * kLda * sizeof(float) <= kLdsExclusive, "LDS of the 16-row feed-forward slice");
// (up to split_rows16_max() rows -- streaming chunks, single utterances, small batches: the 16-row forms; PPASR_SPLIT_ROWS16=0
//  switches them off)
static bool ffn_half16_on() {  // (PPASR_FFN_HALF16=0: whole chunks per workgroup, the A/B knob of the measurement above)
  static const bool on = !(getenv("PPASR_FFN_HALF16") && atoi(getenv("PPASR_FFN_HALF16")) == 0);
  return on;
}
// Rows up to which the split route runs its 16-row forms (PPASR_SPLIT_ROWS16_MAX: tuning knob).  512 rows = 32 blocks x 8
// feed-forward slices = one round of 256 half-as-long workgroups.  Encoder latency of ONE utterance, same box, 16-row forms up
// to 16 rows (a streaming chunk only) / up to 512: 5 s 1.64 / 1.23 ms, 10 s 1.69 / 1.28 ms, 20 s 1.85 / 1.47 ms
// (tools/experiments/r06/single_utt.py).
int split_rows16_max() {
  static const int m = [] {
    if (getenv("PPASR_SPLIT_ROWS16") && atoi(getenv("PPASR_SPLIT_ROWS16")) == 0) return 0;
    const char* e = getenv("PPASR_SPLIT_ROWS16_MAX");
    return e ? atoi(e) : 512;
  }();
  return m;
}
static bool split_rows16(int M) { return M <= split_rows16_max(); }

constexpr size_t kLdsConvPre = 4 * kRows * kLda * sizeof(float);  // (the depthwise window uses the three buffers + halo)
constexpr size_t kLdsFfnPart = 3 * kRows * kLda * sizeof(float);
constexpr size_t kLdsLnQkv = kRows * kLda * sizeof(float);
void launch_conv_pre(const float* g, const float* g_hist, const float* x2, float* x3, const LayerW& w, const int64_t* lens,
                     int M, int Tp, int ksize, int mask_mul, hipStream_t st, bool causal, const PadSkip& ps, bool h3) {
  dim3 grid((M + kRows - 1) / kRows);
  const int left_ctx = causal ? ksize - 1 : (ksize - 1) / 2;
  // one session's chunk of up to 16 frames, or batched rows (single utterances, small batches): the column-split 16-row form
  const bool one_chunk = g_hist && !lens && causal && M == Tp && M <= 16;
  if ((one_chunk || !g_hist) && !h3 && !ps.tab && split_rows16(M) && (ksize == 15 || ksize == 31 || ksize == 7)) {
#define LAUNCH_CC16(KS)                                                                                                  \
  do {                                                                                                                   \
    if (g_hist)                                                                                                          \
      PPASR_LAUNCH((k_conv_pre_cols16<KS, false>), dim3(1, 2), dim3(kThreads), std::max(conv_cols16_lds<KS>(), kLdsExclusive), st, \
                   g, g_hist, x2, x3, w, lens, M, Tp, mask_mul, left_ctx);                                               \
    else                                                                                                                 \
      PPASR_LAUNCH((k_conv_pre_cols16<KS, true>), dim3((M + 15) / 16, 2), dim3(kThreads),                                \
                   std::max(conv_cols16_lds<KS>(), kLdsExclusive), st, g, g_hist, x2, x3, w, lens, M, Tp, mask_mul, left_ctx); \
  } while (0)
    if (ksize == 15) LAUNCH_CC16(15);
    else if (ksize == 31) LAUNCH_CC16(31);
    else LAUNCH_CC16(7);
#undef LAUNCH_CC16
    return;
  }
#define LAUNCH_CP2(KS, STREAM, H3)                                                                                      \
  PPASR_LAUNCH((k_conv_pre<KS, STREAM, H3>), grid, dim3(kThreads), kLdsConvPre, st, g, g_hist, x2, x3, w, lens, M, Tp, \
               mask_mul, left_ctx, ps)
#define LAUNCH_CP(KS)                                      \
  if (g_hist && h3) LAUNCH_CP2(KS, true, true);            \
  else if (g_hist) LAUNCH_CP2(KS, true, false);            \
  else if (h3) LAUNCH_CP2(KS, false, true);                \
  else LAUNCH_CP2(KS, false, false);
  if (ksize == 15) {
    LAUNCH_CP(15)
  } else if (ksize == 31) {  // (31 taps: Squeezeformer's view and Conformer variants the fp16 x3 layer views do not exist for)
    if (g_hist) LAUNCH_CP2(31, true, false);
    else LAUNCH_CP2(31, false, false);
  } else if (ksize == 7) {
    LAUNCH_CP(7)
  }
#undef LAUNCH_CP
#undef LAUNCH_CP2
}
void launch_ffn_split(const float* x, const float* ln_g, const float* ln_b, const f32x4* w1, const float* b1,
                      const f32x4* w2, const float* b2, float scale, const float* out_ln_g, const float* out_ln_b,
                      float* partial, float* out, int M, int n_chunks, int S, hipStream_t st, const PadSkip& ps,
                      bool residual_is_normed, bool h3, int* ticket) {
  if (h3)  // (w1 / w2: the re-packed weights)
    PPASR_LAUNCH(k_ffn_part<true>, dim3((M + kRows - 1) / kRows, S), dim3(kThreads), kLdsFfnPart + kH3ExtraLds, st, x, ln_g, ln_b,
                 w1, b1, w2, partial, M, n_chunks, ps);
  else if (split_rows16(M) && !ps.tab && S == n_chunks && ffn_half16_on() && ((M + 15) / 16) * 2 * S <= 256 &&
           !(getenv("PPASR_STREAM_TICKET") && atoi(getenv("PPASR_STREAM_TICKET")) == 1)) {
    // one chunk per workgroup already: cut the chunks in halves (k_ffn_half16), 2 S partial tiles
    PPASR_LAUNCH(k_ffn_half16, dim3((M + 15) / 16, 2 * S), dim3(kThreads), kLdsFfnPart16, st, x, ln_g, ln_b, w1, b1, w2, partial, M,
                 n_chunks, ps, JoinIn{});
    S *= 2;
  } else if (split_rows16(M) && !ps.tab) {
    // (out == x would let the joining workgroup overwrite rows another slice is still reading: two launches then)
    // OPT-IN (PPASR_STREAM_TICKET=1): measured on one box, one 0.64 s chunk of one session 1.35 ms with the in-kernel join
    // against 1.24 ms with the join as its own launch (the last slice's workgroup joins 16 rows x S partial tiles alone; the
    // join kernel spreads them over 4 workgroups) -- four launches fewer per block, and slower.  32 sessions: 14.26 / 14.5 ms.
    const char* tk = getenv("PPASR_STREAM_TICKET");
    const bool join_in_kernel = tk && atoi(tk) == 1 && ticket != nullptr && out != x && (M + 15) / 16 <= 16;
    PPASR_LAUNCH(k_ffn_part_t<16>, dim3((M + 15) / 16, S), dim3(kThreads), kLdsFfnPart16, st, x, ln_g, ln_b, w1, b1, w2, partial, M,
                 n_chunks, ps,
                 FfnJoin{b2, scale, out_ln_g, out_ln_b, out, residual_is_normed ? ln_g : nullptr, residual_is_normed ? ln_b : nullptr,
                         join_in_kernel ? ticket : nullptr});
    if (join_in_kernel) return;
  } else
    PPASR_LAUNCH(k_ffn_part<false>, dim3((M + kRows - 1) / kRows, S), dim3(kThreads), kLdsFfnPart, st, x, ln_g, ln_b, w1, b1,
                 w2, partial, M, n_chunks, ps);
  PPASR_LAUNCH(k_ffn_join, dim3((M + 3) / 4), dim3(256), 0, st, x, partial, S, b2, scale, out_ln_g, out_ln_b, out, M,
                     ps, residual_is_normed ? ln_g : nullptr, residual_is_normed ? ln_b : nullptr);
}
void launch_ln_qkv(const float* x1, float* qkv, const LayerW& w, int M, hipStream_t st, const PadSkip& ps, float* kc,
                   float* vc, bool h3) {
  if (h3)
    PPASR_LAUNCH(k_ln_qkv<true>, dim3((M + kRows - 1) / kRows, 3), dim3(kThreads), kLdsLnQkv + 512, st, x1, qkv, w, M, ps, kc, vc);
  else if (split_rows16(M) && !ps.tab)
    PPASR_LAUNCH(k_ln_qkv_t<16>, dim3((M + 15) / 16, 3), dim3(kThreads), kLdsLnQkv16, st, x1, qkv, w, M, ps, kc, vc, JoinIn{});
  else
    PPASR_LAUNCH(k_ln_qkv<false>, dim3((M + kRows - 1) / kRows, 3), dim3(kThreads), kLdsLnQkv, st, x1, qkv, w, M, ps, kc, vc);
}
// ---- consumer-side joins of one streaming session's chunk (conformer_kernels.h JoinIn) ----
bool ffn_half16_route(int M, int S, int n_chunks) {
  static const bool fuse = !(getenv("PPASR_JOIN_FUSED") && atoi(getenv("PPASR_JOIN_FUSED")) == 0);  // (A/B switch)
  return fuse && M <= 16 && S > 1 && S == n_chunks && split_rows16(M) && ffn_half16_on() &&
         !(getenv("PPASR_STREAM_TICKET") && atoi(getenv("PPASR_STREAM_TICKET")) == 1);
}
void launch_ffn_half16(const float* x, const JoinIn& jn, const float* ln_g, const float* ln_b, const f32x4* w1, const float* b1,
                       const f32x4* w2, float* partial, int M, int n_chunks, hipStream_t st) {
  PPASR_LAUNCH(k_ffn_half16, dim3(1, 2 * n_chunks), dim3(kThreads), kLdsFfnPart16, st, x, ln_g, ln_b, w1, b1, w2, partial, M,
               n_chunks, PadSkip{}, jn);
}
void launch_join_ln_qkv16(const JoinIn& jn, float* qkv, const LayerW& w, int M, hipStream_t st, float* kc, float* vc) {
  PPASR_LAUNCH(k_ln_qkv_t<16>, dim3(1, 3), dim3(kThreads), kLdsLnQkv16, st, (const float*)jn.out, qkv, w, M, PadSkip{}, kc, vc, jn);
}
void launch_join16(const JoinIn& jn, int M, hipStream_t st) {
  PPASR_LAUNCH(k_ffn_join, dim3((M + 3) / 4), dim3(256), 0, st, jn.x, jn.partial, jn.S, jn.b2, jn.scale, jn.ln_g, jn.ln_b, jn.out, M,
               PadSkip{}, (const float*)nullptr, (const float*)nullptr);
}
unsigned int* split_route_h3_ovf_counter() { return h3_ovf_counter(); }
hipError_t configure_split_route_kernels() {
  hipError_t e = hipSuccess;
#define SET_LDS(fn, bytes)                                                                                     \
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
  if (e != hipSuccess) return e;
  SET_LDS((k_conv_pre<15, false, false>), kLdsConvPre);
  SET_LDS((k_conv_pre<31, false, false>), kLdsConvPre);
  SET_LDS((k_conv_pre<7, false, false>), kLdsConvPre);
  SET_LDS((k_conv_pre<15, true, false>), kLdsConvPre);
  SET_LDS((k_conv_pre<31, true, false>), kLdsConvPre);
  SET_LDS((k_conv_pre<7, true, false>), kLdsConvPre);
  SET_LDS((k_conv_pre<15, false, true>), kLdsConvPre);
  SET_LDS((k_conv_pre<7, false, true>), kLdsConvPre);
  SET_LDS((k_conv_pre<15, true, true>), kLdsConvPre);
  SET_LDS((k_conv_pre<7, true, true>), kLdsConvPre);
  SET_LDS(k_ffn_part<false>, kLdsFfnPart);
  SET_LDS(k_ffn_part_t<16>, kLdsFfnPart16);
  SET_LDS(k_ffn_half16, kLdsFfnPart16);
  SET_LDS((k_conv_pre_cols16<15, false>), std::max(conv_cols16_lds<15>(), kLdsExclusive));
  SET_LDS((k_conv_pre_cols16<15, true>), std::max(conv_cols16_lds<15>(), kLdsExclusive));
  SET_LDS((k_conv_pre_cols16<31, false>), std::max(conv_cols16_lds<31>(), kLdsExclusive));
  SET_LDS((k_conv_pre_cols16<31, true>), std::max(conv_cols16_lds<31>(), kLdsExclusive));
  SET_LDS((k_conv_pre_cols16<7, false>), std::max(conv_cols16_lds<7>(), kLdsExclusive));
  SET_LDS((k_conv_pre_cols16<7, true>), std::max(conv_cols16_lds<7>(), kLdsExclusive));
  SET_LDS(k_ln_qkv_t<16>, kLdsLnQkv16);
  SET_LDS(k_ffn_part<true>, kLdsFfnPart + kH3ExtraLds);
  SET_LDS(k_ln_qkv<true>, kLdsLnQkv + 512);
#undef SET_LDS
  return hipSuccess;
}

}  // namespace ppasr
