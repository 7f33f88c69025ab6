// squeezeformer_kernels.hip -- row-block kernels of the Squeezeformer encoder layer
// (ppasr/model_utils/squeezeformer/encoder.py:386-506: post-LN, order MHA -> FFN -> Conv -> FFN),
// time reduction / recovery (encoder.py:210-230, time_reduction.py:183-206) built from phases.h.
//
// Adaptive scale (x <- ada_scale*x + ada_bias in front of MHA / FFN / conv module,
// attention.py:120-123, positionwise.py:63-64, convolution.py:119-120) is folded into the following
// dense layer on the host:  (s.x + a) W + c = x (diag(s) W) + (a W + c).  The conv module zeroes PAD
// frames AFTER the scale (convolution.py:121-127), so PAD rows bypass the folded GEMM and take
// GLU(bias) = `glu_pad` directly.
#include <cstdlib>

#include "launch.h"
#include "squeezeformer_kernels.h"
#include "conformer_kernels.h"  // ragged_lds / kLdsExclusive

#include "phases_t.h"
#include "h3.h"

namespace ppasr {

// qkv = x * Wqkv' + b'  (first layer, after preln)
__global__ __launch_bounds__(kThreads) void k_sq_qkv(const float* __restrict__ x, float* __restrict__ qkv,
                                                     const f32x4* __restrict__ wqkv, const float* __restrict__ bqkv, int M,
                                                     PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (pad_block_skippable(ps, blockIdx.x * kRows, kRows, M)) return;
  float* bufA = smem;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blockIdx.x * kRows;
  const int valid = min(kRows, M - r0);
  BRing<1> ring;
  ring_prime(ring, wqkv + (size_t)wave * kTs256, 0);
  rb_load_rows(bufA, kLda, x + (size_t)r0 * kD, kRows, valid);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f32x16 acc[1][1];
    acc_zero(acc);
    const f32x4* seg = wqkv + (size_t)(c * 8 + wave) * kTs256;
    rb_gemm<1, 1, kG256>(bufA, kLda, seg, 0, c < 2 ? seg + 8 * kTs256 : nullptr, 0, ring, acc);
    const int col = c * 256 + wave * 32 + (lane & 31);
    const float bv = bqkv[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = acc_row(r, lane);
      if (row < valid) qkv[(size_t)(r0 + row) * 768 + col] = acc[0][0][r] + bv;
    }
  }
}

// shared tail: qkv = bufX * Wqkv + b  (ring already continues into wqkv tile `wave`)
__device__ __forceinline__ void qkv_from_lds(const float* bufX, float* __restrict__ qkv, const f32x4* __restrict__ wqkv,
                                             const float* __restrict__ bqkv, int r0, int valid, BRing<1>& ring) {
  const int lane = lane_id(), wave = wave_id();
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f32x16 acc[1][1];
    acc_zero(acc);
    const f32x4* seg = wqkv + (size_t)(c * 8 + wave) * kTs256;
    rb_gemm<1, 1, kG256>(bufX, kLda, seg, 0, c < 2 ? seg + 8 * kTs256 : nullptr, 0, ring, acc);
    const int col = c * 256 + wave * 32 + (lane & 31);
    const float bv = bqkv[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = acc_row(r, lane);
      if (row < valid) qkv[(size_t)(r0 + row) * 768 + col] = acc[0][0][r] + bv;
    }
  }
}


// ---- K_B / K_C on transposed accumulators, block forms of rbt.h: R = 32 / 16 rows on 8 waves, kW16 = 32 rows on 16 ----
// Same arithmetic as k_sq_mid / k_sq_tail below (R = 32: bit-identical -- the swapped MFMA form computes the same sums in
// the same order); what changed is how the work is issued: every epilogue on 16-byte quads (lane = row), residual rows
// and the pad flag requested at kernel start (one length load per lane instead of one per accumulator register), the
// 31-tap depthwise conv + LayerNorm + swish in registers from two tap chunks (no LDS staging, no dependent global -> LDS
// round trips), the hidden-tile / QKV stores sliced into the next unit's MFMA stream.  R = 16 (v_mfma_f32_16x16x4_f32):
// for launches whose 32-row blocks would leave more than half of the CUs idle.
// H3 (R = 32): the feed-forward module on the fp16 x3 route (h3.h; w.ff1_w1 / w.ff1_w2 are then the re-packed weights; the
// operand planes take bufH's place and 34 KB behind it)
template <int R, bool H3>
__device__ __forceinline__ void sq_mid_body(const float* __restrict__ ctx, const float* __restrict__ x, float* __restrict__ x2,
                                            float* __restrict__ g, float* __restrict__ xhat_out, const SqLayerW& w,
                                            const int64_t* __restrict__ lens, int M, int Tp, int mask_mul, int n_chunks,
                                            const PadSkip& ps) {
  using T = RBT<R>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, T::ROWS, M);
  if (blk < 0) return;
  float* bufX = smem;
  float* bufH = bufX + T::ROWS * kLda;  // 2 buffers; bufH[0] doubles as the ctx staging tile
  const LaneT<R> L;
  const int r0 = blk * T::ROWS;
  const int valid = min(T::ROWS, M - r0);
  typename T::Ring ring;
  const f32x4* seg_o = w.wo + (size_t)L.tile() * kTs256;
  const f32x4* seg_val = w.pw1 + (size_t)L.tile() * kTs256;
  const f32x4* seg_gate = w.pw1 + (size_t)(8 + L.tile()) * kTs256;
  rbt_prime(ring, seg_o);
  rbt_load_rows<R>(bufH, ctx + (size_t)r0 * kD, valid);
  size_t gq[T::NQ];  // place of quad q in x / g (row clamped: branch-free loads)
#pragma unroll
  for (int q = 0; q < T::NQ; ++q) gq[q] = (size_t)(r0 + min(L.row(q), valid - 1)) * kD + L.col(q);
  const int n_ok = L.quads_ok(valid);
  f32x4 res[T::NQ];
#pragma unroll
  for (int q = 0; q < T::NQ; ++q) res[q] = *reinterpret_cast<const f32x4*>(x + gq[q]);
  const PadLaneT<R> pl(lens, r0, M, Tp, mask_mul);
  __syncthreads();
  {
    typename T::Acc acc;
    T::zero(acc);
    rbt_gemm<kG256>(bufH, kLda, seg_o, w.ff1_w1 + (size_t)L.tile() * kTs256, ring, acc);
#pragma unroll
    for (int q = 0; q < T::NQ; ++q) {
      const f32x4 bo = *reinterpret_cast<const f32x4*>(w.bo + L.col(q));
      const f32x4 a = T::quad(acc, q);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = q < n_ok ? res[q][e] + (a[e] + bo[e]) : 0.f;
      *reinterpret_cast<f32x4*>(bufX + L.off(q)) = v;
    }
  }
  __syncthreads();
  rbt_layernorm<R>(bufX, bufX, w.ln1_g, w.ln1_b, 1e-5f);
  __syncthreads();
  typename T::Acc acc2;
  T::zero(acc2);
  if constexpr (H3) {
    static_assert(R == 32, "fp16 x3: the 8-wave 32-row form");
    _Float16* pa = reinterpret_cast<_Float16*>(bufH);
    ffn_phase_h3(bufX, pa, pa + 2 * kPlaneH, w.ff1_w1, w.ff1_b1, w.ff1_w2, n_chunks, seg_val, ring, acc2.v);
  } else {
    ffn_phase_t<R>(bufX, bufH, w.ff1_w1, w.ff1_b1, w.ff1_w2, n_chunks, seg_val, ring, acc2);
  }
  residual_epilogue_q<R>(bufX, acc2, w.ff1_b2, 1.0f);
  __syncthreads();
  rbt_layernorm<R>(bufX, bufX, w.ln2_g, w.ln2_b, 1e-5f);
  rbt_store_rows<R>(x2 + (size_t)r0 * kD, bufX, valid);  // (same wave -> row mapping as the LayerNorm)
  if (xhat_out) {  // streaming: what the reference keeps as cnn_cache = the scaled conv-module input
    const f32x4 sc = *reinterpret_cast<const f32x4*>(w.cm_scale + 4 * L.lane);
    const f32x4 sb = *reinterpret_cast<const f32x4*>(w.cm_bias + 4 * L.lane);
    for (int row = L.wave; row < valid; row += T::WAVES)
      *reinterpret_cast<f32x4*>(xhat_out + (size_t)(r0 + row) * kD + 4 * L.lane) =
          sc * *reinterpret_cast<const f32x4*>(bufX + row * kLda + 4 * L.lane) + sb;
  }
  __syncthreads();
  {
    typename T::Acc av, ag;
    T::zero(av);
    T::zero(ag);
    rbt_gemm<kG256>(bufX, kLda, seg_val, seg_gate, ring, av);
    rbt_gemm<kG256>(bufX, kLda, seg_gate, nullptr, ring, ag);
#pragma unroll
    for (int q = 0; q < T::NQ; ++q) {
      const f32x4 bval = *reinterpret_cast<const f32x4*>(w.pw1_b + L.col(q));
      const f32x4 bgate = *reinterpret_cast<const f32x4*>(w.pw1_b + kD + L.col(q));
      const f32x4 gpad = *reinterpret_cast<const f32x4*>(w.glu_pad + L.col(q));
      const f32x4 a = T::quad(av, q), b = T::quad(ag, q);
      const f32x2 s0 = sigmoid2(f32x2{b[0] + bgate[0], b[1] + bgate[1]});
      const f32x2 s1 = sigmoid2(f32x2{b[2] + bgate[2], b[3] + bgate[3]});
      f32x4 o = {(a[0] + bval[0]) * s0[0], (a[1] + bval[1]) * s0[1], (a[2] + bval[2]) * s1[0], (a[3] + bval[3]) * s1[1]};
      if (pl.pad(q)) o = gpad;
      if (q < n_ok) *reinterpret_cast<f32x4*>(g + gq[q]) = o;
    }
  }
}

template <int R>
__global__ __launch_bounds__(RBT<R>::THREADS) void k_sq_mid_t(const float* __restrict__ ctx, const float* __restrict__ x,
                                                       float* __restrict__ x2, float* __restrict__ g,
                                                       float* __restrict__ xhat_out, SqLayerW w,
                                                       const int64_t* __restrict__ lens, int M, int Tp, int mask_mul,
                                                       int n_chunks, PadSkip ps) {
  sq_mid_body<R, false>(ctx, x, x2, g, xhat_out, w, lens, M, Tp, mask_mul, n_chunks, ps);
}
__global__ __launch_bounds__(kThreads) void k_sq_mid_h3(const float* __restrict__ ctx, const float* __restrict__ x,
                                                        float* __restrict__ x2, float* __restrict__ g,
                                                        float* __restrict__ xhat_out, SqLayerW w,
                                                        const int64_t* __restrict__ lens, int M, int Tp, int mask_mul,
                                                        int n_chunks, PadSkip ps) {
  sq_mid_body<32, true>(ctx, x, x2, g, xhat_out, w, lens, M, Tp, mask_mul, n_chunks, ps);
}

template <int R, int KS, bool H3>
__device__ __forceinline__ void sq_tail_body(const float* __restrict__ g, const float* __restrict__ x2, float* __restrict__ x_out,
                                             float* __restrict__ qkv_next, const SqLayerW& w, const f32x4* __restrict__ wqkv_next,
                                             const float* __restrict__ bqkv_next, const int64_t* __restrict__ lens, int M, int Tp,
                                             int mask_mul, int n_chunks, const PadSkip& ps, int left_ctx) {
  using T = RBT<R>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, T::ROWS, M);
  if (blk < 0) return;
  float* bufX = smem;
  float* bufA = bufX + T::ROWS * kLda;
  float* bufH = bufA + T::ROWS * kLda;
  const LaneT<R> L;
  const int r0 = blk * T::ROWS;
  const int valid = min(T::ROWS, M - r0);
  typename T::Ring ring;
  const f32x4* seg_pw2 = w.pw2 + (size_t)L.tile() * kTs256;
  const PadLaneT<R> pl(lens, r0, M, Tp, mask_mul);
  const int n_ok = L.quads_ok(valid);
  // weight stream and the residual rows of the pointwise_conv2 epilogue: requested after the depthwise multiply-adds
  // (whose register window they would otherwise compete with), in flight during the conv-module LayerNorm; branch-free
  f32x4 res[T::NQ];
  dwconv_ln_phase_t<R, KS>(g, bufA, w.dw_w, w.dw_b, w.glu_pad, w.ln_cm_g, w.ln_cm_b, w.cm_eps, r0, M, Tp, left_ctx, [&] {
    rbt_prime(ring, seg_pw2);
#pragma unroll
    for (int q = 0; q < T::NQ; ++q)
      res[q] = *reinterpret_cast<const f32x4*>(x2 + (size_t)(r0 + min(L.row(q), valid - 1)) * kD + L.col(q));
  });
  __syncthreads();
  {
    typename T::Acc acc;
    T::zero(acc);
    rbt_gemm<kG256>(bufA, kLda, seg_pw2, w.ff2_w1 + (size_t)L.tile() * kTs256, ring, acc);
#pragma unroll
    for (int q = 0; q < T::NQ; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(w.pw2_b + L.col(q));
      const f32x4 a = T::quad(acc, q);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = q < n_ok ? res[q][e] + (pl.pad(q) ? 0.f : a[e] + bv[e]) : 0.f;
      *reinterpret_cast<f32x4*>(bufX + L.off(q)) = o;
    }
  }
  __syncthreads();
  rbt_layernorm<R>(bufX, bufX, w.ln3_g, w.ln3_b, 1e-5f);
  __syncthreads();
  typename T::Acc acc2;
  T::zero(acc2);
  if constexpr (H3) {
    static_assert(R == 32, "fp16 x3: the 8-wave 32-row form");
    _Float16* pa = reinterpret_cast<_Float16*>(bufA);  // (bufA is free behind pointwise_conv2; the planes run on through bufH)
    ffn_phase_h3(bufX, pa, pa + 2 * kPlaneH, w.ff2_w1, w.ff2_b1, w.ff2_w2, n_chunks,
                 wqkv_next ? wqkv_next + (size_t)L.tile() * kTs256 : nullptr, ring, acc2.v);
  } else {
    ffn_phase_t<R>(bufX, bufH, w.ff2_w1, w.ff2_b1, w.ff2_w2, n_chunks, wqkv_next ? wqkv_next + (size_t)L.tile() * kTs256 : nullptr,
                   ring, acc2);
  }
  residual_epilogue_q<R>(bufX, acc2, w.ff2_b2, 1.0f);
  __syncthreads();
  rbt_layernorm<R>(bufX, bufX, w.ln4_g, w.ln4_b, 1e-5f);
  rbt_store_rows<R>(x_out + (size_t)r0 * kD, bufX, valid);
  if (wqkv_next) {
    __syncthreads();
    qkv_phase_t<R>(bufX, qkv_next, wqkv_next, bqkv_next, r0, valid, ring);
  }
}
template <int R, int KS>
__global__ __launch_bounds__(RBT<R>::THREADS) void k_sq_tail_t(const float* __restrict__ g, const float* __restrict__ x2,
                                                        float* __restrict__ x_out, float* __restrict__ qkv_next,
                                                        SqLayerW w, const f32x4* __restrict__ wqkv_next,
                                                        const float* __restrict__ bqkv_next,
                                                        const int64_t* __restrict__ lens, int M, int Tp, int mask_mul,
                                                        int n_chunks, PadSkip ps, int left_ctx) {
  sq_tail_body<R, KS, false>(g, x2, x_out, qkv_next, w, wqkv_next, bqkv_next, lens, M, Tp, mask_mul, n_chunks, ps, left_ctx);
}
template <int KS>
__global__ __launch_bounds__(kThreads) void k_sq_tail_h3(const float* __restrict__ g, const float* __restrict__ x2,
                                                         float* __restrict__ x_out, float* __restrict__ qkv_next, SqLayerW w,
                                                         const f32x4* __restrict__ wqkv_next, const float* __restrict__ bqkv_next,
                                                         const int64_t* __restrict__ lens, int M, int Tp, int mask_mul,
                                                         int n_chunks, PadSkip ps, int left_ctx) {
  sq_tail_body<32, KS, true>(g, x2, x_out, qkv_next, w, wqkv_next, bqkv_next, lens, M, Tp, mask_mul, n_chunks, ps, left_ctx);
}

constexpr size_t kLdsSqMid = 3 * kRows * kLda * sizeof(float);

// ---- split route for under-filled launches (split_route_kernels.hip, k_conv_pre / k_ffn_part / k_ffn_join): K_B and K_C
// cut at their feed-forward modules.  K_B = k_sq_oproj -> FFN1 split -> k_sq_pw1glu ; K_C = k_conv_pre -> FFN2 split
// [-> k_sq_qkv of the next layer] ----
// x1 = LN1(x + ctx Wo + bo)
__global__ __launch_bounds__(kThreads) void k_sq_oproj(const float* __restrict__ ctx, const float* __restrict__ x,
                                                       float* __restrict__ x1, SqLayerW w, int M, PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (pad_block_skippable(ps, blockIdx.x * kRows, kRows, M)) return;
  float* bufX = smem;
  float* bufH = bufX + kRows * kLda;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blockIdx.x * kRows;
  const int valid = min(kRows, M - r0);
  const int col = wave * 32 + (lane & 31);
  BRing<1> ring;
  const f32x4* seg_o = w.wo + (size_t)wave * kTs256;
  ring_prime(ring, seg_o, 0);
  rb_load_rows(bufH, kLda, ctx + (size_t)r0 * kD, kRows, valid);
  __syncthreads();
  float res[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) res[r] = x[(size_t)(r0 + min(acc_row(r, lane), valid - 1)) * kD + col];
  f32x16 acc[1][1];
  acc_zero(acc);
  rb_gemm<1, 1, kG256>(bufH, kLda, seg_o, 0, nullptr, 0, ring, acc);
  const float bv = w.bo[col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane);
    bufX[row * kLda + col] = (row < valid) ? res[r] + (acc[0][0][r] + bv) : 0.f;
  }
  __syncthreads();
  rb_layernorm(bufX, bufX, kLda, kRows, w.ln1_g, w.ln1_b, 1e-5f);
  rb_store_rows(x1 + (size_t)r0 * kD, bufX, kLda, kRows, valid);
}
// g = GLU(pw1(x2)) with PAD frames = GLU(pw1(0)); xhat_out (streaming) = the scaled conv-module input
__global__ __launch_bounds__(kThreads) void k_sq_pw1glu(const float* __restrict__ x2, float* __restrict__ g,
                                                        float* __restrict__ xhat_out, SqLayerW w,
                                                        const int64_t* __restrict__ lens, int M, int Tp, int mask_mul,
                                                        PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (pad_block_skippable(ps, blockIdx.x * kRows, kRows, M)) return;
  float* bufX = smem;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blockIdx.x * kRows;
  const int valid = min(kRows, M - r0);
  const int col = wave * 32 + (lane & 31);
  BRing<1> ring;
  const f32x4* seg_val = w.pw1 + (size_t)wave * kTs256;
  const f32x4* seg_gate = w.pw1 + (size_t)(8 + wave) * kTs256;
  ring_prime(ring, seg_val, 0);
  rb_load_rows(bufX, kLda, x2 + (size_t)r0 * kD, kRows, valid);
  if (xhat_out) {  // (same wave -> row mapping as the load)
    const f32x4 sc = *reinterpret_cast<const f32x4*>(w.cm_scale + 4 * lane);
    const f32x4 sb = *reinterpret_cast<const f32x4*>(w.cm_bias + 4 * lane);
    for (int row = wave; row < valid; row += kWaves)
      *reinterpret_cast<f32x4*>(xhat_out + (size_t)(r0 + row) * kD + 4 * lane) =
          sc * *reinterpret_cast<const f32x4*>(bufX + row * kLda + 4 * lane) + sb;
  }
  __syncthreads();
  f32x16 av[1][1], ag[1][1];
  acc_zero(av);
  acc_zero(ag);
  rb_gemm<1, 1, kG256>(bufX, kLda, seg_val, 0, seg_gate, 0, ring, av);
  rb_gemm<1, 1, kG256>(bufX, kLda, seg_gate, 0, nullptr, 0, ring, ag);
  const float bval = w.pw1_b[col], bgate = w.pw1_b[kD + col];
  const float gpad = w.glu_pad[col];
  PadRows is_pad{lens, r0, Tp, M, mask_mul};
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = acc_row(r, lane);
    float v = (av[0][0][r] + bval) * sigmoidf(ag[0][0][r] + bgate);
    if (is_pad(row)) v = gpad;
    if (row < valid) g[(size_t)(r0 + row) * kD + col] = v;
  }
}

// K_C: x3 = LN3(x2 + mask(pw2(swish(LN(dwconv(g)))))) ; x4 = LN4(x3 + FFN2(x3)) ; [qkv of the next layer]
template <int KS, bool STREAM>
__global__ __launch_bounds__(kThreads) void k_sq_tail(const float* __restrict__ g, const float* __restrict__ g_hist,
                                                      const float* __restrict__ x2,
                                                      float* __restrict__ x_out, float* __restrict__ qkv_next,
                                                      SqLayerW w, const f32x4* __restrict__ wqkv_next,
                                                      const float* __restrict__ bqkv_next,
                                                      const int64_t* __restrict__ lens, int M, int Tp, int mask_mul,
                                                      int n_chunks, PadSkip ps, int left_ctx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (pad_block_skippable(ps, blockIdx.x * kRows, kRows, M)) return;
  float* bufX = smem;
  float* bufA = bufX + kRows * kLda;
  float* bufH = bufA + kRows * kLda;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blockIdx.x * kRows;
  const int valid = min(kRows, M - r0);
  const int col = wave * 32 + (lane & 31);
  BRing<1> ring;
  const f32x4* seg_pw2 = w.pw2 + (size_t)wave * kTs256;
  ring_prime(ring, seg_pw2, 0);
  // residual rows / pad flags of the pointwise_conv2 epilogue, requested first thing and branch-free (see k_conv_ffn)
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
  {
    f32x16 acc[1][1];
    acc_zero(acc);
    rb_gemm<1, 1, kG256>(bufA, kLda, seg_pw2, 0, w.ff2_w1 + (size_t)wave * kTs256, 0, ring, acc);
    const float bv = w.pw2_b[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, lane);
      const float c = ((pad_bits >> r) & 1u) ? 0.f : acc[0][0][r] + bv;
      bufX[row * kLda + col] = (row < valid) ? res[r] + c : 0.f;
    }
  }
  __syncthreads();
  rb_layernorm(bufX, bufX, kLda, kRows, w.ln3_g, w.ln3_b, 1e-5f);
  __syncthreads();
  f32x16 acc2[1][1];
  acc_zero(acc2);
  ffn_phase(bufX, bufH, w.ff2_w1, w.ff2_b1, w.ff2_w2, n_chunks,
            wqkv_next ? wqkv_next + (size_t)wave * kTs256 : nullptr, ring, acc2);
  residual_epilogue(bufX, acc2, w.ff2_b2, 1.0f);
  __syncthreads();
  rb_layernorm(bufX, bufX, kLda, kRows, w.ln4_g, w.ln4_b, 1e-5f);
  rb_store_rows(x_out + (size_t)r0 * kD, bufX, kLda, kRows, valid);
  if (wqkv_next) {
    __syncthreads();
    qkv_from_lds(bufX, qkv_next, wqkv_next, bqkv_next, r0, valid, ring);
  }
}
constexpr size_t kLdsSqTail = 4 * kRows * kLda * sizeof(float);

// Time reduction (TimeReductionLayerStream, time_reduction.py:183-206): zero PAD frames, depthwise
// Conv1D(k=1, stride 2) = per-channel scale+bias of every second frame, pointwise Conv1D 256->256;
// then the QKV projection of the first reduced layer.  Rows here are REDUCED rows (b, j) <- frame 2j.
__global__ __launch_bounds__(kThreads) void k_sq_reduce(const float* __restrict__ x, float* __restrict__ xr,
                                                        float* __restrict__ qkv, SqReduceW rw,
                                                        const f32x4* __restrict__ wqkv, const float* __restrict__ bqkv,
                                                        const int64_t* __restrict__ lens, int B, int Tp, int Tr,
                                                        PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (pad_block_skippable(ps, blockIdx.x * kRows, kRows, B * Tr)) return;  // (ps: the reduced OUTPUT rows)
  float* bufA = smem;
  float* bufX = bufA + kRows * kLda;
  const int lane = lane_id(), wave = wave_id();
  const int Mr = B * Tr;
  const int r0 = blockIdx.x * kRows;
  const int valid = min(kRows, Mr - r0);
  const int col = wave * 32 + (lane & 31);
  BRing<1> ring;
  const f32x4* seg_pw = rw.pw + (size_t)wave * kTs256;
  ring_prime(ring, seg_pw, 0);
  {
    const f32x4 db = *reinterpret_cast<const f32x4*>(rw.dw_b + 4 * lane);
    const int pad = rw.ks > 2 ? rw.ks - 2 : 0;  // Conv1D(kernel ks, stride 2, padding max(0, ks - stride))
    for (int row = wave; row < kRows; row += kWaves) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < valid) {
        const int mr = r0 + row, b = mr / Tr, j = mr - b * Tr;
        v = db;
        for (int k = 0; k < rw.ks; ++k) {
          const int t = 2 * j - pad + k;
          if (t < 0 || t >= Tp) continue;                       // the conv's own zero padding
          if (lens && 4 * (int64_t)t >= lens[b]) continue;      // masked_fill(xs, mask_pad==0, 0)
          v += *reinterpret_cast<const f32x4*>(x + ((size_t)b * Tp + t) * kD + 4 * lane) *
               *reinterpret_cast<const f32x4*>(rw.dw_w + (size_t)k * kD + 4 * lane);
        }
      }
      *reinterpret_cast<f32x4*>(bufA + row * kLda + 4 * lane) = v;
    }
  }
  __syncthreads();
  {
    f32x16 acc[1][1];
    acc_zero(acc);
    rb_gemm<1, 1, kG256>(bufA, kLda, seg_pw, 0, wqkv + (size_t)wave * kTs256, 0, ring, acc);
    const float bv = rw.pw_b[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = acc_row(r, lane);
      float v = acc[0][0][r] + bv;
      bufX[row * kLda + col] = row < valid ? v : 0.f;
      if (row < valid) xr[(size_t)(r0 + row) * kD + col] = v;
    }
  }
  __syncthreads();
  qkv_from_lds(bufX, qkv, wqkv, bqkv, r0, valid, ring);
}

// Time recovery (encoder.py:219-230): repeat_interleave(x, 2) -> Linear -> + saved ; then QKV.
// Rows are full-resolution rows (b, t) <- reduced row (b, t >> 1).
__global__ __launch_bounds__(kThreads) void k_sq_recover(const float* __restrict__ xr, const float* __restrict__ saved,
                                                         float* __restrict__ x, float* __restrict__ qkv,
                                                         const f32x4* __restrict__ wrec, const float* __restrict__ brec,
                                                         const f32x4* __restrict__ wqkv, const float* __restrict__ bqkv,
                                                         int B, int Tp, int Tr, PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (pad_block_skippable(ps, blockIdx.x * kRows, kRows, B * Tp)) return;
  float* bufA = smem;
  float* bufX = bufA + kRows * kLda;
  const int lane = lane_id(), wave = wave_id();
  const int M = B * Tp;
  const int r0 = blockIdx.x * kRows;
  const int valid = min(kRows, M - r0);
  const int col = wave * 32 + (lane & 31);
  BRing<1> ring;
  const f32x4* seg = wrec + (size_t)wave * kTs256;
  ring_prime(ring, seg, 0);
  for (int row = wave; row < kRows; row += kWaves) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < valid) {
      const int m = r0 + row, b = m / Tp, t = m - b * Tp;
      v = *reinterpret_cast<const f32x4*>(xr + ((size_t)b * Tr + (t >> 1)) * kD + 4 * lane);
    }
    *reinterpret_cast<f32x4*>(bufA + row * kLda + 4 * lane) = v;
  }
  __syncthreads();
  {
    f32x16 acc[1][1];
    acc_zero(acc);
    rb_gemm<1, 1, kG256>(bufA, kLda, seg, 0, wqkv + (size_t)wave * kTs256, 0, ring, acc);
    const float bv = brec[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = acc_row(r, lane);
      float v = 0.f;
      if (row < valid) {
        v = saved[(size_t)(r0 + row) * kD + col] + (acc[0][0][r] + bv);
        x[(size_t)(r0 + row) * kD + col] = v;
      }
      bufX[row * kLda + col] = v;
    }
  }
  __syncthreads();
  qkv_from_lds(bufX, qkv, wqkv, bqkv, r0, valid, ring);
}

// in-place LayerNorm of [M][256] rows (preln, encoder.py:207)
__global__ __launch_bounds__(kThreads) void k_ln_rows(float* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ b, int M, PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (pad_block_skippable(ps, blockIdx.x * kRows, kRows, M)) return;
  const int r0 = blockIdx.x * kRows;
  const int valid = min(kRows, M - r0);
  rb_load_rows(smem, kLda, x + (size_t)r0 * kD, kRows, valid);
  rb_layernorm(smem, smem, kLda, kRows, g, b, 1e-5f);
  rb_store_rows(x + (size_t)r0 * kD, smem, kLda, kRows, valid);
}

// ---- launchers ----
static inline dim3 rb_grid(int M) { return dim3((M + kRows - 1) / kRows); }
constexpr size_t kLds1 = kRows * kLda * sizeof(float);
constexpr size_t kLds2 = 2 * kRows * kLda * sizeof(float);

void launch_sq_qkv(const float* x, float* qkv, const f32x4* wqkv, const float* bqkv, int M, hipStream_t st,
                   const PadSkip& ps) {
  PPASR_LAUNCH(k_sq_qkv, rb_grid(M), dim3(kThreads), ragged_lds(kLds1, ps, (int)rb_grid(M).x), st, x, qkv, wqkv, bqkv, M, ps);
}
// 16-row blocks always ask for more than half of a CU's LDS: one workgroup per CU, so that the (<= 256) active blocks of an
// under-filled or ragged launch spread over all CUs instead of pairing up on some
static size_t lds16(size_t own) { return own < kLdsExclusive ? kLdsExclusive : own; }
constexpr size_t kLdsSqMid16 = 3 * 16 * kLda * sizeof(float), kLdsSqTail16 = 4 * 16 * kLda * sizeof(float);

// fp16 x3 forms (h3.h): the residual tile + three operand tiles
constexpr size_t kLdsSqMidH3 = kRows * kLda * sizeof(float) + 3 * kH3TileBytes;
void launch_sq_mid(const float* ctx, const float* x, float* x2, float* g, float* xhat_out, const SqLayerW& w,
                   const int64_t* lens, int M, int Tp, int mask_mul, int n_chunks, hipStream_t st, const PadSkip& ps, int rows,
                   bool h3) {
  if (h3 && rows == 32)  // (w: the layer's fp16 x3 view; the caller decides h3 with sq_h3_route)
    PPASR_LAUNCH(k_sq_mid_h3, rb_grid(M), dim3(kThreads), kLdsSqMidH3, st, ctx, x, x2, g, xhat_out, w, lens, M, Tp, mask_mul,
                 n_chunks, ps);
  else if (rows == 16)
    PPASR_LAUNCH(k_sq_mid_t<16>, dim3((M + 15) / 16), dim3(kThreads), lds16(kLdsSqMid16), st, ctx, x, x2, g, xhat_out, w, lens,
                 M, Tp, mask_mul, n_chunks, ps);
  else if (rows == kW16)
    PPASR_LAUNCH(k_sq_mid_t<kW16>, rb_grid(M), dim3(1024), kLdsSqMid, st, ctx, x, x2, g, xhat_out, w, lens, M, Tp, mask_mul,
                 n_chunks, ps);
  else
    PPASR_LAUNCH(k_sq_mid_t<32>, rb_grid(M), dim3(kThreads), kLdsSqMid, st, ctx, x, x2, g, xhat_out, w, lens, M, Tp, mask_mul,
                 n_chunks, ps);
}
void launch_sq_tail(const float* g, const float* g_hist, const float* x2, float* x_out, float* qkv_next, const SqLayerW& w,
                    const f32x4* wqkv_next, const float* bqkv_next, const int64_t* lens, int M, int Tp, int mask_mul,
                    int n_chunks, int ksize, hipStream_t st, const PadSkip& ps, bool causal, int rows, bool h3) {
  const int left_ctx = causal ? ksize - 1 : (ksize - 1) / 2;
  if (h3 && rows == 32 && sq_h3_supported(ksize, Tp) && !g_hist) {  // (w: the layer's fp16 x3 view)
    if (ksize == 31)
      PPASR_LAUNCH(k_sq_tail_h3<31>, rb_grid(M), dim3(kThreads), kLdsSqTail + kH3ExtraLds, st, g, x2, x_out, qkv_next, w, wqkv_next,
                   bqkv_next, lens, M, Tp, mask_mul, n_chunks, ps, left_ctx);
    else
      PPASR_LAUNCH(k_sq_tail_h3<15>, rb_grid(M), dim3(kThreads), kLdsSqTail + kH3ExtraLds, st, g, x2, x_out, qkv_next, w, wqkv_next,
                   bqkv_next, lens, M, Tp, mask_mul, n_chunks, ps, left_ctx);
    return;
  }
  // the register depthwise conv needs a wave's rows to span at most two utterances (Tp >= rows per wave); streaming
  // chunks (g_hist) keep the LDS-staged form
  if (!g_hist && Tp >= 4 && (ksize == 31 || ksize == 15)) {
#define SQ_TAIL_T(R, KS, LDS)                                                                                         \
  PPASR_LAUNCH((k_sq_tail_t<R, KS>), dim3((M + RBT<R>::ROWS - 1) / RBT<R>::ROWS), dim3(RBT<R>::THREADS), LDS, st, g, x2, \
               x_out, qkv_next, w, wqkv_next, bqkv_next, lens, M, Tp, mask_mul, n_chunks, ps, left_ctx)
    if (rows == 16) {
      if (ksize == 31) SQ_TAIL_T(16, 31, lds16(kLdsSqTail16)); else SQ_TAIL_T(16, 15, lds16(kLdsSqTail16));
    } else if (rows == kW16) {
      if (ksize == 31) SQ_TAIL_T(kW16, 31, kLdsSqTail); else SQ_TAIL_T(kW16, 15, kLdsSqTail);
    } else {
      if (ksize == 31) SQ_TAIL_T(32, 31, kLdsSqTail); else SQ_TAIL_T(32, 15, kLdsSqTail);
    }
#undef SQ_TAIL_T
    return;
  }
#define SQ_TAIL(KS, STREAM)                                                                                          \
  PPASR_LAUNCH((k_sq_tail<KS, STREAM>), rb_grid(M), dim3(kThreads), kLdsSqTail, st, g, g_hist, x2, x_out, qkv_next, w, \
                     wqkv_next, bqkv_next, lens, M, Tp, mask_mul, n_chunks, ps, left_ctx)
  if (ksize == 31) {
    if (g_hist) SQ_TAIL(31, true); else SQ_TAIL(31, false);
  } else if (ksize == 15) {
    if (g_hist) SQ_TAIL(15, true); else SQ_TAIL(15, false);
  }
#undef SQ_TAIL
}
void launch_sq_oproj(const float* ctx, const float* x, float* x1, const SqLayerW& w, int M, hipStream_t st, const PadSkip& ps) {
  PPASR_LAUNCH(k_sq_oproj, rb_grid(M), dim3(kThreads), kLds2, st, ctx, x, x1, w, M, ps);
}
void launch_sq_pw1glu(const float* x2, float* g, float* xhat_out, const SqLayerW& w, const int64_t* lens, int M, int Tp,
                      int mask_mul, hipStream_t st, const PadSkip& ps) {
  PPASR_LAUNCH(k_sq_pw1glu, rb_grid(M), dim3(kThreads), kLds1, st, x2, g, xhat_out, w, lens, M, Tp, mask_mul, ps);
}
void launch_sq_reduce(const float* x, float* xr, float* qkv, const SqReduceW& rw, const f32x4* wqkv, const float* bqkv,
                      const int64_t* lens, int B, int Tp, int Tr, hipStream_t st, const PadSkip& ps) {
  PPASR_LAUNCH(k_sq_reduce, rb_grid(B * Tr), dim3(kThreads), ragged_lds(kLds2, ps, (int)rb_grid(B * Tr).x), st, x, xr, qkv,
               rw, wqkv, bqkv, lens, B, Tp, Tr, ps);
}
void launch_sq_recover(const float* xr, const float* saved, float* x, float* qkv, const f32x4* wrec, const float* brec,
                       const f32x4* wqkv, const float* bqkv, int B, int Tp, int Tr, hipStream_t st, const PadSkip& ps) {
  PPASR_LAUNCH(k_sq_recover, rb_grid(B * Tp), dim3(kThreads), ragged_lds(kLds2, ps, (int)rb_grid(B * Tp).x), st, xr, saved,
               x, qkv, wrec, brec, wqkv, bqkv, B, Tp, Tr, ps);
}
void launch_ln_rows(float* x, const float* g, const float* b, int M, hipStream_t st, const PadSkip& ps) {
  PPASR_LAUNCH(k_ln_rows, rb_grid(M), dim3(kThreads), kLds1, st, x, g, b, M, ps);
}

unsigned int* squeezeformer_h3_ovf_counter() { return h3_ovf_counter(); }

hipError_t configure_squeezeformer_kernels() {
  hipError_t e;
#define SET_LDS(fn, bytes)                                                                                     \
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
  if (e != hipSuccess) return e;
  SET_LDS(k_sq_mid_t<32>, kLdsSqMid);
  SET_LDS(k_sq_mid_h3, kLdsSqMidH3);
  SET_LDS(k_sq_tail_h3<31>, kLdsSqTail + kH3ExtraLds);
  SET_LDS(k_sq_tail_h3<15>, kLdsSqTail + kH3ExtraLds);
  SET_LDS(k_sq_mid_t<16>, kLdsExclusive);
  SET_LDS(k_sq_mid_t<kW16>, kLdsSqMid);
  SET_LDS((k_sq_tail_t<kW16, 31>), kLdsSqTail);
  SET_LDS((k_sq_tail_t<kW16, 15>), kLdsSqTail);
  SET_LDS((k_sq_tail_t<32, 31>), kLdsSqTail);
  SET_LDS((k_sq_tail_t<32, 15>), kLdsSqTail);
  SET_LDS((k_sq_tail_t<16, 31>), kLdsExclusive);
  SET_LDS((k_sq_tail_t<16, 15>), kLdsExclusive);
  SET_LDS((k_sq_tail<31, false>), kLdsSqTail);
  SET_LDS((k_sq_tail<15, false>), kLdsSqTail);
  SET_LDS((k_sq_tail<31, true>), kLdsSqTail);
  SET_LDS((k_sq_tail<15, true>), kLdsSqTail);
  SET_LDS(k_sq_reduce, kLdsExclusive);  // (ragged launches: ragged_lds)
  SET_LDS(k_sq_oproj, kLds2);
  SET_LDS(k_sq_recover, kLdsExclusive);
  SET_LDS(k_sq_qkv, kLdsExclusive);
#undef SET_LDS
  return hipSuccess;
}

}  // namespace ppasr
