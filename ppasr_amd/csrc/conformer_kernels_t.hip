// conformer_kernels_t.hip -- the Conformer-family layer kernels on the v_mfma_f32_16x16x4_f32 block forms of csrc/rbt.h
// (phases_t.h), on the SAME packed weights as the 32-row kernels of conformer_kernels.hip:
//   * 16-row blocks for UNDER-FILLED launches: a call whose 32-row blocks would fill at most half of the chip (small
//     batches, the half-rate layers of a small Efficient-Conformer batch) takes twice as many workgroups, each half as
//     long.  The attention between them is the stand-alone k_attention_t (values row-major in qkv).
//   * 32 rows on 16 waves (kW16) for FULL launches: four waves per SIMD keep the matrix pipe fed where the 8-wave
//     kernels idle it ~16 % of the time (rbt.h); k_conv_ffn_t<kW16, KS, NEXT> replaces k_conv_ffn inside the fused route
//     (it writes the next layer's values in the fragment order k_attn_out_glu reads), k_ffn_qkv_t<kW16> the first S1.
// Same phases, same order of operations per output element as k_ffn_qkv / k_out_glu / k_conv_ffn (encoder.py:380-429,
// convolution.py:104-140); the sums inside a 16-wide k step are taken in another order (1e-6 relative).
#include "conformer_kernels.h"
#include "launch.h"
#include "phases_t.h"

namespace ppasr {

#ifdef PPASR_PHASE_TS
}  // namespace ppasr
extern "C" __attribute__((visibility("default"))) int ppasr_debug_read_phase_ts_t(long long* out) {  // (tools/phase_ts.py --t)
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ppasr::g_phase_ts), sizeof(long long) * 128);
}
namespace ppasr {
#endif

// S1 on LDS-resident rows: x1 = x + 0.5 FFN_macaron(LN(x)) ; qkv = LN_mha(x1) [Wq|Wk|Wv] + b.  `ring` streams w.ffm_w1.
template <int R>
__device__ __forceinline__ void ffn_qkv_body_t(float* bufX, float* bufA, float* bufH, float* __restrict__ x1,
                                               float* __restrict__ qkv, const LayerW& w, int r0, int valid, int n_chunks,
                                               typename RBT<R>::Ring& ring, VtOut vt = VtOut{}) {
  using T = RBT<R>;
  const LaneT<R> L;
  rbt_layernorm<R>(bufX, bufA, w.ln_mac_g, w.ln_mac_b, 1e-5f);
  __syncthreads();
  PPASR_TS(8);
  typename T::Acc acc2;
  T::zero(acc2);
  ffn_phase_t<R>(bufA, bufH, w.ffm_w1, w.ffm_b1, w.ffm_w2, n_chunks, w.wqkv + (size_t)L.tile() * kTs256, ring, acc2);
  PPASR_TS(9);
  residual_epilogue_q<R>(bufX, acc2, w.ffm_b2, 0.5f);
  __syncthreads();
  PPASR_TS(10);
  rbt_store_rows<R>(x1 + (size_t)r0 * kD, bufX, valid);
  rbt_layernorm<R>(bufX, bufA, w.ln_mha_g, w.ln_mha_b, 1e-5f);
  __syncthreads();
  PPASR_TS(11);
  qkv_phase_t<R>(bufA, qkv, w.wqkv, w.bqkv, r0, valid, ring, vt.vt, vt.stride);
  PPASR_TS(14);
}

template <int R>
__global__ __launch_bounds__(RBT<R>::THREADS) void k_ffn_qkv_t(const float* __restrict__ x_in, float* __restrict__ x1,
                                                               float* __restrict__ qkv, LayerW w, int M, int n_chunks,
                                                               PadSkip ps, VtOut vt) {
  using T = RBT<R>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, T::ROWS, M);
  if (blk < 0) return;
  float* bufX = smem;
  float* bufA = bufX + T::ROWS * kLda;
  float* bufH = bufA + T::ROWS * kLda;
  const int r0 = blk * T::ROWS;
  const int valid = min(T::ROWS, M - r0);
  typename T::Ring ring;
  rbt_prime(ring, w.ffm_w1 + (size_t)T::tile(wave_id()) * kTs256);
  rbt_load_rows<R>(bufX, x_in + (size_t)r0 * kD, valid);  // (same wave -> row mapping as the LayerNorm that follows)
  ffn_qkv_body_t<R>(bufX, bufA, bufH, x1, qkv, w, r0, valid, n_chunks, ring, vt);
}

// S3: x2 = x1 + ctx Wo + bo ; g = GLU(pointwise_conv1(mask(LN_conv(x2))))   (attention.py:126, encoder.py:399-409)
// xhat_out != nullptr: the LayerNorm'd (pad-masked) rows are also stored (streaming: what the reference keeps as cnn_cache;
// split route: the input of k_pw1_glu_cols_t); stop_after_ln: the launch ends there (split route)
template <int R>
__global__ __launch_bounds__(RBT<R>::THREADS) void k_out_glu_t(const float* __restrict__ ctx, const float* __restrict__ x1,
                                                               float* __restrict__ x2, float* __restrict__ g, LayerW w,
                                                               const int64_t* __restrict__ lens, int M, int Tp,
                                                               int mask_mul, PadSkip ps, float* __restrict__ xhat_out,
                                                               int stop_after_ln) {
  using T = RBT<R>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, T::ROWS, M);
  if (blk < 0) return;
  float* bufX = smem;
  float* bufA = bufX + T::ROWS * kLda;
  const LaneT<R> L;
  const int r0 = blk * T::ROWS;
  const int valid = min(T::ROWS, M - r0);
  typename T::Ring ring;
  const f32x4* seg_o = w.wo + (size_t)L.tile() * kTs256;
  const f32x4* seg_val = w.pw1 + (size_t)L.tile() * kTs256;
  const f32x4* seg_gate = w.pw1 + (size_t)(8 + L.tile()) * kTs256;
  rbt_prime(ring, seg_o);
  rbt_load_rows<R>(bufA, ctx + (size_t)r0 * kD, valid);
  size_t gq[T::NQ];  // place of quad q in the row-major [M][256] matrices (row clamped: branch-free loads)
#pragma unroll
  for (int q = 0; q < T::NQ; ++q) gq[q] = (size_t)(r0 + min(L.row(q), valid - 1)) * kD + L.col(q);
  const int n_ok = L.quads_ok(valid);
  f32x4 res[T::NQ];
#pragma unroll
  for (int q = 0; q < T::NQ; ++q) res[q] = *reinterpret_cast<const f32x4*>(x1 + gq[q]);
  __syncthreads();
  {
    typename T::Acc acc;
    T::zero(acc);
    rbt_gemm<kG256>(bufA, kLda, seg_o, seg_val, ring, acc);
#pragma unroll
    for (int q = 0; q < T::NQ; ++q) {
      const f32x4 bo = *reinterpret_cast<const f32x4*>(w.bo + L.col(q));
      const f32x4 a = T::quad(acc, q);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = res[q][e] + (a[e] + bo[e]);
      if (q < n_ok) *reinterpret_cast<f32x4*>(x2 + gq[q]) = v;
      else v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(bufX + L.off(q)) = v;
    }
  }
  __syncthreads();
  rbt_layernorm<R, false>(bufX, bufA, w.ln_conv_g, w.ln_conv_b, 1e-5f, PadRows{lens, r0, Tp, M, mask_mul});
  // (LayerNorm and row store use the same wave -> row mapping: no barrier between them)
  if (xhat_out) rbt_store_rows<R>(xhat_out + (size_t)r0 * kD, bufA, valid);
  if (stop_after_ln) return;
  __syncthreads();
  {
    typename T::Acc av, ag;
    T::zero(av);
    T::zero(ag);
    rbt_gemm<kG256>(bufA, kLda, seg_val, seg_gate, ring, av);
    rbt_gemm<kG256>(bufA, kLda, seg_gate, nullptr, ring, ag);
#pragma unroll
    for (int q = 0; q < T::NQ; ++q) {
      const f32x4 bval = *reinterpret_cast<const f32x4*>(w.pw1_b + L.col(q));
      const f32x4 bgate = *reinterpret_cast<const f32x4*>(w.pw1_b + kD + L.col(q));
      const f32x4 a = T::quad(av, q), b = T::quad(ag, q);
      const f32x2 s0 = sigmoid2(f32x2{b[0] + bgate[0], b[1] + bgate[1]});
      const f32x2 s1 = sigmoid2(f32x2{b[2] + bgate[2], b[3] + bgate[3]});
      const f32x4 o = {(a[0] + bval[0]) * s0[0], (a[1] + bval[1]) * s0[1], (a[2] + bval[2]) * s1[0], (a[3] + bval[3]) * s1[1]};
      if (q < n_ok) *reinterpret_cast<f32x4*>(g + gq[q]) = o;
    }
  }
}

// pointwise_conv1 + GLU of LayerNorm'd (and pad-masked) rows on 16-row blocks, the 256 output columns over gridDim.y = 2
// workgroups (conformer_kernels.hip k_pw1_glu_cols): waves 0-3 compute the VALUE tiles of the workgroup's 128 columns,
// waves 4-7 the GATE tiles of the same columns; values cross to the gate waves through LDS.
template <int R>
__global__ __launch_bounds__(RBT<R>::THREADS) void k_pw1_glu_cols_t(const float* __restrict__ xhat, float* __restrict__ g,
                                                                    LayerW w, int M, PadSkip ps, float* __restrict__ hist,
                                                                    int lo, const float* __restrict__ hist_scale,
                                                                    const float* __restrict__ hist_bias,
                                                                    const int64_t* __restrict__ lens, int Tp, int mask_mul) {
  using T = RBT<R>;
  static_assert(R == 16, "the quad view below is the 16-row form's");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, T::ROWS, M);
  if (blk < 0) return;
  float* bufA = smem;
  float* vals = bufA + T::ROWS * kLda;  // [ROWS][132]
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * T::ROWS;
  const int valid = min(T::ROWS, M - r0);
  const int is_gate = wave >> 2, t = wave & 3, y = blockIdx.y;
  const f32x4* seg = w.pw1 + (size_t)((is_gate ? 8 : 0) + 4 * y + t) * kTs256;
  typename T::Ring ring;
  rbt_prime(ring, seg);
  rbt_load_rows<R>(bufA, xhat + (size_t)r0 * kD, valid);
  // hist != nullptr (one streaming session, M <= 16 rows, lo <= 30): the session's conv-module input history of this layer
  // moves on by the chunk's rows -- hist <- last `lo` rows of concat(hist, xhat) (stream_kernels.hip k_hist_update: read all,
  // barrier, write) -- in workgroup y = 0 of this launch instead of a launch of its own (12 per chunk, 4.7 us each)
  f32x4 moved[4];
  const bool mover = hist != nullptr && y == 0 && blockIdx.x == 0;  // (any M: every row is read before the barrier)
  if (mover) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = (int)threadIdx.x + T::THREADS * i;
      if (idx < lo * 64) {
        const int j = M + (idx >> 6), c4 = idx & 63;  // row of concat(hist, xhat)
        if (j < lo) {
          moved[i] = *reinterpret_cast<const f32x4*>(hist + (size_t)j * kD + 4 * c4);
        } else {  // (hist_scale: Squeezeformer keeps ada_scale * x + ada_bias, its pointwise_conv1 here has the scale folded in)
          moved[i] = *reinterpret_cast<const f32x4*>(xhat + (size_t)(j - lo) * kD + 4 * c4);
          if (hist_scale)
            moved[i] = *reinterpret_cast<const f32x4*>(hist_scale + 4 * c4) * moved[i] + *reinterpret_cast<const f32x4*>(hist_bias + 4 * c4);
        }
      }
    }
  }
  __syncthreads();
  if (mover) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = (int)threadIdx.x + T::THREADS * i;
      if (idx < lo * 64) *reinterpret_cast<f32x4*>(hist + (size_t)(idx >> 6) * kD + 4 * (idx & 63)) = moved[i];
    }
  }
  typename T::Acc acc;
  T::zero(acc);
  rbt_gemm<kG256>(bufA, kLda, seg, nullptr, ring, acc);
  const int row = lane & 15;
#pragma unroll
  for (int q = 0; q < T::NQ; ++q) {
    const int c128 = 32 * t + 16 * q + 4 * (lane >> 4);  // column inside the workgroup's 128
    const f32x4 bv = *reinterpret_cast<const f32x4*>(w.pw1_b + (is_gate ? kD : 0) + 128 * y + c128);
    const f32x4 a = T::quad(acc, q) + bv;
    if (!is_gate) *reinterpret_cast<f32x4*>(vals + row * 132 + c128) = a;
    else acc.s[q] = a;
  }
  __syncthreads();
  if (is_gate && row < valid) {
    // (lens != nullptr -- Squeezeformer's batched launches: PAD frames read GLU(pointwise_conv1(0)) = w.glu_pad, k_sq_pw1glu)
    const bool pad = lens != nullptr && PadRows{lens, r0, Tp, M, mask_mul}(row);
#pragma unroll
    for (int q = 0; q < T::NQ; ++q) {
      const int c128 = 32 * t + 16 * q + 4 * (lane >> 4);
      const f32x4 v = *reinterpret_cast<const f32x4*>(vals + row * 132 + c128);
      const f32x4 b = acc.s[q];
      const f32x2 s0 = sigmoid2(f32x2{b[0], b[1]}), s1 = sigmoid2(f32x2{b[2], b[3]});
      f32x4 o = f32x4{v[0] * s0[0], v[1] * s0[1], v[2] * s1[0], v[3] * s1[1]};
      if (pad) o = *reinterpret_cast<const f32x4*>(w.glu_pad + 128 * y + c128);
      *reinterpret_cast<f32x4*>(g + (size_t)(r0 + row) * kD + 128 * y + c128) = o;
    }
  }
}

// S4 [+ the next layer's S1]: depthwise conv -> LN -> swish -> pointwise_conv2 -> pad mask -> +residual -> LN_ff -> FFN ->
// +0.5 residual -> LN_final [-> LN_mac -> FFN_macaron -> ... -> QKV of layer i + 1 on the same LDS-resident rows]
template <int R, int KS, bool NEXT>
__global__ __launch_bounds__(RBT<R>::THREADS) void k_conv_ffn_t(const float* __restrict__ g, const float* __restrict__ x2,
                                                                float* __restrict__ x_out, LayerW w,
                                                                const int64_t* __restrict__ lens, int M, int Tp,
                                                                int n_chunks, int mask_mul, LayerW wn,
                                                                float* __restrict__ x1_next, float* __restrict__ qkv_next,
                                                                int left_ctx, PadSkip ps, VtOut vt_next) {
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
  f32x4 res[T::NQ];
  PPASR_TS(0);
  dwconv_ln_phase_t<R, KS>(g, bufA, w.dw_w, w.dw_b, w.glu_pad, w.ln_cm_g, w.ln_cm_b, w.cm_eps, r0, M, Tp, left_ctx, [&] {
    rbt_prime(ring, seg_pw2);
#pragma unroll
    for (int q = 0; q < T::NQ; ++q)
      res[q] = *reinterpret_cast<const f32x4*>(x2 + (size_t)(r0 + min(L.row(q), valid - 1)) * kD + L.col(q));
  });
  PPASR_TS(1);
  __syncthreads();
  PPASR_TS(2);
  {
    typename T::Acc acc;
    T::zero(acc);
    rbt_gemm<kG256>(bufA, kLda, seg_pw2, w.ff_w1 + (size_t)L.tile() * kTs256, ring, acc);
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
  PPASR_TS(3);
  rbt_layernorm<R>(bufX, bufA, w.ln_ff_g, w.ln_ff_b, 1e-5f);
  __syncthreads();
  PPASR_TS(4);
  typename T::Acc acc2;
  T::zero(acc2);
  ffn_phase_t<R>(bufA, bufH, w.ff_w1, w.ff_b1, w.ff_w2, n_chunks, NEXT ? wn.ffm_w1 + (size_t)L.tile() * kTs256 : nullptr, ring,
                 acc2);
  PPASR_TS(5);
  residual_epilogue_q<R>(bufX, acc2, w.ff_b2, 0.5f);
  __syncthreads();
  PPASR_TS(6);
  rbt_layernorm<R>(bufX, bufX, w.ln_fin_g, w.ln_fin_b, 1e-5f);
  // (LayerNorm, row store and the next LayerNorm use the same wave -> row mapping: no barrier needed between them)
  if (x_out) rbt_store_rows<R>(x_out + (size_t)r0 * kD, bufX, valid);
  PPASR_TS(7);
  if (NEXT) ffn_qkv_body_t<R>(bufX, bufA, bufH, x1_next, qkv_next, wn, r0, valid, n_chunks, ring, vt_next);
  PPASR_TS(15);
}

constexpr size_t kLds16x4 = 4 * 16 * kLda * sizeof(float), kLds16x2 = 2 * 16 * kLda * sizeof(float);
// one workgroup per CU (see lds16 in squeezeformer_kernels.hip)
static size_t excl(size_t own) { return own < kLdsExclusive ? kLdsExclusive : own; }

void launch_ffn_qkv_16(const float* x_in, float* x1, float* qkv, const LayerW& w, int M, int n_chunks, hipStream_t st,
                       const PadSkip& ps) {
  PPASR_LAUNCH(k_ffn_qkv_t<16>, dim3((M + 15) / 16), dim3(kThreads), excl(kLds16x4), st, x_in, x1, qkv, w, M, n_chunks, ps,
               VtOut{});
}
void launch_out_glu_16(const float* ctx, const float* x1, float* x2, float* g, const LayerW& w, const int64_t* lens, int M,
                       int Tp, int mask_mul, hipStream_t st, const PadSkip& ps) {
  PPASR_LAUNCH(k_out_glu_t<16>, dim3((M + 15) / 16), dim3(kThreads), excl(kLds16x2), st, ctx, x1, x2, g, w, lens, M, Tp,
               mask_mul, ps, (float*)nullptr, 0);
}
// split route on 16-row blocks (a streaming chunk): out-projection + LayerNorm (rows -> xhat), then pointwise_conv1 + GLU
// with the columns over two workgroups
void launch_out_glu_split_16(const float* ctx, const float* x1, float* x2, float* g, float* xhat, const LayerW& w,
                             const int64_t* lens, int M, int Tp, int mask_mul, hipStream_t st, const PadSkip& ps, float* hist,
                             int lo) {
  PPASR_LAUNCH(k_out_glu_t<16>, dim3((M + 15) / 16), dim3(kThreads), excl(kLds16x2), st, ctx, x1, x2, g, w, lens, M, Tp,
               mask_mul, ps, xhat, 1);
  PPASR_LAUNCH(k_pw1_glu_cols_t<16>, dim3((M + 15) / 16, 2), dim3(kThreads), excl((16 * kLda + 16 * 132) * sizeof(float)), st,
               xhat, g, w, M, ps, hist, lo, (const float*)nullptr, (const float*)nullptr, (const int64_t*)nullptr, 1, 1);
}
// the two launches on their own, for layers that are not the Conformer's (Squeezeformer's chunk: weight views):
// xhat_out = LN(x1 + ctx Wo + bo) with w.wo / bo / ln_conv_g / ln_conv_b (the plain sum goes to x2_sink)
void launch_oproj_ln_16(const float* ctx, const float* x1, float* x2_sink, float* xhat_out, const LayerW& w, int M,
                        hipStream_t st, const PadSkip& ps) {
  PPASR_LAUNCH(k_out_glu_t<16>, dim3((M + 15) / 16), dim3(kThreads), excl(kLds16x2), st, ctx, x1, x2_sink, (float*)nullptr, w,
               (const int64_t*)nullptr, M, M, 1, ps, xhat_out, 1);
}
// g = GLU(pointwise_conv1(x)) with w.pw1 / pw1_b; hist (M <= 16, one session): moves on by scale * x + bias
void launch_pw1_glu_cols_16(const float* x, float* g, const LayerW& w, int M, hipStream_t st, float* hist, int lo,
                            const float* hist_scale, const float* hist_bias, const PadSkip& ps, const int64_t* lens, int Tp,
                            int mask_mul) {
  PPASR_LAUNCH(k_pw1_glu_cols_t<16>, dim3((M + 15) / 16, 2), dim3(kThreads), excl((16 * kLda + 16 * 132) * sizeof(float)), st, x,
               g, w, M, ps, hist, lo, hist_scale, hist_bias, lens, Tp, mask_mul);
}
bool conv_ffn_16_supported(int ksize, int Tp) { return (ksize == 7 || ksize == 15 || ksize == 31) && Tp >= 2; }
void launch_conv_ffn_16(const float* g, const float* x2, float* x_out, const LayerW& w, const int64_t* lens, int M, int Tp,
                        int n_chunks, int ksize, int mask_mul, const LayerW* next, float* x1_next, float* qkv_next,
                        hipStream_t st, bool causal, const PadSkip& ps) {
  const dim3 grid((M + 15) / 16);
  const int left_ctx = causal ? ksize - 1 : (ksize - 1) / 2;
  const LayerW& wn = next ? *next : w;
#define LAUNCH_CF16(KS)                                                                                               \
  if (next)                                                                                                           \
    PPASR_LAUNCH((k_conv_ffn_t<16, KS, true>), grid, dim3(kThreads), excl(kLds16x4), st, g, x2, x_out, w, lens, M, Tp, \
                 n_chunks, mask_mul, wn, x1_next, qkv_next, left_ctx, ps, VtOut{});                                   \
  else                                                                                                                \
    PPASR_LAUNCH((k_conv_ffn_t<16, KS, false>), grid, dim3(kThreads), excl(kLds16x4), st, g, x2, x_out, w, lens, M, Tp, \
                 n_chunks, mask_mul, wn, x1_next, qkv_next, left_ctx, ps, VtOut{});
  if (ksize == 15) {
    LAUNCH_CF16(15)
  } else if (ksize == 31) {
    LAUNCH_CF16(31)
  } else if (ksize == 7) {
    LAUNCH_CF16(7)
  }
#undef LAUNCH_CF16
}

// ---- 32 rows on 16 waves: drop-in replacements of launch_ffn_qkv / launch_out_glu / launch_conv_ffn for full grids ----
constexpr size_t kLdsW16x4 = 4 * 32 * kLda * sizeof(float), kLdsW16x2 = 2 * 32 * kLda * sizeof(float);
void launch_ffn_qkv_w16(const float* x_in, float* x1, float* qkv, const LayerW& w, int M, int n_chunks, hipStream_t st,
                        const PadSkip& ps, VtOut vt) {
  PPASR_LAUNCH(k_ffn_qkv_t<kW16>, dim3((M + 31) / 32), dim3(1024), kLdsW16x4, st, x_in, x1, qkv, w, M, n_chunks, ps, vt);
}
void launch_out_glu_w16(const float* ctx, const float* x1, float* x2, float* g, const LayerW& w, const int64_t* lens, int M,
                        int Tp, int mask_mul, hipStream_t st, const PadSkip& ps) {
  PPASR_LAUNCH(k_out_glu_t<kW16>, dim3((M + 31) / 32), dim3(1024), excl(kLdsW16x2), st, ctx, x1, x2, g, w, lens, M, Tp,
               mask_mul, ps, (float*)nullptr, 0);
}
bool conv_ffn_w16_supported(int ksize, int Tp) { return (ksize == 7 || ksize == 15 || ksize == 31) && Tp >= 2; }
void launch_conv_ffn_w16(const float* g, const float* x2, float* x_out, const LayerW& w, const int64_t* lens, int M, int Tp,
                         int n_chunks, int ksize, int mask_mul, const LayerW* next, float* x1_next, float* qkv_next,
                         hipStream_t st, bool causal, const PadSkip& ps, VtOut vt_next) {
  const dim3 grid((M + 31) / 32);
  const int left_ctx = causal ? ksize - 1 : (ksize - 1) / 2;
  const LayerW& wn = next ? *next : w;
#define LAUNCH_CFW(KS)                                                                                                 \
  if (next)                                                                                                            \
    PPASR_LAUNCH((k_conv_ffn_t<kW16, KS, true>), grid, dim3(1024), kLdsW16x4, st, g, x2, x_out, w, lens, M, Tp, n_chunks, \
                 mask_mul, wn, x1_next, qkv_next, left_ctx, ps, vt_next);                                              \
  else                                                                                                                 \
    PPASR_LAUNCH((k_conv_ffn_t<kW16, KS, false>), grid, dim3(1024), kLdsW16x4, st, g, x2, x_out, w, lens, M, Tp, n_chunks, \
                 mask_mul, wn, x1_next, qkv_next, left_ctx, ps, vt_next);
  if (ksize == 15) {
    LAUNCH_CFW(15)
  } else if (ksize == 31) {
    LAUNCH_CFW(31)
  } else if (ksize == 7) {
    LAUNCH_CFW(7)
  }
#undef LAUNCH_CFW
}

hipError_t configure_conformer_t_kernels() {
  hipError_t e;
#define SET_LDS(fn)                                                                                                        \
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsExclusive); \
  if (e != hipSuccess) return e;
  SET_LDS(k_ffn_qkv_t<16>);
  SET_LDS(k_out_glu_t<16>);
  SET_LDS(k_pw1_glu_cols_t<16>);
  SET_LDS((k_conv_ffn_t<16, 15, true>));
  SET_LDS((k_conv_ffn_t<16, 15, false>));
  SET_LDS((k_conv_ffn_t<16, 31, true>));
  SET_LDS((k_conv_ffn_t<16, 31, false>));
  SET_LDS((k_conv_ffn_t<16, 7, true>));
  SET_LDS((k_conv_ffn_t<16, 7, false>));
#undef SET_LDS
#define SET_LDS_W(fn, n)                                                                                           \
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(n)); \
  if (e != hipSuccess) return e;
  SET_LDS_W(k_ffn_qkv_t<kW16>, kLdsW16x4);
  SET_LDS_W(k_out_glu_t<kW16>, excl(kLdsW16x2));
  SET_LDS_W((k_conv_ffn_t<kW16, 15, true>), kLdsW16x4);
  SET_LDS_W((k_conv_ffn_t<kW16, 15, false>), kLdsW16x4);
  SET_LDS_W((k_conv_ffn_t<kW16, 31, true>), kLdsW16x4);
  SET_LDS_W((k_conv_ffn_t<kW16, 31, false>), kLdsW16x4);
  SET_LDS_W((k_conv_ffn_t<kW16, 7, true>), kLdsW16x4);
  SET_LDS_W((k_conv_ffn_t<kW16, 7, false>), kLdsW16x4);
#undef SET_LDS_W
  return hipSuccess;
}

}  // namespace ppasr
