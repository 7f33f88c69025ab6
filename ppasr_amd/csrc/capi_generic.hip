// capi_generic.hip -- the GENERAL Conformer layer route: every configuration of ConformerEncoder
// (conformer/encoder.py:24-160) the fused 256-wide row-block kernels are not specialised for --
//   * output_size 512 / 768 / 1024 with heads of 64 (configs/conformer.yml:3-4, "for big data use 512 / 8");
//   * pos_enc_layer_type abs_pos / no_pos (MultiHeadedAttention instead of RelPositionMultiHeadedAttention);
//   * normalize_before = False, concat_after = True, macaron_style = False, use_cnn_module = False;
//   * every activation_type of utils/common.py:189-206;
//   * input_layer = linear (LinearNoSubsampling) besides conv2d / conv2d6 / conv2d8;
//   * any cnn_module_kernel;
// batched (ConformerEncoder.forward) and chunk by chunk (forward_chunk, encoder.py:208-283).
//
// The fused kernels (conformer_kernels.hip) keep a 32 x 256 row block in four LDS buffers per workgroup; a 512-wide row
// block, a post-norm layer or a layer without its macaron half does not fit that scheme.  This route runs the same layer
// (ConformerEncoderLayer.forward, encoder.py:346-431) as a sequence of general pieces:
//   * every Linear / pointwise Conv1D -> k_dense_epi: the fragment-ordered streamed-weight matrix-core GEMM on the SAME
//     packed weights the fused kernels use (ppasr_create packs them for any width), with the activation, the residual
//     add (x += scale * y) and the conv module's pad mask in its epilogue;
//   * attention -> k_attention<64> with any number of heads (AttnArgs::dm = model width); without relative positions
//     the positional half of its score contraction runs on a zero row (q . k + (q + v) . 0: bit-identical to q . k);
//   * LayerNorm, GLU, depthwise conv -> the small row kernels below (HBM-bound, one pass each).
// Results follow the reference exactly like the fused route (same arithmetic, different fusion).
#include <cstdlib>

#include "capi_internal.h"

using namespace ppasr;

namespace {

constexpr int kActNone = -1;

// activation_type (utils/common.py:189-206; Paddle's defaults for every parameter)
__device__ __forceinline__ float act_apply(int act, float v) {
  switch (act) {
    case PPASR_ACT_SWISH: return swishf(v);
    case PPASR_ACT_RELU: return fmaxf(v, 0.f);
    case PPASR_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));  // approximate=False
    case PPASR_ACT_TANH: return tanhf(v);
    case PPASR_ACT_HARDTANH: return fminf(fmaxf(v, -1.0f), 1.0f);
    case PPASR_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.0f);
    case PPASR_ACT_LEAKYRELU: return v >= 0.f ? v : 0.01f * v;
    case PPASR_ACT_SELU: return 1.0507009873554804934193349852946f * (v > 0.f ? v : 1.6732632423543772848170429916717f * expm1f(v));
    case PPASR_ACT_ELU: return v > 0.f ? v : expm1f(v);
    case PPASR_ACT_HARDSWISH: return v * fminf(fmaxf(v + 3.0f, 0.f), 6.0f) * (1.0f / 6.0f);
    case PPASR_ACT_HARDSHRINK: return fabsf(v) > 0.5f ? v : 0.f;
    default: return v;
  }
}

// epilogue of k_dense_epi: v = act((acc + bias) * scale); res == nullptr: out = v; else out = res + rscale * v, with
// the rows of PAD frames (mul * t >= lens[b], rows are [b][t < Tp]) left at res -- the mask behind pointwise_conv2
// (convolution.py:138-140) folded into the residual add
struct GemmEpi {
  int sb = 0;  // 1: acc * scale + bias (Squeezeformer scales the conv features BEFORE input_proj, subsampling.py:66-67)
  int act = kActNone;
  const float* res = nullptr;
  float rscale = 1.f;
  const int64_t* lens = nullptr;
  int Tp = 1, mul = 1;
  PadSkip ps;  // ragged batches (ppasr_set_skip_padding): row tiles behind an utterance's needed frames are not computed
};

// out[M][ldc] (columns < n_valid) = epilogue(A[M][K] Wpacked + bias): the streamed-weight GEMM of k_gemm_stream
// (front_kernels.hip) for dense activations, on 32 * MT-row tiles with K chunks of KC through a double-buffered
// LDS tile.  blockIdx.y = 256-column block, wave = its 32-column tile.  (out may alias res: every element is read and
// written by the same lane.)
template <int MT, int KC>
__global__ __launch_bounds__(kThreads) void k_dense_epi(const float* __restrict__ a, int lda, const f32x4* __restrict__ wp,
                                                        const float* __restrict__ bias, float* out, int M, int n_chunks,
                                                        float scale, int ldc, int n_valid, GemmEpi epi) {
  constexpr int BM = 32 * MT, LD = KC + 4, F4_PER_ROW = KC / 4, NL = BM * F4_PER_ROW / kThreads, G = KC / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  const int r0 = blockIdx.x * BM;
  if (pad_block_skippable(epi.ps, r0, BM, M)) return;
  const int tile_stride = n_chunks * G * 64;
  const f32x4* wbase = wp + (size_t)(blockIdx.y * kWaves + wave) * tile_stride;
  BRing<1> ring;
  ring_prime(ring, wbase, 0);
  const float* tile_base = a + (size_t)r0 * lda;
  const __amdgpu_buffer_rsrc_t rs_a = wstream_rsrc(tile_base);
  int voff[NL], lds_off[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int idx = tid + kThreads * i;
    const int row = idx / F4_PER_ROW, c4 = idx - row * F4_PER_ROW;
    voff[i] = (r0 + row < M) ? row * lda * 4 + 16 * c4 : 0x7fffffff;  // rows >= M read as zeros (offset out of range)
    lds_off[i] = row * LD + 4 * c4;
  }
  f32x4 stg[NL];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NL; ++i) stg[i] = wstream_load(rs_a, voff[i], kc * KC * 4);
  };
  auto write_chunk = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<f32x4*>(buf + lds_off[i]) = stg[i];
  };
  f32x16 acc[MT][1];
  acc_zero(acc);
  load_chunk(0);
  write_chunk(smem);
  __syncthreads();
  for (int kc = 0; kc < n_chunks; ++kc) {
    float* cur = smem + (kc & 1) * BM * LD;
    float* nxt = smem + ((kc + 1) & 1) * BM * LD;
    const bool more = kc + 1 < n_chunks;
    if (more) load_chunk(kc + 1);
    const f32x4* seg = wbase + (size_t)kc * G * 64;
    rb_gemm<MT, 1, G>(cur, LD, seg, 0, more ? seg + G * 64 : nullptr, 0, ring, acc);
    if (more) write_chunk(nxt);
    __syncthreads();
  }
  const int col = blockIdx.y * 256 + wave * 32 + (lane & 31);
  if (col >= n_valid) return;
  const float bv = bias[col];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = r0 + mt * 32 + acc_row(r, lane);
      if (m >= M) continue;
      float v = epi.sb ? acc[mt][0][r] * scale + bv : (acc[mt][0][r] + bv) * scale;
      if (epi.act != kActNone) v = act_apply(epi.act, v);
      if (epi.res) {
        const float rv = epi.res[(size_t)m * ldc + col];
        bool pad = false;
        if (epi.lens) {
          const int bb = m / epi.Tp, t = m - bb * epi.Tp;
          pad = (int64_t)epi.mul * t >= epi.lens[bb];
        }
        v = pad ? rv : rv + epi.rscale * v;
      }
      out[(size_t)m * ldc + col] = v;
    }
}
template <int MT, int KC>
constexpr size_t dense_lds() { return (size_t)2 * (32 * MT) * (KC + 4) * sizeof(float); }

// 32-row tiles with 256-wide K chunks (67 KB of LDS: two workgroups per CU, one's load / store phases behind the other's
// MFMAs).  Taller tiles were measured on the 512-wide model (32 x 10 s, tools/profile_generic.py) and lost: 64 rows x
// KC 128: 11.5 ms of dense time per step against 11.1; 128 rows (135 KB, one workgroup per CU, what conv2's K = 2304 ..
// 4608 contraction runs on) 12.3 ms -- with K = 512 a tile has four chunks and nothing hides its first loads and its
// epilogue.
void dense(const float* a, int lda, const f32x4* w, const float* bias, float* out, int M, int K, int n_cols_padded, int ldc,
           int n_valid, hipStream_t st, float scale = 1.0f, const GemmEpi& epi = GemmEpi{}) {
  PPASR_LAUNCH((k_dense_epi<1, 256>), dim3((M + 31) / 32, n_cols_padded / 256), dim3(kThreads), (dense_lds<1, 256>()), st, a,
               lda, w, bias, out, M, K / 256, scale, ldc, n_valid, epi);
}

// ---- fused feed-forward module for 512-wide models -----------------------------------------------------------------
// out = x + scale * W2 act(W1 LN(x) + b1) + b2 for a 32-row block, the hidden activations never leaving LDS (the 256-wide
// kernels' scheme, phases.h: ffn_phase): LDS = the LayerNorm'd rows [32][516] + two hidden chunks [32][260] = 132.6 KB.
// Wave w owns hidden columns [32w, 32w + 32) of every 256-wide chunk and the output columns [64w, 64w + 64) (two 32-column
// tiles, contracted one after the other so that ONE weight ring streams W1(c), W2(c, tile 0), W2(c, tile 1), W1(c + 1) ..
// without draining).  ln_g == nullptr: no LayerNorm in front (post-norm layers normalise the sum afterwards).
// Replaces k_g_ln + two k_dense_epi launches (the 65 MB hidden tensor of a 32 x 10 s batch stays on chip).
constexpr int kD512 = 512, kLd512 = kD512 + 4;
constexpr size_t kLdsFfn512 = (size_t)(kRows * kLd512 + 2 * kRows * kLda) * sizeof(float);
__global__ __launch_bounds__(kThreads) void k_g_ffn512(const float* x, float* out, const float* __restrict__ ln_g,
                                                       const float* __restrict__ ln_b, const f32x4* __restrict__ w1,
                                                       const float* __restrict__ b1, const f32x4* __restrict__ w2,
                                                       const float* __restrict__ b2, float scale, int act, int M,
                                                       int n_chunks, PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (pad_block_skippable(ps, blockIdx.x * kRows, kRows, M)) return;
  float* bufA = smem;                    // [32][516]
  float* bufH = bufA + kRows * kLd512;   // [2][32][260]
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blockIdx.x * kRows, valid = min(kRows, M - r0);
  BRing<1> ring;
  const int ts1 = (kD512 / 8) * 64;      // W1: K = 512 -> 64 k-groups per 32-column tile
  const int ts2 = n_chunks * 32 * 64;    // W2: K = hidden
  auto w1seg = [&](int c) { return w1 + (size_t)(c * 8 + wave) * ts1; };
  auto w2seg = [&](int c, int t) { return w2 + (size_t)(2 * wave + t) * ts2 + (size_t)c * 32 * 64; };
  ring_prime(ring, w1seg(0), 0);
  {  // rows -> (LayerNorm) -> bufA; a row = two f32x4 per lane (columns 4 lane .. and 256 + 4 lane ..)
    f32x4 g0, g1, be0, be1;
    if (ln_g) {
      g0 = *reinterpret_cast<const f32x4*>(ln_g + 4 * lane);
      g1 = *reinterpret_cast<const f32x4*>(ln_g + 256 + 4 * lane);
      be0 = *reinterpret_cast<const f32x4*>(ln_b + 4 * lane);
      be1 = *reinterpret_cast<const f32x4*>(ln_b + 256 + 4 * lane);
    }
    for (int row = wave; row < kRows; row += kWaves) {
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if (row < valid) {
        v0 = *reinterpret_cast<const f32x4*>(x + (size_t)(r0 + row) * kD512 + 4 * lane);
        v1 = *reinterpret_cast<const f32x4*>(x + (size_t)(r0 + row) * kD512 + 256 + 4 * lane);
        if (ln_g) {
          const float mean = wave_sum(v0[0] + v0[1] + v0[2] + v0[3] + v1[0] + v1[1] + v1[2] + v1[3]) * (1.0f / kD512);
          v0 = v0 - mean;
          v1 = v1 - mean;
          const float var = wave_sum(v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3] + v1[0] * v1[0] +
                                     v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3]) * (1.0f / kD512);
          const float rstd = 1.0f / sqrtf(var + 1e-5f);
          v0 = v0 * rstd * g0 + be0;
          v1 = v1 * rstd * g1 + be1;
        }
      }
      *reinterpret_cast<f32x4*>(bufA + row * kLd512 + 4 * lane) = v0;
      *reinterpret_cast<f32x4*>(bufA + row * kLd512 + 256 + 4 * lane) = v1;
    }
  }
  __syncthreads();
  f32x16 acc2[2][1][1];
  acc_zero(acc2[0]);
  acc_zero(acc2[1]);
  const int hcol = wave * 32 + (lane & 31);
  for (int c = 0; c < n_chunks; ++c) {
    float* hb = bufH + (c & 1) * kRows * kLda;
    f32x16 acc1[1][1];
    acc_zero(acc1);
    rb_gemm<1, 1, kD512 / 8>(bufA, kLd512, w1seg(c), 0, w2seg(c, 0), 0, ring, acc1);
    const float bv = b1[c * 256 + hcol];
#pragma unroll
    for (int r = 0; r < 16; ++r) hb[acc_row(r, lane) * kLda + hcol] = act_apply(act, acc1[0][0][r] + bv);
    __syncthreads();  // (two hidden buffers: the next chunk's epilogue writes the other one, see the comment in ffn_phase)
    rb_gemm<1, 1, kG256>(hb, kLda, w2seg(c, 0), 0, w2seg(c, 1), 0, ring, acc2[0]);
    rb_gemm<1, 1, kG256>(hb, kLda, w2seg(c, 1), 0, c + 1 < n_chunks ? w1seg(c + 1) : nullptr, 0, ring, acc2[1]);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = wave * 64 + t * 32 + (lane & 31);
    const float bv = b2[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, lane);
      if (row < valid) {
        const size_t o = (size_t)(r0 + row) * kD512 + col;
        out[o] = x[o] + scale * (acc2[t][0][0][r] + bv);  // (out may alias x: same lane reads and writes the element)
      }
    }
  }
}

// Row-block projection for 512-wide models: rows -> (LayerNorm | per-channel affine | nothing) -> (pad mask) -> LDS [32][516]
// -> 32-column output tiles, wave w owning n_tiles_per_wave consecutive ones (one weight ring across them).
//   GLU = false: out[m][col] = rows W + bias            (QKV: 1536 columns = 6 tiles per wave)
//   GLU = true : out[m][c]   = (rows W_val + b_val)[c] * sigmoid((rows W_gate + b_gate)[c]), W = pointwise_conv1 with the
//                value channels in tiles [0, 16) and the gate channels in tiles [16, 32) (2 output tiles per wave);
//                a PAD row is all zeros in LDS, so it yields GLU(bias) like the reference's masked input.
// Replaces k_g_ln + k_dense_epi (+ k_g_glu): one launch, the A rows read once, 66 KB of LDS = two workgroups per CU.
// ln_g == nullptr: rows as they are; eps < 0: y = x * g + b (Squeezeformer's ada scale / bias in front of its conv module).
constexpr size_t kLdsProj512 = (size_t)kRows * kLd512 * sizeof(float);
template <bool GLU>
__global__ __launch_bounds__(kThreads) void k_g_proj512(const float* __restrict__ x, float* __restrict__ out, int ldo,
                                                        const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                        float eps, const int64_t* __restrict__ lens, int Tp, int mul,
                                                        const f32x4* __restrict__ w, const float* __restrict__ bias,
                                                        int n_tiles_per_wave, int M, PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (pad_block_skippable(ps, blockIdx.x * kRows, kRows, M)) return;
  float* bufA = smem;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blockIdx.x * kRows, valid = min(kRows, M - r0);
  constexpr int ts = (kD512 / 8) * 64;  // K = 512: 64 k-groups per 32-column tile
  BRing<1> ring;
  const int t0 = GLU ? 2 * wave : wave * n_tiles_per_wave;
  ring_prime(ring, w + (size_t)t0 * ts, 0);
  {
    f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, be0 = {0.f, 0.f, 0.f, 0.f}, be1 = be0;
    if (ln_g) {
      g0 = *reinterpret_cast<const f32x4*>(ln_g + 4 * lane);
      g1 = *reinterpret_cast<const f32x4*>(ln_g + 256 + 4 * lane);
      be0 = *reinterpret_cast<const f32x4*>(ln_b + 4 * lane);
      be1 = *reinterpret_cast<const f32x4*>(ln_b + 256 + 4 * lane);
    }
    for (int row = wave; row < kRows; row += kWaves) {
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
      bool live = row < valid;
      if (live && lens) {
        const int m = r0 + row, bb = m / Tp, t = m - bb * Tp;
        live = (int64_t)mul * t < lens[bb];
      }
      if (live) {
        v0 = *reinterpret_cast<const f32x4*>(x + (size_t)(r0 + row) * kD512 + 4 * lane);
        v1 = *reinterpret_cast<const f32x4*>(x + (size_t)(r0 + row) * kD512 + 256 + 4 * lane);
        if (ln_g && eps >= 0.f) {
          const float mean = wave_sum(v0[0] + v0[1] + v0[2] + v0[3] + v1[0] + v1[1] + v1[2] + v1[3]) * (1.0f / kD512);
          v0 = v0 - mean;
          v1 = v1 - mean;
          const float var = wave_sum(v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3] + v1[0] * v1[0] +
                                     v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3]) * (1.0f / kD512);
          const float rstd = 1.0f / sqrtf(var + eps);
          v0 = v0 * rstd * g0 + be0;
          v1 = v1 * rstd * g1 + be1;
        } else if (ln_g) {
          v0 = v0 * g0 + be0;
          v1 = v1 * g1 + be1;
        }
      }
      *reinterpret_cast<f32x4*>(bufA + row * kLd512 + 4 * lane) = v0;
      *reinterpret_cast<f32x4*>(bufA + row * kLd512 + 256 + 4 * lane) = v1;
    }
  }
  __syncthreads();
  if (GLU) {
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
      const int jt = 2 * wave + t;  // output tile: value channels tile jt, gate channels tile 16 + jt
      f32x16 av[1][1], ag[1][1];
      acc_zero(av);
      acc_zero(ag);
      const f32x4* sv = w + (size_t)jt * ts;
      const f32x4* sg = w + (size_t)(16 + jt) * ts;
      rb_gemm<1, 1, kD512 / 8>(bufA, kLd512, sv, 0, sg, 0, ring, av);
      rb_gemm<1, 1, kD512 / 8>(bufA, kLd512, sg, 0, t == 0 ? w + (size_t)(jt + 1) * ts : nullptr, 0, ring, ag);
      const int col = jt * 32 + (lane & 31);
      const float bv = bias[col], bg = bias[kD512 + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, lane);
        if (row < valid) out[(size_t)(r0 + row) * ldo + col] = (av[0][0][r] + bv) * sigmoidf(ag[0][0][r] + bg);
      }
    }
  } else {
#pragma unroll 1
    for (int t = 0; t < n_tiles_per_wave; ++t) {
      const int nt = t0 + t;
      f32x16 acc[1][1];
      acc_zero(acc);
      rb_gemm<1, 1, kD512 / 8>(bufA, kLd512, w + (size_t)nt * ts, 0, t + 1 < n_tiles_per_wave ? w + (size_t)(nt + 1) * ts : nullptr, 0,
                               ring, acc);
      const int col = nt * 32 + (lane & 31);
      const float bv = bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, lane);
        if (row < valid) out[(size_t)(r0 + row) * ldo + col] = acc[0][0][r] + bv;
      }
    }
  }
}

// LayerNorm over D columns (nn.LayerNorm, biased variance, eps inside the sqrt), one wave per row, optionally followed
// by an activation, optionally with rows t of utterance b zeroed where mul * t >= lens[b] (the conv module's input
// mask, convolution.py:104-106).  eps < 0: per-channel affine only (folded BatchNorm, see capi.hip).  g == nullptr:
// identity (mask / activation only).  The result is multiplied by post_scale (1 everywhere but behind
// LinearNoSubsampling, where the positional encoding's x * sqrt(d) follows the ReLU).  (x and out may be the same buffer, so neither is __restrict__)
__global__ __launch_bounds__(256) void k_g_ln(const float* x, float* out, const float* __restrict__ g,
                                              const float* __restrict__ b, int M, int D, float eps, int act,
                                              const int64_t* __restrict__ lens, int Tp, int mul, float post_scale, PadSkip ps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  if (pad_block_skippable(ps, row, 1, M)) return;  // ragged batches: a row behind its utterance's needed frames
  const float* xr = x + (size_t)row * D;
  float* o = out + (size_t)row * D;
  if (lens) {
    const int bb = row / Tp, t = row - bb * Tp;
    if ((int64_t)mul * t >= lens[bb]) {
      for (int c = lane; c < D; c += 64) o[c] = 0.f;
      return;
    }
  }
  float mean = 0.f, rstd = 1.f;
  if (g && eps >= 0.f) {
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    mean = wave_sum(s) / D;
    float v = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float d = xr[c] - mean;
      v += d * d;
    }
    rstd = 1.0f / sqrtf(wave_sum(v) / D + eps);
  }
  for (int c = lane; c < D; c += 64) {
    float y = g ? (xr[c] - mean) * rstd * g[c] + b[c] : xr[c];
    if (act != kActNone) y = act_apply(act, y);
    o[c] = y * post_scale;
  }
}

// GLU over the channel halves of pointwise_conv1's output: g[m][c] = pg[m][c] * sigmoid(pg[m][D + c])  (convolution.py:126)
__global__ void k_g_glu(const float* __restrict__ pg, float* __restrict__ g, int M, int D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * D) return;
  const size_t row = i / D, c = i - row * D;
  g[i] = pg[row * 2 * D + c] * sigmoidf(pg[row * 2 * D + D + c]);
}

// Depthwise conv over time (KS taps, `left` frames of left context: KS-1 causal, (KS-1)/2 non-causal).  Output row
// (b, t < Tp) reads the input rows (b, t + row_off - left + j) of utterances that are Tp + row_off rows long: row_off = 0
// batched; streaming (one utterance) the input is [lo cached rows | chunk rows] and row_off = left = lo.  Taps outside
// the utterance read pad[c] = GLU(pointwise_conv1 bias) in the causal module (the reference zero-pads BEFORE
// pointwise_conv1, convolution.py:108-126) and 0 in the non-causal one (its depthwise conv pads its own input).
// One workgroup = (tile of TT <= 32 output frames, utterance, 256-channel slab): the input rows it needs and the taps go
// through LDS once (the one-thread-per-output form re-read every input row KS times through L1 / L2: 47 us per layer of a
// 32 x 10 s batch at width 512, 6 x the time of its HBM traffic); thread = channel, fmaf chain in tap order as before.
__global__ __launch_bounds__(256) void k_g_dwconv(const float* __restrict__ g, float* __restrict__ out,
                                                  const float* __restrict__ w /*[KS][D]*/, const float* __restrict__ bias,
                                                  const float* __restrict__ pad, int Tp, int D, int KS, int left, int row_off,
                                                  int stride, int Tp_in, int TT, PadSkip ps) {
  // stride 2: the Efficient-Conformer's stride layer (efficient_conformer/convolution.py:54-60): output frame t of the Tp =
  // ceil(Tp_in / 2) reads the input frames 2 t - left .. ; stride 1: Tp_in == Tp
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int c = blockIdx.z * 256 + threadIdx.x, bb = blockIdx.y, t0 = blockIdx.x * TT;
  if (ps.lens && t0 >= pad_need_steps(ps, bb)) return;  // ragged batches: a tile of OUTPUT frames nobody needs (ps.Tp == Tp)
  const int n_out = min(TT, Tp - t0), Tin = Tp_in + row_off;
  const int n_in = stride * (n_out - 1) + KS;       // input rows of this tile
  const int tt0 = stride * t0 + row_off - left;     // first of them
  float* xs = smem;                                  // [n_in][256]
  float* ws = smem + (size_t)(stride * (TT - 1) + KS) * 256;  // [KS][256]
  const bool causal = left == KS - 1;
  const float pv = causal ? pad[c] : 0.f;
  for (int i = 0; i < n_in; ++i) {
    const int tt = tt0 + i;
    xs[i * 256 + threadIdx.x] = (tt >= 0 && tt < Tin) ? g[((size_t)bb * Tin + tt) * D + c] : pv;
  }
  for (int j = 0; j < KS; ++j) ws[j * 256 + threadIdx.x] = w[(size_t)j * D + c];
  // (every thread reads back only what it wrote -- its own channel column -- so no barrier is needed)
  const float bv = bias[c];
  for (int o = 0; o < n_out; ++o) {
    float acc = bv;
    const float* xr = xs + (size_t)(stride * o) * 256 + threadIdx.x;
    for (int j = 0; j < KS; ++j) acc = fmaf(ws[j * 256 + threadIdx.x], xr[j * 256], acc);
    out[((size_t)bb * Tp + t0 + o) * D + c] = acc;
  }
}
constexpr int kDwMaxKs = 63;                 // cnn_module_kernel of the general route (ppasr_create checks it)
constexpr size_t kDwLdsRows = 150;            // LDS budget in 1 KiB rows: input rows + taps
// B * Tp output rows; tile = up to 32 output frames, fewer when the kernel is long (stride * (TT - 1) + 2 KS rows of LDS)
inline void launch_dwconv(const float* g, float* out, const float* w, const float* bias, const float* pad, int B, int Tp, int D,
                          int KS, int left, int row_off, int stride, int Tp_in, hipStream_t st, const PadSkip& ps = PadSkip{}) {
  const int TT = std::max(1, std::min(32, (int)(kDwLdsRows - 2 * KS) / stride + 1));
  const size_t lds = (size_t)(stride * (TT - 1) + 2 * KS) * 256 * sizeof(float);
  PPASR_LAUNCH(k_g_dwconv, dim3((Tp + TT - 1) / TT, B, D / 256), dim3(256), lds, st, g, out, w, bias, pad, Tp, D, KS, left,
               row_off, stride, Tp_in, TT, ps);
}

// abs_pos (PositionalEncoding.forward, embedding.py:70: x * xscale + pe[offset : offset + T]; the scale is the embed
// GEMM's): x[b][t] += pe[pos0 + t]
__global__ void k_g_add_pe(float* __restrict__ x, const float* __restrict__ pe, int M, int D, int Tp, int pos0) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * D) return;
  const int row = (int)(i / D), c = (int)(i - (size_t)row * D);
  x[i] += pe[(size_t)(pos0 + row % Tp) * D + c];
}

// input_layer = linear: GlobalCMVN (utils/cmvn.py:29-31) of the feature rows, zero-padded to the GEMM's K chunk
__global__ void k_g_cmvn_pad(const float* __restrict__ feats, const float* __restrict__ mean, const float* __restrict__ istd,
                             float* __restrict__ out, int M, int F, int Kp) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * Kp) return;
  const int row = (int)(i / Kp), k = (int)(i - (size_t)row * Kp);
  out[i] = k < F ? (feats[(size_t)row * F + k] - mean[k]) * istd[k] : 0.f;
}

// stride layer, residual branch: AvgPool1D(kernel 2, stride 2, ceil_mode, exclusive) over time
// (efficient_conformer/encoder.py:431-436, 521-527): row (b, j) <- mean of frames 2j, 2j + 1 (the last one alone if T is odd)
__global__ void k_g_avgpool2(const float* __restrict__ x, float* __restrict__ out, int B, int Tp, int Ts, int D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * Ts * D) return;
  const int row = (int)(i / D), c = (int)(i - (size_t)row * D);
  const int b = row / Ts, j = row - b * Ts;
  const float v0 = x[((size_t)b * Tp + 2 * j) * D + c];
  out[i] = 2 * j + 1 < Tp ? (v0 + x[((size_t)b * Tp + 2 * j + 1) * D + c]) * 0.5f : v0;
}

// concat_after: out[m] = [a[m] | b[m]]  (encoder.py:395-396)
__global__ void k_g_concat(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int M, int D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * 2 * D) return;
  const size_t row = i / (2 * D), c = i - row * 2 * D;
  out[i] = c < (size_t)D ? a[row * D + c] : b[row * D + c - D];
}

// streaming: this chunk's keys / values (columns D.. / 2D.. of qkv) -> cache rows
__global__ void k_g_kv_append(const float* __restrict__ qkv, float* __restrict__ kc, float* __restrict__ vc, int n, int D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n * D) return;
  const size_t row = i / D, c = i - row * D;
  kc[i] = qkv[row * 3 * D + D + c];
  vc[i] = qkv[row * 3 * D + 2 * D + c];
}

inline bool fused_ffn512() { return true; }  // width 512: the fused feed-forward kernels (other widths: LayerNorm + two GEMM launches)
inline size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }
inline dim3 blocks(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

struct GenWs {
  size_t y1, y2, x, a, big, y, g, ctx, cat, lg, xs, xr, total;
};
GenWs gen_layout(const ppasr_model_s* m, int B, int T) {
  const int D = m->desc.output_size, H = m->desc.linear_units, V = m->desc.vocab_size;
  const auto fd = m->front_dims(T);
  const size_t M = (size_t)B * fd.Tp;
  const size_t lo = m->desc.cnn_module_kernel > 0 ? m->desc.cnn_module_kernel - 1 : 0;  // streaming: cached conv rows in front
  const size_t wide = (size_t)std::max(3 * D, H);
  GenWs w{};
  size_t o = 0;
  if (m->desc.input_layer == 1) {
    w.y1 = o; o += al64(M * m->lin_kpad);
    w.y2 = o;
  } else {
    w.y1 = o; o += al64((size_t)B * fd.T1 * m->F1 * D);  // (conv2d8: the third conv's output reuses it)
    w.y2 = o; o += al64((size_t)B * (fd.T2 ? fd.T2 : fd.Tp) * m->F2 * D);
  }
  w.x = o; o += al64(M * D);
  w.a = o; o += al64((M + lo) * D);
  w.big = o; o += al64((M + lo) * wide);  // FFN hidden / qkv / pointwise_conv1 output
  w.y = o; o += al64(M * D);
  w.g = o; o += al64((M + lo) * D);
  w.ctx = o; o += al64(M * D);
  w.cat = o; o += m->gen.concat_after ? al64(M * 2 * D) : 0;
  w.lg = o; o += al64(M * (size_t)V);  // logits / probabilities when the caller does not ask for them
  const bool sq = m->desc.model_type == PPASR_MODEL_SQUEEZEFORMER;
  w.xs = o; o += sq ? al64(M * D) : 0;  // Squeezeformer: full-rate activations saved at the time reduction
  w.xr = o; o += sq ? al64(M * D) : 0;  //   and a second activation buffer (reduced / recovered rows)
  w.total = o;
  return w;
}

// one call = the encoder layers + head on M = B * Tp rows that the front end left in x
struct GenRun {
  ppasr_model_s* h;
  hipStream_t st;
  int B, Tp;
  const int64_t* lens;  // nullptr: no masks (forward_chunk)
  float *x, *a, *big, *y, *g, *ctx, *cat, *lg;
  ppasr_stream_s* s = nullptr;  // streaming: caches hold s->cache_t frames, key 0 at positional row pos0
  int pos0 = 0;
  int used_r = 0;               // cached frames of the half-rate layers that take part (Efficient-Conformer, ChunkPlan)
  int* frames_out = nullptr;    // encoder frames this call produced (half of the chunk's behind a stride layer)
};

ppasr_status gen_layers(const GenRun& r, float* probs, float* logits, int32_t* frame_argmax, float* frame_maxprob) {
  ppasr_model_s* h = r.h;
  hipStream_t st = r.st;
  const auto& o = h->gen;
  const int D = h->desc.output_size, H = h->desc.linear_units, V = h->desc.vocab_size, heads = h->desc.attention_heads;
  const int B = r.B;
  // frames per utterance / rows / mask multiplier / positional stride of the CURRENT layer: the Efficient-Conformer's
  // stride layer halves the rate (masks[:, :, ::2], pos_emb[:, ::2], efficient_conformer/encoder.py:252-257)
  int Tp = r.Tp, M = B * Tp, mul = h->sub_rate(), pstride = 1;
  const bool eff = h->desc.model_type == PPASR_MODEL_EFFICIENT_CONFORMER;
  const int64_t* lens = r.lens;
  float *x = r.x, *a = r.a, *big = r.big, *y = r.y, *g = r.g, *ctx = r.ctx;
  const bool rel = o.pos == PPASR_OPT_POS_REL;
  bool half = false;  // behind the stride layer
  // ragged batches (ppasr_set_skip_padding; batched calls only): per utterance only the rows its valid output frames depend
  // on are computed, with the slack of the fused route (capi.hip ppasr_encode): the right context of a non-causal conv
  // module, and with a rate change the stride layer's 2j / 2j + 1 rows and the 3-frame groups of grouped attention.  Rows
  // behind them keep whatever the buffers hold; the valid rows never read them (PAD frames enter the conv module through
  // its mask, keys and values through the attention's) and the outputs behind the valid frames are zeroed at the end.
  const bool skip = h->skip_padding && lens && !r.s;
  const int rc = (h->desc.causal || !o.use_cnn) ? 0 : (h->desc.cnn_module_kernel - 1) / 2;
  const int slack_half = rc + 4, slack_full = eff ? 2 * slack_half + rc + 8 : rc + 4, mul0 = mul;
  auto pskip = [&](int Tcur, int mul_cur) {
    PadSkip p;
    if (skip) {
      p.lens = lens;
      p.Tp = Tcur;
      p.mul = mul_cur;
      p.slack = mul_cur == mul0 ? slack_full : slack_half;
    }
    return p;
  };
  PadSkip ps = pskip(Tp, mul);
  auto ln = [&](const float* in, float* out, const float* gg, const float* bb, float eps, int act, bool mask, int rows) {
    PPASR_LAUNCH(k_g_ln, dim3((rows + 3) / 4), dim3(256), 0, st, in, out, gg, bb, rows, D, eps, act, mask ? lens : nullptr, Tp, mul, 1.0f,
                 rows == M ? ps : PadSkip{});
  };
  auto plain_epi = [&]() {
    GemmEpi e;
    e.ps = ps;
    return e;
  };
  auto act_epi = [&](int act) {
    GemmEpi e;
    e.act = act;
    e.ps = ps;
    return e;
  };
  auto res_epi = [&](float scale, bool mask) {
    GemmEpi e;
    e.res = x;
    e.rscale = scale;
    e.ps = ps;
    if (mask && lens) {
      e.lens = lens;
      e.Tp = Tp;
      e.mul = mul;
    }
    return e;
  };
  // PositionwiseFeedForward (positionwise.py:32-39) inside the layer's residual (encoder.py:380-386 / 411-417)
  const bool fused_ffn = fused_ffn512();
  auto ffn = [&](const float* lg, const float* lb, const f32x4* w1, const float* b1, const f32x4* w2, const float* b2, float scale) {
    if (fused_ffn && D == kD512) {  // one launch, hidden activations in LDS
      PPASR_LAUNCH(k_g_ffn512, dim3((M + kRows - 1) / kRows), dim3(kThreads), kLdsFfn512, st, x, x, o.post_norm ? nullptr : lg,
                   o.post_norm ? nullptr : lb, w1, b1, w2, b2, scale, o.act, M, H / 256, ps);
      if (o.post_norm) ln(x, x, lg, lb, 1e-5f, kActNone, false, M);
      return;
    }
    const float* in = x;
    if (!o.post_norm) {
      ln(x, a, lg, lb, 1e-5f, kActNone, false, M);
      in = a;
    }
    dense(in, D, w1, b1, big, M, D, H, H, H, st, 1.0f, act_epi(o.act));
    dense(big, H, w2, b2, x, M, H, D, D, D, st, 1.0f, res_epi(scale, false));
    if (o.post_norm) ln(x, x, lg, lb, 1e-5f, kActNone, false, M);
  };
  const float ff_scale = o.macaron ? 0.5f : 1.0f;
  for (int i = 0; i < h->desc.num_blocks; ++i) {
    const LayerW& L = h->layers[i];
    // streaming: cached conv-input rows in front of the chunk (this layer's kernel - 1: the slot's first rows, like the
    // fused route) and cached key / value frames (the half-rate layers hold each cached frame once, capi_stream.hip)
    const int lo_s = (r.s && o.use_cnn) ? h->layer_ks[i] - 1 : 0;
    const int n_cache = r.s ? (half ? r.used_r : r.s->cache_t) : 0;
    if (o.macaron) ffn(L.ln_mac_g, L.ln_mac_b, L.ffm_w1, L.ffm_b1, L.ffm_w2, L.ffm_b2, ff_scale);
    // ---- (Rel)MultiHeadedAttention (attention.py:123-262) ----
    {
      const float* in = x;
      if (fused_ffn && D == kD512 && !o.concat_after) {  // LayerNorm + QKV in one launch
        PPASR_LAUNCH(k_g_proj512<false>, dim3((M + kRows - 1) / kRows), dim3(kThreads), kLdsProj512, st, x, big, 3 * D,
                     o.post_norm ? nullptr : L.ln_mha_g, o.post_norm ? nullptr : L.ln_mha_b, 1e-5f, (const int64_t*)nullptr, Tp, mul,
                     L.wqkv, L.bqkv, 3 * D / 32 / kWaves, M, ps);
      } else {
        if (!o.post_norm) {
          ln(x, a, L.ln_mha_g, L.ln_mha_b, 1e-5f, kActNone, false, M);
          in = a;
        }
        dense(in, D, L.wqkv, L.bqkv, big, M, D, 3 * D, 3 * D, 3 * D, st, 1.0f, plain_epi());
      }
      // tokens: frames, or zero-padded groups of 3 (GroupedRelPositionMultiHeadedAttention, pad4group)
      const int grp = h->layer_group[i], Tt = (Tp + grp - 1) / grp;
      AttnArgs at{big, 3 * D, big + D, 3 * D, big + 2 * D, 3 * D, Tt, Tt, rel ? r.pos0 : 0, lens, ctx, L.pos_u, L.pos_v,
                  L.ptab, rel ? pstride : 0, mul * grp, Tp, Tp, grp};
      if (r.s) {  // keys / values: [cache | chunk] in the layer's device caches
        float* kc = r.s->kc + ((size_t)i * r.s->cap + n_cache) * D;
        float* vc = r.s->vc + ((size_t)i * r.s->cap + n_cache) * D;
        PPASR_LAUNCH(k_g_kv_append, blocks((size_t)M * D), dim3(256), 0, st, big, kc, vc, M, D);
        at.k = r.s->kc + (size_t)i * r.s->cap * D;
        at.v = r.s->vc + (size_t)i * r.s->cap * D;
        at.k_stride = at.v_stride = D;
        // grouped attention re-cuts cache + chunk frames into groups of 3 from the START of the cache (pad4group on the
        // concatenated keys, efficient_conformer/attention.py:160-175)
        at.kv_frames = n_cache + Tp;
        at.T2 = (n_cache + Tp + grp - 1) / grp;
      }
      at.pad_skip = skip ? ps.slack + 1 : 0;
      at.dm = D;
      launch_attention(at, B, heads, st);
      if (o.concat_after) {  // x + concat_linear([attention input | linear_out(ctx)])
        dense(ctx, D, L.wo, L.bo, y, M, D, D, D, D, st, 1.0f, plain_epi());
        PPASR_LAUNCH(k_g_concat, blocks((size_t)M * 2 * D), dim3(256), 0, st, in, y, r.cat, M, D);
        dense(r.cat, 2 * D, h->gen_x[i].wcat, h->gen_x[i].bcat, x, M, 2 * D, D, D, D, st, 1.0f, res_epi(1.0f, false));
      } else {
        dense(ctx, D, L.wo, L.bo, x, M, D, D, D, D, st, 1.0f, res_epi(1.0f, false));
      }
      if (o.post_norm) ln(x, x, L.ln_mha_g, L.ln_mha_b, 1e-5f, kActNone, false, M);
    }
    // ---- ConvolutionModule (convolution.py:82-143) ----
    if (o.use_cnn) {
      const int KS = h->layer_ks[i];
      const int left = h->desc.causal ? KS - 1 : (KS - 1) / 2;
      const bool stride2 = eff && ((eff_stride_mask(h->desc) >> i) & 1u);
      float* a_new = a + (size_t)lo_s * D;
      const int rows = lo_s + M;
      if (fused_ffn && D == kD512 && lo_s == 0) {  // (LayerNorm) + pad mask + pointwise_conv1 + GLU in one launch
        PPASR_LAUNCH(k_g_proj512<true>, dim3((M + kRows - 1) / kRows), dim3(kThreads), kLdsProj512, st, x, g, D,
                     o.post_norm ? nullptr : L.ln_conv_g, o.post_norm ? nullptr : L.ln_conv_b, 1e-5f, lens, Tp, mul, L.pw1,
                     L.pw1_b, 2, M, ps);
      } else {
        if (!o.post_norm) ln(x, a_new, L.ln_conv_g, L.ln_conv_b, 1e-5f, kActNone, true, M);  // LN_conv, PAD frames -> 0
        else ln(x, a_new, nullptr, nullptr, 0.f, kActNone, true, M);                           // PAD frames -> 0 only
        if (lo_s) {
          float* hist = r.s->xh_hist + (size_t)i * r.s->lo * D;
          HIP_TRY(hipMemcpyAsync(a, hist, (size_t)lo_s * D * sizeof(float), hipMemcpyDeviceToDevice, st));
          // new cache = the last lo rows of [cache | chunk] (convolution.py:110-116)
          HIP_TRY(hipMemcpyAsync(hist, a + (size_t)M * D, (size_t)lo_s * D * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        dense(a, D, L.pw1, L.pw1_b, big, rows, D, 2 * D, 2 * D, 2 * D, st, 1.0f, lo_s ? GemmEpi{} : plain_epi());
        PPASR_LAUNCH(k_g_glu, blocks((size_t)rows * D), dim3(256), 0, st, big, g, rows, D);
      }
      if (stride2) {
        // StrideConformerEncoderLayer (efficient_conformer/encoder.py:455-548): depthwise conv with stride 2, the residual
        // through AvgPool1D(2, ceil_mode); everything behind runs on ceil(T / 2) frames with masks / positions [::2]
        const int Ts = (Tp + 1) / 2, Ms = B * Ts;
        launch_dwconv(g, y, L.dw_w, L.dw_b, L.glu_pad, B, Ts, D, KS, left, lo_s, 2, Tp, st, pskip(Ts, mul * 2));
        half = true;
        PPASR_LAUNCH(k_g_avgpool2, blocks((size_t)Ms * D), dim3(256), 0, st, x, ctx, B, Tp, Ts, D);
        Tp = Ts;
        M = Ms;
        mul *= 2;
        pstride *= 2;
        ps = pskip(Tp, mul);
        std::swap(x, ctx);  // (res_epi below reads the new x = the pooled residual)
      } else {
        launch_dwconv(g, y, L.dw_w, L.dw_b, L.glu_pad, B, Tp, D, KS, left, lo_s, 1, Tp, st, ps);
      }
      ln(y, y, L.ln_cm_g, L.ln_cm_b, L.cm_eps, o.act, false, M);  // LayerNorm / folded BatchNorm + activation
      dense(y, D, L.pw2, L.pw2_b, x, M, D, D, D, D, st, 1.0f, res_epi(1.0f, true));  // PAD frames of the conv output -> 0, + residual
      if (o.post_norm) ln(x, x, L.ln_conv_g, L.ln_conv_b, 1e-5f, kActNone, false, M);
    }
    ffn(L.ln_ff_g, L.ln_ff_b, L.ff_w1, L.ff_b1, L.ff_w2, L.ff_b2, ff_scale);
    if (o.use_cnn) ln(x, x, L.ln_fin_g, L.ln_fin_b, 1e-5f, kActNone, false, M);
  }
  if (r.frames_out) *r.frames_out = Tp;
  // ---- after_norm (normalize_before only, encoder.py:201) -> ctc_lo -> softmax (ctc.py:62-70) ----
  const float* enc = x;
  if (!o.post_norm) {
    ln(x, a, h->head.ln_g, h->head.ln_b, 1e-5f, kActNone, false, M);
    enc = a;
  }
  float* lg = logits ? logits : (probs ? probs : r.lg);
  dense(enc, D, h->gen_head_w, h->gen_head_b, lg, M, D, h->gen_vpad, V, V, st, 1.0f, plain_epi());
  float* pr = probs;
  if (!pr && (frame_argmax || frame_maxprob)) pr = (lg == r.lg) ? lg : r.lg;
  if (pr) {
    if (pr != lg) HIP_TRY(hipMemcpyAsync(pr, lg, (size_t)M * V * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_softmax_from_stats(pr, nullptr, nullptr, M, V, st);
    if (frame_argmax || frame_maxprob) {
      int32_t* fa = frame_argmax ? frame_argmax : reinterpret_cast<int32_t*>(y);
      float* fp = frame_maxprob ? frame_maxprob : y + M;
      launch_frame_argmax(pr, fa, fp, M, V, st);
    }
  }
  if (skip) launch_zero_pad_rows(probs, logits, frame_argmax, frame_maxprob, lens, B, Tp, mul, V, st);
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

// front end: GlobalCMVN + the subsampling class + the positional encoding's scaling (subsampling.py, embedding.py)
// -> x [B * Tp][D]; pe_off = position of the first output frame (abs_pos)
ppasr_status gen_front(ppasr_model_s* h, const float* feats, int B, int T, float* ws, const GenWs& wl, int pe_off,
                       hipStream_t st) {
  const int D = h->desc.output_size, F = h->desc.input_dim;
  const auto fd = h->front_dims(T);
  const int Tp = fd.Tp, M = B * Tp, il = h->desc.input_layer;
  float *y1 = ws + wl.y1, *y2 = ws + wl.y2, *x = ws + wl.x;
  // x * sqrt(d) belongs to PositionalEncoding / RelPositionalEncoding; NoPositionalEncoding returns x as it is
  const float xscale = h->gen.pos == PPASR_OPT_POS_NONE ? 1.0f : sqrtf((float)D);
  if (il == 1) {
    // LinearNoSubsampling (subsampling.py:39-42): Linear -> LayerNorm(eps 1e-12) -> ReLU
    const int Kp = h->lin_kpad;
    PPASR_LAUNCH(k_g_cmvn_pad, blocks((size_t)M * Kp), dim3(256), 0, st, feats, h->front.cmvn_mean, h->front.cmvn_istd, y1, M, F, Kp);
    dense(y1, Kp, h->front.embed_w, h->front.embed_b, x, M, Kp, D, D, D, st);
    PPASR_LAUNCH(k_g_ln, dim3((M + 3) / 4), dim3(256), 0, st, x, x, h->lin_ln_g, h->lin_ln_b, M, D, 1e-12f, (int)PPASR_ACT_RELU,
                 (const int64_t*)nullptr, Tp, 1, xscale, PadSkip{});
  } else {
    launch_conv1(feats, h->front, y1, B, T, F, fd.T1, h->F1, st, PadSkip{}, D);
    if (il == 8) {  // Conv2dSubsampling8: three 3x3 / 2 convs, the third one over conv1's output buffer
      launch_conv_stage(y1, h->front.conv2_w, h->front.conv2_b, y2, B, fd.T1, h->F1, fd.T2, h->F2, 3, 2, st, PadSkip{}, D);
      launch_conv_stage(y2, h->front.conv3_w, h->front.conv3_b, y1, B, fd.T2, h->F2, Tp, h->F3, 3, 2, st, PadSkip{}, D);
      dense(y1, h->F3 * D, h->front.embed_w, h->front.embed_b, x, M, h->F3 * D, D, D, D, st, xscale);
    } else {  // conv2d (3x3 / 2) or conv2d6 (5x5 / 3): FrontW::conv2_k / _s
      launch_conv_stage(y1, h->front.conv2_w, h->front.conv2_b, y2, B, fd.T1, h->F1, Tp, h->F2, h->front.conv2_k,
                        h->front.conv2_s, st, PadSkip{}, D);
      dense(y2, h->F2 * D, h->front.embed_w, h->front.embed_b, x, M, h->F2 * D, D, D, D, st, xscale);
    }
  }
  if (h->gen.pos == PPASR_OPT_POS_ABS)
    PPASR_LAUNCH(k_g_add_pe, blocks((size_t)M * D), dim3(256), 0, st, x, h->pe_dev, M, D, Tp, pe_off);
  return PPASR_OK;
}

// ---- Squeezeformer pieces ----
// time reduction, depthwise part (TimeReductionLayerStream / TimeReductionLayer1D, time_reduction.py:183-206 / 80-131): zero
// the PAD frames, Conv1D(ks, stride 2, padding max(0, ks - 2)) per channel: reduced row (b, j) <- frames 2j - pad + k
__global__ void k_g_sq_reduce_dw(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ dw_w,
                                 const float* __restrict__ dw_b, int ks, const int64_t* __restrict__ lens, int B, int Tp, int Tr,
                                 int D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * Tr * D) return;
  const int row = (int)(i / D), c = (int)(i - (size_t)row * D);
  const int b = row / Tr, j = row - b * Tr;
  const int pad = ks > 2 ? ks - 2 : 0;
  float v = dw_b[c];
  for (int k = 0; k < ks; ++k) {
    const int t = 2 * j - pad + k;
    if (t < 0 || t >= Tp) continue;                     // the conv's own zero padding
    if (lens && 4 * (int64_t)t >= lens[b]) continue;    // masked_fill(xs, mask_pad == 0, 0)
    v = fmaf(x[((size_t)b * Tp + t) * D + c], dw_w[(size_t)k * D + c], v);
  }
  out[i] = v;
}
// time recovery, gather part (encoder.py:224): repeat_interleave(xs, 2)[:, :T'] -> row (b, t) <- reduced row (b, t >> 1)
__global__ void k_g_sq_repeat(const float* __restrict__ xr, float* __restrict__ out, int B, int Tp, int Tr, int D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * Tp * D) return;
  const int row = (int)(i / D), c = (int)(i - (size_t)row * D);
  const int b = row / Tp, t = row - b * Tp;
  out[i] = xr[((size_t)b * Tr + (t >> 1)) * D + c];
}

}  // namespace

// SqueezeformerEncoder.forward / forward_chunk (squeezeformer/encoder.py:172-236, 260-381; layer :435-506) + ctc softmax as
// general pieces: the post-norm block MHA -> FFN -> conv -> FFN with the adaptive scale / bias of every module folded into
// its first projection at pack time (capi_squeezeformer.hip), time reduction before layer reduce_idx (keys / positions of
// the reduced layers: mask 8 t < len, every second positional row) and recovery before layer recover_idx.
// s != nullptr: one chunk of one stream (B = 1, no masks): keys / values in the stream's caches (the half-rate layers hold
// each cached frame once, capi_stream.hip), the conv modules read [cached scaled inputs | chunk] (convolution.py:119-137).
namespace {
ppasr_status sq_run(ppasr_model_s* h, const float* feats, const int64_t* lens, int B, int T, float* probs, float* logits,
                    int32_t* frame_argmax, float* frame_maxprob, float* ws, hipStream_t st, ppasr_stream_s* s,
                    const ChunkPlan* plan) {
  if (h->taps) return fail(PPASR_EUNSUPPORTED, "debug taps are built for the fused 256-wide route");
  const GenWs wl = gen_layout(h, B, T);
  const int D = h->desc.output_size, H = h->desc.linear_units, V = h->desc.vocab_size, heads = h->desc.attention_heads;
  const int F = h->desc.input_dim, L = h->desc.num_blocks, KS = h->desc.cnn_module_kernel;
  const auto fd = h->front_dims(T);
  const int Tp = fd.Tp, Tr = (Tp + 1) / 2, M = B * Tp;  // Conv1D(stride 2): ceil(T'/2) reduced frames
  float *y1 = ws + wl.y1, *y2 = ws + wl.y2, *xa = ws + wl.x, *a = ws + wl.a, *big = ws + wl.big, *y = ws + wl.y;
  float *g = ws + wl.g, *ctx = ws + wl.ctx, *xs = ws + wl.xs, *xb = ws + wl.xr;
  const bool causal = h->desc.causal != 0;
  const int left = causal ? KS - 1 : (KS - 1) / 2;
  const int lo_s = s ? s->lo : 0;  // cached conv-input rows in front of the chunk
  // ---- DepthwiseConv2DSubsampling4 (dw_stride = False: two ordinary 3x3 / 2 convs) + input_proj + preln ----
  launch_conv1(feats, h->front, y1, B, T, F, fd.T1, h->F1, st, PadSkip{}, D);
  launch_conv_stage(y1, h->front.conv2_w, h->front.conv2_b, y2, B, fd.T1, h->F1, Tp, h->F2, 3, 2, st, PadSkip{}, D);
  {
    GemmEpi e;
    e.sb = 1;
    dense(y2, h->F2 * D, h->front.embed_w, h->front.embed_b, xa, M, h->F2 * D, D, D, D, st, sqrtf((float)D), e);
  }
  // ragged batches (ppasr_set_skip_padding, batched calls): as gen_layers, with the Squeezeformer's slack (capi_squeezeformer.hip:
  // the time reduction reads full-rate rows 2j - 3 .. 2j + 1, the recovery reduced row t / 2)
  const bool skip = h->skip_padding && lens && !s;
  const int rc = causal ? 0 : (KS - 1) / 2;
  auto pskip = [&](int Tcur, int mul_cur) {
    PadSkip p;
    if (skip) {
      p.lens = lens;
      p.Tp = Tcur;
      p.mul = mul_cur;
      p.slack = mul_cur == 4 ? 2 * (rc + 4) + rc + 8 : rc + 4;
    }
    return p;
  };
  const PadSkip psF = pskip(Tp, 4), psH = pskip(Tr, 8);
  auto ln = [&](const float* in, float* out, const float* gg, const float* bb, float eps, int act, bool mask, int rows, int Ti,
                int mul) {
    PPASR_LAUNCH(k_g_ln, dim3((rows + 3) / 4), dim3(256), 0, st, in, out, gg, bb, rows, D, eps, act, mask ? lens : nullptr, Ti, mul, 1.0f,
                 rows == B * Ti ? (mul == 4 ? psF : psH) : PadSkip{});
  };
  ln(xa, xa, h->preln_g, h->preln_b, 1e-5f, kActNone, false, M, Tp, 4);
  float* x = xa;
  bool reduced = false;
  for (int i = 0; i < L; ++i) {
    const SqLayerW& W = h->sq_layers[i];
    if (i == h->desc.reduce_idx) {
      HIP_TRY(hipMemcpyAsync(xs, x, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, st));
      PPASR_LAUNCH(k_g_sq_reduce_dw, blocks((size_t)B * Tr * D), dim3(256), 0, st, x, a, h->sq_reduce.dw_w, h->sq_reduce.dw_b,
                   h->sq_reduce.ks, lens, B, Tp, Tr, D);
      {
        GemmEpi e;
        e.ps = psH;
        dense(a, D, h->sq_reduce.pw, h->sq_reduce.pw_b, xb, B * Tr, D, D, D, D, st, 1.0f, e);
      }
      x = xb;
      reduced = true;
    }
    if (i == h->desc.recover_idx && reduced) {
      PPASR_LAUNCH(k_g_sq_repeat, blocks((size_t)M * D), dim3(256), 0, st, x, a, B, Tp, Tr, D);
      GemmEpi e;
      e.res = xs;  // recover_tensor + time_recover_layer(repeat_interleave(xs, 2))
      e.ps = psF;
      dense(a, D, h->sq_wrec, h->sq_brec, xa, M, D, D, D, D, st, 1.0f, e);
      x = xa;
      reduced = false;
    }
    const int Ti = reduced ? Tr : Tp, Mi = B * Ti, mul = reduced ? 8 : 4;
    const PadSkip& ps = reduced ? psH : psF;
    auto res_epi = [&](bool mask) {
      GemmEpi e;
      e.res = x;
      e.ps = ps;
      if (mask && lens) {
        e.lens = lens;
        e.Tp = Ti;
        e.mul = mul;
      }
      return e;
    };
    auto act_epi = [&]() {
      GemmEpi e;
      e.act = h->gen.act;  // activation_type (swish in every shipped YAML)
      e.ps = ps;
      return e;
    };
    auto plain_epi = [&]() {
      GemmEpi e;
      e.ps = ps;
      return e;
    };
    // normalize_before (squeezeformer/encoder.py:49,467-493; False in every shipped YAML): LayerNorm_k in FRONT of module k
    // (its output feeds the module, the residual stays x) instead of behind the residual sum
    const bool pre = h->gen.sq_pre_norm;
    // ---- x = LN1(x + MHA(x))   [pre: x + MHA(LN1(x))] ----
    if (pre) {
      ln(x, y, W.ln1_g, W.ln1_b, 1e-5f, kActNone, false, Mi, Ti, mul);
      dense(y, D, W.wqkv, W.bqkv, big, Mi, D, 3 * D, 3 * D, 3 * D, st, 1.0f, plain_epi());
    } else if (fused_ffn512() && D == kD512)
      PPASR_LAUNCH(k_g_proj512<false>, dim3((Mi + kRows - 1) / kRows), dim3(kThreads), kLdsProj512, st, x, big, 3 * D,
                   (const float*)nullptr, (const float*)nullptr, 1e-5f, (const int64_t*)nullptr, Ti, mul, W.wqkv, W.bqkv,
                   3 * D / 32 / kWaves, Mi, ps);
    else
      dense(x, D, W.wqkv, W.bqkv, big, Mi, D, 3 * D, 3 * D, 3 * D, st, 1.0f, plain_epi());
    // (pos_enc_layer_type != rel_pos: plain MultiHeadedAttention -- the positional operand is one row of zeros, stride 0)
    const bool plain_mha = h->gen.pos != PPASR_OPT_POS_REL;
    AttnArgs at{big, 3 * D, big + D, 3 * D, big + 2 * D, 3 * D, Ti, Ti, 0, lens, ctx, W.pos_u, W.pos_v, W.ptab,
                plain_mha ? 0 : (reduced ? 2 : 1),
                mul, Ti, Ti, 1};
    if (s) {  // keys / values: [cache | chunk] in the layer's device caches
      const int n_cache = reduced ? plan->used_r : s->cache_t;
      float* kc = s->kc + (size_t)i * s->cap * D;
      float* vc = s->vc + (size_t)i * s->cap * D;
      PPASR_LAUNCH(k_g_kv_append, blocks((size_t)Mi * D), dim3(256), 0, st, big, kc + (size_t)n_cache * D,
                   vc + (size_t)n_cache * D, Mi, D);
      at.k = kc;
      at.v = vc;
      at.k_stride = at.v_stride = D;
      at.T2 = at.kv_frames = n_cache + Ti;
      // (plain MHA: the table is ONE row of zeros -- k_attention_t adds pos0 rows to its base whatever the stride, so a
      //  chunk behind a trimmed cache (pos0 > 0) read past it: garbage in the positional half, a fault when the row sat at
      //  the end of a mapping -- found by the instrumented full suite, round 6)
      at.pos0 = plain_mha ? 0 : plan->pos0;
    }
    at.pad_skip = skip ? ps.slack + 1 : 0;
    at.dm = D;
    launch_attention(at, B, heads, st);
    dense(ctx, D, W.wo, W.bo, x, Mi, D, D, D, D, st, 1.0f, res_epi(false));
    if (!pre) ln(x, x, W.ln1_g, W.ln1_b, 1e-5f, kActNone, false, Mi, Ti, mul);
    // ---- x = LN2(x + FFN1(x))   [pre: x + FFN1(LN2(x))] ----
    auto sq_ffn = [&](const f32x4* w1, const float* b1, const f32x4* w2, const float* b2, const float* lg, const float* lb) {
      if (pre) {
        ln(x, y, lg, lb, 1e-5f, kActNone, false, Mi, Ti, mul);
        dense(y, D, w1, b1, big, Mi, D, H, H, H, st, 1.0f, act_epi());
        dense(big, H, w2, b2, x, Mi, H, D, D, D, st, 1.0f, res_epi(false));
        return;
      }
      if (fused_ffn512() && D == kD512) {
        PPASR_LAUNCH(k_g_ffn512, dim3((Mi + kRows - 1) / kRows), dim3(kThreads), kLdsFfn512, st, x, x, (const float*)nullptr,
                     (const float*)nullptr, w1, b1, w2, b2, 1.0f, h->gen.act, Mi, H / 256, ps);
        return;
      }
      dense(x, D, w1, b1, big, Mi, D, H, H, H, st, 1.0f, act_epi());
      dense(big, H, w2, b2, x, Mi, H, D, D, D, st, 1.0f, res_epi(false));
    };
    sq_ffn(W.ff1_w1, W.ff1_b1, W.ff1_w2, W.ff1_b2, W.ln2_g, W.ln2_b);
    if (!pre) ln(x, x, W.ln2_g, W.ln2_b, 1e-5f, kActNone, false, Mi, Ti, mul);
    // ---- x = LN3(x + conv(x)): ada scale / bias, THEN the pad mask (convolution.py:119-127), the unfolded pointwise_conv1 ----
    {
      float* a_new = a + (size_t)lo_s * D;
      const int rows = lo_s + Mi;
      if (pre) ln(x, y, W.ln3_g, W.ln3_b, 1e-5f, kActNone, false, Mi, Ti, mul);
      const float* cin = pre ? y : x;  // the conv module's input
      if (!pre && fused_ffn512() && D == kD512 && lo_s == 0) {  // ada scale / bias + pad mask + pointwise_conv1 + GLU in one launch
        PPASR_LAUNCH(k_g_proj512<true>, dim3((Mi + kRows - 1) / kRows), dim3(kThreads), kLdsProj512, st, x, g, D, W.cm_scale,
                     W.cm_bias, -1.0f, lens, Ti, mul, W.pw1_raw, W.pw1_b_raw, 2, Mi, ps);
      } else {
        ln(cin, a_new, W.cm_scale, W.cm_bias, -1.0f, kActNone, true, Mi, Ti, mul);
        if (lo_s) {  // the cache holds the SCALED inputs of the previous chunks; new cache = last lo rows of [cache | chunk]
          float* hist = s->xh_hist + (size_t)i * s->lo * D;
          HIP_TRY(hipMemcpyAsync(a, hist, (size_t)lo_s * D * sizeof(float), hipMemcpyDeviceToDevice, st));
          HIP_TRY(hipMemcpyAsync(hist, a + (size_t)Mi * D, (size_t)lo_s * D * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        dense(a, D, W.pw1_raw, W.pw1_b_raw, big, rows, D, 2 * D, 2 * D, 2 * D, st, 1.0f, lo_s ? GemmEpi{} : plain_epi());
        PPASR_LAUNCH(k_g_glu, blocks((size_t)rows * D), dim3(256), 0, st, big, g, rows, D);
      }
      launch_dwconv(g, y, W.dw_w, W.dw_b, W.glu_pad, B, Ti, D, KS, left, lo_s, 1, Ti, st, ps);
      ln(y, y, W.ln_cm_g, W.ln_cm_b, W.cm_eps, h->gen.act, false, Mi, Ti, mul);
      dense(y, D, W.pw2, W.pw2_b, x, Mi, D, D, D, D, st, 1.0f, res_epi(true));
      if (!pre) ln(x, x, W.ln3_g, W.ln3_b, 1e-5f, kActNone, false, Mi, Ti, mul);
    }
    // ---- x = LN4(x + FFN2(x)) ----
    sq_ffn(W.ff2_w1, W.ff2_b1, W.ff2_w2, W.ff2_b2, W.ln4_g, W.ln4_b);
    if (!pre) ln(x, x, W.ln4_g, W.ln4_b, 1e-5f, kActNone, false, Mi, Ti, mul);
  }
  // ---- ctc_lo -> softmax (no after_norm in Squeezeformer, encoder.py:232-235) ----
  float* lg = logits ? logits : (probs ? probs : ws + wl.lg);
  {
    GemmEpi e;
    e.ps = reduced ? PadSkip{} : psF;  // (without a recovery layer the encoder ends at the reduced rate: every row)
    dense(x, D, h->gen_head_w, h->gen_head_b, lg, M, D, h->gen_vpad, V, V, st, 1.0f, e);
  }
  float* pr = probs;
  if (!pr && (frame_argmax || frame_maxprob)) pr = (lg == ws + wl.lg) ? lg : ws + wl.lg;
  if (pr) {
    if (pr != lg) HIP_TRY(hipMemcpyAsync(pr, lg, (size_t)M * V * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_softmax_from_stats(pr, nullptr, nullptr, M, V, st);
    if (frame_argmax || frame_maxprob) {
      int32_t* fa = frame_argmax ? frame_argmax : reinterpret_cast<int32_t*>(y);
      float* fp = frame_maxprob ? frame_maxprob : y + M;
      launch_frame_argmax(pr, fa, fp, M, V, st);
    }
  }
  if (skip && !reduced) launch_zero_pad_rows(probs, logits, frame_argmax, frame_maxprob, lens, B, Tp, 4, V, st);
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}
}  // namespace

ppasr_status generic_sq_encode(ppasr_model_s* h, const float* feats, const int64_t* lens, int B, int T, float* probs,
                               float* logits, int32_t* frame_argmax, float* frame_maxprob, float* ws, hipStream_t st) {
  return sq_run(h, feats, lens, B, T, probs, logits, frame_argmax, frame_maxprob, ws, st, nullptr, nullptr);
}
ppasr_status generic_sq_chunk(ppasr_stream_s* s, const ChunkPlan& p, const float* feats, int T, float* probs,
                              int32_t* frame_argmax, float* frame_maxprob, float* ws, hipStream_t st) {
  return sq_run(s->m, feats, nullptr, 1, T, probs, nullptr, frame_argmax, frame_maxprob, ws, st, s, &p);
}

hipError_t configure_generic_kernels() {
  hipError_t e;
#define SET_LDS(k, bytes)                                                                                        \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                               (int)(bytes))) != hipSuccess)                                                     \
  return e
  SET_LDS((k_dense_epi<1, 256>), (dense_lds<1, 256>()));
  SET_LDS(k_g_ffn512, kLdsFfn512);
  SET_LDS(k_g_dwconv, (kDwLdsRows + 2) * 256 * sizeof(float));
  SET_LDS(k_g_proj512<false>, kLdsProj512);
  SET_LDS(k_g_proj512<true>, kLdsProj512);
#undef SET_LDS
  return hipSuccess;
}

size_t generic_ws_floats(const ppasr_model_s* m, int B, int T) { return gen_layout(m, B, T).total; }

ppasr_status generic_encode(ppasr_model_s* h, const float* feats, const int64_t* lens, int B, int T, float* probs,
                            float* logits, int32_t* frame_argmax, float* frame_maxprob, float* ws, hipStream_t st) {
  if (h->taps) return fail(PPASR_EUNSUPPORTED, "debug taps are built for the fused 256-wide route");
  const GenWs wl = gen_layout(h, B, T);
  ppasr_status rs = gen_front(h, feats, B, T, ws, wl, 0, st);
  if (rs != PPASR_OK) return rs;
  GenRun r{h, st, B, h->front_dims(T).Tp, lens, ws + wl.x, ws + wl.a, ws + wl.big, ws + wl.y, ws + wl.g, ws + wl.ctx,
           ws + wl.cat, ws + wl.lg};
  return gen_layers(r, probs, logits, frame_argmax, frame_maxprob);
}

ppasr_status generic_chunk(ppasr_stream_s* s, const ChunkPlan& p, const float* feats, int T, float* probs,
                           int32_t* frame_argmax, float* frame_maxprob, float* ws, hipStream_t st, int* frames_out) {
  ppasr_model_s* h = s->m;
  const GenWs wl = gen_layout(h, 1, T);
  // PositionalEncoding.forward(x, offset) (abs_pos): the chunk's first frame sits at position `offset`
  ppasr_status rs = gen_front(h, feats, 1, T, ws, wl, s->offset, st);
  if (rs != PPASR_OK) return rs;
  GenRun r{h, st, 1, h->front_dims(T).Tp, nullptr, ws + wl.x, ws + wl.a, ws + wl.big, ws + wl.y, ws + wl.g, ws + wl.ctx,
           ws + wl.cat, ws + wl.lg};
  r.s = s;
  r.pos0 = p.pos0;
  r.used_r = p.used_r;
  r.frames_out = frames_out;
  return gen_layers(r, probs, nullptr, frame_argmax, frame_maxprob);
}
