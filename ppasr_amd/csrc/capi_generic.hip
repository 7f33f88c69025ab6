// capi_generic.hip -- Conformer encoders whose width is NOT 256 (output_size 512 / 768 / 1024 with heads of 64).
//
// The fused row-block kernels (conformer_kernels.hip) are specialised for 256 columns = 8 waves x 32 and four
// 32 x 260 LDS buffers per workgroup; a 512-wide row block does not fit that scheme.  This route runs the same layer
// (ConformerEncoderLayer.forward, conformer/encoder.py:346-431) as a sequence of general pieces instead:
//   * every Linear / pointwise Conv1D  -> launch_dense: the fragment-ordered streamed-weight MFMA GEMM (k_gemm_stream),
//     the SAME packed weights the fused kernels use (ppasr_create packs them for any width);
//   * attention                        -> k_attention<64> with 8+ heads (AttnArgs::dm = model width);
//   * LayerNorm, swish, GLU, depthwise conv, residual / mask updates -> the small row kernels below (HBM-bound, one
//     pass each; none of the activations stays in LDS between them).
// Results follow the reference exactly like the fused route (same arithmetic, different fusion); it is several times
// slower per FLOP than the 256-wide kernels and exists for coverage of the non-shipped `output_size: 512,
// attention_heads: 8` configurations.  Batched encode only: no stream handles, no debug taps, no skip-padding mode.
#include "capi_internal.h"

using namespace ppasr;

namespace {

// LayerNorm over D columns (nn.LayerNorm, biased variance, eps inside the sqrt), one wave per row; optionally followed by
// swish, optionally with rows t of utterance b zeroed where mul * t >= lens[b] (the conv module's input mask,
// convolution.py:104-106).  eps < 0: per-channel affine only (folded BatchNorm1D, see capi.hip).
template <bool SWISH>
// (x and out may be the same buffer -- the encoder normalises in place -- so neither is __restrict__)
__global__ __launch_bounds__(256) void k_g_ln(const float* x, float* out,
                                              const float* __restrict__ g, const float* __restrict__ b, int M, int D,
                                              float eps, const int64_t* __restrict__ lens, int Tp, int mul) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
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
  if (eps >= 0.f) {
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
    float y = (xr[c] - mean) * rstd * g[c] + b[c];
    if (SWISH) y = swishf(y);
    o[c] = y;
  }
}

__global__ void k_g_swish(float* __restrict__ x, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = swishf(x[i]);
}

// x += scale * y, with rows of y dropped where the frame is PAD (lens != nullptr: the mask behind pointwise_conv2,
// convolution.py:138-140)
__global__ void k_g_axpy(float* __restrict__ x, const float* __restrict__ y, float scale, int M, int D,
                         const int64_t* __restrict__ lens, int Tp, int mul) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * D) return;
  if (lens) {
    const int row = (int)(i / D), bb = row / Tp, t = row - bb * Tp;
    if ((int64_t)mul * t >= lens[bb]) return;
  }
  x[i] += scale * y[i];
}

// GLU over the channel halves of pointwise_conv1's output: g[m][c] = pg[m][c] * sigmoid(pg[m][D + c])  (convolution.py:126)
__global__ void k_g_glu(const float* __restrict__ pg, float* __restrict__ g, int M, int D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * D) return;
  const size_t row = i / D, c = i - row * D;
  g[i] = pg[row * 2 * D + c] * sigmoidf(pg[row * 2 * D + D + c]);
}

// Depthwise conv over time of g [B*Tp][D] (KS taps, `left` frames of left context: KS-1 causal, (KS-1)/2 non-causal).
// Taps outside the utterance read pad[c] = GLU(pointwise_conv1 bias) in the causal module (the reference zero-pads
// BEFORE pointwise_conv1, convolution.py:108-126) and 0 in the non-causal one (its depthwise conv pads its own input).
__global__ void k_g_dwconv(const float* __restrict__ g, float* __restrict__ out, const float* __restrict__ w /*[KS][D]*/,
                           const float* __restrict__ bias, const float* __restrict__ pad, int M, int Tp, int D, int KS,
                           int left) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * D) return;
  const int row = (int)(i / D), c = (int)(i - (size_t)row * D);
  const int bb = row / Tp, t = row - bb * Tp;
  const bool causal = left == KS - 1;
  float acc = bias[c];
  for (int j = 0; j < KS; ++j) {
    const int tt = t - left + j;
    float v;
    if (tt >= 0 && tt < Tp) v = g[((size_t)bb * Tp + tt) * D + c];
    else v = causal ? pad[c] : 0.f;
    acc = fmaf(w[(size_t)j * D + c], v, acc);
  }
  out[i] = acc;
}

inline size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

struct GenWs {
  size_t y1, y2, x, a, big, y, g, ctx, lg, total;
};
GenWs gen_layout(const ppasr_model_s* m, int B, int T) {
  const int D = m->desc.output_size, H = m->desc.linear_units, V = m->desc.vocab_size;
  const auto fd = m->front_dims(T);
  const size_t M = (size_t)B * fd.Tp;
  const size_t wide = (size_t)std::max(std::max(3 * D, 2 * D), H);
  GenWs w{};
  size_t o = 0;
  w.y1 = o; o += al64((size_t)B * fd.T1 * m->F1 * D);
  w.y2 = o; o += al64(M * m->F2 * D);
  w.x = o; o += al64(M * D);
  w.a = o; o += al64(M * D);
  w.big = o; o += al64(M * wide);  // FFN hidden / qkv / pointwise_conv1 output
  w.y = o; o += al64(M * D);
  w.g = o; o += al64(M * D);
  w.ctx = o; o += al64(M * D);
  w.lg = o; o += al64(M * (size_t)V);  // logits / probabilities when the caller does not ask for them
  w.total = o;
  return w;
}

}  // namespace

size_t generic_ws_floats(const ppasr_model_s* m, int B, int T) { return gen_layout(m, B, T).total; }

ppasr_status generic_encode(ppasr_model_s* h, const float* feats, const int64_t* lens, int B, int T, float* probs,
                            float* logits, int32_t* frame_argmax, float* frame_maxprob, float* ws, hipStream_t st) {
  if (h->taps) return fail(PPASR_EUNSUPPORTED, "debug taps are built for output_size=256");
  const int D = h->desc.output_size, H = h->desc.linear_units, V = h->desc.vocab_size, heads = h->desc.attention_heads;
  const auto fd = h->front_dims(T);
  const int F = h->desc.input_dim, T1 = fd.T1, F1 = h->F1, Tp = fd.Tp, F2 = h->F2;
  const int M = B * Tp, mul = 4;
  const GenWs wl = gen_layout(h, B, T);
  float *y1 = ws + wl.y1, *y2 = ws + wl.y2, *x = ws + wl.x, *a = ws + wl.a, *big = ws + wl.big, *y = ws + wl.y;
  float *g = ws + wl.g, *ctx = ws + wl.ctx;
  const size_t MD = (size_t)M * D;
  auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
  auto ln = [&](const float* in, float* out, const float* gg, const float* bb, float eps, bool swish, bool mask) {
    if (swish)
      PPASR_LAUNCH(k_g_ln<true>, dim3((M + 3) / 4), dim3(256), 0, st, in, out, gg, bb, M, D, eps, mask ? lens : nullptr, Tp, mul);
    else
      PPASR_LAUNCH(k_g_ln<false>, dim3((M + 3) / 4), dim3(256), 0, st, in, out, gg, bb, M, D, eps, mask ? lens : nullptr, Tp, mul);
  };
  auto axpy = [&](const float* yy, float scale, bool mask) {
    PPASR_LAUNCH(k_g_axpy, blocks(MD), dim3(256), 0, st, x, yy, scale, M, D, mask ? lens : nullptr, Tp, mul);
  };
  // PositionwiseFeedForward (positionwise.py:32-39): x += 0.5 * W2 swish(W1 LN(x) + b1) + b2
  auto ffn = [&](const float* lg, const float* lb, const f32x4* w1, const float* b1, const f32x4* w2, const float* b2) {
    ln(x, a, lg, lb, 1e-5f, false, false);
    launch_dense(a, D, w1, b1, big, M, D, H, H, H, st);
    PPASR_LAUNCH(k_g_swish, blocks((size_t)M * H), dim3(256), 0, st, big, (size_t)M * H);
    launch_dense(big, H, w2, b2, y, M, H, D, D, D, st);
    axpy(y, 0.5f, false);
  };

  // ---- front end: GlobalCMVN + Conv2dSubsampling4 + x * sqrt(d) (subsampling.py:96-115, embedding.py:112) ----
  launch_conv1(feats, h->front, y1, B, T, F, T1, F1, st, PadSkip{}, D);
  launch_conv_stage(y1, h->front.conv2_w, h->front.conv2_b, y2, B, T1, F1, Tp, F2, 3, 2, st, PadSkip{}, D);
  launch_dense(y2, F2 * D, h->front.embed_w, h->front.embed_b, x, M, F2 * D, D, D, D, st, sqrtf((float)D));

  const int left = h->desc.causal ? h->desc.cnn_module_kernel - 1 : (h->desc.cnn_module_kernel - 1) / 2;
  for (int i = 0; i < h->desc.num_blocks; ++i) {
    const LayerW& L = h->layers[i];
    const int KS = h->layer_ks[i];
    ffn(L.ln_mac_g, L.ln_mac_b, L.ffm_w1, L.ffm_b1, L.ffm_w2, L.ffm_b2);
    // ---- RelPositionMultiHeadedAttention (attention.py:198-262) ----
    ln(x, a, L.ln_mha_g, L.ln_mha_b, 1e-5f, false, false);
    launch_dense(a, D, L.wqkv, L.bqkv, big, M, D, 3 * D, 3 * D, 3 * D, st);
    AttnArgs at{big, 3 * D, big + D, 3 * D, big + 2 * D, 3 * D, Tp, Tp, 0, lens, ctx, L.pos_u, L.pos_v, L.ptab, 1, mul, Tp, Tp, 1};
    at.pad_skip = 0;
    at.dm = D;
    launch_attention(at, B, heads, st);
    launch_dense(ctx, D, L.wo, L.bo, y, M, D, D, D, D, st);
    axpy(y, 1.0f, false);
    // ---- ConvolutionModule (convolution.py:82-143) ----
    ln(x, a, L.ln_conv_g, L.ln_conv_b, 1e-5f, false, true);  // LN_conv, PAD frames -> 0
    launch_dense(a, D, L.pw1, L.pw1_b, big, M, D, 2 * D, 2 * D, 2 * D, st);
    PPASR_LAUNCH(k_g_glu, blocks(MD), dim3(256), 0, st, big, g, M, D);
    PPASR_LAUNCH(k_g_dwconv, blocks(MD), dim3(256), 0, st, g, a, L.dw_w, L.dw_b, L.glu_pad, M, Tp, D, KS, left);
    ln(a, a, L.ln_cm_g, L.ln_cm_b, L.cm_eps, true, false);  // LayerNorm / folded BatchNorm + swish
    launch_dense(a, D, L.pw2, L.pw2_b, y, M, D, D, D, D, st);
    axpy(y, 1.0f, true);  // PAD frames of the conv output -> 0, then the residual
    ffn(L.ln_ff_g, L.ln_ff_b, L.ff_w1, L.ff_b1, L.ff_w2, L.ff_b2);
    ln(x, x, L.ln_fin_g, L.ln_fin_b, 1e-5f, false, false);
  }
  // ---- after_norm -> ctc_lo -> softmax (encoder.py:201, ctc.py:62-70) ----
  ln(x, a, h->head.ln_g, h->head.ln_b, 1e-5f, false, false);
  float* lg = logits ? logits : (probs ? probs : ws + wl.lg);
  launch_dense(a, D, h->gen_head_w, h->gen_head_b, lg, M, D, h->gen_vpad, V, V, st);
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
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}
