// capi_stream.hip -- streaming entry points: <Family>Encoder.forward_chunk with the attention K/V cache and the
// conv-module cache resident on the device inside a stream-state object (the reference round-trips both through the
// host on every chunk, infer_utils/inference_predictor.py:196-210).
//   Conformer            conformer/encoder.py:208-283
//   Squeezeformer        squeezeformer/encoder.py:260-381   (time-reduced layers 5..10, recover at 11)
//   Efficient-Conformer  efficient_conformer/encoder.py:266-393 (grouped attention 0..3, stride layer 3, 7-tap convs after)
//
// Cache bookkeeping.  Every layer owns K and V caches [cap][256] (row = frame) and a conv-module input history
// [lo_max][256] (the reference's cnn_cache, frame-major; its first lo_i rows are used, lo_i = kernel_i - 1).
// Layers that run at half rate (after a time reduction / the stride layer) hold each cached frame ONCE; the reference
// stores those caches repeat_interleave'd to the full rate and reads them back with [::2], which is the identity on
// the values.  The frame COUNTS follow the reference's slicing exactly (including Squeezeformer's trim of the reduced
// cache to len(pos_emb) - len(xs) and of the exported cache to the first layer's length); a combination for which the
// reference itself fails with a shape error (odd cache lengths) is refused with PPASR_EINVAL.
#include <algorithm>

#include "capi_internal.h"

namespace {

inline bool is_sq(const ppasr_model_s* h) { return h->desc.model_type == PPASR_MODEL_SQUEEZEFORMER; }
inline bool is_eff(const ppasr_model_s* h) { return h->desc.model_type == PPASR_MODEL_EFFICIENT_CONFORMER; }

// calculate_downsampling_factor (squeezeformer/encoder.py:246-258, efficient_conformer/encoder.py:205-210)
inline int layer_factor(const ppasr_model_s* h, int i) {
  if (is_sq(h)) return (h->desc.reduce_idx >= 0 && i >= h->desc.reduce_idx && !(h->desc.recover_idx >= 0 && i >= h->desc.recover_idx)) ? 2 : 1;
  if (is_eff(h)) return (h->desc.stride_layer_idx >= 0 && i > h->desc.stride_layer_idx) ? 2 : 1;
  return 1;
}
inline int layer_lo(const ppasr_model_s* h, int i) {
  return (is_sq(h) ? h->desc.cnn_module_kernel : h->layer_ks[i]) - 1;
}
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// keep rows [from, from+keep) of a [cap][256] cache at its start
ppasr_status shift_cache(float* buf, int from, int keep, float* tmp, hipStream_t st, int D = kD) {
  if (keep <= 0 || from <= 0) return PPASR_OK;
  const size_t bytes = (size_t)keep * D * sizeof(float);
  HIP_TRY(hipMemcpyAsync(tmp, buf + (size_t)from * D, bytes, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(buf, tmp, bytes, hipMemcpyDeviceToDevice, st));
  return PPASR_OK;
}

// pointwise_conv1 + GLU of the cached conv inputs of EVERY layer -> s->g_hist[i] (lo_i rows each), one launch: the
// histories are last chunk's state (hist_update of layer i runs after layer i has consumed g_hist[i])
void history_glu_all(ppasr_stream_s* s, hipStream_t st) {
  launch_pw1_glu_layers(s->xh_hist, s->g_hist, s->hist_tab, s->m->desc.num_blocks, s->lo, st);
}

// the reference's shape arithmetic for one chunk
ppasr_status plan_chunk(const ppasr_stream_s* s, int c, int required_cache_size, ChunkPlan* p) {
  const ppasr_model_s* h = s->m;
  p->c = c;
  p->c_r = ceil_div(c, 2);  // Conv1D(k=1, s=2) / stride-2 depthwise conv + ceil-mode AvgPool
  p->T2 = s->cache_t + c;
  const int pos_len_r = ceil_div(s->cache_t + c, 2);  // pos_emb[:, ::2]
  if (is_sq(h)) {
    // att_cache[i][:, :, ::2][:, :, :pos_len - xs_len] of a cache exported as repeat_interleave(...)[:max_att_len]
    const int avail = ceil_div(std::min(2 * s->cache_r, s->cache_t), 2);
    p->used_r = std::min(avail, pos_len_r - p->c_r);
  } else {
    p->used_r = s->cache_r;  // att_cache[i][:, :, ::2], no trim
  }
  p->T2_r = p->used_r + p->c_r;
  const bool has_half = is_sq(h) ? h->desc.reduce_idx >= 0 : (is_eff(h) && h->desc.stride_layer_idx >= 0);
  if (has_half && p->T2_r != pos_len_r)
    return fail(PPASR_EINVAL, "half-rate attention cache does not line up with the strided positional table "
                              "(the reference fails on matrix_ac + matrix_bd here): keep cache lengths even");
  if (required_cache_size < 0) p->ncs = 0;
  else if (required_cache_size == 0) p->ncs = p->T2;
  else p->ncs = std::max(p->T2 - required_cache_size, 0);
  // efficient_conformer/encoder.py:305: offset *= calculate_downsampling_factor(num_blocks + 1)
  const int off = is_eff(h) && h->desc.stride_layer_idx >= 0 ? 2 * s->offset : s->offset;
  p->pos0 = off - s->cache_t;
  if (p->pos0 < 0) return fail(PPASR_EINVAL, "offset smaller than the attention cache length");
  if (p->pos0 + p->T2 >= h->desc.max_len) return fail(PPASR_EINVAL, "offset + chunk exceeds the positional table (max_len)");
  if (p->T2 > s->cap) return fail(PPASR_EINVAL, "attention cache capacity exceeded");
  return PPASR_OK;
}

ppasr_status finish_chunk(ppasr_stream_s* s, const ChunkPlan& p, float* shift_tmp, hipStream_t st) {
  ppasr_model_s* h = s->m;
  const int keep = p.T2 - p.ncs;
  const int from_r = p.ncs / 2;
  const int keep_r = std::max(p.T2_r - from_r, 0);
  if (h->desc.num_blocks <= 64 && s->D % 4 == 0 && s->D <= 1024) {  // one launch for every layer's K and V cache
    unsigned long long half_mask = 0;
    for (int i = 0; i < h->desc.num_blocks; ++i)
      if (layer_factor(h, i) == 2) half_mask |= 1ull << i;
    if ((p.ncs > 0 && keep > 0) || (half_mask && from_r > 0 && keep_r > 0))
      launch_shift_caches(s->kc, s->vc, (long long)s->cap * s->D, s->D, h->desc.num_blocks, p.ncs, keep, from_r, keep_r, half_mask, st);
    s->cache_t = keep;
    s->cache_r = keep_r;
    return PPASR_OK;
  }
  for (int i = 0; i < h->desc.num_blocks; ++i) {
    float* bufs[2] = {s->kc + (size_t)i * s->cap * s->D, s->vc + (size_t)i * s->cap * s->D};
    const bool half = layer_factor(h, i) == 2;
    for (float* b : bufs) {
      ppasr_status r = half ? shift_cache(b, from_r, keep_r, shift_tmp, st, s->D) : shift_cache(b, p.ncs, keep, shift_tmp, st, s->D);
      if (r != PPASR_OK) return r;
    }
  }
  s->cache_t = keep;
  s->cache_r = keep_r;
  return PPASR_OK;
}

// ---- Conformer / Efficient-Conformer ----
// `partial`: scratch for the split route of under-filled launches (ppasr_set_ffn_split; a chunk is ONE row block, so by
// default its feed-forward modules are split over 8 workgroups) -- the conv1 buffer, free after the front-end
ppasr_status conformer_chunk(ppasr_stream_s* s, const ChunkPlan& p, float* xa, float* xb, float* xc, float* qkv, float* ctx,
                             float* g, float* xhat, float* partial, int* frames_out, hipStream_t st) {
  ppasr_model_s* h = s->m;
  const int n_chunks = h->desc.linear_units / 256;
  const int H = h->desc.attention_heads;
  int Ti = p.c, mul = 4, pstride = 1;
  bool half = false;
  history_glu_all(s, st);
  // Consumer-side joins (conformer_kernels.h JoinIn; fp32 route, <= 16 rows): a feed-forward module leaves 2 S partial tiles
  // and a PENDING join that the next launch computes in its prologue.  Macaron slices go to partial, final slices to
  // the tiles behind them (the pending final join of block i is read while block i + 1's macaron slices are written).
  JoinIn pending;
  auto flush = [&](int M) {  // a launch that cannot take a pending join in: the join alone
    if (pending.partial) launch_join16(pending, M, st);
    pending = JoinIn{};
  };
  for (int i = 0; i < h->desc.num_blocks; ++i) {
    const LayerW& L = h->layers[i];
    const int grp = h->layer_group[i];
    const int n_cache = half ? p.used_r : s->cache_t;
    const int T2f = n_cache + Ti;
    float* kc = s->kc + (size_t)i * s->cap * kD;
    float* vc = s->vc + (size_t)i * s->cap * kD;
    float* xh = s->xh_hist + (size_t)i * s->lo * kD;
    const int lo_i = layer_lo(h, i);
    const int S = ffn_split_for(h, Ti);
    // ppasr_set_gemm_mode(h, PPASR_GEMM_F16X3): the split route's GEMM units on the fp16 x3 route (Lk = the layer's h3 view: the
    // same LayerNorm / bias pointers, re-packed weights; its ptab are operand planes, so the attention below keeps L's).
    // Out-of-range activations are saturated and counted (ppasr_gemm_guard_stats); a chunk is not re-run.
    const bool h3 = S > 1 && h->gemm_mode == PPASR_GEMM_F16X3 && !h->layers_h3.empty();
    const LayerW& Lk = h3 ? h->layers_h3[i] : L;
    const bool fused_joins = !h3 && ffn_half16_route(Ti, S, n_chunks);
    if (fused_joins) {
      launch_ffn_half16(xa, pending, L.ln_mac_g, L.ln_mac_b, L.ffm_w1, L.ffm_b1, L.ffm_w2, partial, Ti, n_chunks, st);
      const JoinIn mac{partial, 2 * S, L.ffm_b2, 0.5f, xa, nullptr, nullptr, xb};
      launch_join_ln_qkv16(mac, qkv, L, Ti, st, kc + (size_t)n_cache * kD, vc + (size_t)n_cache * kD);
      pending = JoinIn{};
    } else if (S > 1) {
      flush(Ti);
      launch_ffn_split(xa, L.ln_mac_g, L.ln_mac_b, Lk.ffm_w1, L.ffm_b1, Lk.ffm_w2, L.ffm_b2, 0.5f, nullptr, nullptr, partial, xb,
                       Ti, n_chunks, S, st, PadSkip{}, false, h3, s->ticket);
      launch_ln_qkv(xb, qkv, Lk, Ti, st, PadSkip{}, kc + (size_t)n_cache * kD, vc + (size_t)n_cache * kD, h3);
    } else {
      flush(Ti);
      launch_ffn_qkv(xa, xb, qkv, L, Ti, n_chunks, st);
      launch_kv_append(qkv, kc + (size_t)n_cache * kD, vc + (size_t)n_cache * kD, Ti, st);
    }
    // grouped attention re-cuts cache + chunk frames into groups of 3 from the START of the cache (pad4group on the
    // concatenated keys, efficient_conformer/attention.py:160-175), zero-padded tail group
    AttnArgs a{qkv, 768, kc, kD, vc, kD, ceil_div(Ti, grp), ceil_div(T2f, grp), p.pos0, nullptr, ctx, L.pos_u, L.pos_v,
               L.ptab, pstride, mul * grp, Ti, T2f, grp};
    launch_attention(a, 1, H, st);
    float* gh = s->g_hist + (size_t)i * s->lo * kD;
    HistMove hm{xh, lo_i, false};
    launch_out_glu(ctx, xb, xc, g, xhat, Lk, nullptr, Ti, Ti, mul, st, PadSkip{}, S > 1 ? xhat : nullptr, h3, &hm);
    if (is_eff(h) && i == h->desc.stride_layer_idx) {
      const int Ts = ceil_div(Ti, 2);
      const int Ss = ffn_split_for(h, Ts);
      if (Ss > 1) {  // one row block: the conv half of the stride layer alone, its feed-forward module over the slices
        launch_conv_ffn_stride(g, gh, xc, xa, Lk, nullptr, 1, Ti, Ts, n_chunks, h->layer_ks[i], mul * 2, st, PadSkip{}, true, h3, ctx);
        launch_ffn_split(ctx, L.ln_ff_g, L.ln_ff_b, Lk.ff_w1, L.ff_b1, Lk.ff_w2, L.ff_b2, 0.5f, L.ln_fin_g, L.ln_fin_b, partial, xa, Ts,
                         n_chunks, Ss, st, PadSkip{}, false, h3, s->ticket);
      } else {
        launch_conv_ffn_stride(g, gh, xc, xa, Lk, nullptr, 1, Ti, Ts, n_chunks, h->layer_ks[i], mul * 2, st, PadSkip{}, true, h3);
      }
      if (!hm.done) launch_hist_update(xh, xhat, Ti, lo_i, st);
      Ti = Ts;
      mul *= 2;
      pstride *= 2;
      half = true;
    } else {
      if (fused_joins) {
        launch_conv_pre(g, gh, xc, ctx, L, nullptr, Ti, Ti, h->layer_ks[i], mul, st, true, PadSkip{}, false);
        float* part_fin = partial + (size_t)2 * S * Ti * kD;  // (behind the macaron module's 2 S tiles of Ti rows)
        launch_ffn_half16(ctx, JoinIn{}, L.ln_ff_g, L.ln_ff_b, L.ff_w1, L.ff_b1, L.ff_w2, part_fin, Ti, n_chunks, st);
        pending = JoinIn{part_fin, 2 * S, L.ff_b2, 0.5f, ctx, L.ln_fin_g, L.ln_fin_b, xa};
      } else if (S > 1) {
        launch_conv_pre(g, gh, xc, ctx, Lk, nullptr, Ti, Ti, h->layer_ks[i], mul, st, true, PadSkip{}, h3);
        launch_ffn_split(ctx, L.ln_ff_g, L.ln_ff_b, Lk.ff_w1, L.ff_b1, Lk.ff_w2, L.ff_b2, 0.5f, L.ln_fin_g, L.ln_fin_b, partial,
                         xa, Ti, n_chunks, S, st, PadSkip{}, false, h3, s->ticket);
      } else {
        launch_conv_ffn(g, gh, xc, xa, L, nullptr, Ti, Ti, n_chunks, h->layer_ks[i], mul, nullptr, nullptr, nullptr, st);
      }
      if (!hm.done) launch_hist_update(xh, xhat, Ti, lo_i, st);
    }
  }
  flush(Ti);  // (the last block's final join)
  *frames_out = Ti;
  return PPASR_OK;
}

// ---- Squeezeformer ----
ppasr_status squeezeformer_chunk(ppasr_stream_s* s, const ChunkPlan& p, float* xa, float* xb, float* xc, float* qkv,
                                 float* ctx, float* g, float* xs, float* xhat, float* partial, float** x_final,
                                 hipStream_t st) {
  ppasr_model_s* h = s->m;
  const int L = h->desc.num_blocks, H = h->desc.attention_heads;
  const int n_chunks = h->desc.linear_units / 256, KS = h->desc.cnn_module_kernel;
  launch_ln_rows(xa, h->preln_g, h->preln_b, p.c, st);
  history_glu_all(s, st);
  float* x = xa;
  float* other = xb;
  bool reduced = false, have_qkv = false;
  // Round 6: on the split route (fp32) the layer's single-unit launches run on the Conformer's 16-row kernels through weight
  // views -- Q / K / V thirds (no LayerNorm: ln_mha_g = nullptr; K and V straight into the cache rows), out-projection +
  // LayerNorm, pointwise_conv1 + GLU over two column halves (which also moves the SCALED conv-input history on)
  bool kv_in_cache = false;  // this layer's K / V rows were written to its cache by the launch that made its qkv
  auto qkv_view = [](const SqLayerW& w) {
    LayerW v{};
    v.wqkv = w.wqkv;
    v.bqkv = w.bqkv;
    return v;
  };
  for (int i = 0; i < L; ++i) {
    const SqLayerW& W = h->sq_layers[i];
    if (i == h->desc.reduce_idx) {
      HIP_TRY(hipMemcpyAsync(xs, x, (size_t)p.c * kD * sizeof(float), hipMemcpyDeviceToDevice, st));
      launch_sq_reduce(x, other, qkv, h->sq_reduce, W.wqkv, W.bqkv, nullptr, 1, p.c, p.c_r, st);
      std::swap(x, other);
      reduced = true;
      have_qkv = true;
    }
    if (i == h->desc.recover_idx && reduced) {
      launch_sq_recover(x, xs, other, qkv, h->sq_wrec, h->sq_brec, W.wqkv, W.bqkv, 1, p.c, p.c_r, st);
      std::swap(x, other);
      reduced = false;
      have_qkv = true;
    }
    const int Ti = reduced ? p.c_r : p.c;
    const int mul = reduced ? 8 : 4;
    const int n_cache = reduced ? p.used_r : s->cache_t;
    float* kc = s->kc + (size_t)i * s->cap * kD;
    float* vc = s->vc + (size_t)i * s->cap * kD;
    float* xh = s->xh_hist + (size_t)i * s->lo * kD;
    const int S = ffn_split_for(h, Ti);  // one row block: split route (see squeezeformer_encode)
    const bool h3s = S > 1 && h->gemm_mode == PPASR_GEMM_F16X3 && !h->sq_layers_h3.empty();  // (the FFN slices in the mode)
    const bool r16 = S > 1 && !h3s && Ti <= split_rows16_max();
    if (!have_qkv) {
      if (r16) {
        launch_ln_qkv(x, qkv, qkv_view(W), Ti, st, PadSkip{}, kc + (size_t)n_cache * kD, vc + (size_t)n_cache * kD, false);
        kv_in_cache = true;
      } else {
        launch_sq_qkv(x, qkv, W.wqkv, W.bqkv, Ti, st);
      }
    }
    if (!kv_in_cache) launch_kv_append(qkv, kc + (size_t)n_cache * kD, vc + (size_t)n_cache * kD, Ti, st);
    kv_in_cache = false;
    AttnArgs a{qkv, 768, kc, kD, vc, kD, Ti, n_cache + Ti, p.pos0, nullptr, ctx, W.pos_u, W.pos_v, W.ptab, reduced ? 2 : 1,
               mul, Ti, n_cache + Ti, 1};
    launch_attention(a, 1, H, st);
    float* gh = s->g_hist + (size_t)i * s->lo * kD;
    const bool fuse_next = (i + 1 < L) && (i + 1 != h->desc.reduce_idx) && !(i + 1 == h->desc.recover_idx && reduced);
    const SqLayerW* Wn = fuse_next ? &h->sq_layers[i + 1] : nullptr;
    bool hist_moved = false;
    if (S > 1) {
      const SqLayerW& Ws = h3s ? h->sq_layers_h3[i] : W;
      if (r16) {  // x1 = LN1(x + ctx Wo + bo) (the plain sum goes to g, dead until pointwise_conv1 writes it)
        LayerW vo{};
        vo.wo = W.wo; vo.bo = W.bo; vo.ln_conv_g = W.ln1_g; vo.ln_conv_b = W.ln1_b;
        launch_oproj_ln_16(ctx, x, g, other, vo, Ti, st);
      } else {
        launch_sq_oproj(ctx, x, other, W, Ti, st);
      }
      launch_ffn_split(other, nullptr, nullptr, Ws.ff1_w1, W.ff1_b1, Ws.ff1_w2, W.ff1_b2, 1.0f, W.ln2_g, W.ln2_b, partial, xc,
                       Ti, n_chunks, S, st, PadSkip{}, false, h3s, s->ticket);
      if (r16 && KS - 1 <= 30) {
        LayerW vp{};
        vp.pw1 = W.pw1; vp.pw1_b = W.pw1_b;
        launch_pw1_glu_cols_16(xc, g, vp, Ti, st, xh, KS - 1, W.cm_scale, W.cm_bias);
        hist_moved = true;
      } else {
        launch_sq_pw1glu(xc, g, xhat, W, nullptr, Ti, Ti, mul, st);
      }
      launch_conv_pre(g, gh, xc, ctx, sq_conv_view(W), nullptr, Ti, Ti, KS, mul, st);
      launch_ffn_split(ctx, W.ln3_g, W.ln3_b, Ws.ff2_w1, W.ff2_b1, Ws.ff2_w2, W.ff2_b2, 1.0f, W.ln4_g, W.ln4_b, partial, other,
                       Ti, n_chunks, S, st, PadSkip{}, /*residual_is_normed=*/true, h3s, s->ticket);
      if (Wn && r16) {  // (fuse_next: layer i + 1 runs at this layer's rate -- same rows, same cache length)
        float* kcn = s->kc + (size_t)(i + 1) * s->cap * kD;
        float* vcn = s->vc + (size_t)(i + 1) * s->cap * kD;
        launch_ln_qkv(other, qkv, qkv_view(*Wn), Ti, st, PadSkip{}, kcn + (size_t)n_cache * kD, vcn + (size_t)n_cache * kD, false);
        kv_in_cache = true;
      } else if (Wn) {
        launch_sq_qkv(other, qkv, Wn->wqkv, Wn->bqkv, Ti, st);
      }
    } else {
      launch_sq_mid(ctx, x, xc, g, xhat, W, nullptr, Ti, Ti, mul, n_chunks, st);
      launch_sq_tail(g, gh, xc, other, qkv, W, Wn ? Wn->wqkv : nullptr, Wn ? Wn->bqkv : nullptr, nullptr, Ti, Ti, mul,
                     n_chunks, KS, st);
    }
    if (!hist_moved) launch_hist_update(xh, xhat, Ti, KS - 1, st);
    std::swap(x, other);
    have_qkv = fuse_next;
  }
  *x_final = x;
  return PPASR_OK;
}

}  // namespace

extern "C" {

ppasr_status ppasr_stream_create(ppasr_handle h, ppasr_stream* out) {
  if (!h || !out) return fail(PPASR_EINVAL, "null argument");
  if (h->desc.model_type == PPASR_MODEL_DEEPSPEECH2)
    return fail(PPASR_EUNSUPPORTED, "deepspeech2 streams carry their state in the h/c boxes of ppasr_ds2_encode");
  if (!h->desc.causal)
    return fail(PPASR_EUNSUPPORTED, "forward_chunk needs the causal conv module (a streaming=True model)");
  if (h->generic && h->desc.input_layer == 1)
    return fail(PPASR_EUNSUPPORTED, "stream handles are built for the conv front ends (input_layer=linear: batched encode)");
  if (__builtin_popcount(eff_stride_mask(h->desc)) > 1)
    return fail(PPASR_EUNSUPPORTED, "stream handles are built for one stride layer (several: batched encode)");
  if (h->desc.input_layer != 0 && h->desc.model_type != PPASR_MODEL_CONFORMER)
    return fail(PPASR_EUNSUPPORTED, "stream handles with the conv2d6 / conv2d8 front ends are built for model_type=conformer");
  auto* s = new ppasr_stream_s();
  s->m = h;
  s->D = h->desc.output_size;
  s->cap = h->desc.max_len;
  s->lo = (h->generic && !h->gen.use_cnn) ? 0 : h->desc.cnn_module_kernel - 1;
  const size_t L = h->desc.num_blocks;
  const size_t D = s->D, lo_alloc = s->lo > 0 ? s->lo : 1;
  hipError_t e1 = hipMalloc(reinterpret_cast<void**>(&s->kc), L * s->cap * D * sizeof(float));
  hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&s->vc), L * s->cap * D * sizeof(float));
  hipError_t e3 = hipMalloc(reinterpret_cast<void**>(&s->xh_hist), L * lo_alloc * D * sizeof(float));
  hipError_t e4 = hipMalloc(reinterpret_cast<void**>(&s->g_hist), L * lo_alloc * D * sizeof(float));
  s->hist_tab = nullptr;
  if (e4 == hipSuccess) e4 = hipMalloc(reinterpret_cast<void**>(&s->hist_tab), L * sizeof(HistLayer));
  if (e4 == hipSuccess) e4 = hipMalloc(reinterpret_cast<void**>(&s->ticket), 16 * sizeof(int));
  if (e4 == hipSuccess) e4 = hipMemset(s->ticket, 0, 16 * sizeof(int));
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
    (void)hipFree(s->kc); (void)hipFree(s->vc); (void)hipFree(s->xh_hist); (void)hipFree(s->g_hist);
    (void)hipFree(s->hist_tab);
    (void)hipFree(s->ticket);
    delete s;
    return fail(PPASR_EHIP, "hipMalloc failed for the stream caches");
  }
  s->cache_t = 0;
  s->cache_r = 0;
  s->offset = 0;
  hipError_t e5 = hipMemset(s->xh_hist, 0, L * lo_alloc * D * sizeof(float));
  std::vector<HistLayer> tab(L);
  for (size_t i = 0; i < L; ++i) {
    if (h->generic) tab[i] = HistLayer{nullptr, nullptr, 0, 0};  // (the general route recomputes the history's GLU inside the layer)
    else if (is_sq(h)) tab[i] = HistLayer{h->sq_layers[i].pw1_raw, h->sq_layers[i].pw1_b_raw, layer_lo(h, (int)i), 0};
    else tab[i] = HistLayer{h->layers[i].pw1, h->layers[i].pw1_b, layer_lo(h, (int)i), 0};
  }
  if (e5 == hipSuccess) e5 = hipMemcpy(s->hist_tab, tab.data(), L * sizeof(HistLayer), hipMemcpyHostToDevice);
  if (e5 != hipSuccess) {
    (void)ppasr_stream_destroy(s);
    return fail(PPASR_EHIP, "initialising the stream caches failed");
  }
  *out = s;
  return PPASR_OK;
}

ppasr_status ppasr_stream_destroy(ppasr_stream s) {
  if (!s) return PPASR_OK;
  (void)hipFree(s->kc); (void)hipFree(s->vc); (void)hipFree(s->xh_hist); (void)hipFree(s->g_hist);
  (void)hipFree(s->hist_tab);
  (void)hipFree(s->ticket);
  delete s;
  return PPASR_OK;
}

// InferencePredictor.reset_stream (inference_predictor.py:215-220): empty caches, offset 0
ppasr_status ppasr_stream_reset(ppasr_stream s, void* stream) {
  if (!s) return fail(PPASR_EINVAL, "null stream");
  s->cache_t = 0;
  s->cache_r = 0;
  s->offset = 0;
  HIP_TRY(hipMemsetAsync(s->xh_hist, 0, (size_t)s->m->desc.num_blocks * s->lo * s->D * sizeof(float),
                         static_cast<hipStream_t>(stream)));
  return PPASR_OK;
}

int ppasr_stream_offset(ppasr_stream s) { return s ? s->offset : -1; }
int ppasr_stream_cache_frames(ppasr_stream s) { return s ? s->cache_t : -1; }

size_t ppasr_chunk_workspace_bytes(ppasr_handle h, int T) {
  if (!h || T < h->min_frames()) return 0;
  const size_t Tp = h->front_dims(T).Tp;
  if (h->generic) return (generic_ws_floats(h, 1, T) + (size_t)h->desc.max_len * h->desc.output_size) * sizeof(float);
  // the full-utterance layout for B=1, plus the conv-module input rows and a cache-shift scratch
  // (+ the K-split scratch of the chunk's conv2 launch)
  return (ws_layout(h, 1, T).total + Tp * kD + 64 + (size_t)h->desc.max_len * kD + conv_stage_part_floats((int)Tp * h->F2)) *
         sizeof(float);
}

ppasr_status ppasr_encode_chunk(ppasr_stream s, const float* feats, int T, int required_cache_size, float* probs,
                                int32_t* frame_argmax, float* frame_maxprob, int* c_out_host, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!s || !feats || !workspace) return fail(PPASR_EINVAL, "null argument");
  ppasr_model_s* h = s->m;
  if (T < h->min_frames())
    return fail(PPASR_EINVAL, "chunk shorter than the conv front-end's receptive field (7 frames; conv2d6: 11, conv2d8: 15)");
  const auto fd = h->front_dims(T);
  const int F = h->desc.input_dim, T1 = fd.T1, F1 = h->F1, c = fd.Tp, F2 = h->F2;
  if (workspace_bytes < ppasr_chunk_workspace_bytes(h, T)) return fail(PPASR_ENOSPACE, "workspace too small");
  ChunkPlan p{};
  ppasr_status r = plan_chunk(s, c, required_cache_size, &p);
  if (r != PPASR_OK) return r;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const WsLayout wl = ws_layout(h, 1, T);
  float* ws = static_cast<float*>(workspace);
  if (h->generic) {  // the general layer route (capi_generic.hip): same cache bookkeeping, its own layer pieces
    int gframes = c;
    r = is_sq(h) ? generic_sq_chunk(s, p, feats, T, probs, frame_argmax, frame_maxprob, ws, st)
                 : generic_chunk(s, p, feats, T, probs, frame_argmax, frame_maxprob, ws, st, &gframes);
    if (r != PPASR_OK) return r;
    r = finish_chunk(s, p, ws + wl.total, st);
    if (r != PPASR_OK) return r;
    s->offset += gframes;
    if (c_out_host) *c_out_host = gframes;
    return PPASR_OK;
  }
  float *y1 = ws + wl.y1, *y2 = ws + wl.y2, *xa = ws + wl.xa, *xb = ws + wl.xb, *xc = ws + wl.xc;
  float *qkv = ws + wl.qkv, *ctx = ws + wl.ctx, *g = ws + wl.g;
  float* xhat = ws + wl.total;
  float* shift_tmp = xhat + (((size_t)c * kD + 63) & ~(size_t)63);
  launch_conv1(feats, h->front, y1, 1, T, F, T1, F1, st);
  if (h->desc.input_layer == 8) {  // Conv2dSubsampling8: three 3x3 / 2 convs, the third one over conv1's output buffer
    launch_conv_stage(y1, h->front.conv2_w, h->front.conv2_b, y2, 1, T1, F1, fd.T2, F2, 3, 2, st);
    launch_conv_stage(y2, h->front.conv3_w, h->front.conv3_b, y1, 1, fd.T2, F2, c, h->F3, 3, 2, st);
    launch_embed(y1, h->front, xa, c, h->F3 * kD, sqrtf((float)kD), false, st, PadSkip{}, ffn_split_for(h, c), y2);
  } else {
    // (3x3 / 2, or conv2d6's 5x5 / 3: FrontW::conv2_k / _s)
    launch_conv2(y1, h->front, y2, 1, T1, F1, c, F2, st, PadSkip{}, nullptr, nullptr, shift_tmp + (size_t)h->desc.max_len * kD,
                 conv_stage_part_floats(c * F2));
    // (one row block: a workgroup per 256-wide K chunk -- F2 of them -- instead of 8 workgroups of 2 - 3 chunks)
    launch_embed(y2, h->front, xa, c, F2 * kD, sqrtf((float)kD), /*scale_before_bias=*/is_sq(h), st, PadSkip{},
                 (c <= 32 && ffn_split_for(h, c) > 1) ? F2 : ffn_split_for(h, c), y1);
  }
  float* x_final = xa;
  int frames = c;
  if (is_sq(h)) r = squeezeformer_chunk(s, p, xa, xb, xc, qkv, ctx, g, ws + wl.xs, xhat, y1, &x_final, st);
  else r = conformer_chunk(s, p, xa, xb, xc, qkv, ctx, g, xhat, y1, &frames, st);
  if (r != PPASR_OK) return r;
  int32_t* fa = frame_argmax ? frame_argmax : reinterpret_cast<int32_t*>(ws + wl.fa);
  float* fp = frame_maxprob ? frame_maxprob : ws + wl.fp;
  // (one row block: as many column slices as give every wave ONE 32-column vocabulary tile -- 17 at V = 4233 -- not 8)
  const int head_slices = (frames <= 32 && ffn_split_for(h, frames) > 1) ? std::min((h->head.n_tiles + 7) / 8, 32)
                                                                        : ffn_split_for(h, frames);
  launch_ctc_head(x_final, h->head, probs, fa, fp, ws + wl.rmax, ws + wl.rsum, frames, st, PadSkip{}, head_slices, y1);
  if (probs) launch_softmax_from_stats(probs, ws + wl.rmax, ws + wl.rsum, frames, h->head.V, st);
  r = finish_chunk(s, p, shift_tmp, st);
  if (r != PPASR_OK) return r;
  s->offset += frames;
  if (c_out_host) *c_out_host = frames;
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

// Reference-layout views of the caches (what get_encoder_out_chunk returns, conformer/model.py:164-184):
// att_cache [L][h][t][2*dk] (t = ppasr_stream_cache_frames), cnn_cache [L][1][256][lo].
ppasr_status ppasr_stream_export_cache(ppasr_stream s, float* att_cache, float* cnn_cache, void* stream) {
  if (!s) return fail(PPASR_EINVAL, "null stream");
  ppasr_model_s* h = s->m;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int L = h->desc.num_blocks, t = s->cache_t;
  for (int i = 0; i < L; ++i) {
    const int div = layer_factor(h, i);
    if (att_cache && t > 0) {
      // repeat_interleave(cache, 2)[:max_att_len] must cover t frames, or the reference's concat over layers fails
      if (div == 2 && (2 * s->cache_r < t || (is_eff(h) && 2 * s->cache_r != t)))
        return fail(PPASR_EINVAL, "half-rate cache does not match the first layer's cache length (odd cache length)");
      launch_cache_export(s->kc + (size_t)i * s->cap * s->D, s->vc + (size_t)i * s->cap * s->D,
                          att_cache + (size_t)i * (s->D / 64) * t * 128, t, div, st, s->D);
    }
    if (cnn_cache && s->lo > 0)
      launch_cnn_transpose(s->xh_hist + (size_t)i * s->lo * s->D, cnn_cache + (size_t)i * s->D * s->lo, layer_lo(h, i), s->lo, 1, st, s->D);
  }
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

ppasr_status ppasr_stream_import_cache(ppasr_stream s, const float* att_cache, int cache_t, const float* cnn_cache,
                                       int offset, void* stream) {
  if (!s) return fail(PPASR_EINVAL, "null stream");
  if (cache_t < 0 || cache_t > s->cap || offset < 0) return fail(PPASR_EINVAL, "bad cache_t / offset");
  if (cache_t > 0 && !att_cache) return fail(PPASR_EINVAL, "att_cache missing");
  ppasr_model_s* h = s->m;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int L = h->desc.num_blocks;
  for (int i = 0; i < L; ++i) {
    if (cache_t > 0)
      launch_cache_import(att_cache + (size_t)i * (s->D / 64) * cache_t * 128, s->kc + (size_t)i * s->cap * s->D,
                          s->vc + (size_t)i * s->cap * s->D, cache_t, layer_factor(h, i), st, s->D);
    if (cnn_cache && s->lo > 0)
      launch_cnn_transpose(cnn_cache + (size_t)i * s->D * s->lo, s->xh_hist + (size_t)i * s->lo * s->D, layer_lo(h, i), s->lo, 0, st, s->D);
  }
  if (!cnn_cache && s->lo > 0)
    HIP_TRY(hipMemsetAsync(s->xh_hist, 0, (size_t)L * s->lo * s->D * sizeof(float), st));
  s->cache_t = cache_t;
  s->cache_r = ceil_div(cache_t, 2);  // att_cache[i][:, :, ::2]
  s->offset = offset;
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

// =====================================================================================
// Multi-session streaming: a group of Conformer sessions whose caches live in one allocation and advance with ONE set
// of launches per chunk round (the rows of all active sessions are stacked: n x c frames -> ceil(n*c/32) row blocks
// per kernel instead of one).  No reference counterpart: PPASR streams one session per call
// (predict.py:232-337, forward_chunk asserts B = 1); each session here follows exactly the single-session arithmetic
// (required_cache_size < 0: the full history is kept, what PPASRPredictor passes, predict.py:306-307).
// =====================================================================================
struct ppasr_stream_group_s {
  ppasr_model_s* m;
  int n_sessions, cap, lo;
  float *kc, *vc;     // [n_sessions][L][cap][256]
  float* xh_hist;     // [n_sessions][L][lo][256]
  // per-call descriptors: a ring of pinned host staging buffers + device copies, each guarded by an event, so that a
  // call never overwrites a buffer an earlier (still queued) call reads
  static constexpr int kRing = 8;
  SessDesc* desc_host;  // pinned [kRing][n_sessions]
  SessDesc* desc_dev;   // device [kRing][n_sessions]
  hipEvent_t ev[kRing];
  int slot;
  std::vector<int> cache_t, offset;
};

ppasr_status ppasr_stream_group_create(ppasr_handle h, int n_sessions, int max_frames, ppasr_stream_group* out) {
  if (!h || !out || n_sessions < 1) return fail(PPASR_EINVAL, "bad argument");
  if (h->desc.model_type != PPASR_MODEL_CONFORMER || !h->desc.causal)
    return fail(PPASR_EUNSUPPORTED, "session groups are built for streaming (causal) model_type=conformer");
  if (h->desc.input_layer != 0) return fail(PPASR_EUNSUPPORTED, "session groups are built for the conv2d front end only");
  if (h->generic) return fail(PPASR_EUNSUPPORTED, "session groups are built for the fused 256-wide route");
  auto g = std::make_unique<ppasr_stream_group_s>();
  g->m = h;
  g->n_sessions = n_sessions;
  g->cap = (max_frames > 0 && max_frames < h->desc.max_len) ? max_frames : h->desc.max_len;
  g->lo = h->desc.cnn_module_kernel - 1;
  const size_t L = h->desc.num_blocks;
  const size_t kv = (size_t)n_sessions * L * g->cap * kD * sizeof(float);
  const size_t hb = (size_t)n_sessions * L * g->lo * kD * sizeof(float);
  g->kc = g->vc = g->xh_hist = nullptr;
  g->desc_dev = g->desc_host = nullptr;
  g->slot = 0;
  const size_t db = (size_t)ppasr_stream_group_s::kRing * n_sessions * sizeof(SessDesc);
  hipError_t e1 = hipMalloc(reinterpret_cast<void**>(&g->kc), kv);
  hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&g->vc), kv);
  hipError_t e3 = hipMalloc(reinterpret_cast<void**>(&g->xh_hist), hb);
  hipError_t e4 = hipMalloc(reinterpret_cast<void**>(&g->desc_dev), db);
  hipError_t e5 = hipHostMalloc(reinterpret_cast<void**>(&g->desc_host), db, hipHostMallocDefault);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess) {
    (void)hipFree(g->kc); (void)hipFree(g->vc); (void)hipFree(g->xh_hist); (void)hipFree(g->desc_dev);
    (void)hipHostFree(g->desc_host);
    return fail(PPASR_EHIP, "allocation failed for the session-group caches");
  }
  for (int i = 0; i < ppasr_stream_group_s::kRing; ++i) HIP_TRY(hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming));
  HIP_TRY(hipMemset(g->xh_hist, 0, hb));
  g->cache_t.assign(n_sessions, 0);
  g->offset.assign(n_sessions, 0);
  *out = g.release();
  return PPASR_OK;
}

ppasr_status ppasr_stream_group_destroy(ppasr_stream_group g) {
  if (!g) return PPASR_OK;
  (void)hipFree(g->kc); (void)hipFree(g->vc); (void)hipFree(g->xh_hist); (void)hipFree(g->desc_dev);
  (void)hipHostFree(g->desc_host);
  for (int i = 0; i < ppasr_stream_group_s::kRing; ++i) (void)hipEventDestroy(g->ev[i]);
  delete g;
  return PPASR_OK;
}

// session < 0: every session
ppasr_status ppasr_stream_group_reset(ppasr_stream_group g, int session, void* stream) {
  if (!g || session >= g->n_sessions) return fail(PPASR_EINVAL, "bad session");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t per = (size_t)g->m->desc.num_blocks * g->lo * kD;
  if (session < 0) {
    HIP_TRY(hipMemsetAsync(g->xh_hist, 0, per * g->n_sessions * sizeof(float), st));
    std::fill(g->cache_t.begin(), g->cache_t.end(), 0);
    std::fill(g->offset.begin(), g->offset.end(), 0);
  } else {
    HIP_TRY(hipMemsetAsync(g->xh_hist + per * session, 0, per * sizeof(float), st));
    g->cache_t[session] = 0;
    g->offset[session] = 0;
  }
  return PPASR_OK;
}

int ppasr_stream_group_offset(ppasr_stream_group g, int session) {
  return (g && session >= 0 && session < g->n_sessions) ? g->offset[session] : -1;
}

size_t ppasr_group_chunk_workspace_bytes(ppasr_handle h, int n, int T) {
  if (!h || n < 1 || T < 7) return 0;
  const size_t Tp = ((T - 1) / 2 - 1) / 2;
  // the batched layout for B = n, plus the conv-module input rows and the GLU'd histories of the active sessions
  return (ws_layout(h, n, T).total + (size_t)n * Tp * kD + 64 + (size_t)n * (h->desc.cnn_module_kernel - 1) * kD * 2 + 64) *
         sizeof(float);
}

// One chunk [T frames] for each of the n DISTINCT sessions listed in sessions_host; feats [n][T][F] (device).
// Outputs are indexed by position in the list: probs [n][c][V] (or NULL), frame_argmax / frame_maxprob [n][c].
ppasr_status ppasr_encode_chunk_group(ppasr_stream_group g, const int* sessions_host, int n, const float* feats, int T,
                                      float* probs, int32_t* frame_argmax, float* frame_maxprob, int* c_out_host,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || !sessions_host || !feats || !workspace || n < 1 || n > g->n_sessions) return fail(PPASR_EINVAL, "bad argument");
  ppasr_model_s* h = g->m;
  if (T < 7) return fail(PPASR_EINVAL, "chunk shorter than the conv front-end's receptive field (7 frames)");
  if (workspace_bytes < ppasr_group_chunk_workspace_bytes(h, n, T)) return fail(PPASR_ENOSPACE, "workspace too small");
  const int F = h->desc.input_dim, T1 = (T - 1) / 2, F1 = h->F1, c = (T1 - 1) / 2, F2 = h->F2;
  const int slot = g->slot;
  g->slot = (slot + 1) % ppasr_stream_group_s::kRing;
  HIP_TRY(hipEventSynchronize(g->ev[slot]));  // the call that last used this slot has consumed it (no-op when unused)
  SessDesc* desc = g->desc_host + (size_t)slot * g->n_sessions;
  SessDesc* desc_dev = g->desc_dev + (size_t)slot * g->n_sessions;
  std::vector<char> seen(g->n_sessions, 0);
  for (int b = 0; b < n; ++b) {
    const int sidx = sessions_host[b];
    if (sidx < 0 || sidx >= g->n_sessions || seen[sidx]) return fail(PPASR_EINVAL, "session index out of range or repeated");
    seen[sidx] = 1;
    if (g->cache_t[sidx] + c > g->cap) return fail(PPASR_EINVAL, "attention cache capacity exceeded");
    if (g->offset[sidx] + c >= h->desc.max_len) return fail(PPASR_EINVAL, "offset + chunk exceeds the positional table (max_len)");
    desc[b] = SessDesc{sidx, g->cache_t[sidx], g->offset[sidx] - g->cache_t[sidx], 0};
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemcpyAsync(desc_dev, desc, (size_t)n * sizeof(SessDesc), hipMemcpyHostToDevice, st));
  const WsLayout wl = ws_layout(h, n, T);
  float* ws = static_cast<float*>(workspace);
  float *y1 = ws + wl.y1, *y2 = ws + wl.y2, *xa = ws + wl.xa, *xb = ws + wl.xb, *xc = ws + wl.xc;
  float *qkv = ws + wl.qkv, *ctx = ws + wl.ctx, *gg = ws + wl.g;
  float* xhat = ws + wl.total;
  const int lo = g->lo;
  float* xh_act = xhat + (((size_t)n * c * kD + 63) & ~(size_t)63);  // [n][lo][256] gathered histories
  float* g_hist = xh_act + (size_t)n * lo * kD;                       // [n][lo][256] GLU(pointwise_conv1(history))
  const int M = n * c, L = h->desc.num_blocks, H = h->desc.attention_heads;
  const int n_chunks = h->desc.linear_units / 256;
  const long long kv_sess = (long long)L * g->cap * kD, hist_sess = (long long)L * lo * kD;
  launch_conv1(feats, h->front, y1, n, T, F, T1, F1, st);
  launch_conv2(y1, h->front, y2, n, T1, F1, c, F2, st);
  launch_embed(y2, h->front, xa, M, F2 * kD, sqrtf((float)kD), false, st, PadSkip{}, ffn_split_for(h, M), y1);
  for (int i = 0; i < L; ++i) {
    const LayerW& W = h->layers[i];
    float* kc = g->kc + (size_t)i * g->cap * kD;
    float* vc = g->vc + (size_t)i * g->cap * kD;
    float* xh = g->xh_hist + (size_t)i * lo * kD;
    const int S = ffn_split_for(h, M);  // few sessions = an under-filled grid: split route (partial sums in y1)
    const bool h3 = S > 1 && h->gemm_mode == PPASR_GEMM_F16X3 && !h->layers_h3.empty();  // (see conformer_chunk)
    const LayerW& Wk = h3 ? h->layers_h3[i] : W;
    if (S > 1) {
      launch_ffn_split(xa, W.ln_mac_g, W.ln_mac_b, Wk.ffm_w1, W.ffm_b1, Wk.ffm_w2, W.ffm_b2, 0.5f, nullptr, nullptr, y1, xb, M,
                       n_chunks, S, st, PadSkip{}, false, h3);
      launch_ln_qkv(xb, qkv, Wk, M, st, PadSkip{}, nullptr, nullptr, h3);
    } else {
      launch_ffn_qkv(xa, xb, qkv, W, M, n_chunks, st);
    }
    launch_kv_append_group(qkv, kc, vc, kv_sess, desc_dev, n, c, st);
    AttnArgs a{qkv, 768, kc, kD, vc, kD, c, c, 0, nullptr, ctx, W.pos_u, W.pos_v, W.ptab, 1, 4, c, c, 1, desc_dev, kv_sess};
    launch_attention(a, n, H, st);
    launch_hist_gather(xh, hist_sess, desc_dev, xh_act, n, lo, st);
    launch_pw1_glu(xh_act, g_hist, W, n * lo, st);
    launch_out_glu(ctx, xb, xc, gg, xhat, Wk, nullptr, M, c, 4, st, PadSkip{}, S > 1 ? xhat : nullptr, h3);
    if (S > 1) {
      launch_conv_pre(gg, g_hist, xc, ctx, Wk, nullptr, M, c, h->desc.cnn_module_kernel, 4, st, true, PadSkip{}, h3);
      launch_ffn_split(ctx, W.ln_ff_g, W.ln_ff_b, Wk.ff_w1, W.ff_b1, Wk.ff_w2, W.ff_b2, 0.5f, W.ln_fin_g, W.ln_fin_b, y1, xa, M,
                       n_chunks, S, st, PadSkip{}, false, h3);
    } else {
      launch_conv_ffn(gg, g_hist, xc, xa, W, nullptr, M, c, n_chunks, h->desc.cnn_module_kernel, 4, nullptr, nullptr, nullptr, st);
    }
    launch_hist_update_group(xh, hist_sess, desc_dev, xhat, n, c, lo, st);
  }
  int32_t* fa = frame_argmax ? frame_argmax : reinterpret_cast<int32_t*>(ws + wl.fa);
  float* fp = frame_maxprob ? frame_maxprob : ws + wl.fp;
  launch_ctc_head(xa, h->head, probs, fa, fp, ws + wl.rmax, ws + wl.rsum, M, st, PadSkip{}, ffn_split_for(h, M), y1);
  if (probs) launch_softmax_from_stats(probs, ws + wl.rmax, ws + wl.rsum, M, h->head.V, st);
  for (int b = 0; b < n; ++b) {
    g->cache_t[sessions_host[b]] += c;
    g->offset[sessions_host[b]] += c;
  }
  HIP_TRY(hipEventRecord(g->ev[slot], st));
  if (c_out_host) *c_out_host = c;
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

}  // extern "C"
