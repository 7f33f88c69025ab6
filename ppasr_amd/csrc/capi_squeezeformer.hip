// capi_squeezeformer.hip -- weight packing and launch sequence of the Squeezeformer encoder
// (ppasr/model_utils/squeezeformer/{encoder,attention,convolution,positionwise,subsampling,
// time_reduction}.py) behind ppasr_create / ppasr_encode.
#include "capi_internal.h"

namespace {

struct Getter {
  BlobMap& sd;
  std::string missing;
  const float* operator()(const std::string& name, size_t numel) {
    auto it = sd.find(name);
    if (it == sd.end() || it->second.numel() != numel) {
      if (missing.empty()) missing = name;
      return nullptr;
    }
    return it->second.p;
  }
};

std::vector<float> vec_of(const float* p, size_t n) { return std::vector<float>(p, p + n); }

}  // namespace

#define GETW(var, name, numel)                 \
  const float* var = get(name, (size_t)(numel)); \
  if (!var) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + get.missing)
#define UP(vec, dst) \
  if ((st = m->upload(vec, &(dst))) != PPASR_OK) return st
#define UP4(vec, dst) \
  if ((st = m->upload4(vec, &(dst))) != PPASR_OK) return st

ppasr_status squeezeformer_create(ppasr_model_s* m, BlobMap& sd, const float* pe_dev) {
  const ppasr_model_desc& dsc = m->desc;
  const int F = dsc.input_dim, d = dsc.output_size, H = dsc.linear_units, V = dsc.vocab_size, KS = dsc.cnn_module_kernel;
  const int F2 = m->F2, L = dsc.num_blocks, max_len = dsc.max_len;
  if (dsc.reduce_idx >= 0 && (dsc.recover_idx <= dsc.reduce_idx || dsc.recover_idx >= L || dsc.reduce_idx == 0))
    return fail(PPASR_EUNSUPPORTED, "squeezeformer: need 0 < reduce_idx < recover_idx < num_blocks (or reduce_idx = -1)");
  Getter get{sd, ""};
  ppasr_status st;
  // adaptive_scale = False (encoder.py:44): the ada_scale / ada_bias parameters exist in the checkpoint (every module creates
  // them, attention.py:34-37) but are not applied -- fold ones / zeros instead
  const bool no_ada = (dsc.options & PPASR_OPT_SQ_NO_ADAPTIVE_SCALE) != 0;
  const std::vector<float> ones_d(d, 1.f), zeros_d(d, 0.f);
  auto adapt = [&](const float*& as, const float*& ab) {
    if (no_ada) {
      as = ones_d.data();
      ab = zeros_d.data();
    }
  };
  {  // DepthwiseConv2DSubsampling4 (subsampling.py:36-47): two 3x3 / 2 convs; dw_stride = True makes the second one depthwise
    GETW(mean, "encoder.global_cmvn.mean", F);
    GETW(istd, "encoder.global_cmvn.istd", F);
    GETW(c1w, "encoder.embed.pw_conv.weight", d * 9);
    GETW(c1b, "encoder.embed.pw_conv.bias", d);
    // dw_stride = True (groups = odim): weight [d][1][3][3].  Run as the ordinary conv with a block-diagonal weight -- the
    // off-diagonal products are exact zeros, so the sums are the depthwise conv's; no shipped config sets the option
    const float* c2w = get("encoder.embed.dw_conv.weight", (size_t)d * d * 9);
    const bool dw_stride = !c2w;
    if (!c2w) {
      get.missing.clear();
      c2w = get("encoder.embed.dw_conv.weight", (size_t)d * 9);
    }
    if (!c2w) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + get.missing);
    GETW(c2b, "encoder.embed.dw_conv.bias", d);
    GETW(ew, "encoder.embed.input_proj.0.weight", (size_t)d * F2 * d);
    GETW(eb, "encoder.embed.input_proj.0.bias", d);
    GETW(pg, "encoder.preln.weight", d);
    GETW(pb, "encoder.preln.bias", d);
    UP(vec_of(mean, F), m->front.cmvn_mean);
    UP(vec_of(istd, F), m->front.cmvn_istd);
    std::vector<float> c1(9 * d);
    for (int c = 0; c < d; ++c)
      for (int j = 0; j < 9; ++j) c1[j * d + c] = c1w[c * 9 + j];
    UP(c1, m->front.conv1_w);
    UP(vec_of(c1b, d), m->front.conv1_b);
    UP4(pack_b(9 * d, d, [&](int k, int n) {
          if (dw_stride) return (k % d) == n ? c2w[(size_t)n * 9 + (k / d)] : 0.f;
          return c2w[((size_t)n * d + (k % d)) * 9 + (k / d)];
        }), m->front.conv2_w);
    UP(vec_of(c2b, d), m->front.conv2_b);
    UP4(pack_b(F2 * d, d, [&](int k, int n) { return ew[((size_t)(k % d) * F2 + (k / d)) * d + n]; }), m->front.embed_w);
    UP(vec_of(eb, d), m->front.embed_b);
    UP(vec_of(pg, d), m->preln_g);
    UP(vec_of(pb, d), m->preln_b);
  }
  m->sq_layers.resize(L);
  for (int i = 0; i < L; ++i) {
    SqLayerW& W = m->sq_layers[i];
    const std::string p = "encoder.encoders." + std::to_string(i) + ".";
    auto ln = [&](const std::string& n, const float** g, const float** b) -> ppasr_status {
      const float* gw = get(p + n + ".weight", d);
      const float* gb = get(p + n + ".bias", d);
      if (!gw || !gb) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + get.missing);
      ppasr_status s1 = m->upload(vec_of(gw, d), g);
      return s1 != PPASR_OK ? s1 : m->upload(vec_of(gb, d), b);
    };
    if ((st = ln("layer_norm1", &W.ln1_g, &W.ln1_b)) != PPASR_OK) return st;
    if ((st = ln("layer_norm2", &W.ln2_g, &W.ln2_b)) != PPASR_OK) return st;
    if ((st = ln("layer_norm3", &W.ln3_g, &W.ln3_b)) != PPASR_OK) return st;
    if ((st = ln("layer_norm4", &W.ln4_g, &W.ln4_b)) != PPASR_OK) return st;
    // conv-module norm: LayerNorm, or (cnn_norm_type: batch_norm) BatchNorm1D at inference folded into scale / shift,
    // cm_eps < 0 -- same convention as the Conformer loader (capi.hip)
    W.cm_eps = 1e-5f;
    if (sd.find(p + "conv_module.norm._mean") != sd.end()) {
      const float* mean = get(p + "conv_module.norm._mean", d);
      const float* var = get(p + "conv_module.norm._variance", d);
      const float* gw = get(p + "conv_module.norm.weight", d);
      const float* gb = get(p + "conv_module.norm.bias", d);
      if (!mean || !var || !gw || !gb) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + get.missing);
      std::vector<float> sc(d), sh(d);
      for (int c = 0; c < d; ++c) {
        sc[c] = gw[c] / std::sqrt(var[c] + 1e-5f);
        sh[c] = gb[c] - mean[c] * sc[c];
      }
      if ((st = m->upload(sc, &W.ln_cm_g)) != PPASR_OK || (st = m->upload(sh, &W.ln_cm_b)) != PPASR_OK) return st;
      W.cm_eps = -1.f;
    } else if ((st = ln("conv_module.norm", &W.ln_cm_g, &W.ln_cm_b)) != PPASR_OK) {
      return st;
    }
    // FFN with the adaptive scale folded into w_1:  (s.x + a) W1 + b1 = x (diag(s) W1) + (a W1 + b1)
    auto ffn = [&](const std::string& n, const f32x4** w1, const float** b1, const f32x4** w2,
                   const float** b2) -> ppasr_status {
      const float* a1 = get(p + n + ".w_1.weight", (size_t)d * H);
      const float* c1 = get(p + n + ".w_1.bias", H);
      const float* a2 = get(p + n + ".w_2.weight", (size_t)H * d);
      const float* c2 = get(p + n + ".w_2.bias", d);
      const float* as = get(p + n + ".ada_scale", d);
      const float* ab = get(p + n + ".ada_bias", d);
      if (!a1 || !c1 || !a2 || !c2 || !as || !ab) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + get.missing);
      adapt(as, ab);
      std::vector<float> b1f(c1, c1 + H);
      for (int nn = 0; nn < H; ++nn) {
        double acc = 0.0;
        for (int k = 0; k < d; ++k) acc += (double)ab[k] * (double)a1[(size_t)k * H + nn];
        b1f[nn] = (float)((double)c1[nn] + acc);
      }
      ppasr_status s;
      if ((s = m->upload4(pack_b(d, H, [&](int k, int nn) { return as[k] * a1[(size_t)k * H + nn]; }), w1)) != PPASR_OK) return s;
      if ((s = m->upload(b1f, b1)) != PPASR_OK) return s;
      if ((s = m->upload4(pack_b(H, d, [&](int k, int nn) { return a2[(size_t)k * d + nn]; }), w2)) != PPASR_OK) return s;
      return m->upload(vec_of(c2, d), b2);
    };
    if ((st = ffn("ffn1", &W.ff1_w1, &W.ff1_b1, &W.ff1_w2, &W.ff1_b2)) != PPASR_OK) return st;
    if ((st = ffn("ffn2", &W.ff2_w1, &W.ff2_b1, &W.ff2_w2, &W.ff2_b2)) != PPASR_OK) return st;
    {
      GETW(wq, p + "self_attn.linear_q.weight", d * d);
      GETW(wk, p + "self_attn.linear_k.weight", d * d);
      GETW(wv, p + "self_attn.linear_v.weight", d * d);
      GETW(bq, p + "self_attn.linear_q.bias", d);
      GETW(bk, p + "self_attn.linear_k.bias", d);
      GETW(bv, p + "self_attn.linear_v.bias", d);
      GETW(wo, p + "self_attn.linear_out.weight", d * d);
      GETW(bo, p + "self_attn.linear_out.bias", d);
      // pos_enc_layer_type != rel_pos (squeezeformer/encoder.py:101-105): conformer's plain MultiHeadedAttention -- no
      // linear_pos, no pos_bias_u / _v, no adaptive scale.  The attention kernels then contract the positional half with a
      // row of zeros (pos_bias = 0, table stride 0), like the Conformer's abs_pos / no_pos layers
      const bool plain_mha = (dsc.options & PPASR_OPT_POS_MASK) != PPASR_OPT_POS_REL;
      const float *wp = nullptr, *bp = nullptr, *pu = zeros_d.data(), *pv = zeros_d.data(), *as = ones_d.data(), *ab = zeros_d.data();
      if (!plain_mha) {
        wp = get(p + "self_attn.linear_pos.weight", (size_t)d * d);
        bp = get(p + "self_attn.linear_pos.bias", d);  // linear_pos HAS a bias here (squeezeformer/attention.py:28)
        pu = get(p + "self_attn.pos_bias_u", d);
        pv = get(p + "self_attn.pos_bias_v", d);
        as = get(p + "self_attn.ada_scale", d);
        ab = get(p + "self_attn.ada_bias", d);
        if (!wp || !bp || !pu || !pv || !as || !ab) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + get.missing);
        adapt(as, ab);
      }
      const float* ws[3] = {wq, wk, wv};
      const float* bs[3] = {bq, bk, bv};
      UP4(pack_b(d, 3 * d, [&](int k, int n) { return as[k] * ws[n / d][(size_t)k * d + (n % d)]; }), W.wqkv);
      std::vector<float> bqkv(3 * d);
      for (int n = 0; n < 3 * d; ++n) {
        double acc = 0.0;
        for (int k = 0; k < d; ++k) acc += (double)ab[k] * (double)ws[n / d][(size_t)k * d + (n % d)];
        bqkv[n] = (float)((double)bs[n / d][n % d] + acc);
      }
      UP(bqkv, W.bqkv);
      UP4(pack_b(d, d, [&](int k, int n) { return wo[(size_t)k * d + n]; }), W.wo);
      UP(vec_of(bo, d), W.bo);
      UP(vec_of(pu, d), W.pos_u);
      UP(vec_of(pv, d), W.pos_v);
      if (plain_mha) {
        W.ptab = W.pos_u;  // (a row of zeros; read with stride 0)
      } else {
        const float* wpos_dev = nullptr;
        UP(vec_of(wp, (size_t)d * d), wpos_dev);
        const float* bpos_dev = nullptr;
        UP(vec_of(bp, d), bpos_dev);
        void* pt = nullptr;
        HIP_TRY(hipMalloc(&pt, (size_t)max_len * d * sizeof(float)));
        m->allocs.push_back(pt);
        launch_posproj(pe_dev, wpos_dev, bpos_dev, static_cast<float*>(pt), max_len, nullptr, d);
        HIP_TRY(hipGetLastError());
        W.ptab = static_cast<const float*>(pt);
      }
    }
    {
      GETW(p1w, p + "conv_module.pointwise_conv1.weight", 2 * d * d);
      GETW(p1b, p + "conv_module.pointwise_conv1.bias", 2 * d);
      GETW(dww, p + "conv_module.depthwise_conv.weight", d * KS);
      GETW(dwb, p + "conv_module.depthwise_conv.bias", d);
      GETW(p2w, p + "conv_module.pointwise_conv2.weight", d * d);
      GETW(p2b, p + "conv_module.pointwise_conv2.bias", d);
      GETW(as, p + "conv_module.ada_scale", d);
      GETW(ab, p + "conv_module.ada_bias", d);
      adapt(as, ab);
      UP4(pack_b(d, 2 * d, [&](int k, int n) { return as[k] * p1w[(size_t)n * d + k]; }), W.pw1);
      std::vector<float> b1f(2 * d), gp(d);
      for (int n = 0; n < 2 * d; ++n) {
        double acc = 0.0;
        for (int k = 0; k < d; ++k) acc += (double)ab[k] * (double)p1w[(size_t)n * d + k];
        b1f[n] = (float)((double)p1b[n] + acc);
      }
      // zero-padded / PAD frames see pointwise_conv1(0) = the ORIGINAL bias (mask is applied after the scale)
      for (int c = 0; c < d; ++c) gp[c] = p1b[c] * (1.0f / (1.0f + expf(-p1b[c + d])));
      UP(b1f, W.pw1_b);
      UP(gp, W.glu_pad);
      UP(vec_of(as, d), W.cm_scale);
      UP(vec_of(ab, d), W.cm_bias);
      UP4(pack_b(d, 2 * d, [&](int k, int n) { return p1w[(size_t)n * d + k]; }), W.pw1_raw);
      UP(vec_of(p1b, 2 * d), W.pw1_b_raw);
      std::vector<float> dwt((size_t)KS * d);
      for (int c = 0; c < d; ++c)
        for (int j = 0; j < KS; ++j) dwt[(size_t)j * d + c] = dww[(size_t)c * KS + j];
      UP(dwt, W.dw_w);
      UP(vec_of(dwb, d), W.dw_b);
      UP4(pack_b(d, d, [&](int k, int n) { return p2w[(size_t)n * d + k]; }), W.pw2);
      UP(vec_of(p2b, d), W.pw2_b);
    }
  }
  if (dsc.reduce_idx >= 0) {
    // TimeReductionLayerStream: depthwise kernel 1; TimeReductionLayer1D (non-streaming model): kernel 5
    int tr_k = 1;
    const float* rdw = get("encoder.time_reduction_layer.dw_conv.weight", d);
    if (!rdw) {
      get.missing.clear();
      rdw = get("encoder.time_reduction_layer.dw_conv.weight", (size_t)d * 5);
      tr_k = 5;
    }
    if (!rdw) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + get.missing);
    GETW(rdb, "encoder.time_reduction_layer.dw_conv.bias", d);
    GETW(rpw, "encoder.time_reduction_layer.pw_conv.weight", d * d);
    GETW(rpb, "encoder.time_reduction_layer.pw_conv.bias", d);
    GETW(rw, "encoder.time_recover_layer.weight", d * d);
    GETW(rb, "encoder.time_recover_layer.bias", d);
    {
      std::vector<float> taps((size_t)tr_k * d);  // [C][1][k] -> tap-major [k][C]
      for (int c = 0; c < d; ++c)
        for (int k = 0; k < tr_k; ++k) taps[(size_t)k * d + c] = rdw[(size_t)c * tr_k + k];
      UP(taps, m->sq_reduce.dw_w);
      m->sq_reduce.ks = tr_k;
    }
    UP(vec_of(rdb, d), m->sq_reduce.dw_b);
    UP4(pack_b(d, d, [&](int k, int n) { return rpw[(size_t)n * d + k]; }), m->sq_reduce.pw);
    UP(vec_of(rpb, d), m->sq_reduce.pw_b);
    UP4(pack_b(d, d, [&](int k, int n) { return rw[(size_t)k * d + n]; }), m->sq_wrec);
    UP(vec_of(rb, d), m->sq_brec);
  }
  {
    // final_proj (output_size != encoder_dim, encoder.py:165-167,234-235): a Linear between the last layer and ctc_lo with
    // nothing in between -- folded into the head at create time, logits = x (W_fp W_ctc) + (b_fp W_ctc + b_ctc), in double
    std::vector<float> cw_f, cb_f;
    const float *cw = nullptr, *cb = nullptr;
    auto fp = sd.find("encoder.final_proj.weight");
    if (fp != sd.end()) {
      const Blob& fb = fp->second;
      if (fb.ndim != 2 || fb.shape[0] != d) return fail(PPASR_EMISSING, "mis-shaped weight: encoder.final_proj.weight");
      const int O = (int)fb.shape[1];
      const float* fw = fb.p;
      GETW(fbias, "encoder.final_proj.bias", O);
      GETW(cw_o, "ctc.ctc_lo.weight", (size_t)O * V);
      GETW(cb_o, "ctc.ctc_lo.bias", V);
      cw_f.resize((size_t)d * V);
      cb_f.resize(V);
      std::vector<double> acc(V);
      for (int k = 0; k < d; ++k) {
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int o = 0; o < O; ++o) {
          const double f = fw[(size_t)k * O + o];
          const float* row = cw_o + (size_t)o * V;
          for (int n = 0; n < V; ++n) acc[n] += f * (double)row[n];
        }
        for (int n = 0; n < V; ++n) cw_f[(size_t)k * V + n] = (float)acc[n];
      }
      for (int n = 0; n < V; ++n) acc[n] = (double)cb_o[n];
      for (int o = 0; o < O; ++o)
        for (int n = 0; n < V; ++n) acc[n] += (double)fbias[o] * (double)cw_o[(size_t)o * V + n];
      for (int n = 0; n < V; ++n) cb_f[n] = (float)acc[n];
      cw = cw_f.data();
      cb = cb_f.data();
    } else {
      GETW(cw_d, "ctc.ctc_lo.weight", (size_t)d * V);
      GETW(cb_d, "ctc.ctc_lo.bias", V);
      cw = cw_d;
      cb = cb_d;
    }
    m->head.ln_g = nullptr;  // no after_norm in Squeezeformer
    m->head.ln_b = nullptr;
    m->head.V = V;
    m->head.n_tiles = (V + 31) / 32;
    UP4(pack_b(d, V, [&](int k, int n) { return cw[(size_t)k * V + n]; }), m->head.w);
    std::vector<float> cbp((size_t)m->head.n_tiles * 32, 0.f);
    std::memcpy(cbp.data(), cb, V * sizeof(float));
    UP(cbp, m->head.b);
    if (m->generic) {  // general route (encoder_dim 512 ..): the head as a plain dense layer over the padded vocabulary
      m->gen_vpad = (V + 255) / 256 * 256;
      const int Vp = m->gen_vpad;
      UP4(pack_b(d, Vp, [&](int k, int n) { return n < V ? cw[(size_t)k * V + n] : 0.f; }), m->gen_head_w);
      std::vector<float> hb(Vp, 0.f);
      std::memcpy(hb.data(), cb, V * sizeof(float));
      UP(hb, m->gen_head_b);
    }
  }
  return PPASR_OK;
}

// the conv-module fields of a Squeezeformer layer as the LayerW view k_conv_pre reads
LayerW sq_conv_view(const SqLayerW& W) {
  LayerW v{};
  v.dw_w = W.dw_w; v.dw_b = W.dw_b; v.glu_pad = W.glu_pad;
  v.ln_cm_g = W.ln_cm_g; v.ln_cm_b = W.ln_cm_b; v.cm_eps = W.cm_eps;
  v.pw2 = W.pw2; v.pw2_b = W.pw2_b;
  return v;
}

// SqueezeformerEncoder.forward (squeezeformer/encoder.py:172-236) + ctc softmax
ppasr_status squeezeformer_encode(ppasr_model_s* h, const float* feats, const int64_t* lens, int B, int T, float* probs,
                                  float* logits, int32_t* frame_argmax, float* frame_maxprob, float* ws,
                                  const WsLayout& wl, hipStream_t st) {
  const int F = h->desc.input_dim, T1 = (T - 1) / 2, F1 = h->F1, Tp = (T1 - 1) / 2, F2 = h->F2;
  const int Tr = (Tp + 1) / 2;  // Conv1D(k=1, stride 2): ceil(T'/2) frames (time_reduction.py:186,196-199)
  const int M = B * Tp, L = h->desc.num_blocks, H = h->desc.attention_heads;
  const int n_chunks = h->desc.linear_units / 256, KS = h->desc.cnn_module_kernel;
  float *y1 = ws + wl.y1, *y2 = ws + wl.y2, *xa = ws + wl.xa, *xb = ws + wl.xb, *xc = ws + wl.xc;
  float *qkv = ws + wl.qkv, *ctx = ws + wl.ctx, *g = ws + wl.g, *xs = ws + wl.xs;
  size_t tap_off = 0;
  auto tap = [&](const float* src, size_t n) {
    if (h->taps && tap_off + n <= h->taps_floats)
      (void)hipMemcpyAsync(h->taps + tap_off, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    tap_off += n;
  };
  // ragged batches (ppasr_set_skip_padding, see ppasr_encode): the slack covers the time-reduction layer (reduced row
  // j reads full-rate rows 2j - 3 .. 2j + 1 at most), the recovery (row t reads reduced row t/2) and, for the
  // non-streaming model, the right context of the non-causal conv module
  const bool skip = h->skip_padding && lens && !h->taps;
  const bool causal = h->desc.causal != 0;
  const int rc = causal ? 0 : (KS - 1) / 2;
  auto pskip = [&](int Tcur, int mul_cur) {
    PadSkip ps;
    if (skip) {
      ps.lens = lens;
      ps.Tp = Tcur;
      ps.mul = mul_cur;
      ps.slack = mul_cur == 4 ? 2 * (rc + 4) + rc + 8 : rc + 4;
    }
    return ps;
  };
  const PadSkip psF = pskip(Tp, 4), psH = pskip(Tr, 8);
  int* tile_tab = (size_t)B + 2 <= ((size_t)M + 63) / 64 * 64 ? reinterpret_cast<int*>(ws + wl.rmax) : nullptr;
  // ragged batch: the active row blocks of the two frame rates as lists (PadSkip::tab), for the layer kernels K_B / K_C --
  // with the beam search of the previous batch on some CUs a padded grid with early exits runs extra rounds (rowblock.h).
  // The lists live behind conv2's tile table in the CTC head's statistics buffers (2 al(M) floats, unused until the head).
  const int rowsF = h->taps ? 32 : row_block_for(h, B, Tp, 4, psF.slack, skip);
  const int rowsH = h->taps ? 32 : row_block_for(h, B, Tr, 8, psH.slack, skip);
  PadSkip psF_rb = psF, psH_rb = psH;
  if (skip && tile_tab && block_tables_enabled()) {
    const int RF = rowsF == 16 ? 16 : 32, RH = rowsH == 16 ? 16 : 32;
    const size_t nF = 1 + ((size_t)M + RF - 1) / RF, nH = 1 + ((size_t)B * Tr + RH - 1) / RH;
    const size_t o1 = ((size_t)B + 2 + 15) / 16 * 16, o2 = o1 + (nF + 15) / 16 * 16;
    if (o2 + nH <= 2 * (((size_t)M + 63) / 64 * 64)) {
      int* base = reinterpret_cast<int*>(ws + wl.rmax);
      launch_block_table(psF, M, RF, base + o1, st);
      launch_block_table(psH, B * Tr, RH, base + o2, st);
      psF_rb.tab = base + o1;
      psH_rb.tab = base + o2;
    }
  }
  // (fp16 x3 mode, ppasr_set_gemm_mode: conv2 on that route as its own launch behind k_conv1)
  const f32x4* conv2_h3 = h->gemm_mode == PPASR_GEMM_F16X3 ? h->conv2_w_h3 : nullptr;
  if (!conv2_h3 && conv12_enabled(h) && conv12_supported(h->front, F, F2)) {  // both convolutions in one launch (front_fused.hip)
    launch_conv12(feats, h->front, y2, B, T, F, Tp, F2, st, psF, tile_tab);
  } else {
    launch_conv1(feats, h->front, y1, B, T, F, T1, F1, st, psF);
    launch_conv2(y1, h->front, y2, B, T1, F1, Tp, F2, st, psF, tile_tab, conv2_h3);
  }
  // (the embed GEMM works on 32-row blocks: it takes the full-rate list when that is the 32-row one)
  launch_embed(y2, h->front, xa, M, F2 * kD, sqrtf((float)kD), /*scale_before_bias=*/true, st,
               (rowsF != 16 && ffn_split_for(h, M) == 1) ? psF_rb : psF, ffn_split_for(h, M), y1,
               conv2_h3 ? h->embed_w_h3 : nullptr);
  launch_ln_rows(xa, h->preln_g, h->preln_b, M, st, psF);
  tap(xa, (size_t)M * kD);
  float* x = xa;      // current layer input / residual
  float* other = xb;  // ping-pong partner
  bool reduced = false;
  bool have_qkv = false;
  for (int i = 0; i < L; ++i) {
    const SqLayerW& W = h->sq_layers[i];
    if (i == h->desc.reduce_idx) {
      // recover_activations.append(xs) ; time_reduction_layer ; pos_emb[:, ::2]  (encoder.py:210-216)
      HIP_TRY(hipMemcpyAsync(xs, x, (size_t)M * kD * sizeof(float), hipMemcpyDeviceToDevice, st));
      launch_sq_reduce(x, other, qkv, h->sq_reduce, W.wqkv, W.bqkv, lens, B, Tp, Tr, st, psH);
      std::swap(x, other);
      reduced = true;
      have_qkv = true;
    }
    if (i == h->desc.recover_idx && reduced) {
      launch_sq_recover(x, xs, other, qkv, h->sq_wrec, h->sq_brec, W.wqkv, W.bqkv, B, Tp, Tr, st, psF);
      std::swap(x, other);
      reduced = false;
      have_qkv = true;
    }
    const int Ti = reduced ? Tr : Tp;
    const int Mi = B * Ti;
    const int mul = reduced ? 8 : 4;
    const PadSkip& ps = reduced ? psH : psF;
    // under-filled launches up to split_rows16_max() rows (one utterance, small batches; fp32): the single-unit launches on
    // the Conformer's 16-row kernels through weight views, as the streaming chunk does (capi_stream.hip)
    const bool views16 = ffn_split_for(h, Mi) > 1 && !(rowsF == 16 && h->ffn_split < 0) && Mi <= split_rows16_max() &&
                         !(h->gemm_mode == PPASR_GEMM_F16X3 && !h->sq_layers_h3.empty());
    auto qkv_view = [](const SqLayerW& w) {
      LayerW v{};
      v.wqkv = w.wqkv;
      v.bqkv = w.bqkv;
      return v;
    };
    if (!have_qkv) {
      if (views16) launch_ln_qkv(x, qkv, qkv_view(W), Mi, st, ps, nullptr, nullptr, false);
      else launch_sq_qkv(x, qkv, W.wqkv, W.bqkv, Mi, st, ps);
    }
    tap(qkv, (size_t)Mi * 3 * kD);
    AttnArgs a{qkv, 768, qkv + 256, 768, qkv + 512, 768, Ti, Ti, 0, lens, ctx, W.pos_u, W.pos_v, W.ptab, reduced ? 2 : 1, mul, Ti, Ti, 1};
    a.pad_skip = skip ? ps.slack + 1 : 0;
    launch_attention(a, B, H, st);
    tap(ctx, (size_t)Mi * kD);
    // under-filled grid (ppasr_set_ffn_split): K_B / K_C cut at their feed-forward modules, partial sums in the conv1 buffer
    // under-filled launch: 16-row blocks (twice the workgroups, each half as long) before the split route
    // ... and full launches: the 32-row blocks on 16 waves (rbt.h kW16)
    const int rows = reduced ? rowsH : rowsF;
    const PadSkip& ps_rb = reduced ? psH_rb : psF_rb;  // (with the list of active blocks of THIS block size)
    const int S = rows == 16 && h->ffn_split < 0 ? 1 : ffn_split_for(h, Mi);  // (an explicit ffn_split mode wins)
    const bool fuse_next = (i + 1 < L) && (i + 1 != h->desc.reduce_idx) && !(i + 1 == h->desc.recover_idx && reduced);
    const SqLayerW* Wn = fuse_next ? &h->sq_layers[i + 1] : nullptr;
    if (S > 1) {
      // (fp16 x3 mode: the two feed-forward modules' slices on that route -- the re-packed weights of the layer's h3 view)
      const bool h3s = h->gemm_mode == PPASR_GEMM_F16X3 && !h->sq_layers_h3.empty();
      const SqLayerW& Ws = h3s ? h->sq_layers_h3[i] : W;
      // x1 = LN1(x + MHA) in `other` (free until this layer's output)
      if (views16) {
        LayerW vo{};
        vo.wo = W.wo; vo.bo = W.bo; vo.ln_conv_g = W.ln1_g; vo.ln_conv_b = W.ln1_b;
        launch_oproj_ln_16(ctx, x, g, other, vo, Mi, st, ps);  // (the plain sum goes to g, dead until pointwise_conv1 writes it)
      } else {
        launch_sq_oproj(ctx, x, other, W, Mi, st, ps);
      }
      launch_ffn_split(other, nullptr, nullptr, Ws.ff1_w1, W.ff1_b1, Ws.ff1_w2, W.ff1_b2, 1.0f, W.ln2_g, W.ln2_b, y1, xc, Mi,
                       n_chunks, S, st, ps, false, h3s);
      if (views16) {
        LayerW vp{};
        vp.pw1 = W.pw1; vp.pw1_b = W.pw1_b; vp.glu_pad = W.glu_pad;
        launch_pw1_glu_cols_16(xc, g, vp, Mi, st, nullptr, 0, nullptr, nullptr, ps, lens, Ti, mul);
      } else {
        launch_sq_pw1glu(xc, g, nullptr, W, lens, Mi, Ti, mul, st, ps);
      }
      tap(xc, (size_t)Mi * kD);
      tap(g, (size_t)Mi * kD);
      launch_conv_pre(g, nullptr, xc, ctx, sq_conv_view(W), lens, Mi, Ti, KS, mul, st, causal, ps);
      launch_ffn_split(ctx, W.ln3_g, W.ln3_b, Ws.ff2_w1, W.ff2_b1, Ws.ff2_w2, W.ff2_b2, 1.0f, W.ln4_g, W.ln4_b, y1, other, Mi,
                       n_chunks, S, st, ps, /*residual_is_normed=*/true, h3s);
      if (Wn && views16) launch_ln_qkv(other, qkv, qkv_view(*Wn), Mi, st, ps, nullptr, nullptr, false);
      else if (Wn) launch_sq_qkv(other, qkv, Wn->wqkv, Wn->bqkv, Mi, st, ps);
    } else {
      // feed-forward modules on the fp16 x3 route (ppasr_set_gemm_mode): the 8-wave 32-row kernels only
      const bool h3 = h->gemm_mode == PPASR_GEMM_F16X3 && !h->sq_layers_h3.empty() && rows == 32 && !h->taps &&
                      sq_h3_supported(KS, Ti);
      const SqLayerW& Wk = h3 ? h->sq_layers_h3[i] : W;
      launch_sq_mid(ctx, x, xc, g, nullptr, Wk, lens, Mi, Ti, mul, n_chunks, st, ps_rb, rows, h3);
      tap(xc, (size_t)Mi * kD);
      tap(g, (size_t)Mi * kD);
      launch_sq_tail(g, nullptr, xc, other, qkv, Wk, Wn ? Wn->wqkv : nullptr, Wn ? Wn->bqkv : nullptr, lens, Mi, Ti, mul,
                     n_chunks, KS, st, ps_rb, causal, rows, h3);
    }
    std::swap(x, other);
    have_qkv = fuse_next;
    tap(x, (size_t)Mi * kD);
  }
  float* lg = logits ? logits : probs;
  int32_t* fa = frame_argmax ? frame_argmax : reinterpret_cast<int32_t*>(ws + wl.fa);
  float* fp = frame_maxprob ? frame_maxprob : ws + wl.fp;
  // (the encoder ends at the full rate after the recovery; without one it stays reduced and M rows = B * Tp is the
  //  caller's contract either way)
  const PadSkip psO = reduced ? PadSkip{} : psF;
  const bool head_h3 = h->gemm_mode == PPASR_GEMM_F16X3 && h->head_w_h3;
  HeadW hw = h->head;
  if (head_h3) hw.w = h->head_w_h3;
  launch_ctc_head(x, hw, lg, fa, fp, ws + wl.rmax, ws + wl.rsum, M, st, psO, ffn_split_for(h, M), y1, head_h3);
  if (probs) {
    if (logits)
      HIP_TRY(hipMemcpyAsync(probs, logits, (size_t)M * h->head.V * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_softmax_from_stats(probs, ws + wl.rmax, ws + wl.rsum, M, h->head.V, st, psO);
  }
  if (skip && !reduced) launch_zero_pad_rows(probs, logits, fa, fp, lens, B, Tp, 4, h->head.V, st);
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}
