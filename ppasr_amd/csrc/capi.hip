// capi.hip -- the C-ABI of libppasr_hip.so (declared in include/ppasr_hip.h).
// Host side: weight re-packing into MFMA fragment order, workspace carving, launch sequence.
#include <cxxabi.h>
#include <cstdlib>

#include "capi_internal.h"

namespace ppasr {
thread_local LaunchProf g_launch_prof;  // launch.h
}

static thread_local std::string g_err;
std::string& ppasr_err_slot() { return g_err; }


ppasr_status upload_pe_table(ppasr_model_s* m, BlobMap& sd, const float** pe_dev) {
  const int d = m->desc.output_size > 0 ? m->desc.output_size : kD;
  const int max_len = m->desc.max_len > 0 ? m->desc.max_len : 5000;
  m->desc.max_len = max_len;
  std::vector<float> pe((size_t)max_len * d);
  auto it = sd.find("__pe_table__");
  if (it != sd.end() && it->second.numel() == pe.size()) {
    std::memcpy(pe.data(), it->second.p, pe.size() * sizeof(float));
  } else {  // PositionalEncoding.__init__ (embedding.py:38-53)
    for (int i = 0; i < d / 2; ++i) {
      float div = expf((float)(2 * i) * (float)(-(std::log(10000.0) / d)));
      for (int pos = 0; pos < max_len; ++pos) {
        float a = (float)pos * div;
        pe[(size_t)pos * d + 2 * i] = sinf(a);
        pe[(size_t)pos * d + 2 * i + 1] = cosf(a);
      }
    }
  }
  return m->upload(pe, pe_dev);
}

extern "C" {

const char* ppasr_last_error(void) { return g_err.c_str(); }
const char* ppasr_version(void) { return "ppasr_hip 0.1 (gfx950, fp32 MFMA)"; }

ppasr_status ppasr_create(const ppasr_model_desc* desc, const ppasr_weight_blob* blobs, int n_blobs, ppasr_handle* out) {
  if (!desc || !blobs || !out) return fail(PPASR_EINVAL, "null argument");
  if (desc->model_type == PPASR_MODEL_DEEPSPEECH2) {
    if (desc->input_dim > 128 || desc->input_dim < 7) return fail(PPASR_EUNSUPPORTED, "input_dim out of range");
    HIP_TRY(configure_kernels());
    BlobMap sd2;
    for (int i = 0; i < n_blobs; ++i) {
      Blob b{blobs[i].data_host, blobs[i].ndim, {0, 0, 0, 0}};
      for (int j = 0; j < blobs[i].ndim && j < 4; ++j) b.shape[j] = blobs[i].shape[j];
      sd2[blobs[i].name] = b;
    }
    std::unique_ptr<ppasr_model_s> g2(new ppasr_model_s());
    g2->desc = *desc;
    g2->F1 = (desc->input_dim - 1) / 2;
    g2->F2 = (g2->F1 - 1) / 2;
    if (g2->F1 > 40) return fail(PPASR_EUNSUPPORTED, "deepspeech2: input_dim <= 81");
    ppasr_status s2 = ds2_create(g2.get(), sd2);
    if (s2 != PPASR_OK) return s2;
    HIP_TRY(hipDeviceSynchronize());
    *out = g2.release();
    return PPASR_OK;
  }
  if (desc->model_type != PPASR_MODEL_CONFORMER && desc->model_type != PPASR_MODEL_SQUEEZEFORMER &&
      desc->model_type != PPASR_MODEL_EFFICIENT_CONFORMER)
    return fail(PPASR_EUNSUPPORTED, "model_type not built (conformer, efficient_conformer, squeezeformer are)");
  const bool eff = desc->model_type == PPASR_MODEL_EFFICIENT_CONFORMER;
  const unsigned smask = eff_stride_mask(*desc);
  const bool multi_stride = __builtin_popcount(smask) > 1;
  if (eff) {
    if (desc->group_layer_mask != 0 && (desc->group_size < 2 || desc->group_size > 4))
      return fail(PPASR_EUNSUPPORTED, "efficient_conformer: grouped attention is built for group_size 2, 3 and 4");
    if (desc->num_blocks > 31 || (smask >> desc->num_blocks) != 0) return fail(PPASR_EINVAL, "stride layer index out of range");
    if (smask != 0 && !multi_stride && desc->cnn_module_kernel != 15 && desc->output_size == kD)
      return fail(PPASR_EUNSUPPORTED, "efficient_conformer: cnn_module_kernel must be 15 (7 after the stride layer)");
    if ((desc->cnn_module_kernel >> __builtin_popcount(smask)) < 1)
      return fail(PPASR_EINVAL, "efficient_conformer: cnn_module_kernel halves to 0 behind the stride layers");
  }
  // output_size 256 (4 heads of 64) with the shipped constructor arguments: the fused row-block kernels.  Other multiples
  // of 256 up to 1024, non-default ConformerEncoder options (ppasr_model_desc::options), input_layer = linear or another
  // conv kernel size: the general layer route of capi_generic.hip (model_type = conformer).
  if (desc->output_size % 256 != 0 || desc->output_size < 256 || desc->output_size > 1024)
    return fail(PPASR_EUNSUPPORTED, "output_size must be 256, 512, 768 or 1024");
  if (desc->attention_heads * 64 != desc->output_size) return fail(PPASR_EUNSUPPORTED, "kernels are specialised for d_k=64");
  const bool fused_ks = desc->cnn_module_kernel == 15 || desc->cnn_module_kernel == 31 || desc->cnn_module_kernel == 7;
  const bool generic = (desc->model_type == PPASR_MODEL_CONFORMER &&
                        (desc->output_size != kD || desc->options != 0 || desc->input_layer == 1 || !fused_ks)) ||
                       ((desc->model_type == PPASR_MODEL_SQUEEZEFORMER || desc->model_type == PPASR_MODEL_EFFICIENT_CONFORMER) &&
                        desc->output_size != kD) ||
                       (eff && multi_stride) ||  // (several stride layers: kernels 15 -> 7 -> 3 ..., the general route)
                       (desc->model_type == PPASR_MODEL_SQUEEZEFORMER &&
                        (((desc->options >> PPASR_OPT_ACT_SHIFT) & PPASR_OPT_ACT_MASK) != PPASR_ACT_SWISH ||
                         (desc->options & PPASR_OPT_SQ_PRE_NORM) != 0 ||
                         (desc->options & PPASR_OPT_POS_MASK) != PPASR_OPT_POS_REL));  // (activation_type, normalize_before = True,
                                                                                       //  pos_enc_layer_type != rel_pos)
  // Squeezeformer takes two of the option fields: adaptive_scale = False and activation_type (squeezeformer/encoder.py:44-45)
  const int sq_opts = desc->model_type == PPASR_MODEL_SQUEEZEFORMER
                          ? (PPASR_OPT_SQ_NO_ADAPTIVE_SCALE | PPASR_OPT_SQ_PRE_NORM | PPASR_OPT_POS_MASK |
                             (PPASR_OPT_ACT_MASK << PPASR_OPT_ACT_SHIFT)) : 0;
  if (((desc->options & ~sq_opts) != 0 || desc->input_layer == 1) && desc->model_type != PPASR_MODEL_CONFORMER)
    return fail(PPASR_EUNSUPPORTED, "non-default encoder options / input_layer=linear are built for model_type=conformer");
  if ((desc->options & (PPASR_OPT_SQ_NO_ADAPTIVE_SCALE | PPASR_OPT_SQ_PRE_NORM)) && desc->model_type != PPASR_MODEL_SQUEEZEFORMER)
    return fail(PPASR_EINVAL, "PPASR_OPT_SQ_NO_ADAPTIVE_SCALE / PPASR_OPT_SQ_PRE_NORM are Squeezeformer options");
  if (desc->linear_units % 256 != 0 || desc->linear_units <= 0) return fail(PPASR_EUNSUPPORTED, "linear_units % 256 != 0");
  if (!generic && !fused_ks) return fail(PPASR_EUNSUPPORTED, "cnn_module_kernel must be 7, 15 or 31");
  if (generic) {
    const int ks = desc->cnn_module_kernel;
    const bool use_cnn = !(desc->options & PPASR_OPT_NO_CNN);
    if (use_cnn && (ks < 1 || ks > 63 || (!desc->causal && ks % 2 == 0)))
      return fail(PPASR_EINVAL, "cnn_module_kernel: 1..63, odd for the non-causal conv module (convolution.py:38)");
    if ((desc->options & PPASR_OPT_POS_MASK) == 3 || ((desc->options >> PPASR_OPT_ACT_SHIFT) & PPASR_OPT_ACT_MASK) > PPASR_ACT_HARDSHRINK)
      return fail(PPASR_EINVAL, "options: unknown pos_enc_layer_type / activation_type code");
  }
  if (desc->model_type == PPASR_MODEL_SQUEEZEFORMER && desc->cnn_module_kernel == 7)
    return fail(PPASR_EUNSUPPORTED, "squeezeformer: cnn_module_kernel must be 15 or 31");
  if (desc->input_dim > 128 || (desc->input_dim < 7 && desc->input_layer != 1) || desc->input_dim < 1)
    return fail(PPASR_EUNSUPPORTED, "input_dim out of range");
  HIP_TRY(configure_kernels());
  HIP_TRY(configure_generic_kernels());
  HIP_TRY(configure_squeezeformer_kernels());

  BlobMap sd;
  for (int i = 0; i < n_blobs; ++i) {
    Blob b{blobs[i].data_host, blobs[i].ndim, {0, 0, 0, 0}};
    for (int j = 0; j < blobs[i].ndim && j < 4; ++j) b.shape[j] = blobs[i].shape[j];
    sd[blobs[i].name] = b;
  }
  std::string missing;
  auto get = [&](const std::string& name, size_t numel) -> const float* {
    auto it = sd.find(name);
    if (it == sd.end() || it->second.numel() != numel) {
      missing = name;
      return nullptr;
    }
    return it->second.p;
  };
#define GET(var, name, numel)                         \
  const float* var = get(name, (size_t)(numel));      \
  if (!var) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + missing)

  auto* m = new ppasr_model_s();
  std::unique_ptr<ppasr_model_s> guard(m);
  m->desc = *desc;
  if (eff) {  // one representation inside: the mask, and stride_layer_idx = its first (for the shipped shape: only) layer
    m->desc.stride_layer_mask = (int)smask;
    m->desc.stride_layer_idx = smask ? __builtin_ctz(smask) : -1;
  }
  const int F = desc->input_dim, d = desc->output_size, H = desc->linear_units, V = desc->vocab_size, KS = desc->cnn_module_kernel;
  const int il = desc->input_layer;
  if (il != 0 && il != 1 && il != 6 && il != 8)
    return fail(PPASR_EINVAL, "input_layer: 0 (conv2d), 1 (linear), 6 (conv2d6) or 8 (conv2d8)");
  m->generic = generic;
  m->gen.pos = desc->options & PPASR_OPT_POS_MASK;
  m->gen.post_norm = (desc->options & PPASR_OPT_POST_NORM) != 0;
  m->gen.concat_after = (desc->options & PPASR_OPT_CONCAT_AFTER) != 0;
  m->gen.macaron = !(desc->options & PPASR_OPT_NO_MACARON);
  m->gen.use_cnn = !(desc->options & PPASR_OPT_NO_CNN);
  m->gen.act = (desc->options >> PPASR_OPT_ACT_SHIFT) & PPASR_OPT_ACT_MASK;
  m->gen.sq_pre_norm = (desc->options & PPASR_OPT_SQ_PRE_NORM) != 0;
  const auto& go = m->gen;
  if (il != 0 && desc->model_type == PPASR_MODEL_SQUEEZEFORMER)
    return fail(PPASR_EUNSUPPORTED, "squeezeformer: only the conv2d front end is built");
  m->F1 = (F - 1) / 2;
  m->F2 = il == 6 ? (m->F1 - 5) / 3 + 1 : (m->F1 - 1) / 2;
  m->F3 = il == 8 ? (m->F2 - 1) / 2 : 0;
  if (il != 1 && m->F_last() < 1) return fail(PPASR_EINVAL, "input_dim too small for this input_layer");
  const int F2 = m->F_last();  // feature bins entering the linear layer
  ppasr_status st;
#define UP(vec, dst) \
  if ((st = m->upload(vec, &(dst))) != PPASR_OK) return st
#define UP4(vec, dst) \
  if ((st = m->upload4(vec, &(dst))) != PPASR_OK) return st
  auto vec_of = [](const float* p, size_t n) { return std::vector<float>(p, p + n); };

  if (desc->model_type == PPASR_MODEL_SQUEEZEFORMER) {
    const float* pe_sq = nullptr;
    if ((st = upload_pe_table(m, sd, &pe_sq)) != PPASR_OK) return st;
    if ((st = squeezeformer_create(m, sd, pe_sq)) != PPASR_OK) return st;
    HIP_TRY(hipDeviceSynchronize());
    *out = guard.release();
    return PPASR_OK;
  }
  if (il == 1) {  // ---- LinearNoSubsampling (subsampling.py:24-65): out.0 = Linear(idim, odim), out.1 = LayerNorm(eps 1e-12), ReLU ----
    GET(mean, "encoder.global_cmvn.mean", F);
    GET(istd, "encoder.global_cmvn.istd", F);
    GET(ew, "encoder.embed.out.0.weight", (size_t)F * d);
    GET(eb, "encoder.embed.out.0.bias", d);
    GET(lg, "encoder.embed.out.1.weight", d);
    GET(lb, "encoder.embed.out.1.bias", d);
    UP(vec_of(mean, F), m->front.cmvn_mean);
    UP(vec_of(istd, F), m->front.cmvn_istd);
    m->lin_kpad = (F + 255) / 256 * 256;  // the feature rows are zero-padded to whole K chunks (k_g_cmvn_pad)
    UP4(pack_b(m->lin_kpad, d, [&](int k, int n) { return k < F ? ew[(size_t)k * d + n] : 0.f; }), m->front.embed_w);
    UP(vec_of(eb, d), m->front.embed_b);
    UP(vec_of(lg, d), m->lin_ln_g);
    UP(vec_of(lb, d), m->lin_ln_b);
    m->front.conv1_w = m->front.conv1_b = m->front.conv2_b = nullptr;
    m->front.conv2_w = nullptr;
  } else {  // ---- conv front ends ----
    GET(mean, "encoder.global_cmvn.mean", F);
    GET(istd, "encoder.global_cmvn.istd", F);
    GET(c1w, "encoder.embed.conv.0.weight", d * 9);
    GET(c1b, "encoder.embed.conv.0.bias", d);
    const int k2 = il == 6 ? 5 : 3;  // Conv2dSubsampling6: Conv2D(odim, odim, 5, 3) (subsampling.py:139-141)
    GET(c2w, "encoder.embed.conv.2.weight", (size_t)d * d * k2 * k2);
    GET(c2b, "encoder.embed.conv.2.bias", d);
    // Conv2dSubsampling4 names its projection `out` (a Sequential), the 6x / 8x classes `linear` (subsampling.py:142,189)
    const std::string lin = il ? "encoder.embed.linear" : "encoder.embed.out.0";
    GET(ew, lin + ".weight", (size_t)d * F2 * d);
    GET(eb, lin + ".bias", d);
    UP(vec_of(mean, F), m->front.cmvn_mean);
    UP(vec_of(istd, F), m->front.cmvn_istd);
    std::vector<float> c1(9 * d);
    for (int c = 0; c < d; ++c)
      for (int j = 0; j < 9; ++j) c1[j * d + c] = c1w[c * 9 + j];
    UP(c1, m->front.conv1_w);
    UP(vec_of(c1b, d), m->front.conv1_b);
    // conv2: K index = (kh*k+kw)*256 + cin ; weight layout [cout][cin][kh][kw]
    const int taps2 = k2 * k2;
    UP4(pack_b(taps2 * d, d, [&](int k, int n) { return c2w[((size_t)n * d + (k % d)) * taps2 + (k / d)]; }), m->front.conv2_w);
    UP(vec_of(c2b, d), m->front.conv2_b);
    m->front.conv2_k = k2;
    m->front.conv2_s = il == 6 ? 3 : 2;
    m->front.conv3_w = nullptr;
    m->front.conv3_b = nullptr;
    if (il == 8) {  // third Conv2D(odim, odim, 3, 2) (subsampling.py:183-187)
      GET(c3w, "encoder.embed.conv.4.weight", (size_t)d * d * 9);
      GET(c3b, "encoder.embed.conv.4.bias", d);
      UP4(pack_b(9 * d, d, [&](int k, int n) { return c3w[((size_t)n * d + (k % d)) * 9 + (k / d)]; }), m->front.conv3_w);
      UP(vec_of(c3b, d), m->front.conv3_b);
    }
    // embed: our K index = f*256 + c ; Paddle's = c*F2 + f (subsampling.py:113 transpose+reshape)
    UP4(pack_b(F2 * d, d, [&](int k, int n) { return ew[((size_t)(k % d) * F2 + (k / d)) * d + n]; }), m->front.embed_w);
    UP(vec_of(eb, d), m->front.embed_b);
  }

  const float* pe_dev = nullptr;
  if ((st = upload_pe_table(m, sd, &pe_dev)) != PPASR_OK) return st;
  m->pe_dev = pe_dev;
  UP(std::vector<float>(d, 0.f), m->zero_vec);
  if (generic) m->gen_x.resize(desc->num_blocks);
  const int max_len = m->desc.max_len;
  m->layers.resize(desc->num_blocks);
  m->layer_ks.assign(desc->num_blocks, KS);
  m->layer_group.assign(desc->num_blocks, 1);
  for (int i = 0; i < desc->num_blocks; ++i) {
    LayerW& L = m->layers[i];
    // Efficient-Conformer: kernel halves after the stride layer (encoder.py:123-128), grouped attention layers
    const int KSi = eff ? (KS >> eff_strides_before(*desc, i)) : KS;  // (cnn_module_kernels: // 2 per stride layer passed)
    const bool grouped = eff && ((desc->group_layer_mask >> i) & 1);
    m->layer_ks[i] = KSi;
    m->layer_group[i] = grouped ? desc->group_size : 1;
    const int pbn = grouped ? desc->group_size * d : d;  // pos_bias_u/v are [h][dk*group_size] on grouped layers
    const std::string p = "encoder.encoders." + std::to_string(i) + ".";
    auto ln = [&](const std::string& n, const float** g, const float** b) -> ppasr_status {
      const float* gw = get(p + n + ".weight", d);
      const float* gb = get(p + n + ".bias", d);
      if (!gw || !gb) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + missing);
      ppasr_status s1 = m->upload(vec_of(gw, d), g);
      if (s1 != PPASR_OK) return s1;
      return m->upload(vec_of(gb, d), b);
    };
    L = LayerW{};
    // (encoder.py:327-336: norm_ff_macaron exists with the macaron half only, norm_conv / norm_final with the conv module only)
    if (go.macaron && (st = ln("norm_ff_macaron", &L.ln_mac_g, &L.ln_mac_b)) != PPASR_OK) return st;
    if ((st = ln("norm_mha", &L.ln_mha_g, &L.ln_mha_b)) != PPASR_OK) return st;
    if (go.use_cnn && (st = ln("norm_conv", &L.ln_conv_g, &L.ln_conv_b)) != PPASR_OK) return st;
    if ((st = ln("norm_ff", &L.ln_ff_g, &L.ln_ff_b)) != PPASR_OK) return st;
    if (go.use_cnn && (st = ln("norm_final", &L.ln_fin_g, &L.ln_fin_b)) != PPASR_OK) return st;
    // ConvolutionModule.norm (convolution.py:65-71): nn.LayerNorm, or nn.BatchNorm1D (cnn_module_norm: batch_norm), which
    // at inference is the per-channel affine y = (x - _mean) / sqrt(_variance + 1e-5) * weight + bias: folded here into
    // scale / shift vectors in the LayerNorm slots, marked by cm_eps < 0 (ln_rows_inreg then skips the row statistics)
    L.cm_eps = 1e-5f;
    if (!go.use_cnn) {
    } else if (sd.find(p + "conv_module.norm._mean") != sd.end()) {
      const float* mean = get(p + "conv_module.norm._mean", d);
      const float* var = get(p + "conv_module.norm._variance", d);
      const float* gw = get(p + "conv_module.norm.weight", d);
      const float* gb = get(p + "conv_module.norm.bias", d);
      if (!mean || !var || !gw || !gb) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + missing);
      std::vector<float> sc(d), sh(d);
      for (int c = 0; c < d; ++c) {
        sc[c] = gw[c] / std::sqrt(var[c] + 1e-5f);
        sh[c] = gb[c] - mean[c] * sc[c];
      }
      if ((st = m->upload(sc, &L.ln_cm_g)) != PPASR_OK || (st = m->upload(sh, &L.ln_cm_b)) != PPASR_OK) return st;
      L.cm_eps = -1.f;
    } else if ((st = ln("conv_module.norm", &L.ln_cm_g, &L.ln_cm_b)) != PPASR_OK) {
      return st;
    }
    auto ffn = [&](const std::string& n, const f32x4** w1, const float** b1, const f32x4** w2,
                   const float** b2) -> ppasr_status {
      const float* a1 = get(p + n + ".w_1.weight", (size_t)d * H);
      const float* c1 = get(p + n + ".w_1.bias", H);
      const float* a2 = get(p + n + ".w_2.weight", (size_t)H * d);
      const float* c2 = get(p + n + ".w_2.bias", d);
      if (!a1 || !c1 || !a2 || !c2) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + missing);
      ppasr_status s;
      if ((s = m->upload4(pack_b(d, H, [&](int k, int nn) { return a1[(size_t)k * H + nn]; }), w1)) != PPASR_OK) return s;
      if ((s = m->upload(vec_of(c1, H), b1)) != PPASR_OK) return s;
      if ((s = m->upload4(pack_b(H, d, [&](int k, int nn) { return a2[(size_t)k * d + nn]; }), w2)) != PPASR_OK) return s;
      return m->upload(vec_of(c2, d), b2);
    };
    if (go.macaron && (st = ffn("feed_forward_macaron", &L.ffm_w1, &L.ffm_b1, &L.ffm_w2, &L.ffm_b2)) != PPASR_OK) return st;
    if ((st = ffn("feed_forward", &L.ff_w1, &L.ff_b1, &L.ff_w2, &L.ff_b2)) != PPASR_OK) return st;
    {
      GET(wq, p + "self_attn.linear_q.weight", d * d);
      GET(wk, p + "self_attn.linear_k.weight", d * d);
      GET(wv, p + "self_attn.linear_v.weight", d * d);
      GET(bq, p + "self_attn.linear_q.bias", d);
      GET(bk, p + "self_attn.linear_k.bias", d);
      GET(bv, p + "self_attn.linear_v.bias", d);
      GET(wo, p + "self_attn.linear_out.weight", d * d);
      GET(bo, p + "self_attn.linear_out.bias", d);
      const bool rel = go.pos == PPASR_OPT_POS_REL;  // MultiHeadedAttention (abs_pos / no_pos) has no positional parameters
      const float *wp = nullptr, *pu = nullptr, *pv = nullptr;
      if (rel) {
        wp = get(p + "self_attn.linear_pos.weight", (size_t)d * d);
        pu = get(p + "self_attn.pos_bias_u", pbn);
        pv = get(p + "self_attn.pos_bias_v", pbn);
        if (!wp || !pu || !pv) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + missing);
      }
      const float* bp = nullptr;  // linear_pos has a bias only in GroupedRelPositionMultiHeadedAttention
      if (grouped) {
        bp = get(p + "self_attn.linear_pos.bias", d);
        if (!bp) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + missing);
      }
      const float* ws[3] = {wq, wk, wv};
      UP4(pack_b(d, 3 * d, [&](int k, int n) { return ws[n / d][(size_t)k * d + (n % d)]; }), L.wqkv);
      std::vector<float> bqkv(3 * d);
      std::memcpy(&bqkv[0], bq, d * sizeof(float));
      std::memcpy(&bqkv[d], bk, d * sizeof(float));
      std::memcpy(&bqkv[2 * d], bv, d * sizeof(float));
      UP(bqkv, L.bqkv);
      UP4(pack_b(d, d, [&](int k, int n) { return wo[(size_t)k * d + n]; }), L.wo);
      UP(vec_of(bo, d), L.bo);
      if (rel) {
        UP(vec_of(pu, pbn), L.pos_u);
        UP(vec_of(pv, pbn), L.pos_v);
        const float* bpos_dev = nullptr;
        if (bp) UP(vec_of(bp, d), bpos_dev);
        const float* wpos_dev = nullptr;
        UP(vec_of(wp, (size_t)d * d), wpos_dev);
        void* pt = nullptr;
        HIP_TRY(hipMalloc(&pt, (size_t)max_len * d * sizeof(float)));
        m->allocs.push_back(pt);
        launch_posproj(pe_dev, wpos_dev, bpos_dev, static_cast<float*>(pt), max_len, nullptr, d);
        HIP_TRY(hipGetLastError());
        L.ptab = static_cast<const float*>(pt);
      } else {  // the attention kernel's positional half contracts with zeros (capi_generic.hip)
        L.pos_u = L.pos_v = L.ptab = m->zero_vec;
      }
      if (go.concat_after) {  // concat_linear = Linear(2 size, size) (encoder.py:341-342)
        GET(wc, p + "concat_linear.weight", (size_t)2 * d * d);
        GET(bc, p + "concat_linear.bias", d);
        UP4(pack_b(2 * d, d, [&](int k, int n) { return wc[(size_t)k * d + n]; }), m->gen_x[i].wcat);
        UP(vec_of(bc, d), m->gen_x[i].bcat);
      }
    }
    if (go.use_cnn) {
      GET(p1w, p + "conv_module.pointwise_conv1.weight", 2 * d * d);
      GET(p1b, p + "conv_module.pointwise_conv1.bias", 2 * d);
      GET(dww, p + "conv_module.depthwise_conv.weight", d * KSi);
      GET(dwb, p + "conv_module.depthwise_conv.bias", d);
      GET(p2w, p + "conv_module.pointwise_conv2.weight", d * d);
      GET(p2b, p + "conv_module.pointwise_conv2.bias", d);
      // Conv1D weight [out][in][1]: W[k][n] = w[n][k]; GLU value = channels [0,256), gate = [256,512)
      UP4(pack_b(d, 2 * d, [&](int k, int n) { return p1w[(size_t)n * d + k]; }), L.pw1);
      std::vector<float> b1p(p1b, p1b + 2 * d), gp(d);
      for (int c = 0; c < d; ++c) gp[c] = p1b[c] * (1.0f / (1.0f + expf(-p1b[c + d])));
      UP(b1p, L.pw1_b);
      UP(gp, L.glu_pad);
      std::vector<float> dwt((size_t)KSi * d);
      for (int c = 0; c < d; ++c)
        for (int j = 0; j < KSi; ++j) dwt[(size_t)j * d + c] = dww[(size_t)c * KSi + j];
      UP(dwt, L.dw_w);
      UP(vec_of(dwb, d), L.dw_b);
      UP4(pack_b(d, d, [&](int k, int n) { return p2w[(size_t)n * d + k]; }), L.pw2);
      UP(vec_of(p2b, d), L.pw2_b);
    }
  }
  {
    GET(ag, "encoder.after_norm.weight", d);
    GET(ab, "encoder.after_norm.bias", d);
    GET(cw, "ctc.ctc_lo.weight", (size_t)d * V);
    GET(cb, "ctc.ctc_lo.bias", V);
    UP(vec_of(ag, d), m->head.ln_g);
    UP(vec_of(ab, d), m->head.ln_b);
    m->head.V = V;
    m->head.n_tiles = (V + 31) / 32;
    UP4(pack_b(d, V, [&](int k, int n) { return cw[(size_t)k * V + n]; }), m->head.w);
    std::vector<float> cbp((size_t)m->head.n_tiles * 32, 0.f);
    std::memcpy(cbp.data(), cb, V * sizeof(float));
    UP(cbp, m->head.b);
    if (generic) {  // general route: the head as a plain dense layer, vocabulary padded to whole 256-column blocks
      m->gen_vpad = (V + 255) / 256 * 256;
      const int Vp = m->gen_vpad;
      UP4(pack_b(d, Vp, [&](int k, int n) { return n < V ? cw[(size_t)k * V + n] : 0.f; }), m->gen_head_w);
      std::vector<float> hb(Vp, 0.f);
      std::memcpy(hb.data(), cb, V * sizeof(float));
      UP(hb, m->gen_head_b);
    }
  }
  HIP_TRY(hipDeviceSynchronize());
  *out = guard.release();
  return PPASR_OK;
#undef GET
#undef UP
#undef UP4
}

ppasr_status ppasr_destroy(ppasr_handle h) {
  delete h;
  return PPASR_OK;
}

int ppasr_out_frames(ppasr_handle h, int T) {
  if (T < (h ? h->min_frames() : 7)) return 0;
  int tp = h ? h->front_dims(T).Tp : ((T - 1) / 2 - 1) / 2;
  // Efficient-Conformer: the stride-2 conv layer halves the frame rate (ceil), efficient_conformer/encoder.py:252-257
  if (h)
    for (int n = __builtin_popcount(eff_stride_mask(h->desc)); n > 0; --n) tp = (tp + 1) / 2;
  return tp;
}

}  // extern "C"
WsLayout ws_layout(const ppasr_model_s* m, int B, int T) {
  if (m->generic) {
    WsLayout g{};
    g.total = generic_ws_floats(m, B, T);
    return g;
  }
  const auto fd = m->front_dims(T);
  const size_t T1 = fd.T1, Tp = fd.Tp;
  const size_t M = (size_t)B * Tp;
  auto al = [](size_t n) { return (n + 63) & ~(size_t)63; };
  WsLayout w;
  size_t o = 0;
  w.y1 = o; o += al((size_t)B * T1 * m->F1 * kD);  // (conv2d8: the third conv's output reuses it)
  w.y2 = o; o += al((size_t)B * (fd.T2 ? fd.T2 : Tp) * m->F2 * kD);
  w.xa = o; o += al(M * kD);
  w.xb = o; o += al(M * kD);
  w.xc = o; o += al(M * kD);
  w.qkv = o; o += al(M * 3 * kD);
  w.ctx = o; o += al(M * kD);
  w.g = o; o += al(M * kD);
  w.rmax = o; o += al(M);
  w.rsum = o; o += al(M);
  w.fa = o; o += al(M);
  w.fp = o; o += al(M);
  w.xs = o; o += al(M * kD);  // saved full-resolution activations (Squeezeformer time reduction)
  // values in fragment order for the fused attention route (VtOut): a key sub-block of 64 rows may start up to 7 rows
  // before an utterance and end up to 63 rows behind the last one
  w.vt_stride = (int)(al(M) + 128);
  w.vt = o; o += al((size_t)kD * w.vt_stride);
  w.total = o;
  return w;
}
// slices of a launch with `units` independent pieces per row block (K chunks of the input projection, vocabulary tile
// groups of the CTC head) on the split route: as many as fill the chip once, at least the feed-forward split
int wide_slices_for(const ppasr_model_s* m, int M, int units) {
  const int S = ffn_split_for(m, M);
  if (S <= 1) return S;
  const int blocks = (M + kRows - 1) / kRows;
  return std::max(S, std::min(units, 256 / blocks));
}
int ffn_split_for(const ppasr_model_s* m, int M) {
  if (m->desc.model_type == PPASR_MODEL_DEEPSPEECH2) return 1;
  const int n_chunks = m->desc.linear_units / 256;
  if (m->ffn_split >= 0) return m->ffn_split > 1 ? m->ffn_split : 1;
  const int blocks = (M + kRows - 1) / kRows;
  if (blocks > 128) return 1;
  int S = 8;
  while (S > 1 && (blocks * S > 256 || n_chunks % S != 0)) S >>= 1;
  return S;
}

bool block_tables_enabled() {
  static const bool on = [] {
    const char* e = getenv("PPASR_BLOCK_TABLE");
    return !(e && e[0] == '0');
  }();
  return on;
}

bool conv12_enabled(const ppasr_model_s* m) { return m->front_fused != 0; }  // ppasr_set_front_fused

int row_block_for(const ppasr_model_s* m, int B, int Tcur, int mul, int slack, bool skip) {
  // 32 rows on 16 waves (rbt.h kW16) in place of the 8-wave 32-row kernels: OPT-IN per handle,
  // ppasr_set_row_block(PPASR_ROW_BLOCK_32_W16).  Measured on the three bench configurations it is a wash -- the
  // feed-forward streams gain 3 % (92.6 against 89.6 % of the matrix-pipe rate, tools/phase_ts.py --t), the 16-wave
  // depthwise-conv phase requests each window row twice as often and the launch is longer: cfg2 5.99 ms against 5.92,
  // cfg4 9.10 = 9.10, cfg5 6.73 against 6.80 (NOTES.md "Measured and not adopted").
  if (m->row_block == 16 || m->row_block == 32 || m->row_block == kW16) return m->row_block;
  long long rows = (long long)B * Tcur;
  if (skip && (int)m->lens_hint.size() == B) {
    rows = 0;
    for (int b = 0; b < B; ++b) {
      const long long len = m->lens_hint[b];
      const long long need = (len > 0 ? (len + mul - 1) / mul : 0) + slack;
      rows += need < Tcur ? need : Tcur;
    }
  }
  // Rounds of one workgroup per CU: ceil(blocks / 256), a 16-row round 0.52 of a 32-row one (tools/microbench_rb16: 4.3 us
  // per 16-row GEMM unit against 8.4 us per 32-row one; tools/r04_tune_rows.py: whole-encoder timings over batch sizes).
  // 33 .. 128 blocks of 32 rows: one half-as-long round instead of a half-empty one; 257 .. 384: three short rounds instead
  // of two long ones.  Up to 32 blocks the split route (ffn_split_for: 8 hidden slices per row block) fills more CUs than
  // 2 x the blocks would.
  constexpr int min_blocks = 32;
  const long long b32 = (rows + 31) / 32, b16 = (rows + 15) / 16;
  if (b32 <= min_blocks) return 32;  // (split route: the 8-wave kernels)
  const double c32 = (double)((b32 + 255) / 256), c16 = 0.52 * (double)((b16 + 255) / 256);
  return c16 < c32 ? 16 : 32;
}

extern "C" {

ppasr_status ppasr_set_row_block(ppasr_handle h, int rows) {
  if (!h) return fail(PPASR_EINVAL, "null handle");
  if (rows != -1 && rows != 16 && rows != 32 && rows != PPASR_ROW_BLOCK_32_W16)
    return fail(PPASR_EINVAL, "row block: -1 (by grid size), 16, 32 or PPASR_ROW_BLOCK_32_W16");
  h->row_block = rows;
  return PPASR_OK;
}

ppasr_status ppasr_set_lengths_hint(ppasr_handle h, const int64_t* lens_host, int B) {
  if (!h) return fail(PPASR_EINVAL, "null handle");
  if (!lens_host || B <= 0) h->lens_hint.clear();
  else h->lens_hint.assign(lens_host, lens_host + B);
  return PPASR_OK;
}

size_t ppasr_workspace_bytes(ppasr_handle h, int B, int T) {
  if (!h || B <= 0 || T < 7) return 0;
  return ws_layout(h, B, T).total * sizeof(float);
}

ppasr_status ppasr_set_debug_taps(ppasr_handle h, float* taps, size_t n_floats) {
  if (!h) return fail(PPASR_EINVAL, "null handle");
  h->taps = taps;
  h->taps_floats = taps ? n_floats : 0;
  return PPASR_OK;
}

long long ppasr_edit_distance(const int32_t* a, int na, const int32_t* b, int nb) {
  if (na < 0 || nb < 0 || (na > 0 && !a) || (nb > 0 && !b)) return -1;
  if (na < nb) {
    std::swap(a, b);
    std::swap(na, nb);
  }
  if (nb == 0) return na;
  std::vector<int> prev(nb + 1), cur(nb + 1);
  for (int j = 0; j <= nb; ++j) prev[j] = j;
  for (int i = 1; i <= na; ++i) {
    cur[0] = i;
    const int32_t ca = a[i - 1];
    for (int j = 1; j <= nb; ++j) {
      const int sub = prev[j - 1] + (ca != b[j - 1]);
      const int del = prev[j] + 1, ins = cur[j - 1] + 1;
      cur[j] = sub < del ? (sub < ins ? sub : ins) : (del < ins ? del : ins);
    }
    std::swap(prev, cur);
  }
  return prev[nb];
}

ppasr_status ppasr_set_ffn_split(ppasr_handle h, int mode) {
  if (!h) return fail(PPASR_EINVAL, "null handle");
  if (mode != -1 && mode != 0 && mode != 2 && mode != 4 && mode != 8) return fail(PPASR_EINVAL, "ffn split: -1, 0, 2, 4 or 8");
  if (mode > 0 && (h->desc.linear_units / 256) % mode != 0) return fail(PPASR_EINVAL, "ffn split must divide linear_units / 256");
  h->ffn_split = mode;
  return PPASR_OK;
}

ppasr_status ppasr_set_front_fused(ppasr_handle h, int mode) {
  if (!h) return fail(PPASR_EINVAL, "null handle");
  if (mode < -1 || mode > 1) return fail(PPASR_EINVAL, "front fused: -1 (default), 0 or 1");
  h->front_fused = mode;
  return PPASR_OK;
}

// ---- range guard of the fp16 x3 mode (csrc/h3.h): the event counters of the translation units that hold fp16 x3 kernels,
// snapshotted in stream order
struct GuardPtrs {
  const unsigned int* p[ppasr_model_s::kGuardN];
};
__global__ void k_h3_snapshot(GuardPtrs g, unsigned int* dst) {
  for (int i = 0; i < ppasr_model_s::kGuardN; ++i) dst[i] = *g.p[i];
}
static GuardPtrs guard_ptrs(ppasr_handle h) {
  GuardPtrs g;
  for (int i = 0; i < ppasr_model_s::kGuardN; ++i) g.p[i] = h->guard_ctr[i];
  return g;
}

static ppasr_status guard_alloc(ppasr_handle h) {
  if (h->guard_dev) return PPASR_OK;
  constexpr int N = ppasr_model_s::kGuardN;
  h->guard_ctr[0] = conformer_h3_ovf_counter();
  h->guard_ctr[1] = squeezeformer_h3_ovf_counter();
  h->guard_ctr[2] = front_h3_ovf_counter();
  h->guard_ctr[3] = ctc_head_h3_ovf_counter();
  h->guard_ctr[4] = split_route_h3_ovf_counter();
  for (int i = 0; i < N; ++i)
    if (!h->guard_ctr[i]) return fail(PPASR_EHIP, "fp16 x3 range guard: counter symbols not found");
  void* d = nullptr;
  HIP_TRY(hipMalloc(&d, 2 * N * sizeof(unsigned int)));
  h->allocs.push_back(d);
  h->guard_dev = static_cast<unsigned int*>(d);
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, 2 * N * sizeof(unsigned int), hipHostMallocDefault));
  h->guard_host = static_cast<unsigned int*>(p);
  return PPASR_OK;
}

ppasr_status ppasr_set_gemm_mode(ppasr_handle h, int mode) {
  if (!h) return fail(PPASR_EINVAL, "null handle");
  if (mode != PPASR_GEMM_F32 && mode != PPASR_GEMM_F16X3) return fail(PPASR_EINVAL, "gemm mode: PPASR_GEMM_F32 or PPASR_GEMM_F16X3");
  if (mode == PPASR_GEMM_F16X3) {
    const int mt = h->desc.model_type;
    // the layer kernels' feed-forward modules (Conformer, Efficient-Conformer) ...
    bool layers_ok = (mt == PPASR_MODEL_CONFORMER || mt == PPASR_MODEL_EFFICIENT_CONFORMER) && !h->generic && !h->layers.empty();
    for (size_t i = 0; layers_ok && i < h->layers.size(); ++i)
      layers_ok = conv_ffn_h3_supported(h->layer_ks[i]) && h->layers[i].ffm_w1 != nullptr;
    // ... and the second convolution of the 4x front end (Squeezeformer's depthwise-separable one is folded into the same
    // dense [9 * 256][256] weight at create time)
    const bool front_ok = !h->generic && mt != PPASR_MODEL_DEEPSPEECH2 && h->desc.input_layer == 0 && h->front.conv2_k == 3 &&
                          h->front.conv2_w != nullptr;
    // ... Squeezeformer: the two feed-forward modules of a layer (k_sq_mid_h3 / k_sq_tail_h3)
    const bool sq_ok = mt == PPASR_MODEL_SQUEEZEFORMER && !h->generic && !h->sq_layers.empty() &&
                       sq_h3_supported(h->desc.cnn_module_kernel, 4);
    if (!layers_ok && !front_ok && !sq_ok)
      return fail(PPASR_EUNSUPPORTED, "fp16 x3 GEMMs: built for the fused 256-wide routes (feed-forward modules of Conformer / "
                                      "Efficient-Conformer / Squeezeformer layers; conv2 of the 4x front end)");
    const int d = h->desc.output_size, H = h->desc.linear_units;
    {
      const ppasr_status g = guard_alloc(h);
      if (g != PPASR_OK) return g;
    }
    // weights are scaled by 2^8 into fp16: |w| >= 255.9 would overflow (k_repack_h3 counts such weights -- in a word of this
    // call's own, not in the run-time guard's process-wide counters, which a concurrent fp16 x3 encode on another handle
    // may raise at any time)
    unsigned int w_before = 0, w_after = 0;
    unsigned int* w_ovf = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w_ovf), sizeof(unsigned int)));
    struct OvfFree { unsigned int* p; ~OvfFree() { (void)hipFree(p); } } w_ovf_free{w_ovf};
    HIP_TRY(hipMemset(w_ovf, 0, sizeof(unsigned int)));
    const size_t alloc_mark = h->allocs.size();  // re-packed copies made by THIS call start here (freed if the mode is refused)
    HIP_TRY(hipDeviceSynchronize());
    if (layers_ok && h->layers_h3.empty()) {
      std::vector<LayerW> view = h->layers;
      for (LayerW& L : view) {
        // W1 [d][H]: H / 32 column tiles of d / 8 k-groups; W2 [H][d]: d / 32 tiles of H / 8 k-groups; then the d-deep
        // projections [Wq|Wk|Wv] (3d columns), linear_out, pointwise_conv1 (2d columns), pointwise_conv2
        struct { const f32x4** w; int n_tiles, G; } items[8] = {
            {&L.ffm_w1, H / 32, d / 8}, {&L.ffm_w2, d / 32, H / 8}, {&L.ff_w1, H / 32, d / 8}, {&L.ff_w2, d / 32, H / 8},
            {&L.wqkv, 3 * d / 32, d / 8}, {&L.wo, d / 32, d / 8},   {&L.pw1, 2 * d / 32, d / 8}, {&L.pw2, d / 32, d / 8}};
        for (auto& it : items) {
          void* dst = nullptr;
          HIP_TRY(hipMalloc(&dst, (size_t)it.n_tiles * it.G * 256 * sizeof(float)));
          h->allocs.push_back(dst);
          launch_repack_h3(*it.w, static_cast<f32x4*>(dst), it.n_tiles, it.G, w_ovf, nullptr);
          *it.w = static_cast<const f32x4*>(dst);
        }
        // the layer's projected positional table as operand planes (the fused attention's score MFMAs in the mode)
        if (L.ptab != h->zero_vec) {
          void* dst = nullptr;
          HIP_TRY(hipMalloc(&dst, (size_t)h->desc.max_len * d * sizeof(float)));
          h->allocs.push_back(dst);
          launch_split_rows_h3(L.ptab, static_cast<float*>(dst), h->desc.max_len, w_ovf, nullptr);
          L.ptab = static_cast<const float*>(dst);
        }
      }
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipDeviceSynchronize());
      h->layers_h3 = std::move(view);
    }
    if (sq_ok && h->sq_layers_h3.empty()) {
      std::vector<SqLayerW> view = h->sq_layers;
      for (SqLayerW& L : view) {
        const f32x4** w[4] = {&L.ff1_w1, &L.ff1_w2, &L.ff2_w1, &L.ff2_w2};
        for (int j = 0; j < 4; ++j) {
          void* dst = nullptr;
          HIP_TRY(hipMalloc(&dst, (size_t)d * H * sizeof(float)));
          h->allocs.push_back(dst);
          launch_repack_h3(*w[j], static_cast<f32x4*>(dst), (j & 1) ? d / 32 : H / 32, (j & 1) ? H / 8 : d / 8, w_ovf, nullptr);
          *w[j] = static_cast<const f32x4*>(dst);
        }
      }
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipDeviceSynchronize());
      h->sq_layers_h3 = std::move(view);
    }
    if ((layers_ok || sq_ok || front_ok) && !h->head_w_h3 && h->head.w) {  // the CTC head of the fused routes
      void* dst = nullptr;
      HIP_TRY(hipMalloc(&dst, (size_t)h->head.n_tiles * (d / 8) * 256 * sizeof(float)));
      h->allocs.push_back(dst);
      launch_repack_h3(h->head.w, static_cast<f32x4*>(dst), h->head.n_tiles, d / 8, w_ovf, nullptr);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipDeviceSynchronize());
      h->head_w_h3 = static_cast<const f32x4*>(dst);
    }
    if (front_ok && !h->conv2_w_h3) {  // K = 9 * 256
      void* dst = nullptr;
      HIP_TRY(hipMalloc(&dst, (size_t)9 * d * d * sizeof(float)));
      h->allocs.push_back(dst);
      launch_repack_h3(h->front.conv2_w, static_cast<f32x4*>(dst), d / 32, 9 * d / 8, w_ovf, nullptr);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipDeviceSynchronize());
      h->conv2_w_h3 = static_cast<const f32x4*>(dst);
      // the input projection behind it: K = F2 * 256 (a whole number of 256-deep chunks), 256 columns
      const int Ke = h->F2 * d;
      void* dste = nullptr;
      HIP_TRY(hipMalloc(&dste, (size_t)Ke * d * sizeof(float)));
      h->allocs.push_back(dste);
      launch_repack_h3(h->front.embed_w, static_cast<f32x4*>(dste), d / 32, Ke / 8, w_ovf, nullptr);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipDeviceSynchronize());
      h->embed_w_h3 = static_cast<const f32x4*>(dste);
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&w_after, w_ovf, sizeof(unsigned int), hipMemcpyDeviceToHost));
    if (w_after != w_before) {
      // the mode stays off, the views are dropped so that a later call re-checks, and the copies this call made are freed
      // (a retry does not pile up second copies of the weights)
      h->layers_h3.clear();
      h->sq_layers_h3.clear();
      h->head_w_h3 = h->conv2_w_h3 = h->embed_w_h3 = nullptr;
      for (size_t i = alloc_mark; i < h->allocs.size(); ++i) (void)hipFree(h->allocs[i]);
      h->allocs.resize(alloc_mark);
      return fail(PPASR_EUNSUPPORTED, "fp16 x3 GEMMs: a weight of magnitude >= 255.9 (or a positional-table entry >= 4094) does not fit the scaled fp16 pieces");
    }
    for (int i = 0; i < ppasr_model_s::kGuardN; ++i)
      HIP_TRY(hipMemcpy(&h->guard_seen[i], h->guard_ctr[i], sizeof(unsigned int), hipMemcpyDeviceToHost));
    h->gemm_coverage = (layers_ok ? PPASR_GEMM_COVERS_LAYERS : 0) | (sq_ok ? PPASR_GEMM_COVERS_LAYERS : 0) |
                       (front_ok ? PPASR_GEMM_COVERS_FRONT : 0) | (h->head_w_h3 ? PPASR_GEMM_COVERS_HEAD : 0);
  } else {
    h->gemm_coverage = 0;
  }
  h->gemm_mode = mode;
  return PPASR_OK;
}

int ppasr_gemm_coverage(ppasr_handle h) { return h ? h->gemm_coverage : 0; }

ppasr_status ppasr_set_skip_padding(ppasr_handle h, int enable) {
  if (!h) return fail(PPASR_EINVAL, "null handle");
  // the ragged mode is built into the fused 256-column kernels behind the 4x front end; other handles would silently
  // compute (and return) every padded row, which is not what the caller asked for
  if (enable && (h->desc.model_type == PPASR_MODEL_DEEPSPEECH2 || h->desc.input_layer == 1))
    return fail(PPASR_EUNSUPPORTED, "skip_padding: built behind the conv front ends (DeepSpeech2 and input_layer = linear "
                                    "compute every row)");
  h->skip_padding = enable != 0;
  return PPASR_OK;
}

static ppasr_status encode_impl(ppasr_handle h, const float* feats, const int64_t* lens, int B, int T, float* probs,
                                float* logits, int32_t* frame_argmax, float* frame_maxprob, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!h || !feats || !workspace) return fail(PPASR_EINVAL, "null argument");
  if (B <= 0 || T < h->min_frames()) return fail(PPASR_EINVAL, "need B > 0 and T >= 7 (conv2d6: 11, conv2d8: 15) frames");
  if (h->desc.model_type == PPASR_MODEL_DEEPSPEECH2) return fail(PPASR_EINVAL, "deepspeech2 handles use ppasr_ds2_encode");
  const auto fd = h->front_dims(T);
  const int F = h->desc.input_dim, T1 = fd.T1, F1 = h->F1, Tp = fd.Tp, F2 = h->F2;
  const int sub = h->sub_rate();  // frame t of the encoder is PAD iff sub * t >= len (the reference's mask slicing)
  if (Tp >= h->desc.max_len) return fail(PPASR_EINVAL, "utterance longer than the positional table (embedding.py:64-66)");
  const int M = B * Tp;
  const WsLayout wl = ws_layout(h, B, T);
  if (workspace_bytes < wl.total * sizeof(float)) return fail(PPASR_ENOSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  if (h->generic)
    return h->desc.model_type == PPASR_MODEL_SQUEEZEFORMER
               ? generic_sq_encode(h, feats, lens, B, T, probs, logits, frame_argmax, frame_maxprob, ws, st)
               : generic_encode(h, feats, lens, B, T, probs, logits, frame_argmax, frame_maxprob, ws, st);
  if (h->desc.model_type == PPASR_MODEL_SQUEEZEFORMER)
    return squeezeformer_encode(h, feats, lens, B, T, probs, logits, frame_argmax, frame_maxprob, ws, wl, st);
  float *y1 = ws + wl.y1, *y2 = ws + wl.y2, *xa = ws + wl.xa, *xb = ws + wl.xb, *xc = ws + wl.xc;
  float *qkv = ws + wl.qkv, *ctx = ws + wl.ctx, *g = ws + wl.g;
  const VtOut vt_out{ws + wl.vt, wl.vt_stride};
  // the fused attention reads whole 64-row key sub-blocks of V^T: rows outside an utterance (padding behind the last
  // row, frames a ragged batch skips) are multiplied by p = 0 and must be finite
  HIP_TRY(hipMemsetAsync(ws + wl.vt, 0, (size_t)kD * wl.vt_stride * sizeof(float), st));
  size_t tap_off = 0;
  auto tap = [&](const float* src, size_t n) {
    if (h->taps && tap_off + n <= h->taps_floats)
      (void)hipMemcpyAsync(h->taps + tap_off, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    tap_off += n;
  };
  if (h->prof) {
    h->ev_used = 0;
    h->spans.clear();
  }
  auto timed = [&](int cls, auto&& fn) {
    if (!h->prof) {
      fn();
      return;
    }
    // every kernel launched by fn() gets its own (start, stop) events attached to its dispatch (launch.h)
    h->spans.push_back({cls, {}});
    g_launch_prof.ctx = h;
    g_launch_prof.next = [](void* ctx, const void*, hipEvent_t* s, hipEvent_t* e) {
      ppasr_model_s* m = static_cast<ppasr_model_s*>(ctx);
      *s = m->next_event();
      *e = m->next_event();
      m->spans.back().ev.emplace_back(*s, *e);
    };
    fn();
    g_launch_prof = LaunchProf{};
  };
  // ragged batches (ppasr_set_skip_padding): rows behind an utterance's valid frames + slack are skipped.  Slack =
  // what valid outputs read from the rows behind them: the right context of the non-causal conv module, and with a
  // rate change (Efficient-Conformer) the stride layer's 2j / 2j+1 rows and the 3-frame groups of grouped attention.
  const bool eff = h->desc.model_type == PPASR_MODEL_EFFICIENT_CONFORMER;
  // (6x / 8x front ends: the LAYERS and the head skip -- frame t is valid iff 6t / 8t < len --, the front end itself
  //  computes every row: its skip rules are written for the 3x3 / 2 pair of Conv2dSubsampling4)
  const bool skip = h->skip_padding && lens && !h->taps && h->desc.input_layer != 1;
  const bool skip_front = skip && h->desc.input_layer == 0;
  const int rc = h->desc.causal ? 0 : (h->desc.cnn_module_kernel - 1) / 2;
  const int slack_half = rc + 4, slack_full = eff ? 2 * slack_half + rc + 8 : rc + 4;
  auto pskip = [&](int Tcur, int mul_cur) {
    PadSkip ps;
    if (skip) {
      ps.lens = lens;
      ps.Tp = Tcur;
      ps.mul = mul_cur;
      ps.slack = mul_cur == sub ? slack_full : slack_half;
    }
    return ps;
  };
  // Conv2dSubsampling4: both convolutions in one launch, conv1's output never leaves the chip (front_fused.hip)
  // (fp16 x3 mode: conv2 runs on that route as its own launch behind k_conv1)
  const f32x4* conv2_h3 = h->gemm_mode == PPASR_GEMM_F16X3 ? h->conv2_w_h3 : nullptr;
  const bool conv12 = h->desc.input_layer == 0 && !conv2_h3 && conv12_enabled(h) && conv12_supported(h->front, F, F2);
  const PadSkip ps_front = skip_front ? pskip(Tp, sub) : PadSkip{};  // (the front end's own kernels)
  if (!conv12) timed(0, [&] { launch_conv1(feats, h->front, y1, B, T, F, T1, F1, st, ps_front); });
  if (h->desc.input_layer == 8) {
    // Conv2dSubsampling8: conv1 -> conv2 (3x3 / 2) -> conv3 (3x3 / 2, written over conv1's output) -> linear
    timed(1, [&] {
      launch_conv_stage(y1, h->front.conv2_w, h->front.conv2_b, y2, B, T1, F1, fd.T2, F2, 3, 2, st);
      launch_conv_stage(y2, h->front.conv3_w, h->front.conv3_b, y1, B, fd.T2, F2, Tp, h->F3, 3, 2, st);
    });
    timed(2, [&] { launch_embed(y1, h->front, xa, M, h->F3 * kD, sqrtf((float)kD), false, st, PadSkip{}, ffn_split_for(h, M), y2); });
  } else {
    timed(1, [&] {
      // (ragged batches: the active-tile table of conv2 lives in the CTC head's statistics buffer, unused until the head)
      int* tile_tab = (size_t)B + 2 <= ((size_t)M + 63) / 64 * 64 ? reinterpret_cast<int*>(ws + wl.rmax) : nullptr;
      if (conv12) launch_conv12(feats, h->front, y2, B, T, F, Tp, F2, st, ps_front, tile_tab);
      else launch_conv2(y1, h->front, y2, B, T1, F1, Tp, F2, st, ps_front, tile_tab, h->desc.input_layer == 0 ? conv2_h3 : nullptr);
    });
    timed(2, [&] {
      launch_embed(y2, h->front, xa, M, F2 * kD, sqrtf((float)kD), false, st, ps_front, wide_slices_for(h, M, F2), y1,
                   conv2_h3 ? h->embed_w_h3 : nullptr);
    });
  }
  tap(xa, (size_t)M * kD);
  const int n_chunks = h->desc.linear_units / 256;
  // ragged batches: lists of the active row blocks per (frame rate, block size), made on first use (rowblock.h
  // PadSkip::tab; they live behind conv2's tile table in the CTC head's statistics buffers, unused until the head)
  struct BlkTab { int Ti, R; int* tab; } btabs[4];
  int n_bt = 0;
  size_t bt_off = ((size_t)B + 2 + 15) / 16 * 16;
  const bool bt_ok = skip && block_tables_enabled() && (size_t)B + 2 <= ((size_t)M + 63) / 64 * 64;
  auto with_table = [&](PadSkip p, int Tcur, int R) {
    if (!bt_ok) return p;
    for (int k = 0; k < n_bt; ++k)
      if (btabs[k].Ti == Tcur && btabs[k].R == R) { p.tab = btabs[k].tab; return p; }
    const size_t n = 1 + ((size_t)B * Tcur + R - 1) / R;
    if (n_bt == 4 || bt_off + n > 2 * (((size_t)M + 63) / 64 * 64)) return p;
    int* t = reinterpret_cast<int*>(ws + wl.rmax) + bt_off;
    launch_block_table(p, B * Tcur, R, t, st);
    btabs[n_bt++] = BlkTab{Tcur, R, t};
    bt_off += (n + 15) / 16 * 16;
    p.tab = t;
    return p;
  };
  int Ti = Tp, mul = sub, pstride = 1;  // frames per utterance / pad-mask multiplier / positional stride of the current layer
  bool s1_done = false;               // this layer's S1 already ran inside the previous layer's last launch
  for (int i = 0; i < h->desc.num_blocks; ++i) {
    const LayerW& L = h->layers[i];
    const int Mi = B * Ti;
    const int grp = h->layer_group[i];
    // plain 4 x 64 heads: attention and the out-projection / GLU stage run as one launch (context rows stay in LDS);
    // the debug taps need the context tensor, so they take the two-kernel route
    // (an under-filled grid is latency-bound either way, and the two-kernel route then has 4x the workgroups in its
    //  attention half, one per head: 2 - 6 % faster end to end up to 128 row blocks, measured in round 3), 10 %
    //  slower at the bench shape)
    constexpr int fuse_min_blocks = 128;
    auto fusable = [&](int layer) {
      // (the fused kernel reads the values in fragment order, which only the fused QKV stage -- ffn_qkv_body -- writes: a
      //  FORCED split of a large batch (ppasr_set_ffn_split(2 / 4 / 8), k_ln_qkv) therefore takes the two-kernel route)
      return h->layer_group[layer] == 1 && h->desc.attention_heads == 4 && !h->taps && ffn_split_for(h, Mi) == 1 &&
             (h->ffn_split == 0 || (Mi + kRows - 1) / kRows > fuse_min_blocks);  // (ppasr_set_ffn_split(0): always fused)
    };
    const PadSkip ps = pskip(Ti, mul);
    // under-filled launch, 33 .. 128 row blocks: the 16-row-block kernels (conformer_kernels_t.hip) -- twice the
    // workgroups, each half as long -- with the stand-alone attention between them; up to 32 blocks the split route below
    const int rows = (!h->taps && conv_ffn_16_supported(h->layer_ks[i], Ti)) ? row_block_for(h, B, Ti, mul, ps.slack, skip) : 32;
    const bool r16 = rows == 16 && h->ffn_split < 0;
    const bool fuse_attn = !r16 && fusable(i);
    // under-filled grid: FFNs split over S workgroups per row block (partial sums in the conv1 buffer, free by now)
    const int S = r16 ? 1 : ffn_split_for(h, Mi);
    // full grid: the same 32-row blocks on 16 waves (k_*_t<kW16>: drop-in for k_ffn_qkv / k_out_glu / k_conv_ffn)
    const bool w16 = rows == kW16 && S == 1;
    const PadSkip psb = S == 1 ? with_table(ps, Ti, r16 ? 16 : 32) : ps;  // (for the kernels of this layer's block size)
    // feed-forward GEMMs on the fp16 x3 route (ppasr_set_gemm_mode): the 8-wave 32-row kernels only
    const bool h3 = h->gemm_mode == PPASR_GEMM_F16X3 && !h->layers_h3.empty() && !r16 && !w16 && S == 1;
    // ... the split route of under-filled launches likewise (its kernels' units; h3 view for the weights only -- the
    // stand-alone attention keeps the fp32 positional table)
    const bool h3s = h->gemm_mode == PPASR_GEMM_F16X3 && !h->layers_h3.empty() && S > 1;
    const LayerW& Lk = (h3 || h3s) ? h->layers_h3[i] : L;
    // ... and with the fused attention the score MFMAs: the QKV stage then leaves K as fp16 hi / lo planes (VtOut::k_h3) and the
    // attention reads the layer's positional planes.  Producer and consumer follow the same rule: layer j's K is planes iff
    // layer j runs k_attn_out_glu_h3 (the producer of j's QKV is j's own S1 launch or the NEXT tail of j - 1, which shares
    // j's row count and block form; the stride layer has no NEXT tail)
    auto vt_for = [&](bool fused, bool mode) {
      VtOut v = fused ? vt_out : VtOut{};
      v.k_h3 = (fused && mode) ? 1 : 0;
      return v;
    };
    float* partial = y1;
    float* x3 = ctx;
    if (!s1_done) {
      if (S > 1) {
        timed(3, [&] {
          launch_ffn_split(xa, L.ln_mac_g, L.ln_mac_b, Lk.ffm_w1, L.ffm_b1, Lk.ffm_w2, L.ffm_b2, 0.5f, nullptr, nullptr, partial,
                           xb, Mi, n_chunks, S, st, ps, false, h3s);
          launch_ln_qkv(xb, qkv, Lk, Mi, st, ps, nullptr, nullptr, h3s);
        });
      } else if (r16) {
        timed(3, [&] { launch_ffn_qkv_16(xa, xb, qkv, L, Mi, n_chunks, st, psb); });
      } else if (w16) {
        timed(3, [&] { launch_ffn_qkv_w16(xa, xb, qkv, L, Mi, n_chunks, st, psb, fuse_attn ? vt_out : VtOut{}); });
      } else {
        timed(3, [&] { launch_ffn_qkv(xa, xb, qkv, Lk, Mi, n_chunks, st, psb, vt_for(fuse_attn, h3), h3); });
      }
    }
    s1_done = false;
    tap(xb, (size_t)Mi * kD);
    tap(qkv, (size_t)Mi * 3 * kD);
    const int Tt = (Ti + grp - 1) / grp;  // tokens: frames, or zero-padded groups of 3 (pad4group)
    AttnArgs a{qkv, 768, qkv + 256, 768, qkv + 512, 768, Tt, Tt, 0, lens, ctx, L.pos_u, L.pos_v, (h3 && fuse_attn) ? Lk.ptab : L.ptab, pstride,
               mul * grp, Ti, Ti, grp};
    a.pad_skip = skip ? ps.slack + 1 : 0;
    a.vt = vt_out.vt;
    a.vt_stride = vt_out.stride;
    if (fuse_attn) {
      timed(9, [&] { launch_attn_out_glu(a, B, xb, xc, g, Lk, st, h3); });
    } else {
      timed(4, [&] { launch_attention(a, B, h->desc.attention_heads, st); });
      tap(ctx, (size_t)Mi * kD);
      // (under-filled launch: pointwise_conv1 + GLU as its own two-column-half launch; the LayerNorm'd rows pass through
      //  xa, which is free between this layer's S1 and its output)
      if (r16) timed(5, [&] { launch_out_glu_16(ctx, xb, xc, g, L, lens, Mi, Ti, mul, st, psb); });
      else if (w16) timed(5, [&] { launch_out_glu_w16(ctx, xb, xc, g, L, lens, Mi, Ti, mul, st, psb); });
      else timed(5, [&] { launch_out_glu(ctx, xb, xc, g, nullptr, h3s ? Lk : L, lens, Mi, Ti, mul, st, psb, S > 1 ? xa : nullptr, h3s); });
    }
    tap(xc, (size_t)Mi * kD);
    tap(g, (size_t)Mi * kD);
    if (eff && i == h->desc.stride_layer_idx) {
      const int Ts = (Ti + 1) / 2;
      timed(6, [&] {
        const int Ss = r16 ? 1 : ffn_split_for(h, B * Ts);
        if (Ss > 1) {  // under-filled: the conv half alone (x3 -> ctx), the feed-forward module over the slices
          launch_conv_ffn_stride(g, nullptr, xc, xa, h3s ? Lk : L, lens, B, Ti, Ts, n_chunks, h->layer_ks[i], mul * 2, st,
                                 pskip(Ts, mul * 2), h->desc.causal != 0, h3s, ctx);
          launch_ffn_split(ctx, L.ln_ff_g, L.ln_ff_b, Lk.ff_w1, L.ff_b1, Lk.ff_w2, L.ff_b2, 0.5f, L.ln_fin_g, L.ln_fin_b, partial,
                           xa, B * Ts, n_chunks, Ss, st, pskip(Ts, mul * 2), false, h3s);
        } else {
          launch_conv_ffn_stride(g, nullptr, xc, xa, (h3 || h3s) ? Lk : L, lens, B, Ti, Ts, n_chunks, h->layer_ks[i], mul * 2, st,
                                 pskip(Ts, mul * 2), h->desc.causal != 0, h3 || h3s);
        }
      });
      Ti = Ts;  // masks[:, :, ::2], pos_emb[:, ::2]  (efficient_conformer/encoder.py:252-257)
      mul *= 2;
      pstride *= 2;
    } else if (S > 1) {
      timed(6, [&] {
        launch_conv_pre(g, nullptr, xc, x3, Lk, lens, Mi, Ti, h->layer_ks[i], mul, st, h->desc.causal != 0, ps, h3s);
        launch_ffn_split(x3, L.ln_ff_g, L.ln_ff_b, Lk.ff_w1, L.ff_b1, Lk.ff_w2, L.ff_b2, 0.5f, L.ln_fin_g, L.ln_fin_b, partial,
                         xa, Mi, n_chunks, S, st, ps, false, h3s);
      });
    } else {
      // fuse the next layer's S1 into this launch (it writes xb / qkv, which this layer no longer reads)
      const LayerW* next = (i + 1 < h->desc.num_blocks) ? (h3 ? &h->layers_h3[i + 1] : &h->layers[i + 1]) : nullptr;
      timed(next ? 8 : 6, [&] {
        // with the next layer's S1 fused in, the layer output itself is only read by the debug taps: skip its store
        if (r16)
          launch_conv_ffn_16(g, xc, next ? nullptr : xa, L, lens, Mi, Ti, n_chunks, h->layer_ks[i], mul, next, xb, qkv, st,
                             h->desc.causal != 0, psb);
        else if (w16)
          launch_conv_ffn_w16(g, xc, next ? nullptr : xa, L, lens, Mi, Ti, n_chunks, h->layer_ks[i], mul, next, xb, qkv, st,
                              h->desc.causal != 0, psb, (next && fusable(i + 1)) ? vt_out : VtOut{});
        else
          launch_conv_ffn(g, nullptr, xc, (next && !h->taps) ? nullptr : xa, Lk, lens, Mi, Ti, n_chunks, h->layer_ks[i], mul,
                          next, xb, qkv, st, h->desc.causal != 0, psb,
                          vt_for(next && fusable(i + 1), h3), h3);
      });
      s1_done = next != nullptr;
    }
    tap(xa, (size_t)B * Ti * kD);
  }
  const int Mo = B * Ti;
  float* lg = logits ? logits : probs;  // probs are produced in place from the logits tap
  int32_t* fa = frame_argmax ? frame_argmax : reinterpret_cast<int32_t*>(ws + wl.fa);
  float* fp = frame_maxprob ? frame_maxprob : ws + wl.fp;
  // (under-filled launch: the vocabulary tiles are split over several workgroups per row block, scratch = conv1 buffer)
  timed(7, [&] {
    const bool head_h3 = h->gemm_mode == PPASR_GEMM_F16X3 && h->head_w_h3;
    HeadW hw = h->head;
    if (head_h3) hw.w = h->head_w_h3;
    launch_ctc_head(xa, hw, lg, fa, fp, ws + wl.rmax, ws + wl.rsum, Mo, st, pskip(Ti, mul),
                    wide_slices_for(h, Mo, std::min((hw.n_tiles + 7) / 8, 32)), y1, head_h3);
  });
  if (probs) {
    if (logits)
      HIP_TRY(hipMemcpyAsync(probs, logits, (size_t)Mo * h->head.V * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_softmax_from_stats(probs, ws + wl.rmax, ws + wl.rsum, Mo, h->head.V, st, pskip(Ti, mul));
  }
  if (skip) launch_zero_pad_rows(probs, logits, fa, fp, lens, B, Ti, mul, h->head.V, st);
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

ppasr_status ppasr_encode(ppasr_handle h, const float* feats, const int64_t* lens, int B, int T, float* probs,
                          float* logits, int32_t* frame_argmax, float* frame_maxprob, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (!h) return fail(PPASR_EINVAL, "null argument");
  if (h->gemm_mode != PPASR_GEMM_F16X3 || !h->gemm_guard || !h->guard_dev)
    return encode_impl(h, feats, lens, B, T, probs, logits, frame_argmax, frame_maxprob, workspace, workspace_bytes, stream);
  // fp16 x3 mode, guard on: counters before / after the call's launches (stream order), one 16-byte read-back, and on a
  // changed counter the same call again on the fp32 kernels (same weights, same workspace; the inputs are untouched)
  hipStream_t st = static_cast<hipStream_t>(stream);
  constexpr int N = ppasr_model_s::kGuardN;
  PPASR_LAUNCH(k_h3_snapshot, dim3(1), dim3(1), 0, st, guard_ptrs(h), h->guard_dev);
  ppasr_status rc = encode_impl(h, feats, lens, B, T, probs, logits, frame_argmax, frame_maxprob, workspace, workspace_bytes, stream);
  if (rc != PPASR_OK) return rc;
  PPASR_LAUNCH(k_h3_snapshot, dim3(1), dim3(1), 0, st, guard_ptrs(h), h->guard_dev + N);
  HIP_TRY(hipMemcpyAsync(h->guard_host, h->guard_dev, 2 * N * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  long long events = 0;
  for (int i = 0; i < N; ++i) {
    h->guard_seen[i] = h->guard_host[N + i];
    events += (long long)(h->guard_host[N + i] - h->guard_host[i]);
  }
  if (events == 0) return PPASR_OK;
  h->guard_events += events;
  h->guard_fallbacks += 1;
  h->gemm_mode = PPASR_GEMM_F32;
  rc = encode_impl(h, feats, lens, B, T, probs, logits, frame_argmax, frame_maxprob, workspace, workspace_bytes, stream);
  h->gemm_mode = PPASR_GEMM_F16X3;
  return rc;
}

ppasr_status ppasr_set_gemm_guard(ppasr_handle h, int enable) {
  if (!h) return fail(PPASR_EINVAL, "null handle");
  h->gemm_guard = enable != 0;
  return PPASR_OK;
}

ppasr_status ppasr_gemm_guard_stats(ppasr_handle h, long long* fallbacks_host, long long* events_host) {
  if (!h) return fail(PPASR_EINVAL, "null handle");
  if (h->guard_dev) {  // (guard off: the events since this handle last looked -- a device-wide wait, then the counters)
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < ppasr_model_s::kGuardN; ++i) {
      unsigned int now = 0;
      HIP_TRY(hipMemcpy(&now, h->guard_ctr[i], sizeof(unsigned int), hipMemcpyDeviceToHost));
      h->guard_events += (long long)(now - h->guard_seen[i]);
      h->guard_seen[i] = now;
    }
  }
  if (fallbacks_host) *fallbacks_host = h->guard_fallbacks;
  if (events_host) *events_host = h->guard_events;
  return PPASR_OK;
}


// =====================================================================================
// CTC prefix beam search (see ctc_beam.hip)
// =====================================================================================
static ppasr_status beam_config(int V, int beam_size, double cutoff_prob, int cutoff_top_n, int blank, int nbest,
                                int max_tokens, BeamConfig* c) {
  if (V <= 1 || V >= 16384) return fail(PPASR_EUNSUPPORTED, "beam search: vocabulary must be in (1, 16384)");
  if (beam_size < 1 || beam_size > kMaxBeam) return fail(PPASR_EUNSUPPORTED, "beam search: beam_size must be in [1, 512]");
  if (blank < 0 || blank >= V) return fail(PPASR_EINVAL, "beam search: blank id out of range");
  if (nbest < 1 || nbest > beam_size || max_tokens < 1) return fail(PPASR_EINVAL, "beam search: bad nbest / max_tokens");
  if (cutoff_top_n < 1) return fail(PPASR_EINVAL, "beam search: cutoff_top_n < 1");
  // candidates per frame: pruned to cutoff_top_n only when cutoff_prob < 1 (upstream get_pruned_log_probs); with
  // cutoff_prob >= 1 -- the default of the reference's wrappers, swig_wrapper.py:38,71 -- upstream keeps EVERY character
  // (sorted when cutoff_top_n < V, in vocabulary order otherwise), and so does the kernel: wide records and the element
  // lists they produce go through HBM scratch (ppasr_ctc_beam_scratch_bytes)
  const int n_cand = (cutoff_prob < 1.0) ? (cutoff_top_n < V ? cutoff_top_n : V) : V;
  c->V = V; c->beam = beam_size; c->blank = blank; c->cutoff_top_n = cutoff_top_n; c->cutoff_prob = cutoff_prob;
  c->n_cand_max = n_cand; c->nbest = nbest; c->max_tokens = max_tokens; c->max_nodes = 0;
  c->sorted = (cutoff_prob < 1.0 || cutoff_top_n < V) ? 1 : 0;
  {
    const char* m = getenv("PPASR_BEAM_MARGIN");  // (tuning knob: rows of the clipped element list, ctc_beam.hip)
    c->margin = m ? atoi(m) : 2;
    if (c->margin < 0) c->margin = 0;
  }
  c->list_cap = beam_list_cap(beam_size, V, n_cand, c->lm.order > 0);
  if (c->list_cap <= 0) return fail(PPASR_EUNSUPPORTED, "beam search: beam_size x vocabulary does not fit LDS");
  return PPASR_OK;
}

// state buffer = B x [header | beam arrays | arena of 1 + (F+1)*beam nodes | node table] | B status words | scratch of the
// pruning pre-pass (B x F frame records of the largest record size), F = max_frames.  The layout is recomputed from
// (state_bytes, B, beam_size) on every call, so it is the same for every chunk of a streaming decode.

// bytes of one utterance's share of the state buffer for a capacity of F frames: state block (ctc_beam.h: header, beam
// arrays, arena and node table for max_nodes = 1 + (F + 1) * beam nodes) + its status word + F pruning records
static size_t beam_max_nodes(size_t F, int beam_size) { return 1 + (F + 1) * (size_t)beam_size; }
static size_t beam_utt_bytes(size_t F, int beam_size) {
  return 4 * beam_state_words(beam_size, (int)beam_max_nodes(F, beam_size)) + 4 + F * 4 * (size_t)prune_rec_words(kSmallCand);
}
// frame capacity a state buffer of `per_utt_bytes` per utterance was sized for (0: too small for one frame)
static size_t beam_frame_capacity(size_t per_utt_bytes, int beam_size) {
  if (per_utt_bytes < beam_utt_bytes(1, beam_size)) return 0;
  const size_t per_frame = beam_utt_bytes(2, beam_size) - beam_utt_bytes(1, beam_size);
  size_t F = 1 + (per_utt_bytes - beam_utt_bytes(1, beam_size)) / per_frame;
  while (F > 1 && beam_utt_bytes(F, beam_size) > per_utt_bytes) --F;  // (the fixed part is padded to an even word)
  return F;
}

size_t ppasr_ctc_beam_state_bytes(int B, int max_frames, int beam_size) {
  if (B <= 0 || max_frames < 0 || beam_size < 1) return 0;
  return (size_t)B * beam_utt_bytes((size_t)(max_frames < 1 ? 1 : max_frames), beam_size);
}

}  // extern "C"
namespace ppasr { const LmDev* lm_device_view(ppasr_lm_handle lm); }  // lm.hip (internal: C++ linkage, not exported)
extern "C" {

ppasr_status ppasr_ctc_beam_search(const float* probs, const int32_t* frame_lens, int B, int T, int V, int beam_size,
                                   double cutoff_prob, int cutoff_top_n, int blank, int nbest, int max_tokens,
                                   int32_t* tokens, int32_t* lens, double* scores, void* state, size_t state_bytes,
                                   int init_state, void* stream) {
  return ppasr_ctc_beam_search_lm(probs, frame_lens, B, T, V, beam_size, cutoff_prob, cutoff_top_n, blank, nbest, max_tokens,
                                  tokens, lens, scores, state, state_bytes, init_state, nullptr, 0.0, 0.0, stream);
}

ppasr_status ppasr_ctc_beam_search_lm(const float* probs, const int32_t* frame_lens, int B, int T, int V, int beam_size,
                                      double cutoff_prob, int cutoff_top_n, int blank, int nbest, int max_tokens,
                                      int32_t* tokens, int32_t* lens, double* scores, void* state, size_t state_bytes,
                                      int init_state, ppasr_lm_handle lm, double alpha, double beta, void* stream) {
  return ppasr_ctc_beam_search_ws(probs, frame_lens, B, T, V, beam_size, cutoff_prob, cutoff_top_n, blank, nbest, max_tokens,
                                  tokens, lens, scores, state, state_bytes, init_state, lm, alpha, beta, nullptr, 0, stream);
}

size_t ppasr_ctc_beam_scratch_bytes(int B, int T, int V, int beam_size, double cutoff_prob, int cutoff_top_n) {
  BeamConfig c{};
  if (B <= 0 || T < 0) return 0;
  c.lm.order = 1;  // sized for a search WITH a scorer (its context summaries take LDS from the element list): enough for both
  if (beam_config(V, beam_size, cutoff_prob, cutoff_top_n, 0, 1, 1, &c) != PPASR_OK) return 0;
  return beam_scratch_bytes(c, B, T);
}

ppasr_status ppasr_ctc_beam_search_ws(const float* probs, const int32_t* frame_lens, int B, int T, int V, int beam_size,
                                      double cutoff_prob, int cutoff_top_n, int blank, int nbest, int max_tokens,
                                      int32_t* tokens, int32_t* lens, double* scores, void* state, size_t state_bytes,
                                      int init_state, ppasr_lm_handle lm, double alpha, double beta, void* scratch,
                                      size_t scratch_bytes, void* stream) {
  if (!tokens || !lens || !scores || !state || (!probs && T > 0)) return fail(PPASR_EINVAL, "null argument");
  if (B <= 0 || T < 0) return fail(PPASR_EINVAL, "empty batch");
  BeamConfig c{};
  if (lm) {  // before beam_config: the LDS budget depends on it
    c.lm = *ppasr::lm_device_view(lm);
    c.alpha = alpha;
    c.beta = beta;
  }
  ppasr_status s = beam_config(V, beam_size, cutoff_prob, cutoff_top_n, blank, nbest, max_tokens, &c);
  if (s != PPASR_OK) return s;
  {
    const char* e = getenv("PPASR_BEAM_NODE_TABLE");
    c.node_table = e ? (atoi(e) != 0) : (lm && c.lm.word_based);
    const char* f = getenv("PPASR_BEAM_FAST");  // (read per call: the tests run both selections in one process)
    c.fast_path = f ? (atoi(f) != 0) : 1;
  }
  const size_t per_utt_bytes = state_bytes / (size_t)B;
  const size_t F = beam_frame_capacity(per_utt_bytes, beam_size);
  if (F == 0) return fail(PPASR_ENOSPACE, "beam search: state buffer too small");
  if ((size_t)T > F) return fail(PPASR_ENOSPACE, "beam search: more frames in one call than the state buffer was sized for");
  c.max_nodes = (int)beam_max_nodes(F, beam_size);
  const size_t block_words = beam_state_words(beam_size, c.max_nodes);
  int32_t* st_words = static_cast<int32_t*>(state);
  int32_t* status = st_words + (size_t)B * block_words;
  int32_t* prune_recs = status + B;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (init_state) {
    HIP_TRY(hipMemsetAsync(status, 0, (size_t)B * 4, hs));
    // empty node tables (the kernel enters every prefix it creates) -- only where the search uses them: the default route
    // of scorer-less / character-LM searches never reads the table, and clearing 24 bytes per node per call is tens of MB
    // of memset on the latency-bound decoder path
    if (c.node_table) {
    const size_t tab_off = (beam_fixed_words(beam_size) + beam_arena_words(c.max_nodes)) * 4, tab_bytes = 12 * beam_table_slots(c.max_nodes);
    HIP_TRY(hipMemset2DAsync(reinterpret_cast<char*>(state) + tab_off, block_words * 4, 0, tab_bytes, (size_t)B, hs));
    }
  }
  const size_t need_scratch = beam_scratch_bytes(c, B, T);
  if (need_scratch > 0 && (!scratch || scratch_bytes < need_scratch))
    return fail(PPASR_ENOSPACE, "beam search: this pruning configuration keeps more characters per frame than the LDS-resident "
                                "search holds (cutoff_prob >= 1 keeps the whole vocabulary): call ppasr_ctc_beam_search_ws with "
                                "ppasr_ctc_beam_scratch_bytes(...) bytes of device scratch");
  HIP_TRY(launch_ctc_beam(probs, frame_lens, B, T, c, prune_recs, st_words, init_state, 1, tokens, lens, scores, status,
                          need_scratch ? scratch : nullptr, hs));
  return PPASR_OK;
}

// State layout: B blocks of (fixed + 2 * max_nodes) words (beam arrays, then the parent-pointer arena, node ids in
// creation order), B status words, the per-call pruning records.  A larger buffer therefore holds the same search once
// every block's words sit at the start of the new, longer block (the arena simply has room for more nodes behind them).
ppasr_status ppasr_ctc_beam_state_grow(const void* old_state, size_t old_bytes, void* new_state, size_t new_bytes, int B,
                                       int beam_size, void* stream) {
  if (!old_state || !new_state || B <= 0 || beam_size < 1) return fail(PPASR_EINVAL, "null argument");
  const size_t Fo = beam_frame_capacity(old_bytes / (size_t)B, beam_size), Fn = beam_frame_capacity(new_bytes / (size_t)B, beam_size);
  if (Fo == 0 || Fn == 0) return fail(PPASR_ENOSPACE, "beam search: state buffer too small");
  if (Fn < Fo) return fail(PPASR_EINVAL, "beam search: the new state buffer is smaller than the old one");
  const int mo = (int)beam_max_nodes(Fo, beam_size), mn = (int)beam_max_nodes(Fn, beam_size);
  const size_t ow = beam_state_words(beam_size, mo), nw = beam_state_words(beam_size, mn);
  const size_t head_words = beam_fixed_words(beam_size) + beam_arena_words(mo);  // header, beam arrays, arena: same offsets in both
  hipStream_t hs = static_cast<hipStream_t>(stream);
  const int32_t* o = static_cast<const int32_t*>(old_state);
  int32_t* n = static_cast<int32_t*>(new_state);
  HIP_TRY(hipMemcpy2DAsync(n, nw * 4, o, ow * 4, head_words * 4, (size_t)B, hipMemcpyDeviceToDevice, hs));
  HIP_TRY(hipMemcpyAsync(n + (size_t)B * nw, o + (size_t)B * ow, (size_t)B * 4, hipMemcpyDeviceToDevice, hs));  // status
  // the node table is rebuilt for the new size from the arena (every node but the root is an entry)
  const size_t tab_off = (beam_fixed_words(beam_size) + beam_arena_words(mn)) * 4, tab_bytes = 12 * beam_table_slots(mn);
  HIP_TRY(hipMemset2DAsync(reinterpret_cast<char*>(new_state) + tab_off, nw * 4, 0, tab_bytes, (size_t)B, hs));
  HIP_TRY(launch_beam_rehash(n, B, beam_size, mn, hs));
  return PPASR_OK;
}

// Streaming callers of the C-ABI: the kernel flags an utterance whose prefix arena ran out (more cumulative frames than
// the state buffer was sized for) in a status word of the state buffer; this reads the B words back (synchronises the
// stream) and returns PPASR_ENOSPACE if any is set -- the hypotheses of that utterance are then truncated.
ppasr_status ppasr_ctc_beam_status(const void* state, size_t state_bytes, int B, int beam_size, int32_t* status_host,
                                   void* stream) {
  if (!state || B <= 0 || beam_size < 1) return fail(PPASR_EINVAL, "null argument");
  const size_t F = beam_frame_capacity(state_bytes / (size_t)B, beam_size);
  if (F == 0) return fail(PPASR_ENOSPACE, "beam search: state buffer too small");
  const int32_t* status = static_cast<const int32_t*>(state) + (size_t)B * beam_state_words(beam_size, (int)beam_max_nodes(F, beam_size));
  std::vector<int32_t> host(B);
  hipStream_t hs = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemcpyAsync(host.data(), status, (size_t)B * 4, hipMemcpyDeviceToHost, hs));
  HIP_TRY(hipStreamSynchronize(hs));
  bool any = false;
  for (int b = 0; b < B; ++b) {
    if (status_host) status_host[b] = host[b];
    any |= host[b] != 0;
  }
  if (any) return fail(PPASR_ENOSPACE, "beam search: the prefix arena of at least one utterance is exhausted (state sized for fewer frames)");
  return PPASR_OK;
}

// ---- kernel-name profiler: every PPASR_LAUNCH of the calling thread between begin and end carries its own dispatch-attached
// event pair (launch.h); entries are keyed by the kernel's function pointer and named from the code object, so the names
// are the ones rocprofv3's kernel trace prints.  Covers every model family and the decoders (bench.py roofline leg). ----
}  // extern "C"
namespace {
struct KProf {
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  struct Rec { const void* fn; hipEvent_t s, e; };
  std::vector<Rec> recs;
  hipEvent_t next() {
    if (used == pool.size()) {
      hipEvent_t e;
      (void)hipEventCreate(&e);
      pool.push_back(e);
    }
    return pool[used++];
  }
};
thread_local KProf g_kprof;

std::string kernel_display_name(const void* fn) {
  const char* mangled = hipKernelNameRefByPtr(fn, nullptr);
  if (!mangled) return "?";
  int status = 0;
  char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
  std::string n = (status == 0 && dem) ? dem : mangled;
  free(dem);
  // drop the parameter list (the last balanced parenthesis group), "void " and the namespace
  if (!n.empty() && n.back() == ')') {
    int depth = 0;
    for (size_t i = n.size(); i-- > 0;) {
      if (n[i] == ')') ++depth;
      else if (n[i] == '(' && --depth == 0) { n.erase(i); break; }
    }
  }
  if (n.rfind("void ", 0) == 0) n.erase(0, 5);
  for (size_t p; (p = n.find("ppasr::")) != std::string::npos;) n.erase(p, 7);
  for (size_t p; (p = n.find("(anonymous namespace)::")) != std::string::npos;) n.erase(p, 23);
  return n;
}
}  // namespace
extern "C" {

ppasr_status ppasr_kprof_begin(void) {
  g_kprof.used = 0;
  g_kprof.recs.clear();
  g_launch_prof.ctx = &g_kprof;
  g_launch_prof.next = [](void* ctx, const void* fn, hipEvent_t* s, hipEvent_t* e) {
    KProf* k = static_cast<KProf*>(ctx);
    *s = k->next();
    *e = k->next();
    k->recs.push_back({fn, *s, *e});
  };
  return PPASR_OK;
}

ppasr_status ppasr_kprof_end(int max_entries, char* names_host, float* total_ms_host, int* launches_host, int* n_out_host) {
  g_launch_prof = LaunchProf{};
  if (!names_host || !total_ms_host || !launches_host || !n_out_host || max_entries <= 0)
    return fail(PPASR_EINVAL, "null argument");
  std::vector<const void*> order;
  std::unordered_map<const void*, std::pair<double, int>> acc;
  for (auto& r : g_kprof.recs) {
    HIP_TRY(hipEventSynchronize(r.e));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, r.s, r.e));
    auto it = acc.find(r.fn);
    if (it == acc.end()) {
      order.push_back(r.fn);
      it = acc.emplace(r.fn, std::make_pair(0.0, 0)).first;
    }
    it->second.first += ms;
    it->second.second += 1;
  }
  g_kprof.recs.clear();
  int n = 0;
  for (const void* fn : order) {
    if (n == max_entries) break;
    const std::string name = kernel_display_name(fn);
    char* dst = names_host + (size_t)n * PPASR_KPROF_NAME_LEN;
    std::snprintf(dst, PPASR_KPROF_NAME_LEN, "%s", name.c_str());
    total_ms_host[n] = (float)acc[fn].first;
    launches_host[n] = acc[fn].second;
    ++n;
  }
  *n_out_host = n;
  return PPASR_OK;
}

static const char* kKernelClassNames[PPASR_N_KERNEL_CLASSES] = {
    "k_conv1", "k_gemm_stream<conv2>", "k_gemm_stream<embed>", "k_ffn_qkv", "k_attention", "k_out_glu", "k_conv_ffn",
    "k_ctc_head", "k_conv_ffn+ffn_qkv", "k_attn_out_glu"};

ppasr_status ppasr_profile_enable(ppasr_handle h, int enable) {
  if (!h) return fail(PPASR_EINVAL, "null handle");
  if (enable && (h->desc.model_type != PPASR_MODEL_CONFORMER && h->desc.model_type != PPASR_MODEL_EFFICIENT_CONFORMER))
    return fail(PPASR_EUNSUPPORTED, "ppasr_profile_enable: kernel classes exist for the Conformer route only; use ppasr_kprof_begin / _end");
  if (enable && h->generic)
    return fail(PPASR_EUNSUPPORTED, "ppasr_profile_enable: the general layer route has no kernel classes; use ppasr_kprof_begin / _end");
  h->prof = enable != 0;
  h->spans.clear();
  h->ev_used = 0;
  return PPASR_OK;
}

ppasr_status ppasr_profile_read(ppasr_handle h, float* total_ms_host, int* launches_host) {
  if (!h || !total_ms_host || !launches_host) return fail(PPASR_EINVAL, "null argument");
  for (int i = 0; i < PPASR_N_KERNEL_CLASSES; ++i) {
    total_ms_host[i] = 0.f;
    launches_host[i] = 0;
  }
  for (auto& sp : h->spans) {
    if (sp.ev.empty()) continue;  // (a span whose kernels do not go through PPASR_LAUNCH)
    for (auto& pr : sp.ev) {
      HIP_TRY(hipEventSynchronize(pr.second));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
      total_ms_host[sp.cls] += ms;
    }
    launches_host[sp.cls] += 1;
  }
  return PPASR_OK;
}

const char* ppasr_kernel_class_name(int cls) {
  return (cls >= 0 && cls < PPASR_N_KERNEL_CLASSES) ? kKernelClassNames[cls] : "";
}

ppasr_status ppasr_ctc_collapse(const int32_t* frame_argmax, const float* frame_maxprob, const int32_t* frame_lens, int B,
                                int Tp, int blank, int32_t* tokens, int32_t* n_tokens, double* score, void* stream) {
  if (!frame_argmax || !frame_maxprob || !tokens || !n_tokens || !score) return fail(PPASR_EINVAL, "null argument");
  if (B <= 0 || Tp <= 0) return fail(PPASR_EINVAL, "empty batch");
  launch_ctc_collapse(frame_argmax, frame_maxprob, frame_lens, B, Tp, blank, tokens, n_tokens, score,
                      static_cast<hipStream_t>(stream));
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

ppasr_status ppasr_ctc_greedy(const float* probs, const int32_t* frame_lens, int B, int Tp, int V, int blank,
                              int32_t* tokens, int32_t* n_tokens, double* score, void* workspace, size_t workspace_bytes,
                              void* stream) {
  if (!probs || !tokens || !n_tokens || !score || !workspace) return fail(PPASR_EINVAL, "null argument");
  if (B <= 0 || Tp <= 0 || V <= 0) return fail(PPASR_EINVAL, "empty batch");
  const size_t n = (size_t)B * Tp;
  if (workspace_bytes < n * (sizeof(int32_t) + sizeof(float))) return fail(PPASR_ENOSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int32_t* fa = static_cast<int32_t*>(workspace);
  float* fp = reinterpret_cast<float*>(fa + n);
  launch_frame_argmax(probs, fa, fp, B * Tp, V, st);
  launch_ctc_collapse(fa, fp, frame_lens, B, Tp, blank, tokens, n_tokens, score, st);
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

// ---- hypothesis records of the data-parallel path (ppasr_amd/parallel.py): int32 rows
// tokens[cols] (-1 padded) | n_tokens | score (f64 as two words) [| utterance index].  One launch packs a batch's
// hypotheses into its rows of the rank's record, one launch restores the caller's utterance order after the all-gather --
// the record never passes through torch's indexing / elementwise kernels. ----
}  // extern "C"
namespace {
__global__ __launch_bounds__(256) void k_hyp_pack(const int32_t* __restrict__ tokens, long long token_stride, int L,
                                                  const int32_t* __restrict__ n, long long n_stride,
                                                  const double* __restrict__ score, long long score_stride,
                                                  const int32_t* __restrict__ index, int k, int32_t* __restrict__ rec,
                                                  int row0, int cols, int extra) {
  const int W = cols + extra;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)k * W) return;
  const int r = (int)(i / W), j = (int)(i - (long long)r * W);
  int32_t v;
  if (j < cols) {
    v = j < L ? tokens[r * token_stride + j] : -1;
  } else if (j == cols) {
    v = n[r * n_stride];
  } else if (j <= cols + 2) {
    const long long bits = __double_as_longlong(score[r * score_stride]);
    v = (int32_t)(j == cols + 1 ? (uint32_t)bits : (uint32_t)((unsigned long long)bits >> 32));
  } else {
    v = index ? index[r] : -1;
  }
  rec[(size_t)(row0 + r) * W + j] = v;
}
__global__ __launch_bounds__(256) void k_hyp_unpack(const int32_t* __restrict__ rec, const int64_t* __restrict__ order, int N,
                                                    int cols, int extra, int32_t* __restrict__ tokens, int32_t* __restrict__ n,
                                                    double* __restrict__ score, int32_t* __restrict__ index) {
  const int W = cols + extra;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)N * (cols + 1)) return;
  const int r = (int)(i / (cols + 1)), j = (int)(i - (long long)r * (cols + 1));
  const int32_t* src = rec + (size_t)(order ? order[r] : r) * W;
  if (j < cols) {
    tokens[(size_t)r * cols + j] = src[j];
  } else {
    n[r] = src[cols];
    const unsigned long long bits = (unsigned long long)(uint32_t)src[cols + 1] | ((unsigned long long)(uint32_t)src[cols + 2] << 32);
    score[r] = __longlong_as_double((long long)bits);
    if (index) index[r] = extra > 3 ? src[cols + 3] : -1;
  }
}
}  // namespace
extern "C" {

ppasr_status ppasr_hyp_pack(const int32_t* tokens, long long token_stride, int L, const int32_t* n_tokens, long long n_stride,
                            const double* score, long long score_stride, const int32_t* index, int k, int32_t* rec, int row0,
                            int cols, int extra, void* stream) {
  if (!tokens || !n_tokens || !score || !rec) return fail(PPASR_EINVAL, "null argument");
  if (k <= 0 || cols <= 0 || L < 0 || row0 < 0 || (extra != 3 && extra != 4)) return fail(PPASR_EINVAL, "hyp_pack: bad shape");
  const long long total = (long long)k * (cols + extra);
  PPASR_LAUNCH(k_hyp_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), tokens,
               token_stride, L, n_tokens, n_stride, score, score_stride, index, k, rec, row0, cols, extra);
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

ppasr_status ppasr_hyp_unpack(const int32_t* rec, const int64_t* order, int N, int cols, int extra, int32_t* tokens,
                              int32_t* n_tokens, double* score, int32_t* index, void* stream) {
  if (!rec || !tokens || !n_tokens || !score) return fail(PPASR_EINVAL, "null argument");
  if (N <= 0 || cols <= 0 || (extra != 3 && extra != 4)) return fail(PPASR_EINVAL, "hyp_unpack: bad shape");
  const long long total = (long long)N * (cols + 1);
  PPASR_LAUNCH(k_hyp_unpack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rec, order,
               N, cols, extra, tokens, n_tokens, score, index);
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

}  // extern "C"
