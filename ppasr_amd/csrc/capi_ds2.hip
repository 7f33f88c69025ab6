// capi_ds2.hip -- DeepSpeech2Model.get_encoder_out / get_encoder_out_chunk (ppasr/model_utils/deepspeech2/
// model.py:62-72) behind the C-ABI: weight packing and launch sequence of CRNNEncoder.forward
// (deepspeech2/encoder.py:61-104) + ctc softmax.
#include <algorithm>

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
size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }
}  // namespace

#define GETW(var, name, numel)                   \
  const float* var = get(name, (size_t)(numel)); \
  if (!var) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + get.missing)
#define UP(vec, dst) \
  if ((st = m->upload(vec, &(dst))) != PPASR_OK) return st
#define UP4(vec, dst) \
  if ((st = m->upload4(vec, &(dst))) != PPASR_OK) return st

// desc fields for DeepSpeech2: output_size = rnn_size, num_blocks = num_rnn_layers, causal = streaming
// (rnn_direction 'forward', deepspeech2/model.py:40) else 'bidirect'.
ppasr_status ds2_create(ppasr_model_s* m, BlobMap& sd) {
  const ppasr_model_desc& dsc = m->desc;
  const int F = dsc.input_dim, H = dsc.output_size, L = dsc.num_blocks, V = dsc.vocab_size;
  const int dirs = dsc.causal ? 1 : 2;
  if (H % 1024 != 0 || H > 2048) return fail(PPASR_EUNSUPPORTED, "deepspeech2: rnn_size must be 1024 or 2048");
  if (L < 1 || L > 16) return fail(PPASR_EUNSUPPORTED, "deepspeech2: bad num_rnn_layers");
  const int F2 = m->F2, C = 32;
  Getter get{sd, ""};
  ppasr_status st;
  Ds2W& W = m->ds2;
  const int G = dsc.use_gru ? 3 : 4;  // gate rows per hidden unit: nn.GRU (r, z, c) / nn.LSTM (i, f, g, o)
  W.H = H; W.dirs = dirs; W.n_layers = L; W.V = V; W.gates = G;
  W.Vpad = (V + 255) / 256 * 256;
  W.ldx = (C * F2 + 255) / 256 * 256;  // conv feature width padded to the GEMM's K granularity
  {
    GETW(mean, "encoder.global_cmvn.mean", F);
    GETW(istd, "encoder.global_cmvn.istd", F);
    GETW(c1w, "encoder.conv.conv.0.weight", C * 9);
    GETW(c1b, "encoder.conv.conv.0.bias", C);
    GETW(c2w, "encoder.conv.conv.2.weight", C * C * 9);
    GETW(c2b, "encoder.conv.conv.2.bias", C);
    UP(vec_of(mean, F), W.cmvn_mean);
    UP(vec_of(istd, F), W.cmvn_istd);
    std::vector<float> w1(9 * C), w2((size_t)9 * C * C);
    for (int c = 0; c < C; ++c)
      for (int j = 0; j < 9; ++j) w1[j * C + c] = c1w[c * 9 + j];
    for (int co = 0; co < C; ++co)
      for (int ci = 0; ci < C; ++ci)
        for (int j = 0; j < 9; ++j) w2[((size_t)j * C + ci) * C + co] = c2w[((size_t)co * C + ci) * 9 + j];
    UP(w1, W.c1_w);
    UP(vec_of(c1b, C), W.c1_b);
    UP(w2, W.c2_w);
    UP(vec_of(c2b, C), W.c2_b);
  }
  m->ds2_layers.resize(L);
  std::vector<Ds2WaveLayer> wave(L);           // unidirectional models: table of the wavefront path (k_lstm_wave)
  std::vector<float> prev_g, prev_b;           // LayerNorm of the previous layer (folded into this layer's W_ih there)
  for (int l = 0; l < L; ++l) {
    Ds2LayerW& Lw = m->ds2_layers[l];
    const int in_dim = l == 0 ? C * F2 : dirs * H;
    const int in_pad = l == 0 ? W.ldx : dirs * H;
    Lw.in_dim_padded = in_pad;
    const std::string p = "encoder.rnn." + std::to_string(l) + ".";
    const float* wih[2] = {nullptr, nullptr};
    const float* whh[2] = {nullptr, nullptr};
    const float* bih[2] = {nullptr, nullptr};
    const float* bhh[2] = {nullptr, nullptr};
    for (int d = 0; d < dirs; ++d) {
      const std::string sfx = d == 0 ? "_l0" : "_l0_reverse";
      wih[d] = get(p + "weight_ih" + sfx, (size_t)G * H * in_dim);
      whh[d] = get(p + "weight_hh" + sfx, (size_t)G * H * H);
      bih[d] = get(p + "bias_ih" + sfx, G * H);
      bhh[d] = get(p + "bias_hh" + sfx, G * H);
      if (!wih[d] || !whh[d] || !bih[d] || !bhh[d]) return fail(PPASR_EMISSING, "missing or mis-shaped weight: " + get.missing);
    }
    // one GEMM per layer: columns [d*4H, (d+1)*4H) = direction d's gate pre-activations; W[k][n] = weight_ih[n][k]
    const int N = dirs * G * H;
    UP4(pack_b(in_pad, N, [&](int k, int n) {
          if (k >= in_dim) return 0.f;
          const int d = n / (G * H), r = n % (G * H);
          return wih[d][(size_t)r * in_dim + k];
        }), Lw.w_ih);
    std::vector<float> bsum(N), hh((size_t)dirs * G * H * H);
    for (int d = 0; d < dirs; ++d) {
      // LSTM: both biases enter the gate pre-activation; GRU: b_hh stays with the recurrent product (k_gru_step)
      for (int r = 0; r < G * H; ++r) bsum[d * G * H + r] = bih[d][r] + (G == 4 ? bhh[d][r] : 0.f);
      std::memcpy(&hh[(size_t)d * G * H * H], whh[d], (size_t)G * H * H * sizeof(float));
    }
    UP(bsum, Lw.b_sum);
    UP(hh, Lw.w_hh);
    if (G == 3) {
      std::vector<float> bh((size_t)dirs * G * H);
      for (int d = 0; d < dirs; ++d) std::memcpy(&bh[(size_t)d * G * H], bhh[d], (size_t)G * H * sizeof(float));
      UP(bh, Lw.b_hh);
    }
    // fragment-ordered copy for the batched step kernel: per direction, packed column 32 t + 8 gate + u = row
    // gate * H + 8 t + u of weight_hh (every gate of units 8t .. 8t+7 in one 32-column tile)
    // (GRU: three gates r, z, c; the fourth slot of every tile is zero)
    {
      std::vector<float> pk;
      pk.reserve((size_t)dirs * 4 * H * H);
      for (int d = 0; d < dirs; ++d) {
        const float* wd = whh[d];
        std::vector<float> one = pack_b(H, 4 * H, [&](int k, int n) {
          const int t = n / 32, r = n % 32, gate = r / 8, u = r % 8;
          return gate < G ? wd[(size_t)(gate * H + 8 * t + u) * H + k] : 0.f;
        });
        pk.insert(pk.end(), one.begin(), one.end());
      }
      UP4(pk, Lw.w_hh_pk);
    }
    if (dirs == 1) {
      wave[l] = Ds2WaveLayer{Lw.w_hh_pk, nullptr, nullptr, nullptr};
      // packed column n = 32 t + 8 gate + u  <->  row gate * H + 8 t + u of the [G*H][.] weights; -1: the unused fourth gate
      // slot of a GRU tile
      auto rowof = [&](int n) { const int t = n / 32, r = n % 32; return r / 8 < G ? (r / 8) * H + 8 * t + (r % 8) : -1; };
      if (G == 3) {
        std::vector<float> bn(4 * H, 0.f);
        for (int n = 0; n < 4 * H; ++n)
          if (rowof(n) >= 0) bn[n] = bhh[0][rowof(n)];
        UP(bn, wave[l].bhh_n);
      }
      if (l > 0) {
        // W' = W_ih diag(gamma_{l-1}) in the gate-interleaved fragment order; s_n = its column sums;
        // c_n = W_ih beta_{l-1} + b_ih (+ b_hh: LSTM)  (k_lstm_wave applies the LayerNorm through mean / rstd of the raw row)
        const float* wd = wih[0];
        std::vector<float> sn(4 * H, 0.f), cn(4 * H, 0.f);
        for (int n = 0; n < 4 * H; ++n) {
          if (rowof(n) < 0) continue;
          const float* wr = wd + (size_t)rowof(n) * H;
          double a = 0.0, c0 = 0.0;
          for (int k = 0; k < H; ++k) {
            a += (double)prev_g[k] * (double)wr[k];
            c0 += (double)prev_b[k] * (double)wr[k];
          }
          sn[n] = (float)a;
          cn[n] = (float)(c0 + (double)bih[0][rowof(n)] + (G == 4 ? (double)bhh[0][rowof(n)] : 0.0));
        }
        UP4(pack_b(H, 4 * H, [&](int k, int n) { return rowof(n) >= 0 ? prev_g[k] * wd[(size_t)rowof(n) * H + k] : 0.f; }),
            wave[l].wih_pk);
        UP(sn, wave[l].s_n);
        UP(cn, wave[l].c_n);
      }
    }
    GETW(lg, "encoder.layernorm_list." + std::to_string(l) + ".weight", dirs * H);
    GETW(lb, "encoder.layernorm_list." + std::to_string(l) + ".bias", dirs * H);
    UP(vec_of(lg, dirs * H), Lw.ln_g);
    UP(vec_of(lb, dirs * H), Lw.ln_b);
    prev_g = vec_of(lg, dirs * H);
    prev_b = vec_of(lb, dirs * H);
  }
  if (dirs == 1 && H % 64 == 0) {
    void* d = nullptr;
    if (hipMalloc(&d, L * sizeof(Ds2WaveLayer)) != hipSuccess) return fail(PPASR_EHIP, "hipMalloc failed");
    m->allocs.push_back(d);
    if (hipMemcpy(d, wave.data(), L * sizeof(Ds2WaveLayer), hipMemcpyHostToDevice) != hipSuccess)
      return fail(PPASR_EHIP, "hipMemcpy failed");
    W.wave_tab = static_cast<const Ds2WaveLayer*>(d);
  }
  {
    GETW(cw, "decoder.ctc_lo.weight", (size_t)dirs * H * V);
    GETW(cb, "decoder.ctc_lo.bias", V);
    UP4(pack_b(dirs * H, W.Vpad, [&](int k, int n) { return n < V ? cw[(size_t)k * V + n] : 0.f; }), W.ctc_w);
    std::vector<float> cbp(W.Vpad, 0.f);
    std::memcpy(cbp.data(), cb, V * sizeof(float));
    UP(cbp, W.ctc_b);
  }
  return PPASR_OK;
}

struct Ds2Ws {
  size_t y1, x, gx, ya, yb, h0, h1, c, lens32, hbuf, cbuf, yring, part, part_floats, xbuf, pflag, total;  // float offsets
};
static Ds2Ws ds2_ws(const ppasr_model_s* m, int B, int T) {
  const Ds2W& W = m->ds2;
  const size_t T1 = (T - 1) / 2, Tp = (T1 - 1) / 2, M = (size_t)B * Tp;
  Ds2Ws w;
  size_t o = 0;
  w.y1 = o; o += al64((size_t)B * T1 * m->F1 * 32);
  w.x = o; o += al64(M * W.ldx);
  w.gx = o; o += al64(M * W.dirs * W.gates * W.H);
  w.ya = o; o += al64(M * W.dirs * W.H);
  w.yb = o; o += al64(M * W.dirs * W.H);
  w.h0 = o; o += al64((size_t)W.dirs * B * W.H);
  w.h1 = o; o += al64((size_t)W.dirs * B * W.H);
  w.c = o; o += al64((size_t)W.dirs * B * W.H);
  w.lens32 = o; o += al64(B);
  // wavefront path (unidirectional models): per-layer state / output rings
  const size_t Bp = (size_t)(B + 31) / 32 * 32;  // (fragment-ordered buffers hold whole 32-row tiles)
  w.hbuf = o; o += al64((size_t)W.n_layers * 2 * Bp * W.H);
  w.cbuf = o; o += al64((size_t)W.n_layers * B * W.H);
  w.yring = o; o += al64((size_t)W.n_layers * 2 * Bp * W.H);
  // K-slice partial sums of the dense layers when the launch is under-filled (few frames: single utterances)
  w.part_floats = M <= 512 ? (size_t)8 * M * std::max((size_t)W.gates * W.H, (size_t)W.Vpad) : 0;
  w.part = o; o += al64(w.part_floats);
  // persistent recurrence of single utterances (k_lstm_persist): exchange granules [2][dirs][H] x 8 bytes, abort flag
  w.xbuf = o; o += al64((size_t)4 * W.dirs * W.H);
  w.pflag = o; o += 64;
  w.total = o;
  return w;
}

extern "C" size_t ppasr_ds2_workspace_bytes(ppasr_handle h, int B, int T) {
  if (!h || h->desc.model_type != PPASR_MODEL_DEEPSPEECH2 || B <= 0 || T < 7) return 0;
  return ds2_ws(h, B, T).total * sizeof(float);
}

// states: [L*dirs][B][H] in the reference's box layout (layer-major, then direction), or NULL for zeros.
extern "C" ppasr_status ppasr_ds2_encode(ppasr_handle h, const float* feats, const int64_t* lens, int B, int T,
                                         const float* init_h, const float* init_c, float* probs, int64_t* out_lens,
                                         float* final_h, float* final_c, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  if (!h || !feats || !lens || !probs || !workspace) return fail(PPASR_EINVAL, "null argument");
  if (h->desc.model_type != PPASR_MODEL_DEEPSPEECH2) return fail(PPASR_EINVAL, "handle is not a deepspeech2 model");
  if (B <= 0 || T < 7) return fail(PPASR_EINVAL, "need B > 0 and T >= 7 frames");
  const Ds2W& W = h->ds2;
  const Ds2Ws wl = ds2_ws(h, B, T);
  if (workspace_bytes < wl.total * sizeof(float)) return fail(PPASR_ENOSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  const int F = h->desc.input_dim, T1 = (T - 1) / 2, F1 = h->F1, Tp = (T1 - 1) / 2, F2 = h->F2;
  const int M = B * Tp, H = W.H, dirs = W.dirs, G = W.gates;
  float *y1 = ws + wl.y1, *x = ws + wl.x, *gx = ws + wl.gx, *ya = ws + wl.ya, *yb = ws + wl.yb;
  float *h0 = ws + wl.h0, *h1 = ws + wl.h1, *c = ws + wl.c;
  float* part = wl.part_floats ? ws + wl.part : nullptr;
  int32_t* lens32 = reinterpret_cast<int32_t*>(ws + wl.lens32);
  launch_ds2_conv1(feats, W.cmvn_mean, W.cmvn_istd, W.c1_w, W.c1_b, y1, B, T, F, T1, F1, st);
  launch_ds2_conv2(y1, W.c2_w, W.c2_b, x, B, T1, F1, Tp, F2, W.ldx, st);
  launch_ds2_lens(lens, lens32, out_lens, B, Tp, st);
  const float* in = x;
  int in_ld = W.ldx;
  float* out = ya;
  const size_t sbytes = (size_t)dirs * B * H * sizeof(float);
  // (the matrix-core tiles have 32 rows whatever the batch: below 4 utterances the per-step kernels are the faster ones)
  // (the wavefront folds the input projections of layers >= 1 into the step: fewer dependent launches, but those
  //  projections then run on the step kernel's 32-row tiles instead of the dense GEMM; measured, 5 x 1024 LSTM, 5 s
  //  utterances: B = 32 6.6 ms against 9.2 per-step, B = 64 11.4 against 11.2, B = 128 21.9 against 21.0)
  // (round 4: a workgroup of the wavefront kernel takes up to 128 utterances -- 1 / 2 / 4 row tiles -- and streams its gate
  //  columns' weights once for all of them, so the wavefront wins at every batch size; nn.GRU stacks take it too.
  //  PPASR_DS2_WAVE_MAX_B: the per-step route above that batch size, for A/B measurements)
  static const int wave_max_b = getenv("PPASR_DS2_WAVE_MAX_B") ? atoi(getenv("PPASR_DS2_WAVE_MAX_B")) : (1 << 30);
  if (W.wave_tab && dirs == 1 && Tp > 0 && B >= 4 && B <= wave_max_b) {
    // ---- unidirectional stack: wavefront over (layer, time), Tp + L - 1 dependent launches (k_lstm_wave) ----
    const int L = W.n_layers;
    const Ds2LayerW& L0 = h->ds2_layers[0];
    launch_dense(x, W.ldx, L0.w_ih, L0.b_sum, gx, M, L0.in_dim_padded, G * H, G * H, G * H, st, 1.0f, part, wl.part_floats);
    float *hbuf = ws + wl.hbuf, *cbuf = ws + wl.cbuf, *yring = ws + wl.yring;
    const size_t BH = (size_t)B * H, BHp = (size_t)((B + 31) / 32) * 32 * H;  // row-major box / fragment-ordered slot
    for (int l = 0; l < L; ++l) {
      float* h_l = hbuf + (size_t)l * 2 * BHp;  // slot 0 = state before time 0 (fragment order, ds2_kernels.hip)
      HIP_TRY(hipMemsetAsync(h_l, 0, BHp * sizeof(float), st));
      if (init_h) launch_state_reorder(init_h + (size_t)l * BH, h_l, B, H, true, st);
      // (GRU: no cell state; the c box is handed through unchanged, deepspeech2/encoder.py:95-97)
      if (init_c) HIP_TRY(hipMemcpyAsync(cbuf + (size_t)l * BH, init_c + (size_t)l * BH, BH * sizeof(float), hipMemcpyDeviceToDevice, st));
      else HIP_TRY(hipMemsetAsync(cbuf + (size_t)l * BH, 0, BH * sizeof(float), st));
    }
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)M * H * sizeof(float), st));
    for (int s = 0; s < Tp + L - 1; ++s) {
      const int l_lo = s - (Tp - 1) > 0 ? s - (Tp - 1) : 0, l_hi = s < L - 1 ? s : L - 1;
      launch_lstm_wave(gx, W.wave_tab, hbuf, cbuf, yring, out, lens32, B, Tp, H, L, s, l_lo, l_hi - l_lo + 1, st, G == 3);
    }
    for (int l = 0; l < L; ++l) {
      if (final_h) launch_state_reorder(hbuf + ((size_t)l * 2 + (Tp & 1)) * BHp, final_h + (size_t)l * BH, B, H, false, st);
      if (final_c) HIP_TRY(hipMemcpyAsync(final_c + (size_t)l * BH, cbuf + (size_t)l * BH, BH * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    const Ds2LayerW& Ll = h->ds2_layers[L - 1];
    launch_ln_wide(out, Ll.ln_g, Ll.ln_b, M, H, st);
    launch_dense(out, H, W.ctc_w, W.ctc_b, probs, M, H, W.Vpad, W.V, W.V, st, 1.0f, part, wl.part_floats);
    launch_softmax_from_stats(probs, nullptr, nullptr, M, W.V, st);
    HIP_TRY(hipGetLastError());
    return PPASR_OK;
  }
  // (PPASR_DS2_PERSIST=0, read per call: the per-step kernels, for A/B measurements and the route-equality test)
  const char* persist_env = getenv("PPASR_DS2_PERSIST");
  const bool persist_shape = B == 1 && (G == 4 || G == 3) && H == 1024 && Tp > 0 && !(persist_env && persist_env[0] == '0') && lstm_persist_fits(H, dirs);
  const bool persist = persist_shape && h->ds2_persist_hold == 0;
  if (persist_shape && h->ds2_persist_hold > 0) --h->ds2_persist_hold;
  if (persist) {
    HIP_TRY(hipMemsetAsync(ws + wl.xbuf, 0, (size_t)4 * dirs * H * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(ws + wl.pflag, 0, 64 * sizeof(float), st));
  }
  for (int l = 0; l < W.n_layers; ++l) {
    const Ds2LayerW& Lw = h->ds2_layers[l];
    // gate pre-activations of all frames and both directions: [M][dirs*4H]; the step kernel wants [dirs][M][4H]
    // -> run one GEMM per direction into its own slab
    for (int d = 0; d < dirs; ++d)
      launch_dense(in, in_ld, Lw.w_ih + (size_t)d * (G * H / 32) * (Lw.in_dim_padded / 8) * 64, Lw.b_sum + d * G * H,
                   gx + (size_t)d * M * G * H, M, Lw.in_dim_padded, G * H, G * H, G * H, st, 1.0f, part, wl.part_floats);
    // initial states
    if (init_h) HIP_TRY(hipMemcpyAsync(h0, init_h + (size_t)l * dirs * B * H, sbytes, hipMemcpyDeviceToDevice, st));
    else HIP_TRY(hipMemsetAsync(h0, 0, sbytes, st));
    // GRU: there is no cell state; the c box is handed through unchanged (deepspeech2/encoder.py:95-97)
    if (init_c) HIP_TRY(hipMemcpyAsync(c, init_c + (size_t)l * dirs * B * H, sbytes, hipMemcpyDeviceToDevice, st));
    else HIP_TRY(hipMemsetAsync(c, 0, sbytes, st));
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)M * dirs * H * sizeof(float), st));
    float* hp = h0;
    float* hn = h1;
    // the matrix-core step needs H % 64 == 0 (8 waves x whole 8-wide k-groups); the VALU kernel handles the rest
    const bool mfma_step = (H % 64 == 0) && B >= 2;  // (one utterance: the VALU kernel's 5.8 us per step is the faster one)
    if (persist) {
      // one utterance, LSTM or GRU, H = 1024: the layer's recurrence as ONE launch with W_hh in registers (ds2_kernels.hip)
      launch_lstm_persist(gx, Lw.w_hh, Lw.b_hh, G == 3, h0, c, h1, out, lens32, Tp, H, dirs, reinterpret_cast<unsigned long long*>(ws + wl.xbuf),
                          1u + (unsigned int)l * (unsigned int)(Tp + 1), reinterpret_cast<int*>(ws + wl.pflag), st);
      hp = h1;
    } else
    for (int s = 0; s < Tp; ++s) {
      if (G == 3 && mfma_step) launch_gru_step_mfma(gx, Lw.w_hh_pk, Lw.b_hh, hp, hn, out, lens32, B, Tp, H, dirs, s, st);
      else if (G == 3) launch_gru_step(gx, Lw.w_hh, Lw.b_hh, hp, hn, out, lens32, B, Tp, H, dirs, s, st);
      else if (mfma_step) launch_lstm_step_mfma(gx, Lw.w_hh_pk, hp, hn, c, out, lens32, B, Tp, H, dirs, s, st);
      else launch_lstm_step(gx, Lw.w_hh, hp, hn, c, out, lens32, B, Tp, H, dirs, s, st);
      std::swap(hp, hn);
    }
    if (final_h) HIP_TRY(hipMemcpyAsync(final_h + (size_t)l * dirs * B * H, hp, sbytes, hipMemcpyDeviceToDevice, st));
    if (final_c) HIP_TRY(hipMemcpyAsync(final_c + (size_t)l * dirs * B * H, c, sbytes, hipMemcpyDeviceToDevice, st));
    launch_ln_wide(out, Lw.ln_g, Lw.ln_b, M, dirs * H, st);
    in = out;
    in_ld = dirs * H;
    out = (out == ya) ? yb : ya;
  }
  launch_dense(in, in_ld, W.ctc_w, W.ctc_b, probs, M, dirs * H, W.Vpad, W.V, W.V, st, 1.0f, part, wl.part_floats);
  launch_softmax_from_stats(probs, nullptr, nullptr, M, W.V, st);
  HIP_TRY(hipGetLastError());
  if (persist) {
    // the persistent launches wait for one another's workgroups: if one gave up (the chip was not free for the whole
    // grid), say so instead of returning what it left -- this call synchronises, re-runs on the per-step kernels, and the
    // handle stays on them for the next 64 .. 1 024 calls (capi_internal.h ds2_persist_hold)
    int gave_up = 0;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(&gave_up, ws + wl.pflag, sizeof(int), hipMemcpyDeviceToHost));
    if (gave_up) {
      h->ds2_persist_hold = 64 << (h->ds2_persist_giveups < 4 ? h->ds2_persist_giveups : 4);
      ++h->ds2_persist_giveups;
      return ppasr_ds2_encode(h, feats, lens, B, T, init_h, init_c, probs, out_lens, final_h, final_c, workspace, workspace_bytes,
                              stream);
    }
  }
  return PPASR_OK;
}

// Test hook (not used by any product path): holds `n_workgroups` whole CUs for `milliseconds` on `stream` (ds2_kernels.hip
// k_occupy).  tests/test_deepspeech2_gpu.py forces the persistent recurrence's give-up path with it.
extern "C" ppasr_status ppasr_debug_occupy_cus(int n_workgroups, int milliseconds, void* stream) {
  if (n_workgroups <= 0 || milliseconds <= 0 || milliseconds > 5000) return fail(PPASR_EINVAL, "occupy: bad arguments");
  launch_occupy(n_workgroups, milliseconds, nullptr, static_cast<hipStream_t>(stream));
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}
