// capi_internal.h -- definitions shared by the C-ABI translation units (capi*.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ppasr_hip.h"
#include "conformer_kernels.h"
#include "launch.h"
#include "ctc_beam.h"

using namespace ppasr;

#include "squeezeformer_kernels.h"
#include "ds2_kernels.h"

std::string& ppasr_err_slot();  // thread-local error string (defined in capi.hip)
inline ppasr_status fail(ppasr_status s, const std::string& msg) {
  ppasr_err_slot() = msg;
  return s;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return fail(PPASR_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

struct Blob {
  const float* p;
  int ndim;
  int64_t shape[4];
  size_t numel() const {
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    return n;
  }
};

// Fragment-order packing of a [K][N] weight (y = x W): for 32-column tile nt, 8-wide k-group g,
// lane l: 4 consecutive floats = W[8g + 4(l>>5) + 0..3][32 nt + (l&31)].  One wave-level
// global_load_dwordx4 then yields the B operands of 4 successive v_mfma_f32_32x32x2_f32.
template <typename Acc>
inline std::vector<float> pack_b(int K, int N, Acc w) {
  const int n_tiles = (N + 31) / 32, G = K / 8;
  std::vector<float> out((size_t)n_tiles * G * 256, 0.f);
  for (int nt = 0; nt < n_tiles; ++nt)
    for (int g = 0; g < G; ++g)
      for (int l = 0; l < 64; ++l) {
        int n = nt * 32 + (l & 31);
        if (n >= N) continue;
        float* dst = &out[(((size_t)nt * G + g) * 64 + l) * 4];
        for (int j = 0; j < 4; ++j) dst[j] = w(8 * g + 4 * (l >> 5) + j, n);
      }
  return out;
}

typedef std::unordered_map<std::string, Blob> BlobMap;

// Efficient-Conformer: bit i = layer i is a stride-2 layer (ppasr_model_desc::stride_layer_mask, or the single
// stride_layer_idx of the shipped configuration)
inline unsigned eff_stride_mask(const ppasr_model_desc& d) {
  if (d.model_type != PPASR_MODEL_EFFICIENT_CONFORMER) return 0u;
  if (d.stride_layer_mask != 0) return (unsigned)d.stride_layer_mask;
  return d.stride_layer_idx >= 0 ? (1u << d.stride_layer_idx) : 0u;
}
inline int eff_strides_before(const ppasr_model_desc& d, int layer) {  // stride layers among layers 0 .. layer - 1
  return __builtin_popcount(eff_stride_mask(d) & ((1u << layer) - 1u));
}

struct ppasr_model_s {
  ppasr_model_desc desc;
  int F1, F2;
  // general layer route (capi_generic.hip): CTC head as a dense layer over the padded vocabulary
  const f32x4* gen_head_w = nullptr;
  const float* gen_head_b = nullptr;
  int gen_vpad = 0;
  int F3 = 0;  // conv2d8: feature bins behind the third conv (F2 behind the second); 0 otherwise
  // front-end geometry for T input frames (Conv2dSubsampling4 / 6 / 8, subsampling.py): frames behind conv1, behind the
  // intermediate conv (conv2d8 only, else 0) and encoder frames; F_last = feature bins entering the linear layer
  struct Front {
    int T1, T2, Tp;
  };
  int sub_rate() const { return desc.input_layer == 1 ? 1 : desc.input_layer ? desc.input_layer : 4; }
  int F_last() const { return desc.input_layer == 8 ? F3 : F2; }
  int min_frames() const { return desc.input_layer == 1 ? 1 : desc.input_layer == 6 ? 11 : desc.input_layer == 8 ? 15 : 7; }
  Front front_dims(int T) const {
    if (desc.input_layer == 1) return Front{0, 0, T};  // LinearNoSubsampling: one encoder frame per feature frame
    Front f{(T - 1) / 2, 0, 0};
    if (desc.input_layer == 6) {
      f.Tp = (f.T1 - 5) / 3 + 1;
    } else if (desc.input_layer == 8) {
      f.T2 = (f.T1 - 1) / 2;
      f.Tp = (f.T2 - 1) / 2;
    } else {
      f.Tp = (f.T1 - 1) / 2;
    }
    return f;
  }
  // ---- general layer route (capi_generic.hip): any width that is a multiple of 256 and every ConformerEncoder
  // constructor option (ppasr_model_desc::options, input_layer = linear, any conv kernel size) ----
  bool generic = false;
  struct GenOpts {
    int pos = 0;  // PPASR_OPT_POS_*
    bool post_norm = false, concat_after = false, macaron = true, use_cnn = true;
    int act = 0;  // PPASR_ACT_*
    bool sq_pre_norm = false;  // Squeezeformer normalize_before = True
  } gen;
  struct GenLayerX {  // what LayerW has no slot for
    const f32x4* wcat = nullptr;  // concat_linear [2d][d], packed
    const float* bcat = nullptr;
  };
  std::vector<GenLayerX> gen_x;
  const float* pe_dev = nullptr;    // positional table [max_len][d] (abs_pos adds it to the embedded frames)
  const float* zero_vec = nullptr;  // [d] zeros: pos_bias_u / _v and the one-row "positional table" of MultiHeadedAttention
  const float *lin_ln_g = nullptr, *lin_ln_b = nullptr;  // LinearNoSubsampling's LayerNorm (eps 1e-12)
  int lin_kpad = 0;                                      //   and its input width padded to whole 256-wide K chunks
  std::vector<void*> allocs;
  FrontW front;
  std::vector<LayerW> layers;
  std::vector<int> layer_ks;     // depthwise kernel size per layer (Efficient-Conformer halves it after the stride layer)
  std::vector<int> layer_group;  // 1, or 3 on grouped-attention layers
  HeadW head;
  // Squeezeformer (model_type == PPASR_MODEL_SQUEEZEFORMER)
  std::vector<SqLayerW> sq_layers;
  SqReduceW sq_reduce{};
  const f32x4* sq_wrec = nullptr;
  const float* sq_brec = nullptr;
  const float *preln_g = nullptr, *preln_b = nullptr;
  // DeepSpeech2 (model_type == PPASR_MODEL_DEEPSPEECH2)
  Ds2W ds2{};
  // single utterances: the recurrence of a layer as one persistent launch.  It needs every workgroup of its grid resident at
  // once; when a launch gives up (the chip was shared) the next `ds2_persist_hold` calls take the per-step kernels, then the
  // route is tried again (hold doubles with every give-up, 64 .. 1 024 calls)
  int ds2_persist_hold = 0, ds2_persist_giveups = 0;
  std::vector<Ds2LayerW> ds2_layers;
  float* taps = nullptr;
  size_t taps_floats = 0;
  // optional per-kernel timing (bench.py roofline leg): one event pair per launch on the caller's stream
  bool prof = false;
  int front_fused = -1;  // ppasr_set_front_fused: -1 / 1 = conv1 + conv2 of the 4x front end as one launch, 0 = two launches
  int ffn_split = -1;  // ppasr_set_ffn_split: -1 = by grid size, 0 = never, 2 / 4 / 8 = always that many slices
  bool skip_padding = false;  // ppasr_set_skip_padding: ragged batches compute only the rows valid outputs depend on
  int gemm_mode = 0;          // ppasr_set_gemm_mode: PPASR_GEMM_F32 / PPASR_GEMM_F16X3 (feed-forward GEMMs, csrc/h3.h)
  std::vector<LayerW> layers_h3;  // layers[] with the FFN weight pointers replaced by their fp16 x3 re-packing
  const f32x4* conv2_w_h3 = nullptr;  // the 4x front end's second convolution, re-packed likewise
  const f32x4* embed_w_h3 = nullptr;  // ... and its input projection
  const f32x4* head_w_h3 = nullptr;   // the CTC head's weight [256][32 * n_tiles]
  std::vector<SqLayerW> sq_layers_h3;  // Squeezeformer: sq_layers[] with the two feed-forward modules' weights re-packed
  int gemm_coverage = 0;      // PPASR_GEMM_COVERS_* of the current mode (ppasr_gemm_coverage)
  // range guard of the fp16 x3 mode (csrc/h3.h, ppasr_set_gemm_guard / ppasr_gemm_guard_stats)
  bool gemm_guard = true;
  static constexpr int kGuardN = 5;                 // translation units with fp16 x3 kernels (h3.h: one event counter each)
  unsigned int* guard_ctr[kGuardN] = {};            // device addresses of their counters
  unsigned int* guard_dev = nullptr;                // [kGuardN counters before the call | kGuardN after]
  unsigned int* guard_host = nullptr;               // pinned mirror
  unsigned int guard_seen[kGuardN] = {};            // counter values this handle last read
  long long guard_fallbacks = 0, guard_events = 0;
  int row_block = -1;         // ppasr_set_row_block: -1 = by grid size, 32 / 16 / kW16 = always that block form (rbt.h)
  std::vector<int64_t> lens_hint;  // ppasr_set_lengths_hint: host copy of the batch's lengths (route selection only)
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct Span { int cls; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; };  // one pair per kernel launched inside the span
  std::vector<Span> spans;

  ppasr_status upload(const std::vector<float>& v, const float** out) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, v.size() * sizeof(float)));
    allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = static_cast<const float*>(d);
    return PPASR_OK;
  }
  ppasr_status upload4(const std::vector<float>& v, const f32x4** out) {
    const float* p = nullptr;
    ppasr_status s = upload(v, &p);
    *out = reinterpret_cast<const f32x4*>(p);
    return s;
  }
  hipEvent_t next_event() {
    if (ev_used == ev_pool.size()) {
      hipEvent_t e;
      (void)hipEventCreate(&e);
      ev_pool.push_back(e);
    }
    return ev_pool[ev_used++];
  }
  ~ppasr_model_s() {
    for (void* p : allocs) (void)hipFree(p);
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    if (guard_host) (void)hipHostFree(guard_host);
  }
};


// streaming state of one session (capi_stream.hip; the general route of capi_generic.hip shares it)
struct ppasr_stream_s {
  ppasr_model_s* m;
  int D;        // model width = row stride of the caches (256 on the fused route)
  int cap;      // key capacity per layer (frames)
  int cache_t;  // cached key/value frames of the full-rate layers (cache_t1 in the reference)
  int cache_r;  // frames held by the half-rate layers
  int offset;   // encoder-output frames emitted so far (the reference's `offset` argument)
  int lo;       // longest conv left context = cnn_module_kernel - 1
  float *kc, *vc;   // [L][cap][D]
  float* xh_hist;   // [L][lo][D]  conv-module input history
  float* g_hist;    // [L][lo][256] GLU(pointwise_conv1(history)) of every layer, recomputed at the start of each chunk (fused route)
  HistLayer* hist_tab;  // device [L]: per-layer pointwise_conv1 weights / history rows for that launch (fused route)
  int* ticket = nullptr;  // device [16], zero between launches: arrival counters of the feed-forward slices that join in-kernel
};

// the reference's shape arithmetic for one chunk (capi_stream.hip: plan_chunk)
struct ChunkPlan {
  int c;        // full-rate frames of this chunk
  int c_r;      // half-rate frames
  int used_r;   // half-rate cache frames that take part
  int T2, T2_r; // keys of the full-rate / half-rate layers
  int ncs;      // next_cache_start
  int pos0;     // position of key 0
};

struct WsLayout {
  size_t y1, y2, xa, xb, xc, qkv, ctx, g, rmax, rsum, fa, fp, xs, vt, total;  // offsets in floats
  int vt_stride;  // row stride of the transposed values (fused attention route)
};
WsLayout ws_layout(const ppasr_model_s* m, int B, int T);
// hidden-dimension slices per row block for M rows (1 = the fused kernels), see ppasr_set_ffn_split
int ffn_split_for(const ppasr_model_s* m, int M);
int wide_slices_for(const ppasr_model_s* m, int M, int units);  // embed K chunks / head tile groups on the split route
// rows per workgroup (32 or 16) for a row-block launch over B utterances of Tcur rows each: 16 when the rows that will
// actually be computed fill at most half of the chip as 32-row blocks.  With skip_padding the computed rows are
// sum_b min(Tcur, ceil(len_b / mul) + slack) -- known on the host only through ppasr_set_lengths_hint; without a hint the
// padded count decides.
int row_block_for(const ppasr_model_s* m, int B, int Tcur, int mul, int slack, bool skip);
bool block_tables_enabled();  // active-block lists for ragged layer kernels (PPASR_BLOCK_TABLE=0: padded grids; A/B switch)
// the 4x front end as one launch (front_fused.hip)?  ppasr_set_front_fused, then PPASR_CONV12=0 (A/B switch)
bool conv12_enabled(const ppasr_model_s* m);
ppasr::LayerW sq_conv_view(const ppasr::SqLayerW& W);  // capi_squeezeformer.hip

ppasr_status upload_pe_table(ppasr_model_s* m, BlobMap& sd, const float** pe_dev);

// general Conformer layer route (widths 512 / 768 / 1024, non-default constructor options): capi_generic.hip
size_t generic_ws_floats(const ppasr_model_s* m, int B, int T);
ppasr_status generic_encode(ppasr_model_s* h, const float* feats, const int64_t* lens, int B, int T, float* probs,
                            float* logits, int32_t* frame_argmax, float* frame_maxprob, float* ws, hipStream_t st);
// one chunk of ConformerEncoder / EfficientConformerEncoder.forward_chunk (conformer/encoder.py:208-283,
// efficient_conformer/encoder.py:266-393) on the general route: the caches of `s` hold cache_t (cache_r behind the stride
// layer) frames, key 0 sits at positional row p.pos0; appends this chunk's keys / values and conv inputs
ppasr_status generic_chunk(ppasr_stream_s* s, const ChunkPlan& p, const float* feats, int T, float* probs,
                           int32_t* frame_argmax, float* frame_maxprob, float* ws, hipStream_t st, int* frames_out);
// SqueezeformerEncoder.forward_chunk (squeezeformer/encoder.py:260-381) on the general route
ppasr_status generic_sq_chunk(ppasr_stream_s* s, const ChunkPlan& p, const float* feats, int T, float* probs,
                              int32_t* frame_argmax, float* frame_maxprob, float* ws, hipStream_t st);
hipError_t configure_generic_kernels();
// SqueezeformerEncoder.forward (squeezeformer/encoder.py:172-236) on the general route (encoder_dim 512 / 768 / 1024)
ppasr_status generic_sq_encode(ppasr_model_s* h, const float* feats, const int64_t* lens, int B, int T, float* probs,
                               float* logits, int32_t* frame_argmax, float* frame_maxprob, float* ws, hipStream_t st);

// model-family back ends
ppasr_status ds2_create(ppasr_model_s* m, BlobMap& sd);
ppasr_status squeezeformer_create(ppasr_model_s* m, BlobMap& sd, const float* pe_dev);
ppasr_status squeezeformer_encode(ppasr_model_s* h, const float* feats, const int64_t* lens, int B, int T, float* probs,
                                  float* logits, int32_t* frame_argmax, float* frame_maxprob, float* ws,
                                  const WsLayout& wl, hipStream_t st);
