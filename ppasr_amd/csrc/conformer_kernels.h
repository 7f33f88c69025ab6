// conformer_kernels.h -- launch interface between the C-ABI (capi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rowblock.h"

namespace ppasr {

// Device pointers of one ConformerEncoderLayer (conformer/encoder.py:286-344), re-packed:
// dense weights in MFMA fragment order (see pack_b in capi.hip), vectors as plain float arrays.
struct LayerW {
  const float *ln_mac_g, *ln_mac_b, *ln_mha_g, *ln_mha_b, *ln_conv_g, *ln_conv_b;
  const float *ln_ff_g, *ln_ff_b, *ln_fin_g, *ln_fin_b, *ln_cm_g, *ln_cm_b;
  float cm_eps;  // conv-module norm: LayerNorm epsilon (1e-5), or < 0: ln_cm_g / ln_cm_b are a folded BatchNorm's scale / shift
  const f32x4 *ffm_w1, *ffm_w2, *ff_w1, *ff_w2, *wqkv, *wo, *pw1, *pw2;
  const float *ffm_b1, *ffm_b2, *ff_b1, *ff_b2, *bqkv, *bo, *pw1_b, *pw2_b;
  const float *dw_w;     // [k][256] tap-major depthwise weights
  const float *dw_b;     // [256]
  const float *glu_pad;  // [256] GLU(pointwise_conv1(0)) = value of a zero-padded frame after GLU
  const float *pos_u, *pos_v;  // [256] = [h][dk]
  const float *ptab;     // [max_len][256] linear_pos(pe)  (weight-only, folded at create time)
};

struct FrontW {
  const float *cmvn_mean, *cmvn_istd;  // [F]
  const float *conv1_w, *conv1_b;      // [9][256] tap-major, [256]
  const f32x4 *conv2_w;                // packed, K = k*k*256 ordered (kh,kw,cin)
  const float *conv2_b;
  int conv2_k = 3, conv2_s = 2;        // 3, 2 (conv2d / conv2d8, Squeezeformer) or 5, 3 (conv2d6)
  const f32x4 *conv3_w = nullptr;      // conv2d8 only: third 3x3 / 2 conv, packed like conv2_w
  const float *conv3_b = nullptr;
  const f32x4 *embed_w;                // packed, K = f2*256 ordered (f', c)
  const float *embed_b;
};

struct HeadW {
  const float *ln_g, *ln_b;  // encoder.after_norm
  const f32x4 *w;            // packed [256][Vpad]
  const float *b;            // [Vpad]
  int V, n_tiles;
};

// One entry per active session of a multi-session streaming call (ppasr_encode_chunk_group): which cache slot it
// uses, how many frames that cache holds and the positional-table row of its first key.
struct SessDesc {
  int sess;     // slot in the group's cache arrays
  int cache_t;  // cached frames before this chunk
  int pos0;     // offset - cache_t  (encoder.py:253)
  int pad;
};

// Attention operands: queries / keys / values may live in different buffers (streaming reads K/V from
// the per-layer device caches).  Row strides in floats; utterance b starts at row b*T1 (q, ctx) / b*T2 (k, v).
struct AttnArgs {
  const float* q;
  int q_stride;
  const float* k;
  int k_stride;
  const float* v;
  int v_stride;
  int T1, T2;           // query / key TOKENS per utterance (= frames, or ceil(frames/3) with grouped attention)
  int pos0;             // position of key 0 in the positional table (encoder.py:253: offset - cache_t1)
  const int64_t* lens;  // feature lengths for the key-padding mask, or nullptr (streaming: no mask)
  float* ctx;           // [B*T1][256]
  const float *pos_u, *pos_v;  // [256] = [h][dk]
  const float* ptab;           // [max_len][256] projected positional table of this layer
  int pos_stride;       // key j uses table row pos0 + j*pos_stride (2 on Squeezeformer's time-reduced layers)
  int mask_mul;         // key j is PAD iff mask_mul*j >= len (4; 8 on time-reduced layers; x3 when grouped)
  int q_frames, kv_frames;  // valid frames behind the query / key tokens (== T1 / T2 unless grouped)
  int group;            // 1, or 3 = GroupedRelPositionMultiHeadedAttention (pos_u / pos_v are then [h][192])
  // multi-session streaming (plain heads only): utterance b = session sess[b]: keys / values at k + sess*sess_stride,
  // T2 = cache_t + T1 keys, positional rows from pos0; nullptr otherwise
  const SessDesc* sess;
  long long sess_stride;
  // ragged batches: 0 = every query row / key block is computed; n > 0 = query rows behind the valid frames + (n - 1)
  // are skipped and the key loop stops after the last valid key (masked keys contribute exact zeros either way)
  int pad_skip;
  // fused route (k_attn_out_glu) only: the values in MFMA-fragment order (written by the QKV stage, see VtOut) with
  // room for vt_stride rows; a.v is not read then
  const float* vt = nullptr;
  int vt_stride = 0;
  // model width (row stride of ptab / ctx, h * 64 <= dm); 256 everywhere except the general layer route (capi_generic.hip),
  // which runs k_attention<64> with 8 heads on 512-wide activations
  int dm = 256;
};
// Where the QKV stage puts the values when the layer's attention runs fused: in the order the attention's P V MFMAs
// consume them, [8 slabs of 32 columns][stride / 8 row octets][64 lanes = column + 32 * (row quad)][4 rows] -- 1 KiB
// contiguous per wave store / load; stride (rows of the batch) >= 32 * row blocks + 64, a multiple of 8.
// nullptr: row-major in qkv
struct VtOut {
  float* vt = nullptr;
  int stride = 0;
  // fp16 x3 mode, fused attention: the K third of every qkv row as [hi: 256 fp16 | lo: 256 fp16] of 2^4 K (h3.h) instead of
  // 256 floats (QkStoreTailH3 writes it, attn_out_glu_body<true> reads it)
  int k_h3 = 0;
};

// dynamic-LDS request of a ragged (PadSkip) launch: more than half of a CU's LDS, so that workgroups do not share a CU
// (see the comment at ragged_lds in front_kernels.hip); kernels launched with it set kLdsExclusive as their maximum
constexpr size_t kLdsExclusive = 82 * 1024;
size_t ragged_lds(size_t lds, const PadSkip& ps, int n_blocks);

// attention_kernels.hip: the stand-alone attention launch (transposed flash attention, keys split over the waves of a
// workgroup); false = configuration it does not take (the caller falls back to k_attention)
bool launch_attention_t(const AttnArgs& a, int B, int H, hipStream_t st);
hipError_t configure_attention_kernels();

// the list of active R-row blocks of a ragged batch (rowblock.h PadSkip::tab): 1 + ceil(M / R) ints at `tab`
void launch_block_table(const PadSkip& ps, int M, int R, int* tab, hipStream_t st);

// front_fused.hip: conv1 + conv2 of Conv2dSubsampling4 as one launch (y1 is never written); same results bit for bit
bool conv12_supported(const FrontW& fw, int F, int F2);
void launch_conv12(const float* feats, const FrontW& fw, float* y2, int B, int T, int F, int Tp, int F2, hipStream_t st,
                   const PadSkip& ps_frames, int* tile_scratch);
hipError_t configure_front_fused_kernels();

// conformer_kernels_t.hip: the layer kernels on 16-row blocks (under-filled launches; values row-major in qkv)
void launch_ffn_qkv_16(const float* x_in, float* x1, float* qkv, const LayerW& w, int M, int n_chunks, hipStream_t st,
                       const PadSkip& ps);
void launch_out_glu_16(const float* ctx, const float* x1, float* x2, float* g, const LayerW& w, const int64_t* lens, int M,
                       int Tp, int mask_mul, hipStream_t st, const PadSkip& ps);
void launch_out_glu_split_16(const float* ctx, const float* x1, float* x2, float* g, float* xhat, const LayerW& w,
                             const int64_t* lens, int M, int Tp, int mask_mul, hipStream_t st, const PadSkip& ps,
                             float* hist = nullptr, int lo = 0);  // split route, <= 16 rows; hist: see HistMove
void launch_shift_caches(float* kc, float* vc, long long layer_stride, int D, int n_layers, int from_full, int keep_full,
                         int from_half, int keep_half, unsigned long long half_mask, hipStream_t st);  // stream_kernels.hip
void launch_oproj_ln_16(const float* ctx, const float* x1, float* x2_sink, float* xhat_out, const LayerW& w, int M,
                        hipStream_t st, const PadSkip& ps = PadSkip{});  // xhat_out = LN(x1 + ctx Wo + bo): w.wo / bo / ln_conv_g / _b
// (lens != nullptr: PAD frames of the batch read w.glu_pad -- Squeezeformer's batched launches)
void launch_pw1_glu_cols_16(const float* x, float* g, const LayerW& w, int M, hipStream_t st, float* hist = nullptr, int lo = 0,
                            const float* hist_scale = nullptr, const float* hist_bias = nullptr, const PadSkip& ps = PadSkip{},
                            const int64_t* lens = nullptr, int Tp = 1, int mask_mul = 1);
bool conv_ffn_16_supported(int ksize, int Tp);
void launch_conv_ffn_16(const float* g, const float* x2, float* x_out, const LayerW& w, const int64_t* lens, int M, int Tp,
                        int n_chunks, int ksize, int mask_mul, const LayerW* next, float* x1_next, float* qkv_next,
                        hipStream_t st, bool causal, const PadSkip& ps);
// ... and on 32 rows x 16 waves (full launches): drop-in for launch_ffn_qkv / launch_out_glu / launch_conv_ffn
void launch_ffn_qkv_w16(const float* x_in, float* x1, float* qkv, const LayerW& w, int M, int n_chunks, hipStream_t st,
                        const PadSkip& ps, VtOut vt);
void launch_out_glu_w16(const float* ctx, const float* x1, float* x2, float* g, const LayerW& w, const int64_t* lens, int M,
                        int Tp, int mask_mul, hipStream_t st, const PadSkip& ps);
void launch_conv_ffn_w16(const float* g, const float* x2, float* x_out, const LayerW& w, const int64_t* lens, int M, int Tp,
                         int n_chunks, int ksize, int mask_mul, const LayerW* next, float* x1_next, float* qkv_next,
                         hipStream_t st, bool causal, const PadSkip& ps, VtOut vt_next);
hipError_t configure_conformer_t_kernels();

// ---- launchers (all asynchronous on `st`) ----
void launch_posproj(const float* pe, const float* wpos /*[d][d] in,out*/, const float* bpos_or_null, float* ptab,
                    int max_len, hipStream_t st, int d = 256);
// every row-block launcher takes an optional PadSkip (rowblock.h): default = compute all rows
void launch_conv1(const float* feats, const FrontW& fw, float* y1, int B, int T, int F, int T1, int F1, hipStream_t st,
                  const PadSkip& ps = PadSkip{}, int channels = 256);
// one k x k / stride-s 256 -> 256 channel conv + ReLU of the front end on NHWC activations (implicit GEMM)
void launch_conv_stage(const float* y_in, const f32x4* w, const float* bias, float* y_out, int B, int T_in, int F_in,
                       int T_out, int F_out, int k, int s, hipStream_t st, const PadSkip& ps_frames = PadSkip{},
                       int channels = 256, int* tile_scratch = nullptr,  // channels: 256, or a multiple of it (general layer route)
                       bool h3 = false,  // h3: w is the fp16 x3 re-packing (launch_repack_h3), the units run on that route
                       float* part = nullptr, size_t part_floats = 0);  // scratch: under-filled launches split the taps (K)
// floats of `part` that let launch_conv_stage split a launch of M output rows over its K chunks (0: it would not split)
size_t conv_stage_part_floats(int M, int channels = 256);
// tile_scratch (ragged batches): device scratch of B + 2 ints for the active-tile table (k_tile_prefix)
void launch_conv2(const float* y1, const FrontW& fw, float* y2, int B, int T1, int F1, int Tp, int F2, hipStream_t st,
                  const PadSkip& ps = PadSkip{}, int* tile_scratch = nullptr, const f32x4* w_h3 = nullptr, float* part = nullptr,
                  size_t part_floats = 0);
// scale_before_bias: Squeezeformer scales the 4864-wide conv output by sqrt(d) BEFORE input_proj
// (squeezeformer/subsampling.py:66-67) -> acc*scale + b ; Conformer: (acc + b)*scale (embedding.py:112)
// k_slices > 1 (under-filled launches): the contraction is split over that many workgroups per row block, partial sums
// in part (k_slices * M * 256 floats)
void launch_embed(const float* y2, const FrontW& fw, float* x0, int M, int K, float xscale, bool scale_before_bias,
                  hipStream_t st, const PadSkip& ps = PadSkip{}, int k_slices = 1, float* part = nullptr,
                  const f32x4* w_h3 = nullptr);  // w_h3: fw.embed_w re-packed for the fp16 x3 route (full launches take it)
void launch_dense(const float* a, int lda, const f32x4* w, const float* bias, float* out, int M, int K, int n_cols_padded,
                  int ldc, int n_valid, hipStream_t st, float scale = 1.0f, float* part = nullptr, size_t part_floats = 0);  // out = (a W + bias) * scale
// h3: the feed-forward modules on the fp16 x3 route (csrc/h3.h); w (and *next) must then be the layers' h3 views
void launch_ffn_qkv(const float* x_in, float* x1, float* qkv, const LayerW& w, int M, int n_chunks, hipStream_t st,
                    const PadSkip& ps = PadSkip{}, VtOut vt = VtOut{}, bool h3 = false);
// fp32 fragment-packed weight (pack_b: n_tiles x G k-groups x 1 KiB) -> the fp16 x3 packing of csrc/h3.h, same size
void launch_repack_h3(const f32x4* src, f32x4* dst, int n_tiles, int G, unsigned int* ovf, hipStream_t st);  // ovf: device word counting weights beyond the range
inline bool conv_ffn_h3_supported(int ksize) { return ksize == 15 || ksize == 7; }
void launch_split_rows_h3(const float* src, float* dst, long long n_rows, unsigned int* ovf, hipStream_t st);  // [n][256] f32 -> [n][hi 256 | lo 256] fp16 of 2^4 x
unsigned int* conformer_h3_ovf_counter();  // device address of conformer_kernels.hip's range-guard counter (h3.h)
unsigned int* front_h3_ovf_counter();      // ... front_kernels.hip's (conv2 / input projection in the fp16 x3 mode)
unsigned int* ctc_head_h3_ovf_counter();   // ... ctc_head_kernels.hip's
unsigned int* split_route_h3_ovf_counter();  // ... split_route_kernels.hip's
// per-file parts of configure_kernels() (dynamic-LDS limits of the kernels each translation unit owns)
hipError_t configure_front_kernels();
hipError_t configure_stream_kernels();
hipError_t configure_split_route_kernels();
hipError_t configure_ctc_head_kernels();
void launch_attention(const AttnArgs& a, int B, int H, hipStream_t st);
// one streaming session: the layer's conv-module input history [lo][256] is to move on by the chunk's M rows of xhat; a
// launch that can do it on the side sets `done` (otherwise the caller runs launch_hist_update)
struct HistMove {
  float* hist;
  int lo;
  bool done;
};
void launch_out_glu(const float* ctx, const float* x1, float* x2, float* g, float* xhat_out, const LayerW& w,
                    const int64_t* lens, int M, int Tp, int mask_mul, hipStream_t st, const PadSkip& ps = PadSkip{},
                    float* split_xhat = nullptr,  // != nullptr: two launches (under-filled grids), M*256 floats of scratch
                    bool h3 = false,              // the units on the fp16 x3 route (w: the layer's h3 view)
                    HistMove* hm = nullptr);
// next != nullptr: also run the following layer's S1 (writes x1_next, qkv_next) in the same launch
void launch_conv_ffn(const float* g, const float* g_hist, const float* x2, float* x_out, const LayerW& w,
                     const int64_t* lens, int M, int Tp, int n_chunks, int ksize, int mask_mul, const LayerW* next,
                     float* x1_next, float* qkv_next, hipStream_t st, bool causal = true, const PadSkip& ps = PadSkip{},
                     VtOut vt_next = VtOut{}, bool h3 = false);
// ---- split route for under-filled grids (see split_route_kernels.hip): the layer tail cut at its FFNs, each FFN's hidden
// dimension split over S (1, 2, 4 or 8; a divisor of n_chunks) workgroups per row block ----
void launch_conv_pre(const float* g, const float* g_hist, const float* x2, float* x3, const LayerW& w, const int64_t* lens,
                     int M, int Tp, int ksize, int mask_mul, hipStream_t st, bool causal = true,
                     const PadSkip& ps = PadSkip{}, bool h3 = false);
// out = LN_out?(r + scale * FFN(LN?(x))) with r = x (pre-LN blocks) or r = LN(x) (residual_is_normed: post-LN blocks);
// ln_g == nullptr: no LN in front of the FFN; partial: S * M * 256 floats of scratch
void launch_ffn_split(const float* x, const float* ln_g, const float* ln_b, const f32x4* w1, const float* b1,
                      const f32x4* w2, const float* b2, float scale, const float* out_ln_g, const float* out_ln_b,
                      float* partial, float* out, int M, int n_chunks, int S, hipStream_t st, const PadSkip& ps = PadSkip{},
                      bool residual_is_normed = false, bool h3 = false, int* ticket = nullptr);  // h3: w1 / w2 are the re-packed weights
// kc / vc: write the K / V thirds to these cache rows instead of qkv (single-session streaming)
// ---- one streaming session's chunk (<= 16 rows): feed-forward slices whose partial tiles are joined by their CONSUMER ----
// JoinIn describes a pending join  out = LN?(x + scale (sum_s partial[s] + b2))  of S partial tiles [S][M][256]: the launch
// that needs `out` as its input computes it in its prologue (every workgroup for itself, from L2) and workgroup 0 also
// stores it to `out` -- the join launch (4.4 us for 16 KB of work) disappears.  partial == nullptr: no pending join.
struct JoinIn {
  const float* partial = nullptr;
  int S = 0;
  const float* b2 = nullptr;
  float scale = 0.f;
  const float* x = nullptr;                        // residual input of the joined module
  const float *ln_g = nullptr, *ln_b = nullptr;    // LayerNorm behind the residual sum (or nullptr)
  float* out = nullptr;
};
int split_rows16_max();  // rows up to which the split route runs its 16-row forms (split_route_kernels.hip)
// true iff launch_ffn_half16 / launch_join_ln_qkv16 serve this shape (M rows, S = n_chunks slices; PPASR_* switches on)
bool ffn_half16_route(int M, int S, int n_chunks);
// partial[2 S][M][256] <- the 2 S half-chunk slices of FFN(LN(x_in)), x_in = x or the join `jn` (then jn.out is written)
void launch_ffn_half16(const float* x, const JoinIn& jn, const float* ln_g, const float* ln_b, const f32x4* w1, const float* b1,
                       const f32x4* w2, float* partial, int M, int n_chunks, hipStream_t st);
// qkv (K / V thirds to kc / vc) <- LN_mha(jn) [Wq | Wk | Wv]; jn.out is written
void launch_join_ln_qkv16(const JoinIn& jn, float* qkv, const LayerW& w, int M, hipStream_t st, float* kc, float* vc);
// out <- the join alone (the last layer's)
void launch_join16(const JoinIn& jn, int M, hipStream_t st);
void launch_ln_qkv(const float* x1, float* qkv, const LayerW& w, int M, hipStream_t st, const PadSkip& ps = PadSkip{},
                   float* kc = nullptr, float* vc = nullptr, bool h3 = false);
void launch_conv_ffn_stride(const float* g, const float* g_hist, const float* x2, float* x_out, const LayerW& w, const int64_t* lens, int B,
                            int Tp, int Ts, int n_chunks, int ksize, int mask_mul_out, hipStream_t st,
                            const PadSkip& ps = PadSkip{}, bool causal = true, bool h3 = false,
                            float* x3_out = nullptr);  // x3_out: stop at the conv module's output (the caller runs the FFN split)
// streaming helpers
// fused S2+S3 for the batched plain-head path (4 heads x 64): attention + out-projection + LN_conv + pw1 + GLU
void launch_attn_out_glu(const AttnArgs& a, int B, const float* x1, float* x2, float* g, const LayerW& w, hipStream_t st,
                         bool h3 = false);  // h3: out-projection / pointwise_conv1 on the fp16 x3 route (w = the h3 view)
void launch_pw1_glu(const float* xhat, float* g, const LayerW& w, int M, hipStream_t st);
// all layers' conv histories in one launch: layer i reads xh_hist + i*lo_stride*256 (tab[i].rows <= 32 rows), writes
// g_hist + i*lo_stride*256; tab is a DEVICE array
struct HistLayer {
  const f32x4* pw1;    // packed pointwise_conv1 [256][512]
  const float* pw1_b;  // [512]
  int rows, pad;
};
void launch_pw1_glu_layers(const float* xh_hist, float* g_hist, const HistLayer* tab, int n_layers, int lo_stride,
                           hipStream_t st);
void launch_kv_append(const float* qkv, float* kc, float* vc, int n_rows, hipStream_t st);
// multi-session variants: row (b, t) of the chunk batch <-> session sess[b]
void launch_kv_append_group(const float* qkv, float* kc, float* vc, long long sess_stride, const SessDesc* sess, int n, int c,
                            hipStream_t st);
void launch_hist_gather(const float* hist, long long sess_stride, const SessDesc* sess, float* dst, int n, int lo,
                        hipStream_t st);
void launch_hist_update_group(float* hist, long long sess_stride, const SessDesc* sess, const float* fresh, int n, int c,
                              int lo, hipStream_t st);
void launch_hist_update(float* hist, const float* fresh, int n, int lo, hipStream_t st);
// (D = model width: 256, or the general route's 512 / 768 / 1024 -- one thread per column)
void launch_cache_export(const float* kc, const float* vc, float* att, int T, int div, hipStream_t st, int D = 256);
void launch_cache_import(const float* att, float* kc, float* vc, int T, int div, hipStream_t st, int D = 256);
void launch_cnn_transpose(const float* src, float* dst, int lo, int lo_ref, int to_ref, hipStream_t st, int D = 256);
// hw.ln_g == nullptr: no final LayerNorm (Squeezeformer has no after_norm, squeezeformer/encoder.py:232-235)
void launch_ctc_head(const float* x, const HeadW& hw, float* logits, int32_t* fr_argmax, float* fr_maxprob,
                     float* row_max, float* row_sum, int M, hipStream_t st, const PadSkip& ps = PadSkip{},
                     int n_slices = 1, float* part = nullptr, bool h3 = false);  // h3: hw.w re-packed, tiles on the fp16 x3 route  // n_slices > 1: vocabulary tiles over that many workgroups
                                                                // per row block, part = 3 * n_slices * M floats of scratch
void launch_softmax_from_stats(float* probs_inout, const float* row_max, const float* row_sum, int M, int V,
                               hipStream_t st, const PadSkip& ps = PadSkip{});
void launch_frame_argmax(const float* probs, int32_t* fr_argmax, float* fr_maxprob, int M, int V, hipStream_t st);
void launch_ctc_collapse(const int32_t* fr_argmax, const float* fr_maxprob, const int32_t* frame_lens, int B, int Tp,
                         int blank, int32_t* tokens, int32_t* n_tokens, double* score, hipStream_t st);
// ragged batches (skip_padding): rows t with mul*t >= lens[b] of the outputs <- 0 (any pointer may be null)
void launch_zero_pad_rows(float* probs, float* logits, int32_t* fr_argmax, float* fr_maxprob, const int64_t* lens, int B,
                          int Tp, int mul, int V, hipStream_t st);
hipError_t configure_kernels();  // opt in to >64 KiB dynamic LDS

}  // namespace ppasr
