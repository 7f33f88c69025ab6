// squeezeformer_kernels.h -- launch interface of the Squeezeformer row-block kernels
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rowblock.h"

namespace ppasr {

// One SqueezeformerEncoderLayer (squeezeformer/encoder.py:386-433); adaptive scale folded into
// wqkv / ff*_w1 / pw1 (+ their biases) at pack time.
struct SqLayerW {
  const f32x4 *wqkv, *wo, *ff1_w1, *ff1_w2, *pw1, *pw2, *ff2_w1, *ff2_w2;
  const float *bqkv, *bo, *ff1_b1, *ff1_b2, *pw1_b, *pw2_b, *ff2_b1, *ff2_b2;
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b, *ln4_g, *ln4_b, *ln_cm_g, *ln_cm_b;
  float cm_eps;  // as LayerW::cm_eps
  const float *dw_w, *dw_b, *glu_pad;
  const float *pos_u, *pos_v, *ptab;
  // streaming (forward_chunk): the conv-module cache holds SCALED inputs ada_scale*x + ada_bias
  // (convolution.py:119-137), re-projected every chunk with the UNFOLDED pointwise_conv1
  const float *cm_scale, *cm_bias;
  const f32x4* pw1_raw;
  const float* pw1_b_raw;
};
struct SqReduceW {
  const float *dw_w, *dw_b;  // [ks][256] tap-major depthwise stride-2 conv, [256]
  int ks = 1;                // 1: TimeReductionLayerStream; 5: TimeReductionLayer1D (padding ks - 2 on both sides)
  const f32x4* pw;           // packed 256x256
  const float* pw_b;
};

void launch_sq_qkv(const float* x, float* qkv, const f32x4* wqkv, const float* bqkv, int M, hipStream_t st,
                   const PadSkip& ps = PadSkip{});
void launch_sq_mid(const float* ctx, const float* x, float* x2, float* g, float* xhat_out, const SqLayerW& w,
                   const int64_t* lens, int M, int Tp, int mask_mul, int n_chunks, hipStream_t st,
                   const PadSkip& ps = PadSkip{}, int rows = 32,  // rows: 32, or 16 = the 16-row-block kernels (rbt.h)
                   bool h3 = false);  // h3: the feed-forward module on the fp16 x3 route (csrc/h3.h; w = the layer's h3 view; rows 32)
// the register depthwise conv of the transposed forms (and with them the fp16 x3 tail kernel) exists for kernels 31 / 15
inline bool sq_h3_supported(int ksize, int Tp) { return Tp >= 4 && (ksize == 31 || ksize == 15); }
unsigned int* squeezeformer_h3_ovf_counter();  // device address of this translation unit's range-guard counter (h3.h)
// g_hist != nullptr: streaming (single stream, rows = frames of one chunk; left context from g_hist [ksize-1][256])
void launch_sq_tail(const float* g, const float* g_hist, const float* x2, float* x_out, float* qkv_next, const SqLayerW& w,
                    const f32x4* wqkv_next, const float* bqkv_next, const int64_t* lens, int M, int Tp, int mask_mul,
                    int n_chunks, int ksize, hipStream_t st, const PadSkip& ps = PadSkip{}, bool causal = true, int rows = 32,
                    bool h3 = false);
// split route for under-filled launches (ppasr_set_ffn_split): the pieces of K_B / K_C around their feed-forward modules
void launch_sq_oproj(const float* ctx, const float* x, float* x1, const SqLayerW& w, int M, hipStream_t st,
                     const PadSkip& ps = PadSkip{});
void launch_sq_pw1glu(const float* x2, float* g, float* xhat_out, const SqLayerW& w, const int64_t* lens, int M, int Tp,
                      int mask_mul, hipStream_t st, const PadSkip& ps = PadSkip{});
void launch_sq_reduce(const float* x, float* xr, float* qkv, const SqReduceW& rw, const f32x4* wqkv, const float* bqkv,
                      const int64_t* lens, int B, int Tp, int Tr, hipStream_t st, const PadSkip& ps = PadSkip{});
void launch_sq_recover(const float* xr, const float* saved, float* x, float* qkv, const f32x4* wrec, const float* brec,
                       const f32x4* wqkv, const float* bqkv, int B, int Tp, int Tr, hipStream_t st,
                       const PadSkip& ps = PadSkip{});
void launch_ln_rows(float* x, const float* g, const float* b, int M, hipStream_t st, const PadSkip& ps = PadSkip{});
hipError_t configure_squeezeformer_kernels();

}  // namespace ppasr
