// ds2_kernels.h -- launch interface of the DeepSpeech2 kernels
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rowblock.h"

namespace ppasr {

struct Ds2LayerW {
  const f32x4* w_ih;   // packed [in_padded][dirs*4H] (both directions side by side: one GEMM per layer)
  const float* b_sum;  // [dirs*4H]  b_ih + b_hh
  const float* w_hh;   // [dirs][4H][H]
  // the same weights in MFMA fragment order for the batched step kernel: per direction H/8 column tiles of 32 =
  // 8 units x {i, f, g, o} (so that a workgroup owns every gate of its units), [tile][k-group][lane][4]
  const f32x4* w_hh_pk;
  const float *ln_g, *ln_b;  // [dirs*H]
  int in_dim_padded;
};
struct Ds2W {
  const float *cmvn_mean, *cmvn_istd, *c1_w, *c1_b, *c2_w, *c2_b;
  const f32x4* ctc_w;  // packed [dirs*H][Vpad]
  const float* ctc_b;  // [Vpad]
  int H, dirs, n_layers, V, Vpad, ldx;
};

void launch_ds2_conv1(const float* feats, const float* mean, const float* istd, const float* w, const float* bias, float* y1,
                      int B, int T, int F, int T1, int F1, hipStream_t st);
void launch_ds2_conv2(const float* y1, const float* w, const float* bias, float* x, int B, int T1, int F1, int Tp, int F2,
                      int ldx, hipStream_t st);
void launch_ds2_lens(const int64_t* lens, int32_t* out32, int64_t* out64, int B, int Tp, hipStream_t st);
void launch_lstm_step(const float* gx, const float* whh, const float* hprev, float* hnext, float* c, float* y,
                      const int32_t* lens, int B, int T, int H, int dirs, int step, hipStream_t st);
// One LSTM time step for up to 32 utterances per workgroup row tile on the matrix cores (grid H/8 x dirs x ceil(B/32)):
// gates = h_prev W_hh^T as a [32 x H] x [H x 32] MFMA tile per workgroup, the contraction split over its 8 waves.
void launch_lstm_step_mfma(const float* gx, const f32x4* whh_pk, const float* hprev, float* hnext, float* c, float* y,
                           const int32_t* lens, int B, int T, int H, int dirs, int step, hipStream_t st);
void launch_ln_wide(float* x, const float* g, const float* b, int M, int N, hipStream_t st);

}  // namespace ppasr
