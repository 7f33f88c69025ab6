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
  const float* b_hh = nullptr;  // GRU only: [dirs][3H] (the candidate gate needs W_hc h + b_hc on its own); b_sum = b_ih then
};
// unidirectional stack as a wavefront over (layer, time) -- see k_lstm_wave: per-layer device pointers
struct Ds2WaveLayer {
  const f32x4* whh_pk;  // recurrent weights, fragment order, gate-interleaved tiles (as Ds2LayerW::w_hh_pk)
  const f32x4* wih_pk;  // layers >= 1: input weights with the previous layer's LayerNorm gamma folded in, same order
  const float* s_n;     // [4H] column sums of those folded weights (gate-interleaved order)
  const float* c_n;     // [4H] b_ih + b_hh + W_ih * beta of the previous layer's LayerNorm (GRU: without b_hh)
  const float* bhh_n = nullptr;  // GRU only: [4H] b_hh in the gate-interleaved column order (fourth slot zero)
};
struct Ds2W {
  const float *cmvn_mean, *cmvn_istd, *c1_w, *c1_b, *c2_w, *c2_b;
  const f32x4* ctc_w;  // packed [dirs*H][Vpad]
  const float* ctc_b;  // [Vpad]
  int H, dirs, n_layers, V, Vpad, ldx;
  int gates = 4;  // 4 = LSTM (i, f, g, o), 3 = GRU (r, z, c)
  const Ds2WaveLayer* wave_tab = nullptr;  // device [n_layers]; nullptr: no wavefront path (bidirectional models)
};

void launch_ds2_conv1(const float* feats, const float* mean, const float* istd, const float* w, const float* bias, float* y1,
                      int B, int T, int F, int T1, int F1, hipStream_t st);
void launch_ds2_conv2(const float* y1, const float* w, const float* bias, float* x, int B, int T1, int F1, int Tp, int F2,
                      int ldx, hipStream_t st);
void launch_ds2_lens(const int64_t* lens, int32_t* out32, int64_t* out64, int B, int Tp, hipStream_t st);
void launch_lstm_step(const float* gx, const float* whh, const float* hprev, float* hnext, float* c, float* y,
                      const int32_t* lens, int B, int T, int H, int dirs, int step, hipStream_t st);
// The whole recurrence of ONE utterance's layer as a single persistent launch (ds2_kernels.hip k_lstm_persist; LSTM,
// H = 1024): W_hh stays in registers, the time steps exchange h through xbuf (2 * dirs * H 8-byte granules, tags epoch ..
// epoch + T - 1 must not occur in it beforehand: zero it once per call and give every layer its own range).
//   gx [dirs][T][4H], h_init / h_final [dirs][H], c_state [dirs][H] (in: initial, out: final), y [T][dirs * H] pre-zeroed,
//   lens [1] valid steps, *abort_flag != 0 afterwards: a workgroup gave up waiting -- the outputs are invalid
bool lstm_persist_fits(int H, int dirs);
void launch_lstm_persist(const float* gx, const float* whh, const float* bhh, bool gru, const float* h_init, float* c_state,
                         float* h_final, float* y, const int32_t* lens, int T, int H, int dirs, unsigned long long* xbuf,
                         unsigned int epoch, int* abort_flag, hipStream_t st);
void launch_occupy(int n_wg, int ms, float* sink, hipStream_t st);  // test hook: holds n_wg CUs for ms milliseconds
// One GRU time step (paddle.nn.GRU, gate rows r, z, c):  r = s(x_r + h_r), z = s(x_z + h_z), c = tanh(x_c + r * h_c),
// h' = (h - c) * z + c, with x_* = W_ih x + b_ih (gx [dirs][B*T][3H]) and h_* = W_hh h + b_hh.
void launch_gru_step(const float* gx, const float* whh, const float* bhh, const float* hprev, float* hnext, float* y,
                     const int32_t* lens, int B, int T, int H, int dirs, int step, hipStream_t st);
// One LSTM time step for up to 32 utterances per workgroup row tile on the matrix cores (grid H/8 x dirs x ceil(B/32)):
// gates = h_prev W_hh^T as a [32 x H] x [H x 32] MFMA tile per workgroup, the contraction split over its 8 waves.
void launch_lstm_step_mfma(const float* gx, const f32x4* whh_pk, const float* hprev, float* hnext, float* c, float* y,
                           const int32_t* lens, int B, int T, int H, int dirs, int step, hipStream_t st);
// GRU layers on the same tiles (the fourth gate slot of the packed weights is zero); bhh [dirs][3H]
void launch_gru_step_mfma(const float* gx, const f32x4* whh_pk, const float* bhh, const float* hprev, float* hnext, float* y,
                          const int32_t* lens, int B, int T, int H, int dirs, int step, hipStream_t st);
// One wavefront launch of the unidirectional stack: layers l_lo .. l_lo + n_l - 1, layer l at time s - l.
//   gx0 [B*T][4H] layer-0 input projections; hbuf [L][2][B][H] (slot = time parity), cbuf [L][B][H],
//   yring [L][2][B][H] raw outputs of the last two time steps, out [B*T][H] raw outputs of the last layer (pre-zeroed)
void launch_lstm_wave(const float* gx0, const Ds2WaveLayer* tab, float* hbuf, float* cbuf, float* yring, float* out,
                      const int32_t* lens, int B, int T, int H, int L, int s, int l_lo, int n_l, hipStream_t st, bool gru = false);
// [B][H] row-major <-> the MFMA-fragment order of the wavefront kernel's state buffers (hbuf / yring)
void launch_state_reorder(const float* src, float* dst, int B, int H, bool to_frag, hipStream_t st);
void launch_ln_wide(float* x, const float* g, const float* b, int M, int N, hipStream_t st);

}  // namespace ppasr
