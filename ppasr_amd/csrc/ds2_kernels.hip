// ds2_kernels.hip -- DeepSpeech2 conv + LSTM stack (ppasr/model_utils/deepspeech2/{encoder,conv}.py).
//   conv front-end  : Conv2dSubsampling4Pure (conv.py:5-21): Conv2D(1->32,3,s2)+ReLU, Conv2D(32->32,3,s2)+ReLU,
//                     [B,T',32*19] with feature index c*19+f, x_len = ((len-1)//2-1)//2
//   LSTM layer      : input projection for all frames at once on the MFMA GEMM (launch_dense), then the
//                     recurrence as one small kernel per time step: the 4H x H recurrent matrix is sharded
//                     over 256 workgroups (4 hidden units x 4 gates = 16 rows = 64 KiB each, served from the
//                     XCD's own L2 slice after the first step because block->XCD placement is stable);
//                     sequence_length semantics of paddle.nn.LSTM (encoder.py:89-91): steps >= len leave the
//                     state untouched and the output zero; the reverse direction walks t = len-1-s.
//   LayerNorm       : nn.LayerNorm over 1024 / 2048 features (encoder.py:54,93)
// The recurrence is latency-bound (one dependent step = h_{t-1} broadcast + 16 dot products per
// workgroup), not MFMA- or HBM-bound.
#include "launch.h"
#include <cstdlib>

#include "ds2_kernels.h"

#include <math.h>

namespace ppasr {

// CMVN + conv1 (1->32) + ReLU -> y1 [B][T1][F1][32]
__global__ __launch_bounds__(256) void k_ds2_conv1(const float* __restrict__ feats, const float* __restrict__ mean,
                                                   const float* __restrict__ istd, const float* __restrict__ w /*[9][32]*/,
                                                   const float* __restrict__ bias, float* __restrict__ y1, int T, int F,
                                                   int T1, int F1) {
  __shared__ float xs[3][128];
  const int b = blockIdx.y, t1 = blockIdx.x, tid = threadIdx.x;
  for (int idx = tid; idx < 3 * F; idx += 256) {
    int i = idx / F, f = idx - i * F;
    xs[i][f] = (feats[((size_t)b * T + 2 * t1 + i) * F + f] - mean[f]) * istd[f];
  }
  __syncthreads();
  const int c = tid & 31;
  float wr[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) wr[j] = w[j * 32 + c];
  const float bv = bias[c];
  for (int f1 = tid >> 5; f1 < F1; f1 += 8) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc = fmaf(wr[i * 3 + j], xs[i][2 * f1 + j], acc);
    y1[(((size_t)b * T1 + t1) * F1 + f1) * 32 + c] = fmaxf(acc + bv, 0.f);
  }
}

// conv2 (32->32, 3x3, s2) + ReLU -> x [B*Tp][ldx] with feature index c*F2 + f (conv.py:19 transpose+reshape);
// columns [32*F2, ldx) are zero (K padding of the following GEMM).
__global__ __launch_bounds__(256) void k_ds2_conv2(const float* __restrict__ y1, const float* __restrict__ w /*[9][32 ci][32 co]*/,
                                                   const float* __restrict__ bias, float* __restrict__ x, int T1, int F1,
                                                   int Tp, int F2, int ldx) {
  __shared__ float tile[3][40][32];  // 3 rows x F1 (<=40) x 32 channels
  const int b = blockIdx.y, tp = blockIdx.x, tid = threadIdx.x;
  for (int idx = tid; idx < 3 * F1 * 32; idx += 256) {
    int i = idx / (F1 * 32), r = idx - i * F1 * 32;
    tile[i][r >> 5][r & 31] = y1[(((size_t)b * T1 + 2 * tp + i) * F1) * 32 + r];
  }
  __syncthreads();
  const int co = tid & 31;
  float* row = x + ((size_t)b * Tp + tp) * ldx;
  for (int f2 = tid >> 5; f2 < F2; f2 += 8) {
    float acc = 0.f;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const float* wp = w + ((i * 3 + j) * 32) * 32 + co;
        const float* tp_ = &tile[i][2 * f2 + j][0];
#pragma unroll 8
        for (int ci = 0; ci < 32; ++ci) acc = fmaf(wp[ci * 32], tp_[ci], acc);
      }
    row[co * F2 + f2] = fmaxf(acc + bias[co], 0.f);
  }
  for (int c = 32 * F2 + tid; c < ldx; c += 256) row[c] = 0.f;
}

__global__ void k_ds2_lens(const int64_t* __restrict__ lens, int32_t* __restrict__ out32, int64_t* __restrict__ out64, int B,
                           int Tp) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t l = ((lens[b] - 1) / 2 - 1) / 2;  // conv.py:20
  l = l < 0 ? 0 : (l > Tp ? Tp : l);
  out32[b] = (int32_t)l;
  if (out64) out64[b] = l;
}

// One LSTM time step for all utterances, both directions (blockIdx.y).  Gate order i, f, g, o (paddle.nn.LSTM).
//   gx    [dirs][B*T][4H]  input projections + b_ih + b_hh
//   whh   [dirs][4H][H]
//   hprev / hnext [dirs][B][H] (ping-pong), c [dirs][B][H] (in place: a block owns its 4 units)
//   y     [B*T][dirs*H], pre-zeroed (frames >= len stay zero)
__global__ __launch_bounds__(256) void k_lstm_step(const float* __restrict__ gx, const float* __restrict__ whh,
                                                   const float* __restrict__ hprev, float* __restrict__ hnext,
                                                   float* __restrict__ c, float* __restrict__ y,
                                                   const int32_t* __restrict__ lens, int B, int T, int H, int dirs,
                                                   int step) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hs = smem;          // [H]
  float* gates = smem + H;   // [16]
  const int dir = blockIdx.y, j0 = blockIdx.x * 4, tid = threadIdx.x;
  const int r = tid >> 4, part = tid & 15;  // r = gate*4 + unit ; 16 threads per row
  const int gate = r >> 2, unit = r & 3;
  const int per = H / 16;                  // elements per thread (64 for H = 1024)
  // the 16 threads of a row read it INTERLEAVED (thread `part` takes the float4s part, part + 16, ...): one load
  // instruction of the group covers 256 contiguous bytes = two whole cache lines.  (With a contiguous 256-byte slice per
  // thread every lane touched its own line and took 16 bytes of it per instruction, 64 lines in flight per wave.)
  const float* wrow = whh + ((size_t)dir * 4 * H + (size_t)gate * H + j0 + unit) * H + part * 4;
  const float* hp = hprev + (size_t)dir * B * H;
  float* hn = hnext + (size_t)dir * B * H;
  float* cc = c + (size_t)dir * B * H;
  const float* gxd = gx + (size_t)dir * B * T * 4 * H;
  for (int b = 0; b < B; ++b) {
    const int len = lens[b];
    if (step >= len) {  // finished utterance: carry the state (final state = last valid step)
      if (tid < 4) hn[(size_t)b * H + j0 + tid] = hp[(size_t)b * H + j0 + tid];
      continue;
    }
    const int t = dir == 0 ? step : len - 1 - step;
    for (int k = tid * 4; k < H; k += 1024) *reinterpret_cast<f32x4*>(hs + k) = *reinterpret_cast<const f32x4*>(hp + (size_t)b * H + k);
    __syncthreads();
    float acc = 0.f;
    for (int k = 0; k < per; k += 4) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 16 * k);
      const f32x4 hv = *reinterpret_cast<const f32x4*>(hs + part * 4 + 16 * k);
      acc = fmaf(wv[0], hv[0], acc);
      acc = fmaf(wv[1], hv[1], acc);
      acc = fmaf(wv[2], hv[2], acc);
      acc = fmaf(wv[3], hv[3], acc);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (part == 0) gates[r] = acc + gxd[((size_t)b * T + t) * 4 * H + (size_t)gate * H + j0 + unit];
    __syncthreads();
    if (tid < 4) {
      const float gi = 1.0f / (1.0f + expf(-gates[0 + tid]));
      const float gf = 1.0f / (1.0f + expf(-gates[4 + tid]));
      const float gg = tanhf(gates[8 + tid]);
      const float go = 1.0f / (1.0f + expf(-gates[12 + tid]));
      const size_t si = (size_t)b * H + j0 + tid;
      const float cn = gf * cc[si] + gi * gg;
      const float hv = go * tanhf(cn);
      cc[si] = cn;
      hn[si] = hv;
      y[((size_t)b * T + t) * (size_t)(dirs * H) + (size_t)dir * H + j0 + tid] = hv;
    }
    __syncthreads();
  }
}

// One GRU time step for all utterances, both directions (blockIdx.y): the use_gru variant of the stack
// (deepspeech2/encoder.py:36-42; paddle.nn.GRUCell: rows r, z, c of the 3H weights,
//   r = sigmoid(x_r + h_r), z = sigmoid(x_z + h_z), c = tanh(x_c + r * h_c), h' = (h - c) * z + c
// with x = W_ih x_t + b_ih (gx) and h_* = W_hh h + b_hh -- the candidate needs its recurrent part separately, so b_hh is
// not folded into gx).  Same sharding as k_lstm_step: a workgroup owns 4 hidden units = 12 rows of W_hh.
__global__ __launch_bounds__(256) void k_gru_step(const float* __restrict__ gx, const float* __restrict__ whh,
                                                  const float* __restrict__ bhh, const float* __restrict__ hprev,
                                                  float* __restrict__ hnext, float* __restrict__ y,
                                                  const int32_t* __restrict__ lens, int B, int T, int H, int dirs, int step) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hs = smem;          // [H]
  float* hg = smem + H;      // [12] recurrent parts, [12..24) input parts
  const int dir = blockIdx.y, j0 = blockIdx.x * 4, tid = threadIdx.x;
  const int r = tid >> 4, part = tid & 15;  // r = gate*4 + unit ; 16 threads per row; rows 12..15 idle
  const int gate = r >> 2, unit = r & 3;
  const int per = H / 16;
  const bool live_row = r < 12;
  const float* wrow = whh + ((size_t)dir * 3 * H + (size_t)(live_row ? gate : 0) * H + j0 + unit) * H + part * 4;  // (interleaved, see k_lstm_step)
  const float* hp = hprev + (size_t)dir * B * H;
  float* hn = hnext + (size_t)dir * B * H;
  const float* gxd = gx + (size_t)dir * B * T * 3 * H;
  const float* bh = bhh + (size_t)dir * 3 * H;
  for (int b = 0; b < B; ++b) {
    const int len = lens[b];
    if (step >= len) {
      if (tid < 4) hn[(size_t)b * H + j0 + tid] = hp[(size_t)b * H + j0 + tid];
      continue;
    }
    const int t = dir == 0 ? step : len - 1 - step;
    for (int k = tid * 4; k < H; k += 1024) *reinterpret_cast<f32x4*>(hs + k) = *reinterpret_cast<const f32x4*>(hp + (size_t)b * H + k);
    __syncthreads();
    float acc = 0.f;
    if (live_row) {
      for (int k = 0; k < per; k += 4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 16 * k);
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hs + part * 4 + 16 * k);
        acc = fmaf(wv[0], hv[0], acc);
        acc = fmaf(wv[1], hv[1], acc);
        acc = fmaf(wv[2], hv[2], acc);
        acc = fmaf(wv[3], hv[3], acc);
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (part == 0 && live_row) {
      hg[r] = acc + bh[(size_t)gate * H + j0 + unit];
      hg[12 + r] = gxd[((size_t)b * T + t) * 3 * H + (size_t)gate * H + j0 + unit];
    }
    __syncthreads();
    if (tid < 4) {
      const float gr = 1.0f / (1.0f + expf(-(hg[12 + 0 + tid] + hg[0 + tid])));
      const float gz = 1.0f / (1.0f + expf(-(hg[12 + 4 + tid] + hg[4 + tid])));
      const float cand = tanhf(hg[12 + 8 + tid] + gr * hg[8 + tid]);
      const float hv = (hs[j0 + tid] - cand) * gz + cand;
      hn[(size_t)b * H + j0 + tid] = hv;
      y[((size_t)b * T + t) * (size_t)(dirs * H) + (size_t)dir * H + j0 + tid] = hv;
    }
    __syncthreads();
  }
}

// The same step on the matrix cores, for batches: k_lstm_step walks the utterances one by one (B x the time); here a
// workgroup owns 8 hidden units (their 4 gates = one 32-column MFMA tile, weights re-packed accordingly) for up to 32
// utterances (the 32 rows of the tile), its 8 waves split the K = H contraction (a 32x32 tile with K = 1024 on one
// wave would be a chain of 512 dependent MFMAs = 16 us; 64 per wave = 2 us) and the partial tiles are summed through
// LDS.  H / 8 workgroups per direction = 256 for the bidirectional 1024-unit layer: one per CU.
// GRU = true: nn.GRU layers (deepspeech2/encoder.py:36-42).  Same tiles -- 8 units x {r, z, c, (unused)} = 32 columns, the
// fourth gate slot packed as zeros --; the recurrent sums keep b_hh and stay SEPARATE from the input projections because
// the candidate is tanh(x_c + r * (W_hc h + b_hc)); h' = (h - c~) * z + c~ (the arithmetic of k_gru_step, batched).
template <bool GRU>
__global__ __launch_bounds__(kThreads) void k_lstm_step_mfma(const float* __restrict__ gx, const f32x4* __restrict__ whh_pk,
                                                             const float* __restrict__ bhh, const float* __restrict__ hprev,
                                                             float* __restrict__ hnext, float* __restrict__ c,
                                                             float* __restrict__ y, const int32_t* __restrict__ lens, int B,
                                                             int T, int H, int dirs, int step) {
  __shared__ float part[kWaves][32][33];
  __shared__ float gates[32][33];
  const int tile = blockIdx.x, dir = blockIdx.y, b0 = blockIdx.z * 32;
  const int lane = lane_id(), wave = wave_id();
  const int l31 = lane & 31, hh = lane >> 5;
  const int n_groups = H / 8;             // 8-wide k-groups of the contraction
  const int gpw = n_groups / kWaves;      // k-groups per wave (16 for H = 1024)
  const int g0 = wave * gpw;
  const int b = b0 + l31;
  const bool row_live = b < B && step < lens[min(b, B - 1)];
  // A operand straight from global memory in fragment order (lane = utterance row, 4 consecutive k per load); staging
  // the rows through LDS with whole-row loads was measured and is slower here (11.6 against 9.2 ms per 32 x 5 s batch)
  const float* hrow = hprev + ((size_t)dir * B + min(b, B - 1)) * H + 4 * hh;
  const f32x4* wp = whh_pk + ((size_t)dir * (H / 8) + tile) * n_groups * 64 + lane;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int PFD = 8;  // k-groups in flight
  f32x4 ra[PFD], rb[PFD];
#pragma unroll
  for (int s = 0; s < PFD; ++s) {
    ra[s] = row_live ? *reinterpret_cast<const f32x4*>(hrow + 8 * (g0 + s)) : f32x4{0.f, 0.f, 0.f, 0.f};
    rb[s] = wp[(size_t)(g0 + s) * 64];
  }
  for (int g = 0; g < gpw; g += PFD) {
#pragma unroll
    for (int s = 0; s < PFD; ++s) {
      const f32x4 a = ra[s], bq = rb[s];
      if (g + PFD + s < gpw) {
        ra[s] = row_live ? *reinterpret_cast<const f32x4*>(hrow + 8 * (g0 + g + PFD + s)) : f32x4{0.f, 0.f, 0.f, 0.f};
        rb[s] = wp[(size_t)(g0 + g + PFD + s) * 64];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bq[j], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][acc_row(r, lane)][l31] = acc[r];
  __syncthreads();
  // sum of the 8 partial tiles + input projection: thread -> (row, col) = 2 of the 1024 tile elements
  constexpr int NG = GRU ? 3 : 4;
  const float* gxd = gx + (size_t)dir * B * T * NG * H;
  for (int e = threadIdx.x; e < 32 * 32; e += kThreads) {
    const int row = e >> 5, col = e & 31;
    const int bb = b0 + row;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) v += part[w][row][col];
    const int gate = col >> 3, unit = tile * 8 + (col & 7);
    if (GRU) {
      if (gate < 3) v += bhh[(size_t)dir * 3 * H + (size_t)gate * H + unit];  // (the input parts are read in the cell update)
    } else if (bb < B) {
      const int len = lens[bb];
      if (step < len) {
        const int t = dir == 0 ? step : len - 1 - step;
        v += gxd[((size_t)bb * T + t) * 4 * H + (size_t)gate * H + unit];
      }
    }
    gates[row][col] = v;
  }
  __syncthreads();
  if (threadIdx.x < 32 * 8) {
    const int row = threadIdx.x >> 3, u = threadIdx.x & 7;
    const int bb = b0 + row;
    if (bb < B) {
      const int len = lens[bb];
      const size_t si = ((size_t)dir * B + bb) * H + tile * 8 + u;
      if (step >= len) {
        hnext[si] = hprev[si];  // finished utterance: carry the state (final state = last valid step)
      } else if (GRU) {
        const int t = dir == 0 ? step : len - 1 - step;
        const float* gxr = gxd + ((size_t)bb * T + t) * 3 * H + tile * 8 + u;
        const float gr = 1.0f / (1.0f + expf(-(gxr[0] + gates[row][0 + u])));
        const float gz = 1.0f / (1.0f + expf(-(gxr[(size_t)H] + gates[row][8 + u])));
        const float cand = tanhf(gxr[(size_t)2 * H] + gr * gates[row][16 + u]);
        const float hv = (hprev[si] - cand) * gz + cand;
        hnext[si] = hv;
        y[((size_t)bb * T + t) * (size_t)(dirs * H) + (size_t)dir * H + tile * 8 + u] = hv;
      } else {
        const int t = dir == 0 ? step : len - 1 - step;
        const float gi = 1.0f / (1.0f + expf(-gates[row][0 + u]));
        const float gf = 1.0f / (1.0f + expf(-gates[row][8 + u]));
        const float gg = tanhf(gates[row][16 + u]);
        const float go = 1.0f / (1.0f + expf(-gates[row][24 + u]));
        const float cn = gf * c[si] + gi * gg;
        const float hv = go * tanhf(cn);
        c[si] = cn;
        hnext[si] = hv;
        y[((size_t)bb * T + t) * (size_t)(dirs * H) + (size_t)dir * H + tile * 8 + u] = hv;
      }
    }
  }
}

// Wavefront over (layer, time) for the UNIDIRECTIONAL stack (streaming models): layer l+1 at time t only needs layer l at
// time t, so launch s runs every layer l with 0 <= s - l < T at time t = s - l: T + L - 1 dependent launches instead
// of T * L.  For l >= 1 the input projection is folded into the step (there is no [B*T][4H] pre-pass to wait for):
//   W_ih LN(y) + b = rstd * (W' y - mean * colsum(W')) + (W_ih beta + b),   W' = W_ih diag(gamma)
// so the MFMA runs on the RAW previous-layer output y_{l-1}[t] (a second accumulator tile next to h_{t-1} W_hh^T) and
// the LayerNorm of the row enters through its mean / rstd, which the waves accumulate from the very A fragments they
// load (sum and sum of squares per row).  Same workgroup shape as k_lstm_step_mfma.
// Occupancy: at 152 VGPRs (prefetch depth 8) ONE workgroup fits a CU and the 5 x 128 workgroups of a launch ran as 2.5
// rounds (round 3: depth 4, 103 registers: two workgroups per CU, 1.25 rounds).  Round 4: one accumulator tile for both
// products, buffer loads instead of 64-bit per-lane addresses and depth 2 keep the kernel at 80 registers = six waves per
// SIMD = three workgroups per CU (their 53 KB of LDS allow no more), so a launch is ONE round; same-box A/B of the
// depth, 5 x 1024 LSTM, 5 s utterances: B = 32 5.46 ms (depth 4) -> 5.00 (depth 2), B = 128 15.47 -> 15.22, depth 8
// (spills) 12.2 / 42.3.  With six waves per SIMD the occupancy hides the latency: depth 1 is faster still (B = 32
// 5.05 -> 4.75 ms, B = 64 8.09 -> 7.81, B = 128 14.64 -> 14.28, same box; depth 4 at that occupancy: 5.37 / 8.35 / 15.59).
#ifndef PPASR_WAVE_OCC
#define PPASR_WAVE_OCC 6
#endif
#ifndef PPASR_WAVE_PFD
#define PPASR_WAVE_PFD 1
#endif
// RT = 32-row tiles of the batch per workgroup (1, 2, 4): the 8 waves are RT row tiles x KS = 8 / RT slices of the
// contraction, so a workgroup streams its 32 gate columns' weights ONCE for up to 128 utterances (round 3 ran one
// workgroup per 32 utterances: at B = 128 every weight byte was fetched four times per launch and the wavefront lost to
// the per-step kernel; the partial-tile buffer stays [KS][32 RT][33] = 34 KB whatever RT).
// GRU: nn.GRU layers (gate slots r, z, c, unused).  The recurrent sums keep b_hh and stay separate from the input part,
// because the candidate is tanh(x_c + r (W_hc h + b_hc)); h' = (h - c~) z + c~; no cell state.
template <int RT, bool GRU>
__global__ __launch_bounds__(kThreads, PPASR_WAVE_OCC) void k_lstm_wave(const float* __restrict__ gx0, const Ds2WaveLayer* __restrict__ tab,
                                                        float* __restrict__ hbuf, float* __restrict__ cbuf,
                                                        float* __restrict__ yring, float* __restrict__ out,
                                                        const int32_t* __restrict__ lens, int B, int T, int H, int L, int s,
                                                        int l_lo) {
  // every wave contracts its K slice of BOTH products (two accumulator tiles); splitting the waves by product instead
  // (4 + 4) was measured and is slower: 7.9 against 6.8 ms per 32 x 5 s batch
  // (ONE partial-tile buffer used twice -- for h W_hh^T, then for y W'_ih^T -- keeps the workgroup at 40 KB of LDS: the up
  //  to 5 x 128 workgroups of a launch then fit the chip in one round)
  constexpr int KS = kWaves / RT, NR = 32 * RT, NGATE = GRU ? 3 : 4;
  __shared__ float part[KS * NR * 33];
  __shared__ float stat_s[KS * NR], stat_q[KS * NR];
  __shared__ float gates[NR * 33];  // gate pre-activations (GRU: the input part x; the recurrent part W_hh h + b_hh is
                                    // parked in slice 0 of `part`: element (row, col) is read and rewritten by ONE thread)
  const int tile = blockIdx.x, l = l_lo + blockIdx.y, t = s - l, b0 = blockIdx.z * NR;
  const Ds2WaveLayer lay = tab[l];
  const int lane = lane_id(), wave = wave_id();
  const int l31 = lane & 31, hh = lane >> 5;
  const int rt = wave % RT, kq = wave / RT;  // this wave's row tile and K slice
  const int n_groups = H / 8, gpw = n_groups / KS, g0 = kq * gpw;
  const int b = b0 + 32 * rt + l31;
  const bool row_live = b < B && t < lens[min(b, B - 1)];
  // hbuf / yring hold the states and layer outputs in MFMA FRAGMENT order -- [row tile][k-group][64 lanes = row + 32 *
  // (k quad)][4 k] -- so that a wave's A-operand load is 1 KiB contiguous like its weight load (row-major, every lane of a
  // load touched its own 4 KiB-strided row: 32 cache lines per instruction, 64 such instructions per wave and launch)
  const size_t BH = (size_t)((B + 31) / 32) * 32 * H;
  const float* hprev = hbuf + ((size_t)l * 2 + (t & 1)) * BH;
  float* hnext = hbuf + ((size_t)l * 2 + ((t + 1) & 1)) * BH;
  float* cc = cbuf + (size_t)l * B * H;  // (the cell state stays row-major: it is only read and written element-wise)
  const float* yprev = l > 0 ? yring + ((size_t)(l - 1) * 2 + (t & 1)) * BH : nullptr;
  float* ycur = yring + ((size_t)l * 2 + (t & 1)) * BH;
  // ONE accumulator tile, used for h W_hh^T and then for y W'_ih^T: the first product's partial tile goes to LDS and is
  // summed into two values per thread BEFORE the second product starts, so the kernel stays inside 85 registers = six waves
  // per SIMD = three workgroups per CU, and the 5 x 128 workgroups of a launch are one round (at 103 registers -- both
  // tiles live -- two workgroups fitted a CU and every launch ran a second, quarter-full round)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int PFD = PPASR_WAVE_PFD;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto pidx = [&](int k, int row, int col) { return (k * NR + row) * 33 + col; };
  // operands through BUFFER loads: wave-uniform bases (row tile / column tile folded in) in SGPRs, ONE per-lane 32-bit offset
  // (lane * 16; out of range for the rows of finished utterances, which then read zeros), the k-group as the scalar offset --
  // no 64-bit per-lane addresses to keep (and spill) across the loops, no per-load address arithmetic
  const int voff_a = row_live ? lane * 16 : 0x7fffffff, voff_w = lane * 16;
  const size_t rtile_f = (size_t)(blockIdx.z * RT + rt) * n_groups * 256;  // floats in front of this wave's row tile
  {  // ---- h_{t-1} W_hh^T ----
    const __amdgpu_buffer_rsrc_t rs_a = wstream_rsrc(hprev + rtile_f), rs_w = wstream_rsrc(lay.whh_pk + (size_t)tile * n_groups * 64);
    f32x4 ra[PFD], rb[PFD];
#pragma unroll
    for (int q = 0; q < PFD; ++q) {
      ra[q] = wstream_load(rs_a, voff_a, (g0 + q) * 1024);
      rb[q] = wstream_load(rs_w, voff_w, (g0 + q) * 1024);
    }
    for (int g = 0; g < gpw; g += PFD) {
#pragma unroll
      for (int q = 0; q < PFD; ++q) {
        const f32x4 a = ra[q], bq = rb[q];
        if (g + PFD + q < gpw) {
          ra[q] = wstream_load(rs_a, voff_a, (g0 + g + PFD + q) * 1024);
          rb[q] = wstream_load(rs_w, voff_w, (g0 + g + PFD + q) * 1024);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bq[j], acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[pidx(kq, 32 * rt + acc_row(r, lane), l31)] = acc[r];
  __syncthreads();
  constexpr int EPT = 2 * RT;  // tile elements per thread: NR x 32 over 512 threads
  float vh2[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = threadIdx.x + q * kThreads, row = e >> 5, col = e & 31;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < KS; ++w) v += part[pidx(w, row, col)];
    vh2[q] = v;
  }
  if (l > 0) {  // ---- y_{l-1}[t] W'_ih^T on the raw row, LayerNorm statistics on the side ---- (block-uniform)
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const __amdgpu_buffer_rsrc_t rs_a = wstream_rsrc(yprev + rtile_f), rs_w = wstream_rsrc(lay.wih_pk + (size_t)tile * n_groups * 64);
    f32x4 ra[PFD], rb[PFD];
#pragma unroll
    for (int q = 0; q < PFD; ++q) {
      ra[q] = wstream_load(rs_a, voff_a, (g0 + q) * 1024);
      rb[q] = wstream_load(rs_w, voff_w, (g0 + q) * 1024);
    }
    for (int g = 0; g < gpw; g += PFD) {
#pragma unroll
      for (int q = 0; q < PFD; ++q) {
        const f32x4 a = ra[q], bq = rb[q];
        if (g + PFD + q < gpw) {
          ra[q] = wstream_load(rs_a, voff_a, (g0 + g + PFD + q) * 1024);
          rb[q] = wstream_load(rs_w, voff_w, (g0 + g + PFD + q) * 1024);
        }
        sum += a[0] + a[1] + a[2] + a[3];
        sq += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bq[j], acc, 0, 0, 0);
      }
    }
    sum += __shfl_xor(sum, 32);
    sq += __shfl_xor(sq, 32);
    if (hh == 0) {
      stat_s[kq * NR + 32 * rt + l31] = sum;
      stat_q[kq * NR + 32 * rt + l31] = sq;
    }
    __syncthreads();  // every thread has read the first product's partial tiles
#pragma unroll
    for (int r = 0; r < 16; ++r) part[pidx(kq, 32 * rt + acc_row(r, lane), l31)] = acc[r];
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = threadIdx.x + q * kThreads;
    const int row = e >> 5, col = e & 31;
    const int bb = b0 + row;
    float vi = 0.f;   // input part (x W_ih^T + b_ih [+ b_hh for the LSTM])
    float v = vh2[q];  // recurrent part
    if (bb < B && t < lens[bb]) {
      const int n = tile * 32 + col;  // gate-interleaved column
      if (l == 0) {
        if (!GRU || (col >> 3) < 3) vi = gx0[((size_t)bb * T + t) * NGATE * H + (size_t)(col >> 3) * H + tile * 8 + (col & 7)];
      } else {
        float ss = 0.f, qq = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) {
          acc += part[pidx(w, row, col)];
          ss += stat_s[w * NR + row];
          qq += stat_q[w * NR + row];
        }
        const float mean = ss / (float)H;
        const float var = fmaxf(qq / (float)H - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        vi = rstd * (acc - mean * lay.s_n[n]) + lay.c_n[n];
      }
      if (GRU) v += lay.bhh_n[n];
    }
    if (GRU) {
      gates[row * 33 + col] = vi;
      part[pidx(0, row, col)] = v;
    } else {
      gates[row * 33 + col] = v + vi;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < NR * 8; idx += kThreads) {
    const int row = idx >> 3, u = idx & 7;
    const int bb = b0 + row;
    if (bb < B) {
      const int len = lens[bb];
      const size_t si = (size_t)bb * H + tile * 8 + u;                                     // row-major (cell state)
      const size_t fi = ((size_t)(blockIdx.z * RT + (row >> 5)) * n_groups + tile) * 256 + ((row & 31) + 32 * (u >> 2)) * 4 + (u & 3);  // fragment order
      if (t >= len) {
        hnext[fi] = hprev[fi];  // finished utterance: carry the state (final state = last valid step)
      } else if (GRU) {
        const float* gxr = gates + row * 33;
        const float* ghr = part + row * 33;  // (= pidx(0, row, .))
        const float gr = 1.0f / (1.0f + expf(-(gxr[u] + ghr[u])));
        const float gz = 1.0f / (1.0f + expf(-(gxr[8 + u] + ghr[8 + u])));
        const float cand = tanhf(gxr[16 + u] + gr * ghr[16 + u]);
        const float hv = (hprev[fi] - cand) * gz + cand;
        hnext[fi] = hv;
        ycur[fi] = hv;
        if (l == L - 1) out[((size_t)bb * T + t) * (size_t)H + tile * 8 + u] = hv;
      } else {
        const float* gr_ = gates + row * 33;
        const float gi = 1.0f / (1.0f + expf(-gr_[0 + u]));
        const float gf = 1.0f / (1.0f + expf(-gr_[8 + u]));
        const float gg = tanhf(gr_[16 + u]);
        const float go = 1.0f / (1.0f + expf(-gr_[24 + u]));
        const float cn = gf * cc[si] + gi * gg;
        const float hv = go * tanhf(cn);
        cc[si] = cn;
        hnext[fi] = hv;
        ycur[fi] = hv;
        if (l == L - 1) out[((size_t)bb * T + t) * (size_t)H + tile * 8 + u] = hv;
      }
    }
  }
}

// LayerNorm over N features (N % 256 == 0, N <= 4096), in place; one wave per row
__global__ __launch_bounds__(256) void k_ln_wide(float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                 int M, int N) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float* p = x + (size_t)row * N;
  const int n4 = N / 256;  // float4 per lane
  f32x4 v[16];
  float s = 0.f;
  for (int i = 0; i < n4; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(p + (i * 64 + lane) * 4);
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  const float mean = wave_sum(s) / (float)N;
  float q = 0.f;
  for (int i = 0; i < n4; ++i) {
    v[i] = v[i] - mean;
    q += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)N + 1e-5f);
  for (int i = 0; i < n4; ++i) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + (i * 64 + lane) * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + (i * 64 + lane) * 4);
    *reinterpret_cast<f32x4*>(p + (i * 64 + lane) * 4) = v[i] * rstd * gv + bv;
  }
}

void launch_ds2_conv1(const float* feats, const float* mean, const float* istd, const float* w, const float* bias, float* y1,
                      int B, int T, int F, int T1, int F1, hipStream_t st) {
  PPASR_LAUNCH(k_ds2_conv1, dim3(T1, B), dim3(256), 0, st, feats, mean, istd, w, bias, y1, T, F, T1, F1);
}
void launch_ds2_conv2(const float* y1, const float* w, const float* bias, float* x, int B, int T1, int F1, int Tp, int F2,
                      int ldx, hipStream_t st) {
  PPASR_LAUNCH(k_ds2_conv2, dim3(Tp, B), dim3(256), 0, st, y1, w, bias, x, T1, F1, Tp, F2, ldx);
}
void launch_ds2_lens(const int64_t* lens, int32_t* out32, int64_t* out64, int B, int Tp, hipStream_t st) {
  PPASR_LAUNCH(k_ds2_lens, dim3((B + 63) / 64), dim3(64), 0, st, lens, out32, out64, B, Tp);
}
void launch_lstm_step(const float* gx, const float* whh, const float* hprev, float* hnext, float* c, float* y,
                      const int32_t* lens, int B, int T, int H, int dirs, int step, hipStream_t st) {
  PPASR_LAUNCH(k_lstm_step, dim3(H / 4, dirs), dim3(256), (H + 16) * sizeof(float), st, gx, whh, hprev, hnext, c, y,
                     lens, B, T, H, dirs, step);
}
void launch_gru_step(const float* gx, const float* whh, const float* bhh, const float* hprev, float* hnext, float* y,
                     const int32_t* lens, int B, int T, int H, int dirs, int step, hipStream_t st) {
  PPASR_LAUNCH(k_gru_step, dim3(H / 4, dirs), dim3(256), (H + 32) * sizeof(float), st, gx, whh, bhh, hprev, hnext, y,
                     lens, B, T, H, dirs, step);
}
void launch_lstm_step_mfma(const float* gx, const f32x4* whh_pk, const float* hprev, float* hnext, float* c, float* y,
                           const int32_t* lens, int B, int T, int H, int dirs, int step, hipStream_t st) {
  PPASR_LAUNCH(k_lstm_step_mfma<false>, dim3(H / 8, dirs, (B + 31) / 32), dim3(kThreads), 0, st, gx, whh_pk, nullptr, hprev,
               hnext, c, y, lens, B, T, H, dirs, step);
}
void launch_gru_step_mfma(const float* gx, const f32x4* whh_pk, const float* bhh, const float* hprev, float* hnext, float* y,
                          const int32_t* lens, int B, int T, int H, int dirs, int step, hipStream_t st) {
  PPASR_LAUNCH(k_lstm_step_mfma<true>, dim3(H / 8, dirs, (B + 31) / 32), dim3(kThreads), 0, st, gx, whh_pk, bhh, hprev,
               hnext, nullptr, y, lens, B, T, H, dirs, step);
}
void launch_lstm_wave(const float* gx0, const Ds2WaveLayer* tab, float* hbuf, float* cbuf, float* yring, float* out,
                      const int32_t* lens, int B, int T, int H, int L, int s, int l_lo, int n_l, hipStream_t st, bool gru) {
  // row tiles per workgroup: up to 64 utterances stream a layer's weights once.  Beyond that two row tiles per workgroup
  // again (B = 128: 5 x 128 x 2 workgroups = five per CU, dealt as they finish) beat four (640 workgroups = 2.5 per CU,
  // i.e. three on some CUs): 95.3 against 99.8 us per launch, same box; the second read of the weights comes from L2 / MALL
  const int rt = B <= 32 ? 1 : 2;
  const dim3 grid(H / 8, n_l, (B + 32 * rt - 1) / (32 * rt));
#define WAVE_LAUNCH(RT, GRU) \
  PPASR_LAUNCH((k_lstm_wave<RT, GRU>), grid, dim3(kThreads), 0, st, gx0, tab, hbuf, cbuf, yring, out, lens, B, T, H, L, s, l_lo)
  if (gru) {
    if (rt == 1) WAVE_LAUNCH(1, true); else WAVE_LAUNCH(2, true);
  } else {
    if (rt == 1) WAVE_LAUNCH(1, false); else WAVE_LAUNCH(2, false);
  }
#undef WAVE_LAUNCH
}
// ---- persistent recurrence of a single utterance (B = 1, H = 1024, LSTM; any number of directions): ONE launch per
// layer instead of one per time step ----
// k_lstm_step streams the layer's W_hh (33.5 MB for both directions) from L2 / MALL at every step: 125 x the bytes the
// weights have.  Here workgroup (j, dir) keeps the 32 rows of W_hh that belong to its 8 hidden units (4 gates x 8 units x
// 1024 columns = 128 KB) in REGISTERS for the whole layer -- 512 threads x 64 weights: thread (row = lane & 31, column
// segment = 2 wave + (lane >> 5)) holds W[row][64 seg .. 64 seg + 64) -- and the time steps are separated by an exchange
// of the new h through memory instead of a kernel boundary:
//   * publish: the 8 lanes that own a unit store {bits of h, tag = epoch + step} as ONE 8-byte agent-scope store into
//     slot (step & 1) of the exchange buffer (data and tag travel together: no separate flag, no fence);
//   * gather : every thread polls the two granules it stages into LDS (8-byte agent-scope loads) until both carry the tag of
//     the previous step.  Two slots suffice: a workgroup publishes step s + 1 only after it gathered all of step s, which
//     every workgroup published only after ITS gather of step s - 1 was complete.
// Per step: 16 broadcast LDS reads + 64 FMAs per thread, two workgroup barriers, one exchange (MI355X_MICROARCH.md
// "allgather": 8 KB published by 128 CUs, 2.4 - 3 us).  Bounded spins: a workgroup that waits longer than kPersistSpins
// polls raises *abort_flag (so do all others on seeing it) and the host falls back to the per-step kernels -- the launch
// needs all its workgroups resident (H / 8 x dirs <= CUs x occupancy, checked on the host), nothing else may hold the chip.
constexpr int kPersistThreads = 512;
constexpr int kPersistSpins = 200000;  // ~0.2 s
// GRU = true (round 6): nn.GRU layers the same way -- 3 gates x 8 units = 24 of the 32 weight rows of a workgroup are in
// use (96 KB of W_hh in registers), the recurrent parts keep their b_hh and stay apart from the input parts until the cell
// (the candidate gate multiplies only the recurrent part by r: k_gru_step's arithmetic); there is no cell state.
template <bool GRU>
__global__ __launch_bounds__(kPersistThreads) void k_lstm_persist(const float* __restrict__ gx, const float* __restrict__ whh,
                                                                  const float* __restrict__ bhh,
                                                                  const float* __restrict__ h_init, float* __restrict__ c_state,
                                                                  float* __restrict__ h_final, float* __restrict__ y,
                                                                  const int32_t* __restrict__ lens, int T, int dirs,
                                                                  unsigned long long* __restrict__ xbuf, unsigned int epoch,
                                                                  int* __restrict__ abort_flag) {
  constexpr int H = 1024, NG = GRU ? 3 : 4;
  __shared__ __attribute__((aligned(16))) float hs[H];
  __shared__ float part[kPersistThreads / 64][32];
  __shared__ int s_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dir = blockIdx.y, u0 = blockIdx.x * 8;
  const int row = lane & 31, seg = 2 * wave + (lane >> 5);
  const int gate = row >> 3, unit = row & 7;
  const bool live_row = gate < NG;  // (GRU: rows 24 .. 31 of the 32 carry zeros)
  float w[64];
  {
    const float* wr = whh + ((size_t)dir * NG * H + (size_t)(live_row ? gate : 0) * H + u0 + unit) * H + seg * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      f32x4 v = *reinterpret_cast<const f32x4*>(wr + 4 * i);
      if (!live_row) v = f32x4{0.f, 0.f, 0.f, 0.f};
      w[4 * i] = v[0]; w[4 * i + 1] = v[1]; w[4 * i + 2] = v[2]; w[4 * i + 3] = v[3];
    }
  }
  const int len = min(max(lens[0], 0), T);
  const float* gxd = gx + (size_t)dir * T * NG * H + (size_t)(live_row ? gate : 0) * H + u0 + unit;  // + t * NG * H: this lane's gate row (wave 0, lanes < 32)
  unsigned long long* xb = xbuf + (size_t)dir * H;                                  // + (slot) * dirs * H
  const bool owner = wave == 0 && lane < 8;
  const float bh = (GRU && wave == 0 && lane < 32 && live_row) ? bhh[(size_t)dir * NG * H + (size_t)gate * H + u0 + unit] : 0.f;
  float c_reg = (!GRU && owner) ? c_state[(size_t)dir * H + u0 + lane] : 0.f;
  float h_last = owner ? h_init[(size_t)dir * H + u0 + lane] : 0.f;
  if (tid == 0) s_abort = 0;
  for (int s = 0; s < len; ++s) {
    const int t = dir == 0 ? s : len - 1 - s;
    float gxv = 0.f;
    if (wave == 0 && lane < 32 && live_row) gxv = gxd[(size_t)t * NG * H];  // (requested before the gather: off the critical path)
    if (s == 0) {
      hs[2 * tid] = h_init[(size_t)dir * H + 2 * tid];
      hs[2 * tid + 1] = h_init[(size_t)dir * H + 2 * tid + 1];
    } else {
      const unsigned long long* src = xb + (size_t)((s - 1) & 1) * dirs * H + 2 * tid;
      const unsigned int want = epoch + (unsigned int)(s - 1);
      unsigned long long v0, v1;
      int spins = 0;
      for (;;) {
        v0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned int)(v0 >> 32) == want && (unsigned int)(v1 >> 32) == want) break;
        __builtin_amdgcn_s_sleep(2);
        ++spins;
        if (spins > kPersistSpins ||
            ((spins & 63) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_abort = 1;
          break;
        }
      }
      hs[2 * tid] = __uint_as_float((unsigned int)v0);
      hs[2 * tid + 1] = __uint_as_float((unsigned int)v1);
    }
    __syncthreads();
    if (s_abort) break;  // (uniform: written before the barrier)
    float acc = 0.f;
    {
      const float* hp = hs + seg * 64;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hp + 4 * i);
        acc = fmaf(w[4 * i], hv[0], acc);
        acc = fmaf(w[4 * i + 1], hv[1], acc);
        acc = fmaf(w[4 * i + 2], hv[2], acc);
        acc = fmaf(w[4 * i + 3], hv[3], acc);
      }
    }
    acc += __shfl_xor(acc, 32);
    if (lane < 32) part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && GRU) {
      // recurrent part (+ b_hh) and input part of every gate row stay apart (k_gru_step): lane u < 8 owns unit u, its
      // gates r, z, candidate sit in lanes u, 8 + u, 16 + u
      float rec = bh;
      if (lane < 32) {
#pragma unroll
        for (int wv = 0; wv < kPersistThreads / 64; ++wv) rec += part[wv][lane];
      }
      const int u = lane & 7;
      const float rr = __shfl(rec, u), rz = __shfl(rec, 8 + u), rc = __shfl(rec, 16 + u);
      const float xr = __shfl(gxv, u), xz = __shfl(gxv, 8 + u), xc = __shfl(gxv, 16 + u);
      if (lane < 8) {
        const float gr = 1.0f / (1.0f + expf(-(xr + rr)));
        const float gz = 1.0f / (1.0f + expf(-(xz + rz)));
        const float cand = tanhf(xc + gr * rc);
        h_last = (hs[u0 + lane] - cand) * gz + cand;
        const unsigned long long granule = ((unsigned long long)(epoch + (unsigned int)s) << 32) | (unsigned long long)__float_as_uint(h_last);
        __hip_atomic_store(xb + (size_t)(s & 1) * dirs * H + u0 + lane, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        y[(size_t)t * (size_t)(dirs * H) + (size_t)dir * H + u0 + lane] = h_last;
      }
    }
    if (wave == 0 && !GRU) {
      float g = gxv;
      if (lane < 32) {
#pragma unroll
        for (int wv = 0; wv < kPersistThreads / 64; ++wv) g += part[wv][lane];
      }
      // lane u < 8 owns unit u: its gates sit in lanes u, 8 + u, 16 + u, 24 + u
      const float gi = __shfl(g, lane & 7), gf = __shfl(g, 8 + (lane & 7)), gg = __shfl(g, 16 + (lane & 7)),
                  go = __shfl(g, 24 + (lane & 7));
      if (lane < 8) {
        const float ig = 1.0f / (1.0f + expf(-gi));
        const float fg = 1.0f / (1.0f + expf(-gf));
        const float cg = tanhf(gg);
        const float og = 1.0f / (1.0f + expf(-go));
        c_reg = fg * c_reg + ig * cg;
        h_last = og * tanhf(c_reg);
        const unsigned long long granule = ((unsigned long long)(epoch + (unsigned int)s) << 32) | (unsigned long long)__float_as_uint(h_last);
        __hip_atomic_store(xb + (size_t)(s & 1) * dirs * H + u0 + lane, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        y[(size_t)t * (size_t)(dirs * H) + (size_t)dir * H + u0 + lane] = h_last;
      }
    }
  }
  if (owner) {
    if (!GRU) c_state[(size_t)dir * H + u0 + lane] = c_reg;  // (GRU: the c box is handed through unchanged by the host)
    h_final[(size_t)dir * H + u0 + lane] = h_last;
  }
}

// -> false: the launch would not be wholly resident (hipOccupancy x CUs < workgroups); the caller takes the per-step route
bool lstm_persist_fits(int H, int dirs) {
  if (H != 1024) return false;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lstm_persist<false>, kPersistThreads, 0) != hipSuccess) return false;
  // (one workgroup per CU is the intended shape; a partitioned device with fewer CUs takes the per-step kernels)
  return per_cu >= 1 && cus >= (H / 8) * dirs;
}
void launch_lstm_persist(const float* gx, const float* whh, const float* bhh, bool gru, const float* h_init, float* c_state,
                         float* h_final, float* y, const int32_t* lens, int T, int H, int dirs, unsigned long long* xbuf,
                         unsigned int epoch, int* abort_flag, hipStream_t st) {
  (void)H;
  if (gru)
    PPASR_LAUNCH(k_lstm_persist<true>, dim3(1024 / 8, dirs), dim3(kPersistThreads), 0, st, gx, whh, bhh, h_init, c_state,
                 h_final, y, lens, T, dirs, xbuf, epoch, abort_flag);
  else
    PPASR_LAUNCH(k_lstm_persist<false>, dim3(1024 / 8, dirs), dim3(kPersistThreads), 0, st, gx, whh, bhh, h_init, c_state,
                 h_final, y, lens, T, dirs, xbuf, epoch, abort_flag);
}

// ---- test hook (ppasr_debug_occupy_cus): `n_wg` workgroups that each take a whole CU's registers (1 024 threads x 128
// VGPRs) and spin for `ms` milliseconds -- the chip is then PARTLY held, which is the situation the persistent recurrence
// cannot wait out (its grid needs every workgroup resident): tests force its give-up path with this ----
__global__ __launch_bounds__(1024) void k_occupy(long long ticks, float* __restrict__ sink) {
  float r[96];  // (held across the spin: the kernel's register budget is what keeps other workgroups off the CU)
#pragma unroll
  for (int i = 0; i < 96; ++i) r[i] = (float)(threadIdx.x + i);
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < 96; ++i) r[i] = r[i] * 1.0000001f + 1e-9f;
    __builtin_amdgcn_s_sleep(8);
  }
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 96; ++i) a += r[i];
  if (a == 12345.678f) sink[0] = a;  // (keeps the registers live)
}
void launch_occupy(int n_wg, int ms, float* sink, hipStream_t st) {
  PPASR_LAUNCH(k_occupy, dim3(n_wg), dim3(1024), 0, st, (long long)ms * 100000ll /* 100 MHz wall clock */, sink);
}

// [B][H] row-major <-> the fragment order of k_lstm_wave's state buffers (initial / final state boxes)
__global__ __launch_bounds__(256) void k_state_reorder(const float* __restrict__ src, float* __restrict__ dst, int B, int H,
                                                       int to_frag) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * H) return;
  const int bb = (int)(i / H), k = (int)(i - (size_t)bb * H);
  const size_t fi = ((size_t)(bb >> 5) * (H / 8) + (k >> 3)) * 256 + ((bb & 31) + 32 * ((k & 7) >> 2)) * 4 + (k & 3);
  if (to_frag) dst[fi] = src[i];
  else dst[i] = src[fi];
}
void launch_state_reorder(const float* src, float* dst, int B, int H, bool to_frag, hipStream_t st) {
  PPASR_LAUNCH(k_state_reorder, dim3((unsigned)(((size_t)B * H + 255) / 256)), dim3(256), 0, st, src, dst, B, H, to_frag ? 1 : 0);
}
void launch_ln_wide(float* x, const float* g, const float* b, int M, int N, hipStream_t st) {
  PPASR_LAUNCH(k_ln_wide, dim3((M + 3) / 4), dim3(256), 0, st, x, g, b, M, N);
}

}  // namespace ppasr
