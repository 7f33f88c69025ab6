// ctc_head_kernels.hip -- CTC head (after_norm -> ctc_lo -> softmax statistics + argmax, never materialising the
// probability tensor on the greedy route), row softmax, pad-row zeroing, frame argmax and the greedy collapse.
// (Split from conformer_kernels.hip in round 5.)
// Reference: ppasr/model_utils/conformer/encoder.py:201, model_utils/loss/ctc.py:62-70, decoders/ctc_greedy_decoder.py:6-31.
#include <cstdlib>

#include "conformer_kernels.h"
#include "launch.h"
#include "phases.h"
#include "h3.h"

#include <math.h>

namespace ppasr {

// -------------------------------------------------------------------------------------
// CTC head: after_norm (encoder.py:201-202) -> ctc_lo (loss/ctc.py:27) -> per-frame softmax
// statistics + argmax (loss/ctc.py:62-70, ctc_greedy_decoder.py:21-22) without materialising
// the [B,T',V] probability tensor.  Optional logits tap (LOGITS).
// Wave w walks vocabulary tiles w, w+8, ...; each lane keeps a running (max, sum-exp, argmax)
// for its 16 rows, merged across lanes / waves at the end (ties -> lowest index = numpy argmax).
// -------------------------------------------------------------------------------------
// H3: the vocabulary tiles on the fp16 x3 route (h3.h; hw.w is then the re-packed weight; the operand planes replace bufA
// and the reduction arrays move 512 B back)
template <bool LOGITS, bool H3>
__device__ __forceinline__ void ctc_head_body(const float* __restrict__ x, const HeadW& hw, float* __restrict__ logits,
                                              int32_t* __restrict__ fr_argmax, float* __restrict__ fr_maxprob,
                                              float* __restrict__ row_max, float* __restrict__ row_sum, int M, const PadSkip& ps,
                                              float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  // gridDim.y > 1 (under-filled launches): workgroup y walks vocabulary tiles wave + 8 (y + gridDim.y k) and leaves its
  // per-row (max, sum-exp, argmax) in part[3][gridDim.y][M]; k_ctc_merge combines the slices
  const int ny = gridDim.y, y = blockIdx.y;
  float* bufA = smem;                                          // [32][260]
  float* redM = bufA + (H3 ? kH3TileBytes / 4 : kRows * kLda);  // [8][32]
  float* redS = redM + kWaves * 32;                            // [8][32]
  int* redI = reinterpret_cast<int*>(redS + kWaves * 32);      // [8][32]
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  const int V = hw.V;
  BRing<1> ring;
  if (wave + 8 * y < hw.n_tiles) ring_prime(ring, hw.w + (size_t)(wave + 8 * y) * kTs256, 0);
  rb_load_rows(bufA, kLda, x + (size_t)r0 * kD, kRows, valid);
  if (hw.ln_g) rb_layernorm(bufA, bufA, kLda, kRows, hw.ln_g, hw.ln_b, 1e-5f);
  __syncthreads();
  if constexpr (H3) h3_planes_from_tile(bufA, reinterpret_cast<_Float16*>(bufA));
  // Transposed tiles (rb_gemm SWAP): lane = row l&31, its 16 registers = 16 columns of the vocabulary tile in increasing
  // order (col = 8(r>>2) + 4(l>>5) + (r&3)).  The running (max, sum-exp, argmax) of a row is then ONE triple per lane,
  // updated per tile with in-lane arithmetic: tile max (v_max3), one rescale of the running sum, 16 exponentials, and
  // an index scan only when the tile raises the maximum (rare after the first tiles) -- against a triple per (row,
  // column lane) with two exponentials per element and a 5-step cross-lane merge per row at the end.
  float mx = -INFINITY, sm = 0.f;
  int ix = 0x7fffffff;
  const int l31 = lane & 31, hh = lane >> 5;
  constexpr float kLog2e = 1.4426950408889634f;
  const int tstep = kWaves * ny;
  for (int tile = wave + 8 * y; tile < hw.n_tiles; tile += tstep) {
    f32x16 acc[1][1];
    acc_zero(acc);
    const f32x4* seg = hw.w + (size_t)tile * kTs256;
    if constexpr (H3) {
      rb_gemm_h3(reinterpret_cast<const _Float16*>(bufA), seg, tile + tstep < hw.n_tiles ? seg + (size_t)tstep * kTs256 : nullptr,
                 ring, acc[0][0]);
      acc[0][0] *= kH3Inv;
    } else {
      rb_gemm<1, 1, kG256, kPF, NoSide, true>(bufA, kLda, seg, 0, tile + tstep < hw.n_tiles ? seg + (size_t)tstep * kTs256 : nullptr,
                                              0, ring, acc);
    }
    const int c0 = tile * 32 + 4 * hh;  // column of register 0
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(hw.b + c0 + 8 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * q + e] = acc[0][0][4 * q + e] + bq[e];
    }
    if (tile == hw.n_tiles - 1 && (V & 31)) {  // padded columns of the last tile never win and add exp(-inf) = 0
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + 8 * (r >> 2) + (r & 3) >= V) v[r] = -INFINITY;
    }
    if (LOGITS) {
      if (l31 < valid) {
        float* lrow = logits + (size_t)(r0 + l31) * V;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int col = c0 + 8 * (r >> 2) + (r & 3);
          if (col < V) lrow[col] = v[r];
        }
      }
    }
    float tmax = max3f(v[0], v[1], v[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = max3f(tmax, v[r], v[r + 1]);
    tmax = fmaxf(tmax, v[15]);
    if (tmax > mx) {  // first column holding the new maximum (lowest index wins ties: numpy argmax)
#pragma unroll
      for (int r = 15; r >= 0; --r) ix = (v[r] == tmax) ? c0 + 8 * (r >> 2) + (r & 3) : ix;
    }
    const float mn = fmaxf(mx, tmax);
    // (mx = -inf before the first tile: exp2(-inf) = 0; mn is finite from then on -- every tile has a real column)
    sm *= __builtin_amdgcn_exp2f((mx - mn) * kLog2e);
    f32x2 ps = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {  // (subtract first: logits reach +-30, a fused v * log2e - mn * log2e would round at 2e-6)
      const f32x2 t = (f32x2{v[r], v[r + 1]} - f32x2{mn, mn}) * f32x2{kLog2e, kLog2e};
      ps += f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    }
    sm += ps[0] + ps[1];
    mx = mn;
  }
  // the two lane halves of a row (different columns), then one triple per (wave, row)
  {
    const float m2 = __shfl_xor(mx, 32), s2 = __shfl_xor(sm, 32);
    const int i2 = __shfl_xor(ix, 32);
    const float mn = fmaxf(mx, m2);
    const float sa = (mx == -INFINITY) ? 0.f : sm * __expf(mx - mn);
    const float sb = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
    const bool take2 = (m2 > mx) || (m2 == mx && i2 < ix);
    if (hh == 0) {
      redM[wave * 32 + l31] = mn;
      redS[wave * 32 + l31] = sa + sb;
      redI[wave * 32 + l31] = take2 ? i2 : ix;
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int row = threadIdx.x;
    float m = redM[row], s = redS[row];
    int i = redI[row];
    for (int wv = 1; wv < kWaves; ++wv) {
      float m2 = redM[wv * 32 + row], s2 = redS[wv * 32 + row];
      int i2 = redI[wv * 32 + row];
      float mn = fmaxf(m, m2);
      float sa = (m == -INFINITY) ? 0.f : s * __expf(m - mn);
      float sb = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
      bool take2 = (m2 > m) || (m2 == m && i2 < i);
      i = take2 ? i2 : i;
      m = mn;
      s = sa + sb;
    }
    if (row < valid) {
      if (ny > 1) {
        part[(size_t)y * M + r0 + row] = m;
        part[((size_t)ny + y) * M + r0 + row] = s;
        reinterpret_cast<int*>(part)[((size_t)2 * ny + y) * M + r0 + row] = i;
      } else {
        if (fr_argmax) fr_argmax[r0 + row] = i;
        if (fr_maxprob) fr_maxprob[r0 + row] = 1.0f / s;
        if (row_max) row_max[r0 + row] = m;
        if (row_sum) row_sum[r0 + row] = s;
      }
    }
  }
}
__global__ void k_ctc_merge(const float* __restrict__ part, int ny, int32_t* __restrict__ fr_argmax,
                            float* __restrict__ fr_maxprob, float* __restrict__ row_max, float* __restrict__ row_sum, int M,
                            PadSkip ps) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= M) return;
  if (pad_block_skippable(ps, row & ~(kRows - 1), kRows, M)) return;
  float m = part[row], s = part[(size_t)ny * M + row];
  int i = reinterpret_cast<const int*>(part)[(size_t)2 * ny * M + row];
  for (int y = 1; y < ny; ++y) {
    const float m2 = part[(size_t)y * M + row], s2 = part[((size_t)ny + y) * M + row];
    const int i2 = reinterpret_cast<const int*>(part)[((size_t)2 * ny + y) * M + row];
    const float mn = fmaxf(m, m2);
    const float sa = (m == -INFINITY) ? 0.f : s * __expf(m - mn);
    const float sb = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
    const bool take2 = (m2 > m) || (m2 == m && i2 < i);
    i = take2 ? i2 : i;
    m = mn;
    s = sa + sb;
  }
  if (fr_argmax) fr_argmax[row] = i;
  if (fr_maxprob) fr_maxprob[row] = 1.0f / s;
  if (row_max) row_max[row] = m;
  if (row_sum) row_sum[row] = s;
}
template <bool LOGITS>
__global__ __launch_bounds__(kThreads) void k_ctc_head(const float* __restrict__ x, HeadW hw, float* __restrict__ logits,
                                                       int32_t* __restrict__ fr_argmax, float* __restrict__ fr_maxprob,
                                                       float* __restrict__ row_max, float* __restrict__ row_sum, int M,
                                                       PadSkip ps, float* __restrict__ part) {
  ctc_head_body<LOGITS, false>(x, hw, logits, fr_argmax, fr_maxprob, row_max, row_sum, M, ps, part);
}
template <bool LOGITS>
__global__ __launch_bounds__(kThreads) void k_ctc_head_h3(const float* __restrict__ x, HeadW hw, float* __restrict__ logits,
                                                          int32_t* __restrict__ fr_argmax, float* __restrict__ fr_maxprob,
                                                          float* __restrict__ row_max, float* __restrict__ row_sum, int M,
                                                          PadSkip ps, float* __restrict__ part) {
  ctc_head_body<LOGITS, true>(x, hw, logits, fr_argmax, fr_maxprob, row_max, row_sum, M, ps, part);
}
constexpr size_t kLdsCtc = (kRows * kLda + 3 * kWaves * 32) * sizeof(float);
void launch_ctc_head(const float* x, const HeadW& hw, float* logits, int32_t* fr_argmax, float* fr_maxprob, float* row_max,
                     float* row_sum, int M, hipStream_t st, const PadSkip& ps, int n_slices, float* part, bool h3) {
  const int ny = (n_slices > 1 && part) ? n_slices : 1;
  dim3 grid((M + kRows - 1) / kRows, ny);
  const size_t lds = ragged_lds(kLdsCtc + (h3 ? 512 : 0), ps, (int)(grid.x * grid.y));
  if (h3 && logits)  // (hw: the head's fp16 x3 view)
    PPASR_LAUNCH(k_ctc_head_h3<true>, grid, dim3(kThreads), lds, st, x, hw, logits, fr_argmax, fr_maxprob, row_max, row_sum, M, ps,
                 part);
  else if (h3)
    PPASR_LAUNCH(k_ctc_head_h3<false>, grid, dim3(kThreads), lds, st, x, hw, logits, fr_argmax, fr_maxprob, row_max, row_sum, M, ps,
                 part);
  else if (logits)
    PPASR_LAUNCH(k_ctc_head<true>, grid, dim3(kThreads), lds, st, x, hw, logits, fr_argmax, fr_maxprob, row_max,
                       row_sum, M, ps, part);
  else
    PPASR_LAUNCH(k_ctc_head<false>, grid, dim3(kThreads), lds, st, x, hw, logits, fr_argmax, fr_maxprob, row_max,
                       row_sum, M, ps, part);
  if (ny > 1)
    PPASR_LAUNCH(k_ctc_merge, dim3((M + 255) / 256), dim3(256), 0, st, part, ny, fr_argmax, fr_maxprob, row_max, row_sum,
                       M, ps);
}

// probs = softmax(logits) recomputed exactly (max, then exp(x - max) / sum -- not the head's running statistics) in place.
// ONE workgroup per row with the row in registers: one read and one write of the row, 256 lanes on it (the one-wave-per-row
// kernel this replaces read it three times: a 10 s utterance 33 -> 8 us, 15 936 rows of cfg4 96 -> 60 us).  NV = values per
// thread (V <= 256 NV); NV = 0: any V, the row re-read from memory in each pass.
template <int NV>
__global__ __launch_bounds__(256) void k_softmax_row_wg(float* __restrict__ p, int M, int V, PadSkip ps) {
  const int row = blockIdx.x;
  // ragged batch: the CTC head skipped whole 32-row blocks; the same blocks keep their cleared (all-zero) rows here
  if (pad_block_skippable(ps, row & ~(kRows - 1), kRows, M)) return;
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  float* x = p + (size_t)row * V;
  constexpr int NR = NV > 0 ? NV : 1;
  float v[NR];
  float m = -INFINITY;
  if (NV > 0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int c = tid + 256 * i;
      v[i] = c < V ? x[c] : -INFINITY;
      m = fmaxf(m, v[i]);
    }
  } else {
    for (int c = tid; c < V; c += 256) m = fmaxf(m, x[c]);
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  if (NV > 0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      v[i] = expf(v[i] - m);  // (columns past V: exp(-inf) = 0)
      s += v[i];
    }
  } else {
    for (int c = tid; c < V; c += 256) {
      const float e = expf(x[c] - m);
      x[c] = e;
      s += e;
    }
  }
  s = wave_sum(s);
  if (lane == 0) red[4 + wave] = s;
  __syncthreads();
  s = (red[4] + red[5]) + (red[6] + red[7]);
  if (NV > 0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int c = tid + 256 * i;
      if (c < V) x[c] = v[i] / s;
    }
  } else {
    for (int c = tid; c < V; c += 256) x[c] = x[c] / s;
  }
}
void launch_softmax_from_stats(float* probs_inout, const float*, const float*, int M, int V, hipStream_t st,
                               const PadSkip& ps) {
  if (V <= 256 * 8) PPASR_LAUNCH(k_softmax_row_wg<8>, dim3(M), dim3(256), 0, st, probs_inout, M, V, ps);
  else if (V <= 256 * 20) PPASR_LAUNCH(k_softmax_row_wg<20>, dim3(M), dim3(256), 0, st, probs_inout, M, V, ps);
  else PPASR_LAUNCH(k_softmax_row_wg<0>, dim3(M), dim3(256), 0, st, probs_inout, M, V, ps);
}

__global__ __launch_bounds__(256) void k_zero_pad_rows(float* __restrict__ probs, float* __restrict__ logits,
                                                       int32_t* __restrict__ fa, float* __restrict__ fp,
                                                       const int64_t* __restrict__ lens, int M, int Tp, int mul, int V) {
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= M) return;
  const int b = row / Tp, t = row - b * Tp;
  if ((int64_t)mul * t < lens[b]) return;
  const int lane = lane_id();
  if (probs)
    for (int c = lane; c < V; c += 64) probs[(size_t)row * V + c] = 0.f;
  if (logits)
    for (int c = lane; c < V; c += 64) logits[(size_t)row * V + c] = 0.f;
  if (lane == 0) {
    if (fa) fa[row] = 0;
    if (fp) fp[row] = 0.f;
  }
}
void launch_zero_pad_rows(float* probs, float* logits, int32_t* fr_argmax, float* fr_maxprob, const int64_t* lens, int B,
                          int Tp, int mul, int V, hipStream_t st) {
  const int M = B * Tp;
  PPASR_LAUNCH(k_zero_pad_rows, dim3((M + 3) / 4), dim3(256), 0, st, probs, logits, fr_argmax, fr_maxprob, lens, M, Tp,
                     mul, V);
}

// =====================================================================================
// CTC greedy decode (decoders/ctc_greedy_decoder.py:6-31)
// =====================================================================================
// stage 1 from materialised probabilities: np.argmax(axis=1) (first max wins) + prob at argmax
__global__ __launch_bounds__(256) void k_frame_argmax(const float* __restrict__ probs, int32_t* __restrict__ fr_argmax,
                                                      float* __restrict__ fr_maxprob, int M, int V) {
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= M) return;
  const int lane = lane_id();
  const float* x = probs + (size_t)row * V;
  float m = -INFINITY;
  int idx = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    float v = x[c];
    if (v > m || idx == 0x7fffffff) {  // first element always taken (handles -inf / NaN-free inputs)
      m = v;
      idx = c;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float m2 = __shfl_xor(m, o);
    int i2 = __shfl_xor(idx, o);
    bool take2 = (i2 != 0x7fffffff) && (idx == 0x7fffffff || m2 > m || (m2 == m && i2 < idx));
    if (take2) {
      m = m2;
      idx = i2;
    }
  }
  if (lane == 0) {
    fr_argmax[row] = idx;
    fr_maxprob[row] = m;
  }
}
void launch_frame_argmax(const float* probs, int32_t* fr_argmax, float* fr_maxprob, int M, int V, hipStream_t st) {
  PPASR_LAUNCH(k_frame_argmax, dim3((M + 3) / 4), dim3(256), 0, st, probs, fr_argmax, fr_maxprob, M, V);
}

// stage 2: groupby-collapse, drop blank, score = mean(non-blank max probs)*100; one wave per utterance
__global__ __launch_bounds__(64) void k_ctc_collapse(const int32_t* __restrict__ fr_argmax, const float* __restrict__ fr_maxprob,
                                                     const int32_t* __restrict__ frame_lens, int Tp, int blank,
                                                     int32_t* __restrict__ tokens, int32_t* __restrict__ n_tokens,
                                                     double* __restrict__ score) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int n = frame_lens ? frame_lens[b] : Tp;
  n = max(0, min(n, Tp));
  const int32_t* ids = fr_argmax + (size_t)b * Tp;
  const float* pr = fr_maxprob + (size_t)b * Tp;
  int32_t* out = tokens + (size_t)b * Tp;
  int count = 0;
  int prev_last = -2;
  double dsum = 0.0;
  int nnb = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const bool in = i < n;
    const int id = in ? ids[i] : -3;
    int prev = __shfl_up(id, 1);
    if (lane == 0) prev = prev_last;
    const bool nonblank = in && id != blank;
    const bool keep = nonblank && id != prev;
    if (nonblank) {
      dsum += (double)pr[i];
      nnb += 1;
    }
    unsigned long long mask = __ballot(keep);
    int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
    if (keep) out[pos] = id;
    count += __popcll(mask);
    prev_last = __shfl(id, 63);
  }
  // deterministic tree reduction of the fp64 partial sums
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dsum += __shfl_xor(dsum, o);
    nnb += __shfl_xor(nnb, o);
  }
  for (int i = count + lane; i < Tp; i += 64) out[i] = -1;
  if (lane == 0) {
    n_tokens[b] = count;
    score[b] = nnb > 0 ? (dsum / (double)nnb) * 100.0 : 0.0;
  }
}
void launch_ctc_collapse(const int32_t* fr_argmax, const float* fr_maxprob, const int32_t* frame_lens, int B, int Tp,
                         int blank, int32_t* tokens, int32_t* n_tokens, double* score, hipStream_t st) {
  PPASR_LAUNCH(k_ctc_collapse, dim3(B), dim3(64), 0, st, fr_argmax, fr_maxprob, frame_lens, Tp, blank, tokens,
                     n_tokens, score);
}
unsigned int* ctc_head_h3_ovf_counter() { return h3_ovf_counter(); }

hipError_t configure_ctc_head_kernels() {
  hipError_t e = hipSuccess;
#define SET_LDS(fn, bytes)                                                                                     \
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
  if (e != hipSuccess) return e;
  SET_LDS(k_ctc_head<true>, kLdsExclusive);  // (>= kLdsCtc: see ragged_lds)
  SET_LDS(k_ctc_head_h3<true>, kLdsExclusive);
  SET_LDS(k_ctc_head_h3<false>, kLdsExclusive);
  SET_LDS(k_ctc_head<false>, kLdsExclusive);
#undef SET_LDS
  return hipSuccess;
}

}  // namespace ppasr
