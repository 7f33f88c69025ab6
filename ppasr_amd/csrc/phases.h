// phases.h -- row-block phase functions shared by the encoder kernels (Conformer, Squeezeformer,
// Efficient-Conformer): FFN with LDS-resident hidden chunks, residual epilogue, causal depthwise
// conv, pad-row predicate.  See rowblock.h for the execution model.
#pragma once
#include "rowblock.h"

namespace ppasr {

// Epilogue slice of the previous W1 tile, interleaved with the MFMAs of the next one: H[row][col] = swish(acc + b1).
// The W1 units run with swapped MFMA operands (rb_gemm SWAP): lane = row, register quad q = hidden columns
// 8q + 4hh .. +3 of the wave's 32, so a quad is ONE 16-byte LDS store and its four values two packed-math pairs --
// 17 instructions per quad, 68 per tile, against 7 per value / 112 per tile one column per lane.  (Every VALU / LDS
// instruction of a wave takes issue cycles from the SIMD's matrix pipe: tools/microbench_mfma.hip, "side" rows.)
// Quad q is handled during k-groups 8q (first pair) and 8q + 4 (second pair + store).
struct SwishSide {
  const f32x16& acc;
  float* dst;           // hb + (lane & 31) * kLda + wave * 32 + 4 * (lane >> 5)
  const f32x4 (&bias)[4];  // b1 of this lane's columns, quad by quad
  mutable f32x2 lo;
  __device__ __forceinline__ void operator()(int g) const {
    const int q = g >> 3;
    if ((g & 7) == 0) lo = swish2(f32x2{acc[4 * q] + bias[q][0], acc[4 * q + 1] + bias[q][1]});
    if ((g & 7) == 4) {
      const f32x2 hi = swish2(f32x2{acc[4 * q + 2] + bias[q][2], acc[4 * q + 3] + bias[q][3]});
      *reinterpret_cast<f32x4*>(dst + 8 * q) = f32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  }
};

// PositionwiseFeedForward (positionwise.py:32-39): acc2 += swish(A*W1 + b1) * W2.  The hidden
// dimension is processed in 256-wide chunks that never leave LDS (double-buffered bufH); wave w
// owns hidden columns [32w,32w+32) of each chunk and output columns [32w,32w+32).
// Weight stream order: W1(0), W1(1), W2(0), W1(2), W2(1), ..., W2(n-1), then `after`.
// The swish epilogue of chunk c runs inside the W1(c+1) MFMA stream.
// c0 / n_total: the call covers hidden chunks [c0, c0 + n_chunks) of a layer with n_total chunks (a slice of the hidden
// dimension = a partial sum of the output, k_ffn_part); default = all of them.
// T2: the W2 units run swapped as well -- acc2 is then the TRANSPOSED output tile (lane = row; residual_epilogue_t)
template <bool T2 = false>
__device__ __forceinline__ void ffn_phase(const float* bufA, float* bufH, const f32x4* __restrict__ w1,
                                          const float* __restrict__ b1, const f32x4* __restrict__ w2, int n_chunks,
                                          const f32x4* __restrict__ after, BRing<1>& ring, f32x16 (&acc2)[1][1],
                                          int c0 = 0, int n_total = -1) {
  const int lane = lane_id(), wave = wave_id();
  const int ts2 = (n_total > 0 ? n_total : n_chunks) * 32 * 64;  // W2: K = hidden
  b1 += c0 * 256;
  auto w1seg = [&](int c) { return w1 + (size_t)((c0 + c) * 8 + wave) * kTs256; };
  auto w2seg = [&](int c) { return w2 + (size_t)wave * ts2 + (size_t)(c0 + c) * 32 * 64; };
  f32x16 cur[1][1], nx[1][1];
  acc_zero(cur);
  rb_gemm<1, 1, kG256, kPF, NoSide, true>(bufA, kLda, w1seg(0), 0, n_chunks > 1 ? w1seg(1) : w2seg(0), 0, ring, cur);
  const int hoff = (lane & 31) * kLda + wave * 32 + 4 * (lane >> 5);
  for (int c = 0; c < n_chunks; ++c) {
    float* hb = bufH + (c & 1) * kRows * kLda;
    f32x4 bias[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bias[q] = *reinterpret_cast<const f32x4*>(b1 + c * 256 + wave * 32 + 8 * q + 4 * (lane >> 5));
    if (c + 1 < n_chunks) {
      acc_zero(nx);
#ifdef PPASR_ABLATE_SWISH
      rb_gemm<1, 1, kG256, kPF, NoSide, true>(bufA, kLda, w1seg(c + 1), 0, w2seg(c), 0, ring, nx);
#else
      rb_gemm<1, 1, kG256, kPF, SwishSide, true>(bufA, kLda, w1seg(c + 1), 0, w2seg(c), 0, ring, nx,
                                                 SwishSide{cur[0][0], hb + hoff, bias, f32x2{0.f, 0.f}});
#endif
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 lo = swish2(f32x2{cur[0][0][4 * q] + bias[q][0], cur[0][0][4 * q + 1] + bias[q][1]});
        const f32x2 hi = swish2(f32x2{cur[0][0][4 * q + 2] + bias[q][2], cur[0][0][4 * q + 3] + bias[q][3]});
        *reinterpret_cast<f32x4*>(hb + hoff + 8 * q) = f32x4{lo[0], lo[1], hi[0], hi[1]};
      }
    }
    if (c < 8) PPASR_TS(16 + 2 * c);
    if (c < 8) PPASR_WAVE_TS(2 * c);
#ifndef PPASR_ABLATE_FFN_BARRIER
    __syncthreads();
#endif
    if (c < 8) PPASR_TS(17 + 2 * c);
    if (c < 8) PPASR_WAVE_TS(2 * c + 1);
    const f32x4* nseg = (c + 2 < n_chunks) ? w1seg(c + 2) : (c + 1 < n_chunks ? w2seg(c + 1) : after);
    rb_gemm<1, 1, kG256, kPF, NoSide, T2>(hb, kLda, w2seg(c), 0, nseg, 0, ring, acc2);
    cur[0][0] = nx[0][0];
  }
}

// bufX[row][col] += scale * (acc + bias[col])    (residual update, each element owned by one lane)
__device__ __forceinline__ void residual_epilogue(float* bufX, const f32x16 (&acc)[1][1], const float* __restrict__ bias,
                                                  float scale) {
  const int lane = lane_id(), wave = wave_id();
  const int col = wave * 32 + (lane & 31);
  const float bv = bias[col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float* p = bufX + acc_row(r, lane) * kLda + col;
    *p = *p + scale * (acc[0][0][r] + bv);
  }
}

// the same on a transposed accumulator (rb_gemm SWAP: lane = row, register quad q = columns wave*32 + 8q + 4hh .. +3):
// 4 16-byte LDS read-modify-writes per lane instead of 16 4-byte ones
__device__ __forceinline__ void residual_epilogue_t(float* bufX, const f32x16 (&acc)[1][1], const float* __restrict__ bias,
                                                    float scale) {
  const int lane = lane_id(), wave = wave_id();
  const int cq = wave * 32 + 4 * (lane >> 5);
  float* row = bufX + (lane & 31) * kLda + cq;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + cq + 8 * q);
    f32x4 x = *reinterpret_cast<const f32x4*>(row + 8 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = x[e] + scale * (acc[0][0][4 * q + e] + bv[e]);
    *reinterpret_cast<f32x4*>(row + 8 * q) = x;
  }
}

struct PadRows {  // conv-module pad masking (convolution.py:104-106,138-140): frame t of utterance b is PAD iff
                  // mul*t >= len[b]  (mul = 4 after the conv front-end, 8 on time-reduced rows)
  const int64_t* lens;
  int r0, Tp, M;
  int mul = 4;
  __device__ __forceinline__ bool operator()(int row) const {
    if (!lens) return false;
    int m = r0 + row;
    if (m >= M) return false;
    int b = m / Tp, t = m - b * Tp;
    return mul * (int64_t)t >= lens[b];
  }
};


// Causal depthwise conv over the GLU output g [M][256] (row = frame): KS taps, left context KS-1.
// Frames before the utterance start read `gp` = GLU(pointwise_conv1(0)) because the reference
// zero-pads BEFORE pointwise_conv1 (convolution.py:108-126); with STREAM (rows = the frames of one chunk per
// session, Tp frames each) they come from the session's cache rows g_hist [session][KS-1][256] instead.
// The block's input window (KS-1 halo rows + 32 rows) and the tap weights are staged in LDS
// (win_lds: (KS-1+32) x kLda floats, w_lds: KS x kLda floats -- both buffers are free at this point of
// the kernels), so the phase needs few registers whatever KS is.  Output (conv + bias) -> bufA rows.
// `left` = frames of left context: KS-1 for the causal module (lorder, convolution.py:47-49), (KS-1)/2 for the
// non-causal one (streaming=False models), whose depthwise conv zero-pads its INPUT symmetrically
// (convolution.py:50-52: padding=(k-1)//2), so out-of-utterance taps read 0 instead of GLU(bias).
template <int KS, bool STREAM>
__device__ __forceinline__ void dwconv_phase(const float* __restrict__ g, const float* __restrict__ g_hist, float* bufA,
                                             float* win_lds, float* w_lds, const float* __restrict__ dw_w,
                                             const float* __restrict__ dw_b, const float* __restrict__ glu_pad, int r0,
                                             int M, int Tp, int left = KS - 1) {
  const int lane = lane_id(), wave = wave_id();
  constexpr int LO = KS - 1;
  constexpr int RW = kRows / kWaves;
  const bool causal = (left == LO);
  for (int q = wave; q < LO + kRows; q += kWaves) {
    const int mq = r0 - left + q;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (mq >= 0 && mq < M) v = *reinterpret_cast<const f32x4*>(g + (size_t)mq * kD + 4 * lane);
    *reinterpret_cast<f32x4*>(win_lds + q * kLda + 4 * lane) = v;
  }
  for (int j = wave; j < KS; j += kWaves)
    *reinterpret_cast<f32x4*>(w_lds + j * kLda + 4 * lane) = *reinterpret_cast<const f32x4*>(dw_w + j * kD + 4 * lane);
  f32x4 gp = *reinterpret_cast<const f32x4*>(glu_pad + 4 * lane);
  if (!causal) gp = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 bias = *reinterpret_cast<const f32x4*>(dw_b + 4 * lane);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int row = wave * RW + i;
    const int m = r0 + row;
    const int b = m / Tp, t = m - b * Tp;
    // STREAM: left-context frames of this row's session (window rows left of the session start belong to another
    // session's chunk, so they are replaced one by one)
    const float* hist = STREAM ? g_hist + ((size_t)b * LO) * kD + 4 * lane : nullptr;
    f32x4 acc = bias;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const f32x4 wj = *reinterpret_cast<const f32x4*>(w_lds + j * kLda + 4 * lane);
      f32x4 xv = *reinterpret_cast<const f32x4*>(win_lds + (row + j) * kLda + 4 * lane);
      const int tt = t - left + j;  // frame this tap reads inside the utterance / chunk
      if (STREAM) {
        if (tt < 0 && m < M) xv = *reinterpret_cast<const f32x4*>(hist + (size_t)(LO + tt) * kD);
      } else if (!(tt >= 0 && tt < Tp)) {
        xv = gp;
      }
      acc += wj * xv;
    }
    *reinterpret_cast<f32x4*>(bufA + row * kLda + 4 * lane) = acc;
  }
}

// The batched (non-streaming-chunk) form of the conv module's depthwise stage, fused with the LayerNorm + swish that
// follow it (convolution.py:129-134): wave w owns the RW = 4 CONSECUTIVE rows 4w .. 4w+3 of the block, so their
// KS-tap windows overlap and the whole input it needs is NW = RW + KS - 1 rows of g, loaded ONCE from global memory
// straight into registers (one coalesced 1 KiB row per load, lane = 4 columns), the tap weights likewise (KS x f32x4).
// The conv walks the window rows q: row q feeds tap j = q - i of output row i, so every output row still accumulates
// its taps in ascending order (bit-identical to dwconv_phase), and the "outside the utterance -> GLU(bias)" substitution
// is decided once per window row.  A wave spans the whole 256-wide row (4 columns per lane), so the
// LayerNorm runs on the accumulators themselves (wave sums) and the only LDS traffic is the store of the normalised rows.
// (The LDS-staged dwconv_phase -- window + weights through LDS, a barrier, 2 ds_read_b128 per FMA, then a second pass
// for the LayerNorm -- took 12.7 + 1.6 us of the dominant kernel; tools/phase_ts.py.)
// A wave whose 4 rows straddle an utterance boundary runs the walk twice, once per utterance, and keeps per row the
// result of the row's own utterance (needs Tp >= RW: the caller falls back to dwconv_phase below that).
template <int KS>
__device__ __forceinline__ void dwconv_ln_phase(const float* __restrict__ g, float* bufA, const float* __restrict__ dw_w,
                                                const float* __restrict__ dw_b, const float* __restrict__ glu_pad,
                                                const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps, int r0,
                                                int M, int Tp, int left) {
  const int lane = lane_id(), wave = wave_id();
  constexpr int LO = KS - 1, RW = kRows / kWaves, NW = RW + LO;
  const bool causal = (left == LO);
  const int q0 = wave * RW;
  f32x4 x[NW];
#pragma unroll
  for (int q = 0; q < NW; ++q) {
    const int mq = r0 - left + q0 + q;
    const int mc = min(max(mq, 0), M - 1);  // branch-free: clamped address, value masked below
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + (size_t)mc * kD + 4 * lane);
    x[q] = (mq >= 0 && mq < M) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 wt[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) wt[j] = *reinterpret_cast<const f32x4*>(dw_w + j * kD + 4 * lane);
  f32x4 gp = *reinterpret_cast<const f32x4*>(glu_pad + 4 * lane);
  if (!causal) gp = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 bias = *reinterpret_cast<const f32x4*>(dw_b + 4 * lane);
  const int m0 = r0 + q0;
  const int tq0 = m0 - (m0 / Tp) * Tp;           // frame of the wave's first row inside its utterance
  const int npass = (tq0 + RW - 1 < Tp) ? 1 : 2;  // 2: the rows straddle an utterance boundary
  f32x4 out[RW];
#pragma unroll 1
  for (int p = 0; p < npass; ++p) {
    const int tqp = tq0 - p * Tp;  // frame of row 0 relative to the start of utterance p of this wave
    f32x4 acc[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) acc[i] = bias;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int tt = tqp - left + q;  // frame the window row holds, relative to the utterance
      const f32x4 xq = (tt >= 0 && tt < Tp) ? x[q] : gp;
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const int j = q - i;
        if (j >= 0 && j < KS) acc[i] += wt[j] * xq;
      }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i)
      if (npass == 1 || (tqp + i >= 0 && tqp + i < Tp)) out[i] = acc[i];
  }
  const f32x4 gam = *reinterpret_cast<const f32x4*>(ln_g + 4 * lane);
  const f32x4 bet = *reinterpret_cast<const f32x4*>(ln_b + 4 * lane);
  ln_rows_inreg<true, RW>(out, gam, bet, ln_eps);
#pragma unroll
  for (int i = 0; i < RW; ++i) *reinterpret_cast<f32x4*>(bufA + (q0 + i) * kLda + 4 * lane) = out[i];
}

}  // namespace ppasr
