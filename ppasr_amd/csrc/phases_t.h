// phases_t.h -- row-block phase functions on TRANSPOSED accumulators, templated on the block form R (rbt.h: 32 or 16
// rows on 8 waves, kW16 = 32 rows on 16 waves): LayerNorm, feed-forward module with LDS-resident hidden chunks,
// residual / QKV epilogues, depthwise conv + LayerNorm + swish in registers, per-lane pad flags.  Every epilogue sees a
// wave's output tile as NQ quads per lane (quad q = row RBT<R>::row(q, lane), 4 consecutive columns from
// RBT<R>::col(q, lane, wave)), so the same code serves v_mfma_f32_32x32x2_f32 and both v_mfma_f32_16x16x4_f32 forms.
#pragma once
#include "phases.h"
#include "rbt.h"

namespace ppasr {

template <int R>
struct LaneT {  // this lane's place in a transposed wave tile
  int lane, wave;
  __device__ __forceinline__ LaneT() : lane(lane_id()), wave(wave_id()) {}
  __device__ __forceinline__ int row(int q) const { return RBT<R>::row(q, lane); }  // row (of the block) of quad q
  // first of the 4 consecutive columns (of the block's 256) of quad q
  __device__ __forceinline__ int col(int q) const { return RBT<R>::col(q, lane, wave); }
  __device__ __forceinline__ int off(int q) const { return row(q) * kLda + col(q); }  // in an LDS row buffer
  __device__ __forceinline__ int tile() const { return RBT<R>::tile(wave); }        // weight tile the wave streams
  // quads whose row is inside the block's `valid` rows (a lane's rows ascend with q)
  __device__ __forceinline__ int quads_ok(int valid) const {
    int n = 0;
#pragma unroll
    for (int q = 0; q < RBT<R>::NQ; ++q) n += row(q) < valid ? 1 : 0;
    return n;
  }
};

// LDS(row stride kLda) <-> global(row stride 256) with the wave -> row mapping of the LayerNorm below (wave w: rows w,
// w + WAVES, ...), so that a LayerNorm may follow a load or precede a store without a barrier
template <int R>
__device__ __forceinline__ void rbt_load_rows(float* dst, const float* __restrict__ src, int valid) {
  const int lane = lane_id();
  for (int row = wave_id(); row < RBT<R>::ROWS; row += RBT<R>::WAVES) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < valid) v = *reinterpret_cast<const f32x4*>(src + (size_t)row * kD + 4 * lane);
    *reinterpret_cast<f32x4*>(dst + row * kLda + 4 * lane) = v;
  }
}
template <int R>
__device__ __forceinline__ void rbt_store_rows(float* __restrict__ dst, const float* src, int valid) {
  const int lane = lane_id();
  for (int row = wave_id(); row < RBT<R>::ROWS && row < valid; row += RBT<R>::WAVES)
    *reinterpret_cast<f32x4*>(dst + (size_t)row * kD + 4 * lane) = *reinterpret_cast<const f32x4*>(src + row * kLda + 4 * lane);
}

// LayerNorm of the block's rows in LDS (rowblock.h rb_layernorm): wave w normalises rows w, w + WAVES, ...
template <int R, bool SWISH = false, typename ZeroRow = NoZero>
__device__ __forceinline__ void rbt_layernorm(const float* src, float* dst, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, float eps, ZeroRow zero_row = ZeroRow()) {
  const int lane = lane_id(), wave = wave_id();
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * lane);
  const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * lane);
  constexpr int RN = RBT<R>::ROWS / RBT<R>::WAVES, WV = RBT<R>::WAVES;
  f32x4 x[RN];
#pragma unroll
  for (int i = 0; i < RN; ++i) x[i] = *reinterpret_cast<const f32x4*>(src + (wave + i * WV) * kLda + 4 * lane);
  ln_rows_inreg<SWISH, RN>(x, g, b, eps);
#pragma unroll
  for (int i = 0; i < RN; ++i) {
    const int row = wave + i * WV;
    if (zero_row(row)) x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(dst + row * kLda + 4 * lane) = x[i];
  }
}

// frame (row m of the flattened [B][Tp] rows) is PAD iff mul * t >= lens[b]; one length load per row the lane's quads
// touch, to be issued early
template <int R>
struct PadLaneT {
  bool pad_[RBT<R>::NR];
  __device__ __forceinline__ PadLaneT(const int64_t* __restrict__ lens, int r0, int M, int Tp, int mul) {
    const int lane = lane_id();
#pragma unroll
    for (int i = 0; i < RBT<R>::NR; ++i) {
      pad_[i] = false;
      if (lens) {
        const int m = r0 + RBT<R>::row(i, lane);
        const int nb = max(M / Tp, 1);
        const int b = min(m / Tp, nb - 1), t = m - b * Tp;
        pad_[i] = m < M && (int64_t)mul * t >= lens[b];
      }
    }
  }
  __device__ __forceinline__ bool pad(int q) const { return pad_[RBT<R>::NR == 1 ? 0 : q]; }
};

// Swish epilogue of the previous W1 tile inside the next unit's MFMA stream (phases.h SwishSide on the quad view): quad
// q is handled during k-groups q * STEP (first pair) and q * STEP + STEP / 2 (second pair + one 16-byte LDS store)
template <int R>
struct SwishSideT {
  const typename RBT<R>::Acc& acc;
  float* dst;  // hidden buffer + place of the lane's quad 0 (LaneT::off(0))
  const f32x4 (&bias)[RBT<R>::NQ];
  mutable f32x2 lo;
  __device__ __forceinline__ void operator()(int g) const {
    constexpr int STEP = 32 / RBT<R>::NQ;
    const int q = g / STEP;
    if (g % STEP == 0) {
      const f32x4 v = RBT<R>::quad(acc, q);
      lo = swish2(f32x2{v[0] + bias[q][0], v[1] + bias[q][1]});
    }
    if (g % STEP == STEP / 2) {
      const f32x4 v = RBT<R>::quad(acc, q);
      const f32x2 hi = swish2(f32x2{v[2] + bias[q][2], v[3] + bias[q][3]});
      *reinterpret_cast<f32x4*>(dst + q * RBT<R>::QLDS) = f32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  }
};

// PositionwiseFeedForward on LDS-resident rows (phases.h ffn_phase<true>): acc2 += swish(A W1 + b1) W2 with the hidden
// dimension in 256-wide chunks that never leave LDS (bufH: two R x kLda buffers); weight stream W1(0), W1(1), W2(0),
// W1(2), W2(1), ..., W2(n-1), then `after` (k-group 0 of the wave's weight TILE, as every segment passed to rbt_gemm).
// c0 / n_total (split route): this call covers the hidden chunks [c0, c0 + n_chunks) of a module with n_total chunks (its
// partial sum); n_total = 0: the whole module.
template <int R>
__device__ __forceinline__ void ffn_phase_t(const float* bufA, float* bufH, const f32x4* __restrict__ w1,
                                            const float* __restrict__ b1, const f32x4* __restrict__ w2, int n_chunks,
                                            const f32x4* __restrict__ after, typename RBT<R>::Ring& ring,
                                            typename RBT<R>::Acc& acc2, int c0 = 0, int n_total = 0) {
  using T = RBT<R>;
  const LaneT<R> L;
  const int ts2 = (n_total > 0 ? n_total : n_chunks) * 32 * 64;  // W2: K = hidden
  b1 += c0 * 256;
  auto w1seg = [&](int c) { return w1 + (size_t)((c0 + c) * 8 + L.tile()) * kTs256; };
  auto w2seg = [&](int c) { return w2 + (size_t)L.tile() * ts2 + (size_t)(c0 + c) * 32 * 64; };
  typename T::Acc cur, nx;
  T::zero(cur);
  rbt_gemm<kG256>(bufA, kLda, w1seg(0), n_chunks > 1 ? w1seg(1) : w2seg(0), ring, cur);
  const int hoff = L.off(0);
  for (int c = 0; c < n_chunks; ++c) {
    float* hb = bufH + (c & 1) * T::ROWS * kLda;
    f32x4 bias[T::NQ];
#pragma unroll
    for (int q = 0; q < T::NQ; ++q) bias[q] = *reinterpret_cast<const f32x4*>(b1 + c * 256 + L.col(q));
    if (c + 1 < n_chunks) {
      T::zero(nx);
      rbt_gemm<kG256>(bufA, kLda, w1seg(c + 1), w2seg(c), ring, nx, SwishSideT<R>{cur, hb + hoff, bias, f32x2{0.f, 0.f}});
    } else {
#pragma unroll
      for (int q = 0; q < T::NQ; ++q) {
        const f32x4 v = T::quad(cur, q);
        const f32x2 lo = swish2(f32x2{v[0] + bias[q][0], v[1] + bias[q][1]});
        const f32x2 hi = swish2(f32x2{v[2] + bias[q][2], v[3] + bias[q][3]});
        *reinterpret_cast<f32x4*>(hb + hoff + q * T::QLDS) = f32x4{lo[0], lo[1], hi[0], hi[1]};
      }
    }
    if (c < 8) PPASR_TS(16 + 2 * c);
    __syncthreads();
    if (c < 8) PPASR_TS(17 + 2 * c);
    const f32x4* nseg = (c + 2 < n_chunks) ? w1seg(c + 2) : (c + 1 < n_chunks ? w2seg(c + 1) : after);
    rbt_gemm<kG256>(hb, kLda, w2seg(c), nseg, ring, acc2);
    cur = nx;
  }
}

// bufX[row][col] += scale * (acc + bias[col]) on the quad view
template <int R>
__device__ __forceinline__ void residual_epilogue_q(float* bufX, const typename RBT<R>::Acc& acc,
                                                    const float* __restrict__ bias, float scale) {
  const LaneT<R> L;
#pragma unroll
  for (int q = 0; q < RBT<R>::NQ; ++q) {
    float* p = bufX + L.off(q);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + L.col(q));
    const f32x4 a = RBT<R>::quad(acc, q);
    f32x4 x = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = x[e] + scale * (a[e] + bv[e]);
    *reinterpret_cast<f32x4*>(p) = x;
  }
}

// global store of the previous unit's tile (acc + bias) inside the next unit's MFMA stream: one 16-byte store per quad
template <int R>
struct QuadStoreSide {
  const typename RBT<R>::Acc& acc;
  float* out;   // place of the lane's quad 0 in the global matrix
  int qstep;    // floats from one quad of the lane to the next there (RBT<R>::qstep(row stride))
  int n_ok;     // the lane's first n_ok quads lie in valid rows
  const f32x4 (&bias)[RBT<R>::NQ];
  __device__ __forceinline__ void operator()(int g) const {
    constexpr int STEP = 32 / RBT<R>::NQ;
    const int q = g / STEP;
    if (g % STEP == STEP / 2 && q < n_ok) {
      const f32x4 v = RBT<R>::quad(acc, q);
      *reinterpret_cast<f32x4*>(out + q * qstep) = f32x4{v[0] + bias[q][0], v[1] + bias[q][1], v[2] + bias[q][2], v[3] + bias[q][3]};
    }
  }
};

// qkv[r0 + row][0 .. 768) = bufX * [Wq | Wk | Wv] + b (row-major): three transposed units, the stores of unit c sliced
// into the MFMA stream of unit c + 1.  `ring` already streams wqkv tile LaneT::tile().
// vt != nullptr (16-wave form only): the values go to vt in the fragment order of the fused attention kernel instead
// (conformer_kernels.h VtOut: [32-column slab][row octet][lane = column + 32 (row quad)][4 rows]) -- the V unit then runs
// in the plain orientation, where a lane's register quad IS four consecutive rows of one column = one 16-byte piece.
template <int R>
__device__ __forceinline__ void qkv_phase_t(const float* bufX, float* __restrict__ qkv, const f32x4* __restrict__ wqkv,
                                            const float* __restrict__ bqkv, int r0, int valid, typename RBT<R>::Ring& ring,
                                            float* __restrict__ vt = nullptr, int vt_stride = 0) {
  using T = RBT<R>;
  const LaneT<R> L;
  float* q0 = qkv + (size_t)(r0 + L.row(0)) * 768 + L.col(0);
  const int qs = T::qstep(768), n_ok = L.quads_ok(valid);
  f32x4 qb[3][T::NQ];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int q = 0; q < T::NQ; ++q) qb[c][q] = *reinterpret_cast<const f32x4*>(bqkv + c * 256 + L.col(q));
  typename T::Acc tile[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    T::zero(tile[c]);
    const f32x4* seg = wqkv + (size_t)(c * 8 + L.tile()) * kTs256;
    const f32x4* nseg = c < 2 ? seg + 8 * kTs256 : nullptr;
    if (c == 0) {
      rbt_gemm<kG256>(bufX, kLda, seg, nseg, ring, tile[c]);
    } else {
      const QuadStoreSide<R> store{tile[c - 1], q0 + (c - 1) * 256, qs, n_ok, qb[c - 1]};
      if constexpr (R == kW16) {
        if (c == 2 && vt) rbt_gemm<kG256, QuadStoreSide<R>, false>(bufX, kLda, seg, nseg, ring, tile[c], store);
        else rbt_gemm<kG256>(bufX, kLda, seg, nseg, ring, tile[c], store);
      } else {
        rbt_gemm<kG256>(bufX, kLda, seg, nseg, ring, tile[c], store);
      }
    }
  }
  if constexpr (R == kW16) {
    if (vt) {
      const int c32 = 16 * (L.wave & 1) + (L.lane & 15), kq = L.lane >> 4;
      const float bv = bqkv[512 + 16 * L.wave + (L.lane & 15)];
      float* dst = vt + ((size_t)(L.wave >> 1) * (vt_stride >> 3) + (r0 >> 3) + (kq >> 1)) * 256 + 4 * (c32 + 32 * (kq & 1));
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 v = tile[2].s[q];
        *reinterpret_cast<f32x4*>(dst + q * 512) = f32x4{v[0] + bv, v[1] + bv, v[2] + bv, v[3] + bv};
      }
      return;
    }
  }
#pragma unroll
  for (int q = 0; q < T::NQ; ++q) {
    if (q < n_ok) *reinterpret_cast<f32x4*>(q0 + 512 + q * qs) = T::quad(tile[2], q) + qb[2][q];
  }
}

// buffer resource over exactly `bytes` bytes at p (wave-uniform): loads outside read as zeros
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p, size_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  const uint32_t n = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xffffffffull ? 0xffffffffull : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, (int)n, 0x00020000);
}

// One tap chunk [J0, J0 + JN) of the register depthwise conv below: window rows J0 .. J0 + JN + RW - 2 of the wave's window
// and the chunk's tap weights, then the multiply-adds in ascending tap order per output row.  `zero` is an opaque 0 that
// ties the loads to the caller's pass loop (see there).
template <int RW, int J0, int JN>
__device__ __forceinline__ void dw_chunk(__amdgpu_buffer_rsrc_t rs_g, __amdgpu_buffer_rsrc_t rs_w, f32x4 (&acc)[RW],
                                         const f32x4& gp, int mwin0, int tqp, int left, int Tp, int lane, int zero) {
  constexpr int NWC = JN + RW - 1;
  // window rows through BUFFER loads on a resource that spans exactly the M rows of g: rows outside [0, M) (a negative
  // row is a huge unsigned offset) read as zeros without clamps, selects or 64-bit per-row addresses -- with plain global
  // loads the 33 clamped row addresses of a 16-tap chunk alone took 66 registers
  const int vrow = lane * 16 + (mwin0 + J0) * (kD * 4);
  f32x4 x[NWC];
#pragma unroll
  for (int qq = 0; qq < NWC; ++qq) x[qq] = wstream_load(rs_g, vrow + qq * (kD * 4), zero);
  f32x4 wt[JN];
#pragma unroll
  for (int j = 0; j < JN; ++j) wt[j] = wstream_load(rs_w, lane * 16, (J0 + j) * (kD * 4) + zero);
#pragma unroll
  for (int qq = 0; qq < NWC; ++qq) {
    const int tt = tqp - left + J0 + qq;  // frame the window row holds, relative to the utterance (wave-uniform)
    if (!(tt >= 0 && tt < Tp)) x[qq] = gp;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int j = qq - i;  // tap index inside the chunk
      if (j >= 0 && j < JN) acc[i] += wt[j] * x[qq];
    }
  }
}

// all tap chunks [J0, J0 + TC), [J0 + TC, ...) ... of a KS-tap window, in order.  The sched_barrier keeps the next chunk's
// loads behind this chunk's arithmetic: hoisted, the windows + tap sets of a 31-tap module are 280 registers (spills)
template <int RW, int KS, int TC, int J0>
__device__ __forceinline__ void dw_chunks(__amdgpu_buffer_rsrc_t rs_g, __amdgpu_buffer_rsrc_t rs_w, f32x4 (&acc)[RW],
                                          const f32x4& gp, int mwin0, int tqp, int left, int Tp, int lane, int zero) {
  constexpr int JN = KS - J0 < TC ? KS - J0 : TC;
  dw_chunk<RW, J0, JN>(rs_g, rs_w, acc, gp, mwin0, tqp, left, Tp, lane, zero);
  if constexpr (J0 + JN < KS) {
    // (pin the chunk's multiply-adds here: the machine-sink pass otherwise moves them down to the accumulators' final
    //  use, below the next chunk's loads, and both chunks' windows + taps are live at once)
#pragma unroll
    for (int i = 0; i < RW; ++i) asm volatile("" : "+v"(acc[i]));
    __builtin_amdgcn_sched_barrier(0);
    dw_chunks<RW, KS, TC, J0 + JN>(rs_g, rs_w, acc, gp, mwin0, tqp, left, Tp, lane, zero);
  }
}

// Depthwise conv (KS taps) + conv-module LayerNorm + swish of the block's rows, in registers (phases.h dwconv_ln_phase
// for any form and KS): wave w owns the RW = ROWS / WAVES consecutive rows RW w .. RW w + RW - 1; the taps are walked in
// chunks of <= RBT<R>::DW_TC (16; 8 in the 16-wave form) so that window rows + tap weights stay inside the register budget
// (every output row still accumulates its taps in ascending order: bit-identical to the one-pass form).  A wave whose
// rows straddle an utterance boundary runs the walk twice, once per utterance, and keeps per row the result of the
// row's own utterance (needs Tp >= RW).
// `Between` runs after the multiply-adds and before the LayerNorm -- the place to start the weight stream of the GEMM that
// follows: primed before this phase, the ring's registers are live across the window + tap sets and the 31-tap kernels
// spill; primed here, its latency hides behind the LayerNorm.
template <int R, int KS, typename Between>
__device__ __forceinline__ void dwconv_ln_phase_t(const float* __restrict__ g, float* bufA, const float* __restrict__ dw_w,
                                                  const float* __restrict__ dw_b, const float* __restrict__ glu_pad,
                                                  const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
                                                  int r0, int M, int Tp, int left, Between between) {
  const int lane = lane_id(), wave = wave_id();
  constexpr int LO = KS - 1, RW = RBT<R>::RW, TC = KS <= RBT<R>::DW_TC ? 16 : RBT<R>::DW_TC;
  const bool causal = (left == LO);
  const int q0 = wave * RW;
  f32x4 gp = *reinterpret_cast<const f32x4*>(glu_pad + 4 * lane);
  if (!causal) gp = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 bias = *reinterpret_cast<const f32x4*>(dw_b + 4 * lane);
  const int m0 = r0 + q0;
  // frame of the wave's first row inside its utterance.  Wave-uniform; the integer division runs on the vector ALU, so say
  // so explicitly -- otherwise each of the 2 x 34 "is this window row inside the utterance" tests below becomes a 64-bit
  // lane mask and the kernel spills SGPRs
  const int tq0 = __builtin_amdgcn_readfirstlane(m0 - (m0 / Tp) * Tp);
  const int npass = (tq0 + RW - 1 < Tp) ? 1 : 2;  // 2: the rows straddle an utterance boundary
  const __amdgpu_buffer_rsrc_t rs_g = buf_rsrc(g, (size_t)M * kD * sizeof(float)), rs_w = wstream_rsrc(dw_w);
  f32x4 out[RW];
#pragma unroll 1
  for (int p = 0; p < npass; ++p) {
    // an opaque zero added to every load offset of this pass: the (rare) second pass re-requests the window and the taps
    // (cache hits) instead of the compiler keeping all of them live around the loop -- 31 tap rows + 34 window rows are
    // more registers than a wave has
    int zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    const int tqp = tq0 - p * Tp;  // frame of row 0 relative to the start of utterance p of this wave
    f32x4 acc[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) acc[i] = bias;
    dw_chunks<RW, KS, TC, 0>(rs_g, rs_w, acc, gp, m0 - left, tqp, left, Tp, lane, zero);
#pragma unroll
    for (int i = 0; i < RW; ++i)
      if (npass == 1 || (tqp + i >= 0 && tqp + i < Tp)) out[i] = acc[i];
  }
  __builtin_amdgcn_sched_barrier(0);
  between();
  const f32x4 gam = *reinterpret_cast<const f32x4*>(ln_g + 4 * lane);
  const f32x4 bet = *reinterpret_cast<const f32x4*>(ln_b + 4 * lane);
  ln_rows_inreg<true, RW>(out, gam, bet, ln_eps);
#pragma unroll
  for (int i = 0; i < RW; ++i) *reinterpret_cast<f32x4*>(bufA + (q0 + i) * kLda + 4 * lane) = out[i];
}

}  // namespace ppasr
