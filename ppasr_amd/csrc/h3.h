// fp16 x3 matrix-core route of the row-block GEMM units (ppasr_set_gemm_mode(h, PPASR_GEMM_F16X3); opt-in, round 4).
//
// gfx950 has no xf32: an fp32 GEMM on the matrix cores is v_mfma_f32_32x32x2_f32 at 1/16 of the 16-bit rate.  Here every
// operand is written as the sum of TWO fp16 numbers, hi = fp16(x), lo = fp16(x - hi), both rounded to nearest: 22
// significant bits, and a product of two pieces (11 x 11 bits) is exact in fp32.  Per 16-wide k step
//     acc += w_lo a_hi + w_hi a_lo + w_hi a_hi        (three v_mfma_f32_32x32x16_f16, fp32 accumulation)
// replaces eight fp32 MFMAs; w_lo a_lo (2^-22 of the product) is dropped.  Powers of two keep the low pieces normal fp16
// numbers (fp16's smallest normal is 6e-5): weights are scaled by 2^8 when they are re-packed, activations by 2^4 when a
// phase writes them to LDS, and the accumulator is scaled back by 2^-12 -- all exact.  Measured (tools/experiments/r05,
// profiles/r04f_split_bf16_microbench.txt): one 256-deep unit 2.6e-7 of float64 (an fp32 fmaf chain: 5.7e-7), the whole
// feed-forward module 2.7e-7 (fp32 arithmetic 5.5e-7), the 12-block Conformer's logits as close to float64 as fp32
// arithmetic is; 3.0 - 3.2 us per unit against 7.1 (the weight stream -- the same 4 bytes per weight -- is the bound).
// Range guard.  An activation beyond 4 094 (65 504 / 2^4) does not fit the high piece (the GEMM inputs of conv2 and the
// input projection are ReLU outputs, W2's are swish values: unbounded, checkpoint- and feature-dependent).  h3_split4
// saturates such a value to +-65 504 (so nothing becomes Inf / NaN on the way) and counts the event in g_h3_ovf, a
// monotonic device counter of this translation unit.  ppasr_encode snapshots the counters around its launches
// (capi.hip: encode_guarded): with the guard on (default) a changed counter makes it run the call again on the fp32
// kernels -- the caller gets the fp32 result and ppasr_gemm_guard_stats counts the fallback; with the guard off the
// saturated result stands and the caller polls the same statistics.  Weights: |w| >= 255.9 (65 504 / 2^8) is refused by
// ppasr_set_gemm_mode (k_repack_h3 counts them in the same counter).
//
// Layouts.  Weights: the fp32 fragment stream's geometry -- per 32-column tile and 16-wide k step two 1 KiB blocks
// (high pieces, low pieces), lane l = 8 consecutive k (16 ks + 8 (l >> 5) ..) of column 32 tile + (l & 31) -- i.e. the
// same bytes per 256-deep segment (32 KiB per tile) and the same block count as 32 fp32 k-groups, so the ring, the
// segment pointers and the hand-over between fp32 and fp16 units are the fp32 route's.  Activations: two planes
// [32][kLdh] fp16 (row stride 528 B: 16-byte fragment reads are bank-conflict-free), high pieces then low pieces.
#pragma once
#include "rowblock.h"

namespace ppasr {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kLdh = kD + 8;                 // fp16 elements per plane row
constexpr int kPlaneH = kRows * kLdh;        // ... per plane
constexpr int kH3TileBytes = 2 * kPlaneH * 2;  // both planes of a [32][256] operand: 33 792 B (an fp32 tile: 33 280)
// the feed-forward phase keeps three operand tiles (LayerNorm output, two hidden chunks) where the fp32 route keeps three
// fp32 tiles (bufA, bufH[2]): 1 536 B more
constexpr int kH3ExtraLds = 3 * kH3TileBytes - 3 * kRows * kLda * (int)sizeof(float);
constexpr float kH3Sa = 16.f, kH3Sw = 256.f, kH3Inv = 1.f / (16.f * 256.f);

constexpr float kH3Max = 65504.f;  // largest finite fp16

// events of the range guard in this translation unit's kernels (see the header comment); never reset: callers compare
// snapshots.  (No -fgpu-rdc: every .hip that includes this header owns one; h3_ovf_counter() is that one's address.)
static __device__ unsigned int g_h3_ovf = 0;
static inline unsigned int* h3_ovf_counter() {
  void* p = nullptr;
  return hipGetSymbolAddress(&p, HIP_SYMBOL(g_h3_ovf)) == hipSuccess ? static_cast<unsigned int*>(p) : nullptr;
}

// v: already scaled by 2^4.  Out-of-range (and NaN) inputs: saturated; `bad` collects the event (branch-free: the callers
// sit inside MFMA streams, where a branch per quad would cut the schedule into pieces) and h3_note() counts it once per
// phase.
__device__ __forceinline__ void h3_split4(const f32x4 v, f16x4& hi, f16x4& lo, bool& bad) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float c = fminf(fmaxf(v[i], -kH3Max), kH3Max);
    bad |= c != v[i];
    hi[i] = (_Float16)c;
    lo[i] = (_Float16)(c - (float)hi[i]);
  }
}
__device__ __forceinline__ void h3_note(bool bad) {
  if (bad) atomicAdd(&g_h3_ovf, 1u);
}

// fp32 tile [32][kLda] (complete: the caller has synchronised) -> operand planes at dst, which may overlap src: every
// thread reads its 16 values, barrier, writes.  Ends with a barrier (the planes are complete on return).
__device__ __forceinline__ void h3_planes_from_tile(const float* src, _Float16* dst) {
  const int row = threadIdx.x >> 4, c0 = (threadIdx.x & 15) * 16;
  f32x4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + row * kLda + c0 + 4 * i);
  __syncthreads();
  bool bad = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f16x4 hi, lo;
    h3_split4(v[i] * kH3Sa, hi, lo, bad);
    *reinterpret_cast<f16x4*>(dst + row * kLdh + c0 + 4 * i) = hi;
    *reinterpret_cast<f16x4*>(dst + kPlaneH + row * kLdh + c0 + 4 * i) = lo;
  }
  h3_note(bad);
  __syncthreads();
}

// One 256-deep unit on transposed accumulators (rb_gemm's SWAP form: lane = row, register quad q = output features
// wave*32 + 8q + 4(lane>>5) .. +3): acc += 2^12 * A W for this wave's 32 output features.
//   pl  : operand planes of A
//   bp  : this unit's weight segment (tile of this wave), nxt: the segment the stream continues with (fp32 or fp16
//         packed: the ring holds 1 KiB blocks either way); the ring must hold bp's first kPF blocks on entry
template <typename Side = NoSide>
__device__ __forceinline__ void rb_gemm_h3(const _Float16* pl, const f32x4* __restrict__ bp, const f32x4* __restrict__ nxt,
                                           BRing<1>& ring, f32x16& acc, Side side = Side()) {
  static_assert(kPF == 4, "two k steps of two blocks each in flight");
  const int lane = lane_id();
  const _Float16* a_hi = pl + (lane & 31) * kLdh + 8 * (lane >> 5);
  const _Float16* a_lo = a_hi + kPlaneH;
  const __amdgpu_buffer_rsrc_t rs_b = wstream_rsrc(bp), rs_n = wstream_rsrc(nxt);
  const int voff = lane * 16;
  f16x8 a0 = *reinterpret_cast<const f16x8*>(a_hi), a1 = *reinterpret_cast<const f16x8*>(a_lo), n0 = a0, n1 = a1;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int s = (2 * ks) % kPF;
    if (ks + 1 < 16) {
      n0 = *reinterpret_cast<const f16x8*>(a_hi + 16 * (ks + 1));
      n1 = *reinterpret_cast<const f16x8*>(a_lo + 16 * (ks + 1));
    }
    const f16x8 w0 = __builtin_bit_cast(f16x8, ring.q[s][0]), w1 = __builtin_bit_cast(f16x8, ring.q[s + 1][0]);
    if (2 * ks + kPF < 32) {
      ring.q[s][0] = wstream_load(rs_b, voff, (2 * ks + kPF) * 1024);
      ring.q[s + 1][0] = wstream_load(rs_b, voff, (2 * ks + kPF + 1) * 1024);
    } else if (nxt) {
      ring.q[s][0] = wstream_load(rs_n, voff, (2 * ks + kPF - 32) * 1024);
      ring.q[s + 1][0] = wstream_load(rs_n, voff, (2 * ks + kPF + 1 - 32) * 1024);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a0, acc, 0, 0, 0);
    side(ks);
    a0 = n0;
    a1 = n1;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The streamed-A GEMM's unit (k_gemm_stream: conv2's implicit GEMM) on the fp16 x3 route: MT 32-row tiles share every weight
// fragment, accumulators in the standard layout (lane = column).  acc[mt] += 2^12 * A[32 mt ..][16 KS] W.
//   pl: operand planes [2][rows][ldh] of the K chunk (plane = fp16 elements per plane); KS 16-wide k steps
template <int MT, int KS, typename Side = NoSide>
__device__ __forceinline__ void rb_gemm_h3_rows(const _Float16* pl, int ldh, int plane, const f32x4* __restrict__ bp,
                                                const f32x4* __restrict__ nxt, BRing<1>& ring, f32x16 (&acc)[MT][1],
                                                Side side = Side()) {
  static_assert(kPF == 4, "two k steps of two blocks each in flight");
  const int lane = lane_id();
  const _Float16* a_hi = pl + (lane & 31) * ldh + 8 * (lane >> 5);
  const _Float16* a_lo = a_hi + plane;
  const __amdgpu_buffer_rsrc_t rs_b = wstream_rsrc(bp), rs_n = wstream_rsrc(nxt);
  const int voff = lane * 16;
  f16x8 a0[MT], a1[MT], n0[MT], n1[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    a0[mt] = *reinterpret_cast<const f16x8*>(a_hi + mt * 32 * ldh);
    a1[mt] = *reinterpret_cast<const f16x8*>(a_lo + mt * 32 * ldh);
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int s = (2 * ks) % kPF;
    if (ks + 1 < KS) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        n0[mt] = *reinterpret_cast<const f16x8*>(a_hi + mt * 32 * ldh + 16 * (ks + 1));
        n1[mt] = *reinterpret_cast<const f16x8*>(a_lo + mt * 32 * ldh + 16 * (ks + 1));
      }
    }
    const f16x8 w0 = __builtin_bit_cast(f16x8, ring.q[s][0]), w1 = __builtin_bit_cast(f16x8, ring.q[s + 1][0]);
    if (2 * ks + kPF < 2 * KS) {
      ring.q[s][0] = wstream_load(rs_b, voff, (2 * ks + kPF) * 1024);
      ring.q[s + 1][0] = wstream_load(rs_b, voff, (2 * ks + kPF + 1) * 1024);
    } else if (nxt) {
      ring.q[s][0] = wstream_load(rs_n, voff, (2 * ks + kPF - 2 * KS) * 1024);
      ring.q[s + 1][0] = wstream_load(rs_n, voff, (2 * ks + kPF + 1 - 2 * KS) * 1024);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[mt], w1, acc[mt][0], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[mt], w0, acc[mt][0], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[mt], w0, acc[mt][0], 0, 0, 0);
    side(ks);
    if (ks + 1 < KS) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        a0[mt] = n0[mt];
        a1[mt] = n1[mt];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Swish epilogue of hidden chunk c as a slice of the W1(c + 1) MFMA stream (what SwishSide of phases.h is to the fp32
// route): quad q of the wave's 32 hidden columns is handled during k steps 4q .. 4q + 3 -- two packed swish pairs, the
// split into the two fp16 pieces, the two 8-byte plane stores.  Exposed after the unit (round 4) the epilogue and the
// barrier behind it cost 1.5 us per unit, half as much again as the unit itself (tools/phase_ts.py --h3: 16 units 74 us).
struct SwishSideH3 {
  const f32x16& acc;       // raw accumulators of W1(c): 2^12 * (A W1)
  _Float16* dst;           // hb + (lane & 31) * kLdh + wave * 32 + 4 * (lane >> 5)
  const f32x4 (&bias)[4];  // b1 of this lane's columns, quad by quad
  bool& bad;               // range-guard events (h3_split4), counted by the caller once per phase
  mutable f32x2 s01, s23;
  mutable f16x4 hi, lo;
  __device__ __forceinline__ void operator()(int ks) const {
    const int q = ks >> 2;
    switch (ks & 3) {
      case 0: s01 = swish2(f32x2{acc[4 * q] * kH3Inv + bias[q][0], acc[4 * q + 1] * kH3Inv + bias[q][1]}); break;
      case 1: s23 = swish2(f32x2{acc[4 * q + 2] * kH3Inv + bias[q][2], acc[4 * q + 3] * kH3Inv + bias[q][3]}); break;
      case 2: h3_split4(f32x4{s01[0], s01[1], s23[0], s23[1]} * kH3Sa, hi, lo, bad); break;
      default:
        *reinterpret_cast<f16x4*>(dst + 8 * q) = hi;
        *reinterpret_cast<f16x4*>(dst + kPlaneH + 8 * q) = lo;
    }
  }
};

// PositionwiseFeedForward (positionwise.py:32-39) on the fp16 x3 route: acc2 = swish(A W1 + b1) W2 (transposed tile, as
// ffn_phase<true> leaves it: residual_epilogue_t applies).  The hidden dimension in 256-wide chunks that never leave LDS
// (operand planes, double-buffered); weight stream order as on the fp32 route: W1(0), W1(1), W2(0), W1(2), W2(1), ...,
// W2(n-1), `after`; the swish epilogue of chunk c runs inside the W1(c + 1) MFMA stream.
//   bufA : the LayerNorm'd rows, fp32 [32][kLda], complete.  The operand planes take over the LDS from bufA on:
//          [A | hidden chunk, even | hidden chunk, odd] = 3 * kH3TileBytes -- bufA, bufH[0], bufH[1] of the fp32 route
//          plus kH3ExtraLds bytes (the launch asks for them)
//   w1, w2: fp16-x3-packed (pack_h3 / k_repack_h3)
//   general form: src = the fp32 rows, pa = where their operand planes go (kH3TileBytes; may overlap src), ph = the two
//   hidden-chunk tiles (2 * kH3TileBytes)
//   c0 / n_total: the call covers hidden chunks [c0, c0 + n_chunks) of a layer with n_total chunks (k_ffn_part's slice of the
//   hidden dimension = a partial sum of the output); default = all of them
__device__ __forceinline__ void ffn_phase_h3(const float* src, _Float16* pa, _Float16* ph, const f32x4* __restrict__ w1,
                                             const float* __restrict__ b1, const f32x4* __restrict__ w2, int n_chunks,
                                             const f32x4* __restrict__ after, BRing<1>& ring, f32x16 (&acc2)[1][1],
                                             int c0 = 0, int n_total = -1) {
  const int lane = lane_id(), wave = wave_id();
  h3_planes_from_tile(src, pa);
  const int ts2 = (n_total > 0 ? n_total : n_chunks) * 32 * 64;  // W2: K = hidden
  b1 += c0 * 256;
  auto w1seg = [&](int c) { return w1 + (size_t)((c0 + c) * 8 + wave) * kTs256; };
  auto w2seg = [&](int c) { return w2 + (size_t)wave * ts2 + (size_t)(c0 + c) * 32 * 64; };
  const int hoff = (lane & 31) * kLdh + wave * 32 + 4 * (lane >> 5);
  f32x16 acc = acc2[0][0], cur, nx;
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 16; ++r) cur[r] = 0.f;
  rb_gemm_h3(pa, w1seg(0), n_chunks > 1 ? w1seg(1) : w2seg(0), ring, cur);
  for (int c = 0; c < n_chunks; ++c) {
    _Float16* hb = ph + (c & 1) * 2 * kPlaneH;
    f32x4 bias[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bias[q] = *reinterpret_cast<const f32x4*>(b1 + c * 256 + wave * 32 + 8 * q + 4 * (lane >> 5));
    const SwishSideH3 epi{cur, hb + hoff, bias, bad, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f16x4{0, 0, 0, 0}, f16x4{0, 0, 0, 0}};
    if (c + 1 < n_chunks) {
#pragma unroll
      for (int r = 0; r < 16; ++r) nx[r] = 0.f;
      rb_gemm_h3(pa, w1seg(c + 1), w2seg(c), ring, nx, epi);
    } else {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) epi(ks);
    }
    // (two hidden buffers: the buffer written here was last read by W2(c - 2), which every wave has left before it
    //  passed the barrier of chunk c - 1)
    if (c < 8) PPASR_TS(16 + 2 * c);
    __syncthreads();
    if (c < 8) PPASR_TS(17 + 2 * c);
    const f32x4* nseg = (c + 2 < n_chunks) ? w1seg(c + 2) : (c + 1 < n_chunks ? w2seg(c + 1) : after);
    rb_gemm_h3(hb, w2seg(c), nseg, ring, acc);
    cur = nx;
  }
  h3_note(bad);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[0][0][r] = acc[r] * kH3Inv;
}

// (the Conformer kernels' layout: the planes take over bufA, bufH[0], bufH[1])
__device__ __forceinline__ void ffn_phase_h3(float* bufA, const f32x4* __restrict__ w1, const float* __restrict__ b1,
                                             const f32x4* __restrict__ w2, int n_chunks, const f32x4* __restrict__ after,
                                             BRing<1>& ring, f32x16 (&acc2)[1][1], int c0 = 0, int n_total = -1) {
  _Float16* pa = reinterpret_cast<_Float16*>(bufA);
  ffn_phase_h3(bufA, pa, pa + 2 * kPlaneH, w1, b1, w2, n_chunks, after, ring, acc2, c0, n_total);
}

// One 256-deep unit of a complete fp32 tile with the accumulator in the STANDARD layout (lane = column: what the split
// route's kernels store from), on the fp16 x3 route: the tile becomes operand planes in place (they run 512 B past it)
__device__ __forceinline__ void unit_std_h3(float* tile, const f32x4* __restrict__ seg, const f32x4* __restrict__ nxt,
                                            BRing<1>& ring, f32x16 (&acc)[1][1]) {
  h3_planes_from_tile(tile, reinterpret_cast<_Float16*>(tile));
  rb_gemm_h3_rows<1, 16>(reinterpret_cast<const _Float16*>(tile), kLdh, kPlaneH, seg, nxt, ring, acc);
  acc[0][0] *= kH3Inv;
}

}  // namespace ppasr
