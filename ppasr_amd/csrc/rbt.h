// rbt.h -- the row-block GEMM core on TRANSPOSED accumulators for blocks of R = 32 or R = 16 rows.
//
// rowblock.h's execution model (a 512-thread workgroup owns R consecutive rows of the [B*T', 256] activation matrix,
// keeps them in LDS across a chain of dense layers, wave w owns output columns [32w, 32w + 32), weights streamed from
// L2 in MFMA fragment order through a register ring) with the rows per block as a template parameter:
//
//   R = 32: v_mfma_f32_32x32x2_f32, exactly rb_gemm<.., SWAP = true> of rowblock.h.
//   R = 16: v_mfma_f32_16x16x4_f32 (same FLOP rate, 32 cycles per instruction) on the SAME packed weights.  A launch
//           with fewer 32-row blocks than the chip has CUs takes as long as a full one; with 16-row blocks twice as
//           many CUs work and each block is half as long (half-rate layers of ragged batches, small batches,
//           streaming chunks of 16 frames).
//
// 16-row fragment maps.  The packed weights are [32-column tile][k-group of 8][lane L = c32 + 32 hh][4: k = 8g + 4hh + j]
// (capi.hip pack_b).  The wave's 32 columns are two 16-column sub-tiles s; lane l = (c16 = l & 15, kq = l >> 4) loads,
// for the K = 16 super-group G (k-groups 2G, 2G + 1), the f32x4 of packed lane 16s + c16 + 32 (kq & 1) of k-group
// 2G + (kq >> 1): the weights W[k = 16G + 4kq + m][column 16s + c16], m = 0..3 -- one buffer_load_dwordx4 per sub-tile
// per super-group, whole 128-byte lines, every byte used.  MFMA m of the super-group contracts the four k's
// {16G + 4kq + m : kq = 0..3}; the activation operand of the same lane is X[row l & 15][16G + 4kq + m] = element m of ONE
// ds_read_b128 at X[row][16G + 4kq ..] shared by both sub-tiles.  Issued with the weights as the A operand (the
// "swapped" form of rowblock.h), the accumulator of sub-tile s holds, for the lane's row l & 15, the four consecutive
// columns 16s + 4kq .. +3: a 16-byte quad, as in the 32-row form.
//
// Both forms therefore present a tile as NQ column quads per lane:
//   R = 32: NQ = 4, row = lane & 31, quad q = columns 32w + 8q + 4 (lane >> 5) .. +3
//   R = 16: NQ = 2, row = lane & 15, quad q = columns 32w + 16q + 4 (lane >> 4) .. +3
// and every epilogue in phases_t.h is written against that view.
//
// Third form, kW16 = 32 rows on SIXTEEN waves (1024 threads).  A 512-thread workgroup is two waves per SIMD, and the
// 133 KB of LDS a layer kernel holds keep a second workgroup off the CU: whenever both waves of a SIMD wait (weight
// fragment not there yet, LDS operand, barrier) its matrix pipe idles -- tools/microbench_rb16.hip measures 84 % of the
// fp32-MFMA rate for the bare 32-row GEMM stream at one workgroup per CU against 97 % with two per CU.  Four waves per
// SIMD inside ONE workgroup get the same interleaving without a second set of LDS buffers: wave w owns the 16-column
// sub-tile (w & 1) of column tile (w >> 1) for BOTH 16-row halves of the block -- per K = 16 super-group one
// buffer_load_dwordx4 (the 16-row form's fragment of that sub-tile), two ds_read_b128 (rows l & 15 and 16 + (l & 15)) and
// eight v_mfma_f32_16x16x4_f32; the weights cross L1 once per workgroup as in the 8-wave form (94 - 96 % in the same
// microbenchmark).  Presented as NQ = 2 quads per lane, quad q = row 16q + (lane & 15), columns 16w + 4 (lane >> 4) .. +3:
// here the quad index walks ROWS, in the 8-wave forms it walks columns -- RBT<>::row(q, lane) / col(q, lane, wave) hide
// the difference.  128 registers per lane.
#pragma once
#include "rowblock.h"

namespace ppasr {

template <int R>
struct RBT;

#ifndef PPASR_DW_TC32
#define PPASR_DW_TC32 16  // 8-wave 32-row form: taps per chunk of the register depthwise conv (tuning knob)
#endif
#ifndef PPASR_W16_PF
#define PPASR_W16_PF 4  // weight-stream prefetch depth of the 16-wave form, in K = 16 super-groups (256 B per lane-row each)
#endif

template <>
struct RBT<32> {
  static constexpr int ROWS = 32, WAVES = 8, THREADS = 512;
  static constexpr int NQ = 4;
  static constexpr int RW = 4;  // rows per wave in the row-wise phases (LayerNorm, depthwise conv)
  static constexpr int NR = 1;  // distinct rows among a lane's quads
  static constexpr int QLDS = 8;  // floats between a lane's consecutive quads in an LDS row buffer
  static constexpr int DW_TC = PPASR_DW_TC32;  // taps per chunk of the register depthwise conv (modules with more taps)
  static __device__ __forceinline__ int row(int, int lane) { return lane & 31; }
  static __device__ __forceinline__ int col(int q, int lane, int wave) { return wave * 32 + 8 * q + 4 * (lane >> 5); }
  static __device__ __forceinline__ int tile(int wave) { return wave; }  // 32-column weight tile the wave streams
  static __device__ __forceinline__ int qstep(int) { return QLDS; }      // ... and in a row-major global matrix
  struct Acc {
    f32x16 v[1][1];
  };
  static __device__ __forceinline__ f32x4 quad(const Acc& a, int q) {
    return f32x4{a.v[0][0][4 * q], a.v[0][0][4 * q + 1], a.v[0][0][4 * q + 2], a.v[0][0][4 * q + 3]};
  }
  static __device__ __forceinline__ void zero(Acc& a) { acc_zero(a.v); }
  using Ring = BRing<1, kPF>;
};

template <>
struct RBT<16> {
  static constexpr int ROWS = 16, WAVES = 8, THREADS = 512;
  static constexpr int NQ = 2;
  static constexpr int RW = 2;
  static constexpr int NR = 1;
  static constexpr int QLDS = 16;
  static constexpr int DW_TC = 16;
  static __device__ __forceinline__ int row(int, int lane) { return lane & 15; }
  static __device__ __forceinline__ int col(int q, int lane, int wave) { return wave * 32 + 16 * q + 4 * (lane >> 4); }
  static __device__ __forceinline__ int tile(int wave) { return wave; }
  static __device__ __forceinline__ int qstep(int) { return QLDS; }
  struct Acc {
    f32x4 s[2];
  };
  static __device__ __forceinline__ f32x4 quad(const Acc& a, int q) { return a.s[q]; }
  static __device__ __forceinline__ void zero(Acc& a) {
    a.s[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    a.s[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // PF super-groups (K = 16 each) ahead, two sub-tile fragments per super-group: the same bytes and the same lead
  // time in matrix-pipe cycles as the 32-row ring (1 KiB per 128 cycles instead of per 256)
  struct Ring {
    f32x4 q[kPF][2];
  };
};

template <>
struct RBT<kW16> {
  static constexpr int ROWS = 32, WAVES = 16, THREADS = 1024;
  static constexpr int NQ = 2;
  static constexpr int RW = 2;
  static constexpr int NR = 2;
  static constexpr int QLDS = 16 * kLda;
  static constexpr int DW_TC = 8;  // (128 registers per lane: window rows + taps of an 8-tap chunk are 68)
  struct Acc {
    f32x4 s[2];  // s[q]: row 16q + (lane & 15), columns 16 wave + 4 (lane >> 4) .. +3
  };
  static __device__ __forceinline__ int row(int q, int lane) { return 16 * q + (lane & 15); }
  static __device__ __forceinline__ int col(int, int lane, int wave) { return wave * 16 + 4 * (lane >> 4); }
  static __device__ __forceinline__ int tile(int wave) { return wave >> 1; }
  static __device__ __forceinline__ int qstep(int row_stride) { return 16 * row_stride; }
  static __device__ __forceinline__ f32x4 quad(const Acc& a, int q) { return a.s[q]; }
  static __device__ __forceinline__ void zero(Acc& a) {
    a.s[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    a.s[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  struct Ring {
    f32x4 q[PPASR_W16_PF];
  };
};

// byte offset (inside a packed [k-group][64 lanes] f32x4 tile) of the fragment this lane loads for sub-tile s of a
// 16-row unit: packed lane 16s + c16 + 32 (kq & 1) of the super-group's k-group (kq >> 1)
__device__ __forceinline__ int rbt16_voff(int lane, int s) {
  const int c16 = lane & 15, kq = lane >> 4;
  return ((kq >> 1) * 64 + 16 * s + c16 + 32 * (kq & 1)) * 16;
}

__device__ __forceinline__ void rbt_prime(RBT<32>::Ring& ring, const f32x4* __restrict__ bp) { ring_prime(ring, bp, 0); }
__device__ __forceinline__ void rbt_prime(RBT<16>::Ring& ring, const f32x4* __restrict__ bp) {
  const __amdgpu_buffer_rsrc_t rs = wstream_rsrc(bp);
  const int v0 = rbt16_voff(lane_id(), 0), v1 = rbt16_voff(lane_id(), 1);
#pragma unroll
  for (int g = 0; g < kPF; ++g) {
    ring.q[g][0] = wstream_load(rs, v0, g * 2048);
    ring.q[g][1] = wstream_load(rs, v1, g * 2048);
  }
}

__device__ __forceinline__ void rbt_prime(RBT<kW16>::Ring& ring, const f32x4* __restrict__ bp) {
  const __amdgpu_buffer_rsrc_t rs = wstream_rsrc(bp);
  const int v = rbt16_voff(lane_id(), wave_id() & 1);
#pragma unroll
  for (int g = 0; g < PPASR_W16_PF; ++g) ring.q[g] = wstream_load(rs, v, g * 2048);
}

// acc += X[R rows][K = 8 G] * Wpacked (the wave's 32 columns), transposed accumulators.
//   a_lds: LDS rows (row stride lda floats, lda % 64 == 4), bp: this call's packed segment (k-group 0 of the wave's tile),
//   nxt: the segment the weight stream continues with (nullptr: it ends); side(g) once per 8-wide k-group index g in
//   [0, G) -- in the 16-row form two consecutive indices (2Gs, 2Gs + 1) are delivered after super-group Gs.
template <int G, typename Side = NoSide>
__device__ __forceinline__ void rbt_gemm(const float* a_lds, int lda, const f32x4* __restrict__ bp,
                                         const f32x4* __restrict__ nxt, RBT<32>::Ring& ring, RBT<32>::Acc& acc,
                                         Side side = Side()) {
  rb_gemm<1, 1, G, kPF, Side, true>(a_lds, lda, bp, 0, nxt, 0, ring, acc.v, side);
}

template <int G, typename Side = NoSide>
__device__ __forceinline__ void rbt_gemm(const float* a_lds, int lda, const f32x4* __restrict__ bp,
                                         const f32x4* __restrict__ nxt, RBT<16>::Ring& ring, RBT<16>::Acc& acc,
                                         Side side = Side()) {
  static_assert(G % (2 * kPF) == 0, "K must be a multiple of 16 * ring depth");
  constexpr int GS = G / 2;  // super-groups
  const int lane = lane_id();
  const float* a_ptr = a_lds + (lane & 15) * lda + 4 * (lane >> 4);
  const __amdgpu_buffer_rsrc_t rs_b = wstream_rsrc(bp), rs_n = wstream_rsrc(nxt);
  const int v0 = rbt16_voff(lane, 0), v1 = rbt16_voff(lane, 1);
  f32x4 a_cur = *reinterpret_cast<const f32x4*>(a_ptr), a_nxt = a_cur;
#pragma unroll
  for (int g = 0; g < GS; ++g) {
    const int sl = g % kPF;
    if (g + 1 < GS) a_nxt = *reinterpret_cast<const f32x4*>(a_ptr + 16 * (g + 1));
    const f32x4 b0 = ring.q[sl][0], b1 = ring.q[sl][1];
    if (g + kPF < GS) {
      ring.q[sl][0] = wstream_load(rs_b, v0, (g + kPF) * 2048);
      ring.q[sl][1] = wstream_load(rs_b, v1, (g + kPF) * 2048);
    } else if (nxt) {
      ring.q[sl][0] = wstream_load(rs_n, v0, (g + kPF - GS) * 2048);
      ring.q[sl][1] = wstream_load(rs_n, v1, (g + kPF - GS) * 2048);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc.s[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[m], a_cur[m], acc.s[0], 0, 0, 0);
      acc.s[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[m], a_cur[m], acc.s[1], 0, 0, 0);
    }
    side(2 * g);
    side(2 * g + 1);
    a_cur = a_nxt;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// 16-wave form: bp / nxt = k-group 0 of the wave's 32-column TILE (RBT<kW16>::tile(wave)); the wave takes its sub-tile.
// SWAP = false: plain orientation (lane = column 16 (wave & 1) + (lane & 15) of the tile, s[q] = rows 16q + 4 (lane >> 4)
// .. +3 of that column) -- what the fragment-ordered value store of the fused attention route needs.
template <int G, typename Side = NoSide, bool SWAP = true>
__device__ __forceinline__ void rbt_gemm(const float* a_lds, int lda, const f32x4* __restrict__ bp,
                                         const f32x4* __restrict__ nxt, RBT<kW16>::Ring& ring, RBT<kW16>::Acc& acc,
                                         Side side = Side()) {
  constexpr int PF = PPASR_W16_PF;
  static_assert(G % (2 * PF) == 0, "K must be a multiple of 16 * ring depth");
  constexpr int GS = G / 2;
  const int lane = lane_id();
  const float* a_ptr = a_lds + (lane & 15) * lda + 4 * (lane >> 4);
  const __amdgpu_buffer_rsrc_t rs_b = wstream_rsrc(bp), rs_n = wstream_rsrc(nxt);
  const int v = rbt16_voff(lane, wave_id() & 1);
  f32x4 a0 = *reinterpret_cast<const f32x4*>(a_ptr), a1 = *reinterpret_cast<const f32x4*>(a_ptr + 16 * lda);
  f32x4 n0 = a0, n1 = a1;
#pragma unroll
  for (int g = 0; g < GS; ++g) {
    const int sl = g % PF;
    if (g + 1 < GS) {
      n0 = *reinterpret_cast<const f32x4*>(a_ptr + 16 * (g + 1));
      n1 = *reinterpret_cast<const f32x4*>(a_ptr + 16 * lda + 16 * (g + 1));
    }
    const f32x4 b = ring.q[sl];
    if (g + PF < GS) ring.q[sl] = wstream_load(rs_b, v, (g + PF) * 2048);
    else if (nxt) ring.q[sl] = wstream_load(rs_n, v, (g + PF - GS) * 2048);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (SWAP) {
        acc.s[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[m], a0[m], acc.s[0], 0, 0, 0);
        acc.s[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[m], a1[m], acc.s[1], 0, 0, 0);
      } else {
        acc.s[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m], b[m], acc.s[0], 0, 0, 0);
        acc.s[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[m], b[m], acc.s[1], 0, 0, 0);
      }
    }
    side(2 * g);
    side(2 * g + 1);
    a0 = n0;
    a1 = n1;
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace ppasr
