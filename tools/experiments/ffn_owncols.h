// tools/experiments/ffn_owncols.h -- MEASURED AND NOT ADOPTED (round 2): the barrier-free form of ffn_phase.
// Every wave contracts its OWN 32 hidden columns with the matching 32 rows of W2 (8 output tiles x 4 k-groups) so that
// the chunk loop needs no workgroup barrier; the 8 partial [32 x 256] outputs are reduce-scattered through LDS at the
// end (7 ring steps).  Correct (all GPU tests passed), but SLOWER: 312.3 us per launch of the dominant kernel against
// 297.0 us for the shared-hidden-chunk form in phases.h (per-workgroup wall-clock stamps, tools/phase_ts.py): the
// W1 units do run at 99 % of the MFMA rate without barriers (6.9 us), but the 8 short G = 4 GEMM calls per chunk of
// the W2 part start with an exposed LDS read each (7.7 us per unit) and the final reduction costs ~7 us per FFN -- more
// than the 16 chunk barriers it replaces (ablating them in the adopted form gains 7.9 us per launch, 2.7 %).
// Kept for the record; drop-in replacement of ffn_phase (needs bufH >= 2 * 8 * 1024 floats).
#pragma once
#include "../../ppasr_amd/csrc/rowblock.h"

namespace ppasr {

// Epilogue slice of the previous W1 tile into the wave's PRIVATE hidden tile (ffn_phase below)
constexpr int kHpLd = 36;                 // row stride of a private 32x32 hidden tile: b128 A reads conflict-free (36 % 32 == 4)
constexpr int kHpTile = kRows * kHpLd;    // floats per wave
struct SwishSidePriv {
  const f32x16& acc;
  float* hp;
  float bias;
  int lane;
  __device__ __forceinline__ void operator()(int g) const {
    if ((g & 1) == 0) {
      const int r = g >> 1;
      hp[acc_row(r, lane) * kHpLd + (lane & 31)] = swishf(acc[r] + bias);
    }
  }
};

// PositionwiseFeedForward (positionwise.py:32-39): acc2 += swish(A*W1 + b1) * W2, hidden dimension in 256-wide chunks.
//
// Barrier-free inner loop.  Wave w owns hidden columns [32w, 32w+32) of every chunk.  It computes that 32x32 tile
// H_w = swish(A W1[:, cols_w] + b1) (A = the block's 32 rows in LDS, 32 k-groups), keeps it in a PRIVATE LDS tile, and
// contracts it straight away with the 32 rows of W2 that belong to ITS hidden columns: Y_w[32 x 256] += H_w W2[rows_w, :]
// (8 output tiles x 4 k-groups).  No wave ever reads another wave's hidden tile, so the loop over the chunks has no
// workgroup barrier at all (the previous form -- every wave reading the whole 32x256 hidden chunk -- had one per chunk,
// and each cost the matrix pipe ~0.7 us of refill plus the arrival skew of 8 waves: FFN units ran at 7.76 us against
// 6.83 us of MFMA time).  The price is one cross-wave sum at the end: the 8 partial outputs Y_w are reduce-scattered
// over 7 ring steps through LDS (wave w ends up with output columns [32w, 32w+32), summed in a fixed order).
// Same weights in the same packed layout (W2 tile nt, k-groups of the wave's hidden rows), same FLOPs, same bytes.
// Weight stream order: W1(0), W1(1), W2(0; 8 tiles), W1(2), W2(1; ...), ..., W2(n-1), then `after`.
// Register tiles: Yrel[j] = partial of output tile (w + j) & 7 (wave-relative, so that every index is static).
// bufH: >= 2 * 8 * 1024 floats (the private tiles, then the two slot sets of the ring); bufA is only read.
// c0 / n_total: the call covers hidden chunks [c0, c0 + n_chunks) of a layer with n_total chunks (a slice of the hidden
// dimension = a partial sum of the output, k_ffn_part); default = all of them.
__device__ __forceinline__ void ffn_phase(const float* bufA, float* bufH, const f32x4* __restrict__ w1,
                                          const float* __restrict__ b1, const f32x4* __restrict__ w2, int n_chunks,
                                          const f32x4* __restrict__ after, BRing<1>& ring, f32x16 (&acc2)[1][1],
                                          int c0 = 0, int n_total = -1) {
  const int lane = lane_id(), wave = wave_id();
  const int ts2 = (n_total > 0 ? n_total : n_chunks) * 32 * 64;  // W2: K = hidden
  const int col = wave * 32 + (lane & 31);
  float* hp = bufH + wave * kHpTile;
  b1 += c0 * 256;
  auto w1seg = [&](int c) { return w1 + (size_t)((c0 + c) * 8 + wave) * kTs256; };
  // rows [32w, 32w+32) of hidden chunk c (k-groups (c0+c)*32 + 4w .. +4) x output tile (w + j) & 7
  auto w2seg = [&](int c, int j) { return w2 + (size_t)((wave + j) & 7) * ts2 + (size_t)((c0 + c) * 32 + 4 * wave) * 64; };
  f32x16 yrel[8][1][1];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc_zero(yrel[j]);
  f32x16 cur[1][1], nx[1][1];
  acc_zero(cur);
  rb_gemm<1, 1, kG256>(bufA, kLda, w1seg(0), 0, n_chunks > 1 ? w1seg(1) : w2seg(0, 0), 0, ring, cur);
  for (int c = 0; c < n_chunks; ++c) {
    const float bias = b1[c * 256 + col];
    if (c + 1 < n_chunks) {
      acc_zero(nx);
      rb_gemm<1, 1, kG256>(bufA, kLda, w1seg(c + 1), 0, w2seg(c, 0), 0, ring, nx, SwishSidePriv{cur[0][0], hp, bias, lane});
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) hp[acc_row(r, lane) * kHpLd + (lane & 31)] = swishf(cur[0][0][r] + bias);
    }
    if (c < 8) PPASR_TS(16 + 2 * c);
    // (the wave reads back only what it wrote itself: LDS serves a wave's requests in order, no barrier)
    const f32x4* after_chunk = (c + 2 < n_chunks) ? w1seg(c + 2) : (c + 1 < n_chunks ? w2seg(c + 1, 0) : after);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      rb_gemm<1, 1, 4>(hp, kHpLd, w2seg(c, j), 0, j < 7 ? w2seg(c, j + 1) : after_chunk, 0, ring, yrel[j]);
    if (c < 8) PPASR_TS(17 + 2 * c);
    cur[0][0] = nx[0][0];
  }
  // ---- reduce-scatter of the partial outputs: step s hands tile (w + s) & 7 to its owner ----
  __syncthreads();  // every wave is done with its private tile (the slots below reuse that memory)
#pragma unroll
  for (int s = 1; s < 8; ++s) {
    float* set = bufH + (s & 1) * 8 * 1024;
    float* mine = set + wave * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = yrel[s][0][0][r];
    __syncthreads();
    const float* from = set + ((wave - s) & 7) * 1024;  // wave w - s computed its tile (w - s + s) = w as its yrel[s]
#pragma unroll
    for (int r = 0; r < 16; ++r) yrel[0][0][0][r] += from[r * 64 + lane];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[0][0][r] += yrel[0][0][0][r];
}

}  // namespace ppasr
