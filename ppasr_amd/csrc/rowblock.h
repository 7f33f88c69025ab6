// rowblock.h -- device-side toolkit shared by the gfx950 kernels.
//
// Execution model ("row-block"): a 256-thread workgroup (4 wave64, one per SIMD) owns
// 32*MT consecutive rows of the flattened [B*T', d] activation matrix and keeps them in
// LDS across a whole chain of dense layers.  Dense contractions run on the exact-fp32
// matrix core (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain):
//   * the A operand (activations) is read from LDS with one ds_read_b128 per 4 MFMAs;
//     rows are padded by 4 floats so the 16-lane b128 groups hit 16 distinct 16-B slots;
//   * the B operand (weights) is streamed from global/L2 straight into VGPRs in a layout
//     pre-packed on the host in MFMA fragment order, so every load is a fully coalesced
//     1 KiB global_load_dwordx4 per wave and needs no LDS staging (each wave owns its own
//     64 output columns, so B is not shared between waves);
//   * a 4-deep register ring prefetches B four k-groups (2048 MFMA cycles) ahead.
//
// Fragment maps (MI355X guide §3): A lane l holds A[i=l&31][k=l>>5]; B lane l holds
// B[k=l>>5][j=l&31]; C/D lane l, reg r holds D[(r&3)+8*(r>>2)+4*(l>>5)][l&31].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ppasr {

constexpr int kD = 256;        // model width the kernels are specialised for
constexpr int kLda = kD + 4;   // LDS row stride (floats) of a [rows][256] activation buffer
constexpr int kRows = 32;      // rows per MFMA row tile
constexpr int kThreads = 256;  // 4 wave64

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// row of accumulator register r inside a 32x32 tile for this lane
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float swishf(float x) { return x * sigmoidf(x); }

template <int MT, int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[MT][NT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
}

// acc[mt][nt] += A[32*MT x 8*g_count] * Bpacked.
//   a_lds : LDS, row 0 / k 0 of the A block, row stride lda floats (lda % 64 == 4)
//   bp    : packed weights, positioned at (this wave's first n-tile, first k-group);
//           n-tile nt of this call lives at bp + nt*tile_stride (units: f32x4)
//   g_count: number of 8-wide k-groups; must be a multiple of PF
template <int MT, int NT, int PF = 4>
__device__ __forceinline__ void rb_gemm(const float* a_lds, int lda, const f32x4* __restrict__ bp,
                                        int tile_stride, int g_count, f32x16 (&acc)[MT][NT]) {
  const int lane = lane_id();
  const float* a_ptr = a_lds + (lane & 31) * lda + 4 * (lane >> 5);
  const f32x4* b_ptr = bp + lane;
  f32x4 bq[PF][NT];
#pragma unroll
  for (int s = 0; s < PF; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bq[s][nt] = b_ptr[(size_t)nt * tile_stride + s * 64];
  for (int g0 = 0; g0 < g_count; g0 += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int g = g0 + s;
      f32x4 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(a_ptr + mt * 32 * lda + 8 * g);
      f32x4 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = bq[s][nt];
      if (g + PF < g_count) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[s][nt] = b_ptr[(size_t)nt * tile_stride + (size_t)(g + PF) * 64];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
    }
  }
}

// LayerNorm over the 256 columns of LDS rows (biased variance, eps inside the sqrt),
// nn.LayerNorm semantics (utils/base.py:7-21).  Each wave normalises rows w, w+4, ...;
// one ds_read_b128 per lane covers a whole row.  src == dst is allowed.
// If zero_row(row) is true the output row is forced to 0 (conv-module pad masking).
template <typename ZeroRow>
__device__ __forceinline__ void rb_layernorm(const float* src, float* dst, int lda, int nrows,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float eps, ZeroRow zero_row) {
  const int lane = lane_id();
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * lane);
  const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * lane);
  for (int row = wave_id(); row < nrows; row += 4) {
    f32x4 x = *reinterpret_cast<const f32x4*>(src + row * lda + 4 * lane);
    float mean = wave_sum(x[0] + x[1] + x[2] + x[3]) * (1.0f / kD);
    f32x4 c = x - mean;
    float var = wave_sum(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]) * (1.0f / kD);
    float rstd = 1.0f / sqrtf(var + eps);
    f32x4 y = c * rstd * g + b;
    if (zero_row(row)) y = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(dst + row * lda + 4 * lane) = y;
  }
}

struct NoZero {
  __device__ __forceinline__ bool operator()(int) const { return false; }
};

// copy nrows x 256 floats global(row stride 256) -> LDS(row stride lda); rows >= valid are zeroed
__device__ __forceinline__ void rb_load_rows(float* dst, int lda, const float* __restrict__ src, int nrows, int valid) {
  const int lane = lane_id();
  for (int row = wave_id(); row < nrows; row += 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < valid) v = *reinterpret_cast<const f32x4*>(src + (size_t)row * kD + 4 * lane);
    *reinterpret_cast<f32x4*>(dst + row * lda + 4 * lane) = v;
  }
}

// LDS(row stride lda) -> global(row stride 256) for rows < valid
__device__ __forceinline__ void rb_store_rows(float* __restrict__ dst, const float* src, int lda, int nrows, int valid) {
  const int lane = lane_id();
  for (int row = wave_id(); row < nrows && row < valid; row += 4)
    *reinterpret_cast<f32x4*>(dst + (size_t)row * kD + 4 * lane) =
        *reinterpret_cast<const f32x4*>(src + row * lda + 4 * lane);
}

}  // namespace ppasr
