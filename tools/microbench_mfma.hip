// What bounds the fp32 matrix pipe on this chip?  Variants of a register-only v_mfma_f32_32x32x2_f32 loop (no memory):
//   chains = independent accumulator tiles per wave (1: every MFMA depends on the previous one)
//   waves  = waves per SIMD (block = 256 * waves threads, one block per CU)
//   + variants with the LDS read / global weight stream of rb_gemm
// Each prints the achieved TFLOP/s (all 256 CUs) and the implied "cycles per MFMA per SIMD" at 2.4 GHz.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ppasr_amd/csrc tools/microbench_mfma.hip -o tools/mb_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void k_pure(float* out, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = __uint_as_float(0x3f000000u | ((threadIdx.x * 2654435761u) >> 9)) - 0.75f, b = __uint_as_float(0x3f000000u | (((blockIdx.x * 64 + threadIdx.x) * 40503u * 2654435761u) >> 9)) - 0.75f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// + one ds_read_b128 per 4 MFMAs (the A operand of rb_gemm), optional 1 KiB global load per 4 MFMAs (the B ring, PF deep)
// MODE 0: no global loads; 1: plain global_load_dwordx4; >= 2: buffer_load_dwordx4 with cache-policy bits aux = MODE
// (gfx940+: 1 = sc0, 2 = nt, 16 = sc1).  DISTINCT: every workgroup streams its own copy of the weights.
template <int PF, int MODE, bool DISTINCT>
__global__ void k_mem(const f32x4* __restrict__ w, float* out, int iters) {
  constexpr bool GLOBAL = MODE != 0;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < 32 * 260; i += blockDim.x) smem[i] = __uint_as_float(0x3d000000u | ((i * 2654435761u) >> 9)) - 0.04f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* a_ptr = smem + (lane & 31) * 260 + 4 * (lane >> 5);
  const size_t wg_off = DISTINCT ? (size_t)(blockIdx.x % 16) * 64 * 8 * 32 * 64 : 0;
  const size_t wave_stride = (MODE == 101) ? (size_t)8 * 32 * 64 : (size_t)32 * 64;  // 256 KB (W2-like) or 32 KB (W1-like) between waves
  const f32x4* bp = w + wg_off + (size_t)wave * wave_stride + lane;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(w + wg_off), 0, 0x7fffffff, 0x00020000);
  auto ld = [&](const f32x4* p) -> f32x4 {
    if constexpr (MODE == 1) return *p;
    else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((const char*)p - (const char*)(w + wg_off)), 0, (MODE == 100 || MODE == 101) ? 0 : MODE));
  };
  f32x4 ring[PF];
  for (int s = 0; s < PF; ++s) ring[s] = GLOBAL ? ld(bp + s * 64) : f32x4{1.f, 2.f, 3.f, 4.f};
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 a_cur = *reinterpret_cast<const f32x4*>(a_ptr);
  for (int it = 0; it < iters; ++it) {
    const f32x4* seg = bp + ((MODE == 101) ? (size_t)(it & 7) * 32 * 64 : (size_t)((it & 7) * 8) * 32 * 64);
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      const f32x4 a_nxt = *reinterpret_cast<const f32x4*>(a_ptr + 8 * ((g + 1) & 31));
      const f32x4 b = ring[g % PF];
      if (GLOBAL) ring[g % PF] = ld(seg + (size_t)((g + PF) & 31) * 64);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[j], b[j], acc, 0, 0, 0);
      a_cur = a_nxt;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Register-tile shapes of a wave: MT row tiles (A fragments from LDS, one ds_read_b128 each per k-group) x NT column
// tiles (B fragments from the buffer-load ring), MT*NT accumulator chains, 4*MT*NT MFMAs per k-group.
template <int MT, int NT>
__global__ void k_tile(const f32x4* __restrict__ w, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = 132;
  for (int i = threadIdx.x; i < MT * 32 * LD; i += blockDim.x) smem[i] = __uint_as_float(0x3d000000u | ((i * 2654435761u) >> 9)) - 0.04f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* a_ptr = smem + (lane & 31) * LD + 4 * (lane >> 5);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
  const int voff = lane * 16;
  constexpr int PF = 4;
  f32x4 ring[PF][NT];
  auto ld = [&](int g, int nt) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (((g & 1023) * 8 + wave) * NT + nt) * 1024, 0));
  };
  for (int s = 0; s < PF; ++s)
    for (int nt = 0; nt < NT; ++nt) ring[s][nt] = ld(s, nt);
  f32x16 acc[MT][NT];
  for (int mt = 0; mt < MT; ++mt)
    for (int nt = 0; nt < NT; ++nt)
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  f32x4 a_cur[MT], a_nxt[MT];
  for (int mt = 0; mt < MT; ++mt) a_cur[mt] = *reinterpret_cast<const f32x4*>(a_ptr + mt * 32 * LD);
  int pos = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = *reinterpret_cast<const f32x4*>(a_ptr + mt * 32 * LD + 8 * ((g + 1) & 15));
      f32x4 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        b[nt] = ring[g % PF][nt];
        ring[g % PF][nt] = ld(pos + g + PF, nt);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
      __builtin_amdgcn_sched_barrier(0);
    }
    pos += 16;
  }
  float sum = 0.f;
  for (int mt = 0; mt < MT; ++mt)
    for (int nt = 0; nt < NT; ++nt)
      for (int r = 0; r < 16; ++r) sum += acc[mt][nt][r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

// Split-precision candidate (NOTES section 8): fp32 operands as fp16 hi + fp16 (scaled) lo, three
// v_mfma_f32_32x32x16_f16 per 16-wide k step (hi*hi, hi*lo, lo*hi; fp32 accumulation) in the row-block GEMM's structure --
// A (hi, lo) fragments from LDS, B (hi, lo) fragments from the buffer-load ring, 32 x 32 tile per wave.  Reported as
// fp32-EQUIVALENT TFLOP/s (2 * 32 * 32 * 16 per step), i.e. directly comparable with the exact-fp32 MFMA figures.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
template <int TERMS>
__global__ void k_split16(const f32x4* __restrict__ w, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < 2 * 32 * 132; i += blockDim.x) smem[i] = __uint_as_float(0x3c003c00u | ((i * 2654435761u) & 0x03ff03ffu));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // A tile as fp16: [32 rows][256 k] hi, then lo; a lane reads row l&31, k = 16 step + 8 (l>>5) .. +7 = 16 bytes
  const float* a_hi = smem + (lane & 31) * 132 + 4 * (lane >> 5);
  const float* a_lo = a_hi + 32 * 132;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
  const int voff = lane * 16;
  constexpr int PF = 4;
  f32x4 ring[PF][2];
  auto ld = [&](int g, int part) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ((((g & 63) * 8 + wave) * 2 + part)) * 1024, 0));
  };
  for (int s = 0; s < PF; ++s) { ring[s][0] = ld(s, 0); ring[s][1] = ld(s, 1); }
  f32x16 acc, acc_lo;
  for (int r = 0; r < 16; ++r) acc[r] = acc_lo[r] = 0.f;
  int pos = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {  // 16 steps of 16 = K 256 (one GEMM unit)
      const f32x4 ah = *reinterpret_cast<const f32x4*>(a_hi + 8 * g);
      const f32x4 al = *reinterpret_cast<const f32x4*>(a_lo + 8 * g);
      const f32x4 bh = ring[g % PF][0], bl = ring[g % PF][1];
      ring[g % PF][0] = ld(pos + g + PF, 0);
      ring[g % PF][1] = ld(pos + g + PF, 1);
      const h16x8 Ah = __builtin_bit_cast(h16x8, ah), Al = __builtin_bit_cast(h16x8, al);
      const h16x8 Bh = __builtin_bit_cast(h16x8, bh), Bl = __builtin_bit_cast(h16x8, bl);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, acc, 0, 0, 0);
      if (TERMS >= 3) {
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, acc_lo, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, acc_lo, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    pos += 16;
  }
  float sum = 0.f;
  for (int r = 0; r < 16; ++r) sum += acc[r] + acc_lo[r] * (1.0f / 2048.0f);
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

// The FFN chunk loop of phases.h in miniature: unit 1 = A(buf0) x W1 tile -> acc1 (optionally with the swish side
// writes of the previous tile into buf1), [barrier], unit 2 = A(buf1) x W2 tile -> acc2.  Buffer-load weight ring.
//   BAR: workgroup barrier between the units (as ffn_phase has)    SIDE: swish side work    FUSE: both units as ONE
//   64-k-group stream with interleaved k-groups and two accumulator chains (no transition in the middle)
template <bool BAR, int SIDE, bool FUSE>
__global__ void k_ffnlike(const f32x4* __restrict__ w, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* buf0 = smem;
  float* buf1 = smem + 32 * 260;
  for (int i = threadIdx.x; i < 2 * 32 * 260; i += blockDim.x) smem[i] = __uint_as_float(0x3d000000u | ((i * 2654435761u) >> 9)) - 0.04f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* a0 = buf0 + (lane & 31) * 260 + 4 * (lane >> 5);
  const float* a1 = buf1 + (lane & 31) * 260 + 4 * (lane >> 5);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
  const int voff = lane * 16 + wave * 32 * 64 * 16;
  constexpr int PF = 4;
  f32x4 ring[PF];
  int pos = 0;  // k-group counter of the weight stream
  auto ld = [&](int g) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ((g & 2047) * 64) * 16 * 8 / 8, 0)); };
  for (int s = 0; s < PF; ++s) ring[s] = ld(s);
  f32x16 acc1, acc2, prev;
  float live = 0.f;
  for (int r = 0; r < 16; ++r) acc1[r] = acc2[r] = prev[r] = 0.f;
  const int col = wave * 32 + (lane & 31);
  for (int it = 0; it < iters; ++it) {
    if (!FUSE) {
      f32x4 a_cur = *reinterpret_cast<const f32x4*>(a0);
#pragma unroll
      for (int g = 0; g < 32; ++g) {
        const f32x4 a_nxt = *reinterpret_cast<const f32x4*>(a0 + 8 * ((g + 1) & 31));
        const f32x4 b = ring[g % PF];
        ring[g % PF] = ld(pos + g + PF);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[j], b[j], acc1, 0, 0, 0);
        if (SIDE >= 1 && SIDE <= 3 && (g & 1) == 0) {
          const int r = g >> 1;
          const float v = prev[r] + 0.1f;
          const float sw = SIDE == 2 ? v * 0.5f : v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
          if (SIDE != 3) buf1[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 260 + col] = sw;
          else prev[r] = sw;
        }
        if (SIDE == 4 && g == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = prev[r] + 0.1f;
            buf1[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 260 + col] = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
          }
        }
        if (SIDE == 10 && (g & 1) == 0) {  // VALU only, result kept live
          const float v = prev[g >> 1] + 0.1f;
          live += v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        }
        if (SIDE == 11 && (g & 1) == 0) {  // LDS write only (data = a register that needs no VALU work)
          const int r = g >> 1;
          buf1[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 260 + col] = prev[r];
        }
        if (SIDE == 12 && (g & 7) == 0) {  // 4 x ds_write_b128 per unit instead of 16 x b32, no VALU work
          const int q = g >> 3;
          *reinterpret_cast<f32x4*>(buf1 + (lane & 31) * 260 + wave * 32 + 8 * q + 4 * (lane >> 5)) =
              f32x4{prev[4 * q], prev[4 * q + 1], prev[4 * q + 2], prev[4 * q + 3]};
        }
        if (SIDE == 13 && (g & 1) == 0) {  // swish spread over the even k-groups, results kept in registers ...
          const float v = prev[g >> 1] + 0.1f;
          prev[g >> 1] = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        }
        if (SIDE == 13 && (g & 7) == 7) {  // ... and written 4 at a time (the layout a transposed accumulator gives)
          const int q = g >> 3;
          *reinterpret_cast<f32x4*>(buf1 + (lane & 31) * 260 + wave * 32 + 8 * q + 4 * (lane >> 5)) =
              f32x4{prev[4 * q], prev[4 * q + 1], prev[4 * q + 2], prev[4 * q + 3]};
        }
        if (SIDE == 14 && (g & 1) == 0) {  // swish with v_exp/v_rcp replaced by plain multiply-adds of the same count
          const int r = g >> 1;
          const float v = prev[r] + 0.1f;
          float e = v * -1.44f;
          e = e * e + 1.0f;
          e = e * v + 0.3f;
          buf1[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 260 + col] = v * e;
        }
        if (SIDE == 5 && (g & 1) == 0) {  // swish computed here, the 16 LDS writes issued together at g == 31
          const int r = g >> 1;
          const float v = prev[r] + 0.1f;
          prev[r] = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        }
        if (SIDE == 5 && g == 31) {
#pragma unroll
          for (int r = 0; r < 16; ++r) buf1[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 260 + col] = prev[r];
        }
        a_cur = a_nxt;
        __builtin_amdgcn_sched_barrier(0);
      }
      pos += 32;
      if (BAR) __syncthreads();
      a_cur = *reinterpret_cast<const f32x4*>(a1);
#pragma unroll
      for (int g = 0; g < 32; ++g) {
        const f32x4 a_nxt = *reinterpret_cast<const f32x4*>(a1 + 8 * ((g + 1) & 31));
        const f32x4 b = ring[g % PF];
        ring[g % PF] = ld(pos + g + PF);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[j], b[j], acc2, 0, 0, 0);
        a_cur = a_nxt;
        __builtin_amdgcn_sched_barrier(0);
      }
      pos += 32;
    } else {
      // one stream: k-group g of unit 2 (A = buf1, previous chunk's hidden tile) and of unit 1 (A = buf0) alternate
      if (BAR) __syncthreads();
      f32x4 c0 = *reinterpret_cast<const f32x4*>(a0), c1 = *reinterpret_cast<const f32x4*>(a1);
#pragma unroll
      for (int g = 0; g < 32; ++g) {
        const f32x4 n0 = *reinterpret_cast<const f32x4*>(a0 + 8 * ((g + 1) & 31));
        const f32x4 n1 = *reinterpret_cast<const f32x4*>(a1 + 8 * ((g + 1) & 31));
        const f32x4 b0 = ring[(2 * g) % PF], b1 = ring[(2 * g + 1) % PF];
        ring[(2 * g) % PF] = ld(pos + 2 * g + PF);
        ring[(2 * g + 1) % PF] = ld(pos + 2 * g + 1 + PF);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c0[j], b0[j], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[j], b1[j], acc2, 0, 0, 0);
        }
        if (SIDE == 1 && (g & 1) == 0) {
          const int r = g >> 1;
          const float v = prev[r] + 0.1f;
          buf1[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 260 + col] = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        }
        c0 = n0;
        c1 = n1;
        __builtin_amdgcn_sched_barrier(0);
      }
      pos += 64;
    }
    prev = acc1;
  }
  float sum = 0.f;
  for (int r = 0; r < 16; ++r) sum += acc1[r] + acc2[r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum + live;
}

// Co-issue test: waves 0-3 (one per SIMD) run VALU work, waves 4-7 (their SIMD partners) run a saturating MFMA stream
// (2 independent chains).  Each wave reports its own duration (100 MHz ticks) so that the two can be compared with the
// partner idle.  VALU_KIND 0: independent v_fma (8 accumulators)   1: one dependent chain   2: v_exp (transcendental)
template <bool DO_VALU, bool DO_MFMA, int VALU_KIND>
__global__ void k_coissue(float* out, long long* ticks, int n_valu, int n_mfma) {
  const int wave = threadIdx.x >> 6;
  const long long t0 = (long long)wall_clock64();
  float res = 0.f;
  if (wave < 4) {
    if (DO_VALU) {
      float x[8];
      for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
      const float a = 0.999f, b = 0.001f;
      for (int it = 0; it < n_valu; ++it) {
        if (VALU_KIND == 0) {
#pragma unroll
          for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], a, b);
        } else if (VALU_KIND == 1) {
#pragma unroll
          for (int r = 0; r < 64; ++r) x[0] = fmaf(x[0], a, b);
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
        }
      }
      for (int i = 0; i < 8; ++i) res += x[i];
    }
  } else if (DO_MFMA) {
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;
    const float av = threadIdx.x * 0.01f, bv = 0.5f;
    for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, c1, 0, 0, 0);
      }
    }
    for (int r = 0; r < 16; ++r) res += c0[r] + c1[r];
  }
  const long long t1 = (long long)wall_clock64();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = res;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
}

template <class F>
static void timeit(const char* name, double mfma_per_launch, F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 3; ++r) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  ms /= 3;
  const double tf = mfma_per_launch * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%-58s %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3; %.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, ms, tf,
         100.0 * tf / 157.3, 64.0 * 157.3 / tf);
}

int main() {
  const int CUS = 256;
  float* out;
  hipMalloc(&out, (size_t)CUS * 1024 * 4);
  f32x4* w;
  const size_t wbytes = (size_t)17 * 64 * 8 * 32 * 64 * 16 + (1 << 20);
  hipMalloc(&w, wbytes);
  hipMemset(w, 0, wbytes);
  const bool random_data = getenv("MB_RANDOM") != nullptr;   // zeros (default) or realistic operand bits
  if (random_data) {
    std::vector<float> hw(wbytes / 4);
    unsigned x = 12345u;
    for (auto& v : hw) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 9) - (1 << 22)) * (1.0f / (1 << 22)) * 0.1f; }
    hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice);
  }
  printf("operand data: %s\n", random_data ? "random" : "zeros / constants");
  const int iters = 4000;
  for (int waves = 1; waves <= 4; waves *= 2) {
    char nm[96];
    snprintf(nm, sizeof nm, "pure MFMA, 1 chain, %d wave(s)/SIMD", waves);
    timeit(nm, (double)CUS * 4 * waves * iters * 16, [&] { hipLaunchKernelGGL(k_pure<1>, dim3(CUS), dim3(256 * waves), 0, 0, out, iters); });
    snprintf(nm, sizeof nm, "pure MFMA, 2 chains, %d wave(s)/SIMD", waves);
    timeit(nm, (double)CUS * 4 * waves * iters * 32, [&] { hipLaunchKernelGGL(k_pure<2>, dim3(CUS), dim3(256 * waves), 0, 0, out, iters); });
  }
  const int it2 = 2000;
  const size_t lds = 32 * 260 * 4;
#define RUN(PF, MODE, DIST, WAVES, name)                                                                         \
  timeit(name, (double)CUS * 4 * WAVES * it2 * 128,                                                               \
         [&] { hipLaunchKernelGGL((k_mem<PF, MODE, DIST>), dim3(CUS), dim3(256 * WAVES), lds, 0, w, out, it2); })
  RUN(4, 0, false, 2, "LDS A operand, no global, 2 waves/SIMD");
  RUN(4, 1, false, 2, "LDS A + global B ring PF=4 (rb_gemm shape), shared weights");
  RUN(8, 1, false, 2, "  PF=8");
  RUN(4, 1, true, 2, "  PF=4, every workgroup its own weights (16 copies)");
  RUN(4, 100, false, 2, "  buffer_load (no cache-policy bits)");
  RUN(4, 2, false, 2, "  buffer_load nt");
  RUN(4, 3, false, 2, "  buffer_load sc0 nt");
  RUN(4, 1 + 0, false, 1, "  PF=4 plain, 1 wave/SIMD");
  RUN(4, 16, false, 2, "  buffer_load sc1");
  RUN(4, 17, false, 2, "  buffer_load sc0 sc1");
  RUN(4, 18, false, 2, "  buffer_load sc1 nt");
  RUN(4, 100, true, 2, "  buffer_load plain, every workgroup its own weights");
  RUN(4, 101, false, 2, "  buffer_load plain, waves 256 KB apart (W2-like tiles)");
  RUN(8, 100, false, 2, "  buffer_load plain PF=8");
  {
    long long* ticks;
    hipMalloc(&ticks, 64);
    const int nv = 2000, nm = 2000;  // 128k VALU ops per wave; 32k MFMAs per wave (= 2.05 M cycles if the pipe is full)
    auto report = [&](const char* name) {
      hipDeviceSynchronize();
      long long h[8];
      hipMemcpy(h, ticks, 64, hipMemcpyDeviceToHost);
      printf("%-44s VALU wave %8.1f us (%.1f cycles/op)   MFMA wave %8.1f us (%.1f cycles/MFMA)\n", name, h[0] / 100.0,
             h[0] / 100.0 * 2400.0 / (nv * 64.0), h[4] / 100.0, h[4] / 100.0 * 2400.0 / (nm * 16.0));
    };
#define COI(V, M, K, name)                                                                           \
  hipMemset(ticks, 0, 64);                                                                           \
  hipLaunchKernelGGL((k_coissue<V, M, K>), dim3(CUS), dim3(512), 0, 0, out, ticks, nv, nm);          \
  report(name);
    COI(true, false, 0, "independent v_fma alone");
    COI(false, true, 0, "MFMA 2 chains alone");
    COI(true, true, 0, "independent v_fma + partner MFMA stream");
    COI(true, false, 1, "dependent v_fma chain alone");
    COI(true, true, 1, "dependent v_fma chain + partner MFMA stream");
    COI(true, false, 2, "v_exp alone");
    COI(true, true, 2, "v_exp + partner MFMA stream");
  }
  {
    const int it4 = 1000;
#define RUNT(MT, NT, WAVES, name)                                                                                  \
  timeit(name, (double)CUS * 4 * WAVES * (it4 / (MT * NT)) * 16 * 4 * MT * NT,                                     \
         [&] { hipLaunchKernelGGL((k_tile<MT, NT>), dim3(CUS), dim3(256 * WAVES), MT * 32 * 132 * 4, 0, w, out, it4 / (MT * NT)); })
    RUNT(1, 1, 2, "wave tile 32x32 (1 chain), 2 waves/SIMD");
    RUNT(2, 1, 2, "wave tile 64x32 (2 chains), 2 waves/SIMD");
    RUNT(4, 1, 2, "wave tile 128x32 (4 chains, conv2 today), 2 waves/SIMD");
    RUNT(4, 1, 1, "wave tile 128x32 (4 chains), 1 wave/SIMD");
    RUNT(2, 2, 2, "wave tile 64x64 (4 chains), 2 waves/SIMD");
    RUNT(2, 2, 1, "wave tile 64x64 (4 chains), 1 wave/SIMD");
    RUNT(4, 2, 1, "wave tile 128x64 (8 chains), 1 wave/SIMD");
    RUNT(4, 2, 2, "wave tile 128x64 (8 chains), 2 waves/SIMD");
    RUNT(1, 2, 2, "wave tile 32x64 (2 chains), 2 waves/SIMD");
    RUNT(1, 2, 1, "wave tile 32x64 (2 chains), 1 wave/SIMD");
    RUNT(1, 1, 1, "wave tile 32x32 (1 chain), 1 wave/SIMD");
    RUNT(1, 4, 1, "wave tile 32x128 (4 chains), 1 wave/SIMD");
  }
  {
    const int it5 = 4000;
    // mfma_per_launch in units of exact-fp32 MFMAs (4096 FLOP): one K = 16 step of a 32 x 32 tile = 8 of them
#define RUNS(TERMS, WAVES, name)                                                                                  \
  timeit(name, (double)CUS * 4 * WAVES * it5 * 16 * 8,                                                            \
         [&] { hipLaunchKernelGGL((k_split16<TERMS>), dim3(CUS), dim3(256 * WAVES), 2 * 32 * 132 * 4, 0, w, out, it5); })
    RUNS(3, 2, "fp16 hi/lo split, 3 MFMA 32x32x16 per step, 2 waves/SIMD (fp32-equivalent rate)");
    RUNS(3, 1, "fp16 hi/lo split, 3 MFMA per step, 1 wave/SIMD");
    RUNS(1, 2, "plain fp16 (1 MFMA per step) with the same operand traffic, 2 waves/SIMD");
  }
  const size_t lds2 = 2 * 32 * 260 * 4;
  const int it3 = 1000;
#define RUNF(BAR, SIDE, FUSE, name)                                                                       \
  timeit(name, (double)CUS * 8 * it3 * 256,                                                               \
         [&] { hipLaunchKernelGGL((k_ffnlike<BAR, SIDE, FUSE>), dim3(CUS), dim3(512), lds2, 0, w, out, it3); })
  RUNF(false, 0, false, "FFN-like: two 32-k-group units per chunk, no barrier, no side");
  RUNF(true, 0, false, "  + barrier between the units");
  RUNF(true, 1, false, "  + barrier + swish side work (= ffn_phase)");
  RUNF(true, 10, false, "  side = swish VALU only (kept live), no LDS write");
  RUNF(true, 11, false, "  side = 16 ds_write_b32 only, no VALU");
  RUNF(true, 12, false, "  side = 4 ds_write_b128 only, no VALU");
  RUNF(true, 13, false, "  side = swish spread + 4 ds_write_b128");
  RUNF(true, 14, false, "  side = mul/add stand-in for exp/rcp + 16 ds_write_b32");
  return 0;
}
