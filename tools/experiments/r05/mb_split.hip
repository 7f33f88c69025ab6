// Microbenchmark + device-side error check of a SPLIT-bf16 row-block GEMM unit (round-5 groundwork, not product code).
//
// The fused layer kernels are bound by v_mfma_f32_32x32x2_f32, which runs at 1/16 of the bf16 matrix rate (gfx950 has
// no xf32).  Every fp32 number is EXACTLY the sum of three bf16 numbers obtained by truncation (8 + 8 + 8 mantissa
// bits), and a product of two bf16 numbers is exact in fp32, so
//     a * b = sum_{i,j} a_i * b_j,     x6 keeps the six terms with i + j <= 2 (what is dropped is <= 3 * 2^-24 |a||b|),
//                                       x3 keeps the three terms with i + j <= 1 (pieces a0, a1, b0, b1 only; ~2^-16),
// each term one v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6/16 (3/16) of the fp32-MFMA time.
// tools/experiments/r05/split_bf16_numerics.py: through the 12-block Conformer x6 is as accurate as fp32 arithmetic
// itself (5e-7 of the float64 logits against 9e-7), x3 gives 1.1e-5 and changes 1 greedy frame in 4000.
//
// One unit = what one workgroup of the row-block kernels does per GEMM segment: C[32 x 256] += A[32 x 256] * W[256 x 256],
// A from LDS (here: bf16 piece planes written once), W streamed from L2 by every workgroup (host-packed in fragment
// order, piece-interleaved per k step: 2 * NP bytes per weight), 8 waves x one 32 x 32 accumulator tile.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/r05/mb_split.hip -o tools/experiments/r05/mb_split
// run:   mb_split [blocks=256] [units=360] [n_seg=36]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kK = 256, kN = 256, kR = 32, kLda = kK + 8;  // bf16 elements per LDS row: 528 B, b128 reads conflict-free
constexpr int kKS = kK / 16;                                // k steps per unit

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));       \
      exit(1);                                                                                \
    }                                                                                         \
  } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 load16(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

// MODE: 1 = one piece (plain bf16), 3 = x3, 6 = x6, 0 = stream only (no MFMA: what the weight stream alone costs),
// 13 = fp16 x3: two fp16 pieces per operand, rounded to nearest (11 + 11 significant bits, a product of two pieces is still
// exact in fp32), weights pre-scaled by 2^8 on the host so that their low pieces stay normal fp16 numbers, 2^-8 on the result
template <int MODE>
struct Split {
  static constexpr int NP = MODE == 6 ? 3 : ((MODE == 3 || MODE == 13) ? 2 : (MODE == 0 ? 3 : 1));
};

// number of pieces of the weight stream is a separate knob for MODE 0 (NPW)
// NW = 16: two waves per column tile, each takes half of the k steps (the two partial tiles would be added through LDS at the
// end of a unit: not timed here); ACC2: the terms of a k step alternate between two accumulators (no dependent MFMA chain)
template <int MODE, int NPW, int PF, int NW = 8, bool ACC2 = false, int DIAG = 0>
__global__ __launch_bounds__(NW * 64) void k_unit(const u32x4* __restrict__ w, const float* __restrict__ a_in, float* __restrict__ out,
                                              int units, int n_seg, int store) {
  constexpr int NP = MODE == 0 ? NPW : Split<MODE>::NP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* ap = reinterpret_cast<uint16_t*>(smem);  // [NP][32][kLda] bf16 bit patterns
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the phase that produces A (LayerNorm, activation ...) would write the pieces: truncation split, exact
  for (int i = tid; i < kR * kK; i += NW * 64) {
    float v = a_in[i];
    const int r = i / kK, k = i - r * kK;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if constexpr (MODE == 13) {
        const _Float16 h = (_Float16)v;
        ap[(p * kR + r) * kLda + k] = __builtin_bit_cast(uint16_t, h);
        v -= (float)h;
      } else {
        const uint32_t top = __float_as_uint(v) & 0xffff0000u;
        ap[(p * kR + r) * kLda + k] = (uint16_t)(top >> 16);
        v -= __uint_as_float(top);
      }
    }
  }
  __syncthreads();
  const uint16_t* a_lane = ap + (lane & 31) * kLda + 8 * (lane >> 5);
  constexpr int KSW = kKS * 8 / NW;          // k steps per wave and unit
  constexpr int UNIT_W = KSW * NP * 64;      // 16-byte words of one wave's share of a unit
  constexpr int UNIT = NW * UNIT_W;          // ... of a unit
  const int ks0 = NW == 16 ? (wave & 1) * KSW : 0;
  const int voff = lane * 16;
  f32x16 acc, acc_b;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = acc_b[i] = 0.f;
  u32x4 ring[PF][NP];
  uint32_t sink = 0;
  {
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(w + (size_t)wave * UNIT_W);
#pragma unroll
    for (int s = 0; s < PF; ++s)
#pragma unroll
      for (int p = 0; p < NP; ++p) ring[s][p] = load16(rs, voff, (s * NP + p) * 1024);
  }
  bf16x8 a_fix[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) a_fix[p] = *reinterpret_cast<const bf16x8*>(a_lane + p * kR * kLda);
  for (int u = 0; u < units; ++u) {
    const int seg = u % n_seg, seg_n = (u + 1) % n_seg;
    asm volatile("" ::: "memory");  // A changes from unit to unit in the real kernels: no hoisting of its LDS reads
    const __amdgpu_buffer_rsrc_t rs_b = rsrc_of(w + (size_t)seg * UNIT + (size_t)wave * UNIT_W);
    const __amdgpu_buffer_rsrc_t rs_n = rsrc_of(w + (size_t)seg_n * UNIT + (size_t)wave * UNIT_W);
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
      const int s = ks % PF;
      bf16x8 a[NP], b[NP];
      if constexpr (MODE != 0) {
        // DIAG 1: A fragments read once before the loop (no LDS traffic in the loop)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if constexpr (DIAG == 1) {
            asm volatile("" : "+v"(a_fix[p]));
            a[p] = a_fix[p];
          } else {
            a[p] = *reinterpret_cast<const bf16x8*>(a_lane + p * kR * kLda + (ks0 + ks) * 16);
          }
        }
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) b[p] = __builtin_bit_cast(bf16x8, ring[s][p]);
      if constexpr (DIAG == 6) {  // no weight stream: the MFMAs (and the A reads) alone
      } else if (ks + PF < KSW) {
#pragma unroll
        for (int p = 0; p < NP; ++p) ring[s][p] = load16(rs_b, voff, ((ks + PF) * NP + p) * 1024);
      } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) ring[s][p] = load16(rs_n, voff, ((ks + PF - KSW) * NP + p) * 1024);
      }
      if constexpr (DIAG == 3) {  // MFMAs on operands that do not come from the stream; the stream's data only feed the sink
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const u32x4 q = __builtin_bit_cast(u32x4, b[p]);
          sink ^= q[0] ^ q[1] ^ q[2] ^ q[3];
          asm volatile("" : "+v"(a_fix[p]));
          b[p] = a_fix[p];
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch where it is written (the scheduler sinks the loads to their use)
      if constexpr (MODE == 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const u32x4 q = __builtin_bit_cast(u32x4, b[p]);
          sink ^= q[0] ^ q[1] ^ q[2] ^ q[3];
        }
      } else if constexpr (DIAG == 4 || DIAG == 5) {  // the wave is BLOCKED as long as its MFMAs would take, the matrix pipe stays idle
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const u32x4 q = __builtin_bit_cast(u32x4, b[p]), r = __builtin_bit_cast(u32x4, a[p]);
          sink ^= q[0] ^ q[1] ^ q[2] ^ q[3] ^ r[0] ^ r[1] ^ r[2] ^ r[3];
        }
        if constexpr (DIAG == 4) __builtin_amdgcn_s_sleep(MODE == 6 ? 6 : 3);  // 64 cycles each: both waves of a SIMD in turn
        else __builtin_amdgcn_s_sleep(MODE == 6 ? 3 : 2);
      } else if constexpr (DIAG == 2) {  // LDS reads + weight stream, no MFMA
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const u32x4 q = __builtin_bit_cast(u32x4, b[p]), r = __builtin_bit_cast(u32x4, a[p]);
          sink ^= q[0] ^ q[1] ^ q[2] ^ q[3] ^ r[0] ^ r[1] ^ r[2] ^ r[3];
        }
      } else if constexpr (MODE == 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
      } else if constexpr (MODE == 13) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), acc, 0, 0, 0);
      } else if constexpr (MODE == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
      } else if constexpr (ACC2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc_b, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc_b, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc_b, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
      }
    }
  }
  if constexpr (ACC2) acc += acc_b;
  if constexpr (MODE == 13) acc *= (1.0f / 256.0f);
  if (store) {
    // C/D layout of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      out[(size_t)blockIdx.x * kR * kN + row * kN + 32 * (NW == 16 ? wave >> 1 : wave) + (lane & 31)] = acc[r];
    }
  } else {
    float sacc = __uint_as_float(sink & 1u);
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc += acc[r];
    out[(size_t)blockIdx.x * NW * 64 + tid] = sacc;
  }
}

// ---------------------------------------------------------------- host side
static inline float trunc16(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  u &= 0xffff0000u;
  float r;
  memcpy(&r, &u, 4);
  return r;
}
static inline uint16_t top16(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  return (uint16_t)(u >> 16);
}

// W [n_seg][K][N] fp32 -> [seg][wave][k step][piece][lane][8] bf16
static std::vector<uint16_t> pack(const std::vector<float>& W, int n_seg, int NP, int NW, bool half = false) {
  std::vector<uint16_t> P((size_t)n_seg * kK * kN * NP);
  size_t o = 0;
  for (int seg = 0; seg < n_seg; ++seg)
    for (int wv = 0; wv < NW; ++wv)
      for (int ks = (NW == 16 ? (wv & 1) * (kKS / 2) : 0), ke = ks + kKS * 8 / NW; ks < ke; ++ks)
        for (int p = 0; p < NP; ++p)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
              float v = W[((size_t)seg * kK + ks * 16 + 8 * (l >> 5) + e) * kN + 32 * (NW == 16 ? wv >> 1 : wv) + (l & 31)];
              if (half) {
                v *= 256.0f;
                for (int q = 0; q < p; ++q) v -= (float)(_Float16)v;
                const _Float16 h = (_Float16)v;
                uint16_t bits;
                memcpy(&bits, &h, 2);
                P[o++] = bits;
                continue;
              }
              for (int q = 0; q < p; ++q) v -= trunc16(v);
              P[o++] = top16(v);
            }
  return P;
}

template <int MODE, int NPW, int PF, int NW = 8, bool ACC2 = false, int DIAG = 0>
static void run(const char* name, const std::vector<float>& W, const std::vector<float>& A, int n_seg, int blocks, int units) {
  constexpr int NP = MODE == 0 ? NPW : Split<MODE>::NP;
  std::vector<uint16_t> P = pack(W, n_seg, NP, NW, MODE == 13);
  uint16_t* dW;
  float *dA, *dO;
  CHECK(hipMalloc(&dW, P.size() * 2 + 65536));
  CHECK(hipMemcpy(dW, P.data(), P.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&dA, A.size() * 4));
  CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&dO, (size_t)blocks * kR * kN * 4));
  const size_t lds = (size_t)NP * kR * kLda * 2;
  auto kern = k_unit<MODE, NPW, PF, NW, ACC2, DIAG>;
  // ---- error of ONE unit against float64 (and against what fp32 arithmetic itself gives)
  if (MODE != 0 && NW == 8 && DIAG == 0) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(NW * 64), lds, 0, reinterpret_cast<const u32x4*>(dW), dA, dO, 1, n_seg, 1);
    CHECK(hipDeviceSynchronize());
    std::vector<float> C((size_t)kR * kN);
    CHECK(hipMemcpy(C.data(), dO, C.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, e32max = 0, cmax = 0;
    for (int i = 0; i < kR; ++i)
      for (int j = 0; j < kN; ++j) {
        double s = 0;
        float s32 = 0.f;
        for (int k = 0; k < kK; ++k) {
          s += (double)A[i * kK + k] * (double)W[(size_t)k * kN + j];
          s32 = fmaf(A[i * kK + k], W[(size_t)k * kN + j], s32);
        }
        emax = fmax(emax, fabs((double)C[i * kN + j] - s));
        e32max = fmax(e32max, fabs((double)s32 - s));
        cmax = fmax(cmax, fabs(s));
      }
    printf("%-26s one unit vs float64: max |err| / max |c| = %.3e   (an fp32 fmaf chain: %.3e)\n", name, emax / cmax, e32max / cmax);
  }
  // ---- rate
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), lds, 0, reinterpret_cast<const u32x4*>(dW), dA, dO, units, n_seg, 0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  CHECK(hipGetLastError());
  const int rounds = (blocks + 255) / 256;
  const double us_unit = best * 1e3 / units / rounds;
  const double eq_tf = (double)blocks * units * 2.0 * kR * kK * kN / best / 1e9;
  const double bytes = (double)kK * kN * 2 * NP;
  printf("%-26s blocks=%d units=%d: %.3f ms, %.2f us per unit (the fp32-MFMA unit: 6.83 at peak), %.0f fp32-equivalent TFLOP/s, "
         "weight stream %.0f GB/s per CU = %.1f TB/s over the chip\n",
         name, blocks, units, best, us_unit, eq_tf, bytes / us_unit / 1e3, bytes / us_unit / 1e3 * (blocks < 256 ? blocks : 256) / 1e3);
  CHECK(hipFree(dW));
  CHECK(hipFree(dA));
  CHECK(hipFree(dO));
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  const int units = argc > 2 ? atoi(argv[2]) : 360;
  const int n_seg = argc > 3 ? atoi(argv[3]) : 36;
  std::vector<float> W((size_t)n_seg * kK * kN), A((size_t)kR * kK);
  uint32_t s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xffff) / 65536.0f - 0.5f + ((s >> 4) & 0xff) * 1e-7f;
  };
  for (auto& v : W) v = rnd() * 0.125f;
  for (auto& v : A) v = rnd() * 2.0f;
  printf("split-bf16 unit: C[32x256] += A[32x256] W[256x256] per workgroup and unit; %d workgroups, %d units, %d distinct weight segments\n",
         blocks, units, n_seg);
  run<1, 1, 4>("bf16 x1 (one piece)", W, A, n_seg, blocks, units);
  run<13, 2, 4>("fp16 x3, ring 4", W, A, n_seg, blocks, units);
  run<3, 2, 4>("split x3, ring 4", W, A, n_seg, blocks, units);
  run<3, 2, 8>("split x3, ring 8", W, A, n_seg, blocks, units);
  run<6, 3, 4>("split x6, ring 4", W, A, n_seg, blocks, units);
  run<6, 3, 8>("split x6, ring 8", W, A, n_seg, blocks, units);
  run<3, 2, 2, 8, false>("split x3, ring 2", W, A, n_seg, blocks, units);
  run<6, 3, 4, 8, true>("split x6, ring 4, 2 acc", W, A, n_seg, blocks, units);
  run<6, 3, 4, 16, false>("split x6, 16 waves, ring 4", W, A, n_seg, blocks, units);
  run<6, 3, 4, 16, true>("split x6, 16 waves, 2 acc", W, A, n_seg, blocks, units);
  run<3, 2, 4, 16, false>("split x3, 16 waves, ring 4", W, A, n_seg, blocks, units);
  run<6, 3, 4, 8, true, 6>("x6 2 acc, MFMA + A reads only", W, A, n_seg, blocks, units);
  run<6, 3, 4, 16, false, 6>("x6 16 waves, MFMA + A reads only", W, A, n_seg, blocks, units);
  run<6, 3, 4, 16, true, 6>("x6 16 waves 2 acc, MFMA + A only", W, A, n_seg, blocks, units);
  run<6, 3, 4, 8, false, 6>("x6, MFMA + A reads only", W, A, n_seg, blocks, units);
  run<3, 2, 4, 8, false, 6>("x3, MFMA + A reads only", W, A, n_seg, blocks, units);
  run<1, 1, 4, 8, false, 6>("x1, MFMA + A reads only", W, A, n_seg, blocks, units);
  run<6, 3, 4, 8, false, 4>("x6 ring 4, sleep 384 / step", W, A, n_seg, blocks, units);
  run<6, 3, 4, 8, false, 5>("x6 ring 4, sleep 192 / step", W, A, n_seg, blocks, units);
  run<3, 2, 4, 8, false, 4>("x3 ring 4, sleep 192 / step", W, A, n_seg, blocks, units);
  run<6, 3, 4, 8, false, 3>("x6 ring 4, MFMA off-stream", W, A, n_seg, blocks, units);
  run<3, 2, 4, 8, false, 3>("x3 ring 4, MFMA off-stream", W, A, n_seg, blocks, units);
  run<6, 3, 4, 8, false, 1>("x6 ring 4, A in registers", W, A, n_seg, blocks, units);
  run<6, 3, 4, 8, false, 2>("x6 ring 4, no MFMA", W, A, n_seg, blocks, units);
  run<3, 2, 4, 8, false, 1>("x3 ring 4, A in registers", W, A, n_seg, blocks, units);
  run<3, 2, 4, 8, false, 2>("x3 ring 4, no MFMA", W, A, n_seg, blocks, units);
  run<0, 2, 8>("stream only, 4 B/weight", W, A, n_seg, blocks, units);
  run<0, 3, 8>("stream only, 6 B/weight", W, A, n_seg, blocks, units);
  return 0;
}
