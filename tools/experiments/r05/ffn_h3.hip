// Prototype of ONE feed-forward module of the row-block kernels on the fp16 x3 matrix-core route (round-5 groundwork, not
// product code):   y = x + 0.5 * (swish(LN(x) W1 + b1) W2 + b2),   d = 256, hidden 2048, 32 rows per workgroup, 8 waves,
// in the product kernels' pattern (ffn_phase_t, csrc/phases_t.h): the hidden layer in eight 256-wide chunks, W1 chunk ->
// swish -> LDS -> W2 chunk, accumulators transposed (weights are the MFMA's first operand) so that a lane holds quads of
// four consecutive output features of ONE row and can write the next GEMM's operand with 8-byte LDS stores.
//
// fp16 x3: every GEMM operand is the sum of two fp16 pieces (round to nearest; 22 significant bits), pre-scaled by a power
// of two so that the low piece stays a normal fp16 number (weights 2^8 at pack time, activations 2^4 when a phase writes
// them to LDS); a product of two pieces is exact in fp32; three v_mfma_f32_32x32x16_f16 per 16-wide k step instead of eight
// v_mfma_f32_32x32x2_f32, the same 4 bytes per weight in the stream.  mb_split.hip: one unit 2.6e-7 of float64 (an fp32 fmaf
// chain 5.7e-7); split_bf16_numerics.py: the 12-block Conformer's logits as close to float64 as fp32 arithmetic is.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/r05/ffn_h3.hip -o tools/experiments/r05/ffn_h3
// run:   ffn_h3 [blocks=250] [reps=20]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kD = 256, kH = 2048, kR = 32, kNC = kH / 256;
constexpr int kLdx = kD + 4;   // fp32 row stride of the residual tile
constexpr int kLdh = 256 + 8;  // fp16 row stride of an operand plane: 528 B, 16-byte fragment reads conflict-free
constexpr int kKS = 16;        // k steps of a 256-deep unit
constexpr int kPF = 4;         // ring depth in k steps
constexpr float kSA = 16.f, kSW = 256.f, kInv = 1.f / (kSA * kSW);

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 load16(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// the two fp16 pieces of four values (already scaled), as two 8-byte words
__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (_Float16)v[i];
    lo[i] = (_Float16)(v[i] - (float)hi[i]);
  }
}

// One 256-deep unit for this wave's 32 output features: acc[feature quad][row] += W^T pieces x A pieces.
//   a_hi / a_lo: this lane's fragment base in the two operand planes (row lane & 31, k offset 8 * (lane >> 5))
//   ring: holds the unit's first kPF k steps on entry, the NEXT unit's on exit (rs_n; nullptr-equivalent: same unit again)
__device__ __forceinline__ void unit_h3(f32x16& acc, const _Float16* a_hi, const _Float16* a_lo, u32x4 (&ring)[kPF][2],
                                        __amdgpu_buffer_rsrc_t rs_b, __amdgpu_buffer_rsrc_t rs_n, int voff) {
#pragma unroll
  for (int ks = 0; ks < kKS; ++ks) {
    const int s = ks % kPF;
    const f16x8 a0 = *reinterpret_cast<const f16x8*>(a_hi + ks * 16), a1 = *reinterpret_cast<const f16x8*>(a_lo + ks * 16);
    const f16x8 w0 = __builtin_bit_cast(f16x8, ring[s][0]), w1 = __builtin_bit_cast(f16x8, ring[s][1]);
    if (ks + kPF < kKS) {
      ring[s][0] = load16(rs_b, voff, ((ks + kPF) * 2 + 0) * 1024);
      ring[s][1] = load16(rs_b, voff, ((ks + kPF) * 2 + 1) * 1024);
    } else {
      ring[s][0] = load16(rs_n, voff, ((ks + kPF - kKS) * 2 + 0) * 1024);
      ring[s][1] = load16(rs_n, voff, ((ks + kPF - kKS) * 2 + 1) * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a0, acc, 0, 0, 0);
  }
}

// weights: [unit u < 16][wave][k step][piece][lane] 16 bytes; unit 2c = W1 chunk c, unit 2c + 1 = W2 chunk c
__global__ __launch_bounds__(512) void k_ffn_h3(const float* __restrict__ x, const u32x4* __restrict__ w, const float* __restrict__ ln_g,
                                                const float* __restrict__ ln_b, const float* __restrict__ b1, const float* __restrict__ b2,
                                                float* __restrict__ y, int M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);                                  // [32][kLdx] residual input
  _Float16* a0p = reinterpret_cast<_Float16*>(smem + kR * kLdx * 4);          // [2 pieces][32][kLdh] LN(x) * 2^4
  _Float16* hp = a0p + 2 * kR * kLdh;                                          // [2 buffers][2 pieces][32][kLdh] swish chunk * 2^4
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * kR;
  constexpr int UNIT_W = kKS * 2 * 64, UNIT = 8 * UNIT_W;  // 16-byte words per wave / per unit
  const int voff = lane * 16;
  u32x4 ring[kPF][2];
  {  // the weight stream starts before the rows arrive
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(w + (size_t)wave * UNIT_W);
#pragma unroll
    for (int s = 0; s < kPF; ++s) {
      ring[s][0] = load16(rs, voff, (s * 2 + 0) * 1024);
      ring[s][1] = load16(rs, voff, (s * 2 + 1) * 1024);
    }
  }
  // ---- rows in, LayerNorm (wave w: rows 4w .. 4w+3, a lane holds 4 consecutive columns), operand pieces out
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 4 + i, m = m0 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < M) v = *reinterpret_cast<const f32x4*>(x + (size_t)m * kD + lane * 4);
    *reinterpret_cast<f32x4*>(xs + r * kLdx + lane * 4) = v;
    const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / kD);
    f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
    const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / kD);
    const float rstd = rsqrtf(var + 1e-5f);
    const f32x4 g = *reinterpret_cast<const f32x4*>(ln_g + lane * 4), bb = *reinterpret_cast<const f32x4*>(ln_b + lane * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (d[j] * rstd * g[j] + bb[j]) * kSA;
    f16x4 hi, lo;
    split4(o, hi, lo);
    *reinterpret_cast<f16x4*>(a0p + r * kLdh + lane * 4) = hi;
    *reinterpret_cast<f16x4*>(a0p + (kR + r) * kLdh + lane * 4) = lo;
  }
  __syncthreads();
  const int frag = (lane & 31) * kLdh + 8 * (lane >> 5);  // this lane's fragment: row lane & 31, k offset 8 * (lane >> 5)
  f32x16 acc2;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
  // accumulator register r of a lane: output feature 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), row lane & 31
  const int row = lane & 31, fq = 32 * wave + 4 * (lane >> 5);
  for (int c = 0; c < kNC; ++c) {
    _Float16* hb = hp + (c & 1) * 2 * kR * kLdh;
    const __amdgpu_buffer_rsrc_t rs_1 = rsrc_of(w + (size_t)(2 * c) * UNIT + (size_t)wave * UNIT_W);
    const __amdgpu_buffer_rsrc_t rs_2 = rsrc_of(w + (size_t)(2 * c + 1) * UNIT + (size_t)wave * UNIT_W);
    const __amdgpu_buffer_rsrc_t rs_3 = rsrc_of(w + (size_t)((2 * c + 2) % (2 * kNC)) * UNIT + (size_t)wave * UNIT_W);
    f32x16 acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
    unit_h3(acc1, a0p + frag, a0p + kR * kLdh + frag, ring, rs_1, rs_2, voff);
    // swish(h + b1) -> pieces of the W2 operand (hidden features 256 c + 32 wave + ...)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = fq + 8 * q;
      const f32x4 bq = *reinterpret_cast<const f32x4*>(b1 + c * 256 + f);
      f32x4 hq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = acc1[4 * q + j] * kInv + bq[j];
        hq[j] = t / (1.f + __expf(-t)) * kSA;
      }
      f16x4 hi, lo;
      split4(hq, hi, lo);
      *reinterpret_cast<f16x4*>(hb + row * kLdh + f) = hi;
      *reinterpret_cast<f16x4*>(hb + (kR + row) * kLdh + f) = lo;
    }
    __syncthreads();  // (the buffer written two chunks ago is free again: its readers passed this barrier one chunk later)
    unit_h3(acc2, hb + frag, hb + kR * kLdh + frag, ring, rs_2, rs_3, voff);
  }
  // ---- y = x + 0.5 * (acc2 + b2)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = fq + 8 * q, m = m0 + row;
    const f32x4 bq = *reinterpret_cast<const f32x4*>(b2 + f);
    const f32x4 xr = *reinterpret_cast<const f32x4*>(xs + row * kLdx + f);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = xr[j] + 0.5f * (acc2[4 * q + j] * kInv + bq[j]);
    if (m < M) *reinterpret_cast<f32x4*>(y + (size_t)m * kD + f) = o;
  }
}

// ---------------------------------------------------------------- host
static uint16_t h_bits(_Float16 h) {
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
// Wt [K][N] (x @ W layout) rows k0 .. k0+255, columns n0 .. n0+255 -> one unit in stream order
static void pack_unit(const std::vector<float>& W, int ldw, int k0, int n0, std::vector<uint16_t>& P) {
  for (int wv = 0; wv < 8; ++wv)
    for (int ks = 0; ks < kKS; ++ks)
      for (int p = 0; p < 2; ++p)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 8; ++e) {
            float v = W[(size_t)(k0 + ks * 16 + 8 * (l >> 5) + e) * ldw + n0 + 32 * wv + (l & 31)] * kSW;
            const _Float16 hi = (_Float16)v;
            P.push_back(h_bits(p == 0 ? hi : (_Float16)(v - (float)hi)));
          }
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 250;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int M = blocks * kR;
  uint32_t s = 777u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xffffff) / 16777216.0f - 0.5f;
  };
  std::vector<float> X((size_t)M * kD), W1((size_t)kD * kH), W2((size_t)kH * kD), G(kD), Bn(kD), B1(kH), B2(kD);
  for (auto& v : X) v = rnd() * 4.f + 0.3f;
  for (auto& v : W1) v = rnd() * 0.2f;    // ~ xavier for 256 -> 2048
  for (auto& v : W2) v = rnd() * 0.08f;
  for (auto& v : G) v = 1.f + rnd() * 0.2f;
  for (auto& v : Bn) v = rnd() * 0.2f;
  for (auto& v : B1) v = rnd() * 0.2f;
  for (auto& v : B2) v = rnd() * 0.2f;
  std::vector<uint16_t> P;
  P.reserve((size_t)2 * kD * kH * 2);
  for (int c = 0; c < kNC; ++c) {
    pack_unit(W1, kH, 0, c * 256, P);  // hidden features 256 c ..
    pack_unit(W2, kD, c * 256, 0, P);  // ... and the rows of W2 they multiply
  }
  float *dX, *dY, *dG, *dBn, *dB1, *dB2;
  uint16_t* dW;
  CHECK(hipMalloc(&dX, X.size() * 4));
  CHECK(hipMalloc(&dY, X.size() * 4));
  CHECK(hipMalloc(&dW, P.size() * 2 + 65536));
  CHECK(hipMalloc(&dG, kD * 4));
  CHECK(hipMalloc(&dBn, kD * 4));
  CHECK(hipMalloc(&dB1, kH * 4));
  CHECK(hipMalloc(&dB2, kD * 4));
  CHECK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dW, P.data(), P.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dG, G.data(), kD * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dBn, Bn.data(), kD * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB1, B1.data(), kH * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB2, B2.data(), kD * 4, hipMemcpyHostToDevice));
  const size_t lds = (size_t)kR * kLdx * 4 + (size_t)6 * kR * kLdh * 2;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ffn_h3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < reps + 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_ffn_h3, dim3(blocks), dim3(512), lds, 0, dX, reinterpret_cast<const u32x4*>(dW), dG, dBn, dB1, dB2, dY, M);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 2 && ms < best) best = ms;
  }
  CHECK(hipGetLastError());
  std::vector<float> Y(X.size());
  CHECK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
  // ---- reference: float64, and the same module in plain fp32 arithmetic (what the fp32-MFMA kernels compute), rows of 2 blocks
  double e_h3 = 0, e_f32 = 0, ymax = 0;
  const int check_rows[] = {0, 1, 17, 31, kR * (blocks / 2) + 5, M - 1};
  for (int m : check_rows) {
    std::vector<double> ln(kD), hid(kH);
    std::vector<float> lnf(kD), hidf(kH);
    double mean = 0, var = 0;
    for (int j = 0; j < kD; ++j) mean += X[(size_t)m * kD + j];
    mean /= kD;
    for (int j = 0; j < kD; ++j) var += (X[(size_t)m * kD + j] - mean) * (X[(size_t)m * kD + j] - mean);
    var /= kD;
    float meanf = 0, varf = 0;
    for (int j = 0; j < kD; ++j) meanf += X[(size_t)m * kD + j];
    meanf /= kD;
    for (int j = 0; j < kD; ++j) varf += (X[(size_t)m * kD + j] - meanf) * (X[(size_t)m * kD + j] - meanf);
    varf /= kD;
    for (int j = 0; j < kD; ++j) {
      ln[j] = (X[(size_t)m * kD + j] - mean) / sqrt(var + 1e-5) * G[j] + Bn[j];
      lnf[j] = (X[(size_t)m * kD + j] - meanf) / sqrtf(varf + 1e-5f) * G[j] + Bn[j];
    }
    for (int h = 0; h < kH; ++h) {
      double a = B1[h];
      float af = 0.f;
      for (int k = 0; k < kD; ++k) {
        a += ln[k] * W1[(size_t)k * kH + h];
        af = fmaf(lnf[k], W1[(size_t)k * kH + h], af);
      }
      af += B1[h];
      hid[h] = a / (1.0 + exp(-a));
      hidf[h] = af / (1.f + expf(-af));
    }
    for (int j = 0; j < kD; ++j) {
      double a = B2[j];
      float af = 0.f;
      for (int h = 0; h < kH; ++h) {
        a += hid[h] * W2[(size_t)h * kD + j];
        af = fmaf(hidf[h], W2[(size_t)h * kD + j], af);
      }
      af += B2[j];
      const double ref = X[(size_t)m * kD + j] + 0.5 * a;
      const float reff = X[(size_t)m * kD + j] + 0.5f * af;
      e_h3 = fmax(e_h3, fabs((double)Y[(size_t)m * kD + j] - ref));
      e_f32 = fmax(e_f32, fabs((double)reff - ref));
      ymax = fmax(ymax, fabs(ref));
    }
  }
  const double flops = (double)M * 2.0 * 2.0 * kD * kH;
  printf("ffn fp16x3: %d workgroups x 32 rows: %.1f us per launch (16 units: %.2f us per unit incl. LayerNorm, swish, residual), "
         "%.0f fp32-equivalent TFLOP/s\n", blocks, best * 1e3, best * 1e3 / 16 / ((blocks + 255) / 256), flops / best / 1e9);
  printf("  the same module on the fp32-MFMA kernels: 16 units x 6.83 us = 109 us at the matrix-pipe peak\n");
  printf("  max |y - float64| / max |y|: fp16x3 kernel %.3e, plain fp32 arithmetic (fmaf chains) %.3e\n", e_h3 / ymax, e_f32 / ymax);
  return 0;
}
