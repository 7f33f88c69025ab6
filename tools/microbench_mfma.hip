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
  return 0;
}
