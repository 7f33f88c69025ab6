// Microbenchmark of the 16-row row-block GEMM core (csrc/rbt.h, v_mfma_f32_16x16x4_f32 on the 32-row kernels' packed
// weights) next to the 32-row one, in the real kernels' pattern: every workgroup streams the same 4 MiB of weights
// (one FFN's worth) through its register ring, A operand from LDS.  Reports TFLOP/s over the whole chip and the
// fraction of the fp32-MFMA rate of the CUs that have work.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ppasr_amd/csrc tools/microbench_rb16.hip -o /tmp/mb16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rbt.h"
using namespace ppasr;

template <int R>
__global__ __launch_bounds__(kThreads) void k_mb(const f32x4* __restrict__ w, float* __restrict__ out, int iters, int n_seg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = wave_id();
  for (int i = threadIdx.x; i < R * kLda; i += kThreads) smem[i] = (float)(i % 13) * 0.01f;
  __syncthreads();
  typename RBT<R>::Ring ring;
  const f32x4* base = w + (size_t)wave * kTs256;
  rbt_prime(ring, base);
  typename RBT<R>::Acc acc;
  RBT<R>::zero(acc);
  for (int it = 0; it < iters; ++it) {
    const f32x4* seg = base + (size_t)((it % n_seg) * 8) * kTs256;
    const f32x4* nxt = base + (size_t)(((it + 1) % n_seg) * 8) * kTs256;
    rbt_gemm<kG256>(smem, kLda, seg, nxt, ring, acc);
  }
  float s = 0.f;
  for (int q = 0; q < RBT<R>::NQ; ++q) {
    const f32x4 v = RBT<R>::quad(acc, q);
    s += v[0] + v[1] + v[2] + v[3];
  }
  out[(size_t)blockIdx.x * kThreads + threadIdx.x] = s;
}

template <int R>
void run(const f32x4* w, float* out, int blocks, int iters, int n_seg, size_t lds, const char* name) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_mb<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_mb<R>), dim3(blocks), dim3(kThreads), lds, 0, w, out, iters, n_seg);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)blocks * iters * (double)R * 256.0 * 256.0 * 2.0;
  const double tf = flops / ms / 1e9;
  const int cus = blocks < 256 ? blocks : 256;
  printf("%-44s R=%2d blocks=%4d lds=%3zuK: %.3f ms  %6.1f TFLOP/s  = %.1f %% of the fp32-MFMA rate of %d CUs; %.2f us per unit\n", name, R,
         blocks, lds / 1024, ms, tf, 100.0 * tf / (157.3 * cus / 256.0), cus, ms * 1e3 / iters / ((blocks + 255) / 256));
}

// 16 waves on 32 rows: wave w owns the 16-column sub-tile (w & 1) of column tile (w >> 1) for BOTH 16-row halves --
// one buffer load + two ds_read_b128 per eight v_mfma_f32_16x16x4_f32; the weights cross L1 once per workgroup as in
// the 32-row form, and every SIMD has four waves to interleave
template <int PF>
__global__ __launch_bounds__(1024) void k_mb_w16(const f32x4* __restrict__ w, float* __restrict__ out, int iters, int n_seg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32 * kLda; i += 1024) smem[i] = (float)(i % 13) * 0.01f;
  __syncthreads();
  const f32x4* base = w + (size_t)(wave >> 1) * kTs256;
  const int voff = rbt16_voff(lane, wave & 1);
  constexpr int GS = kG256 / 2;
  f32x4 ring[PF];
  {
    const __amdgpu_buffer_rsrc_t rs = wstream_rsrc(base);
#pragma unroll
    for (int g = 0; g < PF; ++g) ring[g] = wstream_load(rs, voff, g * 2048);
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  const float* a_ptr = smem + (lane & 15) * kLda + 4 * (lane >> 4);
  for (int it = 0; it < iters; ++it) {
    const f32x4* seg = base + (size_t)((it % n_seg) * 8) * kTs256;
    const f32x4* nxt = base + (size_t)(((it + 1) % n_seg) * 8) * kTs256;
    const __amdgpu_buffer_rsrc_t rs_b = wstream_rsrc(seg), rs_n = wstream_rsrc(nxt);
    f32x4 a0 = *reinterpret_cast<const f32x4*>(a_ptr), a1 = *reinterpret_cast<const f32x4*>(a_ptr + 16 * kLda);
    f32x4 n0 = a0, n1 = a1;
#pragma unroll
    for (int g = 0; g < GS; ++g) {
      const int sl = g % PF;
      if (g + 1 < GS) {
        n0 = *reinterpret_cast<const f32x4*>(a_ptr + 16 * (g + 1));
        n1 = *reinterpret_cast<const f32x4*>(a_ptr + 16 * kLda + 16 * (g + 1));
      }
      const f32x4 b = ring[sl];
      if (g + PF < GS) ring[sl] = wstream_load(rs_b, voff, (g + PF) * 2048);
      else ring[sl] = wstream_load(rs_n, voff, (g + PF - GS) * 2048);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[m], a0[m], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[m], a1[m], acc1, 0, 0, 0);
      }
      a0 = n0; a1 = n1;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  out[(size_t)blockIdx.x * 1024 + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}

template <int PF>
void run_w16(const f32x4* w, float* out, int blocks, int iters, int n_seg, size_t lds, const char* name) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_mb_w16<PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_mb_w16<PF>), dim3(blocks), dim3(1024), lds, 0, w, out, iters, n_seg);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)blocks * iters * 32.0 * 256.0 * 256.0 * 2.0;
  const double tf = flops / ms / 1e9;
  const int cus = blocks < 256 ? blocks : 256;
  printf("%-44s PF=%d blocks=%4d lds=%3zuK: %.3f ms  %6.1f TFLOP/s  = %.1f %% of the fp32-MFMA rate of %d CUs; %.2f us per unit\n", name, PF,
         blocks, lds / 1024, ms, tf, 100.0 * tf / (157.3 * cus / 256.0), cus, ms * 1e3 / iters / ((blocks + 255) / 256));
}

int main(int argc, char** argv) {
  // weights every workgroup streams: n_seg x 256 KiB (16 = 4 MiB = one XCD's L2; a Conformer layer tail is 9.4 MB = 36)
  const int n_seg = argc > 1 ? atoi(argv[1]) : 16;
  printf("---- %d segments = %.1f MiB of weights per pass ----\n", n_seg, n_seg * 0.25);
  size_t n = (size_t)n_seg * 8 * kTs256 + 8 * kTs256;
  std::vector<float> h(n * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
  f32x4* w; float* out;
  hipMalloc(&w, n * 16); hipMalloc(&out, 4096 * kThreads * 4);
  hipMemcpy(w, h.data(), n * 16, hipMemcpyHostToDevice);
  const int iters = 256;
  run<32>(w, out, 256, iters, n_seg, 133120, "32-row, one block per CU");
  run<32>(w, out, 104, iters, n_seg, 133120, "32-row, 104 blocks (cfg5 half-rate layers)");
  run<16>(w, out, 256, iters, n_seg, 133120, "16-row, one block per CU");
  run<16>(w, out, 208, iters, n_seg, 133120, "16-row, 208 blocks");
  run<16>(w, out, 512, iters, n_seg, 66560, "16-row, two blocks per CU (4 waves / SIMD)");
  run<16>(w, out, 416, iters, n_seg, 66560, "16-row, 416 blocks at two per CU");
  run<16>(w, out, 1024, iters, n_seg, 33280, "16-row, four blocks per CU");
  run<32>(w, out, 512, iters, n_seg, 66560, "32-row, two blocks per CU");
  run_w16<4>(w, out, 256, iters, n_seg, 133120, "32-row on 16 waves (16 columns each)");
  run_w16<2>(w, out, 256, iters, n_seg, 133120, "32-row on 16 waves (16 columns each)");
  run_w16<4>(w, out, 249, iters, n_seg, 133120, "32-row on 16 waves, 249 blocks (cfg2)");
  return 0;
}
