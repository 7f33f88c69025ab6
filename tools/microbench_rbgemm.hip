// Microbenchmark of the row-block GEMM core (rowblock.h) to find what bounds it:
//   mode 0: weight stream from distinct addresses per block-wave (as the real kernels)
//   mode 1: every iteration re-reads the same 32 KiB segment (L1/L2-hot)
//   mode 2: no weight loads in the loop (ring never refilled)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ppasr_amd/csrc tools/microbench_rbgemm.hip -o /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "rowblock.h"
using namespace ppasr;

template <int MT, int MODE>
__global__ __launch_bounds__(kThreads) void k_mb(const f32x4* __restrict__ w, float* __restrict__ out, int iters, int n_seg, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = lane_id(), wave = wave_id();
  for (int i = threadIdx.x; i < 32 * MT * kLda; i += kThreads) smem[i] = (float)(i % 13) * 0.01f;
  __syncthreads();
  BRing<1> ring;
  long long c0 = clock64(), w0 = wall_clock64();
  const f32x4* base = w + (size_t)wave * kTs256;
  ring_prime(ring, base, 0);
  f32x16 acc[MT][1];
  acc_zero(acc);
  for (int it = 0; it < iters; ++it) {
    const f32x4* seg = (MODE == 0) ? base + (size_t)((it % n_seg) * 8) * kTs256 : base;
    const f32x4* nxt = (MODE == 0) ? base + (size_t)(((it + 1) % n_seg) * 8) * kTs256 : base;
    rb_gemm<MT, 1, kG256>(smem, kLda, seg, 0, MODE == 2 ? nullptr : nxt, 0, ring, acc);
  }
  float s = 0.f;
  for (int mt = 0; mt < MT; ++mt)
    for (int r = 0; r < 16; ++r) s += acc[mt][0][r];
  out[(size_t)blockIdx.x * kThreads + threadIdx.x] = s;
  if (blockIdx.x == 3 && threadIdx.x == 0) {
    long long c1 = clock64(), w1 = wall_clock64();
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}

static size_t g_lds_override = 0;
template <int MT, int MODE>
void run(const f32x4* w, float* out, int blocks, int iters, int n_seg, const char* name) {
  size_t lds = g_lds_override ? g_lds_override : 32 * MT * kLda * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_mb<MT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  static long long* clk = nullptr;
  if (!clk) hipMalloc(&clk, 16);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_mb<MT, MODE>), dim3(blocks), dim3(kThreads), lds, 0, w, out, iters, n_seg, clk);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  double flops = (double)blocks * 8 * iters * 128.0 * MT * 4096.0;
  long long hc[2];
  hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
  int wc_khz = 0;
  hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
  double sclk_mhz = (double)hc[0] / ((double)hc[1] / (wc_khz * 1e3)) / 1e6;
  double busy = (double)iters * 128.0 * MT * 2 * 64.0 / (double)hc[0];
  printf("%-28s blocks=%4d MT=%d iters=%d: %.3f ms  %.1f TFLOP/s  sclk=%.0f MHz  in-kernel cycles=%lld  mfma-pipe busy=%.1f%%\n", name, blocks, MT, iters, ms,
         flops / ms / 1e9, sclk_mhz, hc[0], busy * 100);
}

int main() {
  const int n_seg = 16;  // 16 segments x 8 tiles x 32 KiB = 4 MiB weight stream (one FFN's worth)
  size_t n = (size_t)n_seg * 8 * kTs256 + 8 * kTs256;
  std::vector<float> h(n * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
  f32x4* w; float* out;
  hipMalloc(&w, n * 16); hipMalloc(&out, 4096 * kThreads * 4);
  hipMemcpy(w, h.data(), n * 16, hipMemcpyHostToDevice);
  const int iters = 128;
  run<1, 0>(w, out, 249, iters, n_seg, "MT=1 stream (real pattern)");
  run<1, 1>(w, out, 249, iters, n_seg, "MT=1 hot segment");
  run<1, 2>(w, out, 249, iters, n_seg, "MT=1 no loads");
  run<2, 0>(w, out, 249, iters, n_seg, "MT=2 stream");
  run<2, 2>(w, out, 249, iters, n_seg, "MT=2 no loads");
  run<4, 0>(w, out, 249, iters, n_seg, "MT=4 stream");
  run<1, 0>(w, out, 256, iters, n_seg, "MT=1 stream 256 blocks");
  run<1, 0>(w, out, 128, iters, n_seg, "MT=1 stream 128 blocks");
  run<1, 0>(w, out, 64, iters, n_seg, "MT=1 stream 64 blocks");
  run<1, 0>(w, out, 8, iters, n_seg, "MT=1 stream 8 blocks");
  // fixed cost of a row-block kernel: time vs number of 128-MFMA GEMM calls, small and large LDS footprints
  for (size_t lds : {(size_t)33280, (size_t)133120}) {
    g_lds_override = lds;
    for (int it : {0, 1, 2, 4, 8, 16}) {
      char name[64];
      snprintf(name, sizeof name, "lds=%zuK iters=%d", lds / 1024, it);
      run<1, 0>(w, out, 249, it, n_seg, name);
    }
  }
  return 0;
}
