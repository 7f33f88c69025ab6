// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  Launches one-workgroup-per-CU kernels on
// streams with a few candidate masks and prints the (XCC, SE, SH, CU) histogram each one lands on -- the bit order of the
// mask is not documented for multi-XCD parts, so the pipelined decode plans (ppasr_amd/parallel.py) are built on what
// this prints.
// build: hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/cu_mask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <map>
#include <vector>

__global__ __launch_bounds__(512) void k_probe(uint32_t* out, int spin) {
  extern __shared__ float smem[];
  if (threadIdx.x == 0) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  smem[threadIdx.x] = a;
}

static void run(const char* name, const std::vector<uint32_t>& mask, int blocks) {
  hipStream_t st;
  hipError_t e = mask.empty() ? hipStreamCreate(&st) : hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%s: stream create failed: %s\n", name, hipGetErrorString(e)); return; }
  uint32_t* d;
  hipMalloc(&d, blocks * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024);
  hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(512), 82 * 1024, st, d, 20000);
  hipStreamSynchronize(st);
  std::vector<uint32_t> h(blocks * 2);
  hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  std::map<uint32_t, int> cus;  // key: xcc<<16 | se<<8 | sh<<4.. raw fields
  std::map<uint32_t, int> per_xcc;
  for (int b = 0; b < blocks; ++b) {
    const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    per_xcc[xcc]++;
  }
  printf("%-28s %4d blocks -> %3zu distinct CUs; per XCC:", name, blocks, cus.size());
  for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
  printf("\n   (xcc.se.sh.cu):");
  int n = 0;
  for (auto& kv : cus) {
    if (n++ < 40) printf(" %u.%u.%u.%u", kv.first >> 12, (kv.first >> 8) & 0xf, (kv.first >> 4) & 0xf, kv.first & 0xf);
  }
  printf("%s\n", cus.size() > 40 ? " ..." : "");
  hipFree(d);
  hipStreamDestroy(st);
}

// a layer-kernel-shaped launch: `blocks` workgroups of 512 threads and 133 KB of LDS (one per CU), ~100 us each: how many
// CUs does it spread over, and how long does the whole launch take?
__global__ __launch_bounds__(512) void k_busy(uint32_t* out, long long* t, int spin) {
  extern __shared__ float smem[];
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  smem[threadIdx.x] = a;
  (void)c0;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    t[2 * blockIdx.x] = t0;
    t[2 * blockIdx.x + 1] = wall_clock64();
  }
}
__global__ __launch_bounds__(512) void k_spin(float* out, int spin) {
  extern __shared__ float smem[];
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  smem[threadIdx.x] = a;
  if (a == 12345.f) out[0] = a;
}
// ragged launch: the whole padded grid is launched, the workgroups of skipped row blocks exit at once (what the layer
// kernels do under ppasr_set_skip_padding); `active` = the table variant: only the active blocks, in front
__global__ __launch_bounds__(512) void k_ragged(const uint8_t* act, long long* t, int spin) {
  extern __shared__ float smem[];
  if (!act[blockIdx.x]) return;
  const long long t0 = wall_clock64();
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  smem[threadIdx.x] = a;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = wall_clock64(); }
}
static void ragged(const char* name, const std::vector<uint8_t>& act, size_t lds, bool with_search = false) {
  const int blocks = (int)act.size();
  hipStream_t s2;
  (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  float* o2;
  (void)hipMalloc(&o2, 64);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_spin), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  uint8_t* d; long long* t;
  (void)hipMalloc(&d, blocks); (void)hipMalloc(&t, blocks * 16);
  (void)hipMemcpy(d, act.data(), blocks, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ragged), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  double worst = 0, sum = 0;
  int n_act = 0;
  for (auto a : act) n_act += a;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipMemset(t, 0, blocks * 16);
    // (40 KB of LDS: the 16 "search" workgroups cannot share a CU with the 133 KB blocks, they only take CUs away)
    if (with_search) hipLaunchKernelGGL(k_spin, dim3(16), dim3(512), 40 * 1024, s2, o2, 400000);
    hipLaunchKernelGGL(k_ragged, dim3(blocks), dim3(512), lds, 0, d, t, 60000);
    (void)hipDeviceSynchronize();
    std::vector<long long> ht(blocks * 2);
    (void)hipMemcpy(ht.data(), t, blocks * 16, hipMemcpyDeviceToHost);
    long long tmin = 0, tmax = 0;
    for (int b = 0; b < blocks; ++b)
      if (act[b]) { if (!tmin || ht[2 * b] < tmin) tmin = ht[2 * b]; tmax = std::max(tmax, ht[2 * b + 1]); }
    if (rep) { worst = std::max(worst, (tmax - tmin) / 100.0); sum += (tmax - tmin) / 100.0; }
  }
  printf("%-46s grid %3d, %3d active, LDS %3zu KB%s: launch mean %.1f us, worst %.1f us (one block ~820)\n", name, blocks, n_act, lds / 1024,
         with_search ? ", 16 x 512-thread workgroups on another stream" : "", sum / 5, worst);
  (void)hipFree(d); (void)hipFree(t); (void)hipFree(o2); (void)hipStreamDestroy(s2);
}

static void busy(const char* name, const std::vector<uint32_t>& mask, int blocks) {
  hipStream_t st;
  hipError_t e = mask.empty() ? hipStreamCreate(&st) : hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%s: stream create failed\n", name); return; }
  uint32_t* d; long long* t;
  (void)hipMalloc(&d, blocks * 8); (void)hipMalloc(&t, blocks * 16);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_busy), hipFuncAttributeMaxDynamicSharedMemorySize, 133 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(512), 133 * 1024, st, d, t, 60000);
    (void)hipStreamSynchronize(st);
  }
  std::vector<uint32_t> h(blocks * 2); std::vector<long long> ht(blocks * 2);
  (void)hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(ht.data(), t, blocks * 16, hipMemcpyDeviceToHost);
  std::map<uint32_t, int> cus, per_se;
  long long tmin = ht[0], tmax = ht[1], dmax = 0;
  for (int b = 0; b < blocks; ++b) {
    const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    cus[(xcc << 12) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 0xf)]++;
    per_se[(xcc << 4) | ((hw >> 13) & 7)]++;
    tmin = std::min(tmin, ht[2 * b]); tmax = std::max(tmax, ht[2 * b + 1]); dmax = std::max(dmax, ht[2 * b + 1] - ht[2 * b]);
  }
  int twice = 0, se_max = 0;
  for (auto& kv : cus) twice += kv.second > 1;
  for (auto& kv : per_se) se_max = std::max(se_max, kv.second);
  printf("%-34s %3d blocks: %3zu CUs, %d CUs ran 2+ blocks, most blocks on one SE %d; launch %.1f us, longest block %.1f us\n", name, blocks,
         cus.size(), twice, se_max, (tmax - tmin) / 100.0, dmax / 100.0);
  (void)hipFree(d); (void)hipFree(t); (void)hipStreamDestroy(st);
}

// does a kernel slow down merely because ANOTHER queue has a kernel running?  `blocks` busy workgroups (fixed work, ~820 us
// alone) on one stream while 16 spinning workgroups (19 KB of LDS, 512 threads: the beam search's footprint) occupy
// another stream for ~4 ms
static void concurrent(int blocks, bool with_other, int prio, int other_wgs = 16, int other_threads = 512) {
  hipStream_t s1, s2;
  int lo, hi;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  (void)hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, prio ? hi : 0);
  (void)hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, 0);
  uint32_t* d; long long* t; float* o;
  (void)hipMalloc(&d, blocks * 8); (void)hipMalloc(&t, blocks * 16); (void)hipMalloc(&o, 64);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_busy), hipFuncAttributeMaxDynamicSharedMemorySize, 133 * 1024);
  double sum = 0;
  for (int rep = 0; rep < 4; ++rep) {
    if (with_other) hipLaunchKernelGGL(k_spin, dim3(other_wgs), dim3(other_threads), 19 * 1024, s2, o, 300000);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s1);
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(512), 133 * 1024, s1, d, t, 60000);
    (void)hipEventRecord(b, s1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (rep) sum += ms / 3;
  }
  // shader clock of the last launch's blocks: s_memtime cycles / wall-clock time
  std::vector<long long> ht(blocks * 2);
  (void)hipMemcpy(ht.data(), t, blocks * 16, hipMemcpyDeviceToHost);
  printf("%3d busy blocks (133 KB) on stream A%s%s: %.1f us per launch", blocks, with_other ? " + spinning workgroups on stream B" : "",
         prio ? " (A high priority)" : "", sum / 3 * 1000);
  if (with_other) printf(" [B: %d x %d threads]", other_wgs, other_threads);
  long long tmin = ht[0], tmax = ht[1], dmax = 0;
  for (int b = 0; b < blocks; ++b) {
    tmin = std::min(tmin, ht[2 * b]); tmax = std::max(tmax, ht[2 * b + 1]); dmax = std::max(dmax, ht[2 * b + 1] - ht[2 * b]);
  }
  printf("; last launch: first start -> last end %.1f us, longest block %.1f us\n", (tmax - tmin) / 100.0, dmax / 100.0);
  (void)hipFree(d); (void)hipFree(t); (void)hipFree(o); (void)hipStreamDestroy(s1); (void)hipStreamDestroy(s2);
}

int main() {
  concurrent(207, false, 0);
  concurrent(207, true, 0);
  concurrent(207, true, 1);
  concurrent(207, true, 0, 1, 512);
  concurrent(207, true, 0, 1, 64);
  concurrent(207, true, 0, 16, 64);
  concurrent(207, true, 0, 64, 64);
  concurrent(64, false, 0);
  concurrent(64, true, 0);
  concurrent(256, false, 0);
  concurrent(256, true, 0);
  {
    // cfg5's batch: valid encoder frames of the 16 utterances, padded to 743 frames each, 32-row blocks of the flattened rows
    const int fl[16] = {743, 596, 588, 550, 510, 444, 429, 424, 400, 387, 374, 318, 245, 172, 115, 70};
    const int Tp = 743, M = 16 * Tp, nb = (M + 31) / 32;
    std::vector<uint8_t> act(nb, 0);
    for (int b = 0; b < 16; ++b)
      for (int r = b * Tp; r < b * Tp + std::min(fl[b] + 4, Tp); ++r) act[r / 32] = 1;
    int n_act = 0;
    for (auto a : act) n_act += a;
    std::vector<uint8_t> dense(n_act, 1);
    ragged("cfg5 full-rate layer, padded grid + early exits", act, 133 * 1024);
    ragged("the same active blocks as a dense grid", dense, 133 * 1024);
    ragged("cfg5 full-rate layer, padded grid + early exits", act, 133 * 1024, true);
    ragged("the same active blocks as a dense grid", dense, 133 * 1024, true);
    std::vector<uint8_t> act16((M + 15) / 16, 0);
    for (int b = 0; b < 16; ++b)
      for (int r = b * Tp; r < b * Tp + std::min(fl[b] + 4, Tp); ++r) act16[r / 16] = 1;
    int n16 = 0;
    for (auto a : act16) n16 += a;
    std::vector<uint8_t> dense16(n16, 1);
    ragged("16-row blocks, padded grid + early exits", act16, 82 * 1024);
    ragged("16-row blocks, dense grid", dense16, 82 * 1024);
  }
  {
    std::vector<uint32_t> enc(8, 0xffffffffu); enc[0] = 0xffff0000u;
    std::vector<uint32_t> all(8, 0xffffffffu);
    for (int blocks : {207, 240, 256}) {
      busy("133 KB blocks, no mask", {}, blocks);
      busy("133 KB blocks, mask = all CUs", all, blocks);
      busy("133 KB blocks, all but bits 0..15", enc, blocks);
    }
  }
  run("no mask", {}, 1024);
  std::vector<uint32_t> m(8, 0);
  m[0] = 0xffff; run("bits 0..15", m, 256);
  m.assign(8, 0); m[0] = 0xff; run("bits 0..7", m, 256);
  m.assign(8, 0); m[0] = 0x1; run("bit 0", m, 64);
  m.assign(8, 0); m[0] = 0x100; run("bit 8", m, 64);
  m.assign(8, 0); m[1] = 0x1; run("bit 32", m, 64);
  m.assign(8, 0); for (int i = 0; i < 8; ++i) m[i] = 0x1; run("bits 0,32,64,..,224", m, 256);
  m.assign(8, 0); for (int i = 0; i < 8; ++i) m[i] = 0x10001; run("bits 0,16,32,..,240", m, 256);
  m.assign(8, 0xffffffffu); m[0] = 0xffff0000u; run("all but bits 0..15", m, 1024);
  m.assign(1, 0x3); run("1 word, bits 0,1", m, 256);
  return 0;
}
