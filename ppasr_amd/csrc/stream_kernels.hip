// stream_kernels.hip -- cache bookkeeping of the streaming handles (capi_stream.hip): pointwise_conv1 + GLU of the cached
// conv-module frames, K / V append, conv-history update / gather (single sessions and session groups), export / import of
// the reference's att_cache / cnn_cache layouts.  (Split from conformer_kernels.hip in round 5.)
// Reference: ppasr/model_utils/conformer/encoder.py:208-283, convolution.py:108-126.
#include <cstdlib>

#include "conformer_kernels.h"
#include "launch.h"
#include "phases.h"
#include "h3.h"

#include <math.h>

namespace ppasr {

// streaming: g_hist = GLU(pointwise_conv1(cnn_cache rows))  -- the reference re-applies pointwise_conv1+GLU
// to the cached frames on every chunk (convolution.py:113,125-126); here once per chunk on <= 32 rows.
__global__ __launch_bounds__(kThreads) void k_pw1_glu(const float* __restrict__ xhat, float* __restrict__ g, LayerW w, int M) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* bufA = smem;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blockIdx.x * kRows;
  const int valid = min(kRows, M - r0);
  const int col = wave * 32 + (lane & 31);
  BRing<1> ring;
  const f32x4* seg_val = w.pw1 + (size_t)wave * kTs256;
  const f32x4* seg_gate = w.pw1 + (size_t)(8 + wave) * kTs256;
  ring_prime(ring, seg_val, 0);
  rb_load_rows(bufA, kLda, xhat + (size_t)r0 * kD, kRows, valid);
  __syncthreads();
  f32x16 av[1][1], ag[1][1];
  acc_zero(av);
  acc_zero(ag);
  rb_gemm<1, 1, kG256>(bufA, kLda, seg_val, 0, seg_gate, 0, ring, av);
  rb_gemm<1, 1, kG256>(bufA, kLda, seg_gate, 0, nullptr, 0, ring, ag);
  const float bval = w.pw1_b[col];
  const float bgate = w.pw1_b[kD + col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = acc_row(r, lane);
    if (row < valid) g[(size_t)(r0 + row) * kD + col] = (av[0][0][r] + bval) * sigmoidf(ag[0][0][r] + bgate);
  }
}
// the same for every layer's history in ONE launch (single-session streaming: the histories only depend on the previous
// chunk, so the twelve small launches need not sit between the layers): block i = layer i, tab[i] = its weights / rows
__global__ __launch_bounds__(kThreads) void k_pw1_glu_layers(const float* __restrict__ xh_hist, float* __restrict__ g_hist,
                                                             const HistLayer* __restrict__ tab, int lo_stride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* bufA = smem;
  const HistLayer t = tab[blockIdx.x];
  const float* xhat = xh_hist + (size_t)blockIdx.x * lo_stride * kD;
  float* g = g_hist + (size_t)blockIdx.x * lo_stride * kD;
  const int lane = lane_id(), wave = wave_id();
  const int valid = min(kRows, t.rows);
  const int col = wave * 32 + (lane & 31);
  BRing<1> ring;
  const f32x4* seg_val = t.pw1 + (size_t)wave * kTs256;
  const f32x4* seg_gate = t.pw1 + (size_t)(8 + wave) * kTs256;
  ring_prime(ring, seg_val, 0);
  rb_load_rows(bufA, kLda, xhat, kRows, valid);
  __syncthreads();
  f32x16 av[1][1], ag[1][1];
  acc_zero(av);
  acc_zero(ag);
  rb_gemm<1, 1, kG256>(bufA, kLda, seg_val, 0, seg_gate, 0, ring, av);
  rb_gemm<1, 1, kG256>(bufA, kLda, seg_gate, 0, nullptr, 0, ring, ag);
  const float bval = t.pw1_b[col];
  const float bgate = t.pw1_b[kD + col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = acc_row(r, lane);
    if (row < valid) g[(size_t)row * kD + col] = (av[0][0][r] + bval) * sigmoidf(ag[0][0][r] + bgate);
  }
}
constexpr size_t kLdsPw1Glu = kRows * kLda * sizeof(float);
void launch_pw1_glu_layers(const float* xh_hist, float* g_hist, const HistLayer* tab, int n_layers, int lo_stride,
                           hipStream_t st) {
  PPASR_LAUNCH(k_pw1_glu_layers, dim3(n_layers), dim3(kThreads), kLdsPw1Glu, st, xh_hist, g_hist, tab, lo_stride);
}
void launch_pw1_glu(const float* xhat, float* g, const LayerW& w, int M, hipStream_t st) {
  PPASR_LAUNCH(k_pw1_glu, dim3((M + kRows - 1) / kRows), dim3(kThreads), kLdsPw1Glu, st, xhat, g, w, M);
}

// streaming: append this chunk's keys / values (columns 256.. / 512.. of qkv) to the per-layer caches
__global__ void k_kv_append(const float* __restrict__ qkv, float* __restrict__ kc, float* __restrict__ vc, int n_rows) {
  const int row = blockIdx.x, t = threadIdx.x;  // 128 threads x float4 = 512 floats (k | v)
  const f32x4 v = *reinterpret_cast<const f32x4*>(qkv + (size_t)row * 768 + 256 + 4 * t);
  float* dst = (t < 64) ? kc + (size_t)row * kD + 4 * t : vc + (size_t)row * kD + 4 * (t - 64);
  *reinterpret_cast<f32x4*>(dst) = v;
}
void launch_kv_append(const float* qkv, float* kc, float* vc, int n_rows, hipStream_t st) {
  PPASR_LAUNCH(k_kv_append, dim3(n_rows), dim3(128), 0, st, qkv, kc, vc, n_rows);
}

// streaming: hist <- last `lo` rows of concat(hist[lo], fresh[n]); single block, read-all-then-write
__global__ __launch_bounds__(256) void k_hist_update(float* __restrict__ hist, const float* __restrict__ fresh, int n, int lo) {
  const int tid = threadIdx.x;
  constexpr int kMaxPer = 32;  // lo <= 30 rows of 64 float4 = 1920 float4 / 256 threads
  f32x4 tmp[kMaxPer / 4];
  const int total = lo * 64;
#pragma unroll
  for (int i = 0; i < kMaxPer / 4; ++i) {
    int idx = tid + 256 * i;
    if (idx < total) {
      int row = idx >> 6, c4 = idx & 63;
      int j = n + row;  // row index inside concat(hist, fresh)
      tmp[i] = (j < lo) ? *reinterpret_cast<const f32x4*>(hist + (size_t)j * kD + 4 * c4)
                        : *reinterpret_cast<const f32x4*>(fresh + (size_t)(j - lo) * kD + 4 * c4);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kMaxPer / 4; ++i) {
    int idx = tid + 256 * i;
    if (idx < total) *reinterpret_cast<f32x4*>(hist + (size_t)(idx >> 6) * kD + 4 * (idx & 63)) = tmp[i];
  }
}
// Attention caches of ALL layers trimmed in one launch (required_cache_size > 0: every chunk drops its oldest frames): rows
// [from, from + keep) of every layer's K and V cache [cap][D] move to its start.  Workgroup (layer, K | V) walks its rows in
// blocks of 16 -- read a block, barrier, write it: the destination lies below the source, so a later block's source is never
// a block already written.  (Two device-to-device copies through a scratch per cache: 48 launches of ~3.3 us per chunk.)
__global__ __launch_bounds__(256) void k_shift_caches(float* __restrict__ kc, float* __restrict__ vc, long long layer_stride, int D,
                                                      int from_full, int keep_full, int from_half, int keep_half,
                                                      unsigned long long half_mask) {
  const int layer = blockIdx.x;
  float* buf = (blockIdx.y == 0 ? kc : vc) + (size_t)layer * layer_stride;
  const bool half = (half_mask >> layer) & 1ull;
  const int from = half ? from_half : from_full, keep = half ? keep_half : keep_full;
  if (keep <= 0 || from <= 0) return;
  const int d4 = D / 4, per_blk = 16 * d4;  // f32x4 elements of a 16-row block (<= 16 per thread up to D = 1024)
  for (int r0 = 0; r0 < keep; r0 += 16) {
    const int n = min(16, keep - r0) * d4;
    f32x4 tmp[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = (int)threadIdx.x + 256 * i;
      if (idx < n && idx < per_blk) tmp[i] = *reinterpret_cast<const f32x4*>(buf + (size_t)(from + r0) * D + 4 * (size_t)idx);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = (int)threadIdx.x + 256 * i;
      if (idx < n && idx < per_blk) *reinterpret_cast<f32x4*>(buf + (size_t)r0 * D + 4 * (size_t)idx) = tmp[i];
    }
  }
}
void launch_shift_caches(float* kc, float* vc, long long layer_stride, int D, int n_layers, int from_full, int keep_full,
                         int from_half, int keep_half, unsigned long long half_mask, hipStream_t st) {
  PPASR_LAUNCH(k_shift_caches, dim3(n_layers, 2), dim3(256), 0, st, kc, vc, layer_stride, D, from_full, keep_full, from_half,
               keep_half, half_mask);
}
void launch_hist_update(float* hist, const float* fresh, int n, int lo, hipStream_t st) {
  PPASR_LAUNCH(k_hist_update, dim3(1), dim3(256), 0, st, hist, fresh, n, lo);
}

// ---- multi-session streaming helpers (one launch for all active sessions) ----
// keys / values of chunk row (b, t) -> cache row cache_t[b] + t of session sess[b]
__global__ void k_kv_append_group(const float* __restrict__ qkv, float* __restrict__ kc, float* __restrict__ vc,
                                  long long sess_stride, const SessDesc* __restrict__ sess, int c) {
  const int row = blockIdx.x, t = threadIdx.x;  // 128 threads x float4 = 512 floats (k | v)
  const int b = row / c, tt = row - b * c;
  const SessDesc d = sess[b];
  const size_t dst_row = (size_t)d.sess * sess_stride + (size_t)(d.cache_t + tt) * kD;
  const f32x4 v = *reinterpret_cast<const f32x4*>(qkv + (size_t)row * 768 + 256 + 4 * t);
  float* dst = (t < 64) ? kc + dst_row + 4 * t : vc + dst_row + 4 * (t - 64);
  *reinterpret_cast<f32x4*>(dst) = v;
}
void launch_kv_append_group(const float* qkv, float* kc, float* vc, long long sess_stride, const SessDesc* sess, int n, int c,
                            hipStream_t st) {
  PPASR_LAUNCH(k_kv_append_group, dim3(n * c), dim3(128), 0, st, qkv, kc, vc, sess_stride, sess, c);
}
// dst[b][lo][256] <- conv-module input history of session sess[b] (this layer)
__global__ void k_hist_gather(const float* __restrict__ hist, long long sess_stride, const SessDesc* __restrict__ sess,
                              float* __restrict__ dst, int lo) {
  const int b = blockIdx.x / lo, j = blockIdx.x - b * lo, t = threadIdx.x;  // 64 threads x float4
  *reinterpret_cast<f32x4*>(dst + ((size_t)b * lo + j) * kD + 4 * t) =
      *reinterpret_cast<const f32x4*>(hist + (size_t)sess[b].sess * sess_stride + (size_t)j * kD + 4 * t);
}
void launch_hist_gather(const float* hist, long long sess_stride, const SessDesc* sess, float* dst, int n, int lo,
                        hipStream_t st) {
  PPASR_LAUNCH(k_hist_gather, dim3(n * lo), dim3(64), 0, st, hist, sess_stride, sess, dst, lo);
}
// hist[sess[b]] <- last `lo` rows of concat(hist[sess[b]], fresh[b][c]); one 256-thread block per session
__global__ __launch_bounds__(256) void k_hist_update_group(float* __restrict__ hist, long long sess_stride,
                                                           const SessDesc* __restrict__ sess,
                                                           const float* __restrict__ fresh, int c, int lo) {
  const int b = blockIdx.x, tid = threadIdx.x;
  float* h = hist + (size_t)sess[b].sess * sess_stride;
  const float* f = fresh + (size_t)b * c * kD;
  constexpr int kMaxPer = 32;
  f32x4 tmp[kMaxPer / 4];
  const int total = lo * 64;
#pragma unroll
  for (int i = 0; i < kMaxPer / 4; ++i) {
    const int idx = tid + 256 * i;
    if (idx < total) {
      const int row = idx >> 6, c4 = idx & 63;
      const int j = c + row;  // row index inside concat(hist, fresh)
      tmp[i] = (j < lo) ? *reinterpret_cast<const f32x4*>(h + (size_t)j * kD + 4 * c4)
                        : *reinterpret_cast<const f32x4*>(f + (size_t)(j - lo) * kD + 4 * c4);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kMaxPer / 4; ++i) {
    const int idx = tid + 256 * i;
    if (idx < total) *reinterpret_cast<f32x4*>(h + (size_t)(idx >> 6) * kD + 4 * (idx & 63)) = tmp[i];
  }
}
void launch_hist_update_group(float* hist, long long sess_stride, const SessDesc* sess, const float* fresh, int n, int c,
                              int lo, hipStream_t st) {
  PPASR_LAUNCH(k_hist_update_group, dim3(n), dim3(256), 0, st, hist, sess_stride, sess, fresh, c, lo);
}

// [T][256] (col = h*64+f) k/v caches  <->  reference att_cache layout [h][T][2*dk]  (attention.py:232)
// `div` = 2 on time-reduced layers: the reference stores their cache repeat_interleave'd to the full rate and reads it
// back with [::2] (squeezeformer/encoder.py:355,367-369; efficient_conformer/encoder.py:349,368); ours holds each frame once.
__global__ void k_cache_export(const float* __restrict__ kc, const float* __restrict__ vc, float* __restrict__ att, int T,
                               int div) {
  const int t = blockIdx.x, tid = threadIdx.x, D = blockDim.x;  // D = heads * 64 threads: (h, f)
  const int h = tid >> 6, f = tid & 63;
  att[((size_t)h * T + t) * 128 + f] = kc[(size_t)(t / div) * D + tid];
  att[((size_t)h * T + t) * 128 + 64 + f] = vc[(size_t)(t / div) * D + tid];
}
__global__ void k_cache_import(const float* __restrict__ att, float* __restrict__ kc, float* __restrict__ vc, int T, int div) {
  const int j = blockIdx.x, tid = threadIdx.x, D = blockDim.x;  // j = stored frame <- exported frame j * div
  const int h = tid >> 6, f = tid & 63;
  kc[(size_t)j * D + tid] = att[((size_t)h * T + (size_t)j * div) * 128 + f];
  vc[(size_t)j * D + tid] = att[((size_t)h * T + (size_t)j * div) * 128 + 64 + f];
}
// cnn cache: ours [lo][256] (row = frame)  <->  reference [256][lo]
// `lo_ref` >= lo: width of the reference tensor; ours maps to its LAST lo columns, the rest is zero on export
// (F.pad to cnn_module_kernel-1, efficient_conformer/encoder.py:371-374; convolution.py:106 reads cache[:, :, -lorder:]).
__global__ void k_cnn_transpose(const float* __restrict__ src, float* __restrict__ dst, int lo, int lo_ref, int to_ref) {
  const int c = threadIdx.x, D = blockDim.x;
  const int skip = lo_ref - lo;
  if (to_ref)
    for (int j = 0; j < skip; ++j) dst[(size_t)c * lo_ref + j] = 0.f;
  for (int j = 0; j < lo; ++j) {
    if (to_ref) dst[(size_t)c * lo_ref + skip + j] = src[(size_t)j * D + c];
    else dst[(size_t)j * D + c] = src[(size_t)c * lo_ref + skip + j];
  }
}
void launch_cache_export(const float* kc, const float* vc, float* att, int T, int div, hipStream_t st, int D) {
  if (T > 0) PPASR_LAUNCH(k_cache_export, dim3(T), dim3(D), 0, st, kc, vc, att, T, div);
}
void launch_cache_import(const float* att, float* kc, float* vc, int T, int div, hipStream_t st, int D) {
  if (T > 0) PPASR_LAUNCH(k_cache_import, dim3((T + div - 1) / div), dim3(D), 0, st, att, kc, vc, T, div);
}
void launch_cnn_transpose(const float* src, float* dst, int lo, int lo_ref, int to_ref, hipStream_t st, int D) {
  PPASR_LAUNCH(k_cnn_transpose, dim3(1), dim3(D), 0, st, src, dst, lo, lo_ref, to_ref);
}
hipError_t configure_stream_kernels() {
  hipError_t e = hipSuccess;
#define SET_LDS(fn, bytes)                                                                                     \
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
  if (e != hipSuccess) return e;
  SET_LDS(k_pw1_glu, kLdsPw1Glu);
#undef SET_LDS
  return hipSuccess;
}

}  // namespace ppasr
