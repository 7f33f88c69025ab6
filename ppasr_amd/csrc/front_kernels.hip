// front_kernels.hip -- the front end of the *former encoders and the streamed-A GEMM it is built on: positional-table
// folding (create time), CMVN + conv1, conv2 / conv3 as implicit GEMMs (k_gemm_stream<.., Conv2Src>), the input projection
// and the plain dense layers (DenseSrc), their fp16 x3 instantiations, the active-tile / active-block tables of ragged
// batches.  (Split from conformer_kernels.hip in round 5; the one-launch conv1 + conv2 is front_fused.hip.)
// Reference: ppasr/model_utils/conformer/{subsampling,embedding}.py, model_utils/utils/cmvn.py (file:line per kernel).
#include <cstdlib>

#include "conformer_kernels.h"
#include "launch.h"
#include "phases.h"
#include "h3.h"

#include <math.h>

namespace ppasr {
#ifdef PPASR_PHASE_TS
}  // namespace ppasr
extern "C" __attribute__((visibility("default"))) int ppasr_debug_read_wave_ts_front(long long* out) {  // tools/phase_ts.py: conv2's per-wave stamps
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ppasr::g_wave_ts), sizeof(long long) * 512);
}
namespace ppasr {
#endif

// =====================================================================================
// create-time: ptab[pos][n] = sum_k pe[pos][k] * Wpos[k][n]   (attention.py:234, bias-free)
// weight-only constant folding; not on the timed path, so a plain fmaf kernel.
// =====================================================================================
__global__ void k_posproj(const float* __restrict__ pe, const float* __restrict__ wpos,
                          const float* __restrict__ bpos, float* __restrict__ ptab, int max_len, int d) {
  int pos = blockIdx.x;
  int n = threadIdx.x;
  __shared__ float row[1024];
  row[n] = pe[(size_t)pos * d + n];
  __syncthreads();
  float acc = 0.f;
  for (int k = 0; k < d; ++k) acc = fmaf(row[k], wpos[k * d + n], acc);
  if (bpos) acc += bpos[n];  // Squeezeformer / Efficient-Conformer linear_pos has a bias
  ptab[(size_t)pos * d + n] = acc;
}
void launch_posproj(const float* pe, const float* wpos, const float* bpos, float* ptab, int max_len, hipStream_t st, int d) {
  PPASR_LAUNCH(k_posproj, dim3(max_len), dim3(d), 0, st, pe, wpos, bpos, ptab, max_len, d);
}

// =====================================================================================
// conv1: GlobalCMVN (utils/cmvn.py:29-31) + Conv2D(1->256, 3x3, s2) + ReLU
// (conformer/subsampling.py:84-86).  Output NHWC [B][T1][F1][256] so that the implicit-GEMM
// A rows of conv2 are contiguous 1 KiB runs.  One block per (t1, b); thread = channel.
// =====================================================================================
__global__ __launch_bounds__(256) void k_conv1(const float* __restrict__ feats, FrontW fw, float* __restrict__ y1,
                                               int T, int F, int T1, int F1, PadSkip ps) {
  __shared__ float xs[3][128];
  const int b = blockIdx.y, t1 = blockIdx.x, tid = threadIdx.x;
  const int C = 256 * gridDim.z;  // channels (256; the general route: a multiple)
  if (ps.lens && t1 > 2 * pad_need_steps(ps, b)) return;  // conv2 output frame t' reads conv1 frames 2t' .. 2t'+2
  for (int idx = tid; idx < 3 * F; idx += 256) {
    int i = idx / F, f = idx - i * F;
    float v = feats[((size_t)b * T + 2 * t1 + i) * F + f];
    xs[i][f] = (v - fw.cmvn_mean[f]) * fw.cmvn_istd[f];
  }
  __syncthreads();
  // thread = (channel quad cq, f1 phase fp): 16-byte stores, 1 KiB contiguous per wave (one channel per thread and
  // dword stores reached 5.0 TB/s of the 638 MB this kernel writes per 32 x 10 s batch)
  const int cq = tid & 63, fp = tid >> 6;
  const int c4 = 256 * blockIdx.z + 4 * cq;
  f32x4 w[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) w[j] = *reinterpret_cast<const f32x4*>(fw.conv1_w + j * C + c4);
  const f32x4 bias = *reinterpret_cast<const f32x4*>(fw.conv1_b + c4);
  float* out = y1 + ((size_t)(b * T1 + t1) * F1) * C + c4;
  for (int f1 = fp; f1 < F1; f1 += 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float xv = xs[i][2 * f1 + j];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(w[i * 3 + j][e], xv, acc[e]);
      }
    acc += bias;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.f);
    *reinterpret_cast<f32x4*>(out + (size_t)f1 * C) = acc;
  }
}
void launch_conv1(const float* feats, const FrontW& fw, float* y1, int B, int T, int F, int T1, int F1, hipStream_t st,
                  const PadSkip& ps, int channels) {
  PPASR_LAUNCH(k_conv1, dim3(T1, B, channels / 256), dim3(256), 0, st, feats, fw, y1, T, F, T1, F1, ps);
}

// =====================================================================================
// Streamed-A GEMM: out[M][256] = act(A[M][K] * W + b) * scale, A rows gathered from global
// in KC-wide chunks through a double-buffered LDS tile, W streamed in fragment order.
//   conv2  (subsampling.py:87-88): implicit GEMM, K = (kh,kw,cin) = 2304, ReLU, MT=4 (128 rows)
//   embed  (subsampling.py:89,113 + embedding.py:112): K = f2*256, *sqrt(d), MT=1
// =====================================================================================
struct Conv2Src {
  const float* y1;
  int T1, F1, Tp, F2;
  int k = 3, s = 2;  // kernel size / stride (3, 2: Conv2dSubsampling4 / 8; 5, 3: the second conv of Conv2dSubsampling6)
  int C = 256;       // input channels (NHWC)
  __device__ __forceinline__ const float* base(int m) const {
    int f2 = m % F2;
    int bt = m / F2;
    int tp = bt % Tp;
    int b = bt / Tp;
    return y1 + ((size_t)((b * T1 + s * tp) * F1 + s * f2)) * C;
  }
  // KC = 128: chunk kc -> tap kc / (C/128) (kh,kw), 128-channel slice kc % (C/128)
  __device__ __forceinline__ size_t chunk_off(int kc) const {
    const int cpt = C >> 7;
    int tap = kc / cpt, part = kc - tap * cpt;
    int kh = tap / k, kw = tap - k * kh;
    return ((size_t)(kh * F1 + kw)) * C + part * 128;
  }
};
struct DenseSrc {
  const float* a;
  int K, KC;
  __device__ __forceinline__ const float* base(int m) const { return a + (size_t)m * K; }
  __device__ __forceinline__ size_t chunk_off(int kc) const { return (size_t)kc * KC; }
};

// H3: the A chunks are staged as fp16 operand planes and the units run on the fp16 x3 route (h3.h; wp is then the
// re-packed weight)
template <int MT, int KC, bool RELU, bool SB, typename Src, bool H3>
__device__ __forceinline__ void gemm_stream_body(const Src& src, const f32x4* __restrict__ wp, const float* __restrict__ bias,
                                                 float* __restrict__ out, int M, int n_chunks, float scale, int ldc,
                                                 int n_valid, int m0, const PadSkip& ps, const int* __restrict__ tile_tab) {
  constexpr int BM = 32 * MT;
  // Ragged batch with a tile table (k_tile_prefix): workgroup t takes the t-th ACTIVE tile -- tiles are cut per utterance
  // (utterance b: rows b*S + [BM i, BM i + BM) for i < ceil(need rows / BM)), so the active tiles are the first `total`
  // workgroups of the grid and are dealt evenly to the 8 XCDs.  (With the padded row space tiled directly and the tiles
  // behind an utterance's valid frames exiting at once, an XCD that happens to be dealt 129 active tiles for its 32 CUs
  // runs five rounds where four would do: cfg5's conv2 took 1.5 ms against 1.1.)  Every row is computed by the same
  // arithmetic whichever tile it lands in.
  int r0_map = 0, Mlim = M;
  if (tile_tab) {
    const int t = blockIdx.x, nb = tile_tab[0];
    const int* pre = tile_tab + 1;  // pre[b] = active tiles in front of utterance b; pre[nb] = their total
    if (t >= pre[nb]) return;
    int lo = 0, hi = nb;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pre[mid] <= t) lo = mid;
      else hi = mid;
    }
    const int S = ps.Tp * ps.unit;
    r0_map = lo * S + (t - pre[lo]) * BM;
    Mlim = min(M, (lo + 1) * S);
  } else if (ps.tab) {  // list of the active BM-row blocks (rowblock.h PadSkip::tab; whole-matrix launches: m0 = 0)
    const int blk = pad_block_of(ps, BM, M);
    if (blk < 0) return;
    r0_map = blk * BM;
  } else if (pad_block_skippable(ps, m0 + blockIdx.x * BM, BM, M)) {
    return;
  }
  constexpr int LD = H3 ? (KC + 8) / 2 : KC + 4;  // floats per row of a chunk buffer (H3: one fp16 plane row of KC + 8)
  constexpr int LDH = KC + 8, PLANE = BM * LDH;   // fp16 plane geometry (H3)
  constexpr int BUF = H3 ? PLANE : BM * LD;       // floats per chunk buffer (H3: two planes of PLANE fp16 = PLANE floats)
  constexpr int F4_PER_ROW = KC / 4;
  constexpr int NL = BM * F4_PER_ROW / kThreads;  // float4 loads per thread per chunk
  constexpr int G = KC / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  const int r0 = (tile_tab || ps.tab) ? r0_map : m0 + blockIdx.x * BM;  // m0: first row of this launch (row ranges split across launches)
  const int tile_stride = n_chunks * G * 64;
  // gridDim.z > 1 (under-filled launches): workgroup z contracts K chunks [kc0, kc1) only and stores its raw partial sums
  // to out + z * M * ldc; k_gemm_join adds them up and applies bias / scale / activation
  const int kc0 = (int)((long long)blockIdx.z * n_chunks / gridDim.z), kc1 = (int)((long long)(blockIdx.z + 1) * n_chunks / gridDim.z);
  const f32x4* wbase = wp + (size_t)(blockIdx.y * kWaves + wave) * tile_stride;  // blockIdx.y = 256-column block
  BRing<1> ring;
  ring_prime(ring, wbase + (size_t)kc0 * G * 64, 0);
  // A-tile rows through buffer loads: per-lane byte offsets relative to the tile's first row (computed once), the K
  // chunk as the wave-uniform soffset -- the per-chunk request is then 8 VMEM instructions and NO vector ALU work.
  // (With 64-bit per-lane addresses every chunk started with 16 v_add per lane; the younger waves of each SIMD sat
  //  in those for 3 - 7 us while their older partners' MFMA streams had the issue port -- tools/phase_ts.py stamps --
  //  and the workgroup then ran its two wave sets one after the other.)  Rows >= M read as zeros (offset out of range).
  const float* tile_base = src.base(min(r0, M - 1));
  const __amdgpu_buffer_rsrc_t rs_a = wstream_rsrc(tile_base);
  int voff[NL];
  int lds_off[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int idx = tid + kThreads * i;
    int row = idx / F4_PER_ROW, c4 = idx - row * F4_PER_ROW;
    int m = r0 + row;
    voff[i] = (m < Mlim) ? (int)((src.base(m) - tile_base) * sizeof(float)) + 16 * c4 : 0x7fffffff;
    lds_off[i] = H3 ? row * LDH + 4 * c4 : row * LD + 4 * c4;  // (H3: fp16 elements inside a plane)
  }
  f32x4 stg[NL];
  auto load_chunk = [&](int kc) {
    const int soff = (int)(src.chunk_off(kc) * sizeof(float));
#pragma unroll
    for (int i = 0; i < NL; ++i) stg[i] = wstream_load(rs_a, voff[i], soff);
  };
  bool bad = false;  // fp16 x3 range-guard events of this workgroup's loaders (h3.h)
  auto write_piece = [&](float* buf, int i) {  // (H3) f32x4 number i of the staged chunk -> the two operand planes
    _Float16* pl = reinterpret_cast<_Float16*>(buf);
    f16x4 hi, lo;
    h3_split4(stg[i] * kH3Sa, hi, lo, bad);
    *reinterpret_cast<f16x4*>(pl + lds_off[i]) = hi;
    *reinterpret_cast<f16x4*>(pl + PLANE + lds_off[i]) = lo;
  };
  auto write_chunk = [&](float* buf) {
    if constexpr (H3) {
#pragma unroll
      for (int i = 0; i < NL; ++i) write_piece(buf, i);
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) *reinterpret_cast<f32x4*>(buf + lds_off[i]) = stg[i];
    }
  };
  f32x16 acc[MT][1];
  acc_zero(acc);
  load_chunk(kc0);
  write_chunk(smem + (kc0 & 1) * BUF);
  __syncthreads();
  for (int kc = kc0; kc < kc1; ++kc) {
    float* cur = smem + (kc & 1) * BUF;
    float* nxt = smem + ((kc + 1) & 1) * BUF;
    const bool more = kc + 1 < kc1;
    if (more) load_chunk(kc + 1);
    const f32x4* seg = wbase + (size_t)kc * G * 64;
    if (MT == 4 && kc < 8) PPASR_WAVE_TS(32 + 4 * kc);
    if constexpr (H3) {
      // the next chunk's split + plane stores ride inside this chunk's MFMA stream, NL / (KC / 16) pieces per k step (its
      // rows were requested above, before the stream's weight fragments: vmcnt retires in order, so the first fragment
      // wait covers them; the buffer they go to was last read in iteration kc - 1).  After the unit, as on the fp32
      // route, the split was 0.9 - 1.8 us of every 5.6 us chunk with the matrix pipe idle (tools/phase_ts.py --h3)
      constexpr int KS = KC / 16, PER = (NL + KS - 1) / KS;
      auto side = [&](int ks) {
        if (more) {
#pragma unroll
          for (int j = 0; j < PER; ++j)
            if (ks * PER + j < NL) write_piece(nxt, ks * PER + j);
        }
      };
      rb_gemm_h3_rows<MT, KS>(reinterpret_cast<const _Float16*>(cur), LDH, PLANE, seg, more ? seg + G * 64 : nullptr, ring, acc,
                              side);
    } else {
      rb_gemm<MT, 1, G>(cur, LD, seg, 0, more ? seg + G * 64 : nullptr, 0, ring, acc);
    }
    if (MT == 4 && kc < 8) PPASR_WAVE_TS(33 + 4 * kc);
    if constexpr (!H3) {
      if (more) write_chunk(nxt);
    }
    if (MT == 4 && kc < 8) PPASR_WAVE_TS(34 + 4 * kc);
    __syncthreads();
    if (MT == 4 && kc < 8) PPASR_WAVE_TS(35 + 4 * kc);
  }
  if constexpr (H3) {
    h3_note(bad);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] *= kH3Inv;
  }
  const int col = blockIdx.y * 256 + wave * 32 + (lane & 31);
  if (gridDim.z > 1) {
    float* po = out + (size_t)blockIdx.z * M * ldc;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = r0 + mt * 32 + acc_row(r, lane);
        if (m < Mlim && col < n_valid) po[(size_t)m * ldc + col] = acc[mt][0][r];
      }
    return;
  }
  const float bv = bias[col];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int m = r0 + mt * 32 + acc_row(r, lane);
      float v = SB ? acc[mt][0][r] * scale + bv : (acc[mt][0][r] + bv) * scale;
      if (RELU) v = fmaxf(v, 0.f);
      if (m < Mlim && col < n_valid) out[(size_t)m * ldc + col] = v;
    }
}
template <int MT, int KC, bool RELU, bool SB, typename Src>
__global__ __launch_bounds__(kThreads) void k_gemm_stream(Src src, const f32x4* __restrict__ wp,
                                                          const float* __restrict__ bias, float* __restrict__ out, int M,
                                                          int n_chunks, float scale, int ldc, int n_valid, int m0,
                                                          PadSkip ps, const int* __restrict__ tile_tab) {
  gemm_stream_body<MT, KC, RELU, SB, Src, false>(src, wp, bias, out, M, n_chunks, scale, ldc, n_valid, m0, ps, tile_tab);
}
// the convolution stage (conv2's implicit GEMM + ReLU) on the fp16 x3 route
template <int MT>
__global__ __launch_bounds__(kThreads) void k_conv_stage_h3(Conv2Src src, const f32x4* __restrict__ wp,
                                                            const float* __restrict__ bias, float* __restrict__ out, int M,
                                                            int n_chunks, float scale, int ldc, int n_valid, int m0, PadSkip ps,
                                                            const int* __restrict__ tile_tab) {
  gemm_stream_body<MT, 128, true, false, Conv2Src, true>(src, wp, bias, out, M, n_chunks, scale, ldc, n_valid, m0, ps, tile_tab);
}
// the input projection behind the front end (embed GEMM, K = F2 * 256) on the fp16 x3 route
template <bool SB>
__global__ __launch_bounds__(kThreads) void k_embed_h3(DenseSrc src, const f32x4* __restrict__ wp, const float* __restrict__ bias,
                                                       float* __restrict__ out, int M, int n_chunks, float scale, int ldc,
                                                       int n_valid, int m0, PadSkip ps, const int* __restrict__ tile_tab) {
  gemm_stream_body<1, 256, false, SB, DenseSrc, true>(src, wp, bias, out, M, n_chunks, scale, ldc, n_valid, m0, ps, tile_tab);
}
// tab[0] = B, tab[1 + b] = number of BM-row tiles the utterances in front of b need (rows b*S + [0, need(b) * unit)),
// tab[1 + B] = their total: the tile table of a ragged k_gemm_stream launch
__global__ __launch_bounds__(256) void k_tile_prefix(PadSkip ps, int B, int BM, int* __restrict__ tab) {
  __shared__ int cnt[256];
  int run = 0;
  if (threadIdx.x == 0) {
    tab[0] = B;
    tab[1] = 0;
  }
  for (int b0 = 0; b0 < B; b0 += 256) {
    const int b = b0 + threadIdx.x;
    cnt[threadIdx.x] = b < B ? (pad_need_steps(ps, b) * ps.unit + BM - 1) / BM : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 0; i < 256 && b0 + i < B; ++i) {
        run += cnt[i];
        tab[2 + b0 + i] = run;
      }
    }
    __syncthreads();
  }
}
// tab[0] = number of R-row blocks of the flattened [M] rows that hold a row some valid output frame depends on,
// tab[1 + i] = index of the i-th such block (ascending): PadSkip::tab of the ragged row-block launches
__global__ __launch_bounds__(256) void k_block_table(PadSkip ps, int M, int R, int* __restrict__ tab) {
  __shared__ int cnt[256];
  const int nblk = (M + R - 1) / R, tid = threadIdx.x;
  int run = 0;
  for (int b0 = 0; b0 < nblk; b0 += 256) {
    const int i = b0 + tid;
    const int act = (i < nblk && !pad_block_skippable(ps, i * R, R, M)) ? 1 : 0;
    cnt[tid] = act;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // inclusive scan
      const int v = tid >= o ? cnt[tid - o] : 0;
      __syncthreads();
      cnt[tid] += v;
      __syncthreads();
    }
    if (act) tab[1 + run + cnt[tid] - 1] = i;
    run += cnt[255];
    __syncthreads();
  }
  if (tid == 0) tab[0] = run;
}
void launch_block_table(const PadSkip& ps, int M, int R, int* tab, hipStream_t st) {
  PadSkip p = ps;
  p.tab = nullptr;
  PPASR_LAUNCH(k_block_table, dim3(1), dim3(256), 0, st, p, M, R, tab);
}
void launch_tile_prefix(const PadSkip& ps, int B, int BM, int* tab, hipStream_t st) {
  PPASR_LAUNCH(k_tile_prefix, dim3(1), dim3(256), 0, st, ps, B, BM, tab);
}
// out[m][c] = (sum_z part[z][m][c] + bias[c]) * scale   or   sum * scale + bias (scale_before_bias); one float4 per thread
__global__ __launch_bounds__(256) void k_gemm_join(const float* __restrict__ part, int nz, const float* __restrict__ bias,
                                                   float scale, int scale_before_bias, float* __restrict__ out, int M,
                                                   PadSkip ps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  if (pad_block_skippable(ps, row & ~(kRows - 1), kRows, M)) return;
  f32x4 acc = *reinterpret_cast<const f32x4*>(part + (size_t)row * kD + 4 * lane);
  for (int z = 1; z < nz; ++z) acc += *reinterpret_cast<const f32x4*>(part + ((size_t)z * M + row) * kD + 4 * lane);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * lane);
  f32x4 y;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    y[e] = (scale_before_bias & 1) ? acc[e] * scale + bv[e] : (acc[e] + bv[e]) * scale;
    if (scale_before_bias & 2) y[e] = fmaxf(y[e], 0.f);  // (bit 1: ReLU -- the convolution stages)
  }
  *reinterpret_cast<f32x4*>(out + (size_t)row * kD + 4 * lane) = y;
}
constexpr int kConvSplitMax = 9;  // K slices of an under-filled convolution stage (3 x 3 taps x 2 chunks = 18 chunks: 2 each)
size_t conv_stage_part_floats(int M, int channels) {
  return (channels == 256 && M <= 32 * 28) ? (size_t)kConvSplitMax * M * 256 : 0;
}

void launch_conv2(const float* y1, const FrontW& fw, float* y2, int B, int T1, int F1, int Tp, int F2, hipStream_t st,
                  const PadSkip& ps_frames, int* tile_scratch, const f32x4* w_h3, float* part, size_t part_floats) {
  launch_conv_stage(y1, w_h3 ? w_h3 : fw.conv2_w, fw.conv2_b, y2, B, T1, F1, Tp, F2, fw.conv2_k, fw.conv2_s, st, ps_frames, 256,
                    tile_scratch, w_h3 != nullptr, part, part_floats);
}
void launch_conv_stage(const float* y1, const f32x4* conv_w, const float* conv_b, float* y2, int B, int T1, int F1, int Tp,
                       int F2, int ksz, int stride, hipStream_t st, const PadSkip& ps_frames, int channels, int* tile_scratch,
                       bool h3, float* part, size_t part_floats) {
  Conv2Src src{y1, T1, F1, Tp, F2, ksz, stride, channels};
  const int n_kc = ksz * ksz * (channels / 128);  // 128-wide K chunks: channels / 128 per tap
  const int ny = channels / 256;                  // 256-column blocks of the output
  PadSkip ps = ps_frames;
  ps.unit = F2;  // rows are (frame, f2) pairs
  const int M = B * Tp * F2;
  constexpr int KC = 128, kCUs = 256;
  auto lds_of = [h3](int mt) {
    return h3 ? (size_t)2 * 2 * (32 * mt) * (KC + 8) * sizeof(_Float16) : (size_t)2 * (32 * mt) * (KC + 4) * sizeof(float);
  };
  const int* no_tab = nullptr;
  // (conv_w: the fp16 x3 re-packing when h3.  One macro per launch site below picks the kernel.)
#define CONV_STAGE_LAUNCH(MTX, GRID, M0, TAB)                                                                            \
  do {                                                                                                                    \
    if (h3)                                                                                                               \
      PPASR_LAUNCH((k_conv_stage_h3<MTX>), GRID, dim3(kThreads), lds_of(MTX), st, src, conv_w, conv_b, y2, M, n_kc, 1.0f, \
                   channels, channels, M0, ps, TAB);                                                                      \
    else                                                                                                                  \
      PPASR_LAUNCH((k_gemm_stream<MTX, KC, true, false, Conv2Src>), GRID, dim3(kThreads), lds_of(MTX), st, src, conv_w,   \
                   conv_b, y2, M, n_kc, 1.0f, channels, channels, M0, ps, TAB);                                           \
  } while (0)
  if (ps.lens && tile_scratch && M > 128 * kCUs) {
    // ragged batch, more than one round of 128-row tiles: the active tiles in front of the grid (see k_gemm_stream)
    PPASR_LAUNCH(k_tile_prefix, dim3(1), dim3(256), 0, st, ps, B, 128, tile_scratch);
    const int per_utt = (Tp * F2 + 127) / 128;
    CONV_STAGE_LAUNCH(4, dim3(B * per_utt, ny), 0, (const int*)tile_scratch);
    return;
  }
  // Wave quantisation: 128-row tiles over 256 CUs (one workgroup per CU at this LDS footprint) would run
  // ceil(tiles / 256) rounds, the last one mostly empty (1183 tiles = 4.62 rounds for 32 x 10 s).  The whole rounds
  // run with 128-row tiles; the remainder is re-cut into <= 256 tiles of 32 / 64 / 96 rows (one shorter round).
  const int tiles4 = (M + 127) / 128;
  const int full = (tiles4 / kCUs) * kCUs;
  const int rem_rows = M - full * 128;
  int mt_rem = (rem_rows + 32 * kCUs - 1) / (32 * kCUs);  // rows per remainder tile / 32
  if (full == 0) {
    // less than one round of 128-row tiles (a single utterance, a streaming chunk): smaller tiles fill more CUs
    const int mt = (M + 32 * kCUs - 1) / (32 * kCUs);  // 1 .. 4
    // a streaming chunk (16 frames x 19 bins = 10 tiles): each workgroup would walk all 18 K chunks alone, 74 us of a 1 ms
    // chunk -- with scratch the chunks go over up to 9 workgroups per tile (raw partial sums; k_gemm_join adds bias + ReLU)
    const int tiles = (M + 31) / 32;
    if (part && !h3 && ny == 1 && !ps.lens && tiles <= 28) {
      int z = kConvSplitMax;
      while (z > 1 && (n_kc % z != 0 || tiles * z > kCUs || (size_t)z * M * 256 > part_floats)) --z;
      if (z > 1) {
        PPASR_LAUNCH((k_gemm_stream<1, KC, true, false, Conv2Src>), dim3(tiles, 1, z), dim3(kThreads), lds_of(1), st, src, conv_w,
                     conv_b, part, M, n_kc, 1.0f, channels, channels, 0, ps, no_tab);
        PPASR_LAUNCH(k_gemm_join, dim3((M + 3) / 4), dim3(256), 0, st, part, z, conv_b, 1.0f, 2, y2, M, PadSkip{});
        return;
      }
    }
#define CONV2_ALL(MTA) CONV_STAGE_LAUNCH(MTA, dim3((M + 32 * MTA - 1) / (32 * MTA), ny), 0, no_tab)
    if (mt <= 1) CONV2_ALL(1);
    else if (mt == 2) CONV2_ALL(2);
    else if (mt == 3) CONV2_ALL(3);
    else CONV2_ALL(4);
#undef CONV2_ALL
    return;
  }
  if (rem_rows <= 0 || mt_rem >= 4) {
    CONV_STAGE_LAUNCH(4, dim3(tiles4, ny), 0, no_tab);
    return;
  }
  CONV_STAGE_LAUNCH(4, dim3(full, ny), 0, no_tab);
  const int m0 = full * 128;
#define CONV2_REM(MTR) CONV_STAGE_LAUNCH(MTR, dim3((rem_rows + 32 * MTR - 1) / (32 * MTR), ny), m0, no_tab)
  if (mt_rem <= 1) CONV2_REM(1);
  else if (mt_rem == 2) CONV2_REM(2);
  else CONV2_REM(3);
#undef CONV2_REM
#undef CONV_STAGE_LAUNCH
}
// Ragged launches (PadSkip) of kernels whose LDS footprint lets two or more workgroups share a CU: the whole grid is
// resident at once, the workgroups of skipped row blocks exit immediately, and the ACTIVE ones are left wherever they
// were placed -- two on some CUs, none on others (cfg5: 208 active of 375 row blocks: the CTC head ran 307 us where one
// block per CU takes ~ 140).  Asking for more than half of the LDS makes the workgroups exclusive: 256 are placed, a
// skipped one frees its CU for the next, and the active blocks end up one per CU.
size_t ragged_lds(size_t lds, const PadSkip& ps, int n_blocks) {
  return (ps.lens && n_blocks > 256 && lds < kLdsExclusive) ? kLdsExclusive : lds;
}

void launch_embed(const float* y2, const FrontW& fw, float* x0, int M, int K, float xscale, bool scale_before_bias,
                  hipStream_t st, const PadSkip& ps, int k_slices, float* part, const f32x4* w_h3) {
  constexpr int MT = 1, KC = 256;
  DenseSrc src{y2, K, KC};
  size_t lds = ragged_lds(2 * (32 * MT) * (KC + 4) * sizeof(float), ps, (M + 31) / 32);
  if (w_h3 && !(k_slices > 1 && part)) {  // fp16 x3 route (full launches): two chunk buffers of two fp16 planes
    lds = ragged_lds((size_t)2 * 2 * 32 * (KC + 8) * sizeof(_Float16), ps, (M + 31) / 32);
    if (scale_before_bias)
      PPASR_LAUNCH(k_embed_h3<true>, dim3((M + 31) / 32), dim3(kThreads), lds, st, src, w_h3, fw.embed_b, x0, M, K / KC, xscale,
                   kD, kD, 0, ps, (const int*)nullptr);
    else
      PPASR_LAUNCH(k_embed_h3<false>, dim3((M + 31) / 32), dim3(kThreads), lds, st, src, w_h3, fw.embed_b, x0, M, K / KC, xscale,
                   kD, kD, 0, ps, (const int*)nullptr);
    return;
  }
  if (k_slices > 1 && part) {  // under-filled launch: the K = 4864 contraction over k_slices workgroups per row block
    PPASR_LAUNCH((k_gemm_stream<MT, KC, false, false, DenseSrc>), dim3((M + 31) / 32, 1, k_slices), dim3(kThreads), lds,
                       st, src, fw.embed_w, fw.embed_b, part, M, K / KC, xscale, kD, kD, 0, ps, (const int*)nullptr);
    PPASR_LAUNCH(k_gemm_join, dim3((M + 3) / 4), dim3(256), 0, st, part, k_slices, fw.embed_b, xscale,
                       scale_before_bias ? 1 : 0, x0, M, ps);
    return;
  }
  if (scale_before_bias)
    PPASR_LAUNCH((k_gemm_stream<MT, KC, false, true, DenseSrc>), dim3((M + 31) / 32), dim3(kThreads), lds, st, src,
                       fw.embed_w, fw.embed_b, x0, M, K / KC, xscale, kD, kD, 0, ps, (const int*)nullptr);
  else
    PPASR_LAUNCH((k_gemm_stream<MT, KC, false, false, DenseSrc>), dim3((M + 31) / 32), dim3(kThreads), lds, st, src,
                       fw.embed_w, fw.embed_b, x0, M, K / KC, xscale, kD, kD, 0, ps, (const int*)nullptr);
}

// out[m][c] = (sum_z part[z][m][c] + bias[c]) * scale for c < n_valid: the join of launch_dense's K slices (any width)
__global__ __launch_bounds__(256) void k_dense_join(const float* __restrict__ part, int nz, const float* __restrict__ bias,
                                                    float scale, float* __restrict__ out, int M, int ldc, int n_valid) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = (n_valid + 3) / 4;
  if (i >= (size_t)M * per_row) return;
  const int m = (int)(i / per_row), c = 4 * (int)(i - (size_t)m * per_row);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < nz; ++z)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < n_valid) acc[e] += part[((size_t)z * M + m) * ldc + c + e];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (c + e < n_valid) out[(size_t)m * ldc + c + e] = (acc[e] + bias[c + e]) * scale;
}

// out[M][ldc] (columns < n_valid) = A[M][K] * Wpacked + bias ; K % 256 == 0 ; weights / bias padded to a multiple of
// 256 columns.  Used by the DeepSpeech2 path (LSTM input projections, CTC head) and the general layer route.
// Under-filled launches (few rows: one utterance): with a scratch buffer `part` of >= k_slices * M * ldc floats the K
// contraction is cut over up to 8 workgroups per tile (partial sums joined by k_dense_join), like the embed GEMM.
void launch_dense(const float* a, int lda, const f32x4* w, const float* bias, float* out, int M, int K, int n_cols_padded,
                  int ldc, int n_valid, hipStream_t st, float scale, float* part, size_t part_floats) {
  constexpr int MT = 1, KC = 256;
  DenseSrc src{a, lda, KC};
  size_t lds = 2 * (32 * MT) * (KC + 4) * sizeof(float);
  const int tiles = ((M + 31) / 32) * (n_cols_padded / 256), n_kc = K / KC;
  int S = 1;
  if (part && tiles <= 128) {
    S = 8;
    while (S > 1 && (tiles * S > 256 || n_kc % S != 0 || (size_t)S * M * ldc > part_floats)) S >>= 1;
  }
  if (S > 1) {
    PPASR_LAUNCH((k_gemm_stream<MT, KC, false, false, DenseSrc>), dim3((M + 31) / 32, n_cols_padded / 256, S), dim3(kThreads),
                 lds, st, src, w, bias, part, M, n_kc, scale, ldc, n_valid, 0, PadSkip{}, (const int*)nullptr);
    const size_t n4 = (size_t)M * ((n_valid + 3) / 4);
    PPASR_LAUNCH(k_dense_join, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, part, S, bias, scale, out, M, ldc, n_valid);
    return;
  }
  // (128-row tiles -- conv2's shape -- were measured for the big DeepSpeech2 GEMMs and are slower, 0.68 against 0.78 of
  //  the peak at M = 15 872: one 135 KB workgroup per CU, eight K chunks per tile and 64 dword stores per lane leave
  //  prologue and epilogue uncovered, where two 32-row workgroups per CU cover each other's)
  PPASR_LAUNCH((k_gemm_stream<MT, KC, false, false, DenseSrc>), dim3((M + 31) / 32, n_cols_padded / 256),
                     dim3(kThreads), lds, st, src, w, bias, out, M, K / KC, scale, ldc, n_valid, 0, PadSkip{}, (const int*)nullptr);
}

unsigned int* front_h3_ovf_counter() { return h3_ovf_counter(); }

hipError_t configure_front_kernels() {
  hipError_t e = hipSuccess;
#define SET_LDS(fn, bytes)                                                                                     \
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
  if (e != hipSuccess) return e;
  SET_LDS(k_conv_stage_h3<4>, 2 * 2 * 128 * 136 * sizeof(_Float16));
  SET_LDS(k_conv_stage_h3<3>, 2 * 2 * 96 * 136 * sizeof(_Float16));
  SET_LDS(k_conv_stage_h3<2>, 2 * 2 * 64 * 136 * sizeof(_Float16));
  SET_LDS(k_conv_stage_h3<1>, 2 * 2 * 32 * 136 * sizeof(_Float16));
  SET_LDS((k_gemm_stream<4, 128, true, false, Conv2Src>), 2 * 128 * 132 * sizeof(float));
  SET_LDS((k_gemm_stream<3, 128, true, false, Conv2Src>), 2 * 96 * 132 * sizeof(float));
  SET_LDS((k_gemm_stream<2, 128, true, false, Conv2Src>), 2 * 64 * 132 * sizeof(float));
  SET_LDS((k_gemm_stream<1, 128, true, false, Conv2Src>), 2 * 32 * 132 * sizeof(float));
  SET_LDS(k_embed_h3<false>, kLdsExclusive);
  SET_LDS(k_embed_h3<true>, kLdsExclusive);
  SET_LDS((k_gemm_stream<1, 256, false, false, DenseSrc>), kLdsExclusive);
  SET_LDS((k_gemm_stream<1, 256, false, true, DenseSrc>), kLdsExclusive);
#undef SET_LDS
  return hipSuccess;
}

}  // namespace ppasr
