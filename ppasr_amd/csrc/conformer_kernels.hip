// conformer_kernels.hip -- gfx950 kernels of the Conformer encoder + CTC head hot path.
// Reference semantics: ppasr/model_utils/conformer/{encoder,attention,convolution,positionwise,
// subsampling,embedding}.py, model_utils/loss/ctc.py, decoders/ctc_greedy_decoder.py
// (file:line cited per kernel).  All arithmetic is fp32 (the reference's inference dtype).
#include <cstdlib>

#include "conformer_kernels.h"
#include "launch.h"
#include "phases.h"
#include "h3.h"

#include <math.h>

namespace ppasr {
#ifdef PPASR_PHASE_TS
}  // namespace ppasr
extern "C" __attribute__((visibility("default"))) int ppasr_debug_read_phase_ts(long long* out) {  // instrumented builds only (tools/phase_ts.py)
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ppasr::g_phase_ts), sizeof(long long) * 128);
}
extern "C" __attribute__((visibility("default"))) int ppasr_debug_read_wave_ts(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ppasr::g_wave_ts), sizeof(long long) * 512);
}
extern "C" __attribute__((visibility("default"))) int ppasr_debug_read_wg_ts(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ppasr::g_wg_ts), sizeof(long long) * 2 * 1024);
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
  for (int e = 0; e < 4; ++e) y[e] = scale_before_bias ? acc[e] * scale + bv[e] : (acc[e] + bv[e]) * scale;
  *reinterpret_cast<f32x4*>(out + (size_t)row * kD + 4 * lane) = y;
}

void launch_conv2(const float* y1, const FrontW& fw, float* y2, int B, int T1, int F1, int Tp, int F2, hipStream_t st,
                  const PadSkip& ps_frames, int* tile_scratch, const f32x4* w_h3) {
  launch_conv_stage(y1, w_h3 ? w_h3 : fw.conv2_w, fw.conv2_b, y2, B, T1, F1, Tp, F2, fw.conv2_k, fw.conv2_s, st, ps_frames, 256,
                    tile_scratch, w_h3 != nullptr);
}
void launch_conv_stage(const float* y1, const f32x4* conv_w, const float* conv_b, float* y2, int B, int T1, int F1, int Tp,
                       int F2, int ksz, int stride, hipStream_t st, const PadSkip& ps_frames, int channels, int* tile_scratch,
                       bool h3) {
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

// =====================================================================================
// Row-block phases shared by the per-layer kernels
// =====================================================================================

// -------------------------------------------------------------------------------------
// S1: x1 = x + 0.5*FFN_macaron(LN(x)) ; qkv = LN_mha(x1) * [Wq|Wk|Wv] + b
// (encoder.py:380-391, attention.py:75-77)
// -------------------------------------------------------------------------------------
// Epilogue slice of the previous Q / K / V tile inside the next unit's MFMA stream: qkv[row][col] = acc + bias
// (transposed tiles, rb_gemm SWAP: lane = row, register quad q = 4 consecutive columns -> one 16-byte store per quad,
//  issued during k-groups 4, 12, 20, 28 of the next unit)
// fp16 x3 units are a third as long as fp32 ones and gfx9's vmcnt retires in order: a store sliced into the middle of a
// unit makes the next wait for a weight fragment wait for the store's write acknowledge as well (tools/phase_ts.py --h3,
// round 5: Q unit 3.8 us, K unit with Q's four stores inside 6.6, V unit with K's 7.4).  The Q / K tiles are therefore
// stored during the LAST two k steps of the V unit, after the final weight load of the stream: nothing waits behind them.
struct QkStoreTailH3 {
  const f32x16& q_acc;
  const f32x16& k_acc;
  float* out;  // qkv + (r0 + this lane's row) * 768 + first column of this lane's quad 0; nullptr: row >= valid
  const f32x4 (&bias)[2][4];
  __device__ __forceinline__ void operator()(int ks) const {
    if (ks < 14 || !out) return;
    const int c = ks - 14;
    const f32x16& acc = c ? k_acc : q_acc;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<f32x4*>(out + c * 256 + 8 * q) =
          f32x4{acc[4 * q] * kH3Inv + bias[c][q][0], acc[4 * q + 1] * kH3Inv + bias[c][q][1],
                acc[4 * q + 2] * kH3Inv + bias[c][q][2], acc[4 * q + 3] * kH3Inv + bias[c][q][3]};
  }
};
struct QkvStoreSide {
  const f32x16& acc;
  float* out;  // qkv + (r0 + this lane's row) * 768 + first column of this lane's quad 0; nullptr: row >= valid
  const f32x4 (&bias)[4];
  __device__ __forceinline__ void operator()(int g) const {
    if ((g & 7) == 4 && out) {
      const int q = g >> 3;
      *reinterpret_cast<f32x4*>(out + 8 * q) = f32x4{acc[4 * q] + bias[q][0], acc[4 * q + 1] + bias[q][1],
                                                     acc[4 * q + 2] + bias[q][2], acc[4 * q + 3] + bias[q][3]};
    }
  }
};

// Body of S1 on LDS-resident rows (bufX = layer input): FFN_macaron, residual, LN_mha, QKV.
// `ring` must already stream w.ffm_w1 (tile `wave`).
// H3: the feed-forward module on the fp16 x3 route (h3.h; w.ffm_w1 / w.ffm_w2 are then the re-packed weights)
template <bool H3 = false>
__device__ __forceinline__ void ffn_qkv_body(float* bufX, float* bufA, float* bufH, float* __restrict__ x1,
                                             float* __restrict__ qkv, const LayerW& w, int r0, int valid, int n_chunks,
                                             BRing<1>& ring, VtOut vt = VtOut{}) {
  const int lane = lane_id(), wave = wave_id();
  rb_layernorm(bufX, bufA, kLda, kRows, w.ln_mac_g, w.ln_mac_b, 1e-5f);
  __syncthreads();
  PPASR_TS(8);
  f32x16 acc2[1][1];
  acc_zero(acc2);
  const f32x4* wq = w.wqkv + (size_t)wave * kTs256;
  if constexpr (H3) ffn_phase_h3(bufA, w.ffm_w1, w.ffm_b1, w.ffm_w2, n_chunks, wq, ring, acc2);
  else ffn_phase<true>(bufA, bufH, w.ffm_w1, w.ffm_b1, w.ffm_w2, n_chunks, wq, ring, acc2);
  PPASR_TS(9);
  residual_epilogue_t(bufX, acc2, w.ffm_b2, 0.5f);
  __syncthreads();
  PPASR_TS(10);
  rb_store_rows(x1 + (size_t)r0 * kD, bufX, kLda, kRows, valid);
  rb_layernorm(bufX, bufA, kLda, kRows, w.ln_mha_g, w.ln_mha_b, 1e-5f);
  __syncthreads();
  if constexpr (H3) h3_planes_from_tile(bufA, reinterpret_cast<_Float16*>(bufA));  // (in place; 512 B run on into bufH)
  PPASR_TS(11);
  // Q, K, V units: the global stores of unit c's tile (16 per lane) are sliced into the MFMA stream of unit c + 1
  // (QkvStoreSide, one store every second k-group) instead of running between the units with the matrix pipe idle
  // (0.4 - 1.5 us per unit, per-phase stamps); only V's tile is stored after its GEMM.
  f32x16 tile[3][1][1];
  const int cq = wave * 32 + 4 * (lane >> 5);
  float* qrow = (lane & 31) < valid ? qkv + (size_t)(r0 + (lane & 31)) * 768 + cq : nullptr;
  f32x4 qb[2][4];  // biases of this lane's Q / K column quads
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) qb[c][q] = *reinterpret_cast<const f32x4*>(w.bqkv + c * 256 + cq + 8 * q);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    acc_zero(tile[c]);
    const f32x4* seg = w.wqkv + (size_t)(c * 8 + wave) * kTs256;
    const f32x4* nseg = c < 2 ? seg + 8 * kTs256 : nullptr;
    if constexpr (H3) {  // (w.wqkv: the re-packed weight; the LayerNorm'd rows were turned into operand planes above)
      const _Float16* pa = reinterpret_cast<const _Float16*>(bufA);
      if (c < 2)
        rb_gemm_h3(pa, seg, nseg, ring, tile[c][0][0]);
      else
        rb_gemm_h3_rows<1, 16, QkStoreTailH3>(pa, kLdh, kPlaneH, seg, nseg, ring, tile[c],
                                              QkStoreTailH3{tile[0][0][0], tile[1][0][0], qrow, qb});
    } else if (c == 0) {
      rb_gemm<1, 1, kG256, kPF, NoSide, true>(bufA, kLda, seg, 0, nseg, 0, ring, tile[c]);
    } else if (c == 1) {
      rb_gemm<1, 1, kG256, kPF, QkvStoreSide, true>(bufA, kLda, seg, 0, nseg, 0, ring, tile[c],
                                                    QkvStoreSide{tile[0][0][0], qrow, qb[0]});
    } else {  // V stays column-per-lane (its fragment-order store below needs that); K's tile is stored meanwhile
      rb_gemm<1, 1, kG256, kPF, QkvStoreSide, false>(bufA, kLda, seg, 0, nseg, 0, ring, tile[c],
                                                     QkvStoreSide{tile[1][0][0], qrow ? qrow + 256 : nullptr, qb[1]});
    }
    PPASR_TS(12 + c);
  }
  if (vt.vt) {
    // fused attention route: V in the order the attention's P V MFMAs consume it -- [slab = 32 value columns (this
    // wave's)][row octet][lane = column + 32 * (row quad of the octet)][4 rows]: a lane's register quad i (rows 8i +
    // 4hh .. +3 of its column) is one 16-byte piece and the wave's 64 pieces are 1 KiB contiguous, on the store side
    // here and on the load side there (rows >= valid of the last block land in the padding behind row M)
    const float bv = w.bqkv[2 * 256 + wave * 32 + (lane & 31)];
    float* dst = vt.vt + ((size_t)wave * (vt.stride >> 3) + (r0 >> 3)) * 256 + 4 * lane;
    if constexpr (H3) tile[2][0][0] *= kH3Inv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x16& t = tile[2][0][0];
      *reinterpret_cast<f32x4*>(dst + i * 256) = f32x4{t[4 * i] + bv, t[4 * i + 1] + bv, t[4 * i + 2] + bv, t[4 * i + 3] + bv};
    }
  } else {
    const int col = 2 * 256 + wave * 32 + (lane & 31);
    const float bv = w.bqkv[col];
    if constexpr (H3) tile[2][0][0] *= kH3Inv;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = acc_row(r, lane);
      if (row < valid) qkv[(size_t)(r0 + row) * 768 + col] = tile[2][0][0][r] + bv;
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_ffn_qkv(const float* __restrict__ x_in, float* __restrict__ x1,
                                                      float* __restrict__ qkv, LayerW w, int M, int n_chunks, PadSkip ps,
                                                      VtOut vt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  float* bufX = smem;
  float* bufA = bufX + kRows * kLda;
  float* bufH = bufA + kRows * kLda;
  const int wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  BRing<1> ring;
  ring_prime(ring, w.ffm_w1 + (size_t)wave * kTs256, 0);
  rb_load_rows(bufX, kLda, x_in + (size_t)r0 * kD, kRows, valid);
  ffn_qkv_body(bufX, bufA, bufH, x1, qkv, w, r0, valid, n_chunks, ring, vt);
}
// fp32 fragment packing -> fp16 x3 packing (h3.h): thread = (tile, 16-wide k step, lane); its 8 weights k = 16 ks + 8 (l >> 5)
// + e of column 32 tile + (l & 31) sit in k-group 2 ks + (l >> 5) of the source, lanes (l & 31) and (l & 31) + 32
__global__ __launch_bounds__(256) void k_repack_h3(const float* __restrict__ src, _Float16* __restrict__ dst, int n_tiles, int G) {
  const int KS = G >> 1;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)n_tiles * KS * 64) return;
  const int l = (int)(t & 63), ks = (int)((t >> 6) % KS), nt = (int)((t >> 6) / KS);
  const float* g = src + ((size_t)nt * G + 2 * ks + (l >> 5)) * 256;
  f16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = g[((l & 31) + 32 * (e >> 2)) * 4 + (e & 3)] * kH3Sw;
    if (!(fabsf(v) <= kH3Max)) atomicAdd(&g_h3_ovf, 1u);  // |w| >= 255.9: ppasr_set_gemm_mode refuses the mode
    hi[e] = (_Float16)v;
    lo[e] = (_Float16)(v - (float)hi[e]);
  }
  _Float16* d = dst + (((size_t)nt * KS + ks) * 2 * 64 + l) * 8;
  *reinterpret_cast<f16x8*>(d) = hi;
  *reinterpret_cast<f16x8*>(d + 64 * 8) = lo;
}
unsigned int* conformer_h3_ovf_counter() { return h3_ovf_counter(); }
void launch_repack_h3(const f32x4* src, f32x4* dst, int n_tiles, int G, hipStream_t st) {
  const long long n = (long long)n_tiles * (G >> 1) * 64;
  PPASR_LAUNCH(k_repack_h3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float*>(src),
               reinterpret_cast<_Float16*>(dst), n_tiles, G);
}

// the same with the feed-forward module on the fp16 x3 route (ppasr_set_gemm_mode; w: the layer's h3 view)
__global__ __launch_bounds__(kThreads) void k_ffn_qkv_h3(const float* __restrict__ x_in, float* __restrict__ x1,
                                                         float* __restrict__ qkv, LayerW w, int M, int n_chunks, PadSkip ps,
                                                         VtOut vt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);
  if (blk < 0) return;
  float* bufX = smem;
  float* bufA = bufX + kRows * kLda;
  float* bufH = bufA + kRows * kLda;
  const int wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  BRing<1> ring;
  ring_prime(ring, w.ffm_w1 + (size_t)wave * kTs256, 0);
  rb_load_rows(bufX, kLda, x_in + (size_t)r0 * kD, kRows, valid);
  ffn_qkv_body<true>(bufX, bufA, bufH, x1, qkv, w, r0, valid, n_chunks, ring, vt);
}
constexpr size_t kLdsFfnQkv = 4 * kRows * kLda * sizeof(float);
void launch_ffn_qkv(const float* x_in, float* x1, float* qkv, const LayerW& w, int M, int n_chunks, hipStream_t st,
                    const PadSkip& ps, VtOut vt, bool h3) {
  if (h3) {
    PPASR_LAUNCH(k_ffn_qkv_h3, dim3((M + kRows - 1) / kRows), dim3(kThreads), kLdsFfnQkv + kH3ExtraLds, st, x_in, x1, qkv, w,
                 M, n_chunks, ps, vt);
    return;
  }
  PPASR_LAUNCH(k_ffn_qkv, dim3((M + kRows - 1) / kRows), dim3(kThreads), kLdsFfnQkv, st, x_in, x1, qkv, w, M,
                     n_chunks, ps, vt);
}

// -------------------------------------------------------------------------------------
// Attention: RelPositionMultiHeadedAttention.forward + forward_attention (attention.py:198-262, 86-126) and the
// Efficient-Conformer's grouped form (efficient_conformer/attention.py:40-79,128-193): attention_kernels.hip
// (k_attention_t<64 / 192>); the batched plain-head layers run it fused with the out-projection (k_attn_out_glu below).
// -------------------------------------------------------------------------------------
void launch_attention(const AttnArgs& a, int B, int H, hipStream_t st) {
  if (!launch_attention_t(a, B, H, st)) {  // (row strides / widths are multiples of 256 by construction: never taken)
    fprintf(stderr, "ppasr_hip: attention launch refused (group %d, strides %d %d %d, width %d)\n", a.group, a.q_stride,
            a.k_stride, a.v_stride, a.dm);
    abort();
  }
}

// -------------------------------------------------------------------------------------
// S3: x2 = x1 + ctx*Wo + bo ; g = GLU(pointwise_conv1(mask(LN_conv(x2))))
// (attention.py:126, encoder.py:399-409, convolution.py:104-106,125-126)
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_out_glu(const float* __restrict__ ctx, const float* __restrict__ x1,
                                                      float* __restrict__ x2, float* __restrict__ g,
                                                      float* __restrict__ xhat_out, LayerW w,
                                                      const int64_t* __restrict__ lens, int M, int Tp, int mask_mul,
                                                      PadSkip ps, int stop_after_ln) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  float* bufX = smem;
  float* bufA = bufX + kRows * kLda;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  const int col = wave * 32 + (lane & 31);
  BRing<1> ring;
  const f32x4* seg_o = w.wo + (size_t)wave * kTs256;
  const f32x4* seg_val = w.pw1 + (size_t)wave * kTs256;         // GLU value columns [32w, 32w+32)
  const f32x4* seg_gate = w.pw1 + (size_t)(8 + wave) * kTs256;  // GLU gate columns 256 + [32w, 32w+32)
  ring_prime(ring, seg_o, 0);
  rb_load_rows(bufA, kLda, ctx + (size_t)r0 * kD, kRows, valid);
  __syncthreads();
  {
    float res[16];  // residual rows requested before the GEMM, branch-free (clamped row; see k_conv_ffn)
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = x1[(size_t)(r0 + min(acc_row(r, lane), valid - 1)) * kD + col];
    f32x16 acc[1][1];
    acc_zero(acc);
    rb_gemm<1, 1, kG256>(bufA, kLda, seg_o, 0, stop_after_ln ? nullptr : seg_val, 0, ring, acc);
    const float bv = w.bo[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, lane);
      const float v = (row < valid) ? res[r] + (acc[0][0][r] + bv) : 0.f;
      if (row < valid) x2[(size_t)(r0 + row) * kD + col] = v;
      bufX[row * kLda + col] = v;
    }
  }
  __syncthreads();
  rb_layernorm<false>(bufX, bufA, kLda, kRows, w.ln_conv_g, w.ln_conv_b, 1e-5f, PadRows{lens, r0, Tp, M, mask_mul});
  // streaming: the conv-module input (what the reference keeps as cnn_cache, convolution.py:117)
  if (xhat_out) rb_store_rows(xhat_out + (size_t)r0 * kD, bufA, kLda, kRows, valid);
  if (stop_after_ln) return;  // under-filled launches: pointwise_conv1 + GLU run as k_pw1_glu_cols (two column halves)
  __syncthreads();
  {
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
      float val = av[0][0][r] + bval;
      float gate = ag[0][0][r] + bgate;
      if (row < valid) g[(size_t)(r0 + row) * kD + col] = val * sigmoidf(gate);
    }
  }
}
// pointwise_conv1 + GLU of LayerNorm'd (and pad-masked) rows, the 256 output columns over gridDim.y = 2 workgroups:
// waves 0-3 compute the VALUE tiles of the workgroup's 128 columns, waves 4-7 the GATE tiles of the same columns (one
// GEMM unit each instead of two in sequence); values cross to the gate waves through LDS.
__global__ __launch_bounds__(kThreads) void k_pw1_glu_cols(const float* __restrict__ xhat, float* __restrict__ g, LayerW w,
                                                           int M, PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  float* bufA = smem;
  float* vals = bufA + kRows * kLda;  // [32][132]
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  const int is_gate = wave >> 2, t = wave & 3, y = blockIdx.y;
  const int col = 128 * y + 32 * t + (lane & 31);              // output column
  const f32x4* seg = w.pw1 + (size_t)((is_gate ? 8 : 0) + 4 * y + t) * kTs256;
  BRing<1> ring;
  ring_prime(ring, seg, 0);
  rb_load_rows(bufA, kLda, xhat + (size_t)r0 * kD, kRows, valid);
  __syncthreads();
  f32x16 acc[1][1];
  acc_zero(acc);
  rb_gemm<1, 1, kG256>(bufA, kLda, seg, 0, nullptr, 0, ring, acc);
  const float bv = w.pw1_b[(is_gate ? kD : 0) + col];
  if (!is_gate) {
#pragma unroll
    for (int r = 0; r < 16; ++r) vals[acc_row(r, lane) * 132 + 32 * t + (lane & 31)] = acc[0][0][r] + bv;
  }
  __syncthreads();
  if (is_gate) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, lane);
      if (row < valid) g[(size_t)(r0 + row) * kD + col] = vals[row * 132 + 32 * t + (lane & 31)] * sigmoidf(acc[0][0][r] + bv);
    }
  }
}
constexpr size_t kLdsOutGlu = 2 * kRows * kLda * sizeof(float);
constexpr size_t kLdsPw1Cols = (kRows * kLda + kRows * 132) * sizeof(float);
void launch_out_glu(const float* ctx, const float* x1, float* x2, float* g, float* xhat_out, const LayerW& w,
                    const int64_t* lens, int M, int Tp, int mask_mul, hipStream_t st, const PadSkip& ps, float* split_xhat) {
  // split_xhat != nullptr (under-filled launches): out-projection + LayerNorm in one launch (the LayerNorm'd rows go to
  // split_xhat), pointwise_conv1 + GLU in a second one with the columns over two workgroups per row block
  float* xh = xhat_out ? xhat_out : split_xhat;
  PPASR_LAUNCH(k_out_glu, dim3((M + kRows - 1) / kRows), dim3(kThreads), kLdsOutGlu, st, ctx, x1, x2, g, xh, w, lens,
                     M, Tp, mask_mul, ps, split_xhat ? 1 : 0);
  if (split_xhat)
    PPASR_LAUNCH(k_pw1_glu_cols, dim3((M + kRows - 1) / kRows, 2), dim3(kThreads), kLdsPw1Cols, st, xh, g, w, M, ps);
}

// -------------------------------------------------------------------------------------
// S2 + S3 in one launch (batched path, plain 64-wide heads): relative-position attention of a 32-query block for
// ALL heads with the context rows kept in LDS, then k_out_glu's tail (out-projection -> +residual -> LN_conv -> mask
// -> pointwise_conv1 -> GLU) on them.  Saves the context round trip through HBM, one launch and one pipeline
// fill per layer, and gives every CU one workgroup (B x ceil(T'/32) = 256 for 32 x 249 frames).
//
// Attention part: NO workgroup barriers.  Wave w = (head h = w >> 1, key half w & 1) is an independent flash-attention
// worker: it owns the keys [128*half, 128*half + 128) of every 256-key block, keeps its Q' fragments (the MFMA A
// operand) in 64 VGPRs, runs S = Q'K'^T for 64 keys at a time (K' fragments straight from L2), does the online softmax
// on its private 32x64 score tile in LDS (each lane owns half a row), and accumulates O += P V into two accumulator
// tiles.  Only the wave itself reads what it wrote, so `s_waitcnt` replaces every barrier; the two waves of a head sit
// on the same SIMD and fill each other's latency gaps.  One barrier at the end merges the two key halves
// (flash-decoding style: O = (O0 e^{m0-m} + O1 e^{m1-m}) / (l0 e^{m0-m} + l1 e^{m1-m})).
// (The previous design -- waves of a head group sharing score tiles, three barriers per key block -- spent half of
// its time at those barriers and in first-touch latencies that nothing overlapped.)
// -------------------------------------------------------------------------------------
constexpr int kPLd = 68;                       // private score-tile row stride: 64 keys + 4
constexpr int kPTile = 32 * kPLd;              // floats per wave
constexpr int kFusedAttnFloats = kWaves * kPTile + kWaves * 64 + kRows * kLda;
static_assert(2 * kRows * kLda <= kWaves * kPTile + kWaves * 64, "bufX/bufA alias the attention scratch");
static_assert(kRows * kLda + 4 * kPTile <= kWaves * kPTile, "Q'_v and the four merge tiles fit in front of Stat");
static_assert(kRows * kLda * 4 + kH3TileBytes <= kWaves * kPTile * 4, "fp16 x3: operand planes at bufA stay in front of Stat");
static_assert(kFusedAttnFloats * 4 <= 160 * 1024, "LDS budget");
// H3: the out-projection and pointwise_conv1 units on the fp16 x3 route (h3.h; w.wo / w.pw1 are then the re-packed weights)
template <bool H3>
__device__ __forceinline__ void attn_out_glu_body(const AttnArgs& a, int B, const float* __restrict__ x1, float* __restrict__ x2,
                                                  float* __restrict__ g, const LayerW& w) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ps = smem;                          // [32][260] Q'_v = q + pos_bias_v, then 4 merge tiles [head][32][68]
  float* Stat = Ps + kWaves * kPTile;        // [8 waves][2][32]: running max, running sum of each wave's key half
  float* bufC = Stat + kWaves * 64;          // [32][260] Q'_u = q + pos_bias_u during the key loop, then the context rows
  float* bufX = smem;                        // out phase (aliases the tiles)
  float* bufA = bufX + kRows * kLda;
  const int lane = lane_id(), wave = wave_id();
  const int h = wave >> 1, khalf = wave & 1;
  // XCD-aware block -> (utterance, query block) map.  Workgroups are dealt round-robin to the 8 XCDs (linear id % 8),
  // and every XCD has its own 4 MiB L2.  All query blocks of utterance b run on XCD b % 8, so an XCD's L2 holds the
  // keys / values / positional rows of B/8 utterances (3.3 MB for 32 x 249 frames) instead of every XCD streaming
  // all of them from HBM (the plain (qblk, b) grid put the 8 blocks of an utterance on 8 different XCDs).
  const int nq = (a.T1 + 31) / 32;
  const int slot = blockIdx.x >> 3;
  const int b = (slot / nq) * 8 + (blockIdx.x & 7);
  const int q0 = (slot % nq) * 32;
  if (b >= B) return;  // batch not a multiple of 8: the padded slots are empty
  const int T = a.T1;
  const int valid = min(32, T - q0);
  const float* __restrict__ qb = a.q + (size_t)b * T * a.q_stride;
  const float* __restrict__ kbp = a.k + (size_t)b * a.T2 * a.k_stride;
  const float* __restrict__ vbp = a.v + (size_t)b * a.T2 * a.v_stride;
  const float* __restrict__ ptab = a.ptab + (size_t)a.pos0 * kD;
  const int pstride = a.pos_stride;
  const int64_t len_b = a.lens ? a.lens[b] : (int64_t)a.mask_mul * a.T2;
  int T2 = a.T2;  // keys walked: all of them, or (ragged batch) only up to the last valid one
  if (a.pad_skip > 0 && a.lens) {
    const int64_t lb = len_b > 0 ? len_b : 0;
    const int64_t n_valid = (lb + a.mask_mul - 1) / a.mask_mul;
    if (q0 >= min((int64_t)T, n_valid + (a.pad_skip - 1))) return;  // whole row block behind the needed frames
    T2 = (int)max((int64_t)1, min((int64_t)T2, n_valid));
  }
  float* QV = Ps;
  float* P = Ps + kRows * kLda + (wave >> 1) * kPTile;  // merge tile of this wave's head (behind QV)
  BRing<1> ring;
  const f32x4* seg_o = w.wo + (size_t)wave * kTs256;
  const f32x4* seg_val = w.pw1 + (size_t)wave * kTs256;
  const f32x4* seg_gate = w.pw1 + (size_t)(8 + wave) * kTs256;
  const int hh = lane >> 5, l31 = lane & 31;
  constexpr int NG = 16, PF = 4;
  PPASR_TS(32);
  PPASR_WG_TS(512 + 0);

  // ---- key loop: flash attention on TRANSPOSED score tiles, everything between the two MFMA phases in registers ----
  // S^T = K' Q'^T: the K' fragment is the MFMA's A operand and the Q' fragment its B operand, so a lane holds, for ITS
  // query row l31, the scores of 16 keys per 32-key tile (key = (r&3) + 8(r>>2) + 4hh).  The row maximum / sum are
  // then in-lane reductions plus ONE exchange with lane^32, the probabilities never leave the accumulator registers --
  // p[t][4i+j] is exactly the B operand (k slot (hh, j)) of the i-th k-group of O^T += V^T P^T -- and the running
  // rescale of O^T (lane = query row again) is a per-lane multiply.  (The row-major form went through a private LDS
  // score tile: 32 ds_write_b32 + 16 ds_read_b128 + 8 ds_write_b128 + 16 ds_bpermute per 64 keys; every VALU / LDS
  // instruction issued on a SIMD takes its issue cycles away from that SIMD's MFMA pipe -- tools/microbench_mfma.hip.)
  // The keys are walked in the row space of the whole batch (row m = b*T + key) from the utterance's first row rounded
  // DOWN to a multiple of 8: the values (a.vt, written by the QKV stage in fragment order: [32-column slab][row
  // octet][64 lanes][4 rows]) are then read like the packed weights, 1 KiB contiguous per wave-load, whole cache lines;
  // the <= 7 rows in front (u < shift) are masked like the keys >= kv_end.
  const int mrow0 = b * T;
  const int shift = mrow0 & 7;
  int kv_end = (int)min((int64_t)T2, max((int64_t)0, (len_b + a.mask_mul - 1) / a.mask_mul));  // keys >= kv_end are PAD
  const int U = kv_end > 0 ? kv_end + shift : 0;  // shifted key space: u = key + shift in [0, U)
  constexpr float kScale = 0.125f * 1.4426950408889634f;  // 1/sqrt(dk) * log2(e): p = 2^(s*kScale - m*kScale)
  const __amdgpu_buffer_rsrc_t rs_k = wstream_rsrc(kbp + h * 64), rs_p = wstream_rsrc(ptab + h * 64),
                               rs_v = wstream_rsrc(a.vt + ((size_t)(2 * h) * (a.vt_stride >> 3) + ((mrow0 - shift) >> 3)) * 256);
  const int voff_v = lane * 16;
  const int kstride_b = a.k_stride * 4, pstride_b = pstride * kD * 4;
  // byte offsets of this lane's two keys (tile 0 / 1) of the sub-block at u0, in K and in the positional table
  int vk[2], vp[2];
  auto key_offsets = [&](int u0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = min(max(u0 - shift + 32 * t + l31, 0), kv_end - 1);  // out-of-range keys are masked afterwards
      vk[t] = key * kstride_b + 16 * hh;
      vp[t] = key * pstride_b + 16 * hh;
    }
  };
  // K' fragment of k-group gk (features 8gk + 4hh .. +3 of [k | p]) of this lane's key of tile t
  auto kfrag = [&](int t, int gk) -> f32x4 {
    return gk < 8 ? wstream_load(rs_k, vk[t], gk * 32) : wstream_load(rs_p, vp[t], (gk - 8) * 32);
  };
  // K' operands in bursts of 4 k-groups (= one whole 128-byte line of each key row: a lane's 16-byte pieces of 4
  // consecutive k-groups are requested back to back, so the line is fetched from L2 once; one k-group at a time the
  // 8 waves push 64 KiB through the 32 KiB L1 between two uses of a line and every line is fetched 4 times), double
  // buffered: super-group sg + 1 is in flight while sg feeds the MFMAs
  f32x4 kq[2][4][2];
  auto load_sg = [&](int buf, int sg) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kq[buf][i][0] = kfrag(0, 4 * sg + i);
      kq[buf][i][1] = kfrag(1, 4 * sg + i);
    }
  };
  auto prime_k = [&](int u0) {
    key_offsets(u0);
    load_sg(0, 0);
  };

  // ---- Q' = [q + pos_bias_u | q + pos_bias_v] of the block's 32 query rows -> LDS (bufC / QV); the key loop reads
  // its B-operand fragment Q'[row l31][8 gk + 4 hh .. +3] from there, one ds_read_b128 per k-group ----
  {
    const f32x4 pu = *reinterpret_cast<const f32x4*>(a.pos_u + 4 * lane), pv = *reinterpret_cast<const f32x4*>(a.pos_v + 4 * lane);
    for (int row = wave; row < kRows; row += kWaves) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < valid) v = *reinterpret_cast<const f32x4*>(qb + (size_t)(q0 + row) * a.q_stride + 4 * lane);
      *reinterpret_cast<f32x4*>(bufC + row * kLda + 4 * lane) = v + pu;
      *reinterpret_cast<f32x4*>(QV + row * kLda + 4 * lane) = v + pv;
    }
    __syncthreads();
  }
  const float* qfrag_u = bufC + l31 * kLda + h * 64 + 4 * hh;
  const float* qfrag_v = QV + l31 * kLda + h * 64 + 4 * hh;
  auto qfrag = [&](int gk) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(gk < 8 ? qfrag_u + 8 * gk : qfrag_v + 8 * (gk - 8));
  };
  PPASR_TS(33);
  // (requesting these before the Q' staging, or the V operands before the score MFMAs, measured no better: NOTES §4)
  if (khalf * 128 < U) prime_k(khalf * 128);
  f32x16 acc_o[2];  // O^T: acc_o[ct][r] = O[query l31][h*64 + 32ct + (r&3) + 8(r>>2) + 4hh]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // raw-score running max / running sum of row l31 (same in both lane halves)
  bool first = true;

  const int nkb = (U + 255) / 256;
  for (int kb = 0; kb < nkb; ++kb) {
    for (int sb = 0; sb < 2; ++sb) {
      const int u0 = kb * 256 + khalf * 128 + sb * 64;
      if (u0 >= U) break;  // wave-uniform
      int u_next = sb == 0 ? u0 + 64 : (kb + 1) * 256 + khalf * 128;
      if (u_next >= U) u_next = (sb == 0 && (kb + 1) * 256 + khalf * 128 < U) ? (kb + 1) * 256 + khalf * 128 : -1;
      const bool edge = (u0 < shift) || (u0 + 64 > U);  // sub-block holds masked keys (wave-uniform)
      if (kb == 0) PPASR_WAVE_TS(16 + 4 * sb);
      // V^T operands of the first PQ k-groups: requested before the score MFMAs, consumed after the softmax.  k-group q = (tile t = q >> 2,
      // i = q & 3) covers the keys u0 + 32t + 8i + 4hh .. +3; vt[q][ct] = those 4 keys of value column 32ct + l31
      const int soff_v = (u0 >> 3) * 1024, soff_v1 = soff_v + (a.vt_stride >> 3) * 1024;
      auto vload = [&](int q, int ct) -> f32x4 { return wstream_load(rs_v, voff_v, (ct ? soff_v1 : soff_v) + q * 1024); };
      // (all 8 k-groups in one burst: 4 consecutive k-groups share the 128-byte lines of their columns)
      constexpr int PQ = 8;
      f32x4 ringv[PQ][2];
      auto prime_v = [&]() {
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
          ringv[q][0] = vload(q, 0);
          ringv[q][1] = vload(q, 1);
        }
      };
      // ---- S^T = K' Q'^T for 64 keys (two 32-key tiles, two independent accumulator chains) ----
      f32x16 acc_s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_s[t][r] = 0.f;
      {
        f32x4 q_cur = qfrag(0), q_nxt = q_cur;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
          if (sg + 1 < 4) load_sg((sg + 1) & 1, sg + 1);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int gk = 4 * sg + i;
            if (gk + 1 < NG) q_nxt = qfrag(gk + 1);
            const f32x4 k0 = kq[sg & 1][i][0], k1 = kq[sg & 1][i][1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc_s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(k0[j], q_cur[j], acc_s[0], 0, 0, 0);
              acc_s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[j], q_cur[j], acc_s[1], 0, 0, 0);
            }
            q_cur = q_nxt;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (kb == 0) PPASR_WAVE_TS(17 + 4 * sb);
      prime_v();
      // ---- online softmax in registers ----
      if (edge) {
        // element r of tile t is key u = u0 + 4hh + c with c = 32t + (r&3) + 8(r>>2): masked iff c < lo or c >= hi
        const int lo = shift - u0 - 4 * hh, hi = U - u0 - 4 * hh;
        if (u0 < shift) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < lo) acc_s[0][r] = -INFINITY;  // shift <= 7: only the first register quad of tile 0 can be hit
        }
        if (u0 + 64 > U) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (32 * t + (r & 3) + 8 * (r >> 2) >= hi) acc_s[t][r] = -INFINITY;
        }
      }
      float bm = max3f(acc_s[0][0], acc_s[0][1], acc_s[0][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) bm = max3f(bm, acc_s[0][r], acc_s[0][r + 1]);
      bm = max3f(bm, acc_s[0][15], acc_s[1][0]);
#pragma unroll
      for (int r = 1; r < 15; r += 2) bm = max3f(bm, acc_s[1][r], acc_s[1][r + 1]);
      bm = fmaxf(bm, acc_s[1][15]);
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      const float m_new = fmaxf(m_run, bm);
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;  // every key so far masked: p = 0, alpha irrelevant (O = 0)
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_safe) * kScale);
      const float mb = -m_safe * kScale;
      f32x2 ps2 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 e = f32x2{acc_s[t][r], acc_s[t][r + 1]} * f32x2{kScale, kScale} + f32x2{mb, mb};  // v_pk_fma_f32
          acc_s[t][r] = __builtin_amdgcn_exp2f(e[0]);
          acc_s[t][r + 1] = __builtin_amdgcn_exp2f(e[1]);
          ps2 += f32x2{acc_s[t][r], acc_s[t][r + 1]};
        }
      float ps = ps2[0] + ps2[1];
      ps += __shfl_xor(ps, 32);
      m_run = m_new;
      l_run = l_run * alpha + ps;
      if (u_next >= 0) prime_k(u_next);
      if (kb == 0) PPASR_WAVE_TS(18 + 4 * sb);
      // ---- O^T = O^T * alpha + V^T P^T ----
      if (!first) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 o = f32x2{acc_o[ct][r], acc_o[ct][r + 1]} * f32x2{alpha, alpha};  // v_pk_mul_f32
            acc_o[ct][r] = o[0];
            acc_o[ct][r + 1] = o[1];
          }
      }
      first = false;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 v0 = ringv[q][0], v1 = ringv[q][1];
        // (rows outside the utterance meet p = 0 exactly; they hold finite values -- other utterances' rows, or the zeros
        //  ppasr_encode clears the buffer to -- so no select is needed on the values)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pj = acc_s[q >> 2][4 * (q & 3) + j];
          acc_o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[j], pj, acc_o[0], 0, 0, 0);
          acc_o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[j], pj, acc_o[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kb == 0) PPASR_WAVE_TS(19 + 4 * sb);
    }
  }
  PPASR_TS(34);
  // ---- merge the two key halves of every head, normalise -> bufC ----
  // wave 2h+1 hands its O^T (row-major in its scratch tile: [query][64], 4 consecutive columns per register quad)
  // and (m, l) to wave 2h; lane = query row on both sides, so the merge factors are per-lane scalars
  if (khalf == 1) {
    if (hh == 0) {
      Stat[wave * 64 + l31] = m_run;
      Stat[wave * 64 + 32 + l31] = l_run;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(P + l31 * kPLd + 32 * ct + 8 * i + 4 * hh) =
            f32x4{acc_o[ct][4 * i], acc_o[ct][4 * i + 1], acc_o[ct][4 * i + 2], acc_o[ct][4 * i + 3]};
  }
  // fp16 x3: the residual rows of the out-projection epilogue are requested BEFORE the weight stream starts (vmcnt
  // retires in order: behind the stream's first fragments they would hold every later fragment wait for an HBM round trip)
  const int r0 = b * T + q0;
  const int cq = wave * 32 + 4 * hh;                    // first column of this lane's quad 0
  const size_t grow = (size_t)(r0 + min(l31, valid - 1)) * kD;  // this lane's row in x1 / x2 / g
  const bool row_ok = l31 < valid;
  f32x4 res[4];
  if constexpr (H3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) res[q] = *reinterpret_cast<const f32x4*>(x1 + grow + cq + 8 * q);
  }
  ring_prime(ring, seg_o, 0);  // out-projection weights in flight across the barrier
  __syncthreads();
  PPASR_TS(35);
  if (khalf == 0) {
    const float* P1 = P;  // written by wave 2h+1
    const float m1 = Stat[(wave + 1) * 64 + l31], l1 = Stat[(wave + 1) * 64 + 32 + l31];
    const float m = fmaxf(m_run, m1);
    const float e0 = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m) * kScale);
    const float e1 = (m1 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m1 - m) * kScale);
    const float l = l_run * e0 + l1 * e1;
    float inv = (l > 0.f) ? 1.0f / l : 0.f;  // fully masked row -> 0 (attention.py:118)
    if (l31 >= valid) inv = 0.f;
    const float f0 = e0 * inv, f1 = e1 * inv;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 o1 = *reinterpret_cast<const f32x4*>(P1 + l31 * kPLd + 32 * ct + 8 * i + 4 * hh);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc_o[ct][4 * i + e] * f0 + o1[e] * f1;
        *reinterpret_cast<f32x4*>(bufC + l31 * kLda + h * 64 + 32 * ct + 8 * i + 4 * hh) = o;
      }
  }
  __syncthreads();
  PPASR_TS(36);
  // ---- k_out_glu tail on the LDS-resident context ----
  // The three GEMM units run with swapped MFMA operands (rb_gemm SWAP): lane = row l31, register quad q = columns
  // wave*32 + 8q + 4hh .. +3, so residual loads, x2 / g stores and the LDS rows are 16-byte accesses (4 per tile
  // instead of 16) and the epilogue arithmetic is packed.
  f32x4 x2v[4];  // fp16 x3: the x2 rows, stored behind the last weight load of the kernel (with g)
  {
    if constexpr (!H3) {
#pragma unroll
      for (int q = 0; q < 4; ++q) res[q] = *reinterpret_cast<const f32x4*>(x1 + grow + cq + 8 * q);
    }
    f32x16 acc[1][1];
    acc_zero(acc);
    if constexpr (H3) {  // (the context rows' operand planes go where bufA will be: free until the LayerNorm below)
      h3_planes_from_tile(bufC, reinterpret_cast<_Float16*>(bufA));
      rb_gemm_h3(reinterpret_cast<const _Float16*>(bufA), seg_o, seg_val, ring, acc[0][0]);
      acc[0][0] *= kH3Inv;
    } else {
      rb_gemm<1, 1, kG256, kPF, NoSide, true>(bufC, kLda, seg_o, 0, seg_val, 0, ring, acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bo = *reinterpret_cast<const f32x4*>(w.bo + cq + 8 * q);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = res[q][e] + (acc[0][0][4 * q + e] + bo[e]);
      if constexpr (H3) x2v[q] = v;
      else if (row_ok) *reinterpret_cast<f32x4*>(x2 + grow + cq + 8 * q) = v;
      if (!row_ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(bufX + l31 * kLda + cq + 8 * q) = v;
    }
  }
  __syncthreads();
  PPASR_TS(37);
  // conv-module pad mask (frame t of this utterance is PAD iff mask_mul * t >= len_b, convolution.py:104-106) from the
  // length read at kernel start -- the row block lies inside one utterance, no per-row length loads
  struct PadHere {
    int64_t len_b;
    int q0, mul;
    __device__ __forceinline__ bool operator()(int row) const { return (int64_t)mul * (q0 + row) >= len_b; }
  };
  rb_layernorm<false>(bufX, bufA, kLda, kRows, w.ln_conv_g, w.ln_conv_b, 1e-5f,
                      PadHere{a.lens ? len_b : (int64_t)1 << 62, q0, a.mask_mul});
  __syncthreads();
  PPASR_TS(38);
  {
    f32x16 av[1][1], ag[1][1];
    acc_zero(av);
    acc_zero(ag);
    if constexpr (H3) {
      h3_planes_from_tile(bufA, reinterpret_cast<_Float16*>(bufA));  // (in place: 512 B run on into the unused tile space)
      rb_gemm_h3(reinterpret_cast<const _Float16*>(bufA), seg_val, seg_gate, ring, av[0][0]);
      rb_gemm_h3(reinterpret_cast<const _Float16*>(bufA), seg_gate, nullptr, ring, ag[0][0]);
      av[0][0] *= kH3Inv;
      ag[0][0] *= kH3Inv;
    } else {
      rb_gemm<1, 1, kG256, kPF, NoSide, true>(bufA, kLda, seg_val, 0, seg_gate, 0, ring, av);
      rb_gemm<1, 1, kG256, kPF, NoSide, true>(bufA, kLda, seg_gate, 0, nullptr, 0, ring, ag);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bval = *reinterpret_cast<const f32x4*>(w.pw1_b + cq + 8 * q);
      const f32x4 bgate = *reinterpret_cast<const f32x4*>(w.pw1_b + kD + cq + 8 * q);
      const f32x2 s0 = sigmoid2(f32x2{ag[0][0][4 * q] + bgate[0], ag[0][0][4 * q + 1] + bgate[1]});
      const f32x2 s1 = sigmoid2(f32x2{ag[0][0][4 * q + 2] + bgate[2], ag[0][0][4 * q + 3] + bgate[3]});
      const f32x4 o = {(av[0][0][4 * q] + bval[0]) * s0[0], (av[0][0][4 * q + 1] + bval[1]) * s0[1],
                       (av[0][0][4 * q + 2] + bval[2]) * s1[0], (av[0][0][4 * q + 3] + bval[3]) * s1[1]};
      if (row_ok) *reinterpret_cast<f32x4*>(g + grow + cq + 8 * q) = o;
    }
    if constexpr (H3) {
      if (row_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(x2 + grow + cq + 8 * q) = x2v[q];
      }
    }
  }
  PPASR_TS(39);
  PPASR_WG_TS(512 + 1);
}
__global__ __launch_bounds__(kThreads) void k_attn_out_glu(AttnArgs a, int B, const float* __restrict__ x1,
                                                           float* __restrict__ x2, float* __restrict__ g, LayerW w) {
  attn_out_glu_body<false>(a, B, x1, x2, g, w);
}
__global__ __launch_bounds__(kThreads) void k_attn_out_glu_h3(AttnArgs a, int B, const float* __restrict__ x1,
                                                              float* __restrict__ x2, float* __restrict__ g, LayerW w) {
  attn_out_glu_body<true>(a, B, x1, x2, g, w);
}
constexpr size_t kLdsAttnOutGlu = (size_t)kFusedAttnFloats * sizeof(float);
// a: plain-head batched attention arguments (group == 1, T1 == T2 frames, keys/values in the layer's own buffers)
void launch_attn_out_glu(const AttnArgs& a, int B, const float* x1, float* x2, float* g, const LayerW& w, hipStream_t st, bool h3) {
  // 1-D grid of nq * ceil(B/8) * 8 workgroups (see the XCD map in the kernel)
  const int nq = (a.T1 + 31) / 32;
  if (h3)  // (w: the layer's fp16 x3 view)
    PPASR_LAUNCH(k_attn_out_glu_h3, dim3(nq * ((B + 7) / 8) * 8), dim3(kThreads), kLdsAttnOutGlu, st, a, B, x1, x2, g, w);
  else
    PPASR_LAUNCH(k_attn_out_glu, dim3(nq * ((B + 7) / 8) * 8), dim3(kThreads), kLdsAttnOutGlu, st, a, B, x1, x2, g, w);
}

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
void launch_fill_rows(float* dst, const float* row_or_null, int n_rows, hipStream_t st);
__global__ void k_fill_rows(float* __restrict__ dst, const float* __restrict__ row, int n_rows) {
  const int r = blockIdx.x, c = threadIdx.x;
  dst[(size_t)r * kD + c] = row ? row[c] : 0.f;
}
void launch_fill_rows(float* dst, const float* row_or_null, int n_rows, hipStream_t st) {
  PPASR_LAUNCH(k_fill_rows, dim3(n_rows), dim3(256), 0, st, dst, row_or_null, n_rows);
}

// -------------------------------------------------------------------------------------
// S4: causal depthwise conv (k taps, left context k-1; frames before the utterance start
// read GLU(pointwise_conv1(0)) because the reference zero-pads BEFORE pointwise_conv1,
// convolution.py:108-126) -> LayerNorm -> swish -> pointwise_conv2 -> pad mask -> +residual
// -> LN_ff -> FFN -> +0.5 residual -> LN_final     (convolution.py:129-140, encoder.py:416-429)
// -------------------------------------------------------------------------------------
// NEXT: the following layer's S1 (FFN_macaron + QKV, encoder.py:380-391) runs in the same launch on the rows that
// are already LDS-resident (one launch, one store/load of the residual stream and one pipeline fill saved per layer).
// H3: both feed-forward modules (this layer's, the next layer's macaron one) on the fp16 x3 route (h3.h; w / wn are then
// the layers' h3 views: their FFN weight pointers are the re-packed arrays)
template <int KS, bool STREAM, bool NEXT, bool H3>
__device__ __forceinline__ void conv_ffn_body(const float* __restrict__ g, const float* __restrict__ g_hist,
                                              const float* __restrict__ x2, float* __restrict__ x_out, const LayerW& w,
                                              const int64_t* __restrict__ lens, int M, int Tp, int n_chunks, int mask_mul,
                                              const LayerW& wn, float* __restrict__ x1_next, float* __restrict__ qkv_next,
                                              int left_ctx, const PadSkip& ps, VtOut vt_next) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  float* bufX = smem;
  float* bufA = bufX + kRows * kLda;
  float* bufH = bufA + kRows * kLda;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  const int col = wave * 32 + (lane & 31);
  BRing<1> ring;
  const f32x4* seg_pw2 = w.pw2 + (size_t)wave * kTs256;
  PPASR_TS(0);
  if (NEXT) PPASR_WG_TS(0);
  ring_prime(ring, seg_pw2, 0);
  // lengths of the (at most two, when Tp >= 32) utterances this block touches: uniform loads, requested first thing
  const int pb0 = r0 / Tp, pt0 = r0 - pb0 * Tp, pnb = max(M / Tp, 1);
  int64_t plen0 = 0, plen1 = 0;
  if (lens && Tp >= kRows) {
    plen0 = lens[min(pb0, pnb - 1)];
    plen1 = lens[min(pb0 + 1, pnb - 1)];
  }
  if (!STREAM && KS <= 15 && Tp >= kRows / kWaves) {
    // depthwise conv + conv-module LayerNorm (nn.LayerNorm(channels), eps 1e-5, convolution.py:71) + swish in registers
    if constexpr (!STREAM && KS <= 15)
      dwconv_ln_phase<KS>(g, bufA, w.dw_w, w.dw_b, w.glu_pad, w.ln_cm_g, w.ln_cm_b, w.cm_eps, r0, M, Tp, left_ctx);
    PPASR_TS(1);
  } else {
    dwconv_phase<KS, STREAM>(g, g_hist, bufA, bufH, bufX, w.dw_w, w.dw_b, w.glu_pad, r0, M, Tp, left_ctx);
    __syncthreads();
    PPASR_TS(1);
    rb_layernorm<true>(bufA, bufA, kLda, kRows, w.ln_cm_g, w.ln_cm_b, w.cm_eps);
  }
  // residual rows and pad flags of the pointwise_conv2 epilogue: requested before the GEMM, branch-free (clamped row),
  // so that their global round trips (~2.5 us each under load) overlap the pointwise_conv2 GEMM (8 us; issued after
  // the depthwise phase, whose register window they would otherwise compete with).
  // (Inside the epilogue the 16 conditional loads per lane ran as dependent round trips: 9 us.)
  // The pad flags used to come from PadRows per row (a dependent lens[b] load + compare inside the loop): the 16
  // residual loads then ran as 16 serial round trips (6.8 us, per-phase stamps).  A 32-row block touches at most two
  // utterances when Tp >= 32: their lengths are two uniform loads, the flags plain arithmetic.
  // (pointwise_conv2 runs with swapped operands: lane = row, so the residual row is four 16-byte loads and the pad
  //  flag ONE value per lane)
  PadRows is_pad{lens, r0, Tp, M, mask_mul};
  const int l31 = lane & 31, cq = wave * 32 + 4 * (lane >> 5);
  f32x4 res[4];
  {
    const float* rp = x2 + (size_t)(r0 + min(l31, valid - 1)) * kD + cq;
#pragma unroll
    for (int q = 0; q < 4; ++q) res[q] = *reinterpret_cast<const f32x4*>(rp + 8 * q);
  }
  bool pad = false;
  if (lens) {
    if (Tp >= kRows) {
      const int tt = pt0 + l31;
      const bool over = tt >= Tp;
      const int64_t t = over ? tt - Tp : tt, len = over ? plen1 : plen0;
      pad = r0 + l31 < M && (int64_t)mask_mul * t >= len;
    } else {
      pad = is_pad(l31);
    }
  }
  __syncthreads();
  PPASR_TS(2);
  {
    f32x16 acc[1][1];
    acc_zero(acc);
    if constexpr (H3) {  // (w.pw2: the re-packed weight)
      h3_planes_from_tile(bufA, reinterpret_cast<_Float16*>(bufA));
      rb_gemm_h3(reinterpret_cast<const _Float16*>(bufA), seg_pw2, w.ff_w1 + (size_t)wave * kTs256, ring, acc[0][0]);
      acc[0][0] *= kH3Inv;
    } else {
      rb_gemm<1, 1, kG256, kPF, NoSide, true>(bufA, kLda, seg_pw2, 0, w.ff_w1 + (size_t)wave * kTs256, 0, ring, acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(w.pw2_b + cq + 8 * q);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (l31 < valid) ? res[q][e] + (pad ? 0.f : acc[0][0][4 * q + e] + bv[e]) : 0.f;
      *reinterpret_cast<f32x4*>(bufX + l31 * kLda + cq + 8 * q) = o;
    }
  }
  __syncthreads();
  PPASR_TS(3);
  rb_layernorm(bufX, bufA, kLda, kRows, w.ln_ff_g, w.ln_ff_b, 1e-5f);
  __syncthreads();
  PPASR_TS(4);
  f32x16 acc2[1][1];
  acc_zero(acc2);
  if constexpr (H3)
    ffn_phase_h3(bufA, w.ff_w1, w.ff_b1, w.ff_w2, n_chunks, NEXT ? wn.ffm_w1 + (size_t)wave * kTs256 : nullptr, ring, acc2);
  else
    ffn_phase<true>(bufA, bufH, w.ff_w1, w.ff_b1, w.ff_w2, n_chunks, NEXT ? wn.ffm_w1 + (size_t)wave * kTs256 : nullptr,
                    ring, acc2);
  PPASR_TS(5);
  residual_epilogue_t(bufX, acc2, w.ff_b2, 0.5f);
  __syncthreads();
  PPASR_TS(6);
  rb_layernorm(bufX, bufX, kLda, kRows, w.ln_fin_g, w.ln_fin_b, 1e-5f);
  // rb_layernorm and rb_store_rows use the same wave->row mapping: no barrier needed
  if (x_out) rb_store_rows(x_out + (size_t)r0 * kD, bufX, kLda, kRows, valid);  // (nullptr: nobody reads it, see capi.hip)
  PPASR_TS(7);
  if (NEXT) ffn_qkv_body<H3>(bufX, bufA, bufH, x1_next, qkv_next, wn, r0, valid, n_chunks, ring, vt_next);
  PPASR_TS(15);
  if (NEXT) PPASR_WG_TS(1);
}
template <int KS, bool STREAM, bool NEXT>
__global__ __launch_bounds__(kThreads) void k_conv_ffn(const float* __restrict__ g, const float* __restrict__ g_hist,
                                                       const float* __restrict__ x2, float* __restrict__ x_out, LayerW w,
                                                       const int64_t* __restrict__ lens, int M, int Tp, int n_chunks,
                                                       int mask_mul, LayerW wn, float* __restrict__ x1_next,
                                                       float* __restrict__ qkv_next, int left_ctx, PadSkip ps, VtOut vt_next) {
  conv_ffn_body<KS, STREAM, NEXT, false>(g, g_hist, x2, x_out, w, lens, M, Tp, n_chunks, mask_mul, wn, x1_next, qkv_next,
                                         left_ctx, ps, vt_next);
}
template <int KS, bool NEXT>
__global__ __launch_bounds__(kThreads) void k_conv_ffn_h3(const float* __restrict__ g, const float* __restrict__ x2,
                                                          float* __restrict__ x_out, LayerW w, const int64_t* __restrict__ lens,
                                                          int M, int Tp, int n_chunks, int mask_mul, LayerW wn,
                                                          float* __restrict__ x1_next, float* __restrict__ qkv_next,
                                                          int left_ctx, PadSkip ps, VtOut vt_next) {
  conv_ffn_body<KS, false, NEXT, true>(g, nullptr, x2, x_out, w, lens, M, Tp, n_chunks, mask_mul, wn, x1_next, qkv_next,
                                       left_ctx, ps, vt_next);
}
constexpr size_t kLdsConvFfn = 4 * kRows * kLda * sizeof(float);
void launch_conv_ffn(const float* g, const float* g_hist, const float* x2, float* x_out, const LayerW& w,
                     const int64_t* lens, int M, int Tp, int n_chunks, int ksize, int mask_mul, const LayerW* next,
                     float* x1_next, float* qkv_next, hipStream_t st, bool causal, const PadSkip& ps, VtOut vt_next, bool h3) {
  dim3 grid((M + kRows - 1) / kRows);
  const int left_ctx = causal ? ksize - 1 : (ksize - 1) / 2;
  const LayerW& wn = next ? *next : w;
  if (h3) {  // (conv_ffn_h3_supported: kernels 15 and 7, no history rows)
#define LAUNCH_CF_H3(KS, NX)                                                                                                 \
  PPASR_LAUNCH((k_conv_ffn_h3<KS, NX>), grid, dim3(kThreads), kLdsConvFfn + kH3ExtraLds, st, g, x2, x_out, w, lens, M, Tp, \
               n_chunks, mask_mul, wn, x1_next, qkv_next, left_ctx, ps, vt_next)
    if (ksize == 15 && next) LAUNCH_CF_H3(15, true);
    else if (ksize == 15) LAUNCH_CF_H3(15, false);
    else if (next) LAUNCH_CF_H3(7, true);
    else LAUNCH_CF_H3(7, false);
#undef LAUNCH_CF_H3
    return;
  }
#define LAUNCH_CF(KS)                                                                                                  \
  if (g_hist)                                                                                                          \
    PPASR_LAUNCH((k_conv_ffn<KS, true, false>), grid, dim3(kThreads), kLdsConvFfn, st, g, g_hist, x2, x_out, w,  \
                       lens, M, Tp, n_chunks, mask_mul, wn, x1_next, qkv_next, left_ctx, ps, vt_next);                                    \
  else if (next)                                                                                                       \
    PPASR_LAUNCH((k_conv_ffn<KS, false, true>), grid, dim3(kThreads), kLdsConvFfn, st, g, g_hist, x2, x_out, w,  \
                       lens, M, Tp, n_chunks, mask_mul, wn, x1_next, qkv_next, left_ctx, ps, vt_next);                                    \
  else                                                                                                                 \
    PPASR_LAUNCH((k_conv_ffn<KS, false, false>), grid, dim3(kThreads), kLdsConvFfn, st, g, g_hist, x2, x_out, w, \
                       lens, M, Tp, n_chunks, mask_mul, wn, x1_next, qkv_next, left_ctx, ps, vt_next);
  if (ksize == 15) {
    LAUNCH_CF(15)
  } else if (ksize == 31) {
    LAUNCH_CF(31)
  } else if (ksize == 7) {
    LAUNCH_CF(7)
  }
#undef LAUNCH_CF
}

// -------------------------------------------------------------------------------------
// Split route for UNDER-FILLED grids (<= 128 row blocks: small batches, half-rate layers, single streaming sessions).
// A launch with fewer row blocks than CUs takes as long as a full one, and 89 % of the fused kernel's time is the two
// FFNs; here the layer tail is cut at the FFNs and each FFN's hidden dimension is split over S workgroups per row
// block (partial sums through HBM, joined by the next launch -- kernel boundaries are the only synchronisation, so
// nothing can deadlock).  Same arithmetic except for the order of the final sum over hidden chunks.
//   k_conv_pre : dwconv -> LN -> swish -> pw2 -> mask -> +res                        -> x3
//   k_ffn_part : LN(x) -> FFN over hidden chunks [s n/S, (s+1) n/S)                  -> partial[s]   (grid blocks x S)
//   k_ffn_join : x + scale (sum_s partial[s] + b2) [-> LN]                           -> out
//   k_ln_qkv   : LN_mha(x1) -> one 256-column third of [Wq|Wk|Wv]                    -> qkv          (grid blocks x 3)
// -------------------------------------------------------------------------------------
template <int KS, bool STREAM>
__global__ __launch_bounds__(kThreads) void k_conv_pre(const float* __restrict__ g, const float* __restrict__ g_hist,
                                                       const float* __restrict__ x2, float* __restrict__ x3, LayerW w,
                                                       const int64_t* __restrict__ lens, int M, int Tp, int mask_mul,
                                                       int left_ctx, PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  float* bufX = smem;
  float* bufA = bufX + kRows * kLda;
  float* bufH = bufA + kRows * kLda;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  const int col = wave * 32 + (lane & 31);
  BRing<1> ring;
  const f32x4* seg_pw2 = w.pw2 + (size_t)wave * kTs256;
  ring_prime(ring, seg_pw2, 0);
  PadRows is_pad{lens, r0, Tp, M, mask_mul};
  float res[16];
  unsigned pad_bits = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane);
    res[r] = x2[(size_t)(r0 + min(row, valid - 1)) * kD + col];
    pad_bits |= (is_pad(row) ? 1u : 0u) << r;
  }
  dwconv_phase<KS, STREAM>(g, g_hist, bufA, bufH, bufX, w.dw_w, w.dw_b, w.glu_pad, r0, M, Tp, left_ctx);
  __syncthreads();
  rb_layernorm<true>(bufA, bufA, kLda, kRows, w.ln_cm_g, w.ln_cm_b, w.cm_eps);
  __syncthreads();
  f32x16 acc[1][1];
  acc_zero(acc);
  rb_gemm<1, 1, kG256>(bufA, kLda, seg_pw2, 0, nullptr, 0, ring, acc);
  const float bv = w.pw2_b[col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane);
    const float c = ((pad_bits >> r) & 1u) ? 0.f : acc[0][0][r] + bv;
    if (row < valid) x3[(size_t)(r0 + row) * kD + col] = res[r] + c;
  }
}

__global__ __launch_bounds__(kThreads) void k_ffn_part(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                       const float* __restrict__ ln_b, const f32x4* __restrict__ w1,
                                                       const float* __restrict__ b1, const f32x4* __restrict__ w2,
                                                       float* __restrict__ partial, int M, int n_total, PadSkip ps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  float* bufA = smem;
  float* bufH = bufA + kRows * kLda;  // two hidden-chunk buffers
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  const int n_chunks = n_total / gridDim.y, c0 = blockIdx.y * n_chunks;
  BRing<1> ring;
  ring_prime(ring, w1 + (size_t)(c0 * 8 + wave) * kTs256, 0);
  rb_load_rows(bufA, kLda, x + (size_t)r0 * kD, kRows, valid);
  // (same wave -> row mapping as the load: no barrier between; ln_g == nullptr: the rows are used as they are)
  if (ln_g) rb_layernorm(bufA, bufA, kLda, kRows, ln_g, ln_b, 1e-5f);
  __syncthreads();
  f32x16 acc2[1][1];
  acc_zero(acc2);
  ffn_phase(bufA, bufH, w1, b1, w2, n_chunks, nullptr, ring, acc2, c0, n_total);
  float* out = partial + (size_t)blockIdx.y * M * kD;
  const int col = wave * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane);
    if (row < valid) out[(size_t)(r0 + row) * kD + col] = acc2[0][0][r];
  }
}

__device__ __forceinline__ f32x4 ln_row(f32x4 y, const float* __restrict__ g, const float* __restrict__ b, int lane) {
  const float mean = wave_sum(y[0] + y[1] + y[2] + y[3]) * (1.0f / kD);
  const f32x4 c = y - mean;
  const float var = wave_sum(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]) * (1.0f / kD);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  return c * rstd * *reinterpret_cast<const f32x4*>(g + 4 * lane) + *reinterpret_cast<const f32x4*>(b + 4 * lane);
}
// one wave per row: out = LN_out?(LN_pre?(x) + scale * (sum_s partial[s] + b2))
__global__ __launch_bounds__(256) void k_ffn_join(const float* __restrict__ x, const float* __restrict__ partial, int S,
                                                  const float* __restrict__ b2, float scale, const float* __restrict__ ln_g,
                                                  const float* __restrict__ ln_b, float* __restrict__ out, int M,
                                                  PadSkip ps, const float* __restrict__ pre_g,
                                                  const float* __restrict__ pre_b) {
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= M) return;
  if (pad_block_skippable(ps, row & ~(kRows - 1), kRows, M)) return;
  const int lane = lane_id();
  f32x4 acc = *reinterpret_cast<const f32x4*>(partial + (size_t)row * kD + 4 * lane);
  for (int s = 1; s < S; ++s) acc += *reinterpret_cast<const f32x4*>(partial + ((size_t)s * M + row) * kD + 4 * lane);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + 4 * lane);
  f32x4 y = *reinterpret_cast<const f32x4*>(x + (size_t)row * kD + 4 * lane);
  if (pre_g) y = ln_row(y, pre_g, pre_b, lane);
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] = y[e] + scale * (acc[e] + bv[e]);
  if (ln_g) y = ln_row(y, ln_g, ln_b, lane);
  *reinterpret_cast<f32x4*>(out + (size_t)row * kD + 4 * lane) = y;
}

// kc / vc != nullptr (single-session streaming): the K and V thirds go straight to the session's cache rows (row m of
// the chunk -> kc + m*256) instead of qkv -- the separate append launch disappears
__global__ __launch_bounds__(kThreads) void k_ln_qkv(const float* __restrict__ x1, float* __restrict__ qkv, LayerW w, int M,
                                                     PadSkip ps, float* __restrict__ kc, float* __restrict__ vc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  float* bufA = smem;
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  const int c = blockIdx.y;  // 0: q, 1: k, 2: v
  BRing<1> ring;
  const f32x4* seg = w.wqkv + (size_t)(c * 8 + wave) * kTs256;
  ring_prime(ring, seg, 0);
  rb_load_rows(bufA, kLda, x1 + (size_t)r0 * kD, kRows, valid);
  rb_layernorm(bufA, bufA, kLda, kRows, w.ln_mha_g, w.ln_mha_b, 1e-5f);
  __syncthreads();
  f32x16 acc[1][1];
  acc_zero(acc);
  rb_gemm<1, 1, kG256>(bufA, kLda, seg, 0, nullptr, 0, ring, acc);
  const int col = c * 256 + wave * 32 + (lane & 31);
  const float bv = w.bqkv[col];
  float* cache = (c == 1) ? kc : (c == 2 ? vc : nullptr);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane);
    if (row >= valid) continue;
    if (cache) cache[(size_t)(r0 + row) * kD + wave * 32 + (lane & 31)] = acc[0][0][r] + bv;
    else qkv[(size_t)(r0 + row) * 768 + col] = acc[0][0][r] + bv;
  }
}

constexpr size_t kLdsConvPre = 4 * kRows * kLda * sizeof(float);  // (the depthwise window uses the three buffers + halo)
constexpr size_t kLdsFfnPart = 3 * kRows * kLda * sizeof(float);
constexpr size_t kLdsLnQkv = kRows * kLda * sizeof(float);
void launch_conv_pre(const float* g, const float* g_hist, const float* x2, float* x3, const LayerW& w, const int64_t* lens,
                     int M, int Tp, int ksize, int mask_mul, hipStream_t st, bool causal, const PadSkip& ps) {
  dim3 grid((M + kRows - 1) / kRows);
  const int left_ctx = causal ? ksize - 1 : (ksize - 1) / 2;
#define LAUNCH_CP(KS)                                                                                                    \
  if (g_hist)                                                                                                            \
    PPASR_LAUNCH((k_conv_pre<KS, true>), grid, dim3(kThreads), kLdsConvPre, st, g, g_hist, x2, x3, w, lens, M, Tp,   \
                       mask_mul, left_ctx, ps);                                                                          \
  else                                                                                                                   \
    PPASR_LAUNCH((k_conv_pre<KS, false>), grid, dim3(kThreads), kLdsConvPre, st, g, g_hist, x2, x3, w, lens, M, Tp,  \
                       mask_mul, left_ctx, ps);
  if (ksize == 15) {
    LAUNCH_CP(15)
  } else if (ksize == 31) {
    LAUNCH_CP(31)
  } else if (ksize == 7) {
    LAUNCH_CP(7)
  }
#undef LAUNCH_CP
}
void launch_ffn_split(const float* x, const float* ln_g, const float* ln_b, const f32x4* w1, const float* b1,
                      const f32x4* w2, const float* b2, float scale, const float* out_ln_g, const float* out_ln_b,
                      float* partial, float* out, int M, int n_chunks, int S, hipStream_t st, const PadSkip& ps,
                      bool residual_is_normed) {
  PPASR_LAUNCH(k_ffn_part, dim3((M + kRows - 1) / kRows, S), dim3(kThreads), kLdsFfnPart, st, x, ln_g, ln_b, w1, b1,
                     w2, partial, M, n_chunks, ps);
  PPASR_LAUNCH(k_ffn_join, dim3((M + 3) / 4), dim3(256), 0, st, x, partial, S, b2, scale, out_ln_g, out_ln_b, out, M,
                     ps, residual_is_normed ? ln_g : nullptr, residual_is_normed ? ln_b : nullptr);
}
void launch_ln_qkv(const float* x1, float* qkv, const LayerW& w, int M, hipStream_t st, const PadSkip& ps, float* kc,
                   float* vc) {
  PPASR_LAUNCH(k_ln_qkv, dim3((M + kRows - 1) / kRows, 3), dim3(kThreads), kLdsLnQkv, st, x1, qkv, w, M, ps, kc, vc);
}

// -------------------------------------------------------------------------------------
// Efficient-Conformer stride layer (StrideConformerEncoderLayer, efficient_conformer/encoder.py:455-548;
// strided ConvolutionModule, efficient_conformer/convolution.py:80-138): the causal depthwise conv has
// stride 2 (output frame j reads g frames 2j-(K-1)..2j), the residual goes through
// AvgPool1D(2, 2, ceil_mode, exclusive) (encoder.py:171-172,521-531), the pad mask is mask_pad[:, :, ::2].
// Rows of this kernel are OUTPUT rows (b, j), j < Ts = ceil(Tp/2); g and x2 are full-resolution.
// -------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(kThreads) void k_conv_ffn_stride(const float* __restrict__ g, const float* __restrict__ g_hist,
                                                              const float* __restrict__ x2, float* __restrict__ x_out,
                                                              LayerW w,
                                                              const int64_t* __restrict__ lens, int B, int Tp, int Ts,
                                                              int n_chunks, int mask_mul_out, PadSkip ps, int causal) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (pad_block_skippable(ps, blockIdx.x * kRows, kRows, B * Ts)) return;  // (ps describes the OUTPUT rows)
  float* bufX = smem;
  float* bufA = bufX + kRows * kLda;
  float* bufH = bufA + kRows * kLda;
  const int lane = lane_id(), wave = wave_id();
  const int Mo = B * Ts;
  const int r0 = blockIdx.x * kRows;
  const int valid = min(kRows, Mo - r0);
  const int col = wave * 32 + (lane & 31);
  constexpr int LO = KS - 1;
  BRing<1> ring;
  const f32x4* seg_pw2 = w.pw2 + (size_t)wave * kTs256;
  ring_prime(ring, seg_pw2, 0);
  {
    const f32x4 gp = *reinterpret_cast<const f32x4*>(w.glu_pad + 4 * lane);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(w.dw_b + 4 * lane);
    for (int row = wave; row < kRows; row += kWaves) {
      f32x4 out = bias;
      if (row < valid) {
        const int mo = r0 + row, b = mo / Ts, j = mo - b * Ts;
        const float* gb = g + (size_t)b * Tp * kD + 4 * lane;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
          // causal: left context KS-1, frames before the start = GLU(pointwise_conv1(0)) (the reference pads before
          // pointwise_conv1); non-causal: the depthwise conv itself zero-pads (KS-1)/2 frames on both sides
          const int f = 2 * j - (causal ? LO : LO / 2) + t;
          const f32x4 wj = *reinterpret_cast<const f32x4*>(w.dw_w + t * kD + 4 * lane);
          f32x4 v = causal ? gp : f32x4{0.f, 0.f, 0.f, 0.f};
          if (f >= 0 && f < Tp) v = *reinterpret_cast<const f32x4*>(gb + (size_t)f * kD);
          else if (f < 0 && g_hist) v = *reinterpret_cast<const f32x4*>(g_hist + (size_t)(LO + f) * kD + 4 * lane);  // streaming, B = 1
          out += wj * v;
        }
      }
      *reinterpret_cast<f32x4*>(bufA + row * kLda + 4 * lane) = out;
    }
  }
  __syncthreads();
  rb_layernorm<true>(bufA, bufA, kLda, kRows, w.ln_cm_g, w.ln_cm_b, w.cm_eps);
  __syncthreads();
  PadRows is_pad{lens, r0, Ts, Mo, mask_mul_out};
  {
    f32x16 acc[1][1];
    acc_zero(acc);
    rb_gemm<1, 1, kG256>(bufA, kLda, seg_pw2, 0, w.ff_w1 + (size_t)wave * kTs256, 0, ring, acc);
    const float bv = w.pw2_b[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = acc_row(r, lane);
      float v = 0.f;
      if (row < valid) {
        const int mo = r0 + row, b = mo / Ts, j = mo - b * Ts;
        const float* xb = x2 + ((size_t)b * Tp + 2 * j) * kD + col;
        float res = xb[0];
        if (2 * j + 1 < Tp) res = (res + xb[kD]) * 0.5f;  // exclusive average of the (possibly partial) window
        float c = is_pad(row) ? 0.f : acc[0][0][r] + bv;
        v = res + c;
      }
      bufX[row * kLda + col] = v;
    }
  }
  __syncthreads();
  rb_layernorm(bufX, bufA, kLda, kRows, w.ln_ff_g, w.ln_ff_b, 1e-5f);
  __syncthreads();
  f32x16 acc2[1][1];
  acc_zero(acc2);
  ffn_phase(bufA, bufH, w.ff_w1, w.ff_b1, w.ff_w2, n_chunks, nullptr, ring, acc2);
  residual_epilogue(bufX, acc2, w.ff_b2, 0.5f);
  __syncthreads();
  rb_layernorm(bufX, bufX, kLda, kRows, w.ln_fin_g, w.ln_fin_b, 1e-5f);
  rb_store_rows(x_out + (size_t)r0 * kD, bufX, kLda, kRows, valid);
}
void launch_conv_ffn_stride(const float* g, const float* g_hist, const float* x2, float* x_out, const LayerW& w,
                            const int64_t* lens, int B, int Tp, int Ts, int n_chunks, int ksize, int mask_mul_out,
                            hipStream_t st, const PadSkip& ps, bool causal) {
  dim3 grid((B * Ts + kRows - 1) / kRows);
  if (ksize == 15)
    PPASR_LAUNCH(k_conv_ffn_stride<15>, grid, dim3(kThreads), kLdsConvFfn, st, g, g_hist, x2, x_out, w, lens, B, Tp, Ts,
                       n_chunks, mask_mul_out, ps, causal ? 1 : 0);
  else if (ksize == 7)
    PPASR_LAUNCH(k_conv_ffn_stride<7>, grid, dim3(kThreads), kLdsConvFfn, st, g, g_hist, x2, x_out, w, lens, B, Tp, Ts,
                       n_chunks, mask_mul_out, ps, causal ? 1 : 0);
}

// -------------------------------------------------------------------------------------
// CTC head: after_norm (encoder.py:201-202) -> ctc_lo (loss/ctc.py:27) -> per-frame softmax
// statistics + argmax (loss/ctc.py:62-70, ctc_greedy_decoder.py:21-22) without materialising
// the [B,T',V] probability tensor.  Optional logits tap (LOGITS).
// Wave w walks vocabulary tiles w, w+8, ...; each lane keeps a running (max, sum-exp, argmax)
// for its 16 rows, merged across lanes / waves at the end (ties -> lowest index = numpy argmax).
// -------------------------------------------------------------------------------------
// H3: the vocabulary tiles on the fp16 x3 route (h3.h; hw.w is then the re-packed weight; the operand planes replace bufA
// and the reduction arrays move 512 B back)
template <bool LOGITS, bool H3>
__device__ __forceinline__ void ctc_head_body(const float* __restrict__ x, const HeadW& hw, float* __restrict__ logits,
                                              int32_t* __restrict__ fr_argmax, float* __restrict__ fr_maxprob,
                                              float* __restrict__ row_max, float* __restrict__ row_sum, int M, const PadSkip& ps,
                                              float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = pad_block_of(ps, kRows, M);  // (ragged batches: PadSkip::tab or the padded grid)
  if (blk < 0) return;
  // gridDim.y > 1 (under-filled launches): workgroup y walks vocabulary tiles wave + 8 (y + gridDim.y k) and leaves its
  // per-row (max, sum-exp, argmax) in part[3][gridDim.y][M]; k_ctc_merge combines the slices
  const int ny = gridDim.y, y = blockIdx.y;
  float* bufA = smem;                                          // [32][260]
  float* redM = bufA + (H3 ? kH3TileBytes / 4 : kRows * kLda);  // [8][32]
  float* redS = redM + kWaves * 32;                            // [8][32]
  int* redI = reinterpret_cast<int*>(redS + kWaves * 32);      // [8][32]
  const int lane = lane_id(), wave = wave_id();
  const int r0 = blk * kRows;
  const int valid = min(kRows, M - r0);
  const int V = hw.V;
  BRing<1> ring;
  if (wave + 8 * y < hw.n_tiles) ring_prime(ring, hw.w + (size_t)(wave + 8 * y) * kTs256, 0);
  rb_load_rows(bufA, kLda, x + (size_t)r0 * kD, kRows, valid);
  if (hw.ln_g) rb_layernorm(bufA, bufA, kLda, kRows, hw.ln_g, hw.ln_b, 1e-5f);
  __syncthreads();
  if constexpr (H3) h3_planes_from_tile(bufA, reinterpret_cast<_Float16*>(bufA));
  // Transposed tiles (rb_gemm SWAP): lane = row l&31, its 16 registers = 16 columns of the vocabulary tile in increasing
  // order (col = 8(r>>2) + 4(l>>5) + (r&3)).  The running (max, sum-exp, argmax) of a row is then ONE triple per lane,
  // updated per tile with in-lane arithmetic: tile max (v_max3), one rescale of the running sum, 16 exponentials, and
  // an index scan only when the tile raises the maximum (rare after the first tiles) -- against a triple per (row,
  // column lane) with two exponentials per element and a 5-step cross-lane merge per row at the end.
  float mx = -INFINITY, sm = 0.f;
  int ix = 0x7fffffff;
  const int l31 = lane & 31, hh = lane >> 5;
  constexpr float kLog2e = 1.4426950408889634f;
  const int tstep = kWaves * ny;
  for (int tile = wave + 8 * y; tile < hw.n_tiles; tile += tstep) {
    f32x16 acc[1][1];
    acc_zero(acc);
    const f32x4* seg = hw.w + (size_t)tile * kTs256;
    if constexpr (H3) {
      rb_gemm_h3(reinterpret_cast<const _Float16*>(bufA), seg, tile + tstep < hw.n_tiles ? seg + (size_t)tstep * kTs256 : nullptr,
                 ring, acc[0][0]);
      acc[0][0] *= kH3Inv;
    } else {
      rb_gemm<1, 1, kG256, kPF, NoSide, true>(bufA, kLda, seg, 0, tile + tstep < hw.n_tiles ? seg + (size_t)tstep * kTs256 : nullptr,
                                              0, ring, acc);
    }
    const int c0 = tile * 32 + 4 * hh;  // column of register 0
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(hw.b + c0 + 8 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * q + e] = acc[0][0][4 * q + e] + bq[e];
    }
    if (tile == hw.n_tiles - 1 && (V & 31)) {  // padded columns of the last tile never win and add exp(-inf) = 0
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + 8 * (r >> 2) + (r & 3) >= V) v[r] = -INFINITY;
    }
    if (LOGITS) {
      if (l31 < valid) {
        float* lrow = logits + (size_t)(r0 + l31) * V;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int col = c0 + 8 * (r >> 2) + (r & 3);
          if (col < V) lrow[col] = v[r];
        }
      }
    }
    float tmax = max3f(v[0], v[1], v[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = max3f(tmax, v[r], v[r + 1]);
    tmax = fmaxf(tmax, v[15]);
    if (tmax > mx) {  // first column holding the new maximum (lowest index wins ties: numpy argmax)
#pragma unroll
      for (int r = 15; r >= 0; --r) ix = (v[r] == tmax) ? c0 + 8 * (r >> 2) + (r & 3) : ix;
    }
    const float mn = fmaxf(mx, tmax);
    // (mx = -inf before the first tile: exp2(-inf) = 0; mn is finite from then on -- every tile has a real column)
    sm *= __builtin_amdgcn_exp2f((mx - mn) * kLog2e);
    f32x2 ps = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {  // (subtract first: logits reach +-30, a fused v * log2e - mn * log2e would round at 2e-6)
      const f32x2 t = (f32x2{v[r], v[r + 1]} - f32x2{mn, mn}) * f32x2{kLog2e, kLog2e};
      ps += f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    }
    sm += ps[0] + ps[1];
    mx = mn;
  }
  // the two lane halves of a row (different columns), then one triple per (wave, row)
  {
    const float m2 = __shfl_xor(mx, 32), s2 = __shfl_xor(sm, 32);
    const int i2 = __shfl_xor(ix, 32);
    const float mn = fmaxf(mx, m2);
    const float sa = (mx == -INFINITY) ? 0.f : sm * __expf(mx - mn);
    const float sb = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
    const bool take2 = (m2 > mx) || (m2 == mx && i2 < ix);
    if (hh == 0) {
      redM[wave * 32 + l31] = mn;
      redS[wave * 32 + l31] = sa + sb;
      redI[wave * 32 + l31] = take2 ? i2 : ix;
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int row = threadIdx.x;
    float m = redM[row], s = redS[row];
    int i = redI[row];
    for (int wv = 1; wv < kWaves; ++wv) {
      float m2 = redM[wv * 32 + row], s2 = redS[wv * 32 + row];
      int i2 = redI[wv * 32 + row];
      float mn = fmaxf(m, m2);
      float sa = (m == -INFINITY) ? 0.f : s * __expf(m - mn);
      float sb = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
      bool take2 = (m2 > m) || (m2 == m && i2 < i);
      i = take2 ? i2 : i;
      m = mn;
      s = sa + sb;
    }
    if (row < valid) {
      if (ny > 1) {
        part[(size_t)y * M + r0 + row] = m;
        part[((size_t)ny + y) * M + r0 + row] = s;
        reinterpret_cast<int*>(part)[((size_t)2 * ny + y) * M + r0 + row] = i;
      } else {
        if (fr_argmax) fr_argmax[r0 + row] = i;
        if (fr_maxprob) fr_maxprob[r0 + row] = 1.0f / s;
        if (row_max) row_max[r0 + row] = m;
        if (row_sum) row_sum[r0 + row] = s;
      }
    }
  }
}
__global__ void k_ctc_merge(const float* __restrict__ part, int ny, int32_t* __restrict__ fr_argmax,
                            float* __restrict__ fr_maxprob, float* __restrict__ row_max, float* __restrict__ row_sum, int M,
                            PadSkip ps) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= M) return;
  if (pad_block_skippable(ps, row & ~(kRows - 1), kRows, M)) return;
  float m = part[row], s = part[(size_t)ny * M + row];
  int i = reinterpret_cast<const int*>(part)[(size_t)2 * ny * M + row];
  for (int y = 1; y < ny; ++y) {
    const float m2 = part[(size_t)y * M + row], s2 = part[((size_t)ny + y) * M + row];
    const int i2 = reinterpret_cast<const int*>(part)[((size_t)2 * ny + y) * M + row];
    const float mn = fmaxf(m, m2);
    const float sa = (m == -INFINITY) ? 0.f : s * __expf(m - mn);
    const float sb = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
    const bool take2 = (m2 > m) || (m2 == m && i2 < i);
    i = take2 ? i2 : i;
    m = mn;
    s = sa + sb;
  }
  if (fr_argmax) fr_argmax[row] = i;
  if (fr_maxprob) fr_maxprob[row] = 1.0f / s;
  if (row_max) row_max[row] = m;
  if (row_sum) row_sum[row] = s;
}
template <bool LOGITS>
__global__ __launch_bounds__(kThreads) void k_ctc_head(const float* __restrict__ x, HeadW hw, float* __restrict__ logits,
                                                       int32_t* __restrict__ fr_argmax, float* __restrict__ fr_maxprob,
                                                       float* __restrict__ row_max, float* __restrict__ row_sum, int M,
                                                       PadSkip ps, float* __restrict__ part) {
  ctc_head_body<LOGITS, false>(x, hw, logits, fr_argmax, fr_maxprob, row_max, row_sum, M, ps, part);
}
template <bool LOGITS>
__global__ __launch_bounds__(kThreads) void k_ctc_head_h3(const float* __restrict__ x, HeadW hw, float* __restrict__ logits,
                                                          int32_t* __restrict__ fr_argmax, float* __restrict__ fr_maxprob,
                                                          float* __restrict__ row_max, float* __restrict__ row_sum, int M,
                                                          PadSkip ps, float* __restrict__ part) {
  ctc_head_body<LOGITS, true>(x, hw, logits, fr_argmax, fr_maxprob, row_max, row_sum, M, ps, part);
}
constexpr size_t kLdsCtc = (kRows * kLda + 3 * kWaves * 32) * sizeof(float);
void launch_ctc_head(const float* x, const HeadW& hw, float* logits, int32_t* fr_argmax, float* fr_maxprob, float* row_max,
                     float* row_sum, int M, hipStream_t st, const PadSkip& ps, int n_slices, float* part, bool h3) {
  const int ny = (n_slices > 1 && part) ? n_slices : 1;
  dim3 grid((M + kRows - 1) / kRows, ny);
  const size_t lds = ragged_lds(kLdsCtc + (h3 ? 512 : 0), ps, (int)(grid.x * grid.y));
  if (h3 && logits)  // (hw: the head's fp16 x3 view)
    PPASR_LAUNCH(k_ctc_head_h3<true>, grid, dim3(kThreads), lds, st, x, hw, logits, fr_argmax, fr_maxprob, row_max, row_sum, M, ps,
                 part);
  else if (h3)
    PPASR_LAUNCH(k_ctc_head_h3<false>, grid, dim3(kThreads), lds, st, x, hw, logits, fr_argmax, fr_maxprob, row_max, row_sum, M, ps,
                 part);
  else if (logits)
    PPASR_LAUNCH(k_ctc_head<true>, grid, dim3(kThreads), lds, st, x, hw, logits, fr_argmax, fr_maxprob, row_max,
                       row_sum, M, ps, part);
  else
    PPASR_LAUNCH(k_ctc_head<false>, grid, dim3(kThreads), lds, st, x, hw, logits, fr_argmax, fr_maxprob, row_max,
                       row_sum, M, ps, part);
  if (ny > 1)
    PPASR_LAUNCH(k_ctc_merge, dim3((M + 255) / 256), dim3(256), 0, st, part, ny, fr_argmax, fr_maxprob, row_max, row_sum,
                       M, ps);
}

// probs = softmax(logits) recomputed exactly (max, then exp(x-max)/sum) in place; one wave per row.
__global__ __launch_bounds__(256) void k_softmax_rows(float* __restrict__ p, int M, int V, PadSkip ps) {
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= M) return;
  // ragged batch: the CTC head skipped whole 32-row blocks; the same blocks keep their cleared (all-zero) rows here
  if (pad_block_skippable(ps, row & ~(kRows - 1), kRows, M)) return;
  const int lane = lane_id();
  float* x = p + (size_t)row * V;
  float m = -INFINITY;
  for (int c = lane; c < V; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < V; c += 64) {
    float e = expf(x[c] - m);
    x[c] = e;
    s += e;
  }
  s = wave_sum(s);
  for (int c = lane; c < V; c += 64) x[c] = x[c] / s;
}
void launch_softmax_from_stats(float* probs_inout, const float*, const float*, int M, int V, hipStream_t st,
                               const PadSkip& ps) {
  PPASR_LAUNCH(k_softmax_rows, dim3((M + 3) / 4), dim3(256), 0, st, probs_inout, M, V, ps);
}

__global__ __launch_bounds__(256) void k_zero_pad_rows(float* __restrict__ probs, float* __restrict__ logits,
                                                       int32_t* __restrict__ fa, float* __restrict__ fp,
                                                       const int64_t* __restrict__ lens, int M, int Tp, int mul, int V) {
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= M) return;
  const int b = row / Tp, t = row - b * Tp;
  if ((int64_t)mul * t < lens[b]) return;
  const int lane = lane_id();
  if (probs)
    for (int c = lane; c < V; c += 64) probs[(size_t)row * V + c] = 0.f;
  if (logits)
    for (int c = lane; c < V; c += 64) logits[(size_t)row * V + c] = 0.f;
  if (lane == 0) {
    if (fa) fa[row] = 0;
    if (fp) fp[row] = 0.f;
  }
}
void launch_zero_pad_rows(float* probs, float* logits, int32_t* fr_argmax, float* fr_maxprob, const int64_t* lens, int B,
                          int Tp, int mul, int V, hipStream_t st) {
  const int M = B * Tp;
  PPASR_LAUNCH(k_zero_pad_rows, dim3((M + 3) / 4), dim3(256), 0, st, probs, logits, fr_argmax, fr_maxprob, lens, M, Tp,
                     mul, V);
}

// =====================================================================================
// CTC greedy decode (decoders/ctc_greedy_decoder.py:6-31)
// =====================================================================================
// stage 1 from materialised probabilities: np.argmax(axis=1) (first max wins) + prob at argmax
__global__ __launch_bounds__(256) void k_frame_argmax(const float* __restrict__ probs, int32_t* __restrict__ fr_argmax,
                                                      float* __restrict__ fr_maxprob, int M, int V) {
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= M) return;
  const int lane = lane_id();
  const float* x = probs + (size_t)row * V;
  float m = -INFINITY;
  int idx = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    float v = x[c];
    if (v > m || idx == 0x7fffffff) {  // first element always taken (handles -inf / NaN-free inputs)
      m = v;
      idx = c;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float m2 = __shfl_xor(m, o);
    int i2 = __shfl_xor(idx, o);
    bool take2 = (i2 != 0x7fffffff) && (idx == 0x7fffffff || m2 > m || (m2 == m && i2 < idx));
    if (take2) {
      m = m2;
      idx = i2;
    }
  }
  if (lane == 0) {
    fr_argmax[row] = idx;
    fr_maxprob[row] = m;
  }
}
void launch_frame_argmax(const float* probs, int32_t* fr_argmax, float* fr_maxprob, int M, int V, hipStream_t st) {
  PPASR_LAUNCH(k_frame_argmax, dim3((M + 3) / 4), dim3(256), 0, st, probs, fr_argmax, fr_maxprob, M, V);
}

// stage 2: groupby-collapse, drop blank, score = mean(non-blank max probs)*100; one wave per utterance
__global__ __launch_bounds__(64) void k_ctc_collapse(const int32_t* __restrict__ fr_argmax, const float* __restrict__ fr_maxprob,
                                                     const int32_t* __restrict__ frame_lens, int Tp, int blank,
                                                     int32_t* __restrict__ tokens, int32_t* __restrict__ n_tokens,
                                                     double* __restrict__ score) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int n = frame_lens ? frame_lens[b] : Tp;
  n = max(0, min(n, Tp));
  const int32_t* ids = fr_argmax + (size_t)b * Tp;
  const float* pr = fr_maxprob + (size_t)b * Tp;
  int32_t* out = tokens + (size_t)b * Tp;
  int count = 0;
  int prev_last = -2;
  double dsum = 0.0;
  int nnb = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const bool in = i < n;
    const int id = in ? ids[i] : -3;
    int prev = __shfl_up(id, 1);
    if (lane == 0) prev = prev_last;
    const bool nonblank = in && id != blank;
    const bool keep = nonblank && id != prev;
    if (nonblank) {
      dsum += (double)pr[i];
      nnb += 1;
    }
    unsigned long long mask = __ballot(keep);
    int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
    if (keep) out[pos] = id;
    count += __popcll(mask);
    prev_last = __shfl(id, 63);
  }
  // deterministic tree reduction of the fp64 partial sums
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dsum += __shfl_xor(dsum, o);
    nnb += __shfl_xor(nnb, o);
  }
  for (int i = count + lane; i < Tp; i += 64) out[i] = -1;
  if (lane == 0) {
    n_tokens[b] = count;
    score[b] = nnb > 0 ? (dsum / (double)nnb) * 100.0 : 0.0;
  }
}
void launch_ctc_collapse(const int32_t* fr_argmax, const float* fr_maxprob, const int32_t* frame_lens, int B, int Tp,
                         int blank, int32_t* tokens, int32_t* n_tokens, double* score, hipStream_t st) {
  PPASR_LAUNCH(k_ctc_collapse, dim3(B), dim3(64), 0, st, fr_argmax, fr_maxprob, frame_lens, Tp, blank, tokens,
                     n_tokens, score);
}

hipError_t configure_kernels() {
  hipError_t e = configure_attention_kernels();
  if (e != hipSuccess) return e;
  e = configure_conformer_t_kernels();
  if (e != hipSuccess) return e;
  e = configure_front_fused_kernels();
  if (e != hipSuccess) return e;
#define SET_LDS(fn, bytes)                                                                                     \
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
  if (e != hipSuccess) return e;
  SET_LDS(k_ffn_qkv, kLdsFfnQkv);
  SET_LDS(k_ffn_qkv_h3, kLdsFfnQkv + kH3ExtraLds);
  SET_LDS((k_conv_ffn_h3<15, true>), kLdsConvFfn + kH3ExtraLds);
  SET_LDS((k_conv_ffn_h3<15, false>), kLdsConvFfn + kH3ExtraLds);
  SET_LDS((k_conv_ffn_h3<7, true>), kLdsConvFfn + kH3ExtraLds);
  SET_LDS((k_conv_ffn_h3<7, false>), kLdsConvFfn + kH3ExtraLds);
  SET_LDS(k_out_glu, kLdsOutGlu);
  SET_LDS(k_pw1_glu_cols, kLdsPw1Cols);
  SET_LDS((k_conv_ffn<15, false, false>), kLdsConvFfn);
  SET_LDS((k_conv_ffn<31, false, false>), kLdsConvFfn);
  SET_LDS((k_conv_ffn<7, false, false>), kLdsConvFfn);
  SET_LDS((k_conv_ffn<15, false, true>), kLdsConvFfn);
  SET_LDS((k_conv_ffn<31, false, true>), kLdsConvFfn);
  SET_LDS((k_conv_ffn<7, false, true>), kLdsConvFfn);
  SET_LDS((k_conv_ffn<15, true, false>), kLdsConvFfn);
  SET_LDS((k_conv_ffn<31, true, false>), kLdsConvFfn);
  SET_LDS((k_conv_ffn<7, true, false>), kLdsConvFfn);
  SET_LDS(k_pw1_glu, kLdsPw1Glu);
  SET_LDS((k_conv_pre<15, false>), kLdsConvPre);
  SET_LDS((k_conv_pre<31, false>), kLdsConvPre);
  SET_LDS((k_conv_pre<7, false>), kLdsConvPre);
  SET_LDS((k_conv_pre<15, true>), kLdsConvPre);
  SET_LDS((k_conv_pre<31, true>), kLdsConvPre);
  SET_LDS((k_conv_pre<7, true>), kLdsConvPre);
  SET_LDS(k_ffn_part, kLdsFfnPart);
  SET_LDS(k_attn_out_glu, kLdsAttnOutGlu);
  SET_LDS(k_attn_out_glu_h3, kLdsAttnOutGlu);
  SET_LDS(k_conv_ffn_stride<15>, kLdsConvFfn);
  SET_LDS(k_conv_ffn_stride<7>, kLdsConvFfn);
  SET_LDS(k_ctc_head<true>, kLdsExclusive);  // (>= kLdsCtc: see ragged_lds)
  SET_LDS(k_ctc_head_h3<true>, kLdsExclusive);
  SET_LDS(k_ctc_head_h3<false>, kLdsExclusive);
  SET_LDS(k_ctc_head<false>, kLdsExclusive);
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
