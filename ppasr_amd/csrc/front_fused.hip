// front_fused.hip -- Conv2dSubsampling4's two convolutions as ONE launch (conformer/subsampling.py:84-88):
//   y2[b][t'][f2][c2] = relu(b2 + sum_{kh,kw,c} W2[c2][c][kh][kw] * y1[b][2t'+kh][2f2+kw][c]),
//   y1[b][t1][f1][c]  = relu(b1 + sum_{i,j} W1[c][i][j] * cmvn(x)[b][2t1+i][2f1+j]).
// k_conv1 wrote y1 to HBM (B*T1*F1*256 floats: 638 MB for 32 x 10 s, 1.27 GB for cfg4's 64 x 10 s -- at 5 - 6 TB/s the
// largest HBM stream of a step) and conv2 (k_gemm_stream<.., Conv2Src>) gathered its implicit-GEMM A rows from it.  Here
// the A tile of chunk kc + 1 (tap (kh, kw), 128 input channels) is COMPUTED from the features while the matrix pipe works
// on chunk kc: the tile's features (<= 8 output frames x 7 input frames x F, normalised once) sit in LDS, every thread
// owns 4 input channels (its 9 + 1 conv1 weight quads are re-requested per chunk: L1 hits) and 2 MT rows, and the 36
// multiply-adds of a row quad are sliced into the k-groups of the running GEMM (rb_gemm Side) -- vector-ALU work in the
// shadow of the MFMAs.  Each y1 element is recomputed for the ~2.25 output positions x that read it: +2.5 % FLOPs on the
// vector ALU, no y1 traffic at all, one launch less.
// Same arithmetic as k_conv1 followed by k_gemm_stream (same fmaf chain per y1 element, same MFMA order): bit-identical.
#include "conformer_kernels.h"
#include "launch.h"

namespace ppasr {

template <int MT>
__global__ __launch_bounds__(kThreads) void k_conv12(const float* __restrict__ feats, FrontW fw, float* __restrict__ out,
                                                     int T, int F, int Tp, int F2, int M, int m0, PadSkip ps,
                                                     const int* __restrict__ tile_tab) {
  constexpr int BM = 32 * MT, KC = 128, LD = KC + 4, G = KC / 8, NL = 2 * MT, STEP = G / NL, N_CHUNKS = 18;
  // tile -> rows: as k_gemm_stream (ragged batches: the t-th ACTIVE tile of the table, cut per utterance)
  int r0 = m0 + blockIdx.x * BM, Mlim = M;
  if (tile_tab) {
    const int t = blockIdx.x, nb = tile_tab[0];
    const int* pre = tile_tab + 1;
    if (t >= pre[nb]) return;
    int lo = 0, hi = nb;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pre[mid] <= t) lo = mid;
      else hi = mid;
    }
    const int S = ps.Tp * ps.unit;
    r0 = lo * S + (t - pre[lo]) * BM;
    Mlim = min(M, (lo + 1) * S);
  } else if (pad_block_skippable(ps, r0, BM, M)) {
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem + 2 * BM * LD;  // [output frame of the tile][7 input frames][F] normalised features
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  constexpr int tile_stride = N_CHUNKS * G * 64;
  const f32x4* wbase = fw.conv2_w + (size_t)wave * tile_stride;
  BRing<1> ring;
  ring_prime(ring, wbase, 0);
  // ---- the tile's features: output frames bt0 .. bt0 + nbt - 1 of the flattened [B][Tp] frame space, input frames
  // 4 t' .. 4 t' + 6 each (conv1 frame 2 t' + kh reads input frames 4 t' + 2 kh + i) ----
  const int bt0 = r0 / F2, nbt = (min(r0 + BM, Mlim) - 1) / F2 - bt0 + 1;
  for (int idx = tid; idx < nbt * 7 * F; idx += kThreads) {
    const int p = idx / F, f = idx - p * F;
    const int btl = p / 7, fr = p - 7 * btl;
    const int bt = bt0 + btl, b = bt / Tp, tp = bt - b * Tp;
    const int t = min(4 * tp + fr, T - 1);  // (always < T for rows < M)
    xs[idx] = (feats[((size_t)b * T + t) * F + f] - fw.cmvn_mean[f]) * fw.cmvn_istd[f];
  }
  // ---- this thread's rows (2 MT of them, 16 apart) and input-channel quad ----
  const int c4 = tid & 31, rbase = tid >> 5;
  // the rows' windows inside xs, two 16-bit offsets per register (0xffff: a zero row; the 128-row kernel is 2 registers
  // short of spilling)
  uint32_t pb[MT];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int m = r0 + rbase + 16 * i;
    const int bt = m / F2, f2 = m - bt * F2;
    const uint32_t v = m < Mlim ? (uint32_t)((bt - bt0) * 7 * F + 4 * f2) : 0xffffu;
    pb[i >> 1] = (i & 1) ? (pb[i >> 1] | (v << 16)) : v;
  }
  auto pbase = [&](int i) { return (int)((i & 1) ? (pb[i >> 1] >> 16) : (pb[i >> 1] & 0xffffu)); };
  const int lds_off0 = rbase * LD + 4 * c4;  // row i of this thread: + i * 16 * LD
  f32x4 wv[9], bv;
  float xv[9];
  auto load_w = [&](int kc) {  // conv1 weights of the 4 channels this thread produces for chunk kc
    const int c = (kc & 1) * 128 + 4 * c4;
#pragma unroll
    for (int j = 0; j < 9; ++j) wv[j] = *reinterpret_cast<const f32x4*>(fw.conv1_w + j * 256 + c);
    bv = *reinterpret_cast<const f32x4*>(fw.conv1_b + c);
  };
  auto read_x = [&](int i, int toff) {  // toff = 2 kh F + 2 kw: the tap's corner inside a row's window
    const int o = pbase(i);
    const float* p = xs + (o == 0xffff ? 0 : o) + toff;
#pragma unroll
    for (int ii = 0; ii < 3; ++ii)
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) xv[ii * 3 + jj] = p[ii * F + jj];
  };
  // k_conv1's arithmetic for (row i, 4 channels), written straight into the A buffer of the NEXT chunk (free since the
  // barrier that ended the previous chunk: nobody reads it during this chunk's GEMM)
  auto fma_row = [&](int i, float* buf) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = fmaf(wv[j][e], xv[j], a[e]);
    a += bv;
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = pbase(i) != 0xffff ? fmaxf(a[e], 0.f) : 0.f;
    *reinterpret_cast<f32x4*>(buf + lds_off0 + i * 16 * LD) = a;
  };
  auto tap_off = [&](int kc) {
    const int tap = kc >> 1, kh = tap / 3, kw = tap - 3 * kh;
    return 2 * kh * F + 2 * kw;
  };
  f32x16 acc[MT][1];
  acc_zero(acc);
  load_w(0);
  __syncthreads();  // xs complete
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    read_x(i, 0);
    fma_row(i, smem);
  }
  __syncthreads();
  for (int kc = 0; kc < N_CHUNKS; ++kc) {
    float* cur = smem + (kc & 1) * BM * LD;
    float* nxt = smem + ((kc + 1) & 1) * BM * LD;
    const f32x4* seg = wbase + (size_t)kc * G * 64;
    if (kc + 1 < N_CHUNKS) {
      load_w(kc + 1);
      const int toff = tap_off(kc + 1);
      // row i of the next A tile: window read during k-group STEP i, multiply-adds during k-group STEP i + 1
      auto side = [&](int g) {
        const int i = g / STEP, ph = g - i * STEP;
        if (i < NL && ph == 0) read_x(i, toff);
        if (i < NL && ph == 1) fma_row(i, nxt);
      };
      rb_gemm<MT, 1, G, kPF, decltype(side)>(cur, LD, seg, 0, seg + G * 64, 0, ring, acc, side);
    } else {
      rb_gemm<MT, 1, G>(cur, LD, seg, 0, nullptr, 0, ring, acc);
    }
    __syncthreads();
  }
  const int col = wave * 32 + (lane & 31);
  const float b2 = fw.conv2_b[col];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = r0 + mt * 32 + acc_row(r, lane);
      const float v = fmaxf((acc[mt][0][r] + b2) * 1.0f, 0.f);
      if (m < Mlim) out[(size_t)m * 256 + col] = v;
    }
}

static size_t conv12_lds(int mt, int F, int F2) {
  const int bm = 32 * mt, nbt = (bm - 1) / F2 + 2;
  return ((size_t)2 * bm * 132 + (size_t)nbt * 7 * F) * sizeof(float);
}
bool conv12_supported(const FrontW& fw, int F, int F2) {
  // Conv2dSubsampling4 only (3x3 / 2 twice), and the 128-row tile's features + A double buffer must fit the CU's LDS
  return fw.conv2_k == 3 && fw.conv2_s == 2 && F2 >= 1 && conv12_lds(4, F, F2) <= 160 * 1024 &&
         (size_t)((127 / F2) + 2) * 7 * F < 0xffff;  // (16-bit window offsets)
}

// tile_prefix_launch: the ragged launch's tile table (front_kernels.hip k_tile_prefix)
void launch_tile_prefix(const PadSkip& ps, int B, int BM, int* tab, hipStream_t st);

void launch_conv12(const float* feats, const FrontW& fw, float* y2, int B, int T, int F, int Tp, int F2, hipStream_t st,
                   const PadSkip& ps_frames, int* tile_scratch) {
  PadSkip ps = ps_frames;
  ps.unit = F2;  // rows are (frame, f2) pairs
  const int M = B * Tp * F2;
  constexpr int kCUs = 256;
  const int* no_tab = nullptr;
#define CONV12(MTA, GRID, M0, TAB)                                                                                      \
  PPASR_LAUNCH(k_conv12<MTA>, dim3(GRID), dim3(kThreads), conv12_lds(MTA, F, F2), st, feats, fw, y2, T, F, Tp, F2, M, M0, \
               ps, TAB)
  // (the same cut of the row space into launches as launch_conv_stage: whole rounds of 128-row tiles, the remainder
  //  re-cut into <= 256 shorter tiles; ragged batches: the active tiles in front of one grid)
  if (ps.lens && tile_scratch && M > 128 * kCUs) {
    launch_tile_prefix(ps, B, 128, tile_scratch, st);
    const int per_utt = (Tp * F2 + 127) / 128;
    CONV12(4, B * per_utt, 0, (const int*)tile_scratch);
    return;
  }
  const int tiles4 = (M + 127) / 128;
  const int full = (tiles4 / kCUs) * kCUs;
  const int rem_rows = M - full * 128;
  const int mt_rem = (rem_rows + 32 * kCUs - 1) / (32 * kCUs);
  if (full == 0) {
    const int mt = (M + 32 * kCUs - 1) / (32 * kCUs);
    if (mt <= 1) CONV12(1, (M + 31) / 32, 0, no_tab);
    else if (mt == 2) CONV12(2, (M + 63) / 64, 0, no_tab);
    else if (mt == 3) CONV12(3, (M + 95) / 96, 0, no_tab);
    else CONV12(4, (M + 127) / 128, 0, no_tab);
    return;
  }
  if (rem_rows <= 0 || mt_rem >= 4) {
    CONV12(4, tiles4, 0, no_tab);
    return;
  }
  CONV12(4, full, 0, no_tab);
  const int m0 = full * 128;
  if (mt_rem <= 1) CONV12(1, (rem_rows + 31) / 32, m0, no_tab);
  else if (mt_rem == 2) CONV12(2, (rem_rows + 63) / 64, m0, no_tab);
  else CONV12(3, (rem_rows + 95) / 96, m0, no_tab);
#undef CONV12
}

hipError_t configure_front_fused_kernels() {
  hipError_t e;
#define SET_LDS(fn)                                                                                              \
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  if (e != hipSuccess) return e;
  SET_LDS(k_conv12<1>);
  SET_LDS(k_conv12<2>);
  SET_LDS(k_conv12<3>);
  SET_LDS(k_conv12<4>);
#undef SET_LDS
  return hipSuccess;
}

}  // namespace ppasr
