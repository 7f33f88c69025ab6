// rowblock.h -- device-side toolkit shared by the gfx950 kernels.
//
// Execution model ("row-block"): a 512-thread workgroup (8 wave64, two per SIMD) owns
// 32*MT consecutive rows of the flattened [B*T', d] activation matrix and keeps them in
// LDS across a whole chain of dense layers.  Dense contractions run on the exact-fp32
// matrix core (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain):
//   * the A operand (activations) is read from LDS with one ds_read_b128 per 4 MFMAs;
//     rows are padded by 4 floats so the 16-lane b128 groups hit 16 distinct 16-B slots;
//   * the B operand (weights) is streamed from global/L2 straight into VGPRs in a layout
//     pre-packed on the host in MFMA fragment order, so every load is a fully coalesced
//     1 KiB global_load_dwordx4 per wave and needs no LDS staging (each wave owns its own
//     32 output columns, so B is not shared between waves);
//   * a PF-deep register ring prefetches B PF k-groups ahead and is carried ACROSS calls
//     (each call names the segment the stream continues with), so the weight stream never
//     drains at GEMM / phase boundaries or barriers;
//   * a "side" functor is invoked once per k-group so VALU epilogue work of the previous
//     tile can be interleaved with the MFMAs of the current one (separate pipes).
//
// Fragment maps (MI355X guide §3): A lane l holds A[i=l&31][k=l>>5]; B lane l holds
// B[k=l>>5][j=l&31]; C/D lane l, reg r holds D[(r&3)+8*(r>>2)+4*(l>>5)][l&31].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));  // v_pk_{add,mul,fma}_f32 operands

// max(a, b, c) in ONE VALU instruction.  fmaxf() on MFMA results also emits a canonicalising v_max_f32 x, x per operand
// (IEEE sNaN quieting); in an MFMA-bound kernel every VALU instruction costs matrix-pipe issue cycles.
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float d;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

namespace ppasr {

constexpr int kD = 256;        // model width the kernels are specialised for
constexpr int kLda = kD + 4;   // LDS row stride (floats) of a [rows][256] activation buffer
constexpr int kRows = 32;      // rows per MFMA row tile
constexpr int kWaves = 8;      // waves per row-block workgroup
constexpr int kThreads = 64 * kWaves;
constexpr int kW16 = 1032;     // form id of "32 rows on 16 waves" (rbt.h; the 8-wave forms are named by their row count)
constexpr int kG256 = kD / 8;  // k-groups of a K=256 contraction
constexpr int kTs256 = kG256 * 64;  // packed tile stride (f32x4 units) of a K=256 weight

// Optional per-phase time stamps (tools/phase_ts.py builds a second copy of the library with -DPPASR_PHASE_TS):
// thread 0 of one workgroup in the middle of the grid records the 100 MHz wall clock at phase boundaries.
#ifdef PPASR_PHASE_TS
static __device__ long long g_phase_ts[128];  // one copy per translation unit (no -fgpu-rdc); [64..128): shader clock
#define PPASR_TS(i)                                                                              \
  do {                                                                                           \
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {                                       \
      g_phase_ts[i] = (long long)wall_clock64();                                                 \
      g_phase_ts[64 + (i)] = (long long)clock64();                                               \
    }                                                                                            \
  } while (0)
static __device__ long long g_wave_ts[512];  // [slot < 64][wave < 8]: per-wave stamps of one workgroup (PPASR_WAVE_TS)
#define PPASR_WAVE_TS(slot)                                                                          \
  do {                                                                                               \
    if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0)                                      \
      g_wave_ts[(slot) * 8 + (threadIdx.x >> 6)] = (long long)wall_clock64();                        \
  } while (0)
static __device__ long long g_wg_ts[2 * 1024];  // [kernel 0/1][workgroup < 256][start, end] of the last launch that records them (slot 0/1, 512/513)
#define PPASR_WG_TS(slot)                                                                                  \
  do {                                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < 512) g_wg_ts[((slot) & ~1) + 2 * blockIdx.x + ((slot) & 1)] = (long long)wall_clock64(); \
  } while (0)
#else
#define PPASR_TS(i) do { } while (0)
#define PPASR_WAVE_TS(slot) do { } while (0)
#define PPASR_WG_TS(slot) do { } while (0)
#endif

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// wave-uniform by construction; readfirstlane tells the compiler so (everything derived from it -- weight segment
// pointers, buffer-load offsets -- then lives in SGPRs instead of per-lane VALU arithmetic / waterfall loops)
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// Optional skipping of padding rows of a ragged batch (ppasr_set_skip_padding): utterance b occupies `Tp` time steps
// of `unit` rows each, time step t is valid iff mul*t < lens[b], and only the first need(b) = min(Tp, valid + slack)
// time steps are computed -- `slack` covers what valid outputs read from the rows behind them (right context of a
// non-causal conv module, the stride / time-reduction layers, the 3-frame groups of grouped attention).  A workgroup
// whose rows all lie behind need(b) returns at once; its outputs keep the zeros the workspace was cleared to.
struct PadSkip {
  const int64_t* lens = nullptr;  // nullptr: every row is computed (the reference's behaviour)
  int Tp = 0, mul = 4, slack = 0, unit = 1;
  // optional list of the ACTIVE row blocks of this launch's block size (k_block_table: tab[0] = their number, tab[1 + i] =
  // index of the i-th one): workgroup i then takes block tab[1 + i] and the workgroups behind the list exit.  The padded
  // grid with early exits is as fast on an otherwise idle chip (the dispatcher refills a CU as soon as a workgroup exits),
  // but workgroups are dealt to the shader engines by INDEX: with a few CUs taken by another stream's long-running kernel
  // (the beam search of the previous batch) an engine that happens to be dealt more active blocks than it has free CUs
  // runs a second round -- tools/cu_mask_probe.hip: 213 active of 372 blocks beside 16 foreign workgroups 1634 us, the same
  // blocks as the first 213 workgroups of the grid 832 us
  const int* tab = nullptr;
};
__device__ __forceinline__ int pad_need_steps(const PadSkip& s, int b) {
  const long long len = s.lens[b];
  const long long v = (len > 0 ? (len + s.mul - 1) / s.mul : 0) + s.slack;
  return (int)(v < (long long)s.Tp ? v : (long long)s.Tp);
}
// rows [m0, m0 + n_rows) of the flattened [B * Tp * unit] row space
__device__ __forceinline__ bool pad_block_skippable(const PadSkip& s, int m0, int n_rows, int M) {
  if (!s.lens) return false;
  const int m1 = min(m0 + n_rows, M) - 1;
  if (m1 < m0) return false;
  const int stride = s.Tp * s.unit;
  const int b1 = m1 / stride;
  for (int b = m0 / stride; b <= b1; ++b) {
    const int first = max(m0 - b * stride, 0);
    if (first < pad_need_steps(s, b) * s.unit) return false;
  }
  return true;
}

// row block (of R rows) this workgroup of a ragged launch works on, or -1: none
__device__ __forceinline__ int pad_block_of(const PadSkip& s, int R, int M) {
  if (s.tab) return (int)blockIdx.x < s.tab[0] ? s.tab[1 + blockIdx.x] : -1;
  return pad_block_skippable(s, blockIdx.x * R, R, M) ? -1 : (int)blockIdx.x;
}

// row of accumulator register r inside a 32x32 tile for this lane
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Sum over the 64 lanes, returned to every lane.  DPP row shifts / row broadcasts (VALU adds with a lane-permuting
// operand, a few cycles each) instead of the __shfl_xor butterfly, which compiles to ds_bpermute = one LDS crossbar
// round trip per step; the total lands in lane 63 and is read back as a wave-uniform value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or_zero(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_or_zero<0x111, 0xf>(v);  // row_shr:1
  v += dpp_or_zero<0x112, 0xf>(v);  // row_shr:2
  v += dpp_or_zero<0x114, 0xf>(v);  // row_shr:4
  v += dpp_or_zero<0x118, 0xf>(v);  // row_shr:8
  v += dpp_or_zero<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
  v += dpp_or_zero<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// sigmoid / swish on the hardware exp + rcp (about 1 ulp each; well inside the 1e-3 logit budget)
__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf(float x) { return x * sigmoidf(x); }
// sigmoid of two values, packed like swish2
__device__ __forceinline__ f32x2 sigmoid2(f32x2 x) {
  const f32x2 t = x * f32x2{-1.4426950408889634f, -1.4426950408889634f};
  f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  e += f32x2{1.0f, 1.0f};
  return f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
}
// swish of two values with the plain multiplies / adds as packed instructions (same operations, same roundings as
// swishf: x * rcp(1 + exp2(x * -log2 e))): 4 v_pk_* + 4 transcendentals per pair instead of 8 + 4
__device__ __forceinline__ f32x2 swish2(f32x2 x) {
  const f32x2 t = x * f32x2{-1.4426950408889634f, -1.4426950408889634f};
  f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  e += f32x2{1.0f, 1.0f};
  return x * f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
}

template <int MT, int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[MT][NT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
}

// ---- weight stream ring -----------------------------------------------------------------
constexpr int kPF = 4;  // weight-stream prefetch depth (k-groups = 1 KiB loads in flight per wave)
template <int NT, int PF = kPF>
struct BRing {
  f32x4 q[PF][NT];
};

// The weight stream uses BUFFER loads (buffer_load_dwordx4: SGPR resource = the wave's segment base, one constant
// VGPR offset = lane * 16, the k-group offset in an SGPR), not global_load_dwordx4 with a 64-bit per-lane address:
// tools/microbench_mfma.hip -- the same loop (LDS A operand, 1 KiB of B per wave per 4 MFMAs, two waves per SIMD)
// reaches 97.7 % of the fp32-MFMA peak with buffer loads and 90 % with global loads, whatever the prefetch depth or
// cache policy (the per-load 64-bit address arithmetic and address-register traffic of the global form steal issue
// slots from the MFMA stream).  The segment base is made wave-uniform with readfirstlane so that the resource lives
// in SGPRs without a waterfall loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wstream_rsrc(const void* p) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
// 16 bytes per lane at (resource base) + voff (per-lane bytes) + soff (wave-uniform bytes)
__device__ __forceinline__ f32x4 wstream_load(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

// fill the ring with k-groups 0..PF-1 of the segment starting at bp (n-tile nt at bp + nt*tile_stride)
template <int NT, int PF>
__device__ __forceinline__ void ring_prime(BRing<NT, PF>& ring, const f32x4* __restrict__ bp, int tile_stride) {
  const __amdgpu_buffer_rsrc_t rs = wstream_rsrc(bp);
  const int voff = lane_id() * 16;
#pragma unroll
  for (int s = 0; s < PF; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ring.q[s][nt] = wstream_load(rs, voff, (nt * tile_stride + s * 64) * 16);
}

struct NoSide {
  __device__ __forceinline__ void operator()(int) const {}
};

// acc[mt][nt] += A[32*MT x 8*G] * Bpacked, G k-groups (compile-time, multiple of PF).
//   a_lds : LDS, row 0 / k 0 of the A block, row stride lda floats (lda % 64 == 4)
//   bp    : packed weights of THIS call, positioned at (first n-tile, first k-group); the
//           ring must already hold its first PF k-groups (ring_prime or a previous call's nxt)
//   nxt   : segment the stream continues with after this call (nullptr: stream ends); must
//           have the same NT; its n-tile stride is nxt_stride
//   side(g): called once per k-group, after that group's MFMAs were issued
//   SWAP : issue the MFMAs with the operands exchanged (weights as A, activations as B): the accumulator then holds the
//          TRANSPOSED tile -- lane = row (lane & 31), register r = column (r&3) + 8(r>>2) + 4(lane>>5) of the wave's 32 --
//          so that 4 consecutive columns of a row sit in one register quad (16-byte LDS / global stores)
template <int MT, int NT, int G, int PF = kPF, typename Side = NoSide, bool SWAP = false>
__device__ __forceinline__ void rb_gemm(const float* a_lds, int lda, const f32x4* __restrict__ bp, int tile_stride,
                                        const f32x4* __restrict__ nxt, int nxt_stride, BRing<NT, PF>& ring,
                                        f32x16 (&acc)[MT][NT], Side side = Side()) {
  static_assert(G % PF == 0, "k-groups must be a multiple of the ring depth");
  const int lane = lane_id();
  const float* a_ptr = a_lds + (lane & 31) * lda + 4 * (lane >> 5);
  const __amdgpu_buffer_rsrc_t rs_b = wstream_rsrc(bp), rs_n = wstream_rsrc(nxt);
  const int voff = lane * 16;
  f32x4 a_cur[MT], a_nxt[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) a_cur[mt] = *reinterpret_cast<const f32x4*>(a_ptr + mt * 32 * lda);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int s = g % PF;
    // software pipeline: LDS read of the next k-group is issued before this group's MFMAs
    if (g + 1 < G) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = *reinterpret_cast<const f32x4*>(a_ptr + mt * 32 * lda + 8 * (g + 1));
    }
    f32x4 b[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = ring.q[s][nt];
    if (g + PF < G) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) ring.q[s][nt] = wstream_load(rs_b, voff, (nt * tile_stride + (g + PF) * 64) * 16);
    } else if (nxt) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) ring.q[s][nt] = wstream_load(rs_n, voff, (nt * nxt_stride + (g + PF - G) * 64) * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b[nt][j], a_cur[mt][j], acc[mt][nt], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
    side(g);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
    // keep the unrolled k-groups in program order: bounds live ranges (the scheduler would otherwise
    // hoist every LDS read / weight load of the whole tile to the top and spill)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// LayerNorm over the 256 columns of LDS rows (biased variance, eps inside the sqrt),
// nn.LayerNorm semantics (utils/base.py:7-21).  Each wave normalises rows w, w+8, ...;
// one ds_read_b128 per lane covers a whole row.  src == dst is allowed.
// If zero_row(row) is true the output row is forced to 0 (conv-module pad masking);
// POST is applied elementwise after the affine (identity or swish).
struct NoZero {
  __device__ __forceinline__ bool operator()(int) const { return false; }
};
// LayerNorm over the 256 columns of LDS rows: see the comment block above ln_rows_inreg.
// One row of 256 values held as one f32x4 per lane: LayerNorm (+ optional swish) of RN such rows at once.  The RN
// reductions are independent dependency chains (6 DPP adds + a readlane each), written side by side so that the
// scheduler interleaves them; one row at a time the phase is a chain of ~25 dependent cross-lane steps per row
// (measured 1.6 us per 32-row LayerNorm phase, per-phase stamps of tools/phase_ts.py).
template <bool SWISH, int RN>
__device__ __forceinline__ void ln_rows_inreg(f32x4 (&x)[RN], const f32x4& g, const f32x4& b, float eps) {
  if (eps < 0.f) {  // folded BatchNorm (conv module, cnn_module_norm: batch_norm): per-channel scale / shift, no statistics
#pragma unroll
    for (int i = 0; i < RN; ++i) {
      x[i] = x[i] * g + b;
      if (SWISH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[i][e] = swishf(x[i][e]);
      }
    }
    return;
  }
  float mean[RN], var[RN];
#pragma unroll
  for (int i = 0; i < RN; ++i) {
#ifdef PPASR_ABLATE_LN
    mean[i] = 0.f;
#else
    mean[i] = wave_sum(x[i][0] + x[i][1] + x[i][2] + x[i][3]) * (1.0f / kD);
#endif
  }
#pragma unroll
  for (int i = 0; i < RN; ++i) {
    x[i] = x[i] - mean[i];
#ifdef PPASR_ABLATE_LN
    var[i] = 1.f;
#else
    var[i] = wave_sum(x[i][0] * x[i][0] + x[i][1] * x[i][1] + x[i][2] * x[i][2] + x[i][3] * x[i][3]) * (1.0f / kD);
#endif
  }
#pragma unroll
  for (int i = 0; i < RN; ++i) {
    const float rstd = 1.0f / sqrtf(var[i] + eps);
    x[i] = x[i] * rstd * g + b;
    if (SWISH) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[i][e] = swishf(x[i][e]);
    }
  }
}

template <bool SWISH = false, typename ZeroRow = NoZero>
__device__ __forceinline__ void rb_layernorm(const float* src, float* dst, int lda, int nrows,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float eps, ZeroRow zero_row = ZeroRow()) {
  const int lane = lane_id();
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * lane);
  const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * lane);
  constexpr int RN = kRows / kWaves;
  if (nrows == kRows) {  // the row-block case: this wave's RN rows (w, w + 8, ...) side by side
    f32x4 x[RN];
#pragma unroll
    for (int i = 0; i < RN; ++i) x[i] = *reinterpret_cast<const f32x4*>(src + (wave_id() + i * kWaves) * lda + 4 * lane);
    ln_rows_inreg<SWISH, RN>(x, g, b, eps);
#pragma unroll
    for (int i = 0; i < RN; ++i) {
      const int row = wave_id() + i * kWaves;
      if (zero_row(row)) x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(dst + row * lda + 4 * lane) = x[i];
    }
    return;
  }
  for (int row = wave_id(); row < nrows; row += kWaves) {
    f32x4 x[1];
    x[0] = *reinterpret_cast<const f32x4*>(src + row * lda + 4 * lane);
    ln_rows_inreg<SWISH, 1>(x, g, b, eps);
    if (zero_row(row)) x[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(dst + row * lda + 4 * lane) = x[0];
  }
}

// copy nrows x 256 floats global(row stride 256) -> LDS(row stride lda); rows >= valid are zeroed
__device__ __forceinline__ void rb_load_rows(float* dst, int lda, const float* __restrict__ src, int nrows, int valid) {
  const int lane = lane_id();
  for (int row = wave_id(); row < nrows; row += kWaves) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < valid) v = *reinterpret_cast<const f32x4*>(src + (size_t)row * kD + 4 * lane);
    *reinterpret_cast<f32x4*>(dst + row * lda + 4 * lane) = v;
  }
}

// LDS(row stride lda) -> global(row stride 256) for rows < valid
__device__ __forceinline__ void rb_store_rows(float* __restrict__ dst, const float* src, int lda, int nrows, int valid) {
  const int lane = lane_id();
  for (int row = wave_id(); row < nrows && row < valid; row += kWaves)
    *reinterpret_cast<f32x4*>(dst + (size_t)row * kD + 4 * lane) =
        *reinterpret_cast<const f32x4*>(src + row * lda + 4 * lane);
}

}  // namespace ppasr
