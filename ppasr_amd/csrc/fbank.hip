// fbank.hip -- Kaldi-compatible log-mel filterbank front-end on the GPU (the step BEFORE the
// encoder).  Replaces AudioFeaturizer.featurize (ppasr/data_utils/featurizer/audio_featurizer.py:37-67,120-138):
//   AudioSegment.normalize(target_dB)           data_utils/audio.py:287-304   gain = 10^((target - rms_dB)/20)
//   AudioSegment.to('int16')                    data_utils/audio.py:244       clip, truncate to int16
//   paddleaudio.compliance.kaldi.fbank(...)     third-party (paddleaudio>=1.0.1, requirements.txt:14; not in the tree):
//       snip_edges framing (25 ms / 10 ms), remove_dc_offset, pre-emphasis 0.97 (first sample against itself),
//       povey window = hann(periodic=False)^0.85, zero-pad to 512, power spectrum, 80 triangular mel bins on the mel
//       scale 1127 ln(1 + f/700) from 20 Hz to Nyquist over FFT bins 0..255, log(max(e, FLT_EPSILON)), dither 0.
// One 256-thread workgroup per frame: the frame lives in LDS from the raw samples to the 80 log-mel values
// (radix-2 FFT, one butterfly per thread per stage); HBM traffic = the samples once (L2 absorbs the 2.5x frame
// overlap) + 320 B per frame out.  HBM- / latency-bound: 4 B * 160 new samples + 320 B out per frame.
#include "launch.h"
#include <hip/hip_runtime.h>
#include <math.h>

#include <algorithm>
#include <vector>

#include "capi_internal.h"

namespace {

constexpr int kFT = 256;       // threads per frame
constexpr int kNfftMax = 512;  // 25 ms at <= 16 kHz (20.48 kHz would still fit)

struct FbankTables {
  const float* window;   // [win]
  const float* tw_re;    // [nfft/2] cos(-2 pi k / nfft)
  const float* tw_im;    // [nfft/2] sin(-2 pi k / nfft)
  const float* bank;     // [n_mels][nfft/2]
  const int* bank_lo;    // [n_mels] first bin with a non-zero weight
  const int* bank_hi;    // [n_mels] one past the last
};

// ---- mean square of AudioSegment.rms_db (audio.py:526: np.mean(self._samples ** 2) on float32 samples) ----
// numpy reduces a contiguous float32 array in chunks of 8192 elements (the ufunc buffer size), res = 0; res += S(chunk),
// and S is its pairwise sum: n <= 128 -> eight running sums r[j] += a[8i + j], ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
// then the n % 8 tail element by element; n > 128 -> S(first n2) + S(rest), n2 = n/2 rounded down to a multiple of 8.
// All float32, squares rounded before they are added (x ** 2 is an array of its own).  The oracle calls np.mean, so the
// kernel has to reproduce this order bit for bit: one ulp of the mean square moves the gain and with it one int16 sample
// in a few thousand by one LSB (tests/test_fbank_gpu.py, the 10 s case).
// One workgroup per chunk: heap node h (root 1) of the chunk's tree = 8 lanes (the eight running sums of a leaf).
constexpr int kPwChunk = 8192, kPwLeaf = 128;
// A FULL chunk is 2^6 leaves of 128.  A shorter last chunk can be one level deeper: the "rest" half keeps up to 7 extra
// elements per split (8191 -> 4103 -> 2055 -> 1031 -> 519 -> 263 -> 135 -> 71), so 135 > 128 at depth 6 splits once more.
// Depth 7 is the bound for every n <= 8192 (tests/test_fbank_cpu.py walks all 8192 lengths): 256 heap slots.
constexpr int kPwDepth = 7, kPwSlots = 1 << (kPwDepth + 1);

// node h of the pairwise tree over n elements: its range; false when an ancestor is already a leaf
__host__ __device__ __forceinline__ bool pw_node(int h, int n, int& off, int& len) {
  off = 0;
  len = n;
  int top = 0;
  while ((h >> (top + 1)) != 0) ++top;  // depth of h (root 1 = depth 0)
  for (int d = top - 1; d >= 0; --d) {
    if (len <= kPwLeaf) return false;
    int n2 = len / 2;
    n2 -= n2 % 8;
    if ((h >> d) & 1) {
      off += n2;
      len -= n2;
    } else {
      len = n2;
    }
  }
  return true;
}

__global__ __launch_bounds__(1024) void k_sumsq(const float* __restrict__ x, int n, float* __restrict__ chunk_sum) {
  // numpy rounds every square before it is added (x ** 2 is an array of its own): no fused multiply-add here.  hipcc
  // contracts a * b + c -- also when written __fadd_rn(c, __fmul_rn(a, b)): the header's operations carry the contract
  // flag -- into v_fma_f32 under its default -ffp-contract=fast (found in round 6: one ulp of the sum for ~1 length in
  // 3), so this function uses plain operators with contraction switched off.
#pragma clang fp contract(off)
  __shared__ float val[kPwSlots];
  __shared__ unsigned char inner[kPwSlots];  // 1: node with two children
  const int c0 = blockIdx.x * kPwChunk, cn = min(kPwChunk, n - c0);
  const float* a = x + c0;
  const int j = threadIdx.x & 7;  // 8 lanes per heap slot; 128 slots per pass, two passes
  for (int h = threadIdx.x >> 3; h < kPwSlots; h += 128) {
    int off = 0, len = 0;
    const bool node = h >= 1 && pw_node(h, cn, off, len);
    const bool leaf = node && len <= kPwLeaf;
    float r = 0.f;
    if (leaf && len >= 8) {
      const int full = len - (len % 8);
      float v = a[off + j];
      r = v * v;
      for (int i = 8; i < full; i += 8) {
        v = a[off + i + j];
        r = r + v * v;
      }
    }
    // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)): a butterfly over the 8 lanes (float addition commutes)
    r = r + __shfl_xor(r, 1);
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    if (j == 0) {
      inner[h] = node && !leaf;
      if (leaf) {
        int i = len - (len % 8);
        if (len < 8) {
          r = 0.f;
          i = 0;
        }
        for (; i < len; ++i) {
          const float v = a[off + i];
          r = r + v * v;
        }
        val[h] = r;
      }
    }
  }
  __syncthreads();
  for (int d = kPwDepth - 1; d >= 0; --d) {  // inner nodes of depth d: slots [2^d, 2^(d+1))
    const int h = (1 << d) + (int)threadIdx.x;
    if ((int)threadIdx.x < (1 << d) && inner[h]) val[h] = val[2 * h] + val[2 * h + 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) chunk_sum[blockIdx.x] = val[1];
}

// gain of AudioSegment.normalize (audio.py:287-304, gain_db :256-264), once per call: ws[n_chunks] <- the linear gain
__global__ void k_gain(float* __restrict__ ws, int n_chunks, int n, float target_db) {
  float s = 0.f;
  for (int c = 0; c < n_chunks; ++c) s = __fadd_rn(s, ws[c]);
  // the scalar types numpy 1.x gives the reference here (oracle/fbank_oracle.py; pinned by tests/golden/ref_wav.npz): the
  // mean square and its log10 are float32, 10 * log10, target_db - rms_db and the power are float64, the gain is rounded
  // to float32 when it scales the float32 samples
  const float ms = (float)((double)s / (double)n);
  const double rms_db = ms != 0.f ? 10.0 * (double)(float)log10((double)ms) : 0.0;
  ws[n_chunks] = (float)pow(10.0, ((double)target_db - rms_db) / 20.0);
  ws[n_chunks + 1] = (float)((double)target_db - rms_db);  // the host raises beyond max_gain_db = 300 like audio.py:301
}

__global__ __launch_bounds__(kFT) void k_fbank(const float* __restrict__ x, int n, const float* __restrict__ gain_p,
                                               int use_db, float target_db, FbankTables tb, int win, int shift,
                                               int nfft, int log2n, int n_mels, float* __restrict__ feats) {
  __shared__ float re[kNfftMax], im[kNfftMax];
  __shared__ double dred[4];
  __shared__ float s_gain, s_mean;
  const int tid = threadIdx.x, frame = blockIdx.x;
  if (tid == 0) s_gain = use_db ? *gain_p : 1.0f;
  __syncthreads();
  const float gain = s_gain;
  // ---- load: float -> gain -> int16 (clip, truncate) ----
  const float* src = x + (size_t)frame * shift;
  for (int i = tid; i < nfft; i += kFT) {
    float v = 0.f;
    if (i < win) {
      float t = (src[i] * gain) * 32768.0f;
      t = fminf(fmaxf(t, -32768.f), 32767.f);
      v = truncf(t);
    }
    re[i] = v;
  }
  __syncthreads();
  // ---- remove_dc_offset ----
  {
    double s = 0.0;
    for (int i = tid; i < win; i += kFT) s += (double)re[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) dred[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) s_mean = (float)((dred[0] + dred[1] + dred[2] + dred[3]) / (double)win);
    __syncthreads();
  }
  const float mean = s_mean;
  // ---- pre-emphasis + window, written in bit-reversed order for the in-place DIT FFT ----
  float y[2];
  int dst[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = tid + q * kFT;
    float v = 0.f;
    if (i < win && i < nfft) {
      const float cur = re[i] - mean;
      const float prev = re[i > 0 ? i - 1 : 0] - mean;
      v = (cur - 0.97f * prev) * tb.window[i];
    }
    y[q] = v;
    dst[q] = (int)(__brev((unsigned)i) >> (32 - log2n));
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = tid + q * kFT;
    if (i < nfft) {
      re[dst[q]] = y[q];
      im[dst[q]] = 0.f;
    }
  }
  __syncthreads();
  // ---- radix-2 decimation-in-time FFT: nfft/2 butterflies per stage, one (or fewer) per thread ----
  for (int s = 1; s <= log2n; ++s) {
    const int half = 1 << (s - 1);
    for (int bfly = tid; bfly < nfft / 2; bfly += kFT) {
      const int grp = bfly >> (s - 1), k = bfly & (half - 1);
      const int i0 = (grp << s) + k, i1 = i0 + half;
      const int tw = k << (log2n - s);
      const float wr = tb.tw_re[tw], wi = tb.tw_im[tw];
      const float xr = re[i1], xi = im[i1];
      const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
      const float ur = re[i0], ui = im[i0];
      re[i0] = ur + tr; im[i0] = ui + ti;
      re[i1] = ur - tr; im[i1] = ui - ti;
    }
    __syncthreads();
  }
  // ---- power spectrum of bins 0 .. nfft/2-1 (the Nyquist bin is not used by the mel banks) ----
  for (int k = tid; k < nfft / 2; k += kFT) {
    const float p = re[k] * re[k] + im[k] * im[k];
    re[k] = p;
  }
  __syncthreads();
  if (tid < n_mels) {
    const float* w = tb.bank + (size_t)tid * (nfft / 2);
    float e = 0.f;
    for (int k = tb.bank_lo[tid]; k < tb.bank_hi[tid]; ++k) e = fmaf(re[k], w[k], e);
    feats[(size_t)frame * n_mels + tid] = logf(fmaxf(e, 1.1920928955078125e-07f));
  }
}

}  // namespace

struct ppasr_fbank_s {
  int sample_rate, n_mels, win, shift, nfft, log2n;
  FbankTables tb;
  std::vector<void*> allocs;
  ~ppasr_fbank_s() {
    for (void* p : allocs) (void)hipFree(p);
  }
};

extern "C" {

ppasr_status ppasr_fbank_create(int sample_rate, int n_mels, float frame_length_ms, float frame_shift_ms,
                                ppasr_fbank_handle* out) {
  if (!out || sample_rate <= 0 || n_mels < 1 || n_mels > kFT) return fail(PPASR_EINVAL, "fbank: bad arguments");
  auto f = std::make_unique<ppasr_fbank_s>();
  f->sample_rate = sample_rate;
  f->n_mels = n_mels;
  f->win = (int)(sample_rate * 0.001 * frame_length_ms);
  f->shift = (int)(sample_rate * 0.001 * frame_shift_ms);
  if (f->win < 2 || f->shift < 1) return fail(PPASR_EINVAL, "fbank: frame length / shift too small");
  int nfft = 1, l2 = 0;
  while (nfft < f->win) { nfft <<= 1; ++l2; }
  if (nfft > kNfftMax) return fail(PPASR_EUNSUPPORTED, "fbank: frame longer than 512 samples");
  f->nfft = nfft;
  f->log2n = l2;
  const int nb = nfft / 2;
  std::vector<float> window(f->win), twr(nb), twi(nb), bank((size_t)n_mels * nb, 0.f);
  std::vector<int> lo(n_mels), hi(n_mels);
  const double pi = 3.14159265358979323846;
  for (int i = 0; i < f->win; ++i)  // povey: hann (symmetric) ^ 0.85
    window[i] = (float)pow(0.5 - 0.5 * cos(2.0 * pi * i / (f->win - 1)), 0.85);
  for (int k = 0; k < nb; ++k) {
    twr[k] = (float)cos(-2.0 * pi * k / nfft);
    twi[k] = (float)sin(-2.0 * pi * k / nfft);
  }
  // get_mel_banks: low 20 Hz, high = Nyquist, no VTLN; weights in float32 on the mel scale
  auto mel = [](double hz) { return 1127.0 * log(1.0 + hz / 700.0); };
  const double mel_lo = mel(20.0), mel_hi = mel(0.5 * sample_rate);
  const double delta = (mel_hi - mel_lo) / (n_mels + 1);
  const double bin_w = (double)sample_rate / nfft;
  for (int m = 0; m < n_mels; ++m) {
    const double left = mel_lo + m * delta, center = left + delta, right = center + delta;
    int first = -1, last = -1;
    for (int k = 0; k < nb; ++k) {
      const double mk = mel(bin_w * k);
      const double up = (mk - left) / (center - left), down = (right - mk) / (right - center);
      const double w = std::max(0.0, std::min(up, down));
      bank[(size_t)m * nb + k] = (float)w;
      if (w > 0.0) {
        if (first < 0) first = k;
        last = k;
      }
    }
    lo[m] = first < 0 ? 0 : first;
    hi[m] = first < 0 ? 0 : last + 1;
  }
  auto up = [&](const void* src, size_t bytes, const void** dst) -> ppasr_status {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    f->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
    *dst = d;
    return PPASR_OK;
  };
  const void* p = nullptr;
  ppasr_status s;
  if ((s = up(window.data(), window.size() * 4, &p)) != PPASR_OK) return s;
  f->tb.window = static_cast<const float*>(p);
  if ((s = up(twr.data(), twr.size() * 4, &p)) != PPASR_OK) return s;
  f->tb.tw_re = static_cast<const float*>(p);
  if ((s = up(twi.data(), twi.size() * 4, &p)) != PPASR_OK) return s;
  f->tb.tw_im = static_cast<const float*>(p);
  if ((s = up(bank.data(), bank.size() * 4, &p)) != PPASR_OK) return s;
  f->tb.bank = static_cast<const float*>(p);
  if ((s = up(lo.data(), lo.size() * 4, &p)) != PPASR_OK) return s;
  f->tb.bank_lo = static_cast<const int*>(p);
  if ((s = up(hi.data(), hi.size() * 4, &p)) != PPASR_OK) return s;
  f->tb.bank_hi = static_cast<const int*>(p);
  *out = f.release();
  return PPASR_OK;
}

ppasr_status ppasr_fbank_destroy(ppasr_fbank_handle f) {
  delete f;
  return PPASR_OK;
}

int ppasr_fbank_frames(ppasr_fbank_handle f, int n_samples) {
  if (!f || n_samples < f->win) return 0;
  return 1 + (n_samples - f->win) / f->shift;  // snip_edges
}

static size_t fbank_ws_bytes(int n_samples) {
  // one float per 8192-sample chunk of the mean square + the gain + the gain in dB; never less than the 8 KiB earlier
  // versions asked for
  const size_t chunks = n_samples > 0 ? ((size_t)n_samples + kPwChunk - 1) / kPwChunk : 0;
  return std::max<size_t>(8192, (chunks + 2) * sizeof(float));
}

size_t ppasr_fbank_workspace_bytes(ppasr_fbank_handle f, int n_samples) { return f ? fbank_ws_bytes(n_samples) : 0; }

ppasr_status ppasr_fbank_compute(ppasr_fbank_handle f, const float* samples, int n_samples, int use_db_norm,
                                 float target_db, float* feats, void* workspace, size_t workspace_bytes, void* stream) {
  if (!f || !samples || !feats || !workspace) return fail(PPASR_EINVAL, "fbank: null argument");
  if (workspace_bytes < fbank_ws_bytes(n_samples)) return fail(PPASR_ENOSPACE, "fbank: workspace too small");
  const int frames = ppasr_fbank_frames(f, n_samples);
  if (frames <= 0) return PPASR_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  const int chunks = (n_samples + kPwChunk - 1) / kPwChunk;
  if (use_db_norm) {
    PPASR_LAUNCH(k_sumsq, dim3(chunks), dim3(1024), 0, st, samples, n_samples, ws);
    PPASR_LAUNCH(k_gain, dim3(1), dim3(1), 0, st, ws, chunks, n_samples, target_db);
  }
  PPASR_LAUNCH(k_fbank, dim3(frames), dim3(kFT), 0, st, samples, n_samples, ws + chunks, use_db_norm, target_db, f->tb,
                     f->win, f->shift, f->nfft, f->log2n, f->n_mels, feats);
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

}  // extern "C"
