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
constexpr int kMaxPartials = 1024;

struct FbankTables {
  const float* window;   // [win]
  const float* tw_re;    // [nfft/2] cos(-2 pi k / nfft)
  const float* tw_im;    // [nfft/2] sin(-2 pi k / nfft)
  const float* bank;     // [n_mels][nfft/2]
  const int* bank_lo;    // [n_mels] first bin with a non-zero weight
  const int* bank_hi;    // [n_mels] one past the last
};

__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ x, int n, double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += (double)x[i] * (double)x[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(kFT) void k_fbank(const float* __restrict__ x, int n, const double* __restrict__ partial,
                                               int n_partial, int use_db, float target_db, FbankTables tb, int win, int shift,
                                               int nfft, int log2n, int n_mels, float* __restrict__ feats) {
  __shared__ float re[kNfftMax], im[kNfftMax];
  __shared__ double dred[4];
  __shared__ float s_gain, s_mean;
  const int tid = threadIdx.x, frame = blockIdx.x;
  // ---- gain of AudioSegment.normalize: every frame re-reduces the (<= 1024) partial sums ----
  if (use_db) {
    double s = 0.0;
    for (int i = tid; i < n_partial; i += kFT) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) dred[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
      // AudioSegment.rms_db / normalize / gain_db (audio.py:519-530,287-304,256-264) with numpy 1.x's scalar types: the mean
      // square, rms_db and `target_db - rms_db` are float32, the power is taken in float64 and rounded to float32
      // (pinned by tests/golden/ref_wav.npz: a float64 rms_db moves the gain by 3 ulp)
      const float ms = (float)((dred[0] + dred[1] + dred[2] + dred[3]) / (double)n);
      const float rms_db = 10.0f * (float)log10((double)(ms != 0.f ? ms : 1.f));
      const float gain_db = target_db - rms_db;
      s_gain = (float)pow(10.0, (double)gain_db / 20.0);
    }
    __syncthreads();
  } else if (tid == 0) {
    s_gain = 1.0f;
  }
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

size_t ppasr_fbank_workspace_bytes(ppasr_fbank_handle f, int n_samples) {
  (void)n_samples;
  return f ? kMaxPartials * sizeof(double) : 0;
}

ppasr_status ppasr_fbank_compute(ppasr_fbank_handle f, const float* samples, int n_samples, int use_db_norm,
                                 float target_db, float* feats, void* workspace, size_t workspace_bytes, void* stream) {
  if (!f || !samples || !feats || !workspace) return fail(PPASR_EINVAL, "fbank: null argument");
  if (workspace_bytes < kMaxPartials * sizeof(double)) return fail(PPASR_ENOSPACE, "fbank: workspace too small");
  const int frames = ppasr_fbank_frames(f, n_samples);
  if (frames <= 0) return PPASR_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* partial = static_cast<double*>(workspace);
  const int nblk = std::min(kMaxPartials, (n_samples + 4095) / 4096);
  if (use_db_norm) PPASR_LAUNCH(k_sumsq, dim3(nblk), dim3(256), 0, st, samples, n_samples, partial);
  PPASR_LAUNCH(k_fbank, dim3(frames), dim3(kFT), 0, st, samples, n_samples, partial, nblk, use_db_norm, target_db, f->tb,
                     f->win, f->shift, f->nfft, f->log2n, f->n_mels, feats);
  HIP_TRY(hipGetLastError());
  return PPASR_OK;
}

}  // extern "C"
