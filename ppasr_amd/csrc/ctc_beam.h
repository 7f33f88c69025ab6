// ctc_beam.h -- launch interface of the CTC prefix beam search kernel (ctc_beam.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lm.h"

namespace ppasr {

constexpr int kMaxBeamCand = 128;  // pruned characters per frame the kernel can hold
constexpr int kMaxBeam = 512;
constexpr int kLmCtx = kLmMaxOrder - 1;       // LM context words carried per hypothesis
constexpr int kBeamStateArrays = 6 + kLmCtx;  // per-hypothesis words persisted between streaming calls

struct BeamConfig {
  int V, beam, blank;
  int cutoff_top_n;
  double cutoff_prob;
  int n_cand_max;  // min(kMaxBeamCand, what the pruning rule can produce)
  int max_nodes;   // arena capacity per utterance
  int nbest, max_tokens;
  // external scorer (ctc_beam_search_decoder.cpp `ext_scorer`): lm.order == 0 -> none
  LmDev lm;
  double alpha, beta;
};

// words of one frame record of the pruning pre-pass: C, p_blank, n_cand_max characters, n_cand_max log-probs
__host__ __device__ inline int prune_rec_words(int n_cand_max) { return 2 + 2 * n_cand_max; }
size_t beam_lds_bytes(const BeamConfig& c);
size_t beam_state_bytes(const BeamConfig& c);  // per utterance
// prune_recs: device scratch of B * T * prune_rec_words(cfg.n_cand_max) words
hipError_t launch_ctc_beam(const float* probs, const int32_t* frame_lens, int B, int T, const BeamConfig& cfg,
                           int32_t* prune_recs, int32_t* state, int init_state, int finalize, int32_t* out_tokens, int32_t* out_lens,
                           double* out_scores, int32_t* status, hipStream_t st);

}  // namespace ppasr
