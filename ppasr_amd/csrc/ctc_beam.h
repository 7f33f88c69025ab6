// ctc_beam.h -- launch interface of the CTC prefix beam search kernel (ctc_beam.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lm.h"

namespace ppasr {

constexpr int kSmallCand = 128;  // pruning records of up to this many characters per frame live in the state buffer and are
                                 // staged in LDS; wider ones (cutoff_prob >= 1: the whole vocabulary) go through HBM scratch
constexpr int kMaxBeam = 512;
constexpr int kLmCtx = kLmMaxOrder - 1;       // LM context words carried per hypothesis
constexpr int kBeamStateArrays = 7 + kLmCtx;  // per-hypothesis words persisted between streaming calls (node, char, parent,
                                              // log P_b, log P_nb, score, LM context words, dictionary state)

struct BeamConfig {
  int V, beam, blank;
  int cutoff_top_n;
  double cutoff_prob;
  int n_cand_max;  // what the pruning rule can produce: min(cutoff_top_n, V) when cutoff_prob < 1, else V
  int max_nodes;   // arena capacity per utterance
  int nbest, max_tokens;
  int node_table;  // 1: prefixes keep their node id across drop / re-creation (see the state layout below)
  int fast_path;   // 1 (default): without a scorer the rows of the element list are clipped to a verified staircase and
                   // short lists are ranked with ballots (ctc_beam.hip k_ctc_beam); 0 (PPASR_BEAM_FAST=0, tests): always full
                   // rows and the radix selection.  Same results bit for bit
  int sorted;      // 1: the candidate lists are in probability order (anything but cutoff_prob >= 1 with cutoff_top_n >= V)
  int margin;      // extra candidates per row beyond the (rank + 1) (k + 1) <= beam staircase
  int list_cap;    // entries of the LDS element list (beam_list_cap); longer lists use the HBM scratch
  // external scorer (ctc_beam_search_decoder.cpp `ext_scorer`): lm.order == 0 -> none
  LmDev lm;
  double alpha, beta;
};

// ---- per-utterance state block in HBM (int32 words), shared by the kernels and the C-ABI's size arithmetic ----
//   [0] n_beam  [1] n_nodes | kBeamStateArrays arrays of `beam` words | (pad to an even word) |
//   arena: kArenaWords words per node (parent id, character, dictionary state), node ids in creation order, root = 0 |
//   node table: open-addressing hash (parent id, character) -> node id with 2 * max_nodes slots: uint64 keys, then int32 ids.
// The node table gives a prefix ONE identity for the whole search: a prefix that drops out of the beam and is created
// again (while a longer prefix that runs through it survived) gets its old node id back, so the survivor is still
// recognised as its child -- upstream's trie keeps such nodes (PathTrie::remove only deletes childless nodes) and revives
// them in get_path_trie, with the dictionary state they had; without the table the re-created prefix spawns a duplicate
// of its own descendant.  It costs two dependent HBM round trips per frame (look-up, insertion), so it is used where
// such revivals are frequent -- word-based scorers, whose dictionary funnels the beam into few spellings -- and on request
// (BeamConfig::node_table; PPASR_BEAM_NODE_TABLE=1); otherwise every new prefix gets a fresh id (DOCUMENTED DEVIATION: in
// the rare revival case the re-created prefix and the old descendant's line split their probability mass).
__host__ __device__ inline size_t beam_fixed_words(int beam) { return ((size_t)2 + (size_t)kBeamStateArrays * beam + 1) & ~(size_t)1; }
__host__ __device__ inline size_t beam_table_slots(int max_nodes) { return (size_t)2 * max_nodes; }
constexpr int kArenaWords = 3;
__host__ __device__ inline size_t beam_arena_words(int max_nodes) { return ((size_t)kArenaWords * max_nodes + 1) & ~(size_t)1; }
__host__ __device__ inline size_t beam_state_words(int beam, int max_nodes) {
  return beam_fixed_words(beam) + beam_arena_words(max_nodes) + 3 * beam_table_slots(max_nodes);
}
__host__ __device__ inline uint64_t beam_node_key(int parent, int ch) {
  // node ids and characters are non-negative 31-bit values: (parent, character) pairs never share a key, whatever V is
  return ((uint64_t)(uint32_t)parent << 32) | (uint64_t)(uint32_t)ch | (1ull << 63);
}
__host__ __device__ inline size_t beam_node_slot(uint64_t key, size_t slots) {
  return (size_t)(((key * 0x9E3779B97F4A7C15ull) >> 20) % slots);
}

// words of one frame record of the pruning pre-pass: C, p_blank, n_cand_max characters, n_cand_max log-probs
__host__ __device__ inline int prune_rec_words(int n_cand_max) { return 2 + 2 * n_cand_max; }
size_t beam_lds_bytes(const BeamConfig& c);
int beam_list_cap(int beam, int V, int n_cand_max, bool has_lm);  // entries of the LDS element list (0: the fixed arrays alone do not fit)
size_t beam_state_bytes(const BeamConfig& c);  // per utterance
// HBM scratch a call needs beyond the state buffer (0 for the shipped configurations): wide pruning records
// (n_cand_max > kSmallCand) and / or per-utterance element lists that do not fit LDS
size_t beam_scratch_bytes(const BeamConfig& c, int B, int T);
// prune_recs: the state buffer's record area, B * T * prune_rec_words(cfg.n_cand_max) words (narrow records); wide records
// are written to `scratch`.  `scratch`: beam_scratch_bytes(cfg, B, T) bytes or nullptr when that is 0
hipError_t launch_ctc_beam(const float* probs, const int32_t* frame_lens, int B, int T, const BeamConfig& cfg,
                           int32_t* prune_recs, int32_t* state, int init_state, int finalize, int32_t* out_tokens, int32_t* out_lens,
                           double* out_scores, int32_t* status, void* scratch, hipStream_t st);

// rebuilds the node tables of B state blocks (cleared by the caller) from their arenas: after a streaming state buffer grew
hipError_t launch_beam_rehash(int32_t* state, int B, int beam, int max_nodes, hipStream_t st);

}  // namespace ppasr
