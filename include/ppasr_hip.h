/*
 * ppasr_hip.h -- C-ABI of libppasr_hip.so, the MI355X (gfx950) implementation of
 * PPASR's encoder-forward + CTC-decode hot path.
 *
 * PPASR has no FFI of its own (it is pure Python on PaddlePaddle); the boundary a
 * maintainer would bind is the set of Python call sites listed per entry point
 * below (paths relative to the reference checkout, `ppasr/...`).  INTEGRATION.md
 * shows the ctypes stub for each.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - all work is enqueued on the caller's HIP stream (`stream`, a hipStream_t
 *     passed as void*); nothing synchronises, nothing allocates after create();
 *   - the caller owns every buffer including the scratch workspace
 *     (size from ppasr_workspace_bytes); the handle owns only packed weights;
 *   - a handle is not re-entrant; distinct handles are independent;
 *   - no exceptions cross the ABI: int status + ppasr_last_error() (thread-local).
 */
#ifndef PPASR_HIP_H
#define PPASR_HIP_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: these entry points are its whole dynamic symbol table
 * (tests/test_capi_cpu.py compares `nm -D --defined-only` with this header). */
#define PPASR_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ppasr_model_s* ppasr_handle;
typedef struct ppasr_stream_s* ppasr_stream;

typedef enum {
  PPASR_OK = 0,
  PPASR_EINVAL = 1,       /* bad argument / shape */
  PPASR_EHIP = 2,         /* a HIP runtime call failed */
  PPASR_EUNSUPPORTED = 3, /* configuration outside what the kernels are built for */
  PPASR_EMISSING = 4,     /* a required weight blob is absent */
  PPASR_ENOSPACE = 5      /* workspace too small */
} ppasr_status;

enum { PPASR_MODEL_CONFORMER = 0, PPASR_MODEL_EFFICIENT_CONFORMER = 1,
       PPASR_MODEL_SQUEEZEFORMER = 2, PPASR_MODEL_DEEPSPEECH2 = 3 };

/* One named parameter of a Paddle state dict (`model.pdparams`, trainer.py:302-328):
 * float32, C-contiguous, HOST memory, Paddle layout (Linear.weight is [in,out]). */
typedef struct {
  const char* name;
  const float* data_host;
  int ndim;
  int64_t shape[4];
} ppasr_weight_blob;

/* Mirrors `encoder_conf` of configs/conformer.yml:2-16 plus the constructor
 * arguments of ConformerModel (model_utils/conformer/model.py:17-29). */
typedef struct {
  int model_type;        /* PPASR_MODEL_* */
  int input_dim;         /* fbank bins F (80) */
  int vocab_size;        /* V */
  int output_size;       /* d (256) */
  int attention_heads;   /* h (4) */
  int linear_units;      /* FFN hidden (2048) */
  int num_blocks;        /* L (12) */
  int cnn_module_kernel; /* 15 */
  int causal;            /* streaming model => causal depthwise conv (model.py:35-39) */
  int max_len;           /* positional table length (embedding.py:30), 5000 */
  /* Squeezeformer (configs/squeezeformer.yml:7-8, squeezeformer/encoder.py:30-31); -1 = none */
  int reduce_idx;
  int recover_idx;
  /* Efficient-Conformer (configs/efficient_conformer.yml:16-21) */
  int stride_layer_idx;   /* layer with the stride-2 depthwise conv, -1 = none */
  int group_layer_mask;   /* bit i set: layer i uses grouped attention */
  int group_size;         /* 2, 3 (the shipped value) or 4 frames per attention token */
  /* DeepSpeech2 (configs/deepspeech2.yml:5, deepspeech2/encoder.py:36-42): nn.GRU layers instead of nn.LSTM */
  int use_gru;
  /* Conformer / Efficient-Conformer front end, `input_layer` (conformer/encoder.py:93-104, subsampling.py):
     0 = conv2d (Conv2dSubsampling4: 3x3/2, 3x3/2), 6 = conv2d6 (Conv2dSubsampling6: 3x3/2, 5x5/3, `embed.linear`),
     8 = conv2d8 (Conv2dSubsampling8: 3x3/2 three times, `embed.linear`).  6 / 8: batched encode and single stream
     handles (no session groups); key-padding / conv pad masks use 6t / 8t < len (the reference's mask slicing) */
  int input_layer;        /* 1 = linear (LinearNoSubsampling, subsampling.py:24-65: no time reduction; general route only) */
  /* Non-default ConformerEncoder constructor arguments (conformer/encoder.py:38-48), PPASR_OPT_* below; 0 = the shipped
     configuration (rel_pos, normalize_before, macaron_style, use_cnn_module, swish).  Any other value -- like
     output_size != 256, input_layer = linear or a cnn_module_kernel other than 7 / 15 / 31 -- selects the general layer
     route (capi_generic.hip: the same packed weights and matrix-core GEMMs, one launch per layer piece instead of the
     fused 256-wide row-block kernels).  options / input_layer = linear: model_type = conformer only; output_size 512 / 768 /
     1024 (heads of 64): conformer, efficient_conformer and squeezeformer, batched and streaming. */
  int options;
  /* Efficient-Conformer with SEVERAL stride layers (`stride_layer_idx: [1, 3]`, `stride: [2, 2]`, efficient_conformer/
     encoder.py:50-54,117-128): bit i set = layer i is a stride-2 layer (its depthwise conv strides, the residual goes
     through AvgPool1D(2, ceil_mode), every later layer's conv kernel is halved once more).  0 = use stride_layer_idx.  More
     than one bit: the general layer route, batched encode only (stream handles are refused). */
  int stride_layer_mask;
} ppasr_model_desc;

enum {
  PPASR_OPT_POS_REL = 0,       /* pos_enc_layer_type: rel_pos -> RelPositionMultiHeadedAttention */
  PPASR_OPT_POS_ABS = 1,       /*   abs_pos: x * sqrt(d) + pe, MultiHeadedAttention (embedding.py:25-84) */
  PPASR_OPT_POS_NONE = 2,      /*   no_pos: MultiHeadedAttention, x unchanged (embedding.py:10-22) */
  PPASR_OPT_POS_MASK = 3,
  PPASR_OPT_POST_NORM = 4,     /* normalize_before = False (encoder.py:380-428; no after_norm) */
  PPASR_OPT_CONCAT_AFTER = 8,  /* concat_after = True: x + concat_linear([x | att(x)]) (encoder.py:395-397) */
  PPASR_OPT_NO_MACARON = 16,   /* macaron_style = False: no feed_forward_macaron, ff_scale 1 (encoder.py:330-334) */
  PPASR_OPT_NO_CNN = 32,       /* use_cnn_module = False: no conv module, no norm_final */
  PPASR_OPT_SQ_NO_ADAPTIVE_SCALE = 4096, /* model_type squeezeformer only: adaptive_scale = False (squeezeformer/encoder.py:44;
                                  the checkpoint's ada_scale / ada_bias are then not applied).  The two other options of that
                                  encoder need no flag: dw_stride = True is recognised by the [d][1][3][3] shape of
                                  encoder.embed.dw_conv.weight, final_proj by the presence of encoder.final_proj.weight */
  PPASR_OPT_SQ_PRE_NORM = 8192, /* model_type squeezeformer only: normalize_before = True (squeezeformer/encoder.py:49,467-493:
                                  LayerNorm_k in front of module k instead of behind the residual sum); general layer route */
  PPASR_OPT_ACT_SHIFT = 8,     /* activation_type (utils/common.py:189-206) in bits 8..11: */
  PPASR_OPT_ACT_MASK = 15
};
enum { PPASR_ACT_SWISH = 0, PPASR_ACT_RELU = 1, PPASR_ACT_GELU = 2, PPASR_ACT_TANH = 3, PPASR_ACT_HARDTANH = 4,
       PPASR_ACT_RELU6 = 5, PPASR_ACT_LEAKYRELU = 6, PPASR_ACT_SELU = 7, PPASR_ACT_ELU = 8, PPASR_ACT_HARDSWISH = 9,
       PPASR_ACT_HARDSHRINK = 10 };

PPASR_API const char* ppasr_last_error(void);
PPASR_API const char* ppasr_version(void);

/* Replaces: model construction + paddle.inference.create_predictor
 * (infer_utils/inference_predictor.py:41-77) / PPASRTrainer.__setup_model
 * (trainer.py:172-210).  Uploads and re-packs the weights into MFMA fragment order. */
PPASR_API ppasr_status ppasr_create(const ppasr_model_desc* desc, const ppasr_weight_blob* weights_host, int n_weights,
                          ppasr_handle* out);
PPASR_API ppasr_status ppasr_destroy(ppasr_handle h);

/* frames after the conv front-end: ((T-1)/2-1)/2  (conformer/subsampling.py:84-94,115) */
PPASR_API int ppasr_out_frames(ppasr_handle h, int T);
PPASR_API size_t ppasr_workspace_bytes(ppasr_handle h, int B, int T);

/* Replaces ConformerModel.get_encoder_out (model_utils/conformer/model.py:148-162), as called by
 * PPASRTrainer.evaluate (trainer.py:626) and InferencePredictor.predict (inference_predictor.py:103-145).
 *   feats [B,T,F] f32 zero-padded, lens [B] i64  ->  any of
 *   probs  [B,T',V] f32  softmax(ctc_lo(enc))       (what the reference returns)
 *   logits [B,T',V] f32  pre-softmax                (parity tap)
 *   frame_argmax [B,T'] i32, frame_maxprob [B,T'] f32: per-frame argmax / probability at the
 *     argmax, produced inside the CTC-head kernel (fused ctc_greedy first stage,
 *     decoders/ctc_greedy_decoder.py:21-22); pass NULL for outputs not wanted. */
PPASR_API ppasr_status ppasr_encode(ppasr_handle h, const float* feats, const int64_t* lens, int B, int T,
                          float* probs, float* logits, int32_t* frame_argmax, float* frame_maxprob,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Debug taps: when set (host call, before ppasr_encode), intermediate activations are copied
 * to `taps` in the order documented in DESIGN.md; pass NULL to clear. */
PPASR_API ppasr_status ppasr_set_debug_taps(ppasr_handle h, float* taps, size_t n_floats);

/* Ragged batches.  The reference computes every padded row of a batch (its batched decoders even consume them,
 * trainer.py:347).  With enable != 0, ppasr_encode computes, per utterance, only the rows its VALID output frames
 * depend on (t < ceil(len/4), or ceil(len/8) after a rate change, plus a few rows of slack for the grouped-attention /
 * stride / time-reduction reads): whole 32-row blocks behind them are skipped in every kernel and attention stops at
 * the last valid key.  Valid rows are bit-identical to the default mode AS LONG AS both runs use the same block form of the
 * layer kernels (ppasr_set_row_block): with the default rule (-1) the form is chosen from the rows actually computed
 * (ppasr_set_lengths_hint), so a ragged call may take the 16-row kernels where the padded call takes the 32-row ones --
 * the same sums in another order, equal to ~1e-6 relative; pin the form (16 / 32) where bit-identity between the modes,
 * between ranks or between batch compositions is required.  Rows of `probs` / `logits` behind an
 * utterance's last valid frame are set to 0, `frame_argmax` to 0 (blank) and `frame_maxprob` to 0 -- pass frame_lens
 * to the decoders.  Default: off (= the reference's outputs for every row).  Built into the fused 256-column kernels and
 * the general layer route (widths 512 .., constructor options) behind the conv front ends (behind conv2d6 / conv2d8 and on
 * the general route the layers and the head skip, the front end itself computes every row): enabling it on an
 * input_layer = linear or DeepSpeech2 handle returns PPASR_EUNSUPPORTED (those compute every row). */
PPASR_API ppasr_status ppasr_set_skip_padding(ppasr_handle h, int enable);

/* Under-filled launches.  A kernel with fewer 32-row blocks than the chip has CUs takes as long as a full one.  When a
 * call has <= 128 row blocks (small batches, the half-rate layers of the Efficient-Conformer, a single streaming
 * session) the Conformer-family layer tail is cut at its two feed-forward modules and each module's hidden dimension is
 * split over 2 / 4 / 8 workgroups per row block (partial sums joined by the next launch).  Same arithmetic up to the
 * order of the final sum over hidden chunks.  mode: -1 = decide by grid size (default), 0 = never (always the fused
 * kernels, the fused attention kernel included), 2 / 4 / 8 = always that many slices. */
PPASR_API ppasr_status ppasr_set_ffn_split(ppasr_handle h, int mode);

/* Conv2dSubsampling4 (conformer/subsampling.py:84-88) as ONE launch (csrc/front_fused.hip: conv1's output is computed
 * tile by tile inside conv2's implicit GEMM and never written to HBM) or as two (k_conv1, then conv2 reading its output).
 * Bit-identical results.  mode: -1 (default) / 1 = one launch, 0 = two.  The one-launch kernel is 0.5 - 0.8 % faster
 * end to end on its own stream; with a beam search of the previous batch running on a second stream the two-launch form
 * is the faster one (conv2's workgroups leave room on a CU for the search's, the fused kernel's do not), which is why
 * the pipelined plans of ppasr_amd/parallel.py select it.  Other front ends ignore the setting. */
PPASR_API ppasr_status ppasr_set_front_fused(ppasr_handle h, int mode);

/* Block form of the layer kernels (no reference counterpart; csrc/rbt.h): -1 (default) = by grid size -- up to 32 row
 * blocks of 32 rows the split route above; 16-row blocks (v_mfma_f32_16x16x4_f32 on 8 waves) where two short rounds of
 * them beat the 32-row rounds (33 .. 128 blocks: under-filled launches); else 32-row blocks (v_mfma_f32_32x32x2_f32 on
 * 8 waves).  16 / 32 = always that form; PPASR_ROW_BLOCK_32_W16 = 32 rows on SIXTEEN waves (v_mfma_f32_16x16x4_f32 on
 * four waves per SIMD; measured equal to the 8-wave kernels end to end, kept as an option).  Same arithmetic up to the
 * summation order inside a 16-wide k step (1e-6 relative).  Built for the Conformer / Efficient-Conformer /
 * Squeezeformer layer kernels behind the conv2d (4x) front end; other routes ignore it. */
#define PPASR_ROW_BLOCK_32_W16 1032
PPASR_API ppasr_status ppasr_set_row_block(ppasr_handle h, int rows);

/* Arithmetic of the large GEMMs of the fused Conformer-family routes (no reference counterpart; csrc/h3.h).
 * PPASR_GEMM_F32 (default): v_mfma_f32_32x32x2_f32, exact fp32 products.  PPASR_GEMM_F16X3 (opt-in): every operand is the
 * sum of two fp16 pieces (22 significant bits; products of pieces are exact in fp32, accumulation in fp32), three
 * v_mfma_f32_32x32x16_f16 per 16-wide k step instead of eight fp32 MFMAs -- one feed-forward module deviates from float64
 * by 2.7e-7 where fp32 arithmetic deviates by 5.5e-7 (tools/experiments/r05/ffn_h3.hip), but NOT bit-identical to the
 * default mode.  The first call re-packs the weights concerned on the device (second copy, ~ 10 MB per layer).
 * Range.  Weights are scaled by 2^8, activations by 2^4 before they are cut into fp16 pieces: a weight of magnitude >= 255.9
 * makes this call fail with PPASR_EUNSUPPORTED (the handle stays in its previous mode); an activation beyond 4 094 at a
 * GEMM input (ReLU outputs in front of conv2 / the input projection, swish values in front of W2, LayerNorm outputs
 * elsewhere) is handled by the range guard below -- never Inf / NaN.
 * Built for the fused 256-wide routes: Conformer / Efficient-Conformer -- feed-forward modules, Q/K/V, pointwise_conv2
 * (8-wave 32-row layer kernels incl. the Efficient-Conformer's stride layer; depthwise kernel sizes 15 and 7), linear_out +
 * pointwise_conv1 and the score contraction
 * [q+u | q+v] [k | p]^T of the fused attention kernel (K as fp16 hi / lo planes from the QKV stage, the layer's positional
 * table re-packed by this call: + 1 KiB per table row; an entry beyond 4 094 refuses the mode like a weight does);
 * Squeezeformer -- both feed-forward modules of the 32-row layer kernels (depthwise kernel sizes 31 and 15); all three -- the
 * second convolution of the 4x front end (then its own launch behind conv1), the input projection, the CTC head;
 * Conformer / Efficient-Conformer also on the split route of under-filled launches and therefore on their stream handles and
 * session groups (ppasr_encode_chunk*: a saturated chunk is counted in the guard statistics, NOT re-run).  The other block
 * forms (16 rows, 16 waves), the non-feed-forward units of Squeezeformer's split route and streams, the attention's P V product and depthwise convolutions keep fp32 arithmetic.  ppasr_gemm_coverage
 * tells which of the three parts switched (a Conformer with cnn_module_kernel 31 gets the front end and the head only).
 * PPASR_EUNSUPPORTED on DeepSpeech2 handles and on the general layer route.  Measured (NOTES.md 9.8): logits within 1e-6 .. 3e-6
 * of the default mode's, the reference-source pin tests pass with unchanged criteria, 1.2 - 1.7 x faster end to end. */
#define PPASR_GEMM_F32 0
#define PPASR_GEMM_F16X3 1
#define PPASR_GEMM_COVERS_LAYERS 1 /* the encoder layers' GEMMs */
#define PPASR_GEMM_COVERS_FRONT 2  /* conv2 + input projection of the 4x front end */
#define PPASR_GEMM_COVERS_HEAD 4   /* the CTC head */
PPASR_API ppasr_status ppasr_set_gemm_mode(ppasr_handle h, int mode);
PPASR_API int ppasr_gemm_coverage(ppasr_handle h); /* PPASR_GEMM_COVERS_* bits of the current mode; 0 in PPASR_GEMM_F32 */
/* Range guard of PPASR_GEMM_F16X3.  The kernels saturate an out-of-range (or NaN) GEMM input to +-65 504 / 2^4 and count
 * the event in a device counter.  enable = 1 (default): ppasr_encode reads the counters back after its launches (it then
 * SYNCHRONISES `stream` before it returns) and, if the call saturated anything, runs the call again on the fp32 kernels:
 * the caller always gets finite-input-exact results, the fp32 mode's bit for bit in that case.  enable = 0: ppasr_encode
 * stays asynchronous, a saturated result stands, and the caller polls ppasr_gemm_guard_stats.  The counters are shared by
 * the handles of a process: concurrent fp16 x3 handles can cause each other a spurious fp32 re-run, never a missed one. */
PPASR_API ppasr_status ppasr_set_gemm_guard(ppasr_handle h, int enable);
/* -> calls that were re-run on the fp32 kernels / saturation events seen by this handle (guard off: waits for the device,
 * then counts the events since this handle last looked).  Either pointer may be NULL. */
PPASR_API ppasr_status ppasr_gemm_guard_stats(ppasr_handle h, long long* fallbacks_host, long long* events_host);
/* Host copy of the `lens` the following ppasr_encode calls of a batch of B utterances will pass (NULL / 0: forget it).
 * Only ever used to CHOOSE between kernel variants for ragged batches (ppasr_set_skip_padding), whose count of computed
 * rows the host cannot otherwise know; no kernel reads it.  A wrong hint costs speed; the variants differ in the order of
 * their sums (~1e-6 relative, see ppasr_set_row_block), so two calls with different hints need not agree bit for bit. */
PPASR_API ppasr_status ppasr_set_lengths_hint(ppasr_handle h, const int64_t* lens_host, int B);

/* Host helper (no device work): Levenshtein distance between two int32 sequences -- what ppasr/utils/metrics.py:4-29
 * (cer / wer) gets from the `Levenshtein` C extension.  Returns -1 on a null argument with a positive length. */
PPASR_API long long ppasr_edit_distance(const int32_t* a, int na, const int32_t* b, int nb);

/* Replaces the third-party `paddlespeech_ctcdecoders` entry points PPASR calls:
 *   ctc_beam_search_decoding / ctc_beam_search_decoding_batch  (decoders/swig_wrapper.py:61-62,98-100,
 *     from BeamSearchDecoder.decode_beam_search_offline / decode_batch_beam_search_offline,
 *     decoders/beam_search_decoder.py:45-73), with init_state = 1;
 *   CtcBeamSearchDecoderBatch.next() + decode() (swig_wrapper.py:106-121, beam_search_decoder.py:75-96):
 *     call again with init_state = 0 and the SAME state buffer for every further chunk.
 * CTC prefix beam search; the external scorer variant is ppasr_ctc_beam_search_lm below.
 *   probs [B,T,V] f32, frame_lens [B] i32 or NULL; per utterance the `nbest` best prefixes:
 *   tokens [B,nbest,max_tokens] i32 (-1 padded), lens [B,nbest] (-1 = no such hypothesis),
 *   scores [B,nbest] f64 = -log P(prefix) (the upstream return convention).
 *   state: device scratch of ppasr_ctc_beam_state_bytes(B, max total frames, beam_size) bytes that
 *   holds the beam, the prefix arena (both kept between chunk calls) and the per-frame records of the
 *   pruning pre-pass (get_pruned_log_probs of every frame of the call: T <= max total frames). */
PPASR_API size_t ppasr_ctc_beam_state_bytes(int B, int max_frames, int beam_size);
/* Extra device scratch a call with this pruning configuration needs (0 for every configuration the reference ships:
 * cutoff_prob < 1 with cutoff_top_n <= 128).  Non-zero when more than 128 characters of a frame can survive the pruning
 * -- cutoff_prob >= 1, the default of swig_wrapper.py:38,71, where upstream ignores cutoff_top_n and keeps the WHOLE
 * vocabulary; or cutoff_top_n > 128 -- or when beam_size x (1 + candidates) elements do not fit LDS: the per-frame pruning
 * records and the element list of a frame then live in this scratch (T x (2 + 2 V) words + beam x (1 + V) x 5 bytes per
 * utterance).  Pass it to ppasr_ctc_beam_search_ws; the plain entry points return PPASR_ENOSPACE for such a call.
 * There is no cap on the candidates of a frame. */
PPASR_API size_t ppasr_ctc_beam_scratch_bytes(int B, int T, int V, int beam_size, double cutoff_prob, int cutoff_top_n);
/* Streaming past the sized capacity: copies the beams and prefix arenas of a state buffer into a LARGER one (sized with
 * ppasr_ctc_beam_state_bytes for more frames), which then continues the same search.  The reference's decoder object has
 * no frame limit; callers double the buffer when the next chunk would not fit.  Asynchronous on `stream`. */
PPASR_API ppasr_status ppasr_ctc_beam_state_grow(const void* old_state, size_t old_bytes, void* new_state, size_t new_bytes, int B,
                                       int beam_size, void* stream);
/* Reads back the per-utterance status words of a (streaming) state buffer: non-zero = the prefix arena ran out because
 * more cumulative frames were decoded than the buffer was sized for; returns PPASR_ENOSPACE then.  Synchronises. */
PPASR_API ppasr_status ppasr_ctc_beam_status(const void* state, size_t state_bytes, int B, int beam_size, int32_t* status_host,
                                   void* stream);
PPASR_API ppasr_status ppasr_ctc_beam_search(const float* probs, const int32_t* frame_lens, int B, int T, int V, int beam_size,
                                   double cutoff_prob, int cutoff_top_n, int blank, int nbest, int max_tokens,
                                   int32_t* tokens, int32_t* lens, double* scores, void* state, size_t state_bytes,
                                   int init_state, void* stream);

/* External scorer = `Scorer(alpha, beta, model_path, vocabulary)` of paddlespeech_ctcdecoders (decoders/swig_wrapper.py:18-33,
 * built by BeamSearchDecoder.__init__, decoders/beam_search_decoder.py:19-29): back-off n-gram model, either
 * CHARACTER-based (every LM word is one UTF-8 character, as PPASR's Mandarin models are: scored on every extension) or
 * WORD-based (configs/english_example.yml: scored when a space completes a word, the prefixes constrained to spellings
 * of the model's vocabulary by a dictionary -- upstream's OpenFST acceptor, here a character trie; the acoustic vocabulary
 * must contain the space token " " or "<space>").  Model files: the ARPA text format (csrc/lm.hip) and KenLM binaries
 * (.klm, what PPASR ships: decoders/beam_search_decoder.py:19-29) of every model type: "probing" / "rest probing",
 * "trie", and the quantised (-q) / Bhiksha-array (-a) trie variants (csrc/klm.hip).  ppasr_lm_create sniffs the format
 * from the file's magic, like KenLM's loader.
 * vocab_utf8[V]: the acoustic vocabulary (token id -> string), used to map token ids to LM words (unknown -> OOV). */
typedef struct ppasr_lm_s* ppasr_lm_handle;
PPASR_API ppasr_status ppasr_lm_create(const char* model_path, const char* const* vocab_utf8, int V, ppasr_lm_handle* out);
PPASR_API ppasr_status ppasr_lm_create_arpa(const char* arpa_path, const char* const* vocab_utf8, int V, ppasr_lm_handle* out);
PPASR_API ppasr_status ppasr_lm_create_klm(const char* klm_path, const char* const* vocab_utf8, int V, ppasr_lm_handle* out);
PPASR_API const char*  ppasr_lm_format(ppasr_lm_handle lm);              /* "arpa", "klm-probing", "klm-rest-probing", "klm-trie",
                                                                  "klm-quant-trie", "klm-array-trie", "klm-quant-array-trie" */
PPASR_API int          ppasr_lm_word_index(ppasr_lm_handle lm, int token); /* LM word index of an acoustic token, 0 = OOV */
PPASR_API int          ppasr_lm_bos(ppasr_lm_handle lm);
PPASR_API int          ppasr_lm_eos(ppasr_lm_handle lm);
/* Verification hooks of the model-file readers (no device is touched; the decoder never calls them):
 * ppasr_lm_debug_load_host parses a model into the host-side table only, ppasr_lm_debug_host_score evaluates
 * Scorer::get_log_cond_prob for a window of `order` LM word indices (oldest first) on that table. */
PPASR_API ppasr_status ppasr_lm_debug_load_host(const char* model_path, const char* const* vocab_utf8, int V, ppasr_lm_handle* out);
PPASR_API double       ppasr_lm_debug_host_score(ppasr_lm_handle lm, const int32_t* window);
PPASR_API ppasr_status ppasr_lm_destroy(ppasr_lm_handle lm);
PPASR_API int          ppasr_lm_order(ppasr_lm_handle lm);
PPASR_API int          ppasr_lm_is_character_based(ppasr_lm_handle lm);
PPASR_API long long    ppasr_lm_dict_size(ppasr_lm_handle lm);   /* Scorer::get_dict_size(): words in the dictionary (word-based models) */
PPASR_API int          ppasr_lm_space_id(ppasr_lm_handle lm);    /* acoustic token id of the space (" " or "<space>"), -1 if none */
PPASR_API long long    ppasr_lm_ngram_count(ppasr_lm_handle lm);
/* ppasr_ctc_beam_search with the scorer: alpha * ln P_lm(c | prefix) + beta on every extension, the min_cutoff pruning
 * of ctc_beam_search_decoder.cpp, and result scores = -(score - len*beta - alpha*ln P_lm(sentence)) ("approx_ctc").
 * lm == NULL: identical to ppasr_ctc_beam_search. */
PPASR_API ppasr_status ppasr_ctc_beam_search_lm(const float* probs, const int32_t* frame_lens, int B, int T, int V, int beam_size,
                                      double cutoff_prob, int cutoff_top_n, int blank, int nbest, int max_tokens,
                                      int32_t* tokens, int32_t* lens, double* scores, void* state, size_t state_bytes,
                                      int init_state, ppasr_lm_handle lm, double alpha, double beta, void* stream);
/* ... with caller-owned scratch of at least ppasr_ctc_beam_scratch_bytes(B, T, V, beam_size, cutoff_prob, cutoff_top_n)
 * bytes (may be NULL / 0 when that is 0).  The scratch is only used during the call. */
PPASR_API ppasr_status ppasr_ctc_beam_search_ws(const float* probs, const int32_t* frame_lens, int B, int T, int V, int beam_size,
                                      double cutoff_prob, int cutoff_top_n, int blank, int nbest, int max_tokens,
                                      int32_t* tokens, int32_t* lens, double* scores, void* state, size_t state_bytes,
                                      int init_state, ppasr_lm_handle lm, double alpha, double beta, void* scratch,
                                      size_t scratch_bytes, void* stream);

/* ---- hypothesis records of the utterance-parallel path (north_star: "a single RCCL all-gather ... for decoded
 * hypotheses"; no reference counterpart -- the reference has no multi-GPU inference, trainer.py:529-544 is training DP).
 * A record row is int32 [cols + extra]: tokens (-1 padded to cols) | n_tokens | score (f64, low word first)
 * [| utterance index when extra = 4].  ppasr_hyp_pack writes the k hypotheses of one decoder call (tokens [k][L] with
 * row stride token_stride; n_tokens / score with element strides, so that the n-best = 1 column of the beam search's
 * [B][nbest] outputs can be passed in place; index [k] or NULL) into rows row0 .. row0 + k - 1 of `rec`;
 * ppasr_hyp_unpack reads N rows (row order[r], or r when order is NULL) back into tokens [N][cols], n_tokens [N],
 * score [N] and (optionally) the index column.  One launch each, on `stream`. */
PPASR_API ppasr_status ppasr_hyp_pack(const int32_t* tokens, long long token_stride, int L, const int32_t* n_tokens,
                                      long long n_stride, const double* score, long long score_stride, const int32_t* index,
                                      int k, int32_t* rec, int row0, int cols, int extra, void* stream);
PPASR_API ppasr_status ppasr_hyp_unpack(const int32_t* rec, const int64_t* order, int N, int cols, int extra, int32_t* tokens,
                                        int32_t* n_tokens, double* score, int32_t* index, void* stream);

/* ---- streaming: ConformerModel.get_encoder_out_chunk (model_utils/conformer/model.py:164-184) =
 * ConformerEncoder.forward_chunk (conformer/encoder.py:208-283) + ctc softmax, as driven by
 * InferencePredictor.predict_chunk_conformer / reset_stream (inference_predictor.py:184-220) and
 * PPASRPredictor.predict_stream (predict.py:232-337).  The attention key/value cache and the conv-module
 * cache stay resident on the device inside the stream object; `offset` (inference_predictor.py:39,211)
 * is tracked by the object.  B = 1 per stream (encoder.py:238); run several streams for several sessions.
 * Any causal Conformer-family handle behind a conv front end, general-route handles (output_size 512 .. 1024,
 * ppasr_model_desc.options) included; use_cnn_module = 0 handles carry no conv cache (export: cnn_cache untouched). */
PPASR_API ppasr_status ppasr_stream_create(ppasr_handle h, ppasr_stream* out);
PPASR_API ppasr_status ppasr_stream_destroy(ppasr_stream s);
PPASR_API ppasr_status ppasr_stream_reset(ppasr_stream s, void* stream);
PPASR_API int ppasr_stream_offset(ppasr_stream s);        /* encoder frames emitted so far */
PPASR_API int ppasr_stream_cache_frames(ppasr_stream s);  /* cache_t1: key/value frames currently cached */
PPASR_API size_t ppasr_chunk_workspace_bytes(ppasr_handle h, int T);  /* (query it per T: it holds the chunk's activations, the
                                                                         cache-trim scratch and the front end's K-split tiles) */
/*   feats [1,T,F] f32; required_cache_size as in encoder.py:255-260 (<0 keep everything, the value
 *   predict_stream uses; 0 none; >0 last n frames); probs [1,c,V] or NULL; c = ((T-1)/2-1)/2 is also
 *   written to *c_out_host (host int, may be NULL). */
PPASR_API ppasr_status ppasr_encode_chunk(ppasr_stream s, const float* feats, int T, int required_cache_size, float* probs,
                                int32_t* frame_argmax, float* frame_maxprob, int* c_out_host, void* workspace,
                                size_t workspace_bytes, void* stream);
/* Reference tensor layouts of the caches: att_cache [L][h][t][2*dk] (h = attention_heads, dk = 64), cnn_cache [L][1][d][k-1]
 * (d = output_size). */
PPASR_API ppasr_status ppasr_stream_export_cache(ppasr_stream s, float* att_cache, float* cnn_cache, void* stream);
PPASR_API ppasr_status ppasr_stream_import_cache(ppasr_stream s, const float* att_cache, int cache_t, const float* cnn_cache,
                                       int offset, void* stream);

/* ---- DeepSpeech2: DeepSpeech2Model.get_encoder_out / get_encoder_out_chunk (model_utils/deepspeech2/model.py:62-72),
 * as called by trainer.py:626 and InferencePredictor.predict / predict_chunk_deepspeech
 * (inference_predictor.py:103-145,147-182).  Create the handle with model_type = PPASR_MODEL_DEEPSPEECH2,
 * output_size = rnn_size, num_blocks = num_rnn_layers, causal = 1 for the streaming ('forward') model and 0 for
 * the bidirectional one (deepspeech2/model.py:40).
 *   feats [B,T,F], lens [B] i64 -> probs [B,T',V] f32, out_lens [B] i64 (= ((len-1)/2-1)/2, may be NULL);
 *   init_h / init_c / final_h / final_c: [num_rnn_layers*dirs, B, rnn_size] state boxes (NULL = zeros / not wanted).
 * Synchronisation: asynchronous on `stream`, EXCEPT single-utterance LSTM calls (B = 1, rnn_size 1024) on the persistent
 * recurrence route: its launches need every workgroup of their grid resident at once, so the call ends with a stream
 * synchronisation and a 4-byte read-back of the kernels' give-up flag; when a launch gave up (the chip was shared with
 * another stream / process) the call re-runs on the per-step kernels before it returns, and the handle stays on those for
 * the next 64 .. 1 024 calls before it tries the persistent route again.  PPASR_DS2_PERSIST=0 switches the route off. */
PPASR_API size_t ppasr_ds2_workspace_bytes(ppasr_handle h, int B, int T);
/* Test hook, not part of any product path: `n_workgroups` workgroups that each take a whole CU (1 024 threads, 128
 * registers per lane) and spin for `milliseconds` on `stream`.  With part of the chip held like this the persistent
 * recurrence above cannot become fully resident and gives up: the tests check that the call then returns the per-step
 * kernels' result. */
PPASR_API ppasr_status ppasr_debug_occupy_cus(int n_workgroups, int milliseconds, void* stream);
PPASR_API ppasr_status ppasr_ds2_encode(ppasr_handle h, const float* feats, const int64_t* lens, int B, int T, const float* init_h,
                              const float* init_c, float* probs, int64_t* out_lens, float* final_h, float* final_c,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Measurement hook (bench.py roofline leg; no reference counterpart): when enabled, every kernel launch
 * of the next ppasr_encode is bracketed by a HIP event pair on the caller's stream; ppasr_profile_read
 * synchronises those events and returns, per kernel class, the summed duration (ms) and launch count
 * into HOST arrays of PPASR_N_KERNEL_CLASSES entries.  Conformer / Efficient-Conformer handles of width 256 only
 * (PPASR_EUNSUPPORTED otherwise); ppasr_kprof_* below covers every route by kernel name. */
#define PPASR_N_KERNEL_CLASSES 10
PPASR_API ppasr_status ppasr_profile_enable(ppasr_handle h, int enable);
PPASR_API ppasr_status ppasr_profile_read(ppasr_handle h, float* total_ms_host, int* launches_host);
PPASR_API const char* ppasr_kernel_class_name(int cls);

/* Kernel-name profiler (bench.py roofline leg for every model family and the decoders; no reference counterpart).
 * Between begin and end, every kernel the CALLING THREAD launches through this library carries a dispatch-attached HIP
 * event pair; end synchronises them and returns one entry per distinct kernel (the names rocprofv3's kernel trace
 * prints, without the parameter list): names_host [max_entries][PPASR_KPROF_NAME_LEN] chars, total_ms_host /
 * launches_host [max_entries], *n_out_host = entries written.  Not to be combined with ppasr_profile_enable. */
#define PPASR_KPROF_NAME_LEN 160
PPASR_API ppasr_status ppasr_kprof_begin(void);
PPASR_API ppasr_status ppasr_kprof_end(int max_entries, char* names_host, float* total_ms_host, int* launches_host, int* n_out_host);

/* Replaces greedy_decoder / greedy_decoder_batch (decoders/ctc_greedy_decoder.py:6-49), as called
 * from PPASRPredictor.decode (predict.py:128) and PPASRTrainer.__decoder_result (trainer.py:351).
 *   probs [B,Tp,V] f32 (any row-normalised or not: argmax + value at argmax)
 *   frame_lens [B] i32 or NULL (NULL = decode all Tp rows, the reference's batch behaviour)
 *   tokens [B,Tp] i32 (-1 padded), n_tokens [B] i32, score [B] f64 (mean non-blank max prob * 100) */
PPASR_API ppasr_status ppasr_ctc_greedy(const float* probs, const int32_t* frame_lens, int B, int Tp, int V, int blank,
                              int32_t* tokens, int32_t* n_tokens, double* score,
                              void* workspace /* >= 8*B*Tp bytes */, size_t workspace_bytes, void* stream);

/* Second stage only: collapse repeats / drop blank / score, from per-frame argmax+maxprob
 * (the outputs of ppasr_encode).  Same outputs as ppasr_ctc_greedy. */
PPASR_API ppasr_status ppasr_ctc_collapse(const int32_t* frame_argmax, const float* frame_maxprob, const int32_t* frame_lens,
                                int B, int Tp, int blank, int32_t* tokens, int32_t* n_tokens, double* score,
                                void* stream);

/* ---- multi-session streaming (no reference counterpart: PPASR streams one session per call) -----------------------
 * A group of Conformer sessions whose K/V and conv caches live in one allocation; ppasr_encode_chunk_group advances
 * any subset of them by one chunk with ONE set of launches (rows of all listed sessions stacked).  Every session
 * follows the single-session arithmetic of ppasr_encode_chunk with required_cache_size < 0 (full history, what
 * PPASRPredictor.predict_stream passes, predict.py:306-307).  max_frames caps the per-session cache (<= max_len).
 *   sessions_host [n] distinct slot indices; feats [n][T][F]; outputs by list position: probs [n][c][V] or NULL,
 *   frame_argmax / frame_maxprob [n][c] or NULL. */
typedef struct ppasr_stream_group_s* ppasr_stream_group;
PPASR_API ppasr_status ppasr_stream_group_create(ppasr_handle h, int n_sessions, int max_frames, ppasr_stream_group* out);
PPASR_API ppasr_status ppasr_stream_group_destroy(ppasr_stream_group g);
PPASR_API ppasr_status ppasr_stream_group_reset(ppasr_stream_group g, int session /* < 0: all */, void* stream);
PPASR_API int          ppasr_stream_group_offset(ppasr_stream_group g, int session);
PPASR_API size_t       ppasr_group_chunk_workspace_bytes(ppasr_handle h, int n, int T);
PPASR_API ppasr_status ppasr_encode_chunk_group(ppasr_stream_group g, const int* sessions_host, int n, const float* feats, int T,
                                      float* probs, int32_t* frame_argmax, float* frame_maxprob, int* c_out_host,
                                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- Kaldi-compatible fbank front-end --------------------------------------------
 * Replaces AudioFeaturizer.featurize (ppasr/data_utils/featurizer/audio_featurizer.py:37-67,120-138):
 * AudioSegment.normalize(target_dB) (data_utils/audio.py:287-304) -> .to('int16') (audio.py:244) ->
 * paddleaudio.compliance.kaldi.fbank(n_mels, frame_length=25, frame_shift=10, dither=0, sr) (third-party,
 * paddleaudio>=1.0.1).  samples: device f32 mono in [-1,1]; feats: device f32 [frames][n_mels]. */
typedef struct ppasr_fbank_s* ppasr_fbank_handle;
PPASR_API ppasr_status ppasr_fbank_create(int sample_rate, int n_mels, float frame_length_ms, float frame_shift_ms,
                                ppasr_fbank_handle* out);
PPASR_API ppasr_status ppasr_fbank_destroy(ppasr_fbank_handle f);
PPASR_API int          ppasr_fbank_frames(ppasr_fbank_handle f, int n_samples);          /* snip_edges frame count */
/* (workspace: one float per 8192-sample chunk of the mean square -- summed in numpy's order, see csrc/fbank.hip -- + the gain
 *  + the gain in dB (ws[chunks], ws[chunks + 1]: the host mirror raises ValueError beyond 300 dB like audio.py:301); never less
 *  than 8 KiB) */
PPASR_API size_t       ppasr_fbank_workspace_bytes(ppasr_fbank_handle f, int n_samples);
PPASR_API ppasr_status ppasr_fbank_compute(ppasr_fbank_handle f, const float* samples, int n_samples, int use_db_norm,
                                 float target_db, float* feats, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PPASR_HIP_H */
