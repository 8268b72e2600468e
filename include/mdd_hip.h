/*
 * mdd_hip.h -- C ABI of libmdd_hip.so: the MI355X (gfx950) implementation of the
 * CTC-attention mispronunciation-detection hot path.
 *
 * The reference (dyustc/CTC-Attention-Mispronunciation, egs/attention_aug = "AA") has no FFI
 * layer: its seam is Python objects.  Each entry point below names the reference interface it
 * replaces; INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross the boundary
 *   - pointers named *_dev are device (HBM) addresses, everything else is host memory
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all GPU work is
 *     enqueued on it and nothing synchronises unless a function says so
 *   - every function returns 0 on success or a negative mdd_status; mdd_last_error() gives text
 *   - a handle is not thread-safe: one handle per host thread / stream / device
 *   - the caller owns every buffer it passes; the library owns device weights and a workspace
 *     that grows on demand (never inside a stream capture)
 */
#ifndef MDD_HIP_H
#define MDD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdd_model mdd_model;

enum mdd_status {
    MDD_OK = 0,
    MDD_ERR_ARG = -1,       /* bad argument / shape */
    MDD_ERR_HIP = -2,       /* a HIP runtime call failed */
    MDD_ERR_STATE = -3,     /* weights missing / not finalised */
    MDD_ERR_NOMEM = -4,
    MDD_ERR_EMPTY = -5      /* mdd_align with an empty side: the reference raises TypeError */
};

/* per-utterance status written by mdd_beam: the exception the reference's BeamDecoder.decode
 * would raise for that utterance (AA/utils/BeamSearch.py:64,66,103,106,135; AA/utils/NgramLM.py:75-76) */
enum mdd_beam_status { MDD_BEAM_OK = 0, MDD_BEAM_INDEX_ERROR = 1, MDD_BEAM_VALUE_ERROR = 2, MDD_BEAM_KEY_ERROR = 3 };

/* Geometry of CTC_Model.__init__ (AA/models/model_ctc.py:84-158) for the one architecture the
 * reference recipe builds: 2x LayerCNN(k3x3, strides (1,2),(2,2), pad 1) -> `layers` x BiLSTM(hidden)
 * -> Embedding(emb_rows, emb_dim) + BiLSTM text encoder -> dot attention -> BN + Linear(num_class). */
typedef struct mdd_config {
    int32_t feat;       /* stacked input width F (243 = 3 x 81), rnn_param["rnn_input_size"] */
    int32_t hidden;     /* rnn_hidden_size H (384; 256 for BASELINE.json's variant); multiple of 16 */
    int32_t layers;     /* rnn_layers (4) */
    int32_t num_class;  /* C (45 for the 41-phone set) */
    int32_t channels;   /* CNN channels (32) */
    int32_t emb_rows;   /* 44  (model_ctc.py:149) */
    int32_t emb_dim;    /* 512 (model_ctc.py:149-150); multiple of 4 */
    float bn_eps;       /* 1e-5 */
} mdd_config;

const char *mdd_last_error(void);
int mdd_version(void);

/* ---- model lifetime + weights: replaces CTC_Model(...) + load_state_dict (AA/infer.py:251-254) */
int mdd_create(const mdd_config *cfg, int device, mdd_model **out);
void mdd_destroy(mdd_model *m);
/* Copy one state_dict entry (host fp32, contiguous, reference key name and shape) to the device.
 * `num_batches_tracked` entries are accepted and ignored. */
int mdd_load_weight(mdd_model *m, const char *key, const float *data, const int64_t *shape, int32_t ndim);
/* Check that all 55 float entries arrived, fold eval-mode BatchNorm into scale/shift vectors and
 * repack LSTM gate rows for the step kernel.  Synchronises the device. */
int mdd_finalize_weights(mdd_model *m);

/* Arithmetic of the model's contractions.  Modes 2 and 0 are reference width (the arithmetic of the reference's ATen fp32 ops,
 * AA/models/model_ctc.py:27-29,59-66,149-158: every operand with its full 24-bit significand, fp32 accumulation); measured against a
 * float64 evaluation both are CLOSER to it than ATen's own fp32 (tests/test_gpu_parity.py::test_error_against_fp64_beside_aten_fp32).
 * 2 (default) = "f32x6": the large time-batched contractions -- conv0 / conv1 and the BiLSTM / text input projections -- on the bf16
 *     matrix cores with every fp32 operand carried as THREE bf16 planes (hi + mid + lo = its 24 significand bits exactly) and the six
 *     cross products down to 2^-24 of a product, fp32 accumulate, hi.hi in an accumulator of its own (gemm_bf16x6.hip: 6/16 of the cost
 *     of the fp32 MFMA, which on gfx950 runs at the fp32 vector rate); so do the recurrent W_hh.h products where that is the faster of
 *     the two reference-width layer kernels (lstm_x6.hip: W_hh and the state h as three planes each; H = 384 up to 1024 rows, H = 256
 *     up to 128 rows; env MDD_LSTM_X6=0 / force); everything else as mode 0.  Needs contraction lengths that are multiples of 32 (falls
 *     back to 0 otherwise).
 * 0 = every product an exact fp32 MFMA (v_mfma_f32_32x32x2_f32 in the time-batched GEMMs, v_mfma_f32_16x16x4_f32 in the recurrent
 *     W_hh.h products, the attention tail and the convolutions).
 * 1 = split-bf16 "x3" on v_mfma_f32_16x16x32_bf16: each fp32 operand = bf16 hi + bf16 lo (~16 significand bits), products hi.lo +
 *     lo.hi + hi.hi, for EVERY contraction of the forward including the recurrent W_hh.h products (h is re-split every step); cell state,
 *     gates, softmax and the classifier tail stay fp32.  Measured effect on the log-probs <= 1e-5 (tolerance 1e-4).  NARROWER than the
 *     reference's arithmetic: a flagged variant.  Falls back to 0 when a contraction length is not a multiple of 32 or H is not 256 / 384.
 * Env MDD_PRECISION=f32x6 / f32 / bf16x3 selects the mode at mdd_create.  mdd_get_precision returns the mode actually in use. */
int mdd_set_precision(mdd_model *m, int32_t mode);
int32_t mdd_get_precision(mdd_model *m);

/* ---- A1: make_context(feat,0,right) + skip_feat(.,skip) + pad to a multiple of n_down
 * (AA/utils/tools.py:207-227, AA/utils/data_loader.py:138-142) for B equal-length utterances.
 * raw_dev [B,T_raw,D] -> out_dev [B,T_out,(right+1)*D] with T_out = mdd_stack_len(T_raw,skip,n_down). */
int32_t mdd_stack_len(int32_t T_raw, int32_t skip, int32_t n_down);
int mdd_stack_skip(const float *raw_dev, int32_t B, int32_t T_raw, int32_t D, int32_t right, int32_t skip,
                   int32_t n_down, float *out_dev, void *stream);
/* float32 length bookkeeping of create_input / infer (data_loader.py:177, infer.py:296-297) -- host */
int32_t mdd_len_frames(int32_t len, int32_t maxlen, int32_t t_out);

/* ---- A2-A7: CTC_Model.forward(x, x1) in eval mode (AA/models/model_ctc.py:160-223)
 * x_dev [B,T,F] fp32 (T even), x1_dev [B,L] int64 canonical ids (0-padded) ->
 * logp_dev [T/2,B,C] fp32 log-probabilities.  ids outside [0,emb_rows) are an error
 * (the reference raises IndexError) and are reported by the next mdd_sync(). */
int mdd_forward(mdd_model *m, const float *x_dev, int32_t B, int32_t T, const int64_t *x1_dev, int32_t L,
                float *logp_dev, void *stream);
/* ---- A2-A7 for several reference batches of different padded lengths in ONE launch sequence.  The reference pads each
 * batch to its own maximum and masks nothing (model_ctc.py:186,198,204-205; collate AA/utils/data_loader.py:151-181), so an
 * utterance's posteriors depend on its batch's padded length.  frames_dev[b] = T_g / 2 (posterior frames) and canon_dev[b] = L_g
 * (canonical length) of the batch utterance b belongs to; x_dev [B,T,F] holds every batch zero-padded to the common T (zero from
 * its own T_g on), x1_dev [B,L] zero-padded to the common L.  Every utterance's rows t < frames_dev[b] of logp_dev [T/2,B,C] are
 * bit-identical to mdd_forward on its batch alone; rows beyond are undefined. */
int mdd_forward_fused(mdd_model *m, const float *x_dev, int32_t B, int32_t T, const int64_t *x1_dev, int32_t L,
                      const int32_t *frames_dev, const int32_t *canon_dev, float *logp_dev, void *stream);

/* ---- A1 + A2..A7 in one call: raw_dev holds the unstacked frames [B, T_raw, feat/3] (make_context(.,0,2) + skip_feat(.,2)
 * + even padding are applied on the fly: AA/utils/tools.py:207-227, AA/utils/data_loader.py:138-142); logp_dev is
 * [mdd_stack_len(T_raw,2,2)/2, B, C].  Same results, bit for bit, as mdd_stack_skip followed by mdd_forward. */
int mdd_forward_raw(mdd_model *m, const float *raw_dev, int32_t B, int32_t T_raw, const int64_t *x1_dev, int32_t L,
                    float *logp_dev, void *stream);

/* The same forward replayed stage by stage between HIP events on `stream` (measurement aid for bench.py:
 * per-stage wall time, kernel launches and algorithmic flops).  names: comma-separated stage names. */
int32_t mdd_forward_num_stages(mdd_model *m);
int mdd_forward_profile(mdd_model *m, const float *x_dev, int32_t B, int32_t T, const int64_t *x1_dev, int32_t L,
                        float *logp_dev, void *stream, char *names, int32_t names_cap, float *ms, int32_t *launches,
                        double *flops, int32_t cap);
/* Optional taps for parity tests: copies of stage outputs of the last mdd_forward (device buffers,
 * valid until the next forward).  name: "conv1" [T/2,B,ch*W2], "rnn<i>" [T/2,B,2H] (raw, before the
 * next layer's BatchNorm), "text" [L,B,2H], "key" [L,B,2H].  Returns the device pointer or NULL. */
const float *mdd_tap(mdd_model *m, const char *name, int64_t *numel);
/* Same, copied device-to-device into a caller buffer of `capacity` floats on `stream`. */
int mdd_tap_copy(mdd_model *m, const char *name, float *dst_dev, int64_t capacity, void *stream);
/* Keep the raw output of every BiLSTM layer (off by default: only the last layer's is needed). */
int mdd_enable_taps(mdd_model *m, int32_t on);
/* Wait for `stream` and report asynchronous errors of earlier calls on this handle. */
int mdd_sync(mdd_model *m, void *stream);

/* ---- A8: GreedyDecoder.decode (AA/utils/ctcDecoder.py:188-200, 80-92)
 * logp_dev [T,B,C], len_dev [B] -> ids_dev [B,T] (collapsed, blanks removed), nids_dev [B]. */
int mdd_greedy(const float *logp_dev, int32_t T, int32_t B, int32_t C, const int32_t *len_dev, int32_t blank,
               int32_t *ids_dev, int32_t *nids_dev, void *stream);

/* ---- A9: BeamDecoder.decode -> ctcBeamSearch.decode (AA/utils/ctcDecoder.py:215-226,
 * AA/utils/BeamSearch.py:73-153): CTC prefix beam search, float64 scores.
 * lm_dev: dense (C+1)x(C+1) table of natural-log bigram scores T[prev][next] as
 * LanguageModel.get_bi_prob returns them (prev == C: sentence start, next == C: sentence end),
 * NaN where the reference would raise KeyError.  beam <= 64, C <= 256.
 * Outputs: ids_dev [B,T], nids_dev [B], status_dev [B] (mdd_beam_status), score_dev [B] or NULL
 * (length-normalised score of the winner). */
int mdd_beam(const float *logp_dev, int32_t T, int32_t B, int32_t C, const int32_t *len_dev, int32_t beam,
             int32_t blank, const double *lm_dev, double lm_alpha, int32_t *ids_dev, int32_t *nids_dev,
             int32_t *status_dev, double *score_dev, void *stream);

/* ---- A12: nn.CTCLoss(reduction='sum') pieces (AA/steps/train_ctc.py:72,186): alpha/beta lattice.
 * logp_dev [T,B,C], targets_dev [B,Lmax] int64 (padded), in_len_dev/tgt_len_dev [B] int64 ->
 * nll_dev [B] (per-utterance negative log-likelihood; the reference's loss is their sum; +inf for an utterance with no
 * valid alignment, as the reference gives without zero_infinity) and, if grad_dev != NULL, the tensor autograd deposits
 * on the log-probs: [T,B,C] (zero on frames >= in_len).  A target label outside [0,C) makes that utterance's nll NaN
 * and its gradient rows zero (ATen would index past the row).
 * workspace_dev: caller-owned scratch of at least mdd_ctc_workspace_bytes(T,B,C,Lmax, grad_dev != NULL) bytes (16-byte
 * aligned; the alpha/beta rows of the backward sweep); NULL = the library allocates stream-ordered for the call. */
int64_t mdd_ctc_workspace_bytes(int32_t T, int32_t B, int32_t C, int32_t Lmax, int32_t want_grad);
int mdd_ctc_loss(const float *logp_dev, int32_t T, int32_t B, int32_t C, const int64_t *targets_dev, int32_t Lmax,
                 const int64_t *in_len_dev, const int64_t *tgt_len_dev, int32_t blank, float *nll_dev,
                 float *grad_dev, void *workspace_dev, int64_t workspace_bytes, void *stream);

/* ---- A10: Decoder.wer core = _edit_distance + printChanges (AA/utils/ctcDecoder.py:118-184), host.
 * a = hypothesis tokens, b = canonical tokens; ops (capacity >= na+nb): 0 '-', 1 'S', 2 'I', 3 'D'.
 * Either side empty -> MDD_ERR_EMPTY (reference: TypeError). */
int mdd_align(const int32_t *a, int32_t na, const int32_t *b, int32_t nb, int32_t *dist, uint8_t *ops,
              int32_t *nops);
/* The same for the n utterances of a batch in one call (the reference loops over the batch calling decoder.wer per
 * utterance, AA/steps/test_ctc_nosil.py:218-221, AA/infer.py:318-331): row x of a / b (row pitch a_stride / b_stride
 * ids) with a_len[x] / b_len[x] ids; dist[x], nops[x] and row x of ops (pitch ops_stride >= a_len[x] + b_len[x]).
 * A row with an empty side gets dist[x] = -1, nops[x] = 0 (the reference's wer raises TypeError for it) and does not
 * fail the call. */
int mdd_align_batch(const int32_t *a, const int32_t *a_len, int32_t a_stride, const int32_t *b, const int32_t *b_len,
                    int32_t b_stride, int32_t n, int32_t *dist, uint8_t *ops, int32_t ops_stride, int32_t *nops);

/* ---- Kaldi-compatible log-mel filterbank + global CMVN (replaces the reference's subprocess pipe
 * `compute-fbank-feats --config=conf/fbank.conf | apply-cmvn --norm-vars=true data/global_fbank_cmvn.txt`,
 * AA/infer.py:567-574; options of AA/conf/fbank.conf:1-4 over Kaldi's defaults, dither 0).
 * wav_dev: n_samples mono samples at 16 kHz on the int16 scale, as float.  out_dev: [mdd_fbank_num_frames(n), 81],
 * column 0 = raw log energy, 1..80 = log mel energies; if both cmvn pointers are non-NULL (81 floats each, device)
 * every column c is written as value * scale[c] + offset[c].  Parity with Kaldi is unpinned. */
int32_t mdd_fbank_num_frames(int64_t n_samples);
int mdd_fbank(const float *wav_dev, int64_t n_samples, const float *cmvn_scale_dev, const float *cmvn_offset_dev,
              float *out_dev, void *stream);

/* ---- evaluation counts of a batch (AA/steps/test_ctc_nosil.py:33-60,218-298), host.
 * Row x of dec / lab / can (row pitch `stride` ids) holds the decoded, annotated and canonical phoneme ids of utterance x
 * with 'sil' already removed (:196-209).  counts[8] = { phonemes in canonical, TA, FR, FA, TR correctly diagnosed,
 * TR wrongly diagnosed, sum of edit distances decoded vs annotated, annotated phonemes }.  Any empty sequence ->
 * MDD_ERR_EMPTY (the reference's loop dies with TypeError there). */
int mdd_eval_batch(const int32_t *dec, const int32_t *dec_len, const int32_t *lab, const int32_t *lab_len,
                   const int32_t *can, const int32_t *can_len, int32_t n, int32_t stride, int64_t *counts);

/* ---- SURVEY 8(f) #3 / BASELINE config 5: one training step of run_epoch (AA/steps/train_ctc.py:28-105).
 * The drop-in CTC_Model keeps its parameters as torch tensors; every call receives their device pointers.
 *   mdd_train_create            geometry as mdd_create; the handle owns the activations saved between forward and backward
 *   mdd_train_tensor_info       key / element count / "is a BatchNorm running-statistics buffer" of tensor i: the 55 float entries
 *                               of CTC_Model.state_dict() (AA/models/model_ctc.py:84-158), in state_dict order
 *   mdd_train_forward           CTC_Model.forward in TRAIN mode (model_ctc.py:160-223): BatchNorm on batch statistics (biased
 *                               variance; running statistics updated in place with momentum 0.1 and the unbiased variance),
 *                               Dropout(p_drop) behind each LayerCNN and BatchRNN.  masks: NULL (masks drawn from `seed` by a
 *                               counter-based generator) or mdd_train_num_masks() device byte arrays (1 = keep) laid out like the
 *                               reference's tensor at that site -- [B,ch,T,W1], [B,ch,T/2,W2], then [T/2,B,2H] per BatchRNN --
 *                               of mdd_train_mask_bytes() bytes each.  x_dev / x1_dev must stay valid until the backward call.
 *   mdd_train_backward          autograd's backward of that forward: dlogp_dev [T/2,B,C] (e.g. mdd_ctc_loss's gradient scaled
 *                               by 1/B as train_ctc.py:73-74 divides the loss) -> one gradient per parameter into grads[i]
 *                               (entries of running-statistics buffers are ignored and may be NULL)
 *   mdd_adam_step               torch.optim.Adam over n tensors (train_ctc.py:187: lr 1e-3, weight_decay 5e-4 added to the gradient)
 * Arithmetic is exact fp32 (v_mfma_f32_*), as the reference trains; mdd_train_set_precision(w, 1) (or MDD_TRAIN_PRECISION=bf16x3 at create)
 * sends the large projections and their two backward products through the split-bf16 x3 matrix-core GEMM of the decode path instead
 * (operands to 16 mantissa bits, fp32 accumulate); everything else stays fp32 in both modes. */
typedef struct mdd_train_ws mdd_train_ws;
int mdd_train_create(const mdd_config *cfg, int device, mdd_train_ws **out);
int mdd_train_set_precision(mdd_train_ws *w, int32_t mode);   /* 0 exact fp32 (default), 1 split-bf16 x3 contractions */
void mdd_train_destroy(mdd_train_ws *w);
int32_t mdd_train_num_tensors(mdd_train_ws *w);
int mdd_train_tensor_info(mdd_train_ws *w, int32_t i, char *key, int32_t cap, int64_t *numel, int32_t *is_buffer);
int32_t mdd_train_num_masks(mdd_train_ws *w);
int64_t mdd_train_mask_bytes(mdd_train_ws *w, int32_t site, int32_t B, int32_t T);
int mdd_train_forward(mdd_train_ws *w, float *const *tensors, const float *x_dev, int32_t B, int32_t T, const int64_t *x1_dev, int32_t L,
                      const uint8_t *const *masks, uint64_t seed, float p_drop, float *logp_dev, void *stream);
int mdd_train_backward(mdd_train_ws *w, float *const *tensors, const float *dlogp_dev, float *const *grads, void *stream);
int mdd_train_sync(mdd_train_ws *w, void *stream);
int mdd_adam_step(float *const *params, float *const *grads, float *const *exp_avg, float *const *exp_avg_sq, const int64_t *numel, int32_t n,
                  int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay, void *stream);

/* ---- Diagnostics (test and measurement aid; no reference counterpart).
 * mdd_diag_gemm_ph8: race screen of the 8-phase projection GEMM -- the same pseudo-random operands through the single-barrier
 * kernel once and the 8-phase kernel (both DMA placements) `reps` times each; *mismatches_out = C words that ever
 * differed (must be 0; tests/test_gpu_parity.py::test_gemm_8phase_race_screen).  ms_out (nullable, 16 floats): mean kernel times. */
int mdd_diag_gemm_ph8(int M, int N, int K, int reps, unsigned seed, unsigned *mismatches_out, float *ms_out);
/* mdd_diag_gates: the gate nonlinearities of the reference-width recurrences (csrc/lstm_persist.h) evaluated on n device floats:
 * sig_dev[i] = sigmoid(x_dev[i]), tanh_dev[i] = tanh(x_dev[i]) (tests/test_gpu_parity.py::test_gate_functions_accuracy). */
/* mdd_diag_gemm: C_dev[M,N] = A_dev[M,K] . W_dev[N,K]^T through one arithmetic (0 exact fp32 MFMA, 1 split-bf16 x3, 2 the f32x6
 * prototype, 3 the f32x6 kernel), fp32 operands and result on the device; synchronises (tests/test_gpu_parity.py::test_gemm_f32x6_accuracy). */
int mdd_diag_gemm(int mode, const float *A_dev, const float *W_dev, float *C_dev, int M, int N, int K, void *stream);
/* mdd_diag_gemm_time: mean milliseconds of `reps` launches of one GEMM kernel on resident, pre-split pseudo-random operands. */
int mdd_diag_gemm_time(int mode, int M, int N, int K, int reps, float *ms_out);
int mdd_diag_gates(const float *x_dev, float *sig_dev, float *tanh_dev, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MDD_HIP_H */
