/* fvta_hip.h -- C ABI of libfvta_hip.so: the FVTA hot path on MI355X (gfx950).
 *
 * The reference (JunweiLiang/FVTA_MemexQA) has no FFI/plugin layer: its hot path
 * is a TensorFlow-1 graph built in model_v2.py and entered through one
 * `sess.run` per step (trainer.py:36-38, tester.py:21).  This header is the
 * boundary a maintainer would bind instead (ctypes stub in INTEGRATION.md).  Each
 * entry point names the reference code it replaces (file:line in the reference
 * repository).
 *
 * Conventions
 *  - extern "C"; every call returns an int status (FVTA_OK or a negative
 *    FVTA_ERR_*); nothing throws; fvta_last_error() gives the message.
 *  - every buffer is a CALLER-OWNED DEVICE pointer, row-major, innermost =
 *    hidden/feature axis, 16-byte aligned.  The library never allocates:
 *    scratch is passed in (`*_bytes` queries size it) and may be reused
 *    between calls; "saved" buffers carry forward state to the backward call.
 *  - every call is asynchronous on the given hipStream_t, re-entrant, and
 *    keeps no global state.
 *  - masks are bytes (0 / non-0), as the reference feeds bool arrays
 *    (model_v2.py:1171-1200).
 */
#ifndef FVTA_HIP_H
#define FVTA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FVTA_OK 0
#define FVTA_ERR_INVALID_ARG (-1)
#define FVTA_ERR_LAUNCH (-2)
#define FVTA_ERR_UNSUPPORTED (-3)

#define FVTA_F32 0  /* exact fp32 arithmetic (v_mfma_f32_32x32x2_f32 + VALU) */
#define FVTA_BF16 1 /* bf16 MFMA operands, fp32 accumulate (BASELINE.json configs[2]) */
#define FVTA_BF16X3 2 /* bi-LSTM only: every MFMA operand split in two bf16 terms, three products per GEMM (hi hi + hi lo +
                         lo hi, fp32 accumulate; what is dropped is <= 3 * 2^-17 of |a||b| per product), saved gates fp32:
                         the 1e-4 parity engine on the bf16 matrix pipe (model_v2.py:652-661 computes in fp32) */

typedef void* fvta_stream_t; /* hipStream_t */

int fvta_version(void);
const char* fvta_last_error(void);
/* sizeof of a descriptor struct as the library was compiled: which = 0 fvta_attn_desc, 1 fvta_lstm_desc,
 * 2 fvta_scorer_desc, 3 fvta_timewarp_desc, 4 fvta_embed_desc, 5 fvta_imgtrans_desc; -1 otherwise.  A binding compares
 * its own layout with it when it loads the library (fvta_memexqa_amd/_lib.py does). */
int64_t fvta_abi_struct_bytes(int32_t which);

/* ------------------------------------------------------------------------- *
 * Focal attention: model_v2.py:210-298 `attention_3d` (K modalities) and
 * model_v2.py:125-201 / model.py:117-186 `attention` (call with K = 1),
 * including the helpers they inline: `linear` 75-100, `exp_mask`
 * utils.py:210-213, `softsel`/`softmax` model_v2.py:23-48.
 *
 *   hinfo [N,K,T,w]  hq [N,JQ,w]  hmask [N,K,T]  qmask [N,JQ]  (masks may
 *   both be NULL = unmasked, model_v2.py:146/233)
 *   W [F], b [1]: att_logits/{W,b}; F = 3w|2w|4w for simi 1|2|3 (feature
 *   order of model_v2.py:242-248; feat_order=1 selects model.py:149's order
 *   for simi 2); simi 4 (cosine, model_v2.py:250-254) takes W = b = NULL.
 *   out: h_a [N,w]; a_logits [N,K,T,JQ] masked logits (NULL = not wanted;
 *   only tester.py:34-36 `step_vis` reads them).
 * ------------------------------------------------------------------------- */
typedef struct fvta_attn_desc {
  int32_t N, K, T, JQ, w;
  int32_t simi;       /* 1..4 */
  int32_t feat_order; /* 0: model_v2.py order, 1: model.py:149 order (simi 2 only) */
  int32_t add_tanh;   /* model_v2.py:92-93 via linear(add_tanh=) */
  /* Elements between consecutive batch rows n of hinfo / d_hinfo; 0 = dense (K*T*w).  Non-zero needs K == 1 and no
   * tscale: the 1-D attention of model.py:117-186 run on ONE stream of a context arena laid out [N][all streams' rows]
   * (model.py:836-846, per-stream attention; :929-944 runs over the concatenation of the same rows). */
  int64_t hinfo_stride;
} fvta_attn_desc;

size_t fvta_attn_workspace_bytes(const fvta_attn_desc* d);
size_t fvta_attn_saved_bytes(const fvta_attn_desc* d);

int fvta_attn_fwd(const fvta_attn_desc* d, const float* hinfo, const float* hq, const uint8_t* hmask,
                  const uint8_t* qmask, const float* W, const float* b, float* h_a, float* a_logits,
                  void* saved, void* workspace, fvta_stream_t stream);

/* Backward of fvta_attn_fwd given d_h_a [N,w].  accumulate = 0: d_hinfo [N,K,T,w]
 * and d_hq [N,JQ,w] are overwritten (masked rows of d_hinfo get zeros);
 * 1: both are accumulated into; 2: d_hq is accumulated into and only the VALID
 * rows of d_hinfo are written (plain stores, masked rows left untouched: the
 * encoders never read them -- this is what the fused model uses, it saves the
 * clear and the read-modify-write of the largest tensor of the step); 3: d_hq
 * accumulated into, d_hinfo overwritten with zeros on masked rows.  dW [F]
 * and db [1] are always accumulated into (slices of the flat gradient buffer).  Gradient routing through reduce_max goes to the first arg-max
 * (model_v2.py:268,278; TF splits exact ties). */
int fvta_attn_bwd(const fvta_attn_desc* d, const float* hinfo, const float* hq, const uint8_t* hmask,
                  const uint8_t* qmask, const float* W, const float* b, const float* d_h_a,
                  const void* saved, float* d_hinfo, float* d_hq, float* dW, float* db, int accumulate,
                  void* workspace, fvta_stream_t stream);

/* fvta_attn_fwd / fvta_attn_bwd over the encoders' bf16 SHADOW rows instead of an fp32 hinfo (model_v2.py:863-914's
 * context tensor is the concatenation of the two directions' outputs the bi-LSTM has already written as bf16):
 * table [2][N*K*T] device addresses, table[half][(n K + k) T + t] -> the w/2 bf16 values that are channels
 * half * w/2 .. of row (n,k,t) (fvta_lstm_shadow_rows; EVERY entry must be readable: rows no encoder writes point at
 * w/2 zeros).  JQ <= 32, w = 512 or 1024, simiMatrix 1-3, no hinfo_stride, no a_logits; FVTA_ERR_INVALID_ARG otherwise.  Results
 * are those of fvta_attn_fwd / fvta_attn_bwd run on the bf16-rounded rows.  `saved` / `workspace` sizes as above. */
int fvta_attn_fwd_shadow(const fvta_attn_desc* d, const uint64_t* table, const float* hq, const uint8_t* hmask,
                         const uint8_t* qmask, const float* W, const float* b, float* h_a, void* saved, void* workspace,
                         fvta_stream_t stream);
int fvta_attn_bwd_shadow(const fvta_attn_desc* d, const uint64_t* table, const float* hq, const uint8_t* hmask,
                         const uint8_t* qmask, const float* W, const float* b, const float* d_h_a, const void* saved,
                         float* d_hinfo, float* d_hq, float* dW, float* db, int accumulate, void* workspace,
                         fvta_stream_t stream);

/* attention_3d(..., time_warp_att=True, C=C): model_v2.py:269-275.  The max-pooled logits are multiplied by the
 * row sums of C before the softmax over t: tscale [N,T] = sum_t' C[n,t,t'] (with the model's time warp, c[n,t] cnt(t) =
 * fvta_timewarp_fwd's scale_out).  The softmax over K keeps the unscaled maxima.  The reference scales AFTER exp_mask,
 * so a masked row's logit is -1e30 * tscale: for tscale < 0 the masked rows take the softmax; the kernels
 * reproduce that.  tscale NULL = fvta_attn_fwd / fvta_attn_bwd.  d_tscale [N,T] is accumulated into. */
int fvta_attn_fwd_tw(const fvta_attn_desc* d, const float* hinfo, const float* hq, const uint8_t* hmask,
                     const uint8_t* qmask, const float* W, const float* b, const float* tscale, float* h_a,
                     float* a_logits, void* saved, void* workspace, fvta_stream_t stream);
int fvta_attn_bwd_tw(const fvta_attn_desc* d, const float* hinfo, const float* hq, const uint8_t* hmask,
                     const uint8_t* qmask, const float* W, const float* b, const float* tscale, const float* d_h_a,
                     const void* saved, float* d_hinfo, float* d_hq, float* dW, float* db, float* d_tscale,
                     int accumulate, void* workspace, fvta_stream_t stream);

/* ------------------------------------------------------------------------- *
 * bi-LSTM modality encoders: model_v2.py:652-661 (cells), 667-678 (lengths),
 * 694/732/747/760/774/789/802/823 (the eight bidirectional_dynamic_rnn calls)
 * and the context-tensor assembly 863-914 (outputs are written straight into
 * the padded `hall` layout, so tf.pad/tf.stack never happen).
 *
 * One call runs every sequence that shares a cell (all 7 text streams of
 * model_v2.py:692-812 in ONE call; the photo stream in a second call with
 * `cell_img`).  Sequence s reads x + x_off[s] (+ t*in) and writes
 * out + out_off[s] (+ t*out_ld): forward half at [0,d), backward half at
 * [d,2d) of each output row; rows t >= len[s] (up to seq_J[s]) are zeroed as
 * dynamic_rnn does.
 * ------------------------------------------------------------------------- */
typedef struct fvta_lstm_desc {
  int32_t B;           /* sequences */
  int32_t J;           /* max padded steps over the call */
  int32_t in;          /* input features, multiple of 4 */
  int32_t d;           /* hidden size, multiple of 32 */
  int32_t share_fw_bw; /* 1: TF>=1.2 cell reuse, one kernel for both directions */
  int32_t precision;   /* FVTA_F32 | FVTA_BF16 | FVTA_BF16X3 */
  int32_t training;    /* 1: keep gate activations for fvta_bilstm_bwd */
  int32_t reserved;    /* profiling tag: this call's brackets are filed under id + 16*reserved */
  int32_t dx_overwrite; /* fvta_bilstm_bwd*: 0 = dx is accumulated into (the caller zeroes it), 1 = dx is WRITTEN: every row
                        * t < seq_J of every sequence (both copies under fvta_lstm_plan_xdir) holds the input gradient
                        * afterwards, zeros at t >= len, whatever it held before -- the caller's memset and, where the
                        * kernel writes both directions' sum at once, the read of dx go away */
  int32_t out_pads_persist; /* fvta_bilstm_fwd: 1 = the caller promises that nothing but this op writes `out` between forward
                        * calls on this plan memory: only the rows the previous call wrote and this one does not are zeroed
                        * (rows t >= len are zero after every call either way).  Under this flag the forward WRITES the plan
                        * (per sequence: how far the last forward wrote, into which `out`), although it takes it as const:
                        * (1) one plan must not be used by two forwards at once (two streams would race on that state);
                        * (2) the state is keyed by the output row's ADDRESS -- if `out` is freed and another buffer with
                        *     other contents comes back at the same address while the plan memory is kept, stale rows are not
                        *     zeroed: zero the plan memory (fvta_lstm_plan_bytes) whenever `out` is re-allocated, which also
                        *     is the way to invalidate the state; (3) the plan memory must start zeroed. */
  int64_t out_skip;    /* fvta_bilstm_fwd (FVTA_BF16 only): output half-rows at element offsets BELOW this are not stored into
                        * `out` (nor zeroed) -- their readers take the bf16 shadow rows the forward writes anyway
                        * (fvta_lstm_shadow_rows; the focal attention through fvta_attn_fwd_shadow / fvta_attn_bwd_shadow).
                        * 0: every row is stored.  The fp32 store of h is the single largest store of the forward step
                        * (4 of its 18 bytes per row and unit; the step is store-bound: 119 -> 100 us per launch). */
} fvta_lstm_desc;

size_t fvta_lstm_plan_bytes(const fvta_lstm_desc* d);
size_t fvta_lstm_saved_bytes(const fvta_lstm_desc* d);
size_t fvta_lstm_workspace_bytes(const fvta_lstm_desc* d);

/* Length-sorted schedule for one call (device side, no host sync).
 * len [B] int32 (mask row sums, model_v2.py:667-678), seq_J [B] padded length
 * of each sequence, x_off/out_off [B] element offsets, out_ld row stride. */
int fvta_lstm_plan(const fvta_lstm_desc* d, const int32_t* len, const int32_t* seq_J, const int64_t* x_off,
                   const int64_t* out_off, int64_t out_ld, void* plan, fvta_stream_t stream);

/* The same with a separate input for the backward direction: its rows are read (and its dx rows written) x_bw_delta
 * ELEMENTS behind the forward direction's, i.e. the caller passes x = [x_fw | x_bw] and dx = [dx_fw | dx_bw] with
 * x_bw_delta = the size of one copy (a multiple of 4).  This is how DropoutWrapper(cell, input_keep_prob) enters
 * (model_v2.py:657-661): bidirectional_dynamic_rnn calls the wrapped cell in two loops, so each direction sees its own
 * dropped copy of the inputs -- fvta_dropout_pair_fwd makes the two copies, fvta_dropout_pair_bwd folds [dx_fw | dx_bw]
 * back.  x_bw_delta = 0 is fvta_lstm_plan. */
int fvta_lstm_plan_xdir(const fvta_lstm_desc* d, const int32_t* len, const int32_t* seq_J, const int64_t* x_off,
                        const int64_t* out_off, int64_t out_ld, int64_t x_bw_delta, void* plan, fvta_stream_t stream);
/* x2[dir][e] = x[e] * keep(dir, e) / keep_prob, dir = 0 / 1, e < n (tf.nn.dropout's x / keep_prob * floor(keep_prob + u)
 * with u from a counter-based hash of (seed, dir, e) -- TensorFlow's random stream is not reproducible, its distribution
 * is; oracle/fvta_fused.py dropout_keep_masks evaluates the same hash).  Backward: dx[e] (+)= sum_dir dx2[dir][e] *
 * keep(dir, e) / keep_prob. */
int fvta_dropout_pair_fwd(const float* x, float* x2, int64_t n, float keep_prob, uint64_t seed, fvta_stream_t stream);
int fvta_dropout_pair_bwd(const float* dx2, float* dx, int64_t n, float keep_prob, uint64_t seed, int32_t accumulate,
                          fvta_stream_t stream);

/* kernel [in+d, 4d] gate order i,j,f,o; bias [4d]; forget_bias 1.0 added at
 * run time (BasicLSTMCell).  kernel_bw/bias_bw ignored when
 * share_fw_bw. */
int fvta_bilstm_fwd(const fvta_lstm_desc* d, const void* plan, const float* x, float* out,
                    const float* kernel_fw, const float* bias_fw, const float* kernel_bw,
                    const float* bias_bw, void* saved, void* workspace, fvta_stream_t stream);

/* d_out has out's layout.  dx (x's layout, may be NULL), dkernel and dbias are
 * all ACCUMULATED INTO (the caller zeroes them once per step); dx is written instead under desc.dx_overwrite. */
int fvta_bilstm_bwd(const fvta_lstm_desc* d, const void* plan, const float* x, const float* out,
                    const float* d_out, const float* kernel_fw, const float* kernel_bw, void* saved,
                    float* dx, float* dkernel_fw, float* dbias_fw, float* dkernel_bw, float* dbias_bw,
                    void* workspace, fvta_stream_t stream);

/* The same signature with a second stream.  `side_stream` is ACCEPTED AND UNUSED (kept for ABI stability): running dx and
 * the weight gradient beside the recurrence measured slower, so everything runs on `stream`; results are bitwise those
 * of fvta_bilstm_bwd.  How dx is summed (bf16 engines, no atomics anywhere): when the two directions share one input (no
 * input dropout) ONE launch adds both directions in the same accumulators and writes every dx element once; the k-tile
 * order of that sum is rotated per workgroup, so the result is deterministic for a given batch but its last bits depend
 * on where a row's tile sits in the grid (a row moved to another position of the batch may round differently).  With a
 * separate input per direction (fvta_lstm_plan_xdir) two launches write one dx copy each. */
int fvta_bilstm_bwd_overlap(const fvta_lstm_desc* d, const void* plan, const float* x, const float* out,
                            const float* d_out, const float* kernel_fw, const float* kernel_bw, void* saved,
                            float* dx, float* dkernel_fw, float* dbias_fw, float* dkernel_bw, float* dbias_bw,
                            void* workspace, fvta_stream_t stream, fvta_stream_t side_stream);

/* The same with what the HOST knows about the batch: nactive_host[t] (host memory, J entries, or NULL) = sequences with
 * len > t, read while the call enqueues its launches.  The plan is built on the device without a host sync, so the
 * library cannot size a step's launch by its active rows itself; with the hint the backward recurrence gives a step
 * with few active rows the small block tile (bf16 engine; results are bitwise those of fvta_bilstm_bwd_overlap: the
 * order of every sum is the same).  A wrong hint costs time, never correctness.  The library acts on it only under
 * FVTA_LSTM_BWD_HINT=1: measured on ragged batches the shorter text-cell launches lengthen the step, whose end is the
 * photo cell's chain of small launches on the side stream. */
int fvta_bilstm_bwd_hint(const fvta_lstm_desc* d, const void* plan, const float* x, const float* out,
                         const float* d_out, const float* kernel_fw, const float* kernel_bw, void* saved,
                         float* dx, float* dkernel_fw, float* dbias_fw, float* dkernel_bw, float* dbias_bw,
                         void* workspace, fvta_stream_t stream, fvta_stream_t side_stream, const int32_t* nactive_host);

/* Final states = concat(fw .h at t=len-1, bw .h at t=0) of sequences
 * [s0, s0+count): lq model_v2.py:697, lchoices 807-812.  dst [count, 2d]. */
/* The bf16 shadow rows of a call's output (bf16 engine): table [2][nrows] of device addresses -- table[dir][row] points at
 * the d bf16 values that are the output half-row `dir` of output row `row` (= element offset / out_ld) for every row this
 * plan writes below nrows (rows t < len); entries of other rows are left untouched: initialise the table with the address of
 * d zero bf16 values (the rows dynamic_rnn zeroes).  Valid until the next forward on `saved`.  model_v2.py:863-914's
 * context tensor is then never materialised in fp32: fvta_attn_fwd_shadow / fvta_attn_bwd_shadow read these rows. */
int fvta_lstm_shadow_rows(const fvta_lstm_desc* d, const void* plan, const void* saved, int64_t nrows, uint64_t* table,
                          fvta_stream_t stream);
/* out [nrows, out_ld] fp32 <- the rows behind such a table (inspection outputs: Tester.step_vis' hall, tests). */
int fvta_rows_from_shadow(const uint64_t* table, int64_t nrows, int32_t d, int64_t out_ld, float* out, fvta_stream_t stream);
int fvta_lstm_last_state(const fvta_lstm_desc* d, const void* plan, const float* out, int32_t s0,
                         int32_t count, float* dst, fvta_stream_t stream);
/* d_out[...] += d_dst at the same places. */
int fvta_lstm_last_state_bwd(const fvta_lstm_desc* d, const void* plan, const float* d_dst, int32_t s0,
                             int32_t count, float* d_out, fvta_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Answer scorer + loss: model_v2.py:1053-1083 (logits/yp) and 1085-1096
 * (softmax cross-entropy, mean over ALL N rows).
 *   gq [N,w], g1 [N,w], gch [N,C,w], W [5w] (7w with use_eu_output), b [1],
 *   y [N,C] bytes (NULL at test time) -> logits [N,C], yp [N,C], loss [1].
 * ------------------------------------------------------------------------- */
typedef struct fvta_scorer_desc {
  int32_t N, C, w;
  int32_t use_eu_output; /* model_v2.py:1071-1073 */
  int32_t add_tanh;      /* only with use_eu_output */
  int32_t xent_grad;     /* backward only.  0: d logits = softmax - labels on EVERY row, what TF-1's
                          * SoftmaxCrossEntropyWithLogits kernel returns (model_v2.py:1088) -- rows whose labels are all
                          * False (the padded rows of a short batch, model_v2.py:1270) still push softmax/N into the
                          * scorer; 1: the gradient of -sum(y log softmax) proper, softmax * sum(y) - y */
} fvta_scorer_desc;

int fvta_scorer_ce_fwd(const fvta_scorer_desc* d, const float* gq, const float* g1, const float* gch,
                       const float* W, const float* b, const uint8_t* y, float* logits, float* yp,
                       float* loss, fvta_stream_t stream);
/* d loss = loss_scale (a host scalar: 1 for a single GPU, 1/world under data
 * parallelism).  dgq/dg1 [N,w], dgch [N,C,w] overwritten; dW, db accumulated. */
int fvta_scorer_ce_bwd(const fvta_scorer_desc* d, const float* gq, const float* g1, const float* gch,
                       const float* W, const float* b, const uint8_t* y, const float* logits,
                       const float* yp, float loss_scale, float* dgq, float* dg1, float* dgch, float* dW,
                       float* db, fvta_stream_t stream);

/* ------------------------------------------------------------------------- *
 * AttentionGRUCell.__call__: attention_gru_cell.py:50-70 (one step).
 *   inputs [B,d+1] (last column = attention gate g), state [B,d],
 *   Wg [2d,d] + bg [d] (gates), Wc [d,d] (candidate, no bias), Wi [d,d] + bi [d]
 * ------------------------------------------------------------------------- */
int fvta_attgru_fwd(int32_t B, int32_t d, const float* inputs, const float* state, const float* Wg,
                    const float* bg, const float* Wc, const float* Wi, const float* bi, float* new_h,
                    float* saved /* [B,3d]: r, hWc, h_hat */, fvta_stream_t stream);
int fvta_attgru_bwd(int32_t B, int32_t d, const float* inputs, const float* state, const float* Wg,
                    const float* Wc, const float* Wi, const float* saved, const float* d_new_h,
                    float* d_inputs, float* d_state, float* dWg, float* dbg, float* dWc, float* dWi,
                    float* dbi, void* workspace /* B*4d floats */, fvta_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Time warp of the context tensor: model_v2.py:953-1009 with the indicator of
 * time_indication_func (301-341), in the closed form the reference's expression
 * reduces to:  warp_h[n,k,t,:] = hall[n,k,t,:] * c[n,t] * cnt(t).
 *   hall / warp_h / d_* [N,K,T,w]; lq [N,w] (WQ = lq, model_v2.py:970);
 *   WH_W [2w,w] (only its first w rows ever meet a non-zero feature), WH_b [w],
 *   WC_W [w], WC_b [1]: time_warp/{WH,WC}/{W,b}; window_t: time_warp_window_t
 *   (no gradient: tf.ceil, model_v2.py:335).  c_out / scale_out [N,T] are kept
 *   by the caller for the backward call.
 * ------------------------------------------------------------------------- */
typedef struct fvta_timewarp_desc {
  int32_t N, K, T, w;
  int32_t warp_type; /* 1 all, 2 current, 3 past, 4 future, 5 past-future window */
  float window_t;
} fvta_timewarp_desc;

size_t fvta_timewarp_workspace_bytes(const fvta_timewarp_desc* d);
int fvta_timewarp_fwd(const fvta_timewarp_desc* d, const float* hall, const float* lq, const float* WH_W,
                      const float* WH_b, const float* WC_W, const float* WC_b, float* warp_h, float* c_out,
                      float* scale_out, void* workspace, fvta_stream_t stream);
/* d_hall is overwritten; d_lq and the parameter gradients are accumulated into. */
int fvta_timewarp_bwd(const fvta_timewarp_desc* d, const float* hall, const float* lq, const float* WH_W,
                      const float* WH_b, const float* WC_W, const float* WC_b, const float* c_saved,
                      const float* d_warp, float* d_hall, float* d_lq, float* dWH_W, float* dWH_b, float* dWC_W,
                      float* dWC_b, void* workspace, fvta_stream_t stream);
/* The same warp over the bi-LSTM's bf16 SHADOW ROWS (fvta_lstm_shadow_rows; model_v2.py:953-1009 with the context tensor of
 * 863-914 never stored in fp32): row (n,k,t) is read as two bf16 half-rows through table [2][N*K*T] (device addresses),
 * warp_rows [N*K*T][w] receives the warped rows as bf16 -- the rows fvta_attn_fwd_shadow / fvta_attn_bwd_shadow then read
 * through a table of addresses into warp_rows.  w = 512 or 1024, K <= 8.  The backward takes the attention's fp32 gradient
 * of the warped rows (d_warp [N,K,T,w], zeros on masked rows) and overwrites d_hall; no time_warp_att term. */
int fvta_timewarp_fwd_shadow(const fvta_timewarp_desc* d, const uint64_t* table, const float* lq, const float* WH_W,
                             const float* WH_b, const float* WC_W, const float* WC_b, uint16_t* warp_rows, float* c_out,
                             float* scale_out, void* workspace, fvta_stream_t stream);
int fvta_timewarp_bwd_shadow(const fvta_timewarp_desc* d, const uint64_t* table, const float* lq, const float* WH_W,
                             const float* WH_b, const float* WC_W, const float* WC_b, const float* c_saved,
                             const float* d_warp, float* d_hall, float* d_lq, float* dWH_W, float* dWH_b, float* dWC_W,
                             float* dWC_b, void* workspace, fvta_stream_t stream);

/* The same with the attention's gradient w.r.t. the per-position scale (fvta_attn_bwd_tw's d_tscale, [N,T]; NULL =
 * fvta_timewarp_bwd): use_time_warp_att feeds c[n,t] cnt(t) into the attention as well (model_v2.py:1020). */
int fvta_timewarp_bwd_att(const fvta_timewarp_desc* d, const float* hall, const float* lq, const float* WH_W,
                          const float* WH_b, const float* WC_W, const float* WC_b, const float* c_saved,
                          const float* d_warp, const float* d_scale_att, float* d_hall, float* d_lq, float* dWH_W,
                          float* dWH_b, float* dWC_W, float* dWC_b, void* workspace, fvta_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Embedding front-end (model_v2.py:524-645): what turns the
 * reference's token-id feed into the encoder inputs.
 *
 * fvta_embed_fwd replaces, for ALL text inputs of a batch at once (at, ad, when,
 * where, pts, q, choices: model_v2.py:531-620),
 *   tf.nn.embedding_lookup(char_emb, *_c) -> conv1d (52-70: conv2d VALID, +bias,
 *   relu, reduce_max over the width) -> concat([char part, word part]) with the
 *   word table concat([word_emb_mat (trainable, VW rows), existing_emb_mat
 *   (frozen GloVe, VT - VW rows)]) (590).
 * Token t: word_ids[t] in [0, VT), char_ids[t*W .. +W) in [0, VC); its row of
 * cwdim + wdim floats is written at x + tok_off[t] (element offset: the caller
 * lays tokens out as the encoder input arena wants them).  cwdim = 0: use_char
 * off (char_ids / char_emb / filt / bias ignored).  keep_prob is 1.
 *   filt [height, cdim, cwdim] (= TF's [1, height, cdim, cwdim]), bias [cwdim].
 *   argpos [ntok, cwdim] u8 (saved for backward): arg-max window, 255 where the
 *   relu is inactive.  Limits: cwdim <= 128, W <= 64, height*cdim <= 64, VC <= 1024.
 * fvta_embed_bwd: dx rows (same offsets) -> gradients ACCUMULATED into
 *   d_word_emb [VW, wdim] (rows >= VW are frozen; float atomics: the one
 *   order-dependent sum of the library, as TF's own IndexedSlices sum),
 *   d_char_emb [VC, cdim], d_filt, d_bias (fixed-order reductions).
 *   max ties go to the first window (TF splits them), as in the attention.
 * ------------------------------------------------------------------------- */
typedef struct fvta_embed_desc {
  int32_t ntok;
  int32_t W;      /* chars per word (config.max_word_size) */
  int32_t cdim;   /* char_emb_size */
  int32_t cwdim;  /* char_out_size; 0 = no char-CNN */
  int32_t wdim;   /* word_emb_size */
  int32_t VW;     /* trainable word rows */
  int32_t VT;     /* all word rows (trainable + frozen) */
  int32_t VC;     /* char vocabulary */
  int32_t height; /* conv window (5) */
  float keep_prob;       /* conv1d's dropout of the gathered char embeddings while training (model_v2.py:58-62); 0 or 1: off */
  uint64_t dropout_seed; /* element (tok, pos, c) is kept by the counter-based hash of (seed, (tok * W + pos) * cdim + c) */
} fvta_embed_desc;

size_t fvta_embed_workspace_bytes(const fvta_embed_desc* d);
int fvta_embed_fwd(const fvta_embed_desc* d, const int32_t* word_ids, const int32_t* char_ids,
                   const int64_t* tok_off, const float* word_emb, const float* fixed_emb, const float* char_emb,
                   const float* filt, const float* bias, float* x, uint8_t* argpos, fvta_stream_t stream);
int fvta_embed_bwd(const fvta_embed_desc* d, const int32_t* word_ids, const int32_t* char_ids,
                   const int64_t* tok_off, const float* char_emb, const float* filt, const uint8_t* argpos,
                   const float* dx, float* d_word_emb, float* d_char_emb, float* d_filt, float* d_bias,
                   void* workspace, fvta_stream_t stream);

/* Photo features (model_v2.py:634-645): row m of the output, at x + row_off[m]
 * (element offset), = embedding_lookup(image_emb_mat [*, idim], pidx[m]), passed
 * through image_trans_linear W [idim, tdim], b [tdim] (+tanh if add_tanh) when W
 * is not NULL (use_image_trans); with W NULL it is the gathered row itself
 * (tdim = idim).  image_emb_mat is a placeholder in the reference: no gradient.
 * Backward (use_image_trans only): dW, db accumulated; workspace M*tdim floats. */
typedef struct fvta_imgtrans_desc {
  int32_t M, idim, tdim, add_tanh;
} fvta_imgtrans_desc;

int fvta_image_trans_fwd(const fvta_imgtrans_desc* d, const int32_t* pidx, const int64_t* row_off,
                         const float* image_emb_mat, const float* W, const float* b, float* x,
                         fvta_stream_t stream);
int fvta_image_trans_bwd(const fvta_imgtrans_desc* d, const int32_t* pidx, const int64_t* row_off,
                         const float* image_emb_mat, const float* x, const float* dx, float* dW, float* db,
                         void* workspace, fvta_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Parameter update over the flat fp32 parameter buffer: trainer.py:16
 * AdadeltaOptimizer(init_lr) (rho 0.95, eps 1e-8) and the commented-out
 * AdamOptimizer of trainer.py:17.  grad_scale multiplies the gradient first
 * (1/world after the RCCL sum).
 * ------------------------------------------------------------------------- */
int fvta_adadelta_step(float* var, const float* grad, float* accum, float* accum_update, int64_t n,
                       float lr, float rho, float eps, float grad_scale, fvta_stream_t stream);
int fvta_adam_step(float* var, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, int32_t t, float grad_scale, fvta_stream_t stream);
/* add_wd (model_v2.py:347-354; --wd, main.py:105): one l2 term of the "losses" collection for one variable:
 * loss[0] += coef/2 * sum(var^2) (if loss != NULL) and grad += coef * var (if grad != NULL); coef = wd x the number
 * of add_wd calls that cover the variable (the shared char-CNN filter is covered once per conv1d call,
 * model_v2.py:564-571).  Fixed summation order. */
int fvta_weight_decay(const float* var, float* grad, int64_t n, float coef, float* loss, fvta_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Stand-alone forms of the reference's small graph helpers (SURVEY 8b "functional ops"), forward only.  Inside the
 * model they are folded into the attention / scorer / embedding kernels.
 *   fvta_softmax_fwd : softmax(logits) over the last axis, model_v2.py:23-28 (logits viewed as [rows, J])
 *   fvta_softsel_fwd : softsel(target [rows,J,d], logits [rows,J]) -> [rows,d], model_v2.py:39-48
 *   fvta_exp_mask    : val + (1 - mask) * -1e30, utils.py:210-213
 *   fvta_linear_fwd  : flatten(x,1) * W[in,out] + b (+ tanh), model_v2.py:75-100 (b may be NULL)
 * ------------------------------------------------------------------------- */
int fvta_softmax_fwd(const float* logits, float* out, int64_t rows, int32_t J, fvta_stream_t stream);
int fvta_softsel_fwd(const float* target, const float* logits, float* out, int64_t rows, int32_t J, int32_t d,
                     fvta_stream_t stream);
int fvta_exp_mask(const float* val, const uint8_t* mask, float* out, int64_t n, fvta_stream_t stream);
int fvta_linear_fwd(const float* x, const float* W, const float* b, float* y, int64_t M, int32_t in, int32_t out,
                    int32_t add_tanh, fvta_stream_t stream);
/* Backward of fvta_linear_fwd (the `bidrection_squash` / concat linears of model.py:895-991): with dyt = dy, or
 * dy (1 - y^2) under add_tanh (y = the forward output): dx [M,in] = dyt W^T (overwritten, or added to when accumulate_dx),
 * dW [in,out] += x^T dyt, db [out] += sum_m dyt.  dx, dW (with db) may each be NULL. */
int fvta_linear_bwd(const float* x, const float* W, const float* y, const float* dy, float* dx, float* dW, float* db,
                    int64_t M, int32_t in, int32_t out, int32_t add_tanh, int32_t accumulate_dx, fvta_stream_t stream);
/* The same two with x (and dx) rows in blocks: row m starts at (m / rows_per_blk) * blk_stride + (m % rows_per_blk) * in
 * floats -- one context stream's rows inside the model.py graph's [N][all streams] arena (attention_tgif's mlp_h over a
 * stream, model.py:226).  y / dy stay dense [M,out]. */
int fvta_linear_fwd_blk(const float* x, const float* W, const float* b, float* y, int64_t M, int32_t in, int32_t out,
                        int32_t add_tanh, int64_t rows_per_blk, int64_t blk_stride, fvta_stream_t stream);
int fvta_linear_bwd_blk(const float* x, const float* W, const float* y, const float* dy, float* dx, float* dW, float* db,
                        int64_t M, int32_t in, int32_t out, int32_t add_tanh, int32_t accumulate_dx, int64_t rows_per_blk,
                        int64_t blk_stride, fvta_stream_t stream);
/* Backward of fvta_softmax_fwd: dx = p (dp - sum_j p dp), p = the forward output. */
int fvta_softmax_bwd(const float* p, const float* dp, float* dx, int64_t rows, int32_t J, fvta_stream_t stream);
/* sum_j weights[r,j] * target[r,j,:] -> out[r,:] (no softmax): the attended vector of attention_tgif, model.py:236-238 */
int fvta_wsum_fwd(const float* target, const float* weights, float* out, int64_t rows, int32_t J, int32_t d,
                  fvta_stream_t stream);
/* ... with the rows' [J,d] blocks target_ld floats apart, and its backward: d_weights [rows,J] = target . d_out
 * (overwritten; may be NULL), d_target += weights (x) d_out (same layout as target; may be NULL). */
int fvta_wsum_fwd_ld(const float* target, const float* weights, float* out, int64_t rows, int32_t J, int32_t d,
                     int64_t target_ld, fvta_stream_t stream);
int fvta_wsum_bwd(const float* target, const float* weights, const float* d_out, float* d_weights, float* d_target,
                  int64_t rows, int32_t J, int32_t d, int64_t target_ld, fvta_stream_t stream);
/* The feature vector of the DMN+ episode attention, model_dmnplus.py:93-98 (`_get_attention`), for all facts at once:
 * out[n,f,:] = [fact*q, fact*m, |fact-q|, |fact-m|]  (facts [N,F,d], q / m [N,d] -> out [N,F,4d]).  The two
 * fully_connected layers on top of it are fvta_linear_fwd, the softmax over facts fvta_softmax_fwd, the gated recurrence
 * fvta_attgru_fwd per fact (functional.generate_episode). */
int fvta_dmn_features(const float* facts, const float* q, const float* m, float* out, int32_t N, int32_t F, int32_t d,
                      fvta_stream_t stream);
/* Its backward (d_out [N,F,4d]): d_facts [N,F,d], d_q and d_m [N,d] are ACCUMULATED into (the facts and the question feed
 * every hop, model_dmnplus.py:507-509); |x| has derivative 0 at 0, as tf.abs. */
int fvta_dmn_features_bwd(const float* facts, const float* q, const float* m, const float* d_out, float* d_facts, float* d_q,
                          float* d_m, int32_t N, int32_t F, int32_t d, fvta_stream_t stream);
/* relu and its backward from the OUTPUT y (the memory update tf.layers.dense(..., activation=tf.nn.relu),
 * model_dmnplus.py:511-514; the dense itself is fvta_linear_fwd / _bwd). */
int fvta_relu_fwd(const float* x, float* y, int64_t n, fvta_stream_t stream);
int fvta_relu_bwd(const float* y, const float* dy, float* dx, int64_t n, fvta_stream_t stream);
/* The two shape ops the model.py graph (soft-attention baselines) puts between its attentions, forward and backward of
 * each other:
 *   fvta_rows_reduce    : out[r,:] (+)= scale * sum_j x[r,j,:]   x [rows,J,d], out rows `out_ld` floats apart.
 *                         tf.reduce_mean (model.py:874-885 means of the last states, :907 mean over the K streams) with
 *                         scale 1/J; with scale 1 the backward of the tile below.
 *   fvta_rows_broadcast : out[r,j,:] (+)= scale * v[r,:]         v rows `v_ld` floats apart, out [rows,J,d].
 *                         tf.tile (model.py:262 hq per choice) with scale 1; with scale 1/J the backward of reduce_mean.
 * accumulate != 0 adds into the destination. */
int fvta_rows_reduce(const float* x, float* out, int64_t rows, int32_t J, int32_t d, int64_t out_ld, float scale,
                     int32_t accumulate, fvta_stream_t stream);
int fvta_rows_broadcast(const float* v, float* out, int64_t rows, int32_t J, int32_t d, int64_t v_ld, float scale,
                        int32_t accumulate, fvta_stream_t stream);
/* The reversed direction of attention(..., bidirect=True) / attention_keeprank1(..., bidirect=True) (model_v2.py:184-192,
 * model.py:169-177, 297-307): q_a[r,:] = mean_v softsel(hq[r], a_logits[r,v,:]) over the MASKED logits fvta_attn_fwd
 * returns (a_logits [R,V,JQ], hq [R,JQ,w]; JQ <= 64, V*JQ <= 8192: the reference only runs this branch on short row
 * lists).  Backward: d_q_a [R,w] -> dA [R,V,JQ] (overwritten; the additive mask passes it to the raw logits unchanged)
 * and d_hq (accumulated).  fvta_attn_logits_bwd turns a DENSE logit gradient dA [N,T,JQ] of a K = 1 attention (no tanh,
 * simiMatrix 1-3; hinfo_stride honoured) into d_hinfo, d_hq (accumulated), dW, db (accumulated) -- the max-pooled h_a
 * path stays with fvta_attn_bwd. */
int fvta_attn_qside_fwd(const float* a_logits, const float* hq, float* q_a, int32_t R, int32_t V, int32_t JQ, int32_t w,
                        fvta_stream_t stream);
int fvta_attn_qside_bwd(const float* a_logits, const float* hq, const float* d_q_a, float* dA, float* d_hq, int32_t R,
                        int32_t V, int32_t JQ, int32_t w, fvta_stream_t stream);
size_t fvta_attn_logits_bwd_workspace_bytes(const fvta_attn_desc* d);
int fvta_attn_logits_bwd(const fvta_attn_desc* d, const float* hinfo, const float* hq, const float* W, const float* dA,
                         float* d_hinfo, float* d_hq, float* dW, float* db, void* workspace, fvta_stream_t stream);
/* attention_keeprank1 (model.py:247-314) = the per-(n,k) inner softsel of attention_3d without the softmax over k:
 * after fvta_attn_fwd(desc with K = M) this copies that result, u[N,K,w], out of the saved state. */
int fvta_attn_read_u(const fvta_attn_desc* d, const void* saved, float* u_out, fvta_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Test hooks (not part of the reference surface): the MFMA tile engines the
 * LSTM kernels are built from, exposed as plain GEMMs so tests can check the
 * fragment layouts in isolation.  layout 0: C=A[M,K]*B[K,N];
 * 1: C=A[M,K]*B[N,K]^T; 2: C=A[K,M]^T*B[K,N].
 * ------------------------------------------------------------------------- */
int fvta_test_gemm(int32_t precision, int32_t layout, int32_t M, int32_t N, int32_t K, const float* A,
                   const float* B, float* C, fvta_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Opt-in measurement hook (bench.py): while enabled on the calling thread, the
 * kernels below are bracketed by hipEvent pairs recorded on the launch stream.
 * fvta_profile_collect synchronises on them and returns the summed elapsed
 * milliseconds and the number of kernel launches they covered.  The LSTM ids
 * bracket the whole per-step launch sequence of one call (J launches).
 * ------------------------------------------------------------------------- */
#define FVTA_PROF_LSTM_STEP_FWD 1 /* lstm_step_fwd_*  : J launches per fvta_bilstm_fwd */
#define FVTA_PROF_LSTM_STEP_BWD 2 /* the recurrence's step launches of one fvta_bilstm_bwd (fp32 engine: gate + step GEMM, with dx) */
#define FVTA_PROF_LSTM_DW 3       /* weight-gradient GEMM + slab reduce */
#define FVTA_PROF_ATTN_FWD_MAIN 4 /* attn_fwd_main */
#define FVTA_PROF_ATTN_BWD_MAIN 5 /* attn_bwd_main */
#define FVTA_PROF_LSTM_DX 6       /* lstm_dx_bf16: the input gradient of all steps, one launch (bf16 engine) */
int fvta_profile_enable(int32_t on);
int fvta_profile_collect(int32_t id, double* total_ms, int64_t* launches);
/* Test / measurement hook: which bi-LSTM step kernels the bf16 engine may use -- bit 0 the weights-stationary forward
 * (lstm_fwd_wreg_bf16), bit 1 the weights-stationary backward of steps with few rows (lstm_bwd_wreg_bf16), bit 2 the
 * pipelined weights-stationary backward (lstm_bwd_ring_bf16, d = 512); 0 = the tiled kernels for every shape; a negative
 * mask returns to the default (the FVTA_LSTM_WREG environment variable, else every bit set).  Returns the previous mask.
 * Every choice computes the same step (model_v2.py:652-661, 694-823); results differ in the last bits (summation order).
 * Process-wide, not thread-safe.  Not part of the reference surface. */
int fvta_lstm_kernel_select(int32_t mask);
/* Measurement hook: how many backward-step launches of calls with more than 64 sequences (the text cell) ran on which
 * kernel since the last call -- counts[0] the tiled step (lstm_bwd_fused_bf16), counts[1] the pipelined weights-stationary
 * step (lstm_bwd_ring_bf16), counts[2] the round-3 weights-stationary step (lstm_bwd_wreg_bf16); reading resets them.
 * bench.py labels its backward-step roofline by these instead of re-deriving the library's choice.  Process-wide. */
int fvta_lstm_bwd_kernel_counts(int64_t* counts);
/* Test / measurement hook: which focal-attention forward main kernel runs (every choice computes model_v2.py:210-298; the
 * fast kernels' logits carry the 3-term fp16 split, <= 3 2^-22 |h||q| per product).  exact: 1 the exact-fp32 kernel
 * (attn_fwd_main) for every shape, 0 the fast kernels where they cover the shape, negative: the default (environment
 * FVTA_ATTN_EXACT, read once, else 0).  wave16: 0 attn_fwd_rows16, 1 attn_fwd_wave16, 2 / 3 attn_fwd_pair16 with barriers /
 * with its flag hand-shake, negative: the default (FVTA_ATTN_WAVE16, read once, else 3).  Process-wide, not thread-safe. */
int fvta_attn_kernel_select(int32_t exact, int32_t wave16);
/* Measurement hook (bench.py, SURVEY 8d "achievable peak"): one read-only, fully coalesced, non-temporal pass over
 * `bytes` of device memory; the caller times it.  Not part of the reference surface. */
int fvta_probe_hbm_read(const void* buf, size_t bytes, float* sink, fvta_stream_t stream);
/* Measurement hook: `nread` streams read and `nwrite` streams written, bytes_per_stream each, all inside `buf`
 * (which must hold (nread + nwrite) * bytes_per_stream bytes), 16 B per lane, coalesced, non-temporal; the caller times
 * it.  The achievable rate of a mixed read/write stream set -- what the LSTM step epilogues are.  Not part of the
 * reference surface. */
int fvta_probe_hbm_mix(void* buf, size_t bytes_per_stream, int32_t nread, int32_t nwrite, fvta_stream_t stream);
/* Measurement hook (Model side-stream selection): one wave that occupies `stream` for `microseconds` of the 100 MHz
 * wall clock.  Two of them on two streams take one wait if the streams run concurrently (separate hardware queues)
 * and two if HIP mapped both streams onto one queue.  Not part of the reference surface. */
int fvta_probe_spin(int64_t microseconds, fvta_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FVTA_HIP_H */
