/*
 * coattn.h -- C-ABI of the MI355X (gfx950) Hierarchical Parallel Co-Attention path.
 *
 * Drop-in boundary.  The reference exposes no FFI: its boundary is the
 * Python nn.Module surface
 *     ParallelCoAttention.forward(x_img[B,N,d], 3 x x_ques[B,T,d]) -> (3 x v[B,d], 3 x q[B,d])
 *     (/root/reference/model.py:356-397, constructed model.py:167, called model.py:182)
 * with backward = autograd of model.py:372-392 (triggered at main.py:219-220).  This library
 * is what a ctypes binding behind that surface calls: coattn_forward replaces the forward
 * loop (model.py:372-395), coattn_backward replaces its autograd graph.  See INTEGRATION.md
 * for the reference-side stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (torch tensor.data_ptr()) unless marked "host";
 *   - the caller owns every buffer, including `saved` and `ws` (sizes: coattn_workspace_bytes);
 *   - calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream), re-entrant and thread-agnostic (backward runs on the autograd thread);
 *   - return 0 on success, < 0 on error (never throws); coattn_last_error() returns the
 *     calling thread's last message;
 *   - math is fp32 (dtype = COATTN_F32: fp32 storage and accumulation; products fp32-accurate by default, see
 *     "Widths of the fp32 mode" below), row-vector convention Linear(x) = x W^T + b.
 *
 * Layouts
 *   V      : the image features x_img[B,N,d] (model.py:215-217), given as a base pointer plus the element
 *            strides (v_sB, v_sN, v_sD) of that logical view -- no copy is ever made.  Two physical layouts
 *            run on the fused kernels:
 *              channel-major  [B][d][N]: (v_sB >= d*N, v_sN = 1, v_sD = N) -- the buffer behind the permuted
 *                             view the reference's NCHW encoder returns;
 *              location-major [B][N][d]: (v_sB >= N*d, v_sN = d, v_sD = 1) -- what a channels_last encoder
 *                             emits (x_img is then contiguous);
 *            any other strides take the general-shape kernels.  dV is described the same way.
 *   Q[l]   : [B, T, d]  contiguous, l = 0..L-1 (word, phrase, sentence; model.py:298).
 *   W_v,W_q: [d, d] (nn.Linear.weight, out x in); b_v,b_q: [d]; w_v,w_q: [d]; c_v,c_q: [1]
 *            (model.py:350-354).  W_b (model.py:347) is dead in the reference and not passed.
 *   v_out,q_out,gv,gq : [L, B, d].
 */
#ifndef COATTN_H
#define COATTN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COATTN_F32 0
#define COATTN_BF16 1   /* storage type of an INPUT where an entry point says so (coattn_features_native); the math stays fp32 */

/* impl selector (flags & 3): 0 = auto (fused kernels when the shape allows, else general),
 * 1 = general-shape kernels (MFMA GEMM composition), 2 = fused kernels (error if unsupported). */
#define COATTN_IMPL_AUTO 0
#define COATTN_IMPL_GENERAL 1
#define COATTN_IMPL_FUSED 2
/* flags bit 2: the reduced-precision mode that the reference reaches through apex AMP O1 (main.py:185).  The projections
 * P_v, P_q (model.py:380-384) and, in the backward, the d x d contractions that are their gradients (dQ += dP_q W_q,
 * dV += dP_v W_v, dW_v, dW_q) run on the bf16 MFMA (v_mfma_f32_32x32x16_bf16) with operands rounded to bf16 while staged,
 * ONE MFMA per product, fp32 accumulation and fp32 results (gemm_bf.hip at wide shapes, else the single-piece modes of
 * gemm_w.hip / gemm_tn.hip).  On the fused path with d % 512 == 0 the affinity / attention contractions of the fused
 * kernels do the same (single-product instantiations; other widths and the general-shape path keep those exact), and the
 * backward's workspace holds dP_v / dP_q as bf16 when only GEMMs consume them.  Inputs, `saved` and outputs are fp32 in
 * either mode.  Parity then holds to bf16 tolerance (~1e-2), not 1e-4.
 * The same bit selects the bf16 MFMA for the three contractions of coattn_phrase_forward/backward. */
#define COATTN_FLAG_BF16_PROJ 4
#define COATTN_FLAG_BF16_IN 8     /* coattn_linear_forward / coattn_linear_weight_grad, with COATTN_FLAG_BF16_PROJ: x (dy) is STORED as bf16 */
/* Widths of the fp32 mode (dtype COATTN_F32).  An fp32 product runs on the 16-bit MFMAs as partial products of 16-bit
 * PIECES of its operands.
 *   flags = 0 -- the DEFAULT, "exact": three bf16 pieces per operand (hi + mid + lo = the value exactly, six partial
 *     products down to relative order 2^-16, each exact in the fp32 accumulator): every contraction of coattn_forward /
 *     coattn_backward / coattn_phrase_* is fp32-accurate (one fp32 rounding per product) over fp32's whole range, as the
 *     reference's fp32 bmm / Linear (model.py:377-392).  inf / NaN inputs give inf / NaN outputs.
 *   COATTN_FLAG_FAST16 (flags bit 7) -- the "tolerance" mode a caller opts into (train.Trainer does): fewer partial
 *     products, inside the north star's 1e-4 contract on operands of ordinary magnitude:
 *       - forward-side contractions on two FP16 pieces (hi + lo: 22 significand bits, three partial products on
 *         v_mfma_f32_32x32x16_f16, ~2^-22 relative): the affinity A = Q V^T (model.py:377), the projections (model.py:380-384;
 *         the weight image holds 256 W, divided out), C^T P_q and C P_v.  RANGE: exact pieces for |x| <= 65,504 (values below
 *         2^-14 keep 2^-24 absolute; projection weights |W| <= 255); the conversions SATURATE (MODE.FP16_OVFL), so magnitudes
 *         up to 131,008 are still carried with fewer bits and larger ones -- +-inf included -- clamp there.  That event is not
 *         silent: the projection launch records the largest magnitude it converted, and coattn_status() reports it
 *         (-4 = an operand left the exact-piece range in the last coattn_forward on this `saved`);
 *       - gradient contractions of coattn_backward on two bf16 pieces (hi + mid: 16 significand bits, three partial products,
 *         ~2^-16 relative per product, random in sign, fp32's range) -- their operands are gradients of any magnitude.
 *     On the golden cases: v, q within 1e-6, attention maps within 4e-7, H_q within 2e-5, gradients within 1.5e-5 of max|.|
 *     (tests assert 5e-5; the contract is 1e-4); tests/test_split_emulation.py has the budget row by row.
 *     Shapes the hand-scheduled projection kernels do not take (the general-shape path; B N < 128 rows) stay exact.
 *   COATTN_FLAG_EXACT3 (flags bit 4) -- spells the default out (it wins over COATTN_FLAG_FAST16); kept from v0.5.x, where
 *     the tolerance mode was the default of flags = 0. */
#define COATTN_FLAG_EXACT3 16
/* flags bit 5 (coattn_linear_forward; `accumulate` of coattn_linear_weight_grad): the two-bf16-piece width for this product. */
#define COATTN_FLAG_SPLIT2 32
/* flags bit 6 (coattn_linear_forward): two FP16 pieces for this product (the form COATTN_FLAG_FAST16 runs the projections
 * in; no range report: that is coattn_forward's); the weight image written under this flag is read under this flag only. */
#define COATTN_FLAG_F16PAIR 64
#define COATTN_FLAG_FAST16 128
typedef struct coattn_params {
  const void* W_v; const void* b_v;   /* model.py:350 */
  const void* W_q; const void* b_q;   /* model.py:351 */
  const void* w_v; const void* c_v;   /* model.py:353 */
  const void* w_q; const void* c_q;   /* model.py:354 */
} coattn_params;

typedef struct coattn_param_grads {
  void* dW_v; void* db_v; void* dW_q; void* db_q;
  void* dw_v; void* dc_v; void* dw_q; void* dc_q;
} coattn_param_grads;

/* library version: major*10000 + minor*100 + patch */
int coattn_version(void);

/* last error message of the calling thread ("" if none) */
const char* coattn_last_error(void);

/* Per-kernel timing of the calls this THREAD makes between the two (bench.py's per-kernel roofline legs; nothing a
 * training loop calls).  coattn_profile_begin records a HIP event on `stream`; every launch group of coattn_forward /
 * coattn_backward issued afterwards on this thread records one more behind it.  coattn_profile_end synchronises on the
 * last event and returns the number of marks n (<= max_marks), with us[i] = microseconds between mark i-1 and mark i and
 * `names` = the n mark names joined by '\n' (truncated to names_bytes).  Not for use under graph capture. */
int coattn_profile_begin(void* stream);
int coattn_profile_end(float* us, char* names, int names_bytes, int max_marks);

/* Image features as the encoder leaves them -> the layout the kernels run on (the boundary on the image side: what
 * HierarchicalCoAttentionNet does between image_encoder and co_attention, model.py:215-217 view + permute, under AMP
 * main.py:73, :185 with bf16 activations).
 *   x   : element (b, n, c) at x[b sB + n sN + c sD], x_dtype COATTN_F32 or COATTN_BF16, any non-negative strides -- e.g.
 *         the permuted NCHW view (sB >= d N, sN = 1, sD = N) at N = 49, whose 196-byte (98-byte) rows the kernels do not
 *         take in place;
 *   out : fp32 [B, N, d] contiguous (location-major), the V of coattn_forward / coattn_backward with strides (N d, d, 1).
 * One pass at the memory rate (read along n, written along c through an LDS tile).  Features that are already fp32 and
 * location-major, or channel-major with N % 4 == 0, need no call: coattn_forward takes them where they lie. */
int coattn_features_native(const void* x, int x_dtype, int64_t sB, int64_t sN, int64_t sD, void* out, int B, int N, int d,
                           void* stream);

/* 1 if a fused-kernel configuration exists for this shape (for channel-major or location-major V), else 0 */
int coattn_fused_supported(int B, int N, int T, int d, int L, int dtype);

/* Buffer sizes in bytes.  saved: forward -> backward state (P_v, P_q, C, a_v, a_q, H_q, W_q split into bf16
 * pieces for the backward's dQ projection, the status words of the tolerance mode, and the bitmap of the question rows that
 * are not all zeros: one bit per row and level, written by the exact forward, read by the backward's dW_q);
 * ws_fwd / ws_bwd: scratch, contents undefined after the call. */
int coattn_workspace_bytes(int B, int N, int T, int d, int L, int dtype, int flags,
                           size_t* saved, size_t* ws_fwd, size_t* ws_bwd);

/* Forward of model.py:372-395 for all L levels.
 *   Q      : host array of L device pointers.
 *   saved  : NULL for inference (nothing kept), else a buffer of `saved` bytes.
 *   v_out,q_out : [L,B,d].
 * Rows of Q_l that are all zeros -- the pad tokens of the reference's question hierarchy (model.py:263 padding_idx,
 * :292-296 pad_packed_sequence) -- project to the bias alone, and the exact mode (flags = 0) uses that: the launch that
 * splits the weights also reads every question row once, flags the rows that hold anything and writes (0 + b_q) into the
 * other rows of P_q; the projection GEMM then runs over the flagged rows only.  Data-driven (no length argument: the
 * reference's forward(x_img, x_ques_hierarchy) has none), and bit-identical to the dense product for every input: zero rows
 * anywhere, none at all, -0.0, NaN (tests/test_gpu_edges.py::test_zero_question_rows_take_the_bias_path_bit_for_bit).
 * coattn_backward uses the same bitmap (kept in `saved`): dW_q = sum dP_q^T Q contracts over the flagged rows only, and the
 * launch of the two weight gradients shares its split-K parts between dW_v and dW_q on the device (the count is not known on
 * the host).  Deterministic -- a function of the inputs --; the order of the additions differs from the all-rows plan, so dW_v /
 * dW_q agree with it to fp32 rounding, not bit for bit.  (One corner differs in kind: a non-finite dP_q element in a zero row
 * makes the reference's dW_q NaN through inf * 0; here that row is not contracted.  db_q, which sums every row, is non-finite in
 * both, so the step still shows.) */
int coattn_forward(const void* V, int64_t v_sB, int64_t v_sN, int64_t v_sD, const void* const* Q,
                   const coattn_params* p, void* v_out, void* q_out, void* saved, void* ws,
                   int B, int N, int T, int d, int L, int dtype, int flags, void* stream);

/* Second half of coattn_forward only: everything after the projections P_v, P_q (affinity +
 * tanh model.py:377, H_v/H_q :380-384, scores + row softmax :387-388, attended reductions
 * :391-392), reading P_v / P_q from a `saved` buffer that a previous coattn_forward on the same
 * inputs filled -- with the SAME flags (the fused and the general-shape kernels keep P_v / P_q in different scalings).
 * Exists so that tests and bench.py can time / check this kernel in isolation. */
int coattn_attention_forward(const void* V, int64_t v_sB, int64_t v_sN, int64_t v_sD, const void* const* Q,
                             const coattn_params* p, void* v_out, void* q_out, void* saved, void* ws,
                             int B, int N, int T, int d, int L, int dtype, int flags, void* stream);

/* Backward (autograd of model.py:372-392).
 *   gv,gq : [L,B,d] upstream gradients of v_out,q_out.
 *   dV    : gradient of x_img with its own strides (dv_sB, dv_sN, dv_sD) (overwritten), or NULL when the image
 *           features need no gradient (frozen encoder, model.py:239-241);
 *   dQ    : host array of L device pointers [B,T,d] (overwritten).
 *   pg    : parameter gradients; accumulate = 0 overwrites, 1 adds into them (grads of the
 *           three levels are always summed: one weight set is shared, model.py:167, :372).
 *   saved : of a coattn_forward call with the SAME inputs, parameter values and flags (it holds projections of
 *           them and an image of W_q; autograd's forward -> backward order guarantees this). */
int coattn_backward(const void* V, int64_t v_sB, int64_t v_sN, int64_t v_sD, const void* const* Q,
                    const coattn_params* p, const void* saved, const void* gv, const void* gq,
                    void* dV, int64_t dv_sB, int64_t dv_sN, int64_t dv_sD, void* const* dQ,
                    const coattn_param_grads* pg, int accumulate,
                    void* ws, int B, int N, int T, int d, int L, int dtype, int flags, void* stream);

/* Range report of the tolerance mode (COATTN_FLAG_FAST16).  SYNCHRONISES `stream`, reads the status words the last
 * coattn_forward left in `saved` (or in `ws`, when that call was given saved = NULL) and returns
 *    0  every operand converted to FP16 pieces lay inside the exact-piece range (or the call did not use FP16 pieces:
 *       exact mode, reduced-precision mode, general-shape path);
 *   -4  some operand did not: an activation (image / question feature, stored projection) beyond 65,504 or a projection
 *       weight beyond 255.87 in magnitude, +-inf included -- its pieces were clamped (see COATTN_FLAG_FAST16) and the
 *       results of that call are not within tolerance of the reference; coattn_last_error() names the operand class and
 *       the magnitude.  Re-run with flags = 0 (exact), which computes the reference's value over fp32's whole range.
 * amax (host, may be NULL): [0] = largest |activation| beyond the range seen by the projection launch (0 if none),
 * [1] = largest |256 W| over both projection weights.  Call it where the host synchronises anyway (reading the loss). */
int coattn_status(const void* saved, int B, int N, int T, int d, int L, int dtype, void* stream, float* amax);
/* The same report made STICKY across calls without a synchronisation (v0.6.1): every coattn_forward rewrites the status words
 * of its `saved`, so a caller that reads them only now and then (a training loop that reads the loss every log_interval
 * steps) would miss the steps in between.  ASYNCHRONOUS on `stream`: folds the status words the last coattn_forward left in
 * `saved` into `acc`, two floats in DEVICE memory the caller owns and zeroes: acc[0] = max(acc[0], largest out-of-range
 * |activation| of that call), acc[1] = max(acc[1], largest |256 W|) -- maxima on the bit patterns, so a NaN wins.  A call
 * that used no FP16 pieces leaves acc alone.  The caller reads acc where it synchronises anyway: the tolerance mode held
 * for every folded call iff acc[0] <= 65504 and acc[1] <= 65504 (both 0 when nothing was out of range / no weights seen).
 * Steps run before the caller looks are NOT rolled back: a training loop that cannot afford that runs flags = 0. */
int coattn_status_accumulate(const void* saved, int B, int N, int T, int d, int L, int dtype, void* acc, void* stream);

/* ---- PhraseConvPool: the question hierarchy's phrase level ----------------
 * Replaces reference model.py:301-334 (`PhraseConvPool.forward`: 1/2/3-gram Conv1d + Tanh with
 * ConstantPad1d (0,0) / (1,0) / (1,1), concatenated along channels, then MaxPool2d((1,3)) over
 * groups of 3 CONSECUTIVE channels) and its autograd.  Weights in torch Conv1d layout
 * [E_out][E_in][k] (state_dict keys conv_unigram.1.*, conv_bigram.1.*, conv_trigram.1.*). */
typedef struct coattn_phrase_params {
  const void* W1; const void* b1;   /* [E,E,1], [E] */
  const void* W2; const void* b2;   /* [E,E,2], [E] */
  const void* W3; const void* b3;   /* [E,E,3], [E] */
} coattn_phrase_params;

typedef struct coattn_phrase_param_grads {
  void* dW1; void* db1; void* dW2; void* db2; void* dW3; void* db3;
} coattn_phrase_param_grads;

/* saved: forward -> backward state (argmax index per output element, then the status words of coattn_phrase_status);
 * ws_*: scratch.  flags: COATTN_FLAG_BF16_PROJ, COATTN_FLAG_FAST16 (see "Widths of the fp32 mode"); pass the same flags to the
 * forward and the backward of a step. */
int coattn_phrase_workspace_bytes(int B, int T, int E, int dtype, size_t* saved, size_t* ws_fwd, size_t* ws_bwd);

/* X [B,T,E] (rows past a question's length are zeros, as the embedding delivers them) -> out [B,T,E].
 * saved may be NULL for inference. */
int coattn_phrase_forward(const void* X, const coattn_phrase_params* p, void* out, void* saved, void* ws,
                          int B, int T, int E, int dtype, int flags, void* stream);

/* g_out [B,T,E] -> dX [B,T,E] (overwritten; NULL to skip) and the six parameter gradients
 * (accumulate = 0 overwrites, 1 adds).  `out` is the forward's output. */
int coattn_phrase_backward(const void* X, const coattn_phrase_params* p, const void* out, const void* saved,
                           const void* g_out, void* dX, const coattn_phrase_param_grads* pg, int accumulate,
                           void* ws, int B, int T, int E, int dtype, int flags, void* stream);

/* Range report of coattn_phrase_forward under COATTN_FLAG_FAST16 (its product Z = Xcat Wcat^T then runs on two FP16
 * pieces, like the co-attention's projections): as coattn_status -- synchronises, 0 / -4 -- on the status words behind the
 * argmax bytes of `saved` (saved must have been given to the forward call). */
int coattn_phrase_status(const void* saved, int B, int T, int E, void* stream, float* amax);
/* as coattn_status_accumulate, on the status words behind the phrase level's `saved` */
int coattn_phrase_status_accumulate(const void* saved, int B, int T, int E, void* acc, void* stream);

/* ---- cross entropy of the train step ------------------------------------------------
 * coattn_ce_forward replaces `nn.CrossEntropyLoss()(logits, label)` (main.py:94, :214; mean over the batch)
 * together with its gradient.  The MLPClassifier that produces the logits (model.py:400-434) stays on the stock
 * PyTorch-ROCm modules unless the fused head below is used (coattn_head_forward). */
/* Mean cross entropy over B rows of [B,K] logits with int64 labels in [0,K):
 * loss[0] = mean_i (logsumexp(z_i) - z_i[label_i]);  dlogits [B,K] = (softmax(z) - onehot) / B, or NULL to skip.
 * ws: coattn_ce_workspace_bytes.  A label outside [0,K) -- where nn.CrossEntropyLoss raises -- makes the loss NaN and
 * sets a status word in `ws`; the call itself stays asynchronous.  coattn_ce_status(ws, B, stream) SYNCHRONISES the
 * stream and returns -2 (coattn_last_error names the row) if the last coattn_ce_forward on this `ws` met such a label,
 * else 0: call it where the host synchronises anyway (reading the loss). */
int coattn_ce_workspace_bytes(int B, int K, int dtype, size_t* ws);
int coattn_ce_forward(const void* logits, const void* labels, void* loss, void* dlogits, void* ws, int B, int K,
                      int dtype, void* stream);
int coattn_ce_status(const void* ws, int B, void* stream);

/* ---- answer head: MLPClassifier + cross entropy and their backward ----------------------------
 * coattn_head_forward replaces `MLPClassifier.forward` (model.py:414-434, called at model.py:185 with the two lists
 * co-attention returns) and, when labels are given, `criterion(logits, label)` (main.py:214):
 *     h_w = tanh(W_w (q_w + v_w) + b_w); h_p = tanh(W_p [q_p + v_p | h_w] + b_p); h_s = tanh(W_s [q_s + v_s | h_p] + b_s);
 *     logits = W_h h_s + b_h;  loss = mean cross entropy.
 * coattn_head_backward replaces their autograd graph (main.py:219-220).  The adds, the concatenations, bias + tanh and
 * tanh' are folded into the operand addressing / epilogues of 32 x 32-tile products on the exact-fp32 MFMA (head.hip):
 * 4 launches + 2 for the loss forward, 4 launches backward.
 *   v, q   : host arrays of 3 device pointers [B,d] each (word, phrase, sentence: the rows of co-attention's v_out /
 *            q_out, or any three tensors);  W_w [d,d], W_p [d,2d], W_s [mlp,2d], W_h [K,mlp] as nn.Linear stores them.
 *   labels : int64 [B] or NULL (then loss must be NULL too: logits only, e.g. validation's argmax).
 *   logits : [B,K] (written);  loss: [1] (written).  A label outside [0,K) makes the loss NaN and sets the status
 *            word coattn_head_status(saved, ...) reports (-2; it synchronises the stream, like coattn_ce_status).
 *   saved  : forward -> backward state (h_w, h_p, h_s, d loss / d logits, row losses): coattn_head_workspace_bytes.
 * Backward: g_loss [1] (device) scales the saved d loss / d logits; g_logits [B,K] (may be NULL) is added to it -- the two
 * upstream gradients autograd can hand over; at least one must be given.  dv, dq: host arrays of 3 device pointers [B,d]
 * (overwritten; the gradient of q_l + v_l goes to both; dq may be NULL or equal dv: stored once; dv NULL: no input
 * gradients).  pg: the eight parameter gradients (accumulate = 0 overwrites, 1 adds).  ws: scratch of `ws_bwd` bytes. */
/* flags bit 0 of coattn_head_forward / coattn_head_backward: the layers of a direction as phases of ONE launch separated by
 * grid-wide barriers instead of one launch per layer (same tiles, same values; measured slower or equal: opt-in).  The
 * barriers' spins are bounded: should one time out (the grid not resident), the call's first output element -- logits[0][0],
 * and with it the loss; the first gradient element -- is NaN rather than silently stale. */
#define COATTN_HEAD_PERSISTENT 1
/* flags bit 2 (COATTN_FLAG_BF16_PROJ) of both: the reduced-precision mode -- the operands of the four products and of their
 * gradients rounded to bf16 while they are fed to the matrix pipe (ONE v_mfma_f32_32x32x16_bf16 where the exact head issues
 * eight v_mfma_f32_32x32x2_f32), fp32 accumulation, biases, tanh, cross entropy and bias gradients; bf16 tolerance.  Pass the
 * same flag to the forward and the backward of a step.  (The one-launch form is exact only: bits 0 and 2 together are an
 * argument error, -1.) */
typedef struct coattn_head_params {
  const void* W_w; const void* b_w;   /* model.py:409 */
  const void* W_p; const void* b_p;   /* model.py:410 */
  const void* W_s; const void* b_s;   /* model.py:411 */
  const void* W_h; const void* b_h;   /* model.py:412 */
} coattn_head_params;
typedef struct coattn_head_param_grads {
  void* dW_w; void* db_w; void* dW_p; void* db_p; void* dW_s; void* db_s; void* dW_h; void* db_h;
} coattn_head_param_grads;
int coattn_head_workspace_bytes(int B, int d, int mlp, int K, int dtype, size_t* saved, size_t* ws_bwd);
int coattn_head_forward(const void* const* v, const void* const* q, const coattn_head_params* p, const void* labels,
                        void* logits, void* loss, void* saved, int B, int d, int mlp, int K, int dtype, int flags,
                        void* stream);
int coattn_head_status(const void* saved, int B, int d, int mlp, int K, void* stream);
int coattn_head_backward(const void* const* v, const void* const* q, const coattn_head_params* p, const void* saved,
                         const void* g_loss, const void* g_logits, void* const* dv, void* const* dq,
                         const coattn_head_param_grads* pg, int accumulate, void* ws, int B, int d, int mlp, int K,
                         int dtype, int flags, void* stream);

/* ---- building blocks (exported for the per-kernel parity tests) ------------------------ */

/* Strided, batched fp32 GEMM on the f32 MFMA:
 *   C[z](m,n) = act( out_scale * (sum_{i<inner} sum_k A[z,i](m,k) * B[z,i](k,n) + bias_n[n] + bias_m[m])
 *                    + beta * Cin[z](m,n) )
 * Row m of A / C / Cin lives at (m / mdiv) * sdiv + (m % mdiv) * sm (mdiv = 0: plain m * sm).
 * ksplit > 0: batch index z selects the k range [z*ksplit, min(K,(z+1)*ksplit)) instead.
 * inner_total > 0: z is a group of `inner` consecutive inner indices ig = z*inner + i < inner_total.
 * Pointer tables (level-merged launches; NULL entries = unused): a_ptrs/b_ptrs/c_ptrs/cin_ptrs[t]
 * replace A/B/C/Cin with t = z (ptr_by_inner = 0) or t = ig (ptr_by_inner = 1).
 * b_imod > 0: the inner stride of B wraps, B[ig] = B + (ig % b_imod) * b_si.
 * kband_n > 0 (multiple of 128, at most 3 bands): columns [j*kband_n, (j+1)*kband_n) contract only
 * over k in [kband_lo[j], kband_hi[j]) -- block-sparse B operands (n-gram taps of phrase.hip). */
typedef struct coattn_gemm_desc {
  const void* A; const void* B; const void* Cin; void* C;
  const void* bias_n; const void* bias_m;
  int M, N, K, batch, inner, inner_total, ksplit, act; /* act: 0 none, 1 tanh */
  float beta;
  int64_t a_sm, a_sk, a_sz, a_si, a_mdiv, a_sdiv;
  int64_t b_sk, b_sn, b_sz, b_si;
  int64_t c_sm, c_sn, c_sz, c_mdiv, c_sdiv;
  int64_t cin_sm, cin_sn, cin_sz, cin_mdiv, cin_sdiv;
  const void* a_ptrs[8]; const void* b_ptrs[8]; void* c_ptrs[8]; const void* cin_ptrs[8];
  int ptr_by_inner; int b_imod;
  int kband_n; int kband_lo[3]; int kband_hi[3];
  float out_scale;          /* 0 or 1: plain; else C = act(out_scale * (products + biases) + beta * Cin) */
} coattn_gemm_desc;

int coattn_gemm_f32(const coattn_gemm_desc* g, void* stream);

/* ---- nn.Linear against a pre-split weight ----------------------------------------------------------------
 * y[M][N] = out_scale * (x[M][K] W[N][K]^T + bias[N]) in fp32 accuracy (the W_v / W_q projections of
 * ParallelCoAttention.forward, model.py:380-384: x rows ld_x floats apart, W as nn.Linear stores it).
 * The weight is split once into its three bf16 pieces in MFMA-fragment order (`wimg`, device scratch of
 * coattn_linear_workspace_bytes(N, K) bytes) and the GEMM reads the fragments in place (gemm_w.hip) -- the
 * kernel pair coattn_forward uses for both projections.  flags bit 0: `wimg` already holds the image of this
 * W (skip the split) -- WRITTEN BY A CALL WITH THE SAME N, K, precision flags (bits 2, 3) AND ON THE SAME KERNEL: the image's
 * format belongs to the kernel that reads it (three-piece fragments for gemm_w.hip, hi-only 1 KB chunks for gemm_bf.hip), and
 * which kernel runs depends on M (>= 256 and a multiple of 256 for gemm_bf.hip), ld_x and the alignment of x as well.  Reuse
 * an image only across calls of one shape class -- e.g. the steps of a loop over equal batches; a last partial batch with
 * another M must split again.  The library cannot check this (the image carries no tag).  flags bit 2 (COATTN_FLAG_BF16_PROJ): operands rounded to bf16, one MFMA per product.  K % 32 == 0, M >= 128, x 16-byte aligned with ld_x % 4 == 0; other shapes: error -1
 * (use coattn_gemm_f32).  bias may be NULL; out_scale 0 means 1.
 * flags bit 3 (COATTN_FLAG_BF16_IN, with bit 2): x holds bf16 elements (ld_x in elements, % 8 == 0) -- the activations of
 * an autocast encoder, or the fused backward's own bf16 gradients -- read as they are by the wide-shape kernel
 * (gemm_bf.hip: M >= 256, N % 256 == 0, K % 64 == 0; other shapes: error -1). */
size_t coattn_linear_workspace_bytes(int N, int K);
int coattn_linear_forward(const void* x, int64_t ld_x, const void* W, const void* bias, void* y, void* wimg,
                          int M, int N, int K, float out_scale, int flags, void* stream);

/* Weight gradient of the same layer: dW[n_out][n_in] (+)= dY[M][n_out]^T X[M][n_in] in fp32 accuracy
 * (autograd of the W_v / W_q projections, main.py:219-220): split-K parts on the hand-scheduled A^T B kernel
 * (gemm_tn.hip) + a deterministic reduce -- the kernel pair coattn_backward uses for dW_v and dW_q.
 * dY rows ld_dy floats apart, X rows ld_x; `ws`: device scratch of coattn_linear_wgrad_workspace_bytes bytes.
 * n_out, n_in multiples of 128, M >= 16, 16-byte aligned operands with ld % 4 == 0; other shapes: error -1.
 * accumulate: bit 0 adds onto dW; bit 2 (COATTN_FLAG_BF16_PROJ) selects the reduced-precision mode; bit 3
 * (COATTN_FLAG_BF16_IN, with bit 2): dy holds bf16 elements (ld_dy % 8 == 0; n_out, n_in % 256 == 0, M % 32 == 0). */
size_t coattn_linear_wgrad_workspace_bytes(int n_out, int n_in);
int coattn_linear_weight_grad(const void* dy, int64_t ld_dy, const void* x, int64_t ld_x, void* dW, void* ws, int M,
                              int n_out, int n_in, int accumulate, void* stream);

/* Same contract with the operands rounded to bf16 (round to nearest even) while they are staged and
 * contracted on v_mfma_f32_32x32x16_bf16 (fp32 accumulate / output): the arithmetic of
 * COATTN_FLAG_BF16_PROJ.  Shapes the bf16 kernels do not take (unaligned strides, M < 128) are computed
 * in exact fp32 instead. */
int coattn_gemm_bf16(const coattn_gemm_desc* g, void* stream);

/* ---- one-shot gradient exchange over peer-mapped buckets (data-parallel training) -----------
 * The reference trains on one GPU (multi-GPU is a TODO: main.py:71, :102-106).  The co-attention model shards by QA
 * pair, so the only exchange is the gradient mean; on a fully connected xGMI node the pattern that fits is one shot:
 * every rank maps its peers' gradient buckets (HIP IPC, once) and two bandwidth kernels read peer memory directly.
 * peer_bufs[r] = rank r's bucket (world * shard_elems floats) as mapped in THIS process, peer_bufs[rank] its own.
 *   coattn_p2p_reduce_scatter  own bucket, shard `rank`  <-  scale * sum_r peer_bufs[r][shard `rank`], summed in rank
 *                              order (identical and repeatable on every rank);
 *   coattn_p2p_all_gather      own bucket, shard r  <-  peer_bufs[r][shard r] for every r != rank.
 * Both only ENQUEUE on `stream`; the caller orders the phases across ranks (every bucket packed before the first call,
 * every reduce done before the gather, every gather done before a bucket is packed again -- dist.py: a one-element
 * all-reduce on the stream under RCCL).  shard_elems % 4 == 0, 16-byte aligned buckets, world <= 8; else error -1. */
int coattn_p2p_enable_peer(int peer_device);   /* peer access from the current device to the device a mapped bucket lives on
                                                * (once per pair, before the first call below); -1 if they cannot reach each other */
int coattn_p2p_reduce_scatter(const void* const* peer_bufs, int world, int rank, int64_t shard_elems, float scale,
                              void* stream);
int coattn_p2p_all_gather(const void* const* peer_bufs, int world, int rank, int64_t shard_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COATTN_H */
