/*
 * include/ftr.h -- C ABI of the MI355X-native pruned RNN-T loss hot path ("ftr" = fast transducer).
 *
 * This is the drop-in boundary for the native layer of Samsung/tf-fast-rnnt: every entry point
 * replaces one host launcher / op kernel of the reference (cited per function, paths relative to
 * /root/reference).  Conventions, fixed for every function:
 *
 *   - plain C types only: device pointers (hipMalloc'd, or anything HIP can dereference on the
 *     current device), int sizes, an opaque `void* stream` (a hipStream_t; NULL = default stream);
 *   - dense row-major tensors, last axis contiguous, float32 / int32 exactly as the reference's op
 *     registration (tf_fast_rnnt/python/csrc/tf_fast_rnnt_op.cc:27-38);
 *   - asynchronous on `stream`, no host synchronisation, no allocation inside (graph-capturable);
 *     scratch is passed in by the caller (the reference allocates it with allocate_temp,
 *     tf_fast_rnnt_op.cc:66-67,90-91);
 *   - return value: 1 on success -- the reference's launchers return 1 (mutual_information_cuda.cu:810,
 *     873,1011) and its op maps anything else to tf::errors::Internal (tf_fast_rnnt_op.cc:114-116);
 *     <= 0 is an FTR_ERR_* code and ftr_last_error() describes it.  There is no CPU fallback: with no
 *     usable HIP device the functions return FTR_ERR_NO_DEVICE.
 */
#ifndef FTR_H_
#define FTR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTR_OK 1
#define FTR_ERR_INVALID_ARG 0
#define FTR_ERR_UNSUPPORTED (-1)
#define FTR_ERR_LAUNCH (-2)
#define FTR_ERR_NO_DEVICE (-3)

/* ABI version (major*100 + minor) and the package version string the Python layer re-exports
 * (__version__ == "1.2", tf_fast_rnnt/python/tf_fast_rnnt/__init__.py:36). */
int ftr_abi_version(void);
const char* ftr_package_version(void);
/* Thread-local description of the last non-success return on this thread ("" if none). */
const char* ftr_last_error(void);
/* Number of floats of fwd->bwd workspace (`p` below) for a problem size.  `p` is NOT the reference's [B,S+1,T+1] temp
 * (tf_fast_rnnt_op.cc:65-67): it is about TWO lattices of that shape (one split-ratio lattice per direction of the
 * recursion), the values on the cut, and an inter-workgroup hand-off region -- a buffer of B*(S+1)*(T+1) floats is too
 * small and the kernels would write past its end.  Callers that cannot guarantee the size use the _ws entry points
 * below, which check it.  Its CONTENT is implementation defined: the same buffer must be handed
 * unchanged from _fwd to _bwd.  Must be 16-byte aligned. */
size_t ftr_mutual_information_workspace_floats(int B, int S, int T);

/* flags of the _ws entry points */
#define FTR_MI_WS_CLEAN 1 /* the hand-off region of `p` is known to be all zero: it was zeroed once by
                             ftr_mutual_information_workspace_init() and since then only touched by launches of this
                             library that completed with status 0 (they leave it zero again).  Saves the memset node
                             in front of the forward launch.  Graph capture / replay is fine (nothing depends on a
                             launch counter).
                             ONE BUFFER FOR MANY SHAPES.  The hand-off region of a launch is the LAST
                             ftr_mutual_information_handoff_floats(B,S,T) floats of the buffer it is given (p + p_floats
                             is what anchors it), everything else grows from the front.  A caller whose (B,S,T) change
                             from step to step (padded batches) keeps one buffer of
                                 max over shapes of (workspace_floats - handoff_floats)  +  max over shapes of handoff_floats
                             floats, zeroes its last max-handoff floats once (workspace_init with p_floats = the whole
                             buffer and the shape with the largest hand-off part, or a memset), always passes the whole
                             buffer, and may then pass FTR_MI_WS_CLEAN for every one of those shapes: no per-step
                             allocation, no per-step memset (tf_fast_rnnt/mutual_information.py does exactly this). */

/* floats of the hand-off region (control block + granules) within ftr_mutual_information_workspace_floats(B,S,T) */
size_t ftr_mutual_information_handoff_floats(int B, int S, int T);

/* Zeroes the hand-off region of a workspace (asynchronous on `stream`): the last handoff_floats(B,S,T) floats of
 * p[0, p_floats).  Once per buffer (see FTR_MI_WS_CLEAN). */
/* ftr_mutual_information_bwd_ws_f32 with ans_grad = NULL (a seed of ones, no self check) that ALSO writes the loss tail of
 * rnnt_loss.py:333,544-546 -- loss_out = -ans [B] (reduction 0), -mean(ans) (1) or -sum(ans) (2), the values
 * ftr_negated_reduce_f32 gives, bit for bit -- from `ans` as the forward launch left it: the loss nodes' forward is
 * fwd + this, one kernel boundary less than fwd + bwd + negated_reduce. */
int ftr_mutual_information_bwd_loss_ws_f32(const float* px, const float* py, const int32_t* boundary, const float* p,
                                           size_t p_floats, int flags, float* px_grad, float* py_grad, const float* ans,
                                           int reduction, float* loss_out, int B, int S, int T, int modified, void* stream);
int ftr_mutual_information_workspace_init(float* p, size_t p_floats, int B, int S, int T, void* stream);

/* Reads back the sticky status word of a workspace (SYNCHRONISES `stream`): 0 = fine; bit 0 = some band gave up
 * waiting for the band above it (bounded poll; cannot happen unless a producer workgroup never ran): the results of
 * that launch are poisoned (ans = NaN) and the workspace must be re-initialised.  dirty_words_host (nullable,
 * diagnostic) receives the number of non-zero words left in the hand-off region, which is 0 between launches. */
int ftr_mutual_information_status(const float* p, size_t p_floats, int B, int S, int T, int* status_host,
                                  long long* dirty_words_host, void* stream);

/*
 * Forward recursion.  Replaces MutualInformationCuda<float>
 * (tf_fast_rnnt/csrc/mutual_information.h:134-141, mutual_information_cuda.cu:765-811).
 *   px [B,S,T+1] (modified==0) or [B,S,T] (modified!=0); py [B,S+1,T];
 *   boundary [B,4] = (s_begin,t_begin,s_end,t_end) or NULL => (0,0,S,T);
 *   p   workspace, ftr_mutual_information_workspace_floats() floats, written;
 *   ans [B] = p[b,s_end,t_end] of the recursion documented at mutual_information.h:101-126; 0 for an empty
 *       rectangle; -inf when no path exists.
 * NaN inputs: the reference's LogAdd (mutual_information.h:70-83) lets a NaN through or drops it depending on which
 * argument it arrives in; here ANY NaN among the px / py entries inside the boundary rectangle of utterance b gives
 * ans[b] = NaN (and zero occupancies), so that a diverged model is seen.  -inf entries are ordinary ("impossible
 * transition").
 * The _ws form takes the size of `p` (returns FTR_ERR_INVALID_ARG when it is too small instead of writing past it)
 * and the FTR_MI_* flags; the plain form trusts the caller and passes no flags.
 */
int ftr_mutual_information_fwd_f32(const float* px, const float* py, const int32_t* boundary,
                                   float* p, float* ans, int B, int S, int T, int modified,
                                   void* stream);
int ftr_mutual_information_fwd_ws_f32(const float* px, const float* py, const int32_t* boundary, float* p,
                                      size_t p_floats, int flags, float* ans, int B, int S, int T, int modified,
                                      void* stream);

/*
 * Backward recursion.  Replaces MutualInformationBackwardCuda<float>
 * (mutual_information.h:151-162, mutual_information_cuda.cu:817-874) together with the two memsets
 * of the op (tf_fast_rnnt_op.cc:93-96): px_grad (shape of px) and py_grad (shape of py) are FULLY
 * written, zeros outside the boundary rectangle.
 *   p        the workspace written by ftr_mutual_information_fwd_f32 on the same inputs;
 *   p_grad   optional scratch [B,S+1,T+1]; only the "plain" family uses it (may be NULL for
 *            "wavefront", which keeps p_grad on chip);
 *   ans_grad [B], read; if overwrite_ans_grad it is overwritten with p_grad[b,s_begin,t_begin],
 *            which equals the seed when everything is consistent (mutual_information_cuda.cu:510-514).
 *            NULL means "all ones" (what the op itself passes, tf_fast_rnnt_op.cc:104-107) without the
 *            write-back; accepted by the default ("wavefront") family only.
 */
int ftr_mutual_information_bwd_f32(const float* px, const float* py, const int32_t* boundary,
                                   const float* p, float* p_grad, float* px_grad, float* py_grad,
                                   float* ans_grad, int overwrite_ans_grad, int B, int S, int T,
                                   int modified, void* stream);
int ftr_mutual_information_bwd_ws_f32(const float* px, const float* py, const int32_t* boundary, const float* p,
                                      size_t p_floats, int flags, float* p_grad, float* px_grad, float* py_grad,
                                      float* ans_grad, int overwrite_ans_grad, int B, int S, int T, int modified,
                                      void* stream);

/* Inclusive prefix-min along rows of an int32 [rows, cols] matrix.  Replaces CumminCuda<int32_t>
 * (mutual_information.h:164-168, mutual_information_cuda.cu:895-1012; op "Cummin",
 * tf_fast_rnnt_op.cc:36-38,135-165). */
int ftr_cummin_i32(const int32_t* in, int32_t* out, int rows, int cols, void* stream);

/*
 * Prune ranges.  Replaces get_rnnt_prune_ranges + _adjust_pruning_lower_bound +
 * _monotonic_lower_bound (tf_fast_rnnt/python/tf_fast_rnnt/rnnt_loss.py:553-761), i.e. ~15 TF ops
 * and two Cummin launches, by two kernels.
 *   px_grad [B,S,T1] (T1 = T+1 regular, T modified), py_grad [B,S+1,T], boundary [B,4] (required);
 *   s_range as given by the caller; r_eff = (s_range > S ? S+1 : s_range) (rnnt_loss.py:710-711);
 *   ranges  [B,T,r_eff] int32, written; s_begin_scratch [B,T] int32 scratch.
 * Canonical arithmetic (bit-exact with oracle/): cumsum along S sequential in f32, window sum
 * cumsum[s0+r]-cumsum[s0], minus px_grad[s0-1] (0 for s0==0), first maximum wins.
 * Returns FTR_OK; r_eff is returned through *r_eff_out when non-NULL.
 */
int ftr_prune_ranges_i32(const float* px_grad, const float* py_grad, const int32_t* boundary,
                         int32_t* ranges, int32_t* s_begin_scratch, int B, int S, int T, int T1,
                         int s_range, int* r_eff_out, void* stream);

/* Prune gather.  Replaces do_rnnt_pruning (rnnt_loss.py:763-812):
 * am_pruned[b,t,k,:] = am[b,t,:], lm_pruned[b,t,k,:] = lm[b,ranges[b,t,k],:].
 * am [B,T,C], lm [B,S1,C], ranges [B,T,r]; outputs [B,T,r,C].
 * am_pruned may be NULL: only lm_pruned is produced (am_pruned is a broadcast of am over r; a host whose tensors
 * have strides keeps it as a view -- the shipped Python package does -- and saves B*T*r*C*4 bytes of writes). */
int ftr_do_pruning_f32(const float* am, const float* lm, const int32_t* ranges, float* am_pruned,
                       float* lm_pruned, int B, int T, int S1, int C, int r, void* stream);

/* Backward of the prune gather (TF autodiff of rnnt_loss.py:802-811): d_am[b,t,:] = sum_k g_am_pruned[b,t,k,:];
 * d_lm[b,s,:] = sum over {(t,k): ranges[b,t,k] == s} of g_lm_pruned[b,t,k,:] in increasing (t,k) order
 * (deterministic; no atomics).  d_am [B,T,C] and d_lm [B,S1,C] are fully written. */
int ftr_do_pruning_bwd_f32(const float* g_am_pruned, const float* g_lm_pruned, const int32_t* ranges, float* d_am,
                           float* d_lm, int B, int T, int S1, int C, int r, void* stream);

/* The same backward with a caller-provided scratch buffer (ftr_do_pruning_bwd_workspace_bytes() bytes, 16-byte
 * aligned, contents irrelevant): the incoming gradient is streamed once in chunks of 16 frames, each chunk's rows are
 * summed per lattice row in LDS in (t,k) order, and the per-chunk partial rows are then added in chunk order
 * (deterministic, no atomics; the association differs from the plain left-to-right sum of the function above).
 * When g_am_pruned == g_lm_pruned (a joiner that starts with am_pruned + lm_pruned hands the same buffer to both)
 * d_am comes out of the same pass.  Arbitrary `ranges` are accepted.  Falls back to the kernels above when
 * C % 4 != 0. */
size_t ftr_do_pruning_bwd_workspace_bytes(int B, int T, int S1, int C, int r);
int ftr_do_pruning_bwd_ws_f32(const float* g_am_pruned, const float* g_lm_pruned, const int32_t* ranges, float* d_am,
                              float* d_lm, int B, int T, int S1, int C, int r, void* workspace,
                              size_t workspace_bytes, void* stream);

/*
 * Pruned log-probs, forward.  Replaces get_rnnt_logprobs_pruned for rnnt_type "regular"
 * (modified==0) / "modified" (modified!=0) (rnnt_loss.py:853-1020: logsumexp, two gathers, pad,
 * _roll_by_shifts, transpose, fix_for_boundary) and, when delay_penalty > 0, the penalty block
 * (rnnt_loss.py:1097-1114).
 *   logits [B,T,r,C]; symbols [B,S]; ranges [B,T,r]; boundary [B,4] or NULL;
 *   lse [B,T,r] written (kept for the backward);
 *   px [B,S,T1], py [B,S+1,T] written in full: band values, -inf elsewhere, px[:,:,T] = -inf and
 *   px[b,:,t_end[b]] = -inf for regular.
 */
int ftr_pruned_logprobs_fwd_f32(const float* logits, const int32_t* symbols, const int32_t* ranges,
                                const int32_t* boundary, int termination_symbol, double delay_penalty,
                                float* lse, float* px, float* py, int B, int T, int S, int C, int r,
                                int modified, void* stream);

/*
 * Pruned log-probs, backward: d loss / d logits given d loss / d px and d loss / d py lattices
 * (what TF autodiff replays for rnnt_loss.py:942-1016).  With scale != NULL the lattice gradients
 * are multiplied by scale[b] on the fly (fuses _RNNTLossGrad, __init__.py:154-162).
 *   glogits [B,T,r,C] fully written.
 */
int ftr_pruned_logprobs_bwd_f32(const float* logits, const int32_t* symbols, const int32_t* ranges,
                                const int32_t* boundary, int termination_symbol, const float* lse,
                                const float* gpx, const float* gpy, const float* scale,
                                float* glogits, int B, int T, int S, int C, int r, int modified,
                                void* stream);

/*
 * Simple-loss px/py builder (joiner = addition).  Replaces get_rnnt_logprobs (rnnt_loss.py:63-223), fix_for_boundary
 * (:28-61), the delay-penalty block (:305-321) and their autodiff, except for the three dense contractions, which
 * the host issues as library GEMMs between these calls (forward: prod = lm_probs @ am_probs^T [B,S+1,T]; backward:
 * dlmp = W @ am_probs [B,S+1,C], damp = W^T @ lm_probs [B,T,C]).
 *   ftr_rowmax_exp_f32          probs[row,:] = exp(x[row,:] - max), rowmax[row] = max            (:175-178)
 *   ftr_simple_logprobs_fwd_f32 px [B,S,T+1|T], py [B,S+1,T] from am [B,T,C], lm [B,S+1,C], symbols [B,S],
 *                               prod, am_max [B,T], lm_max [B,S+1]; -inf column / boundary column / penalty fused
 *   ftr_simple_logprobs_bwd_w_f32   W = -(gpx' + gpy)/(prod + tiny) [B,S+1,T]; rsx, rsy [B,S+1] row sums
 *   ftr_simple_logprobs_bwd_am_f32  d am [B,T,C] = damp*am_probs + scatter_s gpx'[b,s,t] -> column symbols[b,s]
 *                                   + blank column sums of gpy
 *   ftr_simple_logprobs_bwd_lm_f32  d lm [B,S+1,C] = dlmp*lm_probs + rsx at the symbol column + rsy at blank
 * gpx' = gpx with the cells the forward overwrote with -inf (t == T, t == t_end; regular only) masked out.
 */
int ftr_rowmax_exp_f32(const float* x, float* probs, float* rowmax, long long rows, int C, void* stream);
/* the same for two matrices of C columns in one launch (am [B*T,C] and lm [B*(S+1),C] of the simple builder) */
int ftr_rowmax_exp_pair_f32(const float* x1, float* probs1, float* rowmax1, long long rows1, const float* x2, float* probs2,
                            float* rowmax2, long long rows2, int C, void* stream);
int ftr_simple_logprobs_fwd_f32(const float* am, const float* lm, const int32_t* symbols, const float* prod,
                                const float* am_max, const float* lm_max, const int32_t* boundary,
                                int termination_symbol, double delay_penalty, float* px, float* py, int B, int T,
                                int S, int C, int modified, void* stream);
int ftr_simple_logprobs_bwd_w_f32(const float* gpx, const float* gpy, const float* prod, const int32_t* boundary,
                                  float* W, float* rsx, float* rsy, int B, int T, int S, int modified, void* stream);
int ftr_simple_logprobs_bwd_am_f32(const float* gpx, const float* gpy, const float* damp, const float* am_probs,
                                   const int32_t* symbols, const int32_t* boundary, int termination_symbol,
                                   float* d_am, int B, int T, int S, int C, int modified, void* stream);
int ftr_simple_logprobs_bwd_lm_f32(const float* dlmp, const float* lm_probs, const int32_t* symbols,
                                   const float* rsx, const float* rsy, int termination_symbol, float* d_lm, int B,
                                   int S, int C, void* stream);

/*
 * Smoothed px/py builder.  Replaces get_rnnt_logprobs_smoothed (rnnt_loss.py:1132-1367) and its autodiff: the
 * same kernels as above with the LM-only and AM-only interpolation terms (rnnt_loss.py:1342-1360) folded in,
 *   out = combined_scale (x - normalizers) + lm_only_scale (lm - lmonly_norm[b,s])
 *         + am_only_scale (am + unigram_log[c] - amonly_norm[b,t]).
 * The batch statistics are small vectors the host prepares between calls (rnnt_loss.py:1276-1290):
 *   lmonly_norm [B,S+1] = log(rowsum lm_probs) + lm_max   (ftr_rowmax_exp_sum_f32 also returns the row sums)
 *   unigram [C] = mean_{b,s} lm_probs/rowsum + tiny; unigram_log = log unigram
 *   am_dot [B,T] = am_probs . unigram;  amonly_norm = log am_dot + am_max
 * Backward: W carries combined_scale; bwd_am adds direct_scale (= combined + am_only) on the scattered terms and
 * am_probs * unigram * R with R[b,t] = -am_only_scale * colsum(gpx' + gpy) / am_dot (R is written out, it feeds
 * d unigram); bwd_lm adds direct_scale (= combined + lm_only) and lm_probs * (row_term[b,s] +
 * unigram_grad[c] * inv_rowsum[b,s]).  Zero scales must already be replaced by 1e-20 (rnnt_loss.py:1346-1349).
 */
int ftr_rowmax_exp_sum_f32(const float* x, float* probs, float* rowmax, float* rowsum, long long rows, int C,
                           void* stream);
int ftr_smoothed_logprobs_fwd_f32(const float* am, const float* lm, const int32_t* symbols, const float* prod,
                                  const float* am_max, const float* lm_max, const float* lmonly_norm,
                                  const float* amonly_norm, const float* unigram_log, const int32_t* boundary,
                                  int termination_symbol, float combined_scale, float lm_only_scale,
                                  float am_only_scale, float* px, float* py, int B, int T, int S, int C,
                                  int modified, void* stream);
int ftr_smoothed_logprobs_bwd_w_f32(const float* gpx, const float* gpy, const float* prod, const int32_t* boundary,
                                    float combined_scale, float* W, float* rsx, float* rsy, int B, int T, int S,
                                    int modified, void* stream);
int ftr_smoothed_logprobs_bwd_am_f32(const float* gpx, const float* gpy, const float* damp, const float* am_probs,
                                     const int32_t* symbols, const int32_t* boundary, int termination_symbol,
                                     float direct_scale, const float* unigram, const float* am_dot,
                                     float am_only_scale, float* R, float* d_am, int B, int T, int S, int C,
                                     int modified, void* stream);
int ftr_smoothed_logprobs_bwd_lm_f32(const float* dlmp, const float* lm_probs, const int32_t* symbols,
                                     const float* rsx, const float* rsy, int termination_symbol, float direct_scale,
                                     const float* row_term, const float* inv_rowsum, const float* unigram_grad,
                                     float* d_lm, int B, int S, int C, void* stream);

/*
 * Loss tail and upstream-gradient scaling, so that a whole loss (builder -> recursion -> reduction, and its backward)
 * runs without framework-side elementwise passes over the lattices.
 *   ftr_negated_reduce_f32: the batch reduction of rnnt_loss.py:333,544-546,1124-1126,1487-1489 on the device:
 *     reduction 0 "none": out[b] = -ans[b];  1 "mean": out[0] = -mean(ans);  2 "sum": out[0] = -sum(ans).
 *     One block, fixed summation tree (deterministic).
 *   *_scaled_f32: the backward entry points above with the registered gradient of the op (__init__.py:154-162:
 *     occupancy * upstream gradient) fused in: every incoming lattice gradient of utterance b is multiplied by
 *     (scale ? scale[b * scale_stride] : 1) * scale_mul while it is read.  scale_stride 0 = one device scalar for the
 *     whole batch (the gradient of a "sum"/"mean" loss); scale_mul carries the sign and the 1/B of "mean".
 */
int ftr_negated_reduce_f32(const float* ans, int B, int reduction, float* out, void* stream);
int ftr_pruned_logprobs_bwd_scaled_f32(const float* logits, const int32_t* symbols, const int32_t* ranges,
                                       const int32_t* boundary, int termination_symbol, const float* lse,
                                       const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                       float scale_mul, float* glogits, int B, int T, int S, int C, int r,
                                       int modified, void* stream);
int ftr_simple_logprobs_bwd_w_scaled_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                         float scale_mul, const float* prod, const int32_t* boundary, float* W,
                                         float* rsx, float* rsy, int B, int T, int S, int modified, void* stream);
int ftr_simple_logprobs_bwd_am_scaled_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                          float scale_mul, const float* damp, const float* am_probs,
                                          const int32_t* symbols, const int32_t* boundary, int termination_symbol,
                                          float* d_am, int B, int T, int S, int C, int modified, void* stream);

/* The smoothed loss as one node (rnnt_loss_smoothed, rnnt_loss.py:1369-1494, without framework-side passes over the
 * lattices): the smoothed builder with the delay-penalty block (rnnt_loss.py:1461-1478) folded in, and the _scaled forms
 * of its two lattice-reading backward kernels (upstream gradient applied on the fly, see the simple-loss forms above). */
int ftr_smoothed_logprobs_fwd_pen_f32(const float* am, const float* lm, const int32_t* symbols, const float* prod,
                                      const float* am_max, const float* lm_max, const float* lmonly_norm,
                                      const float* amonly_norm, const float* unigram_log, const int32_t* boundary,
                                      int termination_symbol, double delay_penalty, float combined_scale,
                                      float lm_only_scale, float am_only_scale, float* px, float* py, int B, int T,
                                      int S, int C, int modified, void* stream);
/* Batch statistics of the smoothed builder (rnnt_loss.py:1276-1290 and what autodiff replays for them), MI355X
 * additions that replace four library matrix-vector products:
 *   ftr_rowmax_exp_dot_f32    ftr_rowmax_exp_f32 that also returns dot[row] = probs[row,:] . dotvec   (am_probs . unigram)
 *   ftr_rowdot_f32            dot[row] = x[row,:] . v
 *   ftr_colsum_weighted_f32   out[c] = sum_row w[row] * x[row,c], two deterministic stages through `workspace`
 *                             (ftr_colsum_weighted_workspace_floats(rows, C) floats)                                  */
int ftr_rowmax_exp_dot_f32(const float* x, float* probs, float* rowmax, const float* dotvec, float* dot, long long rows,
                           int C, void* stream);
int ftr_rowdot_f32(const float* x, const float* v, float* dot, long long rows, int C, void* stream);
size_t ftr_colsum_weighted_workspace_floats(long long rows, int C);
int ftr_colsum_weighted_f32(const float* x, const float* w, float* out, float* workspace, size_t workspace_floats,
                            long long rows, int C, void* stream);

/* The builders with the normaliser contraction inside the kernel (f32 MFMA, MI355X addition): replaces the batched
 * matmul of rnnt_loss.py:180-182 / :1270-1272 AND the epilogue above in one launch -- the [B,S+1,T] product never goes
 * through memory unless `prod` is non-NULL (the backward's W kernel still reads it).  am_probs / lm_probs / am_max /
 * lm_max come from ftr_rowmax_exp_f32.  Requires C % 4 == 0 (ftr_simple_logprobs_fused_supported); other sizes take the
 * library-GEMM route above.  Same outputs as ftr_simple_logprobs_fwd_f32 / ftr_smoothed_logprobs_fwd_pen_f32 up to the
 * summation order of the contraction. */
int ftr_simple_logprobs_fused_supported(int C);
int ftr_simple_logprobs_fused_fwd_f32(const float* am, const float* lm, const int32_t* symbols, const float* am_probs,
                                      const float* lm_probs, const float* am_max, const float* lm_max,
                                      const int32_t* boundary, int termination_symbol, double delay_penalty, float* px,
                                      float* py, float* prod, int B, int T, int S, int C, int modified, void* stream);
int ftr_smoothed_logprobs_fused_fwd_f32(const float* am, const float* lm, const int32_t* symbols, const float* am_probs,
                                        const float* lm_probs, const float* am_max, const float* lm_max,
                                        const float* lmonly_norm, const float* amonly_norm, const float* unigram_log,
                                        const int32_t* boundary, int termination_symbol, double delay_penalty,
                                        float combined_scale, float lm_only_scale, float am_only_scale, float* px,
                                        float* py, float* prod, int B, int T, int S, int C, int modified, void* stream);
/* The dense f32 contractions of the simple / smoothed builders that are not inside a hand-written kernel -- what TensorFlow
 * runs as tf.matmul in rnnt_loss.py:180-182 (and :1270-1272) and as the two matmuls of its autodiff -- as rocBLAS
 * strided-batched GEMMs on row-major operands:
 *   kind 0: out [B,S+1,T] = x . y^T   with x = lm_probs [B,S+1,C], y = am_probs [B,T,C]   (forward; only where the fused
 *           forward kernel does not apply: C % 4 != 0)
 *   kind 1: out [B,S+1,C] = x . y     with x = W [B,S+1,T],        y = am_probs [B,T,C]   (backward towards lm)
 *   kind 2: out [B,T,C]   = x^T . y   with x = W [B,S+1,T],        y = lm_probs [B,S+1,C] (backward towards am; where the fused
 *           d am kernel does not apply)
 * S1 = S + 1.  The library's kernel for a shape is chosen by measurement: at the SECOND call with the same shape (and never
 * inside a stream capture) every candidate rocBLAS offers is timed on the caller's stream (~0.2 s, `out` is recomputed by
 * each) and the fastest one is used from then on, for the life of the process (c3: 90 / 84 us with the library's own choice,
 * 62 / 60 us with the measured one).  FTR_GEMM_TUNE=off keeps the library's choice, =first measures at the first call.
 * ftr_normalizer_gemm_choice reports what was chosen for a shape on the current device: returns 1 and fills (any pointer may
 * be NULL) the rocBLAS solution index (0 = library's own), its time and the default's time in us, and the number of candidates
 * timed (-1 = not measured yet); 0 if the shape has not been seen.  ftr_normalizer_gemm_set_choice fixes the solution index
 * for a shape without measuring (a choice recorded by an earlier process on the same rocBLAS; an index the library rejects
 * falls back to its own choice). */
int ftr_normalizer_gemm_f32(int kind, const float* x, const float* y, float* out, int B, int T, int S1, int C, void* stream);
int ftr_normalizer_gemm_choice(int kind, int B, int T, int S1, int C, int* solution, float* us, float* us_default,
                               int* candidates);
int ftr_normalizer_gemm_set_choice(int kind, int B, int T, int S1, int C, int solution);

/* Backward towards am with the W^T . lm_probs contraction inside the kernel (f32 MFMA): replaces one of the two backward
 * matmuls AND ftr_*_logprobs_bwd_am_*: W is formed from g_px, g_py and prod while staging, the scatter by symbol runs as a
 * second small MFMA contraction against a one-hot operand, `damp` [B,T,C] never exists.  Scale arguments as in the _scaled
 * forms above.  Requires ftr_simple_logprobs_fused_bwd_supported(T, C) != 0 (T % 4 == 0 and C % 4 == 0); other sizes take
 * the library-GEMM route. */
int ftr_simple_logprobs_fused_bwd_supported(int T, int C);
int ftr_simple_logprobs_fused_bwd_am_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                         float scale_mul, const float* prod, const float* lm_probs, const float* am_probs,
                                         const int32_t* symbols, const int32_t* boundary, int termination_symbol,
                                         float* d_am, int B, int T, int S, int C, int modified, void* stream);
int ftr_smoothed_logprobs_fused_bwd_am_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                           float scale_mul, const float* prod, const float* lm_probs,
                                           const float* am_probs, const int32_t* symbols, const int32_t* boundary,
                                           int termination_symbol, float combined_scale, float direct_scale,
                                           const float* unigram, const float* am_dot, float am_only_scale, float* R,
                                           float* d_am, int B, int T, int S, int C, int modified, void* stream);
int ftr_smoothed_logprobs_bwd_w_scaled_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                           float scale_mul, const float* prod, const int32_t* boundary,
                                           float combined_scale, float* W, float* rsx, float* rsy, int B, int T, int S,
                                           int modified, void* stream);
int ftr_smoothed_logprobs_bwd_am_scaled_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                            float scale_mul, const float* damp, const float* am_probs,
                                            const int32_t* symbols, const int32_t* boundary, int termination_symbol,
                                            float direct_scale, const float* unigram, const float* am_dot,
                                            float am_only_scale, float* R, float* d_am, int B, int T, int S, int C,
                                            int modified, void* stream);

/*
 * The pruned loss on the band itself.  rnnt_loss_pruned (rnnt_loss.py:1022-1130) pads the band [B,T,r] to full-size px / py
 * lattices (rnnt_loss.py:968-1013) and runs the whole (S+1) x (T+1) recursion on them; these three entry points keep
 * everything band shaped ([B,T,r]; row (b,t,k) <-> lattice cell (ranges[b,t,0] + k, t)):
 *   ftr_pruned_band_fwd_f32           lse [B,T,r] (rnnt_loss.py:942) and px_band / py_band = the values the full-size
 *                                     lattices would hold at the band cells (-inf rules and delay penalty included)
 *   ftr_mutual_information_band_f32   forward recursion, cut, backward recursion in ONE launch (one workgroup per
 *                                     utterance, LDS resident): ans [B] and the occupancies gx_band / gy_band (= px_grad /
 *                                     py_grad at the band cells, seed = ones)
 *   ftr_pruned_band_bwd_scaled_f32    d loss / d logits from the band-shaped occupancies (the _scaled semantics above)
 * PRECONDITION on `ranges` (what get_rnnt_prune_ranges produces): for every utterance ranges[b,t,0] is non-decreasing in
 * t over the frames of the boundary rectangle.  ftr_mutual_information_band_f32 checks it on the device and answers a
 * violation with ans[b] = NaN and zero occupancies; callers with arbitrary ranges use ftr_pruned_logprobs_* + the lattice
 * recursion.  The band entry points also read ranges[b,t,0] only (row k of a frame is lattice row ranges[b,t,0] + k).
 * ftr_band_ranges_check_i32 tells whether a ranges tensor is such a band: flags[0] (device int, written) = 0 when it is,
 * bit 0 = not monotone, bit 1 = ranges[b,t,k] != ranges[b,t,0] + k somewhere, both inside the boundary rectangles only
 * (the Python layer runs it once per ranges tensor that does not come straight from get_rnnt_prune_ranges and routes
 * by the answer).
 * ftr_mutual_information_band_supported(T, S, r): 1 = the LDS-resident kernel (r <= 15 and 12 (S + T + 21) LANES +
 * 4 (T + 34) bytes <= 150 KB with LANES = 8 for r <= 7, else 16); 2 = the streaming kernel for longer utterances, which
 * keeps its wavefront-ordered arrays in a caller-provided workspace of ftr_mutual_information_band_workspace_floats()
 * floats (16-byte aligned; contents need not survive the call) -- pass it to ftr_mutual_information_band_ws_f32;
 * 0 = outside both (r > 15).  ftr_mutual_information_band_f32 is the _ws form without a workspace (kind 1 only).
 * From S + T >= 1100 the library runs the recursion in PARALLEL SEGMENTS instead (csrc/mi_band_seg.hip: transfer matrices of
 * <= 32 segments per direction, float64 chains, occupancies as exp(p + q - ans); c4 124 -> 67 us, c5 527 -> 146 us) whenever
 * the workspace it is given is large enough for that: ftr_mutual_information_band_workspace_floats() returns the larger of
 * the two needs (c3 11.7 MB, c4 21.3 MB, c5 39.2 MB; 0 below S + T = 1100 where the LDS kernel wins), a caller that passes less gets the chain
 * kernels.  FTR_BAND_IMPL = chain | segments forces one implementation for every size (tests, measurements).
 */
int ftr_mutual_information_band_supported(int T, int S, int r);
int ftr_band_ranges_check_i32(const int32_t* ranges, const int32_t* boundary, int32_t* flags, int B, int T, int r, void* stream);
size_t ftr_mutual_information_band_workspace_floats(int B, int T, int S, int r);
int ftr_mutual_information_band_ws_f32(const float* px_band, const float* py_band, const int32_t* ranges,
                                       const int32_t* boundary, float* workspace, size_t workspace_floats, float* ans,
                                       float* gx_band, float* gy_band, int B, int T, int S, int r, int modified,
                                       void* stream);
int ftr_pruned_band_fwd_f32(const float* logits, const int32_t* symbols, const int32_t* ranges, const int32_t* boundary,
                            int termination_symbol, double delay_penalty, float* lse, float* px_band, float* py_band,
                            int B, int T, int S, int C, int r, int modified, void* stream);
int ftr_mutual_information_band_f32(const float* px_band, const float* py_band, const int32_t* ranges,
                                    const int32_t* boundary, float* ans, float* gx_band, float* gy_band, int B, int T,
                                    int S, int r, int modified, void* stream);
int ftr_pruned_band_bwd_scaled_f32(const float* logits, const int32_t* symbols, const int32_t* ranges,
                                   const int32_t* boundary, int termination_symbol, const float* lse,
                                   const float* gx_band, const float* gy_band, const float* scale, int scale_stride,
                                   float scale_mul, float* glogits, int B, int T, int S, int C, int r, int modified,
                                   void* stream);

/* Hardware self-test used by smoke()/tests: checks on the device that the primitives the wavefront
 * kernels rely on behave as assumed (full-wave DPP shift wave_shr:1 with lane 0 keeping its old value;
 * 16-byte global loads/stores at 4-byte alignment).  scratch_dev: >= 8 KiB of device memory; after the
 * stream has drained, ((int*)scratch_dev)[0] == 1 means pass. */
int ftr_selftest(void* scratch_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FTR_H_ */
