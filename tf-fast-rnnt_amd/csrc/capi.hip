// csrc/capi.hip -- the extern "C" surface declared in include/ftr.h: argument validation and error reporting.  No
// allocation, no host synchronisation, no CPU fallback.  Built twice: into libftr_hip.so (the product: the symbols of
// ftr.h and nothing else) and, with -DFTR_DIAG, into the test-only _build/libftr_hip_diag.so, which adds the symbols of
// include/ftr_diag.h: the "plain" kernel family (the reference's arithmetic on the device, mi_plain.hip), its
// process-global switch, and the read-out of the trace / stamp builds.
#include "ftr_common.h"
#ifdef FTR_DIAG
#include "../../include/ftr_diag.h"
#endif
#include <stdlib.h>
#include <string.h>

namespace ftr {
namespace {
thread_local char g_err[512] = {0};
#ifdef FTR_DIAG
int g_mi_impl = -1;  // -1: not yet read from the environment
#endif

int device_ok() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no usable HIP device (%s); this library has no CPU path", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return FTR_ERR_NO_DEVICE;
  }
  return FTR_OK;
}

#ifdef FTR_DIAG
int mi_impl() {
  if (g_mi_impl < 0) {
    const char* e = getenv("FTR_MI_IMPL");
    g_mi_impl = (e && strcmp(e, "plain") == 0) ? 1 : 0;
  }
  return g_mi_impl;
}
#else
constexpr int mi_impl() { return 0; }   // the product library has one kernel family and no switch
#endif
}  // namespace

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
void clear_error() { g_err[0] = 0; }
}  // namespace ftr

using namespace ftr;

#define FTR_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      set_error(__VA_ARGS__);                  \
      return FTR_ERR_INVALID_ARG;              \
    }                                          \
  } while (0)

extern "C" {

int ftr_abi_version(void) { return 133; }
const char* ftr_package_version(void) { return "1.2"; }
const char* ftr_last_error(void) { return g_err; }

#ifdef FTR_DIAG
int ftr_set_mi_impl(int impl) {
  const int prev = mi_impl();
  g_mi_impl = (impl == 1) ? 1 : 0;
  return prev;
}
int ftr_get_mi_impl(void) { return mi_impl(); }
#endif

size_t ftr_mutual_information_workspace_floats(int B, int S, int T) {
  if (B < 0 || S < 0 || T < 0) return 0;
  // the bidirectional wavefront kernels: two ratio lattices, the cut vectors and the hand-off region
  // (mi_wave_bidir.hip); the plain family's p lattice fits in the same buffer
  return mi_bidir_workspace_floats(B, S, T);
}

size_t ftr_mutual_information_handoff_floats(int B, int S, int T) {
  if (B < 0 || S < 0 || T < 0) return 0;
  return mi_bidir_handoff_floats(B, S, T);
}

namespace {
int mi_fwd_common(const char* what, const float* px, const float* py, const int32_t* boundary, float* p, size_t p_floats,
                  int flags, float* ans, int B, int S, int T, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && S >= 0 && T >= 0, "%s: negative size B=%d S=%d T=%d", what, B, S, T);
  FTR_REQUIRE((flags & ~FTR_MI_WS_CLEAN) == 0, "%s: unknown flag bits 0x%x", what, flags);
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(py && p && ans, "%s: null py/p/ans", what);
  FTR_REQUIRE(px || S == 0 || (modified ? T == 0 : false), "%s: null px", what);
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#ifdef FTR_DIAG
  if (mi_impl() == 1) {
    FTR_REQUIRE(p_floats >= (size_t)B * (S + 1) * (T + 1), "%s: workspace too small for the plain family", what);
    return mi_plain_fwd(px, py, boundary, p, ans, B, S, T, modified, st);
  }
#endif
  return mi_bidir_fwd(px, py, boundary, p, p_floats, flags, ans, B, S, T, modified, st);
}

int mi_bwd_common(const char* what, const float* px, const float* py, const int32_t* boundary, const float* p,
                  size_t p_floats, int flags, float* p_grad, float* px_grad, float* py_grad, float* ans_grad,
                  int overwrite_ans_grad, int B, int S, int T, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && S >= 0 && T >= 0, "%s: negative size B=%d S=%d T=%d", what, B, S, T);
  FTR_REQUIRE((flags & ~FTR_MI_WS_CLEAN) == 0, "%s: unknown flag bits 0x%x", what, flags);
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(p && py_grad, "%s: null p/py_grad", what);
  FTR_REQUIRE(ans_grad || mi_impl() == 0, "%s: ans_grad may be NULL (= ones) only with the default kernel family", what);
  FTR_REQUIRE(px_grad || S == 0 || (modified && T == 0), "%s: null px_grad", what);
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#ifdef FTR_DIAG
  if (mi_impl() == 1) {
    FTR_REQUIRE((px || S == 0) && py, "%s: the plain family needs px and py", what);
    FTR_REQUIRE(p_grad, "%s: the plain family needs the p_grad scratch lattice", what);
    return mi_plain_bwd(px, py, boundary, p, p_grad, px_grad, py_grad, ans_grad, overwrite_ans_grad, B, S, T, modified, st);
  }
#endif
  return mi_bidir_bwd(boundary, p, p_floats, flags, px_grad, py_grad, ans_grad, overwrite_ans_grad, B, S, T, modified, st);
}
}  // namespace

int ftr_mutual_information_fwd_f32(const float* px, const float* py, const int32_t* boundary, float* p,
                                   float* ans, int B, int S, int T, int modified, void* stream) {
  return mi_fwd_common("mutual_information_fwd", px, py, boundary, p, (size_t)-1, 0, ans, B, S, T, modified, stream);
}

int ftr_mutual_information_bwd_f32(const float* px, const float* py, const int32_t* boundary,
                                   const float* p, float* p_grad, float* px_grad, float* py_grad,
                                   float* ans_grad, int overwrite_ans_grad, int B, int S, int T,
                                   int modified, void* stream) {
  return mi_bwd_common("mutual_information_bwd", px, py, boundary, p, (size_t)-1, 0, p_grad, px_grad, py_grad, ans_grad,
                       overwrite_ans_grad, B, S, T, modified, stream);
}

int ftr_mutual_information_fwd_ws_f32(const float* px, const float* py, const int32_t* boundary, float* p,
                                      size_t p_floats, int flags, float* ans, int B, int S, int T, int modified,
                                      void* stream) {
  return mi_fwd_common("mutual_information_fwd_ws", px, py, boundary, p, p_floats, flags, ans, B, S, T, modified, stream);
}

int ftr_mutual_information_bwd_ws_f32(const float* px, const float* py, const int32_t* boundary, const float* p,
                                      size_t p_floats, int flags, float* p_grad, float* px_grad, float* py_grad,
                                      float* ans_grad, int overwrite_ans_grad, int B, int S, int T, int modified,
                                      void* stream) {
  return mi_bwd_common("mutual_information_bwd_ws", px, py, boundary, p, p_floats, flags, p_grad, px_grad, py_grad, ans_grad,
                       overwrite_ans_grad, B, S, T, modified, stream);
}

int ftr_mutual_information_bwd_loss_ws_f32(const float* px, const float* py, const int32_t* boundary, const float* p,
                                           size_t p_floats, int flags, float* px_grad, float* py_grad, const float* ans,
                                           int reduction, float* loss_out, int B, int S, int T, int modified, void* stream) {
  clear_error();
  (void)px; (void)py;
  FTR_REQUIRE(B >= 0 && S >= 0 && T >= 0, "mutual_information_bwd_loss_ws: negative size B=%d S=%d T=%d", B, S, T);
  FTR_REQUIRE((flags & ~FTR_MI_WS_CLEAN) == 0, "mutual_information_bwd_loss_ws: unknown flag bits 0x%x", flags);
  FTR_REQUIRE(reduction >= 0 && reduction <= 2, "mutual_information_bwd_loss_ws: reduction %d is not 0 (none), 1 (mean) or 2 (sum)", reduction);
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(p && py_grad && ans && loss_out, "mutual_information_bwd_loss_ws: null p / py_grad / ans / loss_out");
  FTR_REQUIRE(px_grad || S == 0 || (modified && T == 0), "mutual_information_bwd_loss_ws: null px_grad");
  FTR_REQUIRE(mi_impl() == 0, "mutual_information_bwd_loss_ws: only with the default kernel family");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return mi_bidir_bwd(boundary, p, p_floats, flags, px_grad, py_grad, nullptr, 0, B, S, T, modified, reinterpret_cast<hipStream_t>(stream),
                      ans, loss_out, reduction);
}

int ftr_mutual_information_workspace_init(float* p, size_t p_floats, int B, int S, int T, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && S >= 0 && T >= 0, "mutual_information_workspace_init: negative size");
  FTR_REQUIRE(p, "mutual_information_workspace_init: null workspace");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return mi_bidir_ws_init(p, p_floats, B, S, T, reinterpret_cast<hipStream_t>(stream));
}

int ftr_mutual_information_status(const float* p, size_t p_floats, int B, int S, int T, int* status_host,
                                  long long* dirty_words_host, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && S >= 0 && T >= 0, "mutual_information_status: negative size");
  FTR_REQUIRE(p && status_host, "mutual_information_status: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return mi_bidir_status(p, p_floats, B, S, T, status_host, dirty_words_host, reinterpret_cast<hipStream_t>(stream));
}

int ftr_cummin_i32(const int32_t* in, int32_t* out, int rows, int cols, void* stream) {
  clear_error();
  FTR_REQUIRE(rows >= 0 && cols >= 0, "cummin: negative size");
  if (rows == 0 || cols == 0) return FTR_OK;
  FTR_REQUIRE(in && out, "cummin: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return cummin_i32(in, out, rows, cols, reinterpret_cast<hipStream_t>(stream));
}

int ftr_prune_ranges_i32(const float* px_grad, const float* py_grad, const int32_t* boundary,
                         int32_t* ranges, int32_t* s_begin_scratch, int B, int S, int T, int T1,
                         int s_range, int* r_eff_out, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && S >= 1 && T >= 1, "prune_ranges: need S >= 1 and T >= 1 (S=%d T=%d)", S, T);
  FTR_REQUIRE(T1 == T || T1 == T + 1, "prune_ranges: px_grad last dim %d must be T or T+1 (T=%d)", T1, T);
  FTR_REQUIRE(s_range >= 1, "prune_ranges: s_range=%d must be >= 1", s_range);
  const int r = (s_range > S) ? S + 1 : s_range;  // rnnt_loss.py:710-711
  if (r_eff_out) *r_eff_out = r;
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(px_grad && py_grad && boundary && ranges && s_begin_scratch, "prune_ranges: null pointer (boundary is mandatory)");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return prune_ranges(px_grad, py_grad, boundary, ranges, s_begin_scratch, B, S, T, T1, r, reinterpret_cast<hipStream_t>(stream));
}

int ftr_do_pruning_f32(const float* am, const float* lm, const int32_t* ranges, float* am_pruned,
                       float* lm_pruned, int B, int T, int S1, int C, int r, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 0 && S1 >= 1 && C >= 0 && r >= 0, "do_pruning: bad sizes");
  if ((size_t)B * T * r * C == 0) return FTR_OK;
  FTR_REQUIRE(am && lm && ranges && lm_pruned, "do_pruning: null pointer");   // am_pruned may be NULL: gather only
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return do_pruning(am, lm, ranges, am_pruned, lm_pruned, B, T, S1, C, r, reinterpret_cast<hipStream_t>(stream));
}

int ftr_do_pruning_bwd_f32(const float* g_am_pruned, const float* g_lm_pruned, const int32_t* ranges, float* d_am,
                           float* d_lm, int B, int T, int S1, int C, int r, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 0 && S1 >= 1 && C >= 0 && r >= 0, "do_pruning_bwd: bad sizes");
  if ((size_t)B * C == 0) return FTR_OK;
  FTR_REQUIRE(g_am_pruned && g_lm_pruned && ranges && d_am && d_lm, "do_pruning_bwd: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return do_pruning_bwd(g_am_pruned, g_lm_pruned, ranges, d_am, d_lm, B, T, S1, C, r, reinterpret_cast<hipStream_t>(stream));
}

size_t ftr_do_pruning_bwd_workspace_bytes(int B, int T, int S1, int C, int r) {
  if (B < 0 || T < 0 || S1 < 1 || C < 0 || r < 0) return 0;
  return do_pruning_bwd_workspace_bytes(B, T, S1, C, r);
}

int ftr_do_pruning_bwd_ws_f32(const float* g_am_pruned, const float* g_lm_pruned, const int32_t* ranges, float* d_am,
                              float* d_lm, int B, int T, int S1, int C, int r, void* workspace,
                              size_t workspace_bytes, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 0 && S1 >= 1 && C >= 0 && r >= 0, "do_pruning_bwd_ws: bad sizes");
  if ((size_t)B * C == 0) return FTR_OK;
  FTR_REQUIRE(g_am_pruned && g_lm_pruned && ranges && d_am && d_lm, "do_pruning_bwd_ws: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return do_pruning_bwd_ws(g_am_pruned, g_lm_pruned, ranges, d_am, d_lm, B, T, S1, C, r, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

int ftr_pruned_logprobs_fwd_f32(const float* logits, const int32_t* symbols, const int32_t* ranges,
                                const int32_t* boundary, int termination_symbol, double delay_penalty,
                                float* lse, float* px, float* py, int B, int T, int S, int C, int r,
                                int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 1 && C >= 1 && r >= 1, "pruned_logprobs_fwd: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "pruned_logprobs_fwd: termination_symbol %d not in [0,%d)", termination_symbol, C);
  FTR_REQUIRE(r <= S + 1, "pruned_logprobs_fwd: s_range %d > S+1 = %d", r, S + 1);
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(logits && symbols && ranges && lse && px && py, "pruned_logprobs_fwd: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return pruned_logprobs_fwd(logits, symbols, ranges, boundary, termination_symbol, delay_penalty, lse, px, py, B, T, S, C, r, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_pruned_logprobs_bwd_f32(const float* logits, const int32_t* symbols, const int32_t* ranges,
                                const int32_t* boundary, int termination_symbol, const float* lse,
                                const float* gpx, const float* gpy, const float* scale, float* glogits,
                                int B, int T, int S, int C, int r, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 1 && C >= 1 && r >= 1, "pruned_logprobs_bwd: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "pruned_logprobs_bwd: termination_symbol %d not in [0,%d)", termination_symbol, C);
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(logits && symbols && ranges && lse && gpx && gpy && glogits, "pruned_logprobs_bwd: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return pruned_logprobs_bwd(logits, symbols, ranges, boundary, termination_symbol, lse, gpx, gpy, Scale{scale, 1, 1.0f}, glogits, B, T, S, C, r, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_rowmax_exp_f32(const float* x, float* probs, float* rowmax, long long rows, int C, void* stream) {
  clear_error();
  FTR_REQUIRE(rows >= 0 && C >= 0, "rowmax_exp: negative size");
  if (rows == 0 || C == 0) return FTR_OK;
  FTR_REQUIRE(x && probs && rowmax, "rowmax_exp: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_rowmax_exp(x, probs, rowmax, nullptr, nullptr, nullptr, (size_t)rows, C, reinterpret_cast<hipStream_t>(stream));
}

int ftr_rowmax_exp_pair_f32(const float* x1, float* probs1, float* rowmax1, long long rows1, const float* x2, float* probs2,
                            float* rowmax2, long long rows2, int C, void* stream) {
  clear_error();
  FTR_REQUIRE(rows1 >= 0 && rows2 >= 0 && C >= 0, "rowmax_exp_pair: negative size");
  if (rows1 + rows2 == 0 || C == 0) return FTR_OK;
  FTR_REQUIRE((rows1 == 0 || (x1 && probs1 && rowmax1)) && (rows2 == 0 || (x2 && probs2 && rowmax2)), "rowmax_exp_pair: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_rowmax_exp_pair(x1, probs1, rowmax1, (size_t)rows1, x2, probs2, rowmax2, (size_t)rows2, C, reinterpret_cast<hipStream_t>(stream));
}

int ftr_rowmax_exp_sum_f32(const float* x, float* probs, float* rowmax, float* rowsum, long long rows, int C,
                           void* stream) {
  clear_error();
  FTR_REQUIRE(rows >= 0 && C >= 0, "rowmax_exp_sum: negative size");
  if (rows == 0 || C == 0) return FTR_OK;
  FTR_REQUIRE(x && probs && rowmax && rowsum, "rowmax_exp_sum: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_rowmax_exp(x, probs, rowmax, rowsum, nullptr, nullptr, (size_t)rows, C, reinterpret_cast<hipStream_t>(stream));
}

int ftr_rowmax_exp_dot_f32(const float* x, float* probs, float* rowmax, const float* dotvec, float* dot, long long rows,
                           int C, void* stream) {
  clear_error();
  FTR_REQUIRE(rows >= 0 && C >= 0, "rowmax_exp_dot: negative size");
  if (rows == 0 || C == 0) return FTR_OK;
  FTR_REQUIRE(x && probs && rowmax && dotvec && dot, "rowmax_exp_dot: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_rowmax_exp(x, probs, rowmax, nullptr, dotvec, dot, (size_t)rows, C, reinterpret_cast<hipStream_t>(stream));
}

int ftr_rowdot_f32(const float* x, const float* v, float* dot, long long rows, int C, void* stream) {
  clear_error();
  FTR_REQUIRE(rows >= 0 && C >= 0, "rowdot: negative size");
  if (rows == 0) return FTR_OK;
  FTR_REQUIRE(dot && ((x && v) || C == 0), "rowdot: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_rowdot(x, v, dot, (size_t)rows, C, reinterpret_cast<hipStream_t>(stream));
}

size_t ftr_colsum_weighted_workspace_floats(long long rows, int C) {
  return (rows < 0 || C < 0) ? 0 : simple_colsum_workspace_floats((size_t)rows, C);
}

int ftr_colsum_weighted_f32(const float* x, const float* w, float* out, float* workspace, size_t workspace_floats,
                            long long rows, int C, void* stream) {
  clear_error();
  FTR_REQUIRE(rows >= 0 && C >= 0, "colsum_weighted: negative size");
  if (C == 0) return FTR_OK;
  FTR_REQUIRE(out && ((x && w && workspace) || rows == 0), "colsum_weighted: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_colsum_weighted(x, w, out, workspace, workspace_floats, (size_t)rows, C, reinterpret_cast<hipStream_t>(stream));
}

int ftr_simple_logprobs_fwd_f32(const float* am, const float* lm, const int32_t* symbols, const float* prod,
                                const float* am_max, const float* lm_max, const int32_t* boundary,
                                int termination_symbol, double delay_penalty, float* px, float* py, int B, int T,
                                int S, int C, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1, "simple_logprobs_fwd: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "simple_logprobs_fwd: termination_symbol %d not in [0,%d)", termination_symbol, C);
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(am && lm && prod && am_max && lm_max && py && (symbols || S == 0) && (px || S == 0), "simple_logprobs_fwd: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_fwd(am, lm, symbols, prod, am_max, lm_max, boundary, termination_symbol, delay_penalty, nullptr, nullptr, nullptr, 1.0f, 0.0f, 0.0f, px, py, B, T, S, C, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_smoothed_logprobs_fwd_f32(const float* am, const float* lm, const int32_t* symbols, const float* prod,
                                  const float* am_max, const float* lm_max, const float* lmonly_norm,
                                  const float* amonly_norm, const float* unigram_log, const int32_t* boundary,
                                  int termination_symbol, float combined_scale, float lm_only_scale,
                                  float am_only_scale, float* px, float* py, int B, int T, int S, int C,
                                  int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1, "smoothed_logprobs_fwd: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "smoothed_logprobs_fwd: termination_symbol %d not in [0,%d)", termination_symbol, C);
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(am && lm && prod && am_max && lm_max && lmonly_norm && amonly_norm && unigram_log && py && (symbols || S == 0) && (px || S == 0), "smoothed_logprobs_fwd: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_fwd(am, lm, symbols, prod, am_max, lm_max, boundary, termination_symbol, 0.0, lmonly_norm, amonly_norm, unigram_log, combined_scale, lm_only_scale, am_only_scale, px, py, B, T, S, C, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_simple_logprobs_bwd_w_f32(const float* gpx, const float* gpy, const float* prod, const int32_t* boundary,
                                  float* W, float* rsx, float* rsy, int B, int T, int S, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0, "simple_logprobs_bwd_w: bad sizes");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(gpy && prod && W && rsx && rsy && (gpx || S == 0), "simple_logprobs_bwd_w: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_bwd_w(gpx, gpy, scale_none(), prod, boundary, W, rsx, rsy, 1.0f, B, T, S, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_smoothed_logprobs_bwd_w_f32(const float* gpx, const float* gpy, const float* prod, const int32_t* boundary,
                                    float combined_scale, float* W, float* rsx, float* rsy, int B, int T, int S,
                                    int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0, "smoothed_logprobs_bwd_w: bad sizes");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(gpy && prod && W && rsx && rsy && (gpx || S == 0), "smoothed_logprobs_bwd_w: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_bwd_w(gpx, gpy, scale_none(), prod, boundary, W, rsx, rsy, combined_scale, B, T, S, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_simple_logprobs_bwd_am_f32(const float* gpx, const float* gpy, const float* damp, const float* am_probs,
                                   const int32_t* symbols, const int32_t* boundary, int termination_symbol,
                                   float* d_am, int B, int T, int S, int C, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1, "simple_logprobs_bwd_am: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "simple_logprobs_bwd_am: bad termination_symbol");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(gpy && damp && am_probs && d_am && (gpx || S == 0) && (symbols || S == 0), "simple_logprobs_bwd_am: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_bwd_am(gpx, gpy, scale_none(), damp, am_probs, symbols, boundary, termination_symbol, 1.0f, nullptr, nullptr, 0.0f, nullptr, d_am, B, T, S, C, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_smoothed_logprobs_bwd_am_f32(const float* gpx, const float* gpy, const float* damp, const float* am_probs,
                                     const int32_t* symbols, const int32_t* boundary, int termination_symbol,
                                     float direct_scale, const float* unigram, const float* am_dot,
                                     float am_only_scale, float* R, float* d_am, int B, int T, int S, int C,
                                     int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1, "smoothed_logprobs_bwd_am: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "smoothed_logprobs_bwd_am: bad termination_symbol");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(gpy && damp && am_probs && d_am && unigram && am_dot && R && (gpx || S == 0) && (symbols || S == 0), "smoothed_logprobs_bwd_am: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_bwd_am(gpx, gpy, scale_none(), damp, am_probs, symbols, boundary, termination_symbol, direct_scale, unigram, am_dot, am_only_scale, R, d_am, B, T, S, C, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_simple_logprobs_bwd_lm_f32(const float* dlmp, const float* lm_probs, const int32_t* symbols,
                                   const float* rsx, const float* rsy, int termination_symbol, float* d_lm, int B,
                                   int S, int C, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && S >= 0 && C >= 1, "simple_logprobs_bwd_lm: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "simple_logprobs_bwd_lm: bad termination_symbol");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(dlmp && lm_probs && rsx && rsy && d_lm && (symbols || S == 0), "simple_logprobs_bwd_lm: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_bwd_lm(dlmp, lm_probs, symbols, rsx, rsy, termination_symbol, 1.0f, nullptr, nullptr, nullptr, d_lm, B, S, C, reinterpret_cast<hipStream_t>(stream));
}

int ftr_smoothed_logprobs_bwd_lm_f32(const float* dlmp, const float* lm_probs, const int32_t* symbols,
                                     const float* rsx, const float* rsy, int termination_symbol, float direct_scale,
                                     const float* row_term, const float* inv_rowsum, const float* unigram_grad,
                                     float* d_lm, int B, int S, int C, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && S >= 0 && C >= 1, "smoothed_logprobs_bwd_lm: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "smoothed_logprobs_bwd_lm: bad termination_symbol");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(dlmp && lm_probs && rsx && rsy && d_lm && row_term && inv_rowsum && unigram_grad && (symbols || S == 0), "smoothed_logprobs_bwd_lm: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_bwd_lm(dlmp, lm_probs, symbols, rsx, rsy, termination_symbol, direct_scale, row_term, inv_rowsum, unigram_grad, d_lm, B, S, C, reinterpret_cast<hipStream_t>(stream));
}

int ftr_negated_reduce_f32(const float* ans, int B, int reduction, float* out, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 1 && reduction >= 0 && reduction <= 2, "negated_reduce: bad arguments B=%d reduction=%d", B, reduction);
  FTR_REQUIRE(ans && out, "negated_reduce: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return negated_reduce(ans, B, reduction, out, reinterpret_cast<hipStream_t>(stream));
}

int ftr_pruned_logprobs_bwd_scaled_f32(const float* logits, const int32_t* symbols, const int32_t* ranges,
                                       const int32_t* boundary, int termination_symbol, const float* lse,
                                       const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                       float scale_mul, float* glogits, int B, int T, int S, int C, int r,
                                       int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1 && r >= 1, "pruned_logprobs_bwd_scaled: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "pruned_logprobs_bwd_scaled: bad termination_symbol");
  FTR_REQUIRE(scale_stride == 0 || scale_stride == 1, "pruned_logprobs_bwd_scaled: scale_stride must be 0 or 1");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(logits && ranges && lse && gpy && glogits && (symbols || S == 0) && (gpx || S == 0), "pruned_logprobs_bwd_scaled: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return pruned_logprobs_bwd(logits, symbols, ranges, boundary, termination_symbol, lse, gpx, gpy, Scale{scale, scale_stride, scale_mul}, glogits, B, T, S, C, r, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_simple_logprobs_bwd_w_scaled_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                         float scale_mul, const float* prod, const int32_t* boundary, float* W,
                                         float* rsx, float* rsy, int B, int T, int S, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0, "simple_logprobs_bwd_w_scaled: bad sizes");
  FTR_REQUIRE(scale_stride == 0 || scale_stride == 1, "simple_logprobs_bwd_w_scaled: scale_stride must be 0 or 1");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(gpy && prod && W && rsx && rsy && (gpx || S == 0), "simple_logprobs_bwd_w_scaled: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_bwd_w(gpx, gpy, Scale{scale, scale_stride, scale_mul}, prod, boundary, W, rsx, rsy, 1.0f, B, T, S, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_simple_logprobs_bwd_am_scaled_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                          float scale_mul, const float* damp, const float* am_probs,
                                          const int32_t* symbols, const int32_t* boundary, int termination_symbol,
                                          float* d_am, int B, int T, int S, int C, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1, "simple_logprobs_bwd_am_scaled: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "simple_logprobs_bwd_am_scaled: bad termination_symbol");
  FTR_REQUIRE(scale_stride == 0 || scale_stride == 1, "simple_logprobs_bwd_am_scaled: scale_stride must be 0 or 1");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(gpy && damp && am_probs && d_am && (gpx || S == 0) && (symbols || S == 0), "simple_logprobs_bwd_am_scaled: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_bwd_am(gpx, gpy, Scale{scale, scale_stride, scale_mul}, damp, am_probs, symbols, boundary, termination_symbol, 1.0f, nullptr, nullptr, 0.0f, nullptr, d_am, B, T, S, C, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_smoothed_logprobs_fwd_pen_f32(const float* am, const float* lm, const int32_t* symbols, const float* prod,
                                      const float* am_max, const float* lm_max, const float* lmonly_norm,
                                      const float* amonly_norm, const float* unigram_log, const int32_t* boundary,
                                      int termination_symbol, double delay_penalty, float combined_scale,
                                      float lm_only_scale, float am_only_scale, float* px, float* py, int B, int T,
                                      int S, int C, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1, "smoothed_logprobs_fwd_pen: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "smoothed_logprobs_fwd_pen: termination_symbol %d not in [0,%d)", termination_symbol, C);
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(am && lm && prod && am_max && lm_max && lmonly_norm && amonly_norm && unigram_log && py && (symbols || S == 0) && (px || S == 0), "smoothed_logprobs_fwd_pen: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_fwd(am, lm, symbols, prod, am_max, lm_max, boundary, termination_symbol, delay_penalty, lmonly_norm, amonly_norm, unigram_log, combined_scale, lm_only_scale, am_only_scale, px, py, B, T, S, C, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_simple_logprobs_fused_supported(int C) { return simple_fused_supported(C); }
int ftr_simple_logprobs_fused_bwd_supported(int T, int C) { return simple_fused_bwd_supported(T, C); }

int ftr_normalizer_gemm_f32(int kind, const float* x, const float* y, float* out, int B, int T, int S1, int C, void* stream) {
  FTR_REQUIRE(B >= 0 && T >= 0 && S1 >= 0 && C >= 0, "normalizer_gemm: bad sizes");
  if ((size_t)B * T * S1 * C == 0) return FTR_OK;
  FTR_REQUIRE(x && y && out, "normalizer_gemm: null pointer");
  return normalizer_gemm(kind, x, y, out, B, T, S1, C, reinterpret_cast<hipStream_t>(stream));
}

int ftr_normalizer_gemm_choice(int kind, int B, int T, int S1, int C, int* solution, float* us, float* us_default, int* candidates) {
  return normalizer_gemm_choice(kind, B, T, S1, C, solution, us, us_default, candidates);
}

int ftr_normalizer_gemm_set_choice(int kind, int B, int T, int S1, int C, int solution) {
  return normalizer_gemm_set_choice(kind, B, T, S1, C, solution);
}

int ftr_simple_logprobs_fused_fwd_f32(const float* am, const float* lm, const int32_t* symbols, const float* am_probs,
                                      const float* lm_probs, const float* am_max, const float* lm_max,
                                      const int32_t* boundary, int termination_symbol, double delay_penalty, float* px,
                                      float* py, float* prod, int B, int T, int S, int C, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1, "simple_logprobs_fused_fwd: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "simple_logprobs_fused_fwd: termination_symbol %d not in [0,%d)", termination_symbol, C);
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(am && lm && am_probs && lm_probs && am_max && lm_max && py && (symbols || S == 0) && (px || S == 0), "simple_logprobs_fused_fwd: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_fused_fwd(am, lm, symbols, am_probs, lm_probs, am_max, lm_max, boundary, termination_symbol, delay_penalty, nullptr, nullptr, nullptr, 1.0f, 0.0f, 0.0f, px, py, prod, B, T, S, C, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_smoothed_logprobs_fused_fwd_f32(const float* am, const float* lm, const int32_t* symbols, const float* am_probs,
                                        const float* lm_probs, const float* am_max, const float* lm_max,
                                        const float* lmonly_norm, const float* amonly_norm, const float* unigram_log,
                                        const int32_t* boundary, int termination_symbol, double delay_penalty,
                                        float combined_scale, float lm_only_scale, float am_only_scale, float* px,
                                        float* py, float* prod, int B, int T, int S, int C, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1, "smoothed_logprobs_fused_fwd: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "smoothed_logprobs_fused_fwd: termination_symbol %d not in [0,%d)", termination_symbol, C);
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(am && lm && am_probs && lm_probs && am_max && lm_max && lmonly_norm && amonly_norm && unigram_log && py && (symbols || S == 0) && (px || S == 0), "smoothed_logprobs_fused_fwd: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_fused_fwd(am, lm, symbols, am_probs, lm_probs, am_max, lm_max, boundary, termination_symbol, delay_penalty, lmonly_norm, amonly_norm, unigram_log, combined_scale, lm_only_scale, am_only_scale, px, py, prod, B, T, S, C, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_simple_logprobs_fused_bwd_am_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                         float scale_mul, const float* prod, const float* lm_probs, const float* am_probs,
                                         const int32_t* symbols, const int32_t* boundary, int termination_symbol,
                                         float* d_am, int B, int T, int S, int C, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1, "simple_logprobs_fused_bwd_am: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "simple_logprobs_fused_bwd_am: bad termination_symbol");
  FTR_REQUIRE(scale_stride == 0 || scale_stride == 1, "simple_logprobs_fused_bwd_am: scale_stride must be 0 or 1");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(gpy && prod && lm_probs && am_probs && d_am && (gpx || S == 0) && (symbols || S == 0), "simple_logprobs_fused_bwd_am: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_fused_bwd_am(gpx, gpy, Scale{scale, scale_stride, scale_mul}, prod, lm_probs, am_probs, symbols, boundary, termination_symbol, 1.0f, 1.0f, nullptr, nullptr, 0.0f, nullptr, d_am, B, T, S, C, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_smoothed_logprobs_fused_bwd_am_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                           float scale_mul, const float* prod, const float* lm_probs,
                                           const float* am_probs, const int32_t* symbols, const int32_t* boundary,
                                           int termination_symbol, float combined_scale, float direct_scale,
                                           const float* unigram, const float* am_dot, float am_only_scale, float* R,
                                           float* d_am, int B, int T, int S, int C, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1, "smoothed_logprobs_fused_bwd_am: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "smoothed_logprobs_fused_bwd_am: bad termination_symbol");
  FTR_REQUIRE(scale_stride == 0 || scale_stride == 1, "smoothed_logprobs_fused_bwd_am: scale_stride must be 0 or 1");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(gpy && prod && lm_probs && am_probs && d_am && unigram && am_dot && R && (gpx || S == 0) && (symbols || S == 0), "smoothed_logprobs_fused_bwd_am: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_fused_bwd_am(gpx, gpy, Scale{scale, scale_stride, scale_mul}, prod, lm_probs, am_probs, symbols, boundary, termination_symbol, combined_scale, direct_scale, unigram, am_dot, am_only_scale, R, d_am, B, T, S, C, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_smoothed_logprobs_bwd_w_scaled_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                           float scale_mul, const float* prod, const int32_t* boundary,
                                           float combined_scale, float* W, float* rsx, float* rsy, int B, int T, int S,
                                           int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0, "smoothed_logprobs_bwd_w_scaled: bad sizes");
  FTR_REQUIRE(scale_stride == 0 || scale_stride == 1, "smoothed_logprobs_bwd_w_scaled: scale_stride must be 0 or 1");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(gpy && prod && W && rsx && rsy && (gpx || S == 0), "smoothed_logprobs_bwd_w_scaled: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_bwd_w(gpx, gpy, Scale{scale, scale_stride, scale_mul}, prod, boundary, W, rsx, rsy, combined_scale, B, T, S, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_smoothed_logprobs_bwd_am_scaled_f32(const float* gpx, const float* gpy, const float* scale, int scale_stride,
                                            float scale_mul, const float* damp, const float* am_probs,
                                            const int32_t* symbols, const int32_t* boundary, int termination_symbol,
                                            float direct_scale, const float* unigram, const float* am_dot,
                                            float am_only_scale, float* R, float* d_am, int B, int T, int S, int C,
                                            int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1, "smoothed_logprobs_bwd_am_scaled: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "smoothed_logprobs_bwd_am_scaled: bad termination_symbol");
  FTR_REQUIRE(scale_stride == 0 || scale_stride == 1, "smoothed_logprobs_bwd_am_scaled: scale_stride must be 0 or 1");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(gpy && damp && am_probs && d_am && unigram && am_dot && R && (gpx || S == 0) && (symbols || S == 0), "smoothed_logprobs_bwd_am_scaled: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return simple_logprobs_bwd_am(gpx, gpy, Scale{scale, scale_stride, scale_mul}, damp, am_probs, symbols, boundary, termination_symbol, direct_scale, unigram, am_dot, am_only_scale, R, d_am, B, T, S, C, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_mutual_information_band_supported(int T, int S, int r) { return mi_band_supported(T, S, r); }

int ftr_pruned_band_fwd_f32(const float* logits, const int32_t* symbols, const int32_t* ranges, const int32_t* boundary,
                            int termination_symbol, double delay_penalty, float* lse, float* px_band, float* py_band,
                            int B, int T, int S, int C, int r, int modified, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1 && r >= 1, "pruned_band_fwd: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "pruned_band_fwd: termination_symbol %d not in [0,%d)", termination_symbol, C);
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(logits && ranges && lse && px_band && py_band && (symbols || S == 0), "pruned_band_fwd: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // lse and the band gather are two launches: folding the gather into the lse pass (picking the blank / symbol entries out
  // of the registers that hold the row) was built and measured -- 93 - 95 us against 73 + 9 at c3: the extra per-row scalar
  // work (two divisions, the ranges -> symbols dependency) costs the streaming pass more than the second kernel does
  rc = lse_rows(logits, lse, (size_t)B * T * r, C, st);
  if (rc != FTR_OK) return rc;
  return band_gather(logits, symbols, ranges, boundary, lse, termination_symbol, delay_penalty, px_band, py_band, B, T, S, C, r, modified, st);
}

int ftr_band_ranges_check_i32(const int32_t* ranges, const int32_t* boundary, int32_t* flags, int B, int T, int r, void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 0 && r >= 1, "band_ranges_check: bad sizes");
  FTR_REQUIRE(flags && (ranges || B == 0 || T == 0), "band_ranges_check: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return band_ranges_check(ranges, boundary, flags, B, T, r, reinterpret_cast<hipStream_t>(stream));
}

int ftr_mutual_information_band_f32(const float* px_band, const float* py_band, const int32_t* ranges,
                                    const int32_t* boundary, float* ans, float* gx_band, float* gy_band, int B, int T,
                                    int S, int r, int modified, void* stream) {
  return ftr_mutual_information_band_ws_f32(px_band, py_band, ranges, boundary, nullptr, 0, ans, gx_band, gy_band, B, T, S, r, modified, stream);
}

size_t ftr_mutual_information_band_workspace_floats(int B, int T, int S, int r) {
  return (B < 0 || T < 1 || S < 0 || r < 1) ? 0 : mi_band_workspace_floats(B, T, S, r);
}

int ftr_mutual_information_band_ws_f32(const float* px_band, const float* py_band, const int32_t* ranges,
                                       const int32_t* boundary, float* workspace, size_t workspace_floats, float* ans,
                                       float* gx_band, float* gy_band, int B, int T, int S, int r, int modified,
                                       void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && r >= 1, "mutual_information_band: bad sizes");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(px_band && py_band && ranges && ans && gx_band && gy_band, "mutual_information_band: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return mi_band(px_band, py_band, ranges, boundary, workspace, workspace_floats, ans, gx_band, gy_band, B, T, S, r, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_pruned_band_bwd_scaled_f32(const float* logits, const int32_t* symbols, const int32_t* ranges,
                                   const int32_t* boundary, int termination_symbol, const float* lse,
                                   const float* gx_band, const float* gy_band, const float* scale, int scale_stride,
                                   float scale_mul, float* glogits, int B, int T, int S, int C, int r, int modified,
                                   void* stream) {
  clear_error();
  FTR_REQUIRE(B >= 0 && T >= 1 && S >= 0 && C >= 1 && r >= 1, "pruned_band_bwd_scaled: bad sizes");
  FTR_REQUIRE(termination_symbol >= 0 && termination_symbol < C, "pruned_band_bwd_scaled: bad termination_symbol");
  FTR_REQUIRE(scale_stride == 0 || scale_stride == 1, "pruned_band_bwd_scaled: scale_stride must be 0 or 1");
  if (B == 0) return FTR_OK;
  FTR_REQUIRE(logits && ranges && lse && gx_band && gy_band && glogits && (symbols || S == 0), "pruned_band_bwd_scaled: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return band_grad_banded(logits, symbols, ranges, boundary, termination_symbol, lse, gx_band, gy_band, Scale{scale, scale_stride, scale_mul}, glogits, B, T, S, C, r, modified, reinterpret_cast<hipStream_t>(stream));
}

int ftr_selftest(void* scratch_dev, void* stream) {
  clear_error();
  FTR_REQUIRE(scratch_dev, "selftest: need >= 8 KiB of device scratch");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return selftest(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<int*>(scratch_dev));
}

#ifdef FTR_DIAG
int ftr_debug_stamps(unsigned long long* out16) {
  clear_error();
  FTR_REQUIRE(out16, "debug_stamps: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return debug_stamps(out16);
}

int ftr_debug_trace(unsigned long long* out, int n) {
  clear_error();
  FTR_REQUIRE(out || n == 0, "debug_trace: null pointer");
  int rc = device_ok();
  if (rc != FTR_OK) return rc;
  return debug_trace(out, n);
}
#endif

}  // extern "C"
