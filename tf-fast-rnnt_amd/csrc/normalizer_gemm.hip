// csrc/normalizer_gemm.hip -- the three dense f32 contractions of the simple / smoothed builders that are NOT inside a
// hand-written kernel, as rocBLAS strided-batched GEMMs behind the C ABI, with the library's kernel chosen by measurement.
//
//   kind 0  prod[b]  = lm_probs[b] . am_probs[b]^T     [S1,C] x [C,T]  -> [S1,T]   (rnnt_loss.py:180-182; only where the
//                                                                                  fused forward does not apply)
//   kind 1  dlmp[b]  = W[b] . am_probs[b]              [S1,T] x [T,C]  -> [S1,C]   (autodiff of :180-182 towards lm)
//   kind 2  damp[b]  = W[b]^T . lm_probs[b]            [T,S1] x [S1,C] -> [T,C]    (towards am; where the fused d am kernel
//                                                                                  does not apply)
//
// rocBLAS' default kernel for these shapes runs at 62 - 77 TFLOP/s on MI355X, its best one at ~100 (c3: 90 / 84 us ->
// 62 / 60 us), so the library's candidates (rocblas_gemm_strided_batched_ex_get_solutions) are timed once per shape -- by
// default at the SECOND call with a shape (a ragged training loop whose shapes never repeat is never held up; a loop with
// fixed shapes pays ~0.2 s once) and never inside a stream capture -- and the fastest is used from then on.
// FTR_GEMM_TUNE = off | second (default) | first.  The choice lives in the process (no files).
#define ROCBLAS_BETA_FEATURES_API 1
#define ROCBLAS_NO_DEPRECATED_WARNINGS 1
#include <rocblas/rocblas.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "ftr_common.h"

namespace ftr {
namespace {

struct Shape {
  int ta, tb, m, n, k, lda, ldb, ldc, batch, device;
  long long sa, sb, sc;
  bool operator<(const Shape& o) const {
    return std::tie(ta, tb, m, n, k, lda, ldb, ldc, batch, device, sa, sb, sc) <
           std::tie(o.ta, o.tb, o.m, o.n, o.k, o.lda, o.ldb, o.ldc, o.batch, o.device, o.sa, o.sb, o.sc);
  }
};
struct Choice {
  int seen = 0;        // calls with this shape so far
  bool tuned = false;
  int solution = 0;    // 0 = the library's own choice
  float us = 0.f, us_default = 0.f;
  int candidates = 0;
};

std::mutex g_mu;
std::map<int, rocblas_handle> g_handles;   // one per device
std::map<Shape, Choice> g_choices;

int tune_mode() {   // 0 off, 1 at the first call, 2 at the second
  const char* e = getenv("FTR_GEMM_TUNE");
  if (!e || !strcmp(e, "second") || !strcmp(e, "auto")) return 2;
  if (!strcmp(e, "first")) return 1;
  return 0;
}

rocblas_status run(rocblas_handle h, const Shape& s, const float* A, const float* B, float* C, rocblas_gemm_algo algo, int sol) {
  const float one = 1.0f, zero = 0.0f;
  return rocblas_gemm_strided_batched_ex(h, (rocblas_operation)s.ta, (rocblas_operation)s.tb, s.m, s.n, s.k, &one, A,
                                         rocblas_datatype_f32_r, s.lda, s.sa, B, rocblas_datatype_f32_r, s.ldb, s.sb, &zero, C,
                                         rocblas_datatype_f32_r, s.ldc, s.sc, C, rocblas_datatype_f32_r, s.ldc, s.sc, s.batch,
                                         rocblas_datatype_f32_r, algo, sol, rocblas_gemm_flags_none);
}

// times the default kernel and every candidate on the caller's stream (the output is simply recomputed) and keeps the best
void tune(rocblas_handle h, const Shape& s, const float* A, const float* B, float* C, hipStream_t st, Choice& ch) {
  const float one = 1.0f, zero = 0.0f;
  rocblas_int n = 0;
  auto query = [&](rocblas_int* list, rocblas_int* size) {
    return rocblas_gemm_strided_batched_ex_get_solutions(h, (rocblas_operation)s.ta, (rocblas_operation)s.tb, s.m, s.n, s.k, &one, A,
                                                         rocblas_datatype_f32_r, s.lda, s.sa, B, rocblas_datatype_f32_r, s.ldb, s.sb,
                                                         &zero, C, rocblas_datatype_f32_r, s.ldc, s.sc, C, rocblas_datatype_f32_r,
                                                         s.ldc, s.sc, s.batch, rocblas_datatype_f32_r, rocblas_gemm_algo_solution_index,
                                                         rocblas_gemm_flags_none, list, size);
  };
  ch.tuned = true;   // whatever happens below, do not try again
  if (query(nullptr, &n) != rocblas_status_success || n <= 0) return;
  std::vector<rocblas_int> sols((size_t)n);
  if (query(sols.data(), &n) != rocblas_status_success) return;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) return;
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return; }
  auto time_one = [&](rocblas_gemm_algo algo, int sol, int reps) -> float {
    if (run(h, s, A, B, C, algo, sol) != rocblas_status_success) return -1.f;   // also the warm-up
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i)
      if (run(h, s, A, B, C, algo, sol) != rocblas_status_success) return -1.f;
    (void)hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / reps;
  };
  const float def = time_one(rocblas_gemm_algo_standard, 0, 3);
  // one run of every candidate, then four more of those within 15 % of the fastest single run
  std::vector<std::pair<float, int>> timed;
  float fastest = 1e30f;
  for (int i = 0; i < n; ++i) {
    const float t = time_one(rocblas_gemm_algo_solution_index, sols[(size_t)i], 1);
    if (t > 0.f) { timed.emplace_back(t, sols[(size_t)i]); fastest = t < fastest ? t : fastest; }
  }
  float best = def > 0.f ? def : 1e30f;
  int best_sol = 0;
  for (const auto& p : timed) {
    if (p.first > 1.15f * fastest) continue;
    const float t = time_one(rocblas_gemm_algo_solution_index, p.second, 4);
    if (t > 0.f && t < 0.97f * best) { best = t; best_sol = p.second; }   // a candidate has to beat the default by 3 %
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipGetLastError();
  ch.solution = best_sol;
  ch.us = best < 1e29f ? best : 0.f;
  ch.us_default = def;
  ch.candidates = (int)timed.size();
}

}  // namespace

namespace {
// a row-major product  out = X . Y  is the column-major product  out^T = Y^T . X^T : rocBLAS gets (Y, X) swapped
bool make_shape(int kind, int B, int T, int S1, int C, int dev, Shape& s) {
  if (kind == 0)        // prod[S1,T] = lm_probs[S1,C] . am_probs[T,C]^T
    s = Shape{rocblas_operation_transpose, rocblas_operation_none, T, S1, C, C, C, T, B, dev, (long long)T * C, (long long)S1 * C, (long long)S1 * T};
  else if (kind == 1)   // dlmp[S1,C] = W[S1,T] . am_probs[T,C]
    s = Shape{rocblas_operation_none, rocblas_operation_none, C, S1, T, C, T, C, B, dev, (long long)T * C, (long long)S1 * T, (long long)S1 * C};
  else if (kind == 2)   // damp[T,C] = W[S1,T]^T . lm_probs[S1,C]
    s = Shape{rocblas_operation_none, rocblas_operation_transpose, C, T, S1, C, T, C, B, dev, (long long)S1 * C, (long long)S1 * T, (long long)T * C};
  else
    return false;
  return true;
}
}  // namespace

// row-major operands as the builders hold them; see the table at the top
int normalizer_gemm(int kind, const float* x, const float* y, float* out, int B, int T, int S1, int C, hipStream_t st) {
  if ((size_t)B * T * S1 * C == 0) return FTR_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("normalizer_gemm: hipGetDevice failed"); return FTR_ERR_LAUNCH; }
  Shape s{};
  if (!make_shape(kind, B, T, S1, C, dev, s)) {
    set_error("normalizer_gemm: kind %d is not 0, 1 or 2", kind);
    return FTR_ERR_INVALID_ARG;
  }
  const float *A = y, *Bm = x;   // swapped: x = lm_probs / W / W, y = am_probs / am_probs / lm_probs
  std::lock_guard<std::mutex> lock(g_mu);
  rocblas_handle& h = g_handles[dev];
  if (!h) {
    if (rocblas_create_handle(&h) != rocblas_status_success) { h = nullptr; set_error("normalizer_gemm: rocblas_create_handle failed"); return FTR_ERR_LAUNCH; }
    (void)rocblas_set_pointer_mode(h, rocblas_pointer_mode_host);
  }
  if (rocblas_set_stream(h, st) != rocblas_status_success) { set_error("normalizer_gemm: rocblas_set_stream failed"); return FTR_ERR_LAUNCH; }
  Choice& ch = g_choices[s];
  ++ch.seen;
  const int mode = tune_mode();
  if (!ch.tuned && mode != 0 && ch.seen >= mode) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) tune(h, s, A, Bm, out, st, ch);
  }
  rocblas_status rs = ch.solution ? run(h, s, A, Bm, out, rocblas_gemm_algo_solution_index, ch.solution)
                                  : run(h, s, A, Bm, out, rocblas_gemm_algo_standard, 0);
  if (rs != rocblas_status_success && ch.solution) {   // a stale choice: back to the library's own
    ch.solution = 0;
    rs = run(h, s, A, Bm, out, rocblas_gemm_algo_standard, 0);
  }
  if (rs != rocblas_status_success) { set_error("normalizer_gemm: rocblas_gemm_strided_batched_ex failed with status %d", (int)rs); return FTR_ERR_LAUNCH; }
  return FTR_OK;
}

// what the selection did for a shape: returns 1 and fills the outputs if the shape has been seen, else 0
int normalizer_gemm_choice(int kind, int B, int T, int S1, int C, int* solution, float* us, float* us_default, int* candidates) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  Shape s{};
  if (!make_shape(kind, B, T, S1, C, dev, s)) return 0;
  std::lock_guard<std::mutex> lock(g_mu);
  const auto it = g_choices.find(s);
  if (it == g_choices.end()) return 0;
  if (solution) *solution = it->second.solution;
  if (us) *us = it->second.us;
  if (us_default) *us_default = it->second.us_default;
  if (candidates) *candidates = it->second.tuned ? it->second.candidates : -1;
  return 1;
}

// fixes the kernel for a shape without measuring (a choice stored by an earlier run of the same shape on the same library
// version; an index the library rejects falls back to its own choice at the first call)
int normalizer_gemm_set_choice(int kind, int B, int T, int S1, int C, int solution) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  Shape s{};
  if (!make_shape(kind, B, T, S1, C, dev, s)) { set_error("normalizer_gemm_set_choice: kind %d is not 0, 1 or 2", kind); return FTR_ERR_INVALID_ARG; }
  std::lock_guard<std::mutex> lock(g_mu);
  Choice& ch = g_choices[s];
  ch.tuned = true;
  ch.solution = solution;
  ch.candidates = 0;
  return FTR_OK;
}

}  // namespace ftr
