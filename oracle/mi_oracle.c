/*
 * oracle/mi_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, sequential CPU restatement of the hot path of Samsung/tf-fast-rnnt
 * (reference tree read as text only; no reference source is copied here).
 * It is the checker for tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing in the product package may import, link or call
 * this file.
 *
 * What each function restates (paths relative to /root/reference):
 *   oracle_logadd_*        LogAdd                      tf_fast_rnnt/csrc/mutual_information.h:54-83
 *   oracle_mi_fwd_*        mutual_information_kernel   tf_fast_rnnt/csrc/mutual_information_cuda.cu:174-422
 *                          (recursion: mutual_information.h:101-126; validity guards
 *                          :291-303,316-331; origin :346-347; ans :413-419)
 *   oracle_safe_exp_*      safe_exp                    mutual_information_cuda.cu:430-439
 *   oracle_mi_bwd_*        mutual_information_backward_kernel  :490-760
 *                          (eqs. 3a-c/4a-b :474-481; guards :608-637; terms :654-659;
 *                          seed :692-704; recursion :719-720; outputs :733-758)
 *   oracle_cummin_i32      tensor_kernel_scan_innermost_dim_with_indices / CumminCuda :895-1012
 *   oracle_prune_ranges    get_rnnt_prune_ranges + _adjust_pruning_lower_bound +
 *                          _monotonic_lower_bound      tf_fast_rnnt/python/tf_fast_rnnt/rnnt_loss.py:553-761
 *   oracle_do_pruning      do_rnnt_pruning             rnnt_loss.py:763-812
 *   oracle_pruned_band_*   the per-(b,t,k) arithmetic of get_rnnt_logprobs_pruned  rnnt_loss.py:942-996
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - cummin / monotonic lower bound and roll-by-shifts are pinned against the
 *     reference's own docstring vectors (rnnt_loss.py:561-574, 823-834).
 *   - the floating-point recursion has NO golden vectors anywhere in the
 *     reference (its tests only print): PARITY UNPINNED by reference data for
 *     those outputs.  They are cross-checked instead by brute-force path
 *     enumeration, by an independent float64 autograd DP and by invariants
 *     (tests/test_oracle_mi.py).
 *
 * Canonical orders chosen where the reference leaves them unspecified
 * (TensorFlow GPU reductions): cumsum along S is sequential f32; argmax takes
 * the first maximum; logsumexp is max + log(sequential f32 sum of exp).
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off so no FMA contraction
 * changes the arithmetic from what is written).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define IDX3(a, b_, s_, t_, S_, T_) ((a)[((size_t)(b_) * (size_t)(S_) + (size_t)(s_)) * (size_t)(T_) + (size_t)(t_)])

/* ---- LogAdd (mutual_information.h:54-83) -------------------------------- */
float oracle_logadd_f32(float x, float y) {
  float diff;
  if (x < y) { diff = x - y; x = y; } else { diff = y - x; }
  if (diff - diff != 0) return x;           /* +-inf / nan: return the larger */
  return x + log1pf(expf(diff));
}
double oracle_logadd_f64(double x, double y) {
  double diff;
  if (x < y) { diff = x - y; x = y; } else { diff = y - x; }
  if (diff - diff != 0) return x;
  return x + log1p(exp(diff));
}
/* ---- safe_exp (mutual_information_cuda.cu:430-439) ---------------------- */
float oracle_safe_exp_f32(float x) {
  if (x - x != 0) return 0;
  float a = expf(x);
  if (a - a != 0.0f) return 0;
  return a;
}
double oracle_safe_exp_f64(double x) {
  if (x - x != 0) return 0;
  double a = exp(x);
  if (a - a != 0.0) return 0;
  return a;
}

static void get_boundary(const int32_t* boundary, int b, int S, int T, int* sb, int* tb, int* se, int* te) {
  if (boundary) {
    *sb = boundary[4 * b + 0]; *tb = boundary[4 * b + 1];
    *se = boundary[4 * b + 2]; *te = boundary[4 * b + 3];
  } else { *sb = 0; *tb = 0; *se = S; *te = T; }
}

/*
 * Forward.  px [B,S,T1] with T1 = T+1 (regular) or T (modified); py [B,S+1,T];
 * p [B,S+1,T+1] (only cells inside the boundary rectangle are written, as in
 * the reference); ans [B].  Returns 1 like the reference launcher (.cu:810).
 */
#define DEFINE_MI_FWD(NAME, REAL, LOGADD)                                                      \
  int NAME(const REAL* px, const REAL* py, const int32_t* boundary, REAL* p, REAL* ans, int B, \
           int S, int T, int modified) {                                                       \
    const int T1 = modified ? T : T + 1;                                                       \
    const int toff = modified ? -1 : 0;                                                        \
    for (int b = 0; b < B; ++b) {                                                              \
      int sb, tb, se, te;                                                                      \
      get_boundary(boundary, b, S, T, &sb, &tb, &se, &te);                                     \
      for (int s = sb; s <= se; ++s) {                                                         \
        for (int t = tb; t <= te; ++t) {                                                       \
          REAL v;                                                                              \
          if (s == sb && t == tb) {                                                            \
            v = 0;                                                                             \
          } else {                                                                             \
            REAL a = -INFINITY, c = -INFINITY;                                                 \
            const int t_off = t + toff;                                                        \
            if (s > sb && t_off >= tb) /* px valid: .cu:295; p context: .cu:317 */             \
              a = IDX3(p, b, s - 1, t_off, S + 1, T + 1) + IDX3(px, b, s - 1, t_off, S, T1);   \
            if (t > tb) /* py valid: .cu:301 */                                                \
              c = IDX3(p, b, s, t - 1, S + 1, T + 1) + IDX3(py, b, s, t - 1, S + 1, T);        \
            v = LOGADD(a, c);                                                                  \
          }                                                                                    \
          IDX3(p, b, s, t, S + 1, T + 1) = v;                                                  \
        }                                                                                      \
      }                                                                                        \
      ans[b] = (se >= sb && te >= tb) ? IDX3(p, b, se, te, S + 1, T + 1) : (REAL)0;            \
    }                                                                                          \
    return 1;                                                                                  \
  }
DEFINE_MI_FWD(oracle_mi_fwd_f32, float, oracle_logadd_f32)
DEFINE_MI_FWD(oracle_mi_fwd_f64, double, oracle_logadd_f64)

/*
 * Backward, in the reference's own arithmetic: term1/term2 from the stored p
 * (clamped at -1e30), the p_grad recursion (3a) and outputs (3b),(3c).
 * px_grad has the shape of px ([B,S,T1]), py_grad of py; both must be
 * zero-filled by the caller (the reference memsets them,
 * tf_fast_rnnt_op.cc:93-96) -- only in-boundary cells are written here.
 * p_grad [B,S+1,T+1] scratch (in-boundary cells written).
 * If overwrite_ans_grad, ans_grad[b] := p_grad[b,s_begin,t_begin] (.cu:756-758).
 */
#define DEFINE_MI_BWD(NAME, REAL, SAFE_EXP)                                                       \
  int NAME(const REAL* px, const REAL* py, const int32_t* boundary, const REAL* p, REAL* p_grad,  \
           REAL* px_grad, REAL* py_grad, REAL* ans_grad, int overwrite_ans_grad, int B, int S,    \
           int T, int modified) {                                                                 \
    const int T1 = modified ? T : T + 1;                                                          \
    const int noff = modified ? 1 : 0;                                                            \
    for (int b = 0; b < B; ++b) {                                                                 \
      int sb, tb, se, te;                                                                         \
      get_boundary(boundary, b, S, T, &sb, &tb, &se, &te);                                        \
      if (se < sb || te < tb) continue;                                                           \
      for (int s = se; s >= sb; --s) {                                                            \
        for (int t = te; t >= tb; --t) {                                                          \
          /* p with out-of-range := 0 and clamp at -1e30 (.cu:629-637) */                         \
          REAL p00 = IDX3(p, b, s, t, S + 1, T + 1);                                              \
          if (p00 < (REAL)-1.0e+30) p00 = (REAL)-1.0e+30;                                         \
          REAL p10 = 0, p01 = 0;                                                                  \
          if (s + 1 <= se && t + noff <= te) {                                                    \
            p10 = IDX3(p, b, s + 1, t + noff, S + 1, T + 1);                                      \
            if (p10 < (REAL)-1.0e+30) p10 = (REAL)-1.0e+30;                                       \
          }                                                                                       \
          if (t + 1 <= te) {                                                                      \
            p01 = IDX3(p, b, s, t + 1, S + 1, T + 1);                                             \
            if (p01 < (REAL)-1.0e+30) p01 = (REAL)-1.0e+30;                                       \
          }                                                                                       \
          /* px/py with out-of-range := -inf (.cu:608-615) */                                     \
          REAL x = -INFINITY, y = -INFINITY;                                                      \
          if (s < se && t <= te && t < T1) x = IDX3(px, b, s, t, S, T1);                          \
          if (t < te) y = IDX3(py, b, s, t, S + 1, T);                                            \
          const REAL term1 = SAFE_EXP(p00 + x - p10); /* (4a) .cu:654-655 */                      \
          const REAL term2 = SAFE_EXP(p00 + y - p01); /* (4b) .cu:659 */                          \
          /* p_grad context out of range := 0 (.cu:670-683) */                                    \
          REAL g10 = 0, g01 = 0;                                                                  \
          if (s + 1 <= se && t + noff <= te) g10 = IDX3(p_grad, b, s + 1, t + noff, S + 1, T + 1);\
          if (t + 1 <= te) g01 = IDX3(p_grad, b, s, t + 1, S + 1, T + 1);                         \
          REAL g;                                                                                 \
          if (s == se && t == te) g = ans_grad[b]; /* .cu:702 */                                  \
          else g = g10 * term1 + g01 * term2;       /* (3a) .cu:719-720 */                        \
          IDX3(p_grad, b, s, t, S + 1, T + 1) = g;                                                \
          if (s < se && t <= te - noff) IDX3(px_grad, b, s, t, S, T1) = g10 * term1; /* (3b) */   \
          if (t < te) IDX3(py_grad, b, s, t, S + 1, T) = g01 * term2;                /* (3c) */   \
        }                                                                                         \
      }                                                                                           \
      if (overwrite_ans_grad) ans_grad[b] = IDX3(p_grad, b, sb, tb, S + 1, T + 1);                \
    }                                                                                             \
    return 1;                                                                                     \
  }
DEFINE_MI_BWD(oracle_mi_bwd_f32, float, oracle_safe_exp_f32)
DEFINE_MI_BWD(oracle_mi_bwd_f64, double, oracle_safe_exp_f64)

/* ---- cummin: inclusive prefix-min per row (.cu:895-1012) ----------------- */
int oracle_cummin_i32(const int32_t* in, int32_t* out, int rows, int cols) {
  for (int r = 0; r < rows; ++r) {
    int32_t m = INT32_MAX; /* init = numeric_limits::max(), .cu:1001 */
    for (int c = 0; c < cols; ++c) {
      int32_t v = in[(size_t)r * cols + c];
      if (m > v) m = v; /* binary_op_update: if (rhs > lhs) rhs = lhs, .cu:876-882 */
      out[(size_t)r * cols + c] = m;
    }
  }
  return 1;
}

/* _monotonic_lower_bound (rnnt_loss.py:553-585): reverse -> cummin -> reverse = suffix-min. */
static void monotonic_lower_bound_row(int32_t* x, int n) {
  int32_t m = INT32_MAX;
  for (int i = n - 1; i >= 0; --i) { if (m > x[i]) m = x[i]; x[i] = m; }
}
int oracle_monotonic_lower_bound_i32(const int32_t* in, int32_t* out, int rows, int cols) {
  memcpy(out, in, sizeof(int32_t) * (size_t)rows * cols);
  for (int r = 0; r < rows; ++r) monotonic_lower_bound_row(out + (size_t)r * cols, cols);
  return 1;
}
/* _adjust_pruning_lower_bound (rnnt_loss.py:587-641), in place on [rows, T]. */
int oracle_adjust_pruning_lower_bound_i32(int32_t* s_begin, int rows, int T, int s_range) {
  for (int r = 0; r < rows; ++r) {
    int32_t* x = s_begin + (size_t)r * T;
    monotonic_lower_bound_row(x, T);                                   /* :628 */
    for (int t = 0; t < T; ++t) x[t] = -(x[t] - (s_range - 1) * t);    /* :630-632 */
    monotonic_lower_bound_row(x, T);                                   /* :634 */
    for (int t = 0; t < T; ++t) if (x[t] < 0) x[t] = 0;                /* :636 */
    for (int t = 0; t < T; ++t) x[t] = -(x[t] - (s_range - 1) * t);    /* :638-640 */
  }
  return 1;
}

/*
 * get_rnnt_prune_ranges (rnnt_loss.py:647-761).
 * px_grad [B,S,T1], py_grad [B,S+1,T], boundary [B,4] (mandatory here as in
 * the reference), s_range as passed by the caller.  ranges must have room for
 * [B,T,r_eff] where r_eff = (s_range > S ? S+1 : s_range) (:710-711); r_eff is
 * returned.  s_begin_raw (optional, [B,T]) receives the argmax before any
 * adjustment, for diagnostics.
 */
int oracle_prune_ranges(const float* px_grad, const float* py_grad, const int32_t* boundary,
                        int32_t* ranges, int32_t* s_begin_raw, int B, int S, int T, int T1,
                        int s_range) {
  const int S1 = S + 1;
  if (s_range > S) s_range = S + 1;
  const int r = s_range;
  const int nwin = S1 - r + 1;
  float* cum = (float*)malloc(sizeof(float) * (size_t)(S1 + 1));
  int32_t* sbeg = (int32_t*)malloc(sizeof(int32_t) * (size_t)B * T);
  for (int b = 0; b < B; ++b) {
    for (int t = 0; t < T; ++t) {
      /* cumsum along S with a leading zero (:722-724), sequential f32 */
      cum[0] = 0.0f;
      float acc = 0.0f;
      for (int s = 0; s < S1; ++s) { acc = acc + IDX3(py_grad, b, s, t, S1, T); cum[s + 1] = acc; }
      int best = 0; float bestv = 0;
      for (int s0 = 0; s0 < nwin; ++s0) {
        float blk = cum[s0 + r] - cum[s0];                                     /* :725 */
        float pxp = (s0 == 0) ? 0.0f : IDX3(px_grad, b, s0 - 1, t, S, T1);     /* :726-727 */
        float fin = blk - pxp;                                                 /* :728 */
        if (s0 == 0 || fin > bestv) { best = s0; bestv = fin; }                /* :729 first max */
      }
      if (s_begin_raw) s_begin_raw[(size_t)b * T + t] = best;
      /* padding frames (:741-748) */
      int32_t pad = boundary[4 * b + 2] - r + 1;
      if (pad < 0) pad = 0;
      sbeg[(size_t)b * T + t] = (t < boundary[4 * b + 3] - 1) ? best : pad;
    }
  }
  oracle_adjust_pruning_lower_bound_i32(sbeg, B, T, (T1 == T) ? 2 : r);         /* :756 */
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < r; ++k)
        ranges[((size_t)b * T + t) * r + k] = sbeg[(size_t)b * T + t] + k;      /* :758-759 */
  free(cum); free(sbeg);
  return r;
}

/* do_rnnt_pruning (rnnt_loss.py:763-812): am [B,T,C], lm [B,S+1,C], ranges [B,T,r]. */
int oracle_do_pruning(const float* am, const float* lm, const int32_t* ranges, float* am_pruned,
                      float* lm_pruned, int B, int T, int S1, int C, int r) {
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < r; ++k) {
        const int s = ranges[((size_t)b * T + t) * r + k];
        float* ao = am_pruned + (((size_t)b * T + t) * r + k) * C;
        float* lo = lm_pruned + (((size_t)b * T + t) * r + k) * C;
        memcpy(ao, am + ((size_t)b * T + t) * C, sizeof(float) * C);
        memcpy(lo, lm + ((size_t)b * S1 + s) * C, sizeof(float) * C);
      }
  return 1;
}

/*
 * Band arithmetic of get_rnnt_logprobs_pruned (rnnt_loss.py:942-965, 995-996):
 * lse[b,t,k] = logsumexp_c logits[b,t,k,:]; px_band = logits[..., sym] - lse with
 * sym = concat(symbols, blank)[b, ranges[b,t,k]]; py_band = logits[..., blank] - lse.
 */
int oracle_pruned_band_fwd(const float* logits, const int32_t* symbols, const int32_t* ranges,
                           int termination_symbol, float* lse, float* px_band, float* py_band,
                           int B, int T, int S, int C, int r) {
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < r; ++k) {
        const size_t row = ((size_t)b * T + t) * r + k;
        const float* x = logits + row * C;
        float m = x[0];
        for (int c = 1; c < C; ++c) if (x[c] > m) m = x[c];
        float sum = 0.0f;
        for (int c = 0; c < C; ++c) sum = sum + expf(x[c] - m);
        const float l = m + logf(sum);
        const int s = ranges[row];
        const int sym = (s < S) ? symbols[(size_t)b * S + s] : termination_symbol;
        lse[row] = l;
        px_band[row] = x[sym] - l;
        py_band[row] = x[termination_symbol] - l;
      }
  return 1;
}

/*
 * Gradient of the band values w.r.t. logits (what TF autodiff of
 * rnnt_loss.py:942-996 produces): d/dlogits[c] = gx*(1[c==sym]-softmax) + gy*(1[c==blank]-softmax).
 */
int oracle_pruned_band_bwd(const float* logits, const int32_t* symbols, const int32_t* ranges,
                           int termination_symbol, const float* lse, const float* gpx_band,
                           const float* gpy_band, float* glogits, int B, int T, int S, int C,
                           int r) {
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < r; ++k) {
        const size_t row = ((size_t)b * T + t) * r + k;
        const float* x = logits + row * C;
        float* g = glogits + row * C;
        const float gx = gpx_band[row], gy = gpy_band[row], l = lse[row];
        const int s = ranges[row];
        const int sym = (s < S) ? symbols[(size_t)b * S + s] : termination_symbol;
        const float tot = gx + gy;
        for (int c = 0; c < C; ++c) g[c] = -tot * expf(x[c] - l);
        g[sym] += gx;
        g[termination_symbol] += gy;
      }
  return 1;
}

/* Timing helper for bench.py's cpu_baseline leg: fwd+bwd over the batch with
 * OpenMP over utterances when built with -fopenmp (utterances are independent,
 * mutual_information_cuda.cu:247-248). */
int oracle_mi_fwd_bwd_f32_mt(const float* px, const float* py, const int32_t* boundary, float* p,
                             float* p_grad, float* px_grad, float* py_grad, float* ans,
                             float* ans_grad, int B, int S, int T, int modified) {
  const int T1 = modified ? T : T + 1;
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < B; ++b) {
    const int32_t* bd = boundary ? boundary + 4 * (size_t)b : NULL;
    int32_t full[4] = {0, 0, S, T};
    if (!bd) bd = full;
    oracle_mi_fwd_f32(px + (size_t)b * S * T1, py + (size_t)b * (S + 1) * T, bd,
                      p + (size_t)b * (S + 1) * (T + 1), ans + b, 1, S, T, modified);
    ans_grad[b] = 1.0f;
    oracle_mi_bwd_f32(px + (size_t)b * S * T1, py + (size_t)b * (S + 1) * T, bd,
                      p + (size_t)b * (S + 1) * (T + 1), p_grad + (size_t)b * (S + 1) * (T + 1),
                      px_grad + (size_t)b * S * T1, py_grad + (size_t)b * (S + 1) * T, ans_grad + b,
                      1, 1, S, T, modified);
  }
  return 1;
}
