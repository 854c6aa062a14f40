// csrc/mi_plain.hip -- "plain" mutual-information kernels: one workgroup per utterance, one thread
// per lattice row, one barrier per anti-diagonal, the reference's own arithmetic (stored p, LogAdd,
// safe_exp terms, p_grad recursion).  Diagnostic family: it is what the reference algorithm does when
// run on this GPU, kept so the wavefront family can be bisected against it on device.
//
// Arithmetic follows (reference paths relative to /root/reference/tf_fast_rnnt/csrc):
//   LogAdd              mutual_information.h:70-83
//   forward recursion   mutual_information.h:101-126, mutual_information_cuda.cu:291-303,346-351,388-389
//   safe_exp / backward mutual_information_cuda.cu:430-439, 608-660, 692-704, 719-720, 733-758
// The tiling of the reference (32x32 tiles, one launch per tile diagonal) is NOT reproduced.
#include "ftr_common.h"

namespace ftr {
namespace {

__device__ __forceinline__ float logadd_ref(float x, float y) {
  float diff;
  if (x < y) { diff = x - y; x = y; } else { diff = y - x; }
  if (diff - diff != 0) return x;
  return x + log1pf(expf(diff));
}
__device__ __forceinline__ float safe_exp_ref(float x) {
  if (x - x != 0) return 0.0f;
  float a = expf(x);
  if (a - a != 0.0f) return 0.0f;
  return a;
}

// Step j: row r (0-based from s_begin) is at column c = MOD ? j : j - r.  prev[] holds every row's
// value of step j-1: prev[r-1] is p[s-1, t(+off)] and prev[r] is p[s, t-1] in both variants.
template <bool MOD>
__global__ void mi_plain_fwd_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                    const int32_t* __restrict__ boundary, float* __restrict__ p,
                                    float* __restrict__ ans, int S, int T) {
  extern __shared__ float sm[];
  const int b = blockIdx.x;
  const Bound bd = load_boundary(boundary, b, S, T);
  const int T1 = MOD ? T : T + 1;
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  if (Sn <= 0 || Tn <= 0) { if (threadIdx.x == 0) ans[b] = 0.0f; return; }
  float* prev = sm;
  float* cur = sm + (S + 2);
  for (int r = threadIdx.x; r < Sn; r += blockDim.x) prev[r] = -INFINITY;
  __syncthreads();
  const float* pxb = px + (size_t)b * S * T1;
  const float* pyb = py + (size_t)b * (S + 1) * T;
  float* pb = p + (size_t)b * (S + 1) * (T + 1);
  const int nsteps = MOD ? Tn : Tn + Sn - 1;
  for (int j = 0; j < nsteps; ++j) {
    for (int r = threadIdx.x; r < Sn; r += blockDim.x) {
      const int c = MOD ? j : j - r;
      float v = -INFINITY;
      if (c >= 0 && c < Tn) {
        const int s = bd.sb + r, t = bd.tb + c;
        if (r == 0 && c == 0) {
          v = 0.0f;
        } else {
          float a = -INFINITY, d = -INFINITY;
          const int coff = MOD ? c - 1 : c;
          if (r > 0 && coff >= 0) a = prev[r - 1] + pxb[(size_t)(s - 1) * T1 + bd.tb + coff];
          if (c > 0) d = prev[r] + pyb[(size_t)s * T + t - 1];
          v = logadd_ref(a, d);
        }
        pb[(size_t)s * (T + 1) + t] = v;
        if (r == Sn - 1 && c == Tn - 1) ans[b] = v;
      } else if (c >= Tn) {
        v = prev[r];
      }
      cur[r] = v;
    }
    __syncthreads();
    float* tmp = prev; prev = cur; cur = tmp;
  }
}

// Reversed coordinates r = s_end - s, c = t_end - t; step j: c = MOD ? j : j - r.  prev[] holds p_grad
// of step j-1: prev[r-1] is p_grad[s+1, t(+1 if MOD)], prev[r] is p_grad[s, t+1].
template <bool MOD>
__global__ void mi_plain_bwd_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                    const int32_t* __restrict__ boundary, const float* __restrict__ p,
                                    float* __restrict__ p_grad, float* __restrict__ px_grad,
                                    float* __restrict__ py_grad, float* __restrict__ ans_grad,
                                    int overwrite, int S, int T) {
  extern __shared__ float sm[];
  const int b = blockIdx.x;
  const Bound bd = load_boundary(boundary, b, S, T);
  const int T1 = MOD ? T : T + 1;
  const int noff = MOD ? 1 : 0;
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  if (Sn <= 0 || Tn <= 0) return;
  float* prev = sm;
  float* cur = sm + (S + 2);
  for (int r = threadIdx.x; r < Sn; r += blockDim.x) prev[r] = 0.0f;
  __syncthreads();
  const float* pxb = px + (size_t)b * S * T1;
  const float* pyb = py + (size_t)b * (S + 1) * T;
  const float* pb = p + (size_t)b * (S + 1) * (T + 1);
  float* pgb = p_grad ? p_grad + (size_t)b * (S + 1) * (T + 1) : nullptr;
  float* pxg = px_grad + (size_t)b * S * T1;
  float* pyg = py_grad + (size_t)b * (S + 1) * T;
  const float seed = ans_grad[b];
  const int nsteps = MOD ? Tn : Tn + Sn - 1;
  for (int j = 0; j < nsteps; ++j) {
    for (int r = threadIdx.x; r < Sn; r += blockDim.x) {
      const int c = MOD ? j : j - r;
      float g = 0.0f;
      if (c >= 0 && c < Tn) {
        const int s = bd.se - r, t = bd.te - c;
        float p00 = pb[(size_t)s * (T + 1) + t];
        if (p00 < -1.0e+30f) p00 = -1.0e+30f;
        float p10 = 0.0f, p01 = 0.0f, g10 = 0.0f, g01 = 0.0f;
        const bool up_ok = (r > 0) && (MOD ? (c > 0) : true);   // (s+1, t+noff) inside the rectangle
        if (up_ok) {
          p10 = pb[(size_t)(s + 1) * (T + 1) + t + noff];
          if (p10 < -1.0e+30f) p10 = -1.0e+30f;
          g10 = prev[r - 1];
        }
        if (c > 0) {
          p01 = pb[(size_t)s * (T + 1) + t + 1];
          if (p01 < -1.0e+30f) p01 = -1.0e+30f;
          g01 = prev[r];
        }
        float x = -INFINITY, y = -INFINITY;
        if (s < bd.se && t < T1) x = pxb[(size_t)s * T1 + t];
        if (t < bd.te) y = pyb[(size_t)s * T + t];
        const float term1 = safe_exp_ref(p00 + x - p10);
        const float term2 = safe_exp_ref(p00 + y - p01);
        g = (r == 0 && c == 0) ? seed : (g10 * term1 + g01 * term2);
        if (pgb) pgb[(size_t)s * (T + 1) + t] = g;
        if (s < bd.se && t <= bd.te - noff) pxg[(size_t)s * T1 + t] = g10 * term1;
        if (t < bd.te) pyg[(size_t)s * T + t] = g01 * term2;
        if (overwrite && r == Sn - 1 && c == Tn - 1) ans_grad[b] = g;
      } else if (c >= Tn) {
        g = prev[r];
      }
      cur[r] = g;
    }
    __syncthreads();
    float* tmp = prev; prev = cur; cur = tmp;
  }
}

}  // namespace

int mi_plain_fwd(const float* px, const float* py, const int32_t* boundary, float* p, float* ans,
                 int B, int S, int T, int modified, hipStream_t st) {
  const size_t lds = sizeof(float) * 2 * (size_t)(S + 2);
  if (lds > 64 * 1024) { set_error("mi_plain_fwd: S=%d too large for the plain family", S); return FTR_ERR_UNSUPPORTED; }
  const int threads = 256;
  if (modified) hipLaunchKernelGGL(mi_plain_fwd_kernel<true>, dim3(B), dim3(threads), lds, st, px, py, boundary, p, ans, S, T);
  else hipLaunchKernelGGL(mi_plain_fwd_kernel<false>, dim3(B), dim3(threads), lds, st, px, py, boundary, p, ans, S, T);
  return check_launch("mi_plain_fwd");
}

int mi_plain_bwd(const float* px, const float* py, const int32_t* boundary, const float* p,
                 float* p_grad, float* px_grad, float* py_grad, float* ans_grad, int overwrite, int B,
                 int S, int T, int modified, hipStream_t st) {
  const size_t lds = sizeof(float) * 2 * (size_t)(S + 2);
  if (lds > 64 * 1024) { set_error("mi_plain_bwd: S=%d too large for the plain family", S); return FTR_ERR_UNSUPPORTED; }
  const int T1 = modified ? T : T + 1;
  // the reference zero-fills both outputs before the kernel (tf_fast_rnnt_op.cc:93-96)
  if (zero_words(px_grad, (size_t)B * S * T1, st, "mi_plain_bwd") != FTR_OK ||
      zero_words(py_grad, (size_t)B * (S + 1) * T, st, "mi_plain_bwd") != FTR_OK) return FTR_ERR_LAUNCH;
  const int threads = 256;
  if (modified) hipLaunchKernelGGL(mi_plain_bwd_kernel<true>, dim3(B), dim3(threads), lds, st, px, py, boundary, p, p_grad, px_grad, py_grad, ans_grad, overwrite, S, T);
  else hipLaunchKernelGGL(mi_plain_bwd_kernel<false>, dim3(B), dim3(threads), lds, st, px, py, boundary, p, p_grad, px_grad, py_grad, ans_grad, overwrite, S, T);
  return check_launch("mi_plain_bwd");
}

}  // namespace ftr
