// csrc/ftr_common.h -- shared by every translation unit of libftr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ftr.h"

namespace ftr {

// error text of the last failing call on this thread (ftr_last_error()).
void set_error(const char* fmt, ...);
void clear_error();

// Checks the launch that was just enqueued (no sync: hipGetLastError only reports launch-time errors).
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return FTR_ERR_LAUNCH;
  }
  return FTR_OK;
}

// Zero fill / device copy of 32-bit words as KERNELS.  Not hipMemsetAsync / hipMemcpyAsync: a memset node captured into a
// hipGraph writes the right value on the first replay only on this ROCm (scripts/graph_memset_probe.py: garbage from the
// second replay on), and every launch of this library must be capturable -- the recursion's hand-off region is cleared by
// such a node whenever the caller does not vouch for a clean workspace.
__global__ void zero_words_kernel(uint32_t* __restrict__ p, size_t n);
__global__ void copy_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n);
inline int zero_words(void* p, size_t n_words, hipStream_t st, const char* what) {
  if (n_words == 0) return FTR_OK;
  const size_t blocks = (n_words + 4 * 256 - 1) / (4 * 256);
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st, static_cast<uint32_t*>(p), n_words);
  return check_launch(what);
}
inline int copy_words(void* dst, const void* src, size_t n_words, hipStream_t st, const char* what) {
  if (n_words == 0) return FTR_OK;
  const size_t blocks = (n_words + 255) / 256;
  hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st, static_cast<uint32_t*>(dst),
                     static_cast<const uint32_t*>(src), n_words);
  return check_launch(what);
}

// A "minus infinity" that stays finite under the additions of the recursion, so the dependent chain
// needs no NaN guard: anything <= NEG_THRESH is reported as -inf when it leaves a kernel.
constexpr float kNeg = -1.0e30f;
constexpr float kNegThresh = -1.0e29f;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// 16-byte vector with 4-byte alignment: lattice rows have odd lengths (T+1), so row starts are only
// dword aligned; amdhsa runs in unaligned-access mode and these lower to global_load/store_dwordx4.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Per-utterance factor applied to incoming lattice gradients on the fly: (p ? p[b * stride] : 1) * mul.
// stride 0 = one scalar for the whole batch (reduction "sum"/"mean"), mul carries the sign and the 1/B of "mean".
struct Scale {
  const float* p; int stride; float mul;
  __device__ __forceinline__ float at(int b) const { return (p ? p[(size_t)b * stride] : 1.0f) * mul; }
};
inline Scale scale_none() { return Scale{nullptr, 0, 1.0f}; }

// Wave64 reductions on the VALU (DPP row shifts + row broadcasts, result read from lane 63) instead of six LDS
// round trips (ds_bpermute): the row kernels do two of these per 2-4 KB row.  Every lane gets the result.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_take(float identity, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, src),
                                                               CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_take<0x111, 0xf, 0xf>(0.0f, v);   // row_shr:1
  v += dpp_take<0x112, 0xf, 0xf>(0.0f, v);   // row_shr:2
  v += dpp_take<0x114, 0xf, 0xe>(0.0f, v);   // row_shr:4
  v += dpp_take<0x118, 0xf, 0xc>(0.0f, v);   // row_shr:8  -> lane 15 of every row holds the row's sum
  v += dpp_take<0x142, 0xa, 0xf>(0.0f, v);   // row_bcast:15 into rows 1 and 3
  v += dpp_take<0x143, 0xc, 0xf>(0.0f, v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  constexpr float ninf = -__builtin_inff();
  v = fmaxf(v, dpp_take<0x111, 0xf, 0xf>(ninf, v));
  v = fmaxf(v, dpp_take<0x112, 0xf, 0xf>(ninf, v));
  v = fmaxf(v, dpp_take<0x114, 0xf, 0xe>(ninf, v));
  v = fmaxf(v, dpp_take<0x118, 0xf, 0xc>(ninf, v));
  v = fmaxf(v, dpp_take<0x142, 0xa, 0xf>(ninf, v));
  v = fmaxf(v, dpp_take<0x143, 0xc, 0xf>(ninf, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- per-utterance shift of the recursion's operands (mi_wave_bidir.hip, mi_band.hip)
// The recursion is invariant under px -> px - cx, py -> py - cy with constants per utterance: every complete path takes
// the same number of px steps (s_end - s_begin) and of py steps (t_end - t_begin; modified: minus the px steps), so both
// incoming terms of every cell move together -- split ratios, occupancies and gradients are unchanged and `ans` moves
// by a known amount, which the cut reduction adds back.  What the shift buys is float32 accuracy: unshifted, p(s,t)
// reaches -5e3 .. -1e5 (log2 units) on a long lattice, one ulp there is 5e-4 .. 8e-3, and the split ratio is formed from
// the DIFFERENCE of two such numbers (the reference's own arithmetic has the same defect: term1 / term2 at
// mutual_information_cuda.cu:642-660 subtract them in the backward pass).  With cx, cy chosen so that a path along the
// rectangle's diagonal gains nothing on average -- cx = mean(px) - ln f, cy = mean(py) - ln(1 - f), f = the fraction of
// px steps -- p stays within tens to hundreds along the paths that carry the occupancy.  The means are estimates (a
// fixed sample, or the band itself): any finite constants are exact in exact arithmetic.
struct Shift { float cx2, cy2; };   // log2 domain
__device__ __forceinline__ bool shift_sample_ok(float v) { return v > -1.0e4f && v < 1.0e4f; }   // not -inf / NaN / a huge stand-in for -inf
template <bool MOD>
__device__ __forceinline__ Shift shift_from_sums(float sumx, float nx, float sumy, float ny, int Sn, int Tn) {
  const float steps = MOD ? (float)(Tn - 1) : (float)(Sn - 1 + Tn - 1);
  float f = steps > 0.0f ? (float)(Sn - 1) / steps : 0.5f;
  f = fminf(fmaxf(f, 1.0e-3f), 1.0f - 1.0e-3f);
  const float mx = nx > 0.0f ? sumx / nx : 0.0f, my = ny > 0.0f ? sumy / ny : 0.0f;
  Shift s;
  s.cx2 = mx * kLog2e - __builtin_amdgcn_logf(f);
  s.cy2 = my * kLog2e - __builtin_amdgcn_logf(1.0f - f);
  return s;
}
// what the shifts took out of `ans` (log2 units): px steps * cx2 + py steps * cy2 of any complete path
template <bool MOD>
__device__ __forceinline__ double shift_total(const Shift s, int Sn, int Tn) {
  const int nx = Sn - 1, ny = MOD ? (Tn - 1) - (Sn - 1) : (Tn - 1);
  return (double)nx * (double)s.cx2 + (double)ny * (double)s.cy2;
}

struct Bound { int sb, tb, se, te; };
__device__ __forceinline__ Bound load_boundary(const int32_t* __restrict__ boundary, int b, int S, int T) {
  Bound r;
  if (boundary) {
    // clamped into the lattice: a malformed row then behaves like its intersection with [0,S] x [0,T] instead of
    // sending a kernel out of bounds (the reference only asserts shapes, mutual_information_cuda.cu:773-785)
    r.sb = max(boundary[4 * b + 0], 0); r.tb = max(boundary[4 * b + 1], 0);
    r.se = min(boundary[4 * b + 2], S); r.te = min(boundary[4 * b + 3], T);
  } else { r.sb = 0; r.tb = 0; r.se = S; r.te = T; }
  return r;
}

}  // namespace ftr

// launchers implemented in the kernel files (all return FTR_OK / FTR_ERR_*), called from capi.hip
namespace ftr {
int mi_plain_fwd(const float* px, const float* py, const int32_t* boundary, float* p, float* ans, int B, int S, int T, int modified, hipStream_t st);
int mi_plain_bwd(const float* px, const float* py, const int32_t* boundary, const float* p, float* p_grad, float* px_grad, float* py_grad, float* ans_grad, int overwrite, int B, int S, int T, int modified, hipStream_t st);
int mi_bidir_fwd(const float* px, const float* py, const int32_t* boundary, float* ws, size_t ws_floats, int flags, float* ans, int B, int S, int T, int modified, hipStream_t st);
int mi_bidir_bwd(const int32_t* boundary, const float* ws, size_t ws_floats, int flags, float* px_grad, float* py_grad, float* ans_grad, int overwrite, int B, int S, int T, int modified, hipStream_t st, const float* ans = nullptr, float* loss_out = nullptr, int loss_code = 0);
int mi_bidir_ws_init(float* ws, size_t ws_floats, int B, int S, int T, hipStream_t st);
int mi_bidir_status(const float* ws, size_t ws_floats, int B, int S, int T, int* status_host, long long* dirty_host, hipStream_t st);
size_t mi_bidir_workspace_floats(int B, int S, int T);
size_t mi_bidir_handoff_floats(int B, int S, int T);
int cummin_i32(const int32_t* in, int32_t* out, int rows, int cols, hipStream_t st);
int prune_ranges(const float* px_grad, const float* py_grad, const int32_t* boundary, int32_t* ranges, int32_t* s_begin, int B, int S, int T, int T1, int r, hipStream_t st);
int do_pruning(const float* am, const float* lm, const int32_t* ranges, float* am_p, float* lm_p, int B, int T, int S1, int C, int r, hipStream_t st);
size_t do_pruning_bwd_workspace_bytes(int B, int T, int S1, int C, int r);
int do_pruning_bwd_ws(const float* g_am_p, const float* g_lm_p, const int32_t* ranges, float* d_am, float* d_lm, int B, int T, int S1, int C, int r, void* ws, size_t ws_bytes, hipStream_t st);
int do_pruning_bwd(const float* g_am_p, const float* g_lm_p, const int32_t* ranges, float* d_am, float* d_lm, int B, int T, int S1, int C, int r, hipStream_t st);
int pruned_logprobs_fwd(const float* logits, const int32_t* symbols, const int32_t* ranges, const int32_t* boundary, int blank, double delay_penalty, float* lse, float* px, float* py, int B, int T, int S, int C, int r, int modified, hipStream_t st);
int pruned_logprobs_bwd(const float* logits, const int32_t* symbols, const int32_t* ranges, const int32_t* boundary, int blank, const float* lse, const float* gpx, const float* gpy, Scale scale, float* glogits, int B, int T, int S, int C, int r, int modified, hipStream_t st);
int simple_rowmax_exp(const float* x, float* probs, float* rowmax, float* rowsum, const float* dotvec, float* dot, size_t rows, int C, hipStream_t st);
int simple_rowmax_exp_pair(const float* x1, float* probs1, float* rowmax1, size_t rows1, const float* x2, float* probs2, float* rowmax2, size_t rows2, int C, hipStream_t st);
int simple_rowdot(const float* x, const float* v, float* dot, size_t rows, int C, hipStream_t st);
size_t simple_colsum_workspace_floats(size_t rows, int C);
int simple_colsum_weighted(const float* x, const float* w, float* out, float* ws, size_t ws_floats, size_t rows, int C, hipStream_t st);
int simple_logprobs_fwd(const float* am, const float* lm, const int32_t* symbols, const float* prod, const float* am_max, const float* lm_max, const int32_t* boundary, int blank, double delay_penalty, const float* lmonly_norm, const float* amonly_norm, const float* ulog, float cs, float ls, float as, float* px, float* py, int B, int T, int S, int C, int modified, hipStream_t st);
int simple_logprobs_bwd_w(const float* gpx, const float* gpy, Scale scale, const float* prod, const int32_t* boundary, float* W, float* rsx, float* rsy, float cs, int B, int T, int S, int modified, hipStream_t st);
int simple_logprobs_bwd_am(const float* gpx, const float* gpy, Scale scale, const float* damp, const float* am_probs, const int32_t* symbols, const int32_t* boundary, int blank, float kdir, const float* uvec, const float* amdot, float as, float* Rout, float* d_am, int B, int T, int S, int C, int modified, hipStream_t st);
int simple_logprobs_bwd_lm(const float* dlmp, const float* lm_probs, const int32_t* symbols, const float* rsx, const float* rsy, int blank, float kdir, const float* arow, const float* invsum, const float* gu, float* d_lm, int B, int S, int C, hipStream_t st);
int simple_fused_supported(int C);
int simple_fused_bwd_supported(int T, int C);
int normalizer_gemm(int kind, const float* x, const float* y, float* out, int B, int T, int S1, int C, hipStream_t st);
int normalizer_gemm_choice(int kind, int B, int T, int S1, int C, int* solution, float* us, float* us_default, int* candidates);
int normalizer_gemm_set_choice(int kind, int B, int T, int S1, int C, int solution);
int simple_fused_fwd(const float* am, const float* lm, const int32_t* symbols, const float* am_probs, const float* lm_probs, const float* am_max, const float* lm_max, const int32_t* boundary, int blank, double delay_penalty, const float* lmonly_norm, const float* amonly_norm, const float* ulog, float cs, float ls, float as, float* px, float* py, float* prod_out, int B, int T, int S, int C, int modified, hipStream_t st);
int simple_fused_bwd_am(const float* gpx, const float* gpy, Scale scale, const float* prod, const float* lm_probs, const float* am_probs, const int32_t* symbols, const int32_t* boundary, int blank, float cs, float kdir, const float* uvec, const float* amdot, float as, float* Rout, float* d_am, int B, int T, int S, int C, int modified, hipStream_t st);
int negated_reduce(const float* ans, int B, int reduction, float* out, hipStream_t st);
int lse_rows(const float* logits, float* lse, size_t rows, int C, hipStream_t st);
int mi_band_supported(int T, int S, int r);
int band_ranges_check(const int32_t* ranges, const int32_t* boundary, int* flags, int B, int T, int r, hipStream_t st);
int band_gather(const float* logits, const int32_t* symbols, const int32_t* ranges, const int32_t* boundary, const float* lse, int blank, double delay_penalty, float* pxb, float* pyb, int B, int T, int S, int C, int r, int modified, hipStream_t st);
size_t mi_band_workspace_floats(int B, int T, int S, int r);
int mi_band_seg_supported(int T, int S, int r);
size_t mi_band_seg_workspace_floats(int B, int T, int S, int r);
int mi_band_seg(const float* pxb, const float* pyb, const int32_t* ranges, const int32_t* boundary, float* ws, size_t ws_floats, float* ans, float* gxb, float* gyb, int B, int T, int S, int r, int modified, hipStream_t st);
int mi_band(const float* pxb, const float* pyb, const int32_t* ranges, const int32_t* boundary, float* ws, size_t ws_floats, float* ans, float* gxb, float* gyb, int B, int T, int S, int r, int modified, hipStream_t st);
int band_grad_banded(const float* logits, const int32_t* symbols, const int32_t* ranges, const int32_t* boundary, int blank, const float* lse, const float* gxb, const float* gyb, Scale scale, float* glogits, int B, int T, int S, int C, int r, int modified, hipStream_t st);
int selftest(hipStream_t st, int* result_dev);
int debug_stamps(unsigned long long* out16);
int debug_trace(unsigned long long* out, int n);
}
