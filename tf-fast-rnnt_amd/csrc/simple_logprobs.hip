// csrc/simple_logprobs.hip -- px/py builder of rnnt_loss_simple (joiner = addition), forward and backward,
// gfx950.  Replaces the ~25 un-fused TensorFlow ops of get_rnnt_logprobs
// (/root/reference/tf_fast_rnnt/python/tf_fast_rnnt/rnnt_loss.py:175-221), fix_for_boundary (:28-61), the
// delay-penalty block (:305-321) and what TF autodiff replays for them.  The one dense contraction
// (normalisers = lm_probs @ am_probs^T, and its two transposes in the backward) stays a library GEMM on the
// host side ("may stay in the framework's BLAS"); everything around it is here:
//
//   rowmax_exp_kernel      :175-178   row max + exp(x - max), one wave per row, 16-byte accesses
//   simple_fwd_kernel      :180-221, 305-321   log + max add-back, symbol / blank gathers, -inf column,
//                          fix_for_boundary, penalty; px and py written once, coalesced along t.  am rows of a
//                          32-frame tile are staged in LDS (row stride C+1: conflict-free column gathers), so
//                          am is read from HBM exactly once, coalesced (TF transposes am through HBM for this).
//   simple_bwd_w_kernel    W = -(gpx + gpy) / (prod + tiny)  and the row sums that feed d lm
//   simple_bwd_am_kernel   d am = (W^T lm_probs) * am_probs + scatter of gpx by symbol + blank column sums;
//                          the scatter-add over symbols happens in an LDS tile owned by the workgroup
//                          (each accumulator cell is owned by exactly one thread: deterministic, no atomics)
//   simple_bwd_lm_kernel   d lm = (W am_probs) * lm_probs + row sums at the symbol / blank columns
#include "ftr_common.h"

namespace ftr {
namespace {

constexpr float kTiny = 1.401298464324817e-45f;  // tf.math.nextafter(0., 1.)  (rnnt_loss.py:181)
#ifndef FTR_TT_NARROW_ABOVE
#define FTR_TT_NARROW_ABOVE 300
#endif
constexpr int kTTwide = 32;                      // frames per tile for small vocabularies
constexpr int kTTnarrow = 16;                    // above FTR_TT_NARROW_ABOVE columns: more workgroups per CU (LDS) beats longer
                                                 // row segments (measured at C = 500: fwd 60 -> 47 us, bwd_am 73 -> 65 us; C = 1024: 374 -> 242, 546 -> 420)

__device__ __forceinline__ float wave_max(float v) { return wave_max_dpp(v); }
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }

// probs[row, :] = exp(x[row, :] - max(x[row, :])), rowmax[row] = max; optionally rowsum[row] = sum of the row of probs
// and dot[row] = probs[row, :] . dotvec (the am-only normaliser of the smoothed builder, rnnt_loss.py:1281-1286, taken
// while the row is in registers instead of a second pass over the [B*T, C] matrix).  One wave per row.
template <bool VEC>
__device__ __forceinline__ void rowmax_exp_row(const float* __restrict__ x, float* __restrict__ probs,
                                               float* __restrict__ rowmax, float* __restrict__ rowsum,
                                               const float* __restrict__ dotvec, float* __restrict__ dot, size_t row, int C, int lane) {
  const float* xr = x + row * C;
  float* pr = probs + row * C;
  float m = -INFINITY;
  float sum = 0.0f, dsum = 0.0f;
  if (VEC) {
    const f4u* x4 = reinterpret_cast<const f4u*>(xr);
    f4u* p4 = reinterpret_cast<f4u*>(pr);
    const f4u* d4 = reinterpret_cast<const f4u*>(dotvec);
    const int n4 = C >> 2;
    for (int i = lane; i < n4; i += 64) { const f4 v = x4[i]; m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3])); }
    m = wave_max(m);
    for (int i = lane; i < n4; i += 64) {
      const f4 v = x4[i];
      f4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = expf(v[e] - m); sum += o[e]; }
      if (dotvec) { const f4 dv = d4[i]; dsum += (o[0] * dv[0] + o[1] * dv[1]) + (o[2] * dv[2] + o[3] * dv[3]); }
      p4[i] = o;
    }
  } else {
    for (int i = lane; i < C; i += 64) m = fmaxf(m, xr[i]);
    m = wave_max(m);
    for (int i = lane; i < C; i += 64) { const float o = expf(xr[i] - m); pr[i] = o; sum += o; if (dotvec) dsum += o * dotvec[i]; }
  }
  if (rowsum) { sum = wave_sum(sum); if (lane == 0) rowsum[row] = sum; }
  if (dotvec) { dsum = wave_sum(dsum); if (lane == 0) dot[row] = dsum; }
  if (lane == 0) rowmax[row] = m;
}

template <bool VEC>
__global__ void rowmax_exp_kernel(const float* __restrict__ x, float* __restrict__ probs,
                                  float* __restrict__ rowmax, float* __restrict__ rowsum,
                                  const float* __restrict__ dotvec, float* __restrict__ dot, size_t rows, int C) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  rowmax_exp_row<VEC>(x, probs, rowmax, rowsum, dotvec, dot, row, C, lane);
}

// the same for TWO matrices of C columns in one launch (am [B*T, C] and lm [B*(S+1), C] of the simple builder: the second is
// a 3-us kernel of its own otherwise, and every kernel boundary of the step costs 1.7 us of idle time on top)
template <bool VEC>
__global__ void rowmax_exp_pair_kernel(const float* __restrict__ x1, float* __restrict__ probs1, float* __restrict__ rowmax1, size_t rows1,
                                       const float* __restrict__ x2, float* __restrict__ probs2, float* __restrict__ rowmax2, size_t rows2, int C) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row < rows1) rowmax_exp_row<VEC>(x1, probs1, rowmax1, nullptr, nullptr, nullptr, row, C, lane);
  else if (row - rows1 < rows2) rowmax_exp_row<VEC>(x2, probs2, rowmax2, nullptr, nullptr, nullptr, row - rows1, C, lane);
}

// dot[row] = x[row, :] . v.  One wave per row.
template <bool VEC>
__global__ void rowdot_kernel(const float* __restrict__ x, const float* __restrict__ v, float* __restrict__ dot, size_t rows, int C) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  float acc = 0.0f;
  if (VEC) {
    const f4u* x4 = reinterpret_cast<const f4u*>(xr);
    const f4u* v4 = reinterpret_cast<const f4u*>(v);
    for (int i = lane; i < (C >> 2); i += 64) { const f4 a = x4[i], b = v4[i]; acc += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]); }
  } else {
    for (int i = lane; i < C; i += 64) acc += xr[i] * v[i];
  }
  acc = wave_sum(acc);
  if (lane == 0) dot[row] = acc;
}

// out[c] = sum_row w[row] * x[row, c] in two deterministic stages (fixed summation order, no atomics): each workgroup
// folds a slab of kColsumSlab rows for a strip of columns (one 16-byte piece per thread and row: a row of a strip is one
// contiguous 4 KB read), a second launch adds the slabs up in order.  (rocBLAS' gemv for this shape -- [64000 x 1024]^T
// times a vector -- takes 1.1 ms on MI355X; this takes the time of reading the matrix once.)
constexpr int kColsumSlab = 64;
template <bool VEC>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ partial, size_t rows, int C) {
  const size_t r0 = (size_t)blockIdx.x * kColsumSlab;
  const size_t r1 = r0 + kColsumSlab < rows ? r0 + kColsumSlab : rows;
  if (VEC) {
    const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    size_t r = r0;
    for (; r + 16 <= r1; r += 16) {     // sixteen independent 16-byte loads in flight per lane
      f4 v[16]; float ww[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { v[u] = *reinterpret_cast<const f4u*>(x + (r + u) * C + c); ww[u] = w[r + u]; }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u] * ww[u];
    }
    for (; r < r1; ++r) acc += *reinterpret_cast<const f4u*>(x + r * C + c) * w[r];
    *reinterpret_cast<f4u*>(partial + (size_t)blockIdx.x * C + c) = acc;
  } else {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    float acc = 0.0f;
    for (size_t r = r0; r < r1; ++r) acc += x[r * C + c] * w[r];
    partial[(size_t)blockIdx.x * C + c] = acc;
  }
}
// 64 columns x 4 slab groups per workgroup; the four partial sums of a column are added in a fixed order
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nslab, int C) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, gq = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int per = (nslab + 3) / 4;
  const int i0 = gq * per, i1 = min(i0 + per, nslab);
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  if (c < C) {
    int i = i0;
    for (; i + 4 <= i1; i += 4) {
      a0 += partial[(size_t)i * C + c]; a1 += partial[(size_t)(i + 1) * C + c];
      a2 += partial[(size_t)(i + 2) * C + c]; a3 += partial[(size_t)(i + 3) * C + c];
    }
    for (; i < i1; ++i) a0 += partial[(size_t)i * C + c];
  }
  part[gq][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (gq == 0 && c < C) out[c] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// grid (ceil(T1 / TT), B); block 256 = (256/TT) row-groups x TT frames.  LDS: am tile [TT][C+1].
template <bool MOD, int TT>
__global__ void simple_fwd_kernel(const float* __restrict__ am, const float* __restrict__ lm,
                                  const int32_t* __restrict__ symbols, const float* __restrict__ prod,
                                  const float* __restrict__ am_max, const float* __restrict__ lm_max,
                                  const int32_t* __restrict__ boundary, int blank, double delay_penalty,
                                  const float* __restrict__ lmonly_norm, const float* __restrict__ amonly_norm,
                                  const float* __restrict__ ulog, float cs, float ls, float as,
                                  float* __restrict__ px, float* __restrict__ py, int T, int S, int C) {
  // With lmonly_norm != NULL this is get_rnnt_logprobs_smoothed (rnnt_loss.py:1296-1365):
  //   out = cs * (am + lm - normalizers) + ls * (lm - lmonly_norm[b,s]) + as * (am + ulog[c] - amonly_norm[b,t])
  extern __shared__ float tile[];  // [TT][C + 1]
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TT;
  const int T1 = MOD ? T : T + 1;
  const int ld = C + 1;
  const float* amb = am + (size_t)b * T * C;
  // stage TT frames of am (zeros past T): two rows per pass, 16 bytes per lane along c where C allows
  {
    const int half = threadIdx.x >> 7, cl = threadIdx.x & 127;
    if ((C & 3) == 0) {
      const int n4 = C >> 2;
      for (int tt = half; tt < TT; tt += 2) {
        const bool ok = t0 + tt < T;
        const f4u* src = reinterpret_cast<const f4u*>(amb + (size_t)(t0 + tt) * C);
        for (int c4 = cl; c4 < n4; c4 += 128) {
          f4 v = {0.f, 0.f, 0.f, 0.f};
          if (ok) v = src[c4];
          float* dst = tile + tt * ld + 4 * c4;
          dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
      }
    } else {
      for (int tt = half; tt < TT; tt += 2)
        for (int c = cl; c < C; c += 128)
          tile[tt * ld + c] = (t0 + tt < T) ? amb[(size_t)(t0 + tt) * C + c] : 0.0f;
    }
  }
  // per-row scalars (symbol, lm at the symbol, lm at blank, lm_max, lmonly_norm) staged once: in the sweep below only
  // `prod` is a global load, so the compiler can keep several rows in flight
  const float* lmb = lm + (size_t)b * (S + 1) * C;
  const bool smooth = lmonly_norm != nullptr;
  float* rs_lmsym = tile + TT * ld;          // [S+1] each
  float* rs_lmblank = rs_lmsym + (S + 1);
  float* rs_lmmax = rs_lmblank + (S + 1);
  float* rs_lon = rs_lmmax + (S + 1);
  float* rs_ulog = rs_lon + (S + 1);
  int* rs_sym = reinterpret_cast<int*>(rs_ulog + (S + 1));
  for (int s = threadIdx.x; s <= S; s += blockDim.x) {
    const int sym = (s < S) ? min(max(symbols[(size_t)b * S + s], 0), C - 1) : blank;   // kept in bounds
    rs_sym[s] = sym;
    rs_lmsym[s] = lmb[(size_t)s * C + sym];
    rs_lmblank[s] = lmb[(size_t)s * C + blank];
    rs_lmmax[s] = lm_max[(size_t)b * (S + 1) + s];
    rs_lon[s] = smooth ? lmonly_norm[(size_t)b * (S + 1) + s] : 0.0f;
    rs_ulog[s] = smooth ? ulog[sym] : 0.0f;
  }
  __syncthreads();
  const int tx = threadIdx.x & (TT - 1);
  constexpr int NY = 256 / TT;
  const int ty = threadIdx.x / TT;
  const int t = t0 + tx;
  const int te = boundary ? boundary[4 * b + 3] : T;
  const float amx = (t < T) ? am_max[(size_t)b * T + t] : 0.0f;
  const float am_blank = tile[tx * ld + blank];
  float pen = 0.0f;
  if (delay_penalty > 0.0) pen = (float)((((double)te - 1.0) / 2.0 - (double)t) * delay_penalty);  // :305-321
  const float aon = (smooth && t < T) ? amonly_norm[(size_t)b * T + t] : 0.0f;
  const float ulog_blank = smooth ? ulog[blank] : 0.0f;
  const float* prodb = prod + (size_t)b * (S + 1) * T + t;
#pragma unroll 4
  for (int s = ty; s <= S; s += NY) {
    float nrm = 0.0f;
    const float lon = rs_lon[s];
    if (t < T) {
      nrm = logf(prodb[(size_t)s * T] + kTiny) + rs_lmmax[s] + amx;                                          // :180-186
      const float lmv = rs_lmblank[s];
      float v = am_blank + lmv - nrm;                                                                       // :214-216
      if (smooth) v = v * cs + (lmv - lon) * ls + (am_blank + ulog_blank - aon) * as;                       // :1333-1360
      py[((size_t)b * (S + 1) + s) * T + t] = v;
    }
    if (s < S && t < T1) {
      float v = -INFINITY;  // px[:, :, T] (:193-203) and fix_for_boundary (:218-219)
      if (t < T && (MOD || t != te)) {
        const float amv = tile[tx * ld + rs_sym[s]], lmv = rs_lmsym[s];
        v = amv + lmv - nrm;                                                                               // :187-211
        if (smooth) v = v * cs + (lmv - lon) * ls + (amv + rs_ulog[s] - aon) * as;                          // :1323-1355
      }
      if (delay_penalty > 0.0) v += pen;
      px[((size_t)b * S + s) * T1 + t] = v;
    }
  }
}

// one WAVE per (b, s) row (four rows per workgroup, no barrier): W[b,s,:] and the two row sums.
// rsx[b,s] = sum_t gpx'[b,s,t] (0 for s == S), rsy[b,s] = sum_t gpy[b,s,t]; gpx' = gpx with the overwritten cells
// (t == T, t == t_end) masked.  Rows are only 4-byte aligned (T+1 columns): 16-byte accesses through f4u.
template <bool MOD>
__global__ __launch_bounds__(256) void simple_bwd_w_kernel(const float* __restrict__ gpx, const float* __restrict__ gpy,
                                                           const Scale scale, const float* __restrict__ prod,
                                                           const int32_t* __restrict__ boundary, float* __restrict__ W,
                                                           float* __restrict__ rsx, float* __restrict__ rsy, float cs,
                                                           int T, int S, int B) {
  const int lane = threadIdx.x & 63;
  const size_t rowid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);     // b * (S+1) + s
  if (rowid >= (size_t)B * (S + 1)) return;
  const int b = (int)(rowid / (S + 1));
  const int s = (int)(rowid - (size_t)b * (S + 1));
  const int T1 = MOD ? T : T + 1;
  const int te = boundary ? boundary[4 * b + 3] : T;
  const size_t rowy = rowid * T;
  const size_t rowx = ((size_t)b * S + s) * T1;
  const float sc = scale.at(b);
  const bool hasx = s < S;
  float sx = 0.0f, sy = 0.0f;
  const int n4 = T >> 2;
  for (int q = lane; q < n4; q += 64) {
    const int t = 4 * q;
    f4 gx = {0.f, 0.f, 0.f, 0.f};
    if (hasx) gx = *reinterpret_cast<const f4u*>(gpx + rowx + t);
    const f4 gy = *reinterpret_cast<const f4u*>(gpy + rowy + t);
    const f4 pr = *reinterpret_cast<const f4u*>(prod + rowy + t);
    f4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = gx[e] * sc;
      if (!MOD && t + e == te) x = 0.0f;
      const float y = gy[e] * sc;
      sx += x; sy += y;
      w[e] = -cs * (x + y) / (pr[e] + kTiny);   // cs = 1 for the simple loss
    }
    *reinterpret_cast<f4u*>(W + rowy + t) = w;
  }
  for (int t = 4 * n4 + lane; t < T; t += 64) {
    float x = 0.0f;
    if (hasx && (MOD || t != te)) x = gpx[rowx + t] * sc;
    const float y = gpy[rowy + t] * sc;
    sx += x; sy += y;
    W[rowy + t] = -cs * (x + y) / (prod[rowy + t] + kTiny);
  }
  sx = wave_sum(sx); sy = wave_sum(sy);
  if (lane == 0) { rsx[rowid] = sx; rsy[rowid] = sy; }
}

#ifndef FTR_BWD_AM_PF
#define FTR_BWD_AM_PF 1     // frames whose write-out operands are fetched at the top of the kernel.  Measured (scripts/bwd_am_ab.sh,
                            // c3 / c2 / c5 / c4 in us): 1 -> 53 / 23.6 / 370 / 281, 4 -> 54.7 / 24.2 / 369 / 277, 6 -> 58.7 / 26.5 / 383 / 290,
                            // 8 at three workgroups per CU -> 63 / 34.5 / 370 / 281 (170 registers), 2 -> 57 / 25.5 / 488 / 306 (!);
                            // 0 -> 51.5 / 24.0 / 402 / 284 (one frame's operands are what gives the write-out its head start);
                            // the previous revision (no prefetch, four loads in flight in the sweeps): 54 / 23.0 / 431 / 299.
#endif
#ifndef FTR_BWD_AM_WGS
#define FTR_BWD_AM_WGS 4    // workgroups per CU the register budget is set for
#endif
// grid (ceil(T / TT), B); block 256 = (256/TT) column-owner groups x TT frames.  LDS: acc [TT][C + 1] + csy/csx [256].
// Thread (ty, tx) owns the accumulator cells acc[tx][c] with c % 8 == ty: every cell has one owner.
template <bool MOD, int TT>
__global__ __launch_bounds__(256, FTR_BWD_AM_WGS) void simple_bwd_am_kernel(const float* __restrict__ gpx, const float* __restrict__ gpy, const Scale scale,
                                     const float* __restrict__ damp, const float* __restrict__ am_probs,
                                     const int32_t* __restrict__ symbols, const int32_t* __restrict__ boundary,
                                     int blank, float kdir, const float* __restrict__ uvec,
                                     const float* __restrict__ amdot, float as, float* __restrict__ Rout,
                                     float* __restrict__ d_am, int T, int S, int C) {
  // Smoothed extension (uvec != NULL): the direct terms carry kdir = cs + as, and the AM-only normaliser
  // log(am_probs . u) + am_max contributes  am_probs[b,t,c] * u[c] * R[b,t],  R = -as (colsum gpx' + colsum gpy) / dot;
  // R is also written out (it feeds d u on the host side).
  extern __shared__ float acc[];  // [TT][C + 1], then csy [8][TT], csx [8][TT], then u16 row lists [8][S]
  // Neighbouring frame tiles on ONE XCD.  A tile of 16 frames reads 64-byte pieces of the g_px / g_py rows, half a cache
  // line; workgroups are dealt to the eight XCDs round robin by linear id and every XCD has its own L2, so in launch order
  // the two halves of a line are fetched by two different L2s (PMC: 1.37 x the algorithmic bytes).  Within every 16
  // consecutive linear ids (two per XCD) XCD x now takes the tiles 2x and 2x + 1 of the (utterance-major) tile list.
  int tile = blockIdx.x, b = blockIdx.y;
#ifndef FTR_EXP_BWD_AM_LAUNCH_ORDER
  if (TT == 16) {
    const unsigned total = gridDim.x * gridDim.y;
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    if ((lin | 15u) < total) {                     // a whole group of 16 (the last, partial group keeps its order)
      const unsigned nl = (lin & ~15u) + 2 * (lin & 7u) + ((lin >> 3) & 1u);
      tile = (int)(nl % gridDim.x); b = (int)(nl / gridDim.x);
    }
  }
#endif
  const int t0 = tile * TT;
  const int T1 = MOD ? T : T + 1;
  const int ld = C + 1;
  // A workgroup is a chain of dependent memory round trips (symbols -> row lists -> the g_px sweep -> the g_py sweep -> the
  // write-out's operands).  With long symbol sequences the sweeps are most of it: they keep 8 loads in flight now instead
  // of 4 (c5 431 -> 369 us, c4 299 -> 278).  Fetching the write-out's operands up here, before the sweeps, buys nothing that
  // four workgroups per CU do not already hide, and costs occupancy beyond a frame or two (see FTR_BWD_AM_PF).
  constexpr int NF = TT / 2, PF = NF < FTR_BWD_AM_PF ? NF : FTR_BWD_AM_PF;   // write-out: frames per thread, of which prefetched
  const int wo_half = threadIdx.x >> 7, wo_cl = threadIdx.x & 127;
  f4 pdp[PF > 0 ? PF : 1], pap[PF > 0 ? PF : 1];   // (PF = 0: no prefetch, the arrays are unused)
  if ((C & 3) == 0) {
    const int c4 = min(wo_cl, (C >> 2) - 1);
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const size_t o = ((size_t)b * T + min(t0 + wo_half + 2 * k, T - 1)) * C;
      pdp[k] = reinterpret_cast<const f4u*>(damp + o)[c4];
      pap[k] = reinterpret_cast<const f4u*>(am_probs + o)[c4];
    }
  }
  float* csy = acc + TT * ld;
  constexpr int NY = 256 / TT;       // thread rows = symbol classes (sym % NY)
  float* csx = csy + NY * TT;
  for (int i = threadIdx.x; i < TT * ld; i += blockDim.x) acc[i] = 0.0f;
  __syncthreads();
  const int tx = threadIdx.x & (TT - 1);
  const int ty = threadIdx.x / TT;
  const int t = t0 + tx;
  const int te = boundary ? boundary[4 * b + 3] : T;
  const bool tok = t < T;
  const bool xok = tok && (MOD || t != te);
  const int32_t* symb = symbols + (size_t)b * S;
  auto symc = [&](int si) { return min(max(symb[si], 0), C - 1); };   // caller data kept inside the accumulator tile
  // rows grouped by symbol class: thread row ty owns the columns with sym % 8 == ty (no two threads ever touch the
  // same accumulator, rows are added in ascending s: deterministic).  Each 32-thread row first builds the ordered
  // list of its rows in LDS (ballot compaction), then sweeps only those -- S/8 iterations instead of S.
  unsigned short* slist = reinterpret_cast<unsigned short*>(csx + NY * TT) + (size_t)ty * S;
  int cnt = 0;
  {
    const int hshift = TT * (ty % (64 / TT));   // which part of the wave this thread row is
    constexpr unsigned kRowMask = (TT == 32) ? 0xffffffffu : ((1u << (TT & 31)) - 1u);
    for (int s0 = 0; s0 < S; s0 += TT) {
      const int s = s0 + tx;
      const bool mine = s < S && (symc(s) & (NY - 1)) == ty;
      const unsigned m32 = (unsigned)(__ballot(mine) >> hshift) & kRowMask;
      if (mine) slist[cnt + __popc(m32 & ((1u << tx) - 1u))] = (unsigned short)s;
      cnt += __popc(m32);
    }
  }
  __syncthreads();
  float cs = 0.0f, cx = 0.0f;
  const float sc = scale.at(b);
  if (xok) {
    const float* gcol = gpx + (size_t)b * S * T1 + t;
    int i = 0;
    for (; i + 7 < cnt; i += 8) {                   // eight rows in flight, added in list order
      int sr[8], cr[8];
      float g[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) sr[e] = slist[i + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) { g[e] = gcol[(size_t)sr[e] * T1]; cr[e] = symc(sr[e]); }
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float ge = g[e] * sc; acc[tx * ld + cr[e]] += ge; cx += ge; }
    }
    for (; i + 3 < cnt; i += 4) {
      const int s_0 = slist[i], s_1 = slist[i + 1], s_2 = slist[i + 2], s_3 = slist[i + 3];
      const float g0 = gcol[(size_t)s_0 * T1] * sc, g1 = gcol[(size_t)s_1 * T1] * sc;
      const float g2 = gcol[(size_t)s_2 * T1] * sc, g3 = gcol[(size_t)s_3 * T1] * sc;
      acc[tx * ld + symc(s_0)] += g0; cx += g0;
      acc[tx * ld + symc(s_1)] += g1; cx += g1;
      acc[tx * ld + symc(s_2)] += g2; cx += g2;
      acc[tx * ld + symc(s_3)] += g3; cx += g3;
    }
    for (; i < cnt; ++i) {
      const int s_0 = slist[i];
      const float g0 = gcol[(size_t)s_0 * T1] * sc;
      acc[tx * ld + symc(s_0)] += g0; cx += g0;
    }
  }
  if (tok) {
    const float* ycol = gpy + (size_t)b * (S + 1) * T + t;
#pragma unroll 8
    for (int s = ty; s <= S; s += NY) cs += ycol[(size_t)s * T] * sc;
  }
  csy[ty * TT + tx] = cs;
  csx[ty * TT + tx] = cx;
  __syncthreads();
  if (threadIdx.x < TT) {                          // per-frame totals, summed once (group order), into row 0
    float col = 0.0f, colx = 0.0f;
#pragma unroll
    for (int g = 0; g < NY; ++g) { col += csy[g * TT + threadIdx.x]; colx += csx[g * TT + threadIdx.x]; }
    csy[threadIdx.x] = col; csx[threadIdx.x] = colx;
  }
  __syncthreads();
  // write-out: two frames per pass, 16 bytes per lane along c where C allows
  {
    const int half = wo_half, cl = wo_cl;
    const bool vec = (C & 3) == 0;
    const int n4 = C >> 2;
    auto frame_terms = [&](int tt, float& col, float& R) {   // the blank column's sum and the smoothed term's factor of frame tt
      const float colx = csx[tt];
      col = csy[tt];
      R = 0.0f;
      if (uvec) R = -as * (colx + col) / amdot[(size_t)b * T + t0 + tt];
      col *= kdir;
    };
    auto quad = [&](size_t o, int c4, const float* arow, f4 dp, f4 ap, float col, float R) {
      f4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = dp[e] * ap[e] + kdir * arow[4 * c4 + e];
        if (4 * c4 + e == blank) v[e] += col;
        if (uvec) v[e] += ap[e] * uvec[4 * c4 + e] * R;
      }
      reinterpret_cast<f4u*>(d_am + o)[c4] = v;
    };
    if (vec) {                                     // first column pass of the first PF frames: operands fetched at the top
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        const int tt = half + 2 * k;
        if (t0 + tt >= T || cl >= n4) continue;
        float col, R;
        frame_terms(tt, col, R);
        quad(((size_t)b * T + t0 + tt) * C, cl, acc + tt * ld, pdp[k], pap[k], col, R);
      }
    }
    for (int tt = half; tt < TT; tt += 2) {        // everything else
      if (t0 + tt >= T) continue;
      float col, R;
      frame_terms(tt, col, R);
      if (uvec && cl == 0) Rout[(size_t)b * T + t0 + tt] = R;
      const size_t o = ((size_t)b * T + t0 + tt) * C;
      const float* arow = acc + tt * ld;
      if (vec) {
        for (int c4 = (tt < 2 * PF ? cl + 128 : cl); c4 < n4; c4 += 128)
          quad(o, c4, arow, reinterpret_cast<const f4u*>(damp + o)[c4], reinterpret_cast<const f4u*>(am_probs + o)[c4], col, R);
      } else {
        for (int c = cl; c < C; c += 128) {
          float v = damp[o + c] * am_probs[o + c] + kdir * arow[c];
          if (c == blank) v += col;
          if (uvec) v += am_probs[o + c] * uvec[c] * R;
          d_am[o + c] = v;
        }
      }
    }
  }
}

// one thread per element of d lm [B, S+1, C]
__global__ void simple_bwd_lm_kernel(const float* __restrict__ dlmp, const float* __restrict__ lm_probs,
                                     const int32_t* __restrict__ symbols, const float* __restrict__ rsx,
                                     const float* __restrict__ rsy, int blank, float kdir,
                                     const float* __restrict__ arow, const float* __restrict__ invsum,
                                     const float* __restrict__ gu, float* __restrict__ d_lm, int S, int C,
                                     size_t total) {
  // Smoothed extension (arow != NULL): direct terms carry kdir = cs + ls; the LM-only normaliser and the batch
  // unigram contribute  lm_probs[b,s,c] * (arow[b,s] + gu[c] * invsum[b,s])  (host side prepares the vectors).
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / C;  // b*(S+1) + s
    const int c = (int)(i - row * C);
    const int s = (int)(row % (size_t)(S + 1));
    const size_t b = row / (size_t)(S + 1);
    float v = dlmp[i] * lm_probs[i];
    if (s < S && c == symbols[b * S + s]) v += kdir * rsx[row];
    if (c == blank) v += kdir * rsy[row];
    if (arow) v += lm_probs[i] * (arow[row] + gu[c] * invsum[row]);
    d_lm[i] = v;
  }
}

// the same, one WAVE per row (b, s), 16 bytes per lane (C % 4 == 0): the row's scalars once per wave, no per-element index
// division (the element kernel above spends its time on three 64-bit divisions per element: 16 us at c3 for 39 MB)
__global__ __launch_bounds__(256) void simple_bwd_lm_rows_kernel(const float* __restrict__ dlmp, const float* __restrict__ lm_probs,
                                                                 const int32_t* __restrict__ symbols, const float* __restrict__ rsx,
                                                                 const float* __restrict__ rsy, int blank, float kdir,
                                                                 const float* __restrict__ arow, const float* __restrict__ invsum,
                                                                 const float* __restrict__ gu, float* __restrict__ d_lm, int S, int C,
                                                                 size_t rows) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);     // b * (S+1) + s
  if (row >= rows) return;
  const int s = (int)(row % (size_t)(S + 1));
  const size_t b = row / (size_t)(S + 1);
  const int sym = s < S ? symbols[b * S + s] : -1;
  const float ax = kdir * rsx[row], ay = kdir * rsy[row];
  const float ar = arow ? arow[row] : 0.0f, is = arow ? invsum[row] : 0.0f;
  const int n4 = C >> 2;
  const f4u* dp4 = reinterpret_cast<const f4u*>(dlmp + row * C);
  const f4u* lp4 = reinterpret_cast<const f4u*>(lm_probs + row * C);
  f4u* o4 = reinterpret_cast<f4u*>(d_lm + row * C);
  for (int q = lane; q < n4; q += 64) {
    const f4 dp = dp4[q], lp = lp4[q];
    f4 g = {0.f, 0.f, 0.f, 0.f};
    if (arow) g = reinterpret_cast<const f4u*>(gu)[q];
    f4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * q + e;
      float x = dp[e] * lp[e];                       // the element kernel's expressions, term by term
      if (c == sym) x += ax;
      if (c == blank) x += ay;
      if (arow) x += lp[e] * (ar + g[e] * is);
      v[e] = x;
    }
    o4[q] = v;
  }
}

}  // namespace

int simple_rowmax_exp(const float* x, float* probs, float* rowmax, float* rowsum, const float* dotvec, float* dot,
                      size_t rows, int C, hipStream_t st) {
  if (rows == 0 || C == 0) return FTR_OK;
  const int wpb = 4;
  const unsigned blocks = (unsigned)((rows + wpb - 1) / wpb);
  if ((C & 3) == 0) hipLaunchKernelGGL(rowmax_exp_kernel<true>, dim3(blocks), dim3(64 * wpb), 0, st, x, probs, rowmax, rowsum, dotvec, dot, rows, C);
  else hipLaunchKernelGGL(rowmax_exp_kernel<false>, dim3(blocks), dim3(64 * wpb), 0, st, x, probs, rowmax, rowsum, dotvec, dot, rows, C);
  return check_launch("rowmax_exp");
}

int simple_rowmax_exp_pair(const float* x1, float* probs1, float* rowmax1, size_t rows1, const float* x2, float* probs2, float* rowmax2,
                           size_t rows2, int C, hipStream_t st) {
  if (rows1 + rows2 == 0 || C == 0) return FTR_OK;
  const int wpb = 4;
  const unsigned blocks = (unsigned)((rows1 + rows2 + wpb - 1) / wpb);
  if ((C & 3) == 0) hipLaunchKernelGGL(rowmax_exp_pair_kernel<true>, dim3(blocks), dim3(64 * wpb), 0, st, x1, probs1, rowmax1, rows1, x2, probs2, rowmax2, rows2, C);
  else hipLaunchKernelGGL(rowmax_exp_pair_kernel<false>, dim3(blocks), dim3(64 * wpb), 0, st, x1, probs1, rowmax1, rows1, x2, probs2, rowmax2, rows2, C);
  return check_launch("rowmax_exp_pair");
}

int simple_rowdot(const float* x, const float* v, float* dot, size_t rows, int C, hipStream_t st) {
  if (rows == 0) return FTR_OK;
  const int wpb = 4;
  const unsigned blocks = (unsigned)((rows + wpb - 1) / wpb);
  if ((C & 3) == 0) hipLaunchKernelGGL(rowdot_kernel<true>, dim3(blocks), dim3(64 * wpb), 0, st, x, v, dot, rows, C);
  else hipLaunchKernelGGL(rowdot_kernel<false>, dim3(blocks), dim3(64 * wpb), 0, st, x, v, dot, rows, C);
  return check_launch("rowdot");
}

size_t simple_colsum_workspace_floats(size_t rows, int C) { return ((rows + kColsumSlab - 1) / kColsumSlab) * (size_t)C; }

int simple_colsum_weighted(const float* x, const float* w, float* out, float* ws, size_t ws_floats, size_t rows, int C,
                           hipStream_t st) {
  if (C == 0) return FTR_OK;
  const size_t nslab = (rows + kColsumSlab - 1) / kColsumSlab;
  if (ws_floats < nslab * (size_t)C) { set_error("colsum_weighted: workspace of %zu floats, %zu needed", ws_floats, nslab * (size_t)C); return FTR_ERR_INVALID_ARG; }
  if (nslab > 0x7fffffff) { set_error("colsum_weighted: too many rows"); return FTR_ERR_UNSUPPORTED; }
  if (nslab > 0) {
    if ((C & 3) == 0) hipLaunchKernelGGL(colsum_partial_kernel<true>, dim3((unsigned)nslab, (C / 4 + 255) / 256), dim3(256), 0, st, x, w, ws, rows, C);
    else hipLaunchKernelGGL(colsum_partial_kernel<false>, dim3((unsigned)nslab, (C + 255) / 256), dim3(256), 0, st, x, w, ws, rows, C);
    int rc = check_launch("colsum_weighted (partial)");
    if (rc != FTR_OK) return rc;
  }
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, st, ws, out, (int)nslab, C);
  return check_launch("colsum_weighted (final)");
}

static int tile_lds_ok(size_t lds, const char* what) {
  if (lds > 160 * 1024) { set_error("%s: C too large for the LDS tile (%zu bytes)", what, lds); return FTR_ERR_UNSUPPORTED; }
  return FTR_OK;
}

template <typename K>
static int reserve_lds(K kernel, size_t lds, const char* what) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("%s: cannot reserve %zu bytes of LDS: %s", what, lds, hipGetErrorString(e)); return FTR_ERR_LAUNCH; }
  }
  return FTR_OK;
}

int simple_logprobs_fwd(const float* am, const float* lm, const int32_t* symbols, const float* prod,
                        const float* am_max, const float* lm_max, const int32_t* boundary, int blank,
                        double delay_penalty, const float* lmonly_norm, const float* amonly_norm, const float* ulog,
                        float cs, float ls, float as, float* px, float* py, int B, int T, int S, int C, int modified,
                        hipStream_t st) {
  const int T1 = modified ? T : T + 1;
  const bool narrow = C > FTR_TT_NARROW_ABOVE;
  const int TT = narrow ? kTTnarrow : kTTwide;
  const size_t lds = sizeof(float) * ((size_t)TT * (C + 1) + 6 * (size_t)(S + 1));
  int rc = tile_lds_ok(lds, "simple_logprobs_fwd");
  if (rc != FTR_OK) return rc;
  const dim3 grid((T1 + TT - 1) / TT, B);
#define FTR_LAUNCH_FWD(MODV, TTV)                                                                                       \
  do {                                                                                                                  \
    if ((rc = reserve_lds(simple_fwd_kernel<MODV, TTV>, lds, "simple_logprobs_fwd")) != FTR_OK) return rc;              \
    hipLaunchKernelGGL((simple_fwd_kernel<MODV, TTV>), grid, dim3(256), lds, st, am, lm, symbols, prod, am_max, lm_max, \
                       boundary, blank, delay_penalty, lmonly_norm, amonly_norm, ulog, cs, ls, as, px, py, T, S, C);    \
  } while (0)
  if (modified) { if (narrow) FTR_LAUNCH_FWD(true, kTTnarrow); else FTR_LAUNCH_FWD(true, kTTwide); }
  else { if (narrow) FTR_LAUNCH_FWD(false, kTTnarrow); else FTR_LAUNCH_FWD(false, kTTwide); }
#undef FTR_LAUNCH_FWD
  return check_launch("simple_logprobs_fwd");
}

int simple_logprobs_bwd_w(const float* gpx, const float* gpy, Scale scale, const float* prod, const int32_t* boundary,
                          float* W, float* rsx, float* rsy, float cs, int B, int T, int S, int modified, hipStream_t st) {
  const size_t rows = (size_t)B * (S + 1);
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (modified) hipLaunchKernelGGL(simple_bwd_w_kernel<true>, grid, dim3(256), 0, st, gpx, gpy, scale, prod, boundary, W, rsx, rsy, cs, T, S, B);
  else hipLaunchKernelGGL(simple_bwd_w_kernel<false>, grid, dim3(256), 0, st, gpx, gpy, scale, prod, boundary, W, rsx, rsy, cs, T, S, B);
  return check_launch("simple_logprobs_bwd_w");
}

int simple_logprobs_bwd_am(const float* gpx, const float* gpy, Scale scale, const float* damp, const float* am_probs,
                           const int32_t* symbols, const int32_t* boundary, int blank, float kdir, const float* uvec,
                           const float* amdot, float as, float* Rout, float* d_am, int B, int T, int S, int C,
                           int modified, hipStream_t st) {
  if (S > 65535) { set_error("simple_logprobs_bwd_am: S = %d > 65535 is not supported", S); return FTR_ERR_UNSUPPORTED; }
  const bool narrow = C > FTR_TT_NARROW_ABOVE;
  const int TT = narrow ? kTTnarrow : kTTwide;
  const size_t lds = sizeof(float) * ((size_t)TT * (C + 1) + 2 * 256) + sizeof(unsigned short) * (256 / TT) * (size_t)S;
  int rc = tile_lds_ok(lds, "simple_logprobs_bwd_am");
  if (rc != FTR_OK) return rc;
  const dim3 grid((T + TT - 1) / TT, B);
#define FTR_LAUNCH_AM(MODV, TTV)                                                                                        \
  do {                                                                                                                  \
    if ((rc = reserve_lds(simple_bwd_am_kernel<MODV, TTV>, lds, "simple_logprobs_bwd_am")) != FTR_OK) return rc;        \
    hipLaunchKernelGGL((simple_bwd_am_kernel<MODV, TTV>), grid, dim3(256), lds, st, gpx, gpy, scale, damp, am_probs,    \
                       symbols, boundary, blank, kdir, uvec, amdot, as, Rout, d_am, T, S, C);                           \
  } while (0)
  if (modified) { if (narrow) FTR_LAUNCH_AM(true, kTTnarrow); else FTR_LAUNCH_AM(true, kTTwide); }
  else { if (narrow) FTR_LAUNCH_AM(false, kTTnarrow); else FTR_LAUNCH_AM(false, kTTwide); }
#undef FTR_LAUNCH_AM
  return check_launch("simple_logprobs_bwd_am");
}

int simple_logprobs_bwd_lm(const float* dlmp, const float* lm_probs, const int32_t* symbols, const float* rsx,
                           const float* rsy, int blank, float kdir, const float* arow, const float* invsum,
                           const float* gu, float* d_lm, int B, int S, int C, hipStream_t st) {
  const size_t total = (size_t)B * (S + 1) * C;
  if (total == 0) return FTR_OK;
  if ((C & 3) == 0) {
    const size_t rows = (size_t)B * (S + 1);
    hipLaunchKernelGGL(simple_bwd_lm_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, dlmp, lm_probs, symbols, rsx, rsy,
                       blank, kdir, arow, invsum, gu, d_lm, S, C, rows);
    return check_launch("simple_logprobs_bwd_lm");
  }
  const size_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(simple_bwd_lm_kernel, dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)), dim3(256), 0, st,
                     dlmp, lm_probs, symbols, rsx, rsy, blank, kdir, arow, invsum, gu, d_lm, S, C, total);
  return check_launch("simple_logprobs_bwd_lm");
}

}  // namespace ftr
