// csrc/prune.hip -- integer side of the pruning pipeline + the prune gather, gfx950.
//   cummin_kernel          replaces tensor_kernel_scan_innermost_dim_with_indices
//                          (tf_fast_rnnt/csrc/mutual_information_cuda.cu:895-1012)
//   prune_argmax_kernel    replaces rnnt_loss.py:722-748 (cumsum, window sums, argmax, padding frames)
//   prune_adjust_kernel    replaces _adjust_pruning_lower_bound + _monotonic_lower_bound + the ranges
//                          broadcast (rnnt_loss.py:553-641, 756-759): both suffix-min scans, the two
//                          linear transforms, the clip and the [B,T,r] write in one launch
//   do_pruning_kernel      replaces do_rnnt_pruning (rnnt_loss.py:763-812)
// (paths relative to /root/reference/tf_fast_rnnt/python/tf_fast_rnnt/ unless they start with csrc).
// All integer results are bit-exact with oracle/mi_oracle.c; the float window sums use the same
// canonical order (sequential f32 cumsum along S, additions only -> no contraction possible).
#include "ftr_common.h"
#include <cstring>
#include <cstdlib>
#include <limits.h>

namespace ftr {
namespace {

// inclusive min-scan across the 64 lanes of a wave, in lane order
__device__ __forceinline__ int wave_incl_min_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off, 64);
    if (lane >= off) v = min(v, t);
  }
  return v;
}

// one wave per row, 64 columns per pass, running minimum carried in a register
__global__ void cummin_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int32_t* src = in + (size_t)row * cols;
  int32_t* dst = out + (size_t)row * cols;
  int carry = INT_MAX;  // init = numeric_limits<int>::max(), mutual_information_cuda.cu:1001
  for (int c0 = 0; c0 < cols; c0 += 64) {
    const int c = c0 + lane;
    int v = (c < cols) ? src[c] : INT_MAX;
    v = wave_incl_min_scan(v, lane);
    v = min(v, carry);
    if (c < cols) dst[c] = v;
    carry = __shfl(v, 63, 64);
  }
}

// one thread per (b, t); consecutive threads take consecutive t (coalesced along the lattice rows).
// lead = cumsum[s0 + r], lag = cumsum[s0], both accumulated in the canonical sequential order.
__global__ void prune_argmax_kernel(const float* __restrict__ px_grad, const float* __restrict__ py_grad,
                                    const int32_t* __restrict__ boundary, int32_t* __restrict__ s_begin,
                                    int B, int S, int T, int T1, int r) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  const int S1 = S + 1;
  const int nwin = S1 - r + 1;
  const float* pyc = py_grad + (size_t)b * S1 * T + t;
  const float* pxc = px_grad + (size_t)b * S * T1 + t;
  float lead = 0.0f, lag = 0.0f;
  for (int s = 0; s < r; ++s) lead = lead + pyc[(size_t)s * T];
  int best = 0;
  float bestv = 0.0f;
  // The cumulative sums must be taken in this order (bit-exact ranges), but the loads do not depend on them: fetch
  // CHK windows' worth of operands first (3 * CHK independent loads in flight), then run the serial arithmetic.
  constexpr int CHK = 16;
  float ld[2][CHK], lg[2][CHK], xv[2][CHK];
  auto fetch = [&](int c0, float (&l)[CHK], float (&g)[CHK], float (&x)[CHK]) {
#pragma unroll
    for (int u = 0; u < CHK; ++u) {
      const int s0 = c0 + u;
      const bool more = s0 + 1 < nwin;
      l[u] = more ? pyc[(size_t)(s0 + r) * T] : 0.0f;
      g[u] = more ? pyc[(size_t)s0 * T] : 0.0f;
      x[u] = (s0 > 0 && s0 < nwin) ? pxc[(size_t)(s0 - 1) * T1] : 0.0f;
    }
  };
  auto consume = [&](int c0, const float (&l)[CHK], const float (&g)[CHK], const float (&x)[CHK]) {
#pragma unroll
    for (int u = 0; u < CHK; ++u) {
      const int s0 = c0 + u;
      if (s0 < nwin) {
        const float blk = lead - lag;                                      // rnnt_loss.py:725
        const float fin = blk - x[u];                                      // :726-728 (px_pad[.,0] = 0)
        if (s0 == 0 || fin > bestv) { best = s0; bestv = fin; }            // :729, first maximum
        if (s0 + 1 < nwin) {
          lead = lead + l[u];
          lag = lag + g[u];
        }
      }
    }
  };
  // two chunks in flight: the next chunk's 3 * CHK loads are issued before the current chunk's serial adds run
  fetch(0, ld[0], lg[0], xv[0]);
  for (int c0 = 0; c0 < nwin; c0 += 2 * CHK) {
    if (c0 + CHK < nwin) fetch(c0 + CHK, ld[1], lg[1], xv[1]);
    consume(c0, ld[0], lg[0], xv[0]);
    if (c0 + CHK < nwin) {
      if (c0 + 2 * CHK < nwin) fetch(c0 + 2 * CHK, ld[0], lg[0], xv[0]);
      consume(c0 + CHK, ld[1], lg[1], xv[1]);
    }
  }
  const int se = boundary[4 * b + 2], te = boundary[4 * b + 3];
  int pad = se - r + 1;                                                  // :744-746
  if (pad < 0) pad = 0;
  s_begin[(size_t)b * T + t] = (t < te - 1) ? best : pad;                // :741-748
}

// The same walk with its LOADS spread over the waves of a workgroup.  The cumulative sums have to be taken in the canonical
// sequential order (bit-exact ranges), but the loads do not depend on them, and a column walk of S rows is 3 S loads of 256
// bytes (64 columns) at rows 4 KB apart: one wave that takes them 64 rows at a time pays a memory round trip per 64 rows
// (c3: 4 rounds of ~8 us, c5: 16).  Here chunk c of kSplitQ windows belongs to wave c mod NW: every wave requests its first
// chunk at once, then the chunks are consumed in order -- the owner continues the running sums (lead, lag, best) from LDS
// where the previous owner left them, the same additions in the same order, and requests its next chunk -- with a barrier
// between chunks.  One round trip for up to 64 NW rows.
constexpr int kSplitQ = 64;
__global__ __launch_bounds__(512) void prune_argmax_split_kernel(const float* __restrict__ px_grad, const float* __restrict__ py_grad,
                                                                 const int32_t* __restrict__ boundary, int32_t* __restrict__ s_begin,
                                                                 int B, int S, int T, int T1, int r) {
  __shared__ float st_lead[64], st_lag[64], st_bestv[64];
  __shared__ int st_best[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  const int b = blockIdx.y;
  const bool live = t < T;
  const int tc = live ? t : T - 1;                 // clamped: every lane loads, dead lanes store nothing
  const int S1 = S + 1;
  const int nwin = S1 - r + 1;
  const float* pyc = py_grad + (size_t)b * S1 * T + tc;
  const float* pxc = px_grad + (size_t)b * S * T1 + tc;
  constexpr int Q = kSplitQ;
  const int NC = (nwin + Q - 1) / Q;
  float n[Q], g[Q], x[Q];
  auto fetch = [&](int c) {
    const int c0 = c * Q;
#pragma unroll
    for (int u = 0; u < Q; ++u) {
      const int s0 = c0 + u;
      const bool more = s0 + 1 < nwin;
      n[u] = more ? pyc[(size_t)(s0 + r) * T] : 0.0f;
      g[u] = more ? pyc[(size_t)s0 * T] : 0.0f;
      x[u] = (s0 > 0 && s0 < nwin) ? pxc[(size_t)(s0 - 1) * T1] : 0.0f;
    }
  };
  if (wave < NC) fetch(wave);
  float lead = 0.0f, lag = 0.0f, bestv = 0.0f;
  int best = 0;
  if (wave == 0)
    for (int s = 0; s < r; ++s) lead = lead + pyc[(size_t)s * T];
  for (int c = 0; c < NC; ++c) {
    if (c % NW == wave) {
      if (c > 0) { lead = st_lead[lane]; lag = st_lag[lane]; bestv = st_bestv[lane]; best = st_best[lane]; }
      const int c0 = c * Q;
#pragma unroll
      for (int u = 0; u < Q; ++u) {
        const int s0 = c0 + u;
        if (s0 < nwin) {
          const float blk = lead - lag;                                      // rnnt_loss.py:725
          const float fin = blk - x[u];                                      // :726-728 (px_pad[.,0] = 0)
          if (s0 == 0 || fin > bestv) { best = s0; bestv = fin; }            // :729, first maximum
          if (s0 + 1 < nwin) {
            lead = lead + n[u];
            lag = lag + g[u];
          }
        }
      }
      if (c + 1 < NC) { st_lead[lane] = lead; st_lag[lane] = lag; st_bestv[lane] = bestv; st_best[lane] = best; }
      if (c + NW < NC) fetch(c + NW);
      if (c + 1 == NC && live) {
        const int se = boundary[4 * b + 2], te = boundary[4 * b + 3];
        int pad = se - r + 1;                                                // :744-746
        if (pad < 0) pad = 0;
        s_begin[(size_t)b * T + t] = (t < te - 1) ? best : pad;              // :741-748
      }
    }
    if (c + 1 < NC) __syncthreads();
  }
}

#ifndef FTR_PRUNE_CHK
#define FTR_PRUNE_CHK 64
#endif
// The same with the window length as a template parameter (1 <= R <= 16): every py_grad value is LOADED ONCE.  The lag
// cumsum consumes, R rows later, what the lead cumsum loaded (carried in registers across chunks), in the same order
// and with the same additions, so the sums -- and the ranges -- are bit-identical to the kernel above.
template <int R>
__global__ __launch_bounds__(64) void prune_argmax_once_kernel(const float* __restrict__ px_grad, const float* __restrict__ py_grad,
                                         const int32_t* __restrict__ boundary, int32_t* __restrict__ s_begin,
                                         int B, int S, int T, int T1) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  const int S1 = S + 1;
  const int nwin = S1 - R + 1;
  const float* pyc = py_grad + (size_t)b * S1 * T + t;
  const float* pxc = px_grad + (size_t)b * S * T1 + t;
  constexpr int CHK = FTR_PRUNE_CHK;   // rows in flight per array and register set: the walk is sequential, so its time is (rows / CHK) memory round trips
  // v[i] of a chunk starting at window c0 is py_grad row c0 + i (i < CHK + R): rows c0 .. c0+R-1 come from the previous
  // chunk's tail (`carry`), rows c0+R .. c0+CHK+R-1 are loaded (the "lead" rows of this chunk's windows)
  float carry[R];
  float lead = 0.0f, lag = 0.0f;
#pragma unroll
  for (int s = 0; s < R; ++s) { carry[s] = pyc[(size_t)s * T]; }
#pragma unroll
  for (int s = 0; s < R; ++s) lead = lead + carry[s];
  int best = 0;
  float bestv = 0.0f;
  float nv[2][CHK], xv[2][CHK];
  auto fetch = [&](int c0, float (&n)[CHK], float (&x)[CHK]) {
#pragma unroll
    for (int u = 0; u < CHK; ++u) {
      const int s0 = c0 + u;
      n[u] = (s0 + 1 < nwin) ? pyc[(size_t)(s0 + R) * T] : 0.0f;
      x[u] = (s0 > 0 && s0 < nwin) ? pxc[(size_t)(s0 - 1) * T1] : 0.0f;
    }
  };
  auto consume = [&](int c0, const float (&n)[CHK], const float (&x)[CHK]) {
#pragma unroll
    for (int u = 0; u < CHK; ++u) {
      const int s0 = c0 + u;
      const float g = (u < R) ? carry[u < R ? u : 0] : n[u >= R ? u - R : 0];     // py_grad row s0
      if (s0 < nwin) {
        const float blk = lead - lag;                                      // rnnt_loss.py:725
        const float fin = blk - x[u];                                      // :726-728 (px_pad[.,0] = 0)
        if (s0 == 0 || fin > bestv) { best = s0; bestv = fin; }            // :729, first maximum
        if (s0 + 1 < nwin) {
          lead = lead + n[u];
          lag = lag + g;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) carry[i] = n[CHK - R + i];                 // rows c0+CHK .. c0+CHK+R-1 for the next chunk
  };
  fetch(0, nv[0], xv[0]);
  for (int c0 = 0; c0 < nwin; c0 += 2 * CHK) {
    if (c0 + CHK < nwin) fetch(c0 + CHK, nv[1], xv[1]);
    consume(c0, nv[0], xv[0]);
    if (c0 + CHK < nwin) {
      if (c0 + 2 * CHK < nwin) fetch(c0 + 2 * CHK, nv[0], xv[0]);
      consume(c0 + CHK, nv[1], xv[1]);
    }
  }
  const int se = boundary[4 * b + 2], te = boundary[4 * b + 3];
  int pad = se - R + 1;                                                  // :744-746
  if (pad < 0) pad = 0;
  s_begin[(size_t)b * T + t] = (t < te - 1) ? best : pad;                // :741-748
}

// inclusive SUFFIX min-scan across the 64 lanes of a wave (lane l gets the minimum over lanes l..63)
__device__ __forceinline__ int wave_incl_suffix_min(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_down(v, off, 64);
    if (lane + off < 64) v = min(v, t);
  }
  return v;
}

// one wave per utterance, ONE right-to-left sweep over T in super-blocks of 1024 frames (16 consecutive frames per
// lane, held in registers): suffix-min (rnnt_loss.py:628), x = -(x - (r-1) t) (:630-632), suffix-min again (:634),
// clip at 0 (:636), transform back (:638-640), ranges write (:758-759).  Both scans run right to left, so the second
// one can follow the first super-block by super-block with its own carry: no round trip through memory in between.
__global__ void prune_adjust_kernel(const int32_t* __restrict__ s_begin, int32_t* __restrict__ ranges, int B,
                                    int T, int r_out, int r_con) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  constexpr int PER = 16, SB = 64 * PER;
  const int32_t* x = s_begin + (size_t)b * T;
  int carry1 = INT_MAX, carry2 = INT_MAX;
  for (int base = ((T - 1) / SB) * SB; base >= 0; base -= SB) {
    const int t0 = base + lane * PER;
    int v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = (t0 + i < T) ? x[t0 + i] : INT_MAX;
    // pass 1
#pragma unroll
    for (int i = PER - 2; i >= 0; --i) v[i] = min(v[i], v[i + 1]);
    int incl = wave_incl_suffix_min(v[0], lane);              // lanes l..63 of this super-block
    int right = __shfl_down(incl, 1, 64);                     // lanes l+1..63
    right = (lane == 63) ? carry1 : min(right, carry1);
    carry1 = min(__shfl(incl, 0, 64), carry1);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int t = t0 + i;
      v[i] = (t < T) ? -(min(v[i], right) - (r_con - 1) * t) : INT_MAX;
    }
    // pass 2
#pragma unroll
    for (int i = PER - 2; i >= 0; --i) v[i] = min(v[i], v[i + 1]);
    incl = wave_incl_suffix_min(v[0], lane);
    right = __shfl_down(incl, 1, 64);
    right = (lane == 63) ? carry2 : min(right, carry2);
    carry2 = min(__shfl(incl, 0, 64), carry2);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int t = t0 + i;
      if (t < T) {
        const int z = max(min(v[i], right), 0);
        const int sb = -(z - (r_con - 1) * t);
        int32_t* dst = ranges + ((size_t)b * T + t) * r_out;
        for (int k = 0; k < r_out; ++k) dst[k] = sb + k;
      }
    }
  }
}

// one thread per 16 bytes of output row (vector path, C % 4 == 0) or per element (scalar path)
template <bool VEC>
__global__ __launch_bounds__(128) void do_pruning_kernel(const float* __restrict__ am, const float* __restrict__ lm,
                                                         const int32_t* __restrict__ ranges, float* __restrict__ am_p,
                                                         float* __restrict__ lm_p, int T, int S1, int C, int r) {
  // one block per frame (b,t): the am row is read once and written r times, the r lm rows are gathered; the outputs
  // (2 * N bytes, far beyond any cache) are written with non-temporal stores.  No per-thread index divisions.
  // am_p == nullptr: only the gather (the host keeps am_pruned as a broadcast view of am, which is all it is).
  const bool copy_am = am_p != nullptr;
  // XCD-aware frame order: workgroups are dealt to the 8 XCDs round robin and every XCD has its own L2; in launch order each
  // of them fetches every utterance's lm rows.  Here XCD k takes a contiguous eighth of the frames: the rows of an utterance
  // are fetched by one L2.
  size_t bt = blockIdx.x;
  if ((gridDim.x & 7u) == 0) bt = (size_t)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const size_t b = bt / T;
  const int32_t* rg = ranges + bt * r;
  const float* lmb = lm + b * S1 * C;
  float* ao = am_p + bt * r * C;
  float* lo = lm_p + bt * r * C;
  if (VEC) {
    const int n4 = C >> 2;
    const f4u* arow = reinterpret_cast<const f4u*>(am + bt * C);
    for (int c4 = threadIdx.x; c4 < n4; c4 += 128) {
      f4 a = {0.0f, 0.0f, 0.0f, 0.0f};
      if (copy_am) a = arow[c4];
      for (int k0 = 0; k0 < r; k0 += 8) {
        f4 l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < r) l[u] = reinterpret_cast<const f4u*>(lmb + (size_t)min(max(rg[k0 + u], 0), S1 - 1) * C)[c4];   // caller data: kept in bounds
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < r) {
            // (plain stores make this kernel 12 us faster at c3 and the caller's next kernel 10 us slower, with or without
            // writing the frames last to first: measured inside the step, scripts/order_ab.sh)
            if (copy_am) __builtin_nontemporal_store(a, reinterpret_cast<f4u*>(ao + (size_t)(k0 + u) * C) + c4);
            __builtin_nontemporal_store(l[u], reinterpret_cast<f4u*>(lo + (size_t)(k0 + u) * C) + c4);
          }
      }
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 128) {
      const float a = copy_am ? am[bt * C + c] : 0.0f;
      for (int k = 0; k < r; ++k) {
        if (copy_am) ao[(size_t)k * C + c] = a;
        lo[(size_t)k * C + c] = lmb[(size_t)min(max(rg[k], 0), S1 - 1) * C + c];
      }
    }
  }
}

// ---- backward of the prune gather (what TF autodiff does for rnnt_loss.py:802-811: a reduce_sum over the
// broadcast axis for am, an unsorted-segment-sum for the gather of lm).
// d am[b,t,:] = sum_k g_am_p[b,t,k,:]: one thread per 16 bytes of d am.
template <bool VEC>
__global__ void do_pruning_bwd_am_kernel(const float* __restrict__ g_am_p, float* __restrict__ d_am, int C, int r,
                                         size_t total) {
  const int per_row = VEC ? (C >> 2) : C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bt = i / per_row;
    const int c = (int)(i - bt * per_row);
    if (VEC) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < r; ++k) acc += reinterpret_cast<const f4u*>(g_am_p + (bt * r + k) * C)[c];
      reinterpret_cast<f4u*>(d_am + bt * C)[c] = acc;
    } else {
      float acc = 0.f;
      for (int k = 0; k < r; ++k) acc += g_am_p[(bt * r + k) * C + c];
      d_am[bt * C + c] = acc;
    }
  }
}

// d lm[b,s,:] = sum over (t,k) with ranges[b,t,k] == s of g_lm_p[b,t,k,:], in increasing (t,k) order.
// One WAVE per (b,s): pass 1 lists the matching rows in LDS with wave ballots (ordered, no barrier), pass 2
// sums them 16 bytes per lane -- deterministic, no atomics, arbitrary `ranges` (not only the monotone bands
// get_rnnt_prune_ranges produces).
template <bool VEC>
__global__ void do_pruning_bwd_lm_kernel(const float* __restrict__ g_lm_p, const int32_t* __restrict__ ranges,
                                         float* __restrict__ d_lm, int T, int S1, int C, int r) {
  extern __shared__ int lds_list[];   // [T*r] matching row indices of this wave
  const int s = blockIdx.x, b = blockIdx.y;
  const int n = T * r;
  const int32_t* rg = ranges + (size_t)b * n;
  const int lane = threadIdx.x;
  int count = 0;
  const unsigned long long lt = (1ull << lane) - 1ull;
  // 4 candidates per lane and pass (16-byte loads where the row of `ranges` allows), 8 passes' loads issued
  // back to back before any of them is consumed (the scan is otherwise a chain of L2 round trips); the list
  // is kept in index order
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(rg) & 15) == 0);
  constexpr int UN = 8;
  for (int base = 0; base < n; base += 256 * UN) {
    int v[UN][4];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i0 = base + 256 * u + 4 * lane;
      if (vec_ok && i0 + 3 < n) {
        const int4 q = *reinterpret_cast<const int4*>(rg + i0);
        v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = (i0 + e < n) ? rg[i0 + e] : -1;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i0 = base + 256 * u + 4 * lane;
      bool hit[4];
      int before = 0, total = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hit[e] = (v[u][e] == s);
        const unsigned long long m = __ballot(hit[e]);
        before += __popcll(m & lt);
        total += __popcll(m);
      }
      if (total != 0) {   // wave-uniform
        int pos = count + before;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (hit[e]) lds_list[pos++] = i0 + e;
        count += total;
      }
    }
  }
  __syncthreads();   // one wave per block: orders the LDS list writes before the reads below
  const float* gb = g_lm_p + (size_t)b * n * C;
  float* out = d_lm + ((size_t)b * S1 + s) * C;
  if (VEC) {
    const int n4 = C >> 2;
    // two column quads per lane (c, c + 64) x eight rows = 16 independent 16-byte loads in flight per lane; the
    // eight partial sums are combined in a fixed order, so the result is deterministic
    for (int c = lane; c < n4; c += 128) {
      const int c2 = c + 64;
      const bool two = c2 < n4;
      const f4 z = {0.f, 0.f, 0.f, 0.f};
      f4 a[4] = {z, z, z, z}, d[4] = {z, z, z, z};
      int j = 0;
      for (; j + 7 < count; j += 8) {
        f4 x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const f4u* row = reinterpret_cast<const f4u*>(gb + (size_t)lds_list[j + u] * C);
          x[u] = row[c];
          y[u] = two ? row[c2] : z;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[u] += x[u]; d[u] += y[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[u] += x[u + 4]; d[u] += y[u + 4];
        }
      }
      for (; j < count; ++j) {
        const f4u* row = reinterpret_cast<const f4u*>(gb + (size_t)lds_list[j] * C);
        a[0] += row[c];
        if (two) d[0] += row[c2];
      }
      reinterpret_cast<f4u*>(out)[c] = (a[0] + a[1]) + (a[2] + a[3]);
      if (two) reinterpret_cast<f4u*>(out)[c2] = (d[0] + d[1]) + (d[2] + d[3]);
    }
  } else {
    for (int c = lane; c < C; c += 64) {
      float acc = 0.f;
      for (int j = 0; j < count; ++j) acc += gb[(size_t)lds_list[j] * C + c];
      out[c] = acc;
    }
  }
}


// ---- segmented backward of the prune gather (16-byte path, C % 4 == 0, r <= 16): g is streamed exactly once.
// Pass 1, one block per (utterance, SEG consecutive frames): the SEG*r rows of the segment are contiguous in memory; thread i
// owns the column quads i, i+128, ... and walks the frames in order, two batches of frames in flight.
// get_rnnt_prune_ranges produces bands -- ranges[b,t,k] = base[b,t] + k with base non-decreasing in t and no lattice row
// skipped -- so a thread keeps the r lattice rows of the current frame in REGISTERS (acc[k] = row base + k; no LDS, no
// atomics, no barriers after the prologue, fixed order): a frame with the same base adds into them; when the base moves up
// by `step`, the rows below it have received everything this segment has for them and are FLUSHED, the registers shift down.
// A flushed row goes straight into d lm when no other segment touches it (row > the top row of the frame in front of the
// segment and < the base of the frame behind it), otherwise into one of at most 2r partial rows of the segment
// (`partial` [B][nseg][2r][C]: slots [0,r) rows shared with earlier segments, [r,2r) rows shared with later ones only).
// When the two incoming gradients are the same tensor (the joiner starts with am_pruned + lm_pruned, so autograd hands the
// same buffer to both) the sum over k that gives d am is taken from the same registers (FUSE).
// No loop of the streaming part contains a memory instruction under a data-dependent trip count (every flush is R stores under
// uniform branches), so that the compiler keeps counted s_waitcnt vmcnt(n) and the second batch stays in flight.
// A segment whose rows are not such a band (arbitrary caller data) is marked irregular and left to pass 2.
// meta[b][seg] = {lowest row, highest row, top row of the frame in front (or INT_MAX: irregular), base of the frame behind}.
// Pass 2, one block per (b,s): nothing to do for a row that one segment wrote directly; otherwise it adds the partial rows
// in segment order, or -- irregular segments, or segments that disagree about who owns the row -- rescans the ranges of the
// segments that touch the row and adds the matching rows of g directly.
constexpr int kSegIrregular = INT_MAX;
#ifndef FTR_SEG_ROWS
#define FTR_SEG_ROWS 16
#endif

template <int R, bool FUSE>
__global__ __launch_bounds__(128) void do_pruning_bwd_seg_kernel(
    const float* __restrict__ g_lm_p, const int32_t* __restrict__ ranges, float* __restrict__ d_am, float* __restrict__ d_lm,
    float* __restrict__ partial, int4* __restrict__ meta, int T, int S1, int C, int SEG) {
  extern __shared__ int bs[];      // base[SEG]
  __shared__ int red[8];
  constexpr int r = R;
  const int seg = blockIdx.x, b = blockIdx.y, nseg = gridDim.x;   // (last segments first, as lse_rows_reg_kernel walks its rows, was measured: 93 - 97 us either way)
  const int t0 = seg * SEG;
  const int nf = min(SEG, T - t0);
  const int nrows = nf * r;
  const int32_t* rg = ranges + ((size_t)b * T + t0) * r;
  // the segment's ranges: lowest / highest row, and is it a band?
  int lo = INT_MAX, hi = INT_MIN, bad = 0;
  for (int i = threadIdx.x; i < nrows; i += 128) {
    const int v = rg[i];
    const int f = i / r, k = i - f * r;
    const int v0 = rg[f * r];
    lo = min(lo, v); hi = max(hi, v);
    bad |= (v != v0 + k) | (v < 0) | (v >= S1);
    if (k == 0) {
      bs[f] = v;
      if (f + 1 < nf) { const int nx = rg[(f + 1) * r]; bad |= (nx < v) | (nx > v + r); }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
    bad |= __shfl_xor(bad, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; red[3 * w] = lo; red[3 * w + 1] = hi; red[3 * w + 2] = bad; }
  if (threadIdx.x == 127) {   // the neighbours (wave 1 writes red[6..7], read after the barrier like the rest)
    red[6] = t0 > 0 ? rg[-r] + r - 1 : -1;
    red[7] = t0 + nf < T ? rg[nrows] : INT_MAX - 1;
  }
  __syncthreads();
  const int smin = min(red[0], red[3]), smax = max(red[1], red[4]);
  const int prevtop = red[6], nextbase = red[7];
  // a band whose neighbours continue it: at most r rows shared with each side
  const bool ok = !(red[2] | red[5]) && prevtop - (r - 1) <= smin && nextbase >= smax - (r - 1);
  if (threadIdx.x == 0)
    meta[(size_t)b * nseg + seg] = make_int4(min(smin, smax), max(smin, smax), ok ? prevtop : kSegIrregular, nextbase);
  if (!ok && !FUSE) return;
  const int n4 = C >> 2;
  const size_t row0 = ((size_t)b * T + t0) * r;
  float* pseg = partial + ((size_t)b * nseg + seg) * (size_t)(2 * r) * C;
  float* dlm_b = d_lm + (size_t)b * S1 * C;
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  constexpr int FB = (FTR_SEG_ROWS + R - 1) / R;    // frames per batch: about 16 rows; two batches in flight
  for (int cb = 0; cb < n4; cb += 128) {
    const int c4 = cb + (int)threadIdx.x;
    const bool live = c4 < n4;
    const int cc = live ? c4 : n4 - 1;                       // clamped: every load is issued, what it brings is not stored
    const f4u* g = reinterpret_cast<const f4u*>(g_lm_p + row0 * C) + cc;   // row i at g[i * n4]
    f4 acc[R];                                               // lattice rows cur .. cur + R - 1
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = z;
    int cur = smin;
    auto flush = [&](int j) {                                // row cur + j is complete as far as this segment goes
      const int row = cur + j;
      // one store instruction per destination buffer (a pointer chosen by ?: becomes a flat access through a pointer table)
      const int slot = row <= prevtop ? row - smin : r + row - nextbase;
      if (row <= prevtop || row >= nextbase) {
        if (live) reinterpret_cast<f4u*>(pseg + (size_t)slot * C)[c4] = acc[j];
      } else {
        if (live) reinterpret_cast<f4u*>(dlm_b + (size_t)row * C)[c4] = acc[j];
      }
    };
    auto load = [&](f4 (&v)[FB * R], int f0) {
#pragma unroll
      for (int u = 0; u < FB * R; ++u) v[u] = g[(size_t)min(f0 * R + u, nrows - 1) * n4];
    };
    auto consume = [&](const f4 (&v)[FB * R], int f0) {
#pragma unroll
      for (int q = 0; q < FB; ++q) {
        const int f = f0 + q;
        if (f < nf) {
          if (ok) {
            const int step = __builtin_amdgcn_readfirstlane(bs[f]) - cur;   // 0 ... R in a band
            if (step > 0) {
#pragma unroll
              for (int j = 0; j < R; ++j)
                if (j < step) flush(j);
              for (int sft = 0; sft < step; ++sft) {          // registers only: no memory instruction in this loop
#pragma unroll
                for (int k = 0; k + 1 < R; ++k) acc[k] = acc[k + 1];
                acc[R - 1] = z;
              }
              cur += step;
            }
#pragma unroll
            for (int k = 0; k < R; ++k) acc[k] += v[q * R + k];
          }
          if (FUSE) {
            f4 asum = v[q * R];
#pragma unroll
            for (int k = 1; k < R; ++k) asum += v[q * R + k];
            if (live) reinterpret_cast<f4u*>(d_am + ((size_t)b * T + t0 + f) * C)[c4] = asum;
          }
        }
      }
    };
    f4 va[FB * R], vb[FB * R];
    load(va, 0);
    for (int f0 = 0; f0 < nf; f0 += 2 * FB) {
      load(vb, f0 + FB);
      consume(va, f0);
      load(va, f0 + 2 * FB);
      consume(vb, f0 + FB);
    }
    if (ok) {
#pragma unroll
      for (int j = 0; j < R; ++j) flush(j);
    }
  }
}

// items of the reduction list: bit 31 clear = row index into `partial`, bit 31 set = row index into g_lm_p
constexpr int RED_CAP = 512;
#ifndef FTR_RED_WAVES
#define FTR_RED_WAVES 2
#endif
constexpr int RED_WAVES = FTR_RED_WAVES;   // 1, 2 or 4 (c3 / c4 / c5 with 128-frame segments: 8.4 / 16.0 / 27.4, 8.6 / 15.1 / 22.9, 11.0 / 18.2 / 21.8 us)
// One block of RED_WAVES waves per (b,s).  Wave 0 lists the segments that touch the row with ballots and decides: nobody -> zeros;
// one band segment that owns the row -> pass 1 has written it; band segments that all hold a partial row for it -> those, in
// segment order; anything else (an irregular segment, or band segments that disagree about the owner: only caller data that is
// not a band does that) -> the ranges of the touching segments are rescanned, 128 rows per pass, and the matching rows of g are
// the items.  Then wave w adds the items w, w + RED_WAVES, ... (four loads in flight) and the sums are combined as
// w0 + w1 (or (w0 + w1) + (w2 + w3)): a fixed order, so the result does not depend on scheduling.
__global__ __launch_bounds__(64 * RED_WAVES) void do_pruning_bwd_reduce_kernel(
    const float* __restrict__ g_lm_p, const int32_t* __restrict__ ranges, const float* __restrict__ partial,
    const int4* __restrict__ meta, float* __restrict__ d_lm, int T, int S1, int C, int r, int SEG, int nseg) {
  extern __shared__ __attribute__((aligned(16))) int dyn[];   // hits [2 * nseg], then (16-byte aligned) sums [RED_WAVES - 1][256] quads
  __shared__ unsigned items[RED_CAP];
  __shared__ int ctl[4];           // nh (or -1: nothing to do), hi, j0, cnt of the batch
  int* hits = dyn;                 // segment id, partial slot of the row (or -1: rescan the segment)
  f4* sums = reinterpret_cast<f4*>(dyn + ((2 * nseg + 3) & ~3));
  const int s = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int n4 = C >> 2;
  if (wv == 0) {
    const int4* mb = meta + (size_t)b * nseg;
    int nh = 0, ndirect = 0, nresc = 0;
    for (int c0 = 0; c0 < nseg; c0 += 64) {
      const int sg = c0 + lane;
      bool hit = false, direct = false, resc = false;
      int slot = -1;
      if (sg < nseg) {
        const int4 m = mb[sg];
        hit = s >= m.x && s <= m.y;
        if (hit) {
          if (m.z == kSegIrregular) resc = true;
          else if (s <= m.z) slot = s - m.x;
          else if (s >= m.w) slot = r + s - m.w;
          else direct = true;
        }
      }
      const unsigned long long mask = __ballot(hit);
      if (hit) {
        const int pos = nh + __popcll(mask & lt);
        hits[2 * pos] = sg; hits[2 * pos + 1] = slot;
      }
      nh += __popcll(mask);
      ndirect += __popcll(__ballot(direct));
      nresc += __popcll(__ballot(resc));
    }
    const bool done = nh == 1 && ndirect == 1;                 // pass 1 wrote the row
    const bool rescan = !done && (nresc > 0 || ndirect > 0);   // not (only) bands: take every touching segment from g
    if (rescan)
      for (int i = lane; i < nh; i += 64) hits[2 * i + 1] = -1;
    if (lane == 0) { ctl[0] = done ? -1 : nh; ctl[1] = 0; ctl[2] = 0; }
  }
  __syncthreads();
  const int nh = ctl[0];
  if (nh < 0) return;
  float* out = d_lm + ((size_t)b * S1 + s) * C;
  f4 tot[4];                       // wave 0's lanes: the row's running total over the batches (4 column quads per lane and sweep)
  for (int cb = 0; cb < n4; cb += 256) {
#pragma unroll
    for (int q = 0; q < 4; ++q) tot[q] = f4{0.f, 0.f, 0.f, 0.f};
    if (wv == 0 && lane == 0) { ctl[1] = 0; ctl[2] = 0; }
    __syncthreads();
    while (true) {
      // ---- wave 0: the next batch of items
      if (wv == 0) {
        int hi = ctl[1], j0 = ctl[2], cnt = 0;
        while (hi < nh && cnt <= RED_CAP - 128) {
          const int sg = hits[2 * hi], slot = hits[2 * hi + 1];
          if (slot >= 0) {
            if (lane == 0) items[cnt] = (unsigned)(((size_t)b * nseg + sg) * (2 * r) + slot);
            ++cnt; ++hi;
          } else {
            const int t0 = sg * SEG;
            const int nrows = min(SEG, T - t0) * r;
            const size_t row0 = ((size_t)b * T + t0) * r;
            while (j0 < nrows && cnt <= RED_CAP - 128) {
              const int ja = j0 + lane, jb = j0 + 64 + lane;          // two independent loads per pass
              const int va = ja < nrows ? ranges[row0 + ja] : -1;
              const int vb = jb < nrows ? ranges[row0 + jb] : -1;
              const unsigned long long ma = __ballot(va == s), mk = __ballot(vb == s);
              if (va == s) items[cnt + __popcll(ma & lt)] = 0x80000000u | (unsigned)(row0 + ja);
              cnt += __popcll(ma);
              if (vb == s) items[cnt + __popcll(mk & lt)] = 0x80000000u | (unsigned)(row0 + jb);
              cnt += __popcll(mk);
              j0 += 128;
            }
            if (j0 >= nrows) { ++hi; j0 = 0; }
          }
        }
        if (lane == 0) { ctl[1] = hi; ctl[2] = j0; ctl[3] = cnt; }
      }
      __syncthreads();
      const int cnt = ctl[3];
      const bool last = ctl[1] >= nh;
      // ---- every wave: its share of the batch
      f4 acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
      int i = wv;
      for (; i + 3 * RED_WAVES < cnt; i += 4 * RED_WAVES) {
        const float* src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned it = items[i + u * RED_WAVES];
          src[u] = (it & 0x80000000u) ? g_lm_p + (size_t)(it & 0x7fffffffu) * C : partial + (size_t)it * C;
        }
        f4 v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c4 = cb + lane + 64 * q;
            v[u][q] = (c4 < n4) ? (f4)reinterpret_cast<const f4u*>(src[u])[c4] : f4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] += v[u][q];
      }
      for (; i < cnt; i += RED_WAVES) {
        const unsigned it = items[i];
        const float* src = (it & 0x80000000u) ? g_lm_p + (size_t)(it & 0x7fffffffu) * C : partial + (size_t)it * C;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c4 = cb + lane + 64 * q;
          if (c4 < n4) acc[q] += reinterpret_cast<const f4u*>(src)[c4];
        }
      }
      if (RED_WAVES > 1 && cnt > 1) {   // uniform over the block: a single item is wave 0's alone
        if (wv != 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c4 = lane + 64 * q;
            if (cb + c4 < n4) sums[(size_t)(wv - 1) * 256 + c4] = acc[q];
          }
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c4 = lane + 64 * q;
            if (cb + c4 < n4) tot[q] += RED_WAVES == 4 ? (acc[q] + sums[c4]) + (sums[256 + c4] + sums[512 + c4]) : acc[q] + sums[c4];
          }
        }
      } else if (wv == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) tot[q] += acc[q];
      }
      if (last) break;
      __syncthreads();             // the batch's items and sums are consumed: wave 0 may overwrite them
    }
    if (wv == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c4 = cb + lane + 64 * q;
        if (c4 < n4) reinterpret_cast<f4u*>(out)[c4] = tot[q];
      }
    }
    __syncthreads();
  }
}

// frames per segment: a block per CU and more (two 16-row batches in flight per thread keep a wave busy: measured at c3, pass 1
// takes 80 - 83 us with anything from 4 to 128 frames per segment, i.e. 8000 to 256 blocks), as long as possible otherwise:
// every segment costs up to 2r partial rows where the band is flat, and pass 2 one addition per segment and shared row
// (c3: 71 / 36 / 22 / 11 us with 4 / 8 / 16 / 128 frames)
inline int seg_frames(int B, int T) {
  const char* e = getenv("FTR_PRUNE_SEG");   // study / test knob, read per call (one getenv per launch)
  const int forced = e ? atoi(e) : 0;
  if (forced >= 4) return forced;
  const long long per = ((long long)B * T + 255) / 256;
  int seg = 16;
  while (seg < per && seg < 128) seg <<= 1;
  return seg;
}
constexpr int kSegMaxR = 16;   // widest band the register window is instantiated for; wider ones take the generic kernels

}  // namespace

int do_pruning_bwd(const float* g_am_p, const float* g_lm_p, const int32_t* ranges, float* d_am, float* d_lm, int B,
                   int T, int S1, int C, int r, hipStream_t st) {
  if ((size_t)B * T * C == 0) return FTR_OK;
  const int threads = 256;
  if ((C & 3) == 0) {
    const size_t total = (size_t)B * T * (C >> 2);
    hipLaunchKernelGGL(do_pruning_bwd_am_kernel<true>, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0, st, g_am_p, d_am, C, r, total);
  } else {
    const size_t total = (size_t)B * T * C;
    hipLaunchKernelGGL(do_pruning_bwd_am_kernel<false>, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0, st, g_am_p, d_am, C, r, total);
  }
  int rc = check_launch("do_pruning_bwd_am");
  if (rc != FTR_OK) return rc;
  const size_t lds = sizeof(int) * (size_t)T * r;   // worst case: every (t,k) selects the same row
  if (lds > 150 * 1024) { set_error("do_pruning_bwd: T*s_range = %d too large for the LDS row list", T * r); return FTR_ERR_UNSUPPORTED; }
  const bool vec = (C & 3) == 0;
  if (lds > 64 * 1024) {
    hipError_t e = vec ? hipFuncSetAttribute(reinterpret_cast<const void*>(do_pruning_bwd_lm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                       : hipFuncSetAttribute(reinterpret_cast<const void*>(do_pruning_bwd_lm_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("do_pruning_bwd: cannot reserve LDS: %s", hipGetErrorString(e)); return FTR_ERR_LAUNCH; }
  }
  if (vec) hipLaunchKernelGGL(do_pruning_bwd_lm_kernel<true>, dim3(S1, B), dim3(64), lds, st, g_lm_p, ranges, d_lm, T, S1, C, r);
  else hipLaunchKernelGGL(do_pruning_bwd_lm_kernel<false>, dim3(S1, B), dim3(64), lds, st, g_lm_p, ranges, d_lm, T, S1, C, r);
  return check_launch("do_pruning_bwd_lm");
}

int cummin_i32(const int32_t* in, int32_t* out, int rows, int cols, hipStream_t st) {
  if (rows == 0 || cols == 0) return FTR_OK;
  const int waves_per_block = 4;
  hipLaunchKernelGGL(cummin_kernel, dim3((rows + waves_per_block - 1) / waves_per_block),
                     dim3(64 * waves_per_block), 0, st, in, out, rows, cols);
  return check_launch("cummin_i32");
}

int prune_ranges(const float* px_grad, const float* py_grad, const int32_t* boundary, int32_t* ranges,
                 int32_t* s_begin, int B, int S, int T, int T1, int r, hipStream_t st) {
  if (B == 0 || T == 0) return FTR_OK;
  const int threads = 64;  // small blocks: B*T threads is only ~32k at the headline shape, spread them
  const dim3 grid((T + threads - 1) / threads, B);
#define FTR_ARGMAX_ONCE(R) case R: hipLaunchKernelGGL(prune_argmax_once_kernel<R>, grid, dim3(threads), 0, st, px_grad, py_grad, boundary, s_begin, B, S, T, T1); break;
  const int nwin = S + 1 - r + 1;
  static const bool split_off = getenv("FTR_PRUNE_ARGMAX_SPLIT") && !strcmp(getenv("FTR_PRUNE_ARGMAX_SPLIT"), "0");   // A/B knob
  // ... where one wave per 64 columns leaves the chip short of waves (c3: 512 of them on 256 CUs, 41.5 -> 34 us); with a
  // thousand and more the one-wave kernel already keeps 8 waves per CU loading and the split only costs (c4 67 -> 109 us,
  // c5 191 -> 436)
  if (nwin > ftr::kSplitQ && (size_t)((T + 63) / 64) * B <= 768 && !split_off) {
    const int nc = (nwin + ftr::kSplitQ - 1) / ftr::kSplitQ;
    const int nw = nc < 8 ? nc : 8;
    hipLaunchKernelGGL(prune_argmax_split_kernel, dim3((T + 63) / 64, B), dim3(64 * nw), 0, st, px_grad, py_grad, boundary, s_begin, B, S, T, T1, r);
  } else
  switch (r) {   // window lengths up to 16: py_grad is loaded once; longer windows: the generic kernel
    FTR_ARGMAX_ONCE(1) FTR_ARGMAX_ONCE(2) FTR_ARGMAX_ONCE(3) FTR_ARGMAX_ONCE(4) FTR_ARGMAX_ONCE(5) FTR_ARGMAX_ONCE(6)
    FTR_ARGMAX_ONCE(7) FTR_ARGMAX_ONCE(8) FTR_ARGMAX_ONCE(9) FTR_ARGMAX_ONCE(10) FTR_ARGMAX_ONCE(11) FTR_ARGMAX_ONCE(12)
    FTR_ARGMAX_ONCE(13) FTR_ARGMAX_ONCE(14) FTR_ARGMAX_ONCE(15) FTR_ARGMAX_ONCE(16)
    default: hipLaunchKernelGGL(prune_argmax_kernel, grid, dim3(threads), 0, st, px_grad, py_grad, boundary, s_begin, B, S, T, T1, r);
  }
#undef FTR_ARGMAX_ONCE
  int rc = check_launch("prune_argmax");
  if (rc != FTR_OK) return rc;
  const int r_con = (T1 == T) ? 2 : r;  // rnnt_loss.py:756
  hipLaunchKernelGGL(prune_adjust_kernel, dim3(B), dim3(64), 0, st, s_begin, ranges, B, T, r, r_con);
  return check_launch("prune_adjust");
}

int do_pruning(const float* am, const float* lm, const int32_t* ranges, float* am_p, float* lm_p, int B, int T, int S1,
               int C, int r, hipStream_t st) {
  const size_t frames = (size_t)B * T;
  if (frames == 0 || C == 0 || r == 0) return FTR_OK;
  if (frames > 0x7fffffffull) { set_error("do_pruning: B*T = %zu exceeds the grid limit", frames); return FTR_ERR_UNSUPPORTED; }
  if ((C & 3) == 0) hipLaunchKernelGGL(do_pruning_kernel<true>, dim3((unsigned)frames), dim3(128), 0, st, am, lm, ranges, am_p, lm_p, T, S1, C, r);
  else hipLaunchKernelGGL(do_pruning_kernel<false>, dim3((unsigned)frames), dim3(128), 0, st, am, lm, ranges, am_p, lm_p, T, S1, C, r);
  return check_launch("do_pruning");
}

size_t do_pruning_bwd_workspace_bytes(int B, int T, int S1, int C, int r) {
  if ((C & 3) != 0 || (size_t)B * T * C == 0 || r <= 0) return 0;
  const int seg = ftr::seg_frames(B, T);
  const size_t nseg = (size_t)(T + seg - 1) / seg;
  const size_t partial = sizeof(float) * (size_t)B * nseg * (2 * (size_t)r) * C;
  return partial + sizeof(int4) * (size_t)B * nseg;
}

int do_pruning_bwd_ws(const float* g_am_p, const float* g_lm_p, const int32_t* ranges, float* d_am, float* d_lm,
                      int B, int T, int S1, int C, int r, void* ws, size_t ws_bytes, hipStream_t st) {
  if ((size_t)B * T * C == 0) return FTR_OK;
  if ((C & 3) != 0 || r <= 0) return do_pruning_bwd(g_am_p, g_lm_p, ranges, d_am, d_lm, B, T, S1, C, r, st);
  const size_t need = do_pruning_bwd_workspace_bytes(B, T, S1, C, r);
  if (!ws || ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 15) != 0) {
    set_error("do_pruning_bwd_ws: workspace of %zu bytes (16-byte aligned) required, got %zu", need, ws_bytes);
    return FTR_ERR_INVALID_ARG;
  }
  const int seg = ftr::seg_frames(B, T);
  const int nseg = (T + seg - 1) / seg;
  float* partial = reinterpret_cast<float*>(ws);
  int4* meta = reinterpret_cast<int4*>(partial + (size_t)B * nseg * (2 * (size_t)r) * C);
  const size_t lds = sizeof(int) * (size_t)seg;
  if (r > ftr::kSegMaxR || lds > 60 * 1024 || (size_t)B * nseg * (2 * (size_t)r) >= 0x7fffffffull || (size_t)B * T * r >= 0x7fffffffull)
    return do_pruning_bwd(g_am_p, g_lm_p, ranges, d_am, d_lm, B, T, S1, C, r, st);
  const bool fuse = (g_am_p == g_lm_p);
  if (!fuse) {
    const size_t total = (size_t)B * T * (C >> 2);
    hipLaunchKernelGGL(ftr::do_pruning_bwd_am_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g_am_p, d_am, C, r, total);
    int rc = check_launch("do_pruning_bwd_am");
    if (rc != FTR_OK) return rc;
  }
#define FTR_SEG(R) case R: \
    if (fuse) hipLaunchKernelGGL((ftr::do_pruning_bwd_seg_kernel<R, true>), dim3(nseg, B), dim3(128), lds, st, g_lm_p, ranges, d_am, d_lm, partial, meta, T, S1, C, seg); \
    else hipLaunchKernelGGL((ftr::do_pruning_bwd_seg_kernel<R, false>), dim3(nseg, B), dim3(128), lds, st, g_lm_p, ranges, d_am, d_lm, partial, meta, T, S1, C, seg); \
    break;
  switch (r) {
    FTR_SEG(1) FTR_SEG(2) FTR_SEG(3) FTR_SEG(4) FTR_SEG(5) FTR_SEG(6) FTR_SEG(7) FTR_SEG(8)
    FTR_SEG(9) FTR_SEG(10) FTR_SEG(11) FTR_SEG(12) FTR_SEG(13) FTR_SEG(14) FTR_SEG(15) FTR_SEG(16)
  }
#undef FTR_SEG
  int rc = check_launch("do_pruning_bwd_seg");
  if (rc != FTR_OK) return rc;
  hipLaunchKernelGGL(ftr::do_pruning_bwd_reduce_kernel, dim3(S1, B), dim3(64 * ftr::RED_WAVES),
                     sizeof(int) * (size_t)((2 * nseg + 3) & ~3) + 3 * 256 * sizeof(float) * 4, st, g_lm_p, ranges, partial, meta, d_lm, T, S1, C, r, seg, nseg);
  return check_launch("do_pruning_bwd_reduce");
}

}  // namespace ftr
