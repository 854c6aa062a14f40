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
  for (int s0 = 0; s0 < nwin; ++s0) {
    const float blk = lead - lag;                                        // rnnt_loss.py:725
    const float pxp = (s0 == 0) ? 0.0f : pxc[(size_t)(s0 - 1) * T1];     // :726-727
    const float fin = blk - pxp;                                         // :728
    if (s0 == 0 || fin > bestv) { best = s0; bestv = fin; }              // :729, first maximum
    if (s0 + 1 < nwin) {
      lead = lead + pyc[(size_t)(s0 + r) * T];
      lag = lag + pyc[(size_t)s0 * T];
    }
  }
  const int se = boundary[4 * b + 2], te = boundary[4 * b + 3];
  int pad = se - r + 1;                                                  // :744-746
  if (pad < 0) pad = 0;
  s_begin[(size_t)b * T + t] = (t < te - 1) ? best : pad;                // :741-748
}

// one wave per utterance: two right-to-left min-scans over T with a register carry.
__global__ void prune_adjust_kernel(int32_t* __restrict__ s_begin, int32_t* __restrict__ ranges, int B,
                                    int T, int r_out, int r_con) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  int32_t* x = s_begin + (size_t)b * T;
  const int nblk = (T + 63) / 64;
  // pass 1: suffix-min (rnnt_loss.py:628), then x = -(x - (r-1) t) (:630-632)
  int carry = INT_MAX;
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int t = blk * 64 + (63 - lane);  // lane 0 holds the right-most column of the block
    int v = (t < T) ? x[t] : INT_MAX;
    v = wave_incl_min_scan(v, lane);
    v = min(v, carry);
    carry = __shfl(v, 63, 64);
    if (t < T) x[t] = -(v - (r_con - 1) * t);
  }
  // pass 2 re-reads, in each lane, exactly the elements that lane wrote in pass 1 (same t mapping)
  // pass 2: suffix-min (:634), clip at 0 (:636), transform back (:638-640), write ranges (:758-759)
  carry = INT_MAX;
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int t = blk * 64 + (63 - lane);
    int v = (t < T) ? x[t] : INT_MAX;
    v = wave_incl_min_scan(v, lane);
    v = min(v, carry);
    carry = __shfl(v, 63, 64);
    if (t < T) {
      const int z = max(v, 0);
      const int sb = -(z - (r_con - 1) * t);
      int32_t* dst = ranges + ((size_t)b * T + t) * r_out;
      for (int k = 0; k < r_out; ++k) dst[k] = sb + k;
    }
  }
}

// one thread per 16 bytes of output row (vector path, C % 4 == 0) or per element (scalar path)
template <bool VEC>
__global__ void do_pruning_kernel(const float* __restrict__ am, const float* __restrict__ lm,
                                  const int32_t* __restrict__ ranges, float* __restrict__ am_p,
                                  float* __restrict__ lm_p, int T, int S1, int C, int r, size_t total) {
  const int per_row = VEC ? (C >> 2) : C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / per_row;           // (b*T + t)*r + k
    const int c = (int)(i - row * per_row);
    const size_t bt = row / r;                // b*T + t
    const size_t b = bt / T;
    const int s = ranges[row];
    if (VEC) {
      const f4 a = reinterpret_cast<const f4u*>(am + bt * C)[c];
      const f4 l = reinterpret_cast<const f4u*>(lm + (b * S1 + s) * C)[c];
      reinterpret_cast<f4u*>(am_p + row * C)[c] = a;
      reinterpret_cast<f4u*>(lm_p + row * C)[c] = l;
    } else {
      am_p[row * C + c] = am[bt * C + c];
      lm_p[row * C + c] = lm[(b * S1 + s) * C + c];
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
    for (int c = lane; c < n4; c += 64) {
      // four independent partial sums (rows j = 0,1,2,3 mod 4) keep four loads in flight; they are combined in a
      // fixed order, so the result is still deterministic
      f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
      int j = 0;
      for (; j + 3 < count; j += 4) {
        a0 += reinterpret_cast<const f4u*>(gb + (size_t)lds_list[j] * C)[c];
        a1 += reinterpret_cast<const f4u*>(gb + (size_t)lds_list[j + 1] * C)[c];
        a2 += reinterpret_cast<const f4u*>(gb + (size_t)lds_list[j + 2] * C)[c];
        a3 += reinterpret_cast<const f4u*>(gb + (size_t)lds_list[j + 3] * C)[c];
      }
      for (; j < count; ++j) a0 += reinterpret_cast<const f4u*>(gb + (size_t)lds_list[j] * C)[c];
      reinterpret_cast<f4u*>(out)[c] = (a0 + a1) + (a2 + a3);
    }
  } else {
    for (int c = lane; c < C; c += 64) {
      float acc = 0.f;
      for (int j = 0; j < count; ++j) acc += gb[(size_t)lds_list[j] * C + c];
      out[c] = acc;
    }
  }
}

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
  hipLaunchKernelGGL(prune_argmax_kernel, dim3((T + threads - 1) / threads, B), dim3(threads), 0, st,
                     px_grad, py_grad, boundary, s_begin, B, S, T, T1, r);
  int rc = check_launch("prune_argmax");
  if (rc != FTR_OK) return rc;
  const int r_con = (T1 == T) ? 2 : r;  // rnnt_loss.py:756
  hipLaunchKernelGGL(prune_adjust_kernel, dim3(B), dim3(64), 0, st, s_begin, ranges, B, T, r, r_con);
  return check_launch("prune_adjust");
}

int do_pruning(const float* am, const float* lm, const int32_t* ranges, float* am_p, float* lm_p, int B,
               int T, int S1, int C, int r, hipStream_t st) {
  const size_t rows = (size_t)B * T * r;
  if (rows == 0 || C == 0) return FTR_OK;
  const int threads = 256;
  if ((C & 3) == 0) {
    const size_t total = rows * (size_t)(C >> 2);
    const size_t blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(do_pruning_kernel<true>, dim3((unsigned)(blocks > 0x7fffffffull ? 0x7fffffffull : blocks)), dim3(threads), 0, st,
                       am, lm, ranges, am_p, lm_p, T, S1, C, r, total);
  } else {
    const size_t total = rows * (size_t)C;
    const size_t blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(do_pruning_kernel<false>, dim3((unsigned)(blocks > 0x7fffffffull ? 0x7fffffffull : blocks)), dim3(threads), 0, st,
                       am, lm, ranges, am_p, lm_p, T, S1, C, r, total);
  }
  return check_launch("do_pruning");
}

}  // namespace ftr
