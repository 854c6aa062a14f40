// csrc/pruned_logprobs.hip -- pruned joiner log-probs, forward and backward, gfx950.
// Replaces get_rnnt_logprobs_pruned (+ _roll_by_shifts, fix_for_boundary and the delay-penalty block)
// of /root/reference/tf_fast_rnnt/python/tf_fast_rnnt/rnnt_loss.py:853-1020, 814-851, 28-61, 1097-1114
// and what TensorFlow autodiff replays for them in the backward pass.
//   lse_rows_kernel        :942  reduce_logsumexp over C, one wave per (b,t,k) row, 16-byte loads
//   band_to_lattice_kernel :943-1016 gathers + pad + roll + transpose + fix_for_boundary, one thread per
//                          lattice cell, coalesced along t: px/py are written exactly once, complete
//   band_grad_kernel       gradient w.r.t. logits: -(gx+gy) softmax + gx 1[sym] + gy 1[blank]
#include "ftr_common.h"

namespace ftr {
namespace {

__device__ __forceinline__ float wave_max(float v) { return wave_max_dpp(v); }
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }

// logsumexp of each row, the row held in registers (C % 4 == 0, C <= 256 * NQ): one wave per row, one 16-byte load per lane
// and quad, all issued before the first is used, a single pass over the data, one short-lived wave per row and as many waves
// as the chip holds.  (Rounds 1 - 2 ran this as 2048 persistent blocks, two rows per pass, the next pass's loads in flight
// while one is reduced: 73 us at c3 = 4.4 TB/s; scripts/probes/stream_probe.hip measures the plain form below at 55 us =
// 5.8 TB/s on the same tensor -- a flat read reaches 6.7 -- and 205 against 250 us at c4.  The memory system likes many short
// waves better than few clever ones.)
template <int NQ>
__global__ __launch_bounds__(256) void lse_rows_reg_kernel(const float* __restrict__ logits, float* __restrict__ lse,
                                                           size_t rows, int C) {
  const int lane = threadIdx.x & 63;
  // LAST ROWS FIRST.  The joiner has just written `logits` front to back, 320 MB at c3 against 256 MB of memory-side cache:
  // what is still in the cache is the tail.  Walking front to back misses the cache on the head AND pushes the dirty tail out
  // before it is read; walking back to front reads the tail from the cache: 75 -> 53 us inside the c3 step (the kernel alone,
  // on a tensor at rest, takes 55 us either way).
  const size_t rowi = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (rowi >= rows) return;
  const size_t row = rows - 1 - rowi;
  const int n4 = C >> 2;
  const f4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  const f4u* x4 = reinterpret_cast<const f4u*>(logits + row * C);
  f4 v[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = lane + 64 * q;
    v[q] = (i < n4) ? (f4)x4[i] : ninf;
  }
  float m = -INFINITY;
#pragma unroll
  for (int q = 0; q < NQ; ++q) m = fmaxf(fmaxf(m, fmaxf(v[q][0], v[q][1])), fmaxf(v[q][2], v[q][3]));
  m = wave_max(m);
  float sum = 0.0f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (lane + 64 * q < n4)   // same per-lane order as the two-pass kernel
      sum += __expf(v[q][0] - m) + __expf(v[q][1] - m) + __expf(v[q][2] - m) + __expf(v[q][3] - m);
  }
  sum = wave_sum(sum);
  if (lane == 0) lse[row] = m + __logf(sum);
}

// logsumexp of each row of length C; rows = B*T*r.  One wave per row (any C).
template <bool VEC>
__global__ void lse_rows_kernel(const float* __restrict__ logits, float* __restrict__ lse, size_t rows, int C) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = logits + row * C;
  float m = -INFINITY;
  if (VEC) {
    const f4u* x4 = reinterpret_cast<const f4u*>(x);
    const int n4 = C >> 2;
    for (int i = lane; i < n4; i += 64) { const f4 v = x4[i]; m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3])); }
    m = wave_max(m);
    float s = 0.0f;
    for (int i = lane; i < n4; i += 64) {  // second pass hits L1/L2: a row is 2-4 KB
      const f4 v = x4[i];
      s += __expf(v[0] - m) + __expf(v[1] - m) + __expf(v[2] - m) + __expf(v[3] - m);
    }
    s = wave_sum(s);
    if (lane == 0) lse[row] = m + __logf(s);
  } else {
    for (int i = lane; i < C; i += 64) m = fmaxf(m, x[i]);
    m = wave_max(m);
    float s = 0.0f;
    for (int i = lane; i < C; i += 64) s += __expf(x[i] - m);
    s = wave_sum(s);
    if (lane == 0) lse[row] = m + __logf(s);
  }
}

// grid: (ceil((T+1)/256), S+1, B); thread <-> (b, s, t).  Writes py[b,s,t] (t < T) and px[b,s,t] (s < S, t < T1).
template <bool MOD>
__global__ void band_to_lattice_kernel(const float* __restrict__ logits, const int32_t* __restrict__ symbols,
                                       const int32_t* __restrict__ ranges, const int32_t* __restrict__ boundary,
                                       const float* __restrict__ lse, int blank, double delay_penalty,
                                       float* __restrict__ px, float* __restrict__ py, int T, int S, int C, int r) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y, b = blockIdx.z;
  const int T1 = MOD ? T : T + 1;
  if (t >= T1 && t >= T) return;
  const int te = boundary ? boundary[4 * b + 3] : T;
  float vx = -INFINITY, vy = -INFINITY;
  if (t < T) {
    const size_t bt = (size_t)b * T + t;
    const int s0 = ranges[bt * r];
    int k = s - s0;                     // _roll_by_shifts: out[s] = padded[(s - s0) mod (S+1)]  (:849)
    if (k < 0) k += S + 1;
    if (k < r) {
      const size_t row = bt * r + k;
      const float l = lse[row];
      vy = logits[row * C + blank] - l;                       // :995-996
      if (s < S) vx = logits[row * C + min(max(symbols[(size_t)b * S + s], 0), C - 1)] - l;   // :961-965 (symbol kept in bounds)
    }
  }
  if (t < T) py[((size_t)b * (S + 1) + s) * T + t] = vy;
  if (s < S && t < T1) {
    if (!MOD && t == te) vx = -INFINITY;                      // fix_for_boundary (:1015-1016), px[:,:,T] (:984-993)
    if (delay_penalty > 0.0) {                                // :1097-1114, float64 then cast
      const double offset = ((double)te - 1.0) / 2.0;
      vx += (float)((offset - (double)t) * delay_penalty);
    }
    px[((size_t)b * S + s) * T1 + t] = vx;
  }
}

// one wave per (b,t,k) row of glogits.
template <bool MOD, bool VEC>
__global__ void band_grad_kernel(const float* __restrict__ logits, const int32_t* __restrict__ symbols,
                                 const int32_t* __restrict__ ranges, const int32_t* __restrict__ boundary,
                                 const float* __restrict__ lse, const float* __restrict__ gpx,
                                 const float* __restrict__ gpy, const Scale scale, int blank,
                                 float* __restrict__ glogits, size_t rows, int T, int S, int C, int r) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int T1 = MOD ? T : T + 1;
  const size_t bt = row / r;
  const int k = (int)(row - bt * r);
  const int b = (int)(bt / T);
  const int t = (int)(bt - (size_t)b * T);
  const int s0 = ranges[bt * r];
  int s = s0 + k;                        // inverse of the roll: band slot k <-> lattice row (s0 + k) mod (S+1)
  if (s > S) s -= S + 1;
  const int te = boundary ? boundary[4 * b + 3] : T;
  const float sc = scale.at(b);
  float gx = 0.0f;
  int sym = blank;
  const bool sok = s >= 0 && s <= S;     // ranges are caller data: a row outside the lattice gets no gradient
  if (sok && s < S) {
    sym = symbols[(size_t)b * S + s];
    if (MOD || t != te) gx = gpx[((size_t)b * S + s) * T1 + t] * sc;   // overwritten cells get no gradient
  }
  const float gy = sok ? gpy[((size_t)b * (S + 1) + s) * T + t] * sc : 0.0f;
  const float tot = gx + gy;
  const float l = lse[row];
  const float* x = logits + row * C;
  float* g = glogits + row * C;
  if (VEC) {
    const f4u* x4 = reinterpret_cast<const f4u*>(x);
    f4u* g4 = reinterpret_cast<f4u*>(g);
    const int n4 = C >> 2;
    for (int i = lane; i < n4; i += 64) {
      const f4 v = x4[i];
      f4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * i + e;
        float val = -tot * __expf(v[e] - l);
        if (c == sym) val += gx;
        if (c == blank) val += gy;
        o[e] = val;
      }
      g4[i] = o;
    }
  } else {
    for (int c = lane; c < C; c += 64) {
      float val = -tot * __expf(x[c] - l);
      if (c == sym) val += gx;
      if (c == blank) val += gy;
      g[c] = val;
    }
  }
}

// loss tail (rnnt_loss.py:333,544-546,1124-1126,1487-1489): out = -ans (reduction 0), -mean (1) or -sum (2) over the
// batch, one block, fixed summation tree (deterministic).
__global__ __launch_bounds__(256) void negated_reduce_kernel(const float* __restrict__ ans, int B, int reduction,
                                                             float* __restrict__ out) {
  __shared__ float red[4];
  if (reduction == 0) {
    for (int b = threadIdx.x; b < B; b += 256) out[b] = -ans[b];
    return;
  }
  float s = 0.0f;
  for (int b = threadIdx.x; b < B; b += 256) s += ans[b];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    out[0] = (reduction == 1) ? -(t / (float)B) : -t;
  }
}
}  // namespace

int negated_reduce(const float* ans, int B, int reduction, float* out, hipStream_t st) {
  hipLaunchKernelGGL(negated_reduce_kernel, dim3(1), dim3(256), 0, st, ans, B, reduction, out);
  return check_launch("negated_reduce");
}

// logsumexp over the last axis of [rows, C] (rnnt_loss.py:942): picks the register-resident kernel where it fits
int lse_rows(const float* logits, float* lse, size_t rows, int C, hipStream_t st) {
  if (rows == 0) return FTR_OK;
  const int wpb = 4;
  const unsigned blocks = (unsigned)((rows + wpb - 1) / wpb);
  if ((C & 3) == 0 && C <= 256) hipLaunchKernelGGL((lse_rows_reg_kernel<1>), dim3(blocks), dim3(64 * wpb), 0, st, logits, lse, rows, C);
  else if ((C & 3) == 0 && C <= 512) hipLaunchKernelGGL((lse_rows_reg_kernel<2>), dim3(blocks), dim3(64 * wpb), 0, st, logits, lse, rows, C);
  else if ((C & 3) == 0 && C <= 1024) hipLaunchKernelGGL((lse_rows_reg_kernel<4>), dim3(blocks), dim3(64 * wpb), 0, st, logits, lse, rows, C);
  else if ((C & 3) == 0 && C <= 2048) hipLaunchKernelGGL((lse_rows_reg_kernel<8>), dim3(blocks), dim3(64 * wpb), 0, st, logits, lse, rows, C);
  else if ((C & 3) == 0) hipLaunchKernelGGL(lse_rows_kernel<true>, dim3(blocks), dim3(64 * wpb), 0, st, logits, lse, rows, C);
  else hipLaunchKernelGGL(lse_rows_kernel<false>, dim3(blocks), dim3(64 * wpb), 0, st, logits, lse, rows, C);
  return check_launch("lse_rows");
}

int pruned_logprobs_fwd(const float* logits, const int32_t* symbols, const int32_t* ranges,
                        const int32_t* boundary, int blank, double delay_penalty, float* lse, float* px,
                        float* py, int B, int T, int S, int C, int r, int modified, hipStream_t st) {
  const size_t rows = (size_t)B * T * r;
  if (rows == 0) return FTR_OK;
  int rc = lse_rows(logits, lse, rows, C, st);
  if (rc != FTR_OK) return rc;
  const int threads = 256;
  const dim3 grid((T + 1 + threads - 1) / threads, S + 1, B);
  if (modified) hipLaunchKernelGGL(band_to_lattice_kernel<true>, grid, dim3(threads), 0, st, logits, symbols, ranges, boundary, lse, blank, delay_penalty, px, py, T, S, C, r);
  else hipLaunchKernelGGL(band_to_lattice_kernel<false>, grid, dim3(threads), 0, st, logits, symbols, ranges, boundary, lse, blank, delay_penalty, px, py, T, S, C, r);
  return check_launch("band_to_lattice");
}

int pruned_logprobs_bwd(const float* logits, const int32_t* symbols, const int32_t* ranges,
                        const int32_t* boundary, int blank, const float* lse, const float* gpx,
                        const float* gpy, Scale scale, float* glogits, int B, int T, int S, int C,
                        int r, int modified, hipStream_t st) {
  const size_t rows = (size_t)B * T * r;
  if (rows == 0) return FTR_OK;
  const int wpb = 4;
  const unsigned blocks = (unsigned)((rows + wpb - 1) / wpb);
  const bool vec = (C & 3) == 0;
#define FTR_LAUNCH_BG(MODV, VECV) hipLaunchKernelGGL((band_grad_kernel<MODV, VECV>), dim3(blocks), dim3(64 * wpb), 0, st, \
    logits, symbols, ranges, boundary, lse, gpx, gpy, scale, blank, glogits, rows, T, S, C, r)
  if (modified) { if (vec) FTR_LAUNCH_BG(true, true); else FTR_LAUNCH_BG(true, false); }
  else { if (vec) FTR_LAUNCH_BG(false, true); else FTR_LAUNCH_BG(false, false); }
#undef FTR_LAUNCH_BG
  return check_launch("band_grad");
}

}  // namespace ftr
