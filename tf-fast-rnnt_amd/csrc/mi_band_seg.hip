// csrc/mi_band_seg.hip -- the band recursion of mi_band.hip cut into SEGMENTS that run in parallel (gfx950).
//
// mi_band.hip runs the pruned loss's recursion as two dependent chains of (S_n + T_n) / 2 steps each, forward and then flow,
// on ONE wave per utterance: 59 us at c3 on 32 of 256 CUs, 620 us at c5 on 8.  But the chain knows nothing about bands -- its
// state after walk step j is one value per lane, LANES = 8 or 16 of them, and a step is the same linear map in the
// (logsumexp, +) semiring for every input:
//        v'[l] = logadd(v[l-1] + OX[j][l], v[l] + OY[j][l])
// -- so the walk can be cut into K segments.  A segment's TRANSFER MATRIX M (LANES x LANES: where does a unit at lane i of
// the segment's first step end up) is LANES chains run side by side on as many 16-lane DPP rows, all K segments of both
// directions at once; a segment's true initial state is the product of the matrices in front of it with the origin (K - 1
// small matrix-vector products, one wave per chain); then every segment runs the one true chain from its initial state.
// Depth 2 (S_n + T_n) / K + K instead of S_n + T_n.
// Both directions run over the WHOLE band -- p(s,t) from the origin in chain A's wavefront order, q(s,t) from the end cell in
// chain B's -- and the occupancy of a transition is exp(p(source) + transition + q(destination) - ans), an elementwise pass:
// no split ratios, no flow chain.  That formula does not forgive rounding the way the flow does (a split ratio is the difference
// of two neighbours that share their history; p + q - ans is not), and a float32 chain picks up ~ulp(|value|) per step:
// 1.4e-4 in the occupancies after 2700 steps, measured.  So the chains ACCUMULATE IN FLOAT64 -- v = max(a, b) + log2(1 + 2^-|a-b|)
// with a, b, v double and only the bounded last term (in [0, 1]) through the float32 exp2 / log2 units -- which also makes
// frames unnecessary here: ~6e-8 per step, absolute, whatever the magnitude of the log-probabilities.
//
// Six launches on a per-utterance workspace (wavefront-ordered arrays, row = walk step, LANES values per row):
//   band_seg_init      neutral operands (-inf, 0) everywhere, header, the utterance's shift constants (ftr_common.h)
//   band_seg_scatter   every band cell's operands into chain A's slot (transitions INTO the cell) and chain B's (OUT of it)
//   band_seg_transfer  grid (K, 2, B): M of every segment but the last
//   band_seg_prefix    grid (2, B): every segment's initial state (K - 1 small matrix-vector products, matrices in LDS)
//   band_seg_final     grid (K, 2, B): the true chain from the segment's initial state, p / q of every step
//   band_seg_occupancy one thread per band cell: px_grad / py_grad band shaped, ans
// The geometry (walk length D, K, steps per segment) depends on the utterance's boundary, i.e. on device data: every kernel
// derives it the same way (seg_geom) and surplus workgroups leave.
#include "ftr_common.h"
#include <cstdlib>
#include <cstring>

namespace ftr {
namespace {

constexpr int kSegMaxK = 32;      // segments per chain at most
constexpr int kSegTarget = 64;    // steps per segment aimed at while K < kSegMaxK
constexpr int kSegU = 4;          // steps per operand fetch group (as kBandAhead in mi_band.hip)
constexpr int kSegFlagBad = 1, kSegFlagNaN = 2, kSegFlagNoOrigin = 4, kSegFlagNoEnd = 8;

__device__ __forceinline__ float seg_ror1(float v) {   // lane i of each 16-lane row receives lane (i-1) mod 16
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
}
template <int N>
__device__ __forceinline__ float seg_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float seg_row_max(float v) {
  v = fmaxf(v, seg_ror<8>(v)); v = fmaxf(v, seg_ror<4>(v));
  v = fmaxf(v, seg_ror<2>(v)); v = fmaxf(v, seg_ror<1>(v));
  return v;
}

__device__ __forceinline__ double seg_ror1d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x121, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x121, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ double seg_rord(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
constexpr double kNegD = -1.0e30, kNegThreshD = -1.0e29;

struct SegGeom { int D, K, L; };
__host__ __device__ inline SegGeom seg_geom(int D) {   // D >= 0 walk steps after the first cell: D + 1 rows
  const int steps = D + 1;
  int K = (steps + kSegTarget - 1) / kSegTarget;
  K = K > kSegMaxK ? kSegMaxK : (K < 1 ? 1 : K);
  const int L = (steps + K - 1) / K;
  K = (steps + L - 1) / L;     // no empty segment
  return SegGeom{D, K, L};
}
__host__ __device__ inline int seg_lmax(int T, int S) {
  const int steps = S + T + 1;
  const int l = (steps + kSegMaxK - 1) / kSegMaxK;
  return l > kSegTarget ? l : kSegTarget;
}
// rows of a wavefront array: one front row, the D + 1 <= S + T + 1 steps, and a tail as long as a segment (the last segment
// and the operand fetches run past the walk's end, over neutral rows)
__host__ __device__ inline size_t seg_rows(int T, int S) { return (size_t)S + T + 2 + seg_lmax(T, S) + 2 * kSegU + 2; }
template <int LANES>
__host__ __device__ inline size_t seg_floats_per_utt(int T, int S) {
  const size_t nr = seg_rows(T, S);
  const size_t f = 8 * nr * LANES + 2 * 2 * (size_t)kSegMaxK * (LANES * LANES + LANES) + 16;   // OA, OB (float2), PA, QB (double), M and the segments' initial states (double), header
  return (f + 3) & ~(size_t)3;
}
template <int LANES>
struct SegWs {
  float2* OA; float2* OB; double* PA; double* QB; double* M; double* ST; float* hdr;
  __device__ SegWs(float* base, int T, int S) {
    const size_t nr = seg_rows(T, S);
    OA = reinterpret_cast<float2*>(base); OB = OA + nr * LANES;
    PA = reinterpret_cast<double*>(base + 4 * nr * LANES); QB = PA + nr * LANES;
    M = QB + nr * LANES; ST = M + 2 * (size_t)kSegMaxK * (LANES * LANES);
    hdr = reinterpret_cast<float*>(ST + 2 * (size_t)kSegMaxK * LANES);
  }
  // header: [0] flags (int), [1] cx2, [2] cy2, [4..5] shift_back (double)
  __device__ int* flags() const { return reinterpret_cast<int*>(hdr); }
  __device__ double* shift_back() const { return reinterpret_cast<double*>(hdr + 4); }
  __device__ double* mat(int dir, int seg) const { return M + ((size_t)dir * kSegMaxK + seg) * (LANES * LANES); }   // [LANES][LANES]
  __device__ double* state(int dir, int seg) const { return ST + ((size_t)dir * kSegMaxK + seg) * LANES; }          // [LANES]
};
constexpr int kRow0 = 1;   // row of walk step 0

struct SegUtt { Bound bd; int Sn, Tn, D; bool degenerate; };
template <bool MOD>
__device__ __forceinline__ SegUtt seg_utt(const int32_t* boundary, int b, int S, int T) {
  SegUtt u;
  u.bd = load_boundary(boundary, b, S, T);
  u.Sn = u.bd.se - u.bd.sb + 1; u.Tn = u.bd.te - u.bd.tb + 1;
  u.degenerate = u.Sn <= 0 || u.Tn <= 0 || u.Tn == 1;
  u.D = (MOD ? 0 : (u.Sn - 1)) + (u.Tn - 1);
  return u;
}

// ---------------------------------------------------------------------------------------- init: neutral slots, header, shift
template <bool MOD, int LANES>
__global__ __launch_bounds__(256) void band_seg_init_kernel(const float* __restrict__ pxb, const float* __restrict__ pyb,
                                                            const int32_t* __restrict__ ranges, const int32_t* __restrict__ boundary,
                                                            float* __restrict__ ws, int T, int S, int r) {
  const int b = blockIdx.y;
  const SegWs<LANES> w(ws + (size_t)b * seg_floats_per_utt<LANES>(T, S), T, S);
  const size_t n = 2 * seg_rows(T, S) * LANES;   // OA and OB are adjacent
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) w.OA[i] = make_float2(kNeg, 0.0f);
  if (blockIdx.x != 0) return;
  // the utterance's shift constants from a fixed sample of the band arrays (as mi_band_stream_kernel: 2048 entries of each)
  __shared__ float red[16];
  const SegUtt u = seg_utt<MOD>(boundary, b, S, T);
  if (threadIdx.x == 0) {
    // a chain may only start from a cell that IS a band cell (a lane keeps its value over slots without a cell: a start value
    // in a lane whose first cell is not the origin would leak into whatever row takes that lane later); flags also collects
    // what the scatter pass finds (band start decreasing, NaN inputs)
    int f = 0;
    if (!u.degenerate) {
      const int32_t* rg = ranges + (size_t)b * T * r;
      const int lo_b = rg[(size_t)min(u.bd.tb, u.bd.te - 1) * r], lo_e = rg[(size_t)(u.bd.te - 1) * r];
      if (!(u.bd.sb >= lo_b && u.bd.sb <= lo_b + r - 1)) f |= kSegFlagNoOrigin;
      if (!((u.bd.se >= lo_e && u.bd.se <= lo_e + r - 1) || (MOD && u.bd.se == lo_e + r))) f |= kSegFlagNoEnd;
    }
    *w.flags() = f;
  }
  if (u.degenerate) return;
  const float* pxu = pxb + (size_t)b * T * r;
  const float* pyu = pyb + (size_t)b * T * r;
  const unsigned nfr = (unsigned)(u.bd.te - u.bd.tb) * (unsigned)r;
  float sx = 0.0f, nx = 0.0f, sy = 0.0f, ny = 0.0f;
  float vx[8], vy[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const unsigned i = (unsigned)u.bd.tb * (unsigned)r + __umulhi((unsigned)(threadIdx.x + 256 * q) * 0x9E3779B1u, nfr);
    vx[q] = pxu[i]; vy[q] = pyu[i];
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (shift_sample_ok(vx[q])) { sx += vx[q]; nx += 1.0f; }
    if (shift_sample_ok(vy[q])) { sy += vy[q]; ny += 1.0f; }
  }
  sx = wave_sum_dpp(sx); nx = wave_sum_dpp(nx); sy = wave_sum_dpp(sy); ny = wave_sum_dpp(ny);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave] = sx; red[4 + wave] = nx; red[8 + wave] = sy; red[12 + wave] = ny; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, c = 0.f, d = 0.f, e = 0.f;
    for (int q = 0; q < 4; ++q) { a += red[q]; c += red[4 + q]; d += red[8 + q]; e += red[12 + q]; }
    const Shift sh = shift_from_sums<MOD>(a, c, d, e, u.Sn, u.Tn);
    w.hdr[1] = sh.cx2; w.hdr[2] = sh.cy2;
    *w.shift_back() = shift_total<MOD>(sh, u.Sn, u.Tn);
  }
}

// ---------------------------------------------------------------------------------------- scatter: one thread per band cell
// Geometry helpers shared by the scatter and the occupancy pass (the conventions are mi_band.hip's).
template <bool MOD, int LANES>
struct SegBand {
  const int32_t* rg; int r, sb, tb, se, te; bool end_above; int lo_te;
  __device__ SegBand(const int32_t* ranges, int b, int T, int r_, const Bound& bd) : rg(ranges + (size_t)b * T * r_), r(r_), sb(bd.sb), tb(bd.tb), se(bd.se), te(bd.te) {
    lo_te = lo(te);
    end_above = MOD && se == lo_te + r;
  }
  // band start per lattice column: the column t_end has no frame of its own, it continues the last frame's band
  __device__ int lo(int t) const { return rg[(size_t)min(t, te - 1) * r]; }
  __device__ bool in_band(int s, int t, int l) const { return (s >= l && s <= l + r - 1) || (end_above && t == te && s == se); }
  __device__ int dgA(int s, int t) const { return MOD ? (t - tb) : (s - sb) + (t - tb); }
  __device__ int dgB(int s, int t) const { return MOD ? (te - t) : (se - s) + (te - t); }
  __device__ int slotA(int s, int t) const { return (kRow0 + dgA(s, t)) * LANES + ((s - sb) & (LANES - 1)); }
  __device__ int slotB(int s, int t) const { return (kRow0 + dgB(s, t)) * LANES + ((se - s) & (LANES - 1)); }
};

// operands of band cell (s, t): (ax, ay) = the transitions INTO it (chain A's slot), (bx, by) = the transitions OUT of it
// (chain B's slot); log2 domain, shifted; kNeg where the transition does not exist, leaves the rectangle or leaves the band.
// l0 / lm / lp: band start of columns t, t - 1, t + 1.
template <bool MOD, int LANES>
__device__ __forceinline__ void seg_cell_operands(const SegBand<MOD, LANES>& g, const float* __restrict__ pxu, const float* __restrict__ pyu,
                                                  float cx2, float cy2, int s, int t, int l0, int lm, int lp,
                                                  float& ax, float& ay, float& bx, float& by) {
  const int r = g.r, sb = g.sb, tb = g.tb, se = g.se, te = g.te;
  auto at = [&](const float* src, float c2, int ss, int tt, int l) { return __builtin_fmaf(src[(size_t)tt * r + (ss - l)], kLog2e, -c2); };
  ax = kNeg; ay = kNeg; bx = kNeg; by = kNeg;
  {
    const int tx = MOD ? t - 1 : t;
    const int lx = MOD ? lm : l0;
    if (s - 1 >= sb && tx >= tb && tx <= te - 1 && g.in_band(s - 1, tx, lx)) ax = at(pxu, cx2, s - 1, tx, lx);
    if (t - 1 >= tb && g.in_band(s, t - 1, lm)) ay = at(pyu, cy2, s, t - 1, lm);
    if (s == sb && t == tb) ay = 0.0f;                               // origin trick
  }
  if (t <= te - 1) {
    const int tnx = MOD ? t + 1 : t;
    const int ln = MOD ? lp : l0;
    if (s + 1 <= se && tnx <= te && g.in_band(s + 1, tnx, ln)) bx = at(pxu, cx2, s, t, l0);
    if (g.in_band(s, t + 1, lp)) by = at(pyu, cy2, s, t, l0);
  }
  if (s == se && t == te) by = 0.0f;                                 // chain B's origin is the end cell
}

template <bool MOD, int LANES>
__global__ __launch_bounds__(256) void band_seg_scatter_kernel(const float* __restrict__ pxb, const float* __restrict__ pyb,
                                                               const int32_t* __restrict__ ranges, const int32_t* __restrict__ boundary,
                                                               float* __restrict__ ws, int T, int S, int r) {
  const int b = blockIdx.y;
  const SegUtt u = seg_utt<MOD>(boundary, b, S, T);
  if (u.degenerate) return;
  const SegWs<LANES> w(ws + (size_t)b * seg_floats_per_utt<LANES>(T, S), T, S);
  const SegBand<MOD, LANES> g(ranges, b, T, r, u.bd);
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int q = i / r, k = i - q * r;
  const int t = u.bd.tb + q;
  if (t > u.bd.te) return;
  const float cx2 = w.hdr[1], cy2 = w.hdr[2];
  const float* pxu = pxb + (size_t)b * T * r;
  const float* pyu = pyb + (size_t)b * T * r;
  const int sb = u.bd.sb, tb = u.bd.tb, se = u.bd.se, te = u.bd.te;
  const int l0 = g.lo(t), lm = t > tb ? g.lo(t - 1) : l0, lp = t < te ? g.lo(t + 1) : l0;
  unsigned nan_acc = 0;
  int bad = 0;
  if (k == 0 && t > tb && t < te && l0 < lm) bad = kSegFlagBad;   // the band start must not decrease (mi_band.hip)
  auto at = [&](const float* src, float c2, int s, int tt, int l) { return __builtin_fmaf(src[(size_t)tt * r + (s - l)], kLog2e, -c2); };
  auto put = [&](float2* O, int slot, float x, float y) {
    nan_acc = max(nan_acc, max(__float_as_uint(x) & 0x7fffffffu, __float_as_uint(y) & 0x7fffffffu));
    O[slot] = make_float2(fmaxf(x, kNeg), fmaxf(y, kNeg));
  };
  const int s = l0 + k;
  if (s >= sb && s <= se) {
    float ax, ay, bx, by;
    seg_cell_operands<MOD, LANES>(g, pxu, pyu, cx2, cy2, s, t, l0, lm, lp, ax, ay, bx, by);
    put(w.OA, g.slotA(s, t), ax, ay);
    put(w.OB, g.slotB(s, t), bx, by);
  }
  // the end cell outside the band of column t_end (modified type, see mi_band.hip): reached by the last frame's top px only
  if (g.end_above && i == 0) {
    float ax = kNeg;
    const int lq = g.lo(te - 1);
    if (se - 1 >= sb && te - 1 >= tb && g.in_band(se - 1, te - 1, lq)) ax = at(pxu, cx2, se - 1, te - 1, lq);
    put(w.OA, g.slotA(se, te), ax, kNeg);
    w.OB[g.slotB(se, te)] = make_float2(kNeg, 0.0f);
  }
  int fl = bad | ((nan_acc > 0x7f800000u) ? kSegFlagNaN : 0);
  if (__any(fl != 0)) {
    int all = 0;
    for (int bit = 1; bit <= 2; bit <<= 1) if (__any((fl & bit) != 0)) all |= bit;
    if ((threadIdx.x & 63) == 0) atomicOr(w.flags(), all);
  }
}

// ---------------------------------------------------------------------------------------- the chain over one segment
// Operands of the segment's rows are staged in LDS (seg[row][lg], L + 2 U rows: the fetches run two groups ahead); every
// 16-lane DPP row of the calling waves runs its own chain over the SAME operands.  With LANES = 8 both halves of a row carry
// the same 8-lane chain, which makes the 16-lane rotate an 8-lane one (mi_band.hip).
// STORE: keep every step's values in LDS (pbuf[row][lg]; written by DPP row 0 of wave 0).
template <int LANES, bool STORE>
__device__ __forceinline__ void seg_chain(const float2* __restrict__ seg, int L, int lg, double& val, double* __restrict__ pbuf, bool writer) {
  constexpr int U = kSegU;
  auto fwd = [&](int row, float2 o) {
    const double up = seg_ror1d(val);
    const double a_ = up + (double)o.x, b_ = val + (double)o.y;
    const float d = (float)(a_ - b_);
    const float ex = __builtin_amdgcn_exp2f(-__builtin_fabsf(d));
    val = fmax(a_, b_) + (double)__builtin_amdgcn_logf(1.0f + ex);
    if (STORE && writer) pbuf[row * LANES + lg] = val;
  };
  float2 oa[U], ob[U];
#pragma unroll
  for (int q = 0; q < U; ++q) oa[q] = seg[q * LANES + lg];
  int i = 0;
  for (; i + 2 * U <= L; i += 2 * U) {
#pragma unroll
    for (int q = 0; q < U; ++q) ob[q] = seg[(i + U + q) * LANES + lg];
#pragma unroll
    for (int q = 0; q < U; ++q) fwd(i + q, oa[q]);
#pragma unroll
    for (int q = 0; q < U; ++q) oa[q] = seg[(i + 2 * U + q) * LANES + lg];
#pragma unroll
    for (int q = 0; q < U; ++q) fwd(i + U + q, ob[q]);
  }
  for (; i < L; ++i) fwd(i, seg[i * LANES + lg]);
}

template <int LANES>
__device__ __forceinline__ void seg_stage(float2* __restrict__ seg, const float2* __restrict__ O, int row0, int L, int tid, int nthreads) {
  const int n = (L + 2 * kSegU) * LANES;
  const float2* src = O + (size_t)row0 * LANES;
  for (int i = tid; i < n; i += nthreads) seg[i] = src[i];
}

// grid (kSegMaxK, 2, B), 16 LANES threads: LANES chains, one per DPP row, from the unit vectors
template <bool MOD, int LANES>
__global__ __launch_bounds__(16 * LANES) void band_seg_transfer_kernel(const int32_t* __restrict__ boundary, float* __restrict__ ws, int T, int S) {
  extern __shared__ __attribute__((aligned(16))) float2 seg[];
  const int sg = blockIdx.x, dir = blockIdx.y, b = blockIdx.z;
  const SegUtt u = seg_utt<MOD>(boundary, b, S, T);
  if (u.degenerate) return;
  const SegGeom gm = seg_geom(u.D);
  if (sg >= gm.K - 1) return;                      // nobody needs the last segment's matrix
  const SegWs<LANES> w(ws + (size_t)b * seg_floats_per_utt<LANES>(T, S), T, S);
  seg_stage<LANES>(seg, dir ? w.OB : w.OA, kRow0 + sg * gm.L, gm.L, threadIdx.x, 16 * LANES);
  __syncthreads();
  const int l16 = threadIdx.x & 15, lg = l16 & (LANES - 1);
  const int basis = threadIdx.x >> 4;              // this DPP row's unit vector
  double val = (lg == basis) ? 0.0 : kNegD;
  seg_chain<LANES, false>(seg, gm.L, lg, val, nullptr, false);
  double* m = w.mat(dir, sg);
  if (l16 < LANES) m[basis * LANES + lg] = val;
}

// (Folding this kernel into the transfer kernel -- the workgroup of a chain that finishes last, found with a counter behind a
// __threadfence(), does the products -- was built and is 70 us SLOWER at c3: a device-scope fence on this part writes back
// the whole L2 of its XCD, once per workgroup.  Kernel boundaries do that once.)
// grid (2, B), one wave: the initial state of every segment -- the start cell (lane 0 = 0) pushed through the matrices one
// after the other: s'[o] = log2 sum_i 2^(s[i] + M_g[i][o]) (double; only the bounded differences go through the float32
// exp2 / log2).  All matrices are brought into LDS first: the K - 1 products are a dependent chain, their loads are not.
template <bool MOD, int LANES>
__global__ __launch_bounds__(64) void band_seg_prefix_kernel(const int32_t* __restrict__ boundary, float* __restrict__ ws, int T, int S) {
  extern __shared__ __attribute__((aligned(16))) double mats[];   // [K - 1][LANES][LANES]
  const int dir = blockIdx.x, b = blockIdx.y;
  const SegUtt u = seg_utt<MOD>(boundary, b, S, T);
  if (u.degenerate) return;
  const SegGeom gm = seg_geom(u.D);
  const SegWs<LANES> w(ws + (size_t)b * seg_floats_per_utt<LANES>(T, S), T, S);
  const int lane = threadIdx.x, l16 = lane & 15, lg = l16 & (LANES - 1);
  const int nm = (gm.K - 1) * LANES * LANES;
  const double* m0 = w.mat(dir, 0);
  for (int i = lane; i < nm; i += 64) mats[i] = m0[i];
  const bool start_ok = (*w.flags() & (dir ? kSegFlagNoEnd : kSegFlagNoOrigin)) == 0;
  double val = (lg == 0 && start_ok) ? 0.0 : kNegD;
  if (lane < LANES) w.state(dir, 0)[lg] = val;
  __syncthreads();
  for (int g = 0; g + 1 < gm.K; ++g) {
    const double* m = mats + (size_t)g * LANES * LANES;
    double tt[LANES];
    double best = kNegD;
#pragma unroll
    for (int i = 0; i < LANES; ++i) {
      // rotating by i brings the state of lane (o - i) to lane o: source index src = (o - i) mod LANES
      double si = val;
      switch (i) {   // compile-time rotate counts
        case 0: break;
#define FTR_ROT(N) case N: si = seg_rord<N>(val); break;
        FTR_ROT(1) FTR_ROT(2) FTR_ROT(3) FTR_ROT(4) FTR_ROT(5) FTR_ROT(6) FTR_ROT(7) FTR_ROT(8)
        FTR_ROT(9) FTR_ROT(10) FTR_ROT(11) FTR_ROT(12) FTR_ROT(13) FTR_ROT(14) FTR_ROT(15)
#undef FTR_ROT
      }
      const int src = (lg - i) & (LANES - 1);
      tt[i] = si + m[src * LANES + lg];
      best = fmax(best, tt[i]);
    }
    if (best > kNegThreshD) {
      float sum = 0.0f;
#pragma unroll
      for (int i = 0; i < LANES; ++i) sum += __builtin_amdgcn_exp2f((float)(tt[i] - best));
      val = best + (double)__builtin_amdgcn_logf(sum);
    } else {
      val = kNegD;
    }
    if (lane < LANES) w.state(dir, g + 1)[lg] = val;
  }
}

// grid (kSegMaxK, 2, B), one wave: the true chain from the segment's initial state
template <bool MOD, int LANES>
__global__ __launch_bounds__(64) void band_seg_final_kernel(const int32_t* __restrict__ boundary, float* __restrict__ ws, int T, int S) {
  extern __shared__ __attribute__((aligned(16))) float2 seg[];
  const int sg = blockIdx.x, dir = blockIdx.y, b = blockIdx.z;
  const SegUtt u = seg_utt<MOD>(boundary, b, S, T);
  if (u.degenerate) return;
  const SegGeom gm = seg_geom(u.D);
  if (sg >= gm.K) return;
  const SegWs<LANES> w(ws + (size_t)b * seg_floats_per_utt<LANES>(T, S), T, S);
  const int lane = threadIdx.x, l16 = lane & 15, lg = l16 & (LANES - 1);
  double* pbuf = reinterpret_cast<double*>(seg + (size_t)(gm.L + 2 * kSegU) * LANES);   // [L][LANES]
  seg_stage<LANES>(seg, dir ? w.OB : w.OA, kRow0 + sg * gm.L, gm.L, lane, 64);
  double val = w.state(dir, sg)[lg];   // the segment's initial state (band_seg_prefix_kernel)
  __syncthreads();
  seg_chain<LANES, true>(seg, gm.L, lg, val, pbuf, lane < LANES);
  __syncthreads();
  // ---- values of the segment's rows, coalesced
  double* P = (dir ? w.QB : w.PA) + (size_t)(kRow0 + sg * gm.L) * LANES;
  for (int i = lane; i < gm.L * LANES; i += 64) P[i] = pbuf[i];
}

// ---------------------------------------------------------------------------------------- occupancies, ans
template <bool MOD, int LANES>
__global__ __launch_bounds__(256) void band_seg_occupancy_kernel(const int32_t* __restrict__ ranges, const int32_t* __restrict__ boundary,
                                                                 const float* __restrict__ ws, float* __restrict__ ans,
                                                                 float* __restrict__ gxb, float* __restrict__ gyb, int T, int S, int r) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const size_t cells = (size_t)T * r;
  const SegUtt u = seg_utt<MOD>(boundary, b, S, T);
  float* gx_g = gxb + (size_t)b * cells;
  float* gy_g = gyb + (size_t)b * cells;
  if (u.degenerate) {
    if ((size_t)i < cells) { gx_g[i] = 0.0f; gy_g[i] = 0.0f; }
    if (i == 0) ans[b] = (u.Sn <= 0 || u.Tn <= 0) ? 0.0f : ((u.Sn == 1) ? 0.0f : -INFINITY);
    return;
  }
  const SegWs<LANES> w(const_cast<float*>(ws) + (size_t)b * seg_floats_per_utt<LANES>(T, S), T, S);
  const SegBand<MOD, LANES> g(ranges, b, T, r, u.bd);
  const int flags = *w.flags();
  const int fl = flags & (kSegFlagBad | kSegFlagNaN);
  const int sb = u.bd.sb, tb = u.bd.tb, se = u.bd.se, te = u.bd.te;
  // ans = p(end cell): chain A's last step (if the end cell is a band cell at all: its slot holds somebody else's value otherwise)
  const double pend = w.PA[g.slotA(se, te)];
  const double ans2 = pend;
  const bool dead = (flags & (kSegFlagNoOrigin | kSegFlagNoEnd)) != 0 || !(pend > kNegThreshD);
  if (i == 0) ans[b] = fl ? __builtin_nanf("") : (dead ? -INFINITY : (float)((ans2 + *w.shift_back()) * 0.6931471805599453));
  if ((size_t)i >= cells) return;
  const int t = i / r, k = i - t * r;
  float fx = 0.0f, fy = 0.0f;
  if (!fl && !dead && t >= tb && t < te) {
    const int l0 = g.lo(t);
    const int s = l0 + k;
    if (s >= sb && s <= se) {
      const double pa = w.PA[g.slotA(s, t)];
      if (pa > kNegThreshD) {
        const float2 ob = w.OB[g.slotB(s, t)];
        if (ob.x > kNegThresh) {           // px(s,t) exists: its destination (s+1, t [t+1 if modified]) is in the band
          const double qv = w.QB[g.slotB(s + 1, MOD ? t + 1 : t)];
          if (qv > kNegThreshD) fx = exp2f((float)((pa + (double)ob.x + qv) - ans2));
        }
        if (ob.y > kNegThresh) {
          const double qv = w.QB[g.slotB(s, t + 1)];
          if (qv > kNegThreshD) fy = exp2f((float)((pa + (double)ob.y + qv) - ans2));
        }
      }
    }
  }
  gx_g[i] = fx;
  gy_g[i] = fy;
}

inline int seg_lanes(int r) { return r <= 7 ? 8 : (r <= 15 ? 16 : 0); }
constexpr int kSegFrom = 1100;   // S + T from which this route is taken (below: the chain kernels of mi_band.hip)
inline size_t seg_lds_bytes(int T, int S, int lanes, bool store) {
  const size_t L = (size_t)seg_lmax(T, S);
  return sizeof(float2) * (L + 2 * kSegU) * lanes + (store ? sizeof(double) * (L * lanes) : 0) + 64;
}

}  // namespace

// 0 unless the segmented route covers the shape (r <= 15, the segment's operands fit LDS) AND pays: its six launches cost
// ~45 us whatever the size, the chain kernels 43 ns per walk step -- measured (scripts/band_bench.py, one MI355X, B = 32 / 8):
//   S + T =  480: 39.3 us against 28.4     850: 44.8 against 43.5     1200 (c3): 51.0 against 57.1
//           1800: 60.4 against 100.0      2300 (c4): 65.8 against 123.4      9000 (c5): 145.7 against 526.6
// so it is taken from S + T >= 1100 (in the c3 step: 60.8 -> 53.8 us).  FTR_BAND_IMPL = chain | segments forces one (tests,
// A/B measurements).  (A single-launch form for bands whose operands fit LDS -- scatter, matrices, states and chains of one
// direction in one 16-wave workgroup -- was built and measured: 29.9 / 41.2 / 52.4 / 70.4 us at the first four sizes, never
// better than the better of the other two, because a float64 chain step costs 125 ns where the float32 one costs 43; dropped.)
int mi_band_seg_supported(int T, int S, int r) {
  bool forced = false;
  if (const char* e = getenv("FTR_BAND_IMPL")) { if (!strcmp(e, "chain")) return 0; forced = !strcmp(e, "segments"); }
  const int lanes = seg_lanes(r);
  if (!lanes || T < 1 || S < 0 || r < 1) return 0;
  if (!forced && (long long)S + T < kSegFrom) return 0;
  if (seg_lds_bytes(T, S, lanes, true) > (size_t)150 * 1024) return 0;
  if ((uint64_t)(T + 1) * r >= (1ull << 31) || (uint64_t)seg_rows(T, S) * lanes >= (1ull << 31)) return 0;
  return 1;
}
size_t mi_band_seg_workspace_floats(int B, int T, int S, int r) {
  if (!mi_band_seg_supported(T, S, r)) return 0;
  return (size_t)B * (seg_lanes(r) == 8 ? seg_floats_per_utt<8>(T, S) : seg_floats_per_utt<16>(T, S));
}

int mi_band_seg(const float* pxb, const float* pyb, const int32_t* ranges, const int32_t* boundary, float* ws, size_t ws_floats,
                float* ans, float* gxb, float* gyb, int B, int T, int S, int r, int modified, hipStream_t st) {
  if (B == 0) return FTR_OK;
  const size_t need = mi_band_seg_workspace_floats(B, T, S, r);
  if (!need || !ws || ws_floats < need || (reinterpret_cast<uintptr_t>(ws) & 15) != 0) {
    set_error("mutual_information_band (segments): workspace of %zu floats (16-byte aligned) required, got %zu", need, ws_floats);
    return FTR_ERR_INVALID_ARG;
  }
  const int lanes = seg_lanes(r);
  const size_t lds_t = seg_lds_bytes(T, S, lanes, false), lds_f = seg_lds_bytes(T, S, lanes, true);
  static bool big_ok = false;
  if (!big_ok) {
    const void* ks[8] = {reinterpret_cast<const void*>(band_seg_transfer_kernel<true, 8>), reinterpret_cast<const void*>(band_seg_transfer_kernel<false, 8>),
                         reinterpret_cast<const void*>(band_seg_transfer_kernel<true, 16>), reinterpret_cast<const void*>(band_seg_transfer_kernel<false, 16>),
                         reinterpret_cast<const void*>(band_seg_final_kernel<true, 8>), reinterpret_cast<const void*>(band_seg_final_kernel<false, 8>),
                         reinterpret_cast<const void*>(band_seg_final_kernel<true, 16>), reinterpret_cast<const void*>(band_seg_final_kernel<false, 16>)};
    for (const void* k : ks)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) {
        (void)hipGetLastError(); set_error("mutual_information_band (segments): cannot raise the dynamic LDS limit"); return FTR_ERR_LAUNCH;
      }
    big_ok = true;
  }
  const unsigned cells_blocks = (unsigned)(((size_t)(T + 1) * r + 255) / 256);
  const unsigned fill_blocks = (unsigned)((2 * seg_rows(T, S) * lanes + 255) / 256);
  const dim3 gseg(kSegMaxK, 2, B);
#define FTR_SEG_RUN(MODV, LV)                                                                                                         \
  do {                                                                                                                                \
    hipLaunchKernelGGL((band_seg_init_kernel<MODV, LV>), dim3(fill_blocks, B), dim3(256), 0, st, pxb, pyb, ranges, boundary, ws, T, S, r); \
    hipLaunchKernelGGL((band_seg_scatter_kernel<MODV, LV>), dim3(cells_blocks, B), dim3(256), 0, st, pxb, pyb, ranges, boundary, ws, T, S, r); \
    hipLaunchKernelGGL((band_seg_transfer_kernel<MODV, LV>), gseg, dim3(16 * LV), lds_t, st, boundary, ws, T, S);                      \
    hipLaunchKernelGGL((band_seg_prefix_kernel<MODV, LV>), dim3(2, B), dim3(64), sizeof(double) * (kSegMaxK - 1) * LV * LV, st, boundary, ws, T, S); \
    hipLaunchKernelGGL((band_seg_final_kernel<MODV, LV>), gseg, dim3(64), lds_f, st, boundary, ws, T, S);                              \
    hipLaunchKernelGGL((band_seg_occupancy_kernel<MODV, LV>), dim3((unsigned)(((size_t)T * r + 255) / 256), B), dim3(256), 0, st,    \
                       ranges, boundary, ws, ans, gxb, gyb, T, S, r);                                                                 \
  } while (0)
  if (lanes == 8) { if (modified) FTR_SEG_RUN(true, 8); else FTR_SEG_RUN(false, 8); }
  else { if (modified) FTR_SEG_RUN(true, 16); else FTR_SEG_RUN(false, 16); }
#undef FTR_SEG_RUN
  return check_launch("mi_band_seg");
}

}  // namespace ftr
