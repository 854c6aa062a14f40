// csrc/mi_band.hip -- the recursion on the PRUNED band itself (gfx950): rnnt_loss_pruned without the full-size lattices.
//
// get_rnnt_logprobs_pruned (/root/reference/tf_fast_rnnt/python/tf_fast_rnnt/rnnt_loss.py:968-1013) pads the band
// [B,T,r] to full-size px / py lattices that are > 97 % -inf and the op runs the whole (S+1) x (T+1) recursion on them.
// Here the band arrays go into ONE launch that runs forward, cut reduction and backward for an utterance inside one
// wave, entirely out of LDS:
//
//   band cell (t,k)  <->  lattice cell (s,t), s = s0[t] + k,  s0[t] = ranges[b,t,0]   (monotone, steps <= r-1: what
//   get_rnnt_prune_ranges produces; the host only routes such ranges here)
//
// Wavefront over the band: the cells of one anti-diagonal (regular) / one column (modified) that lie inside the band
// are at most r consecutive lattice rows (the band is monotone), so with LANES >= r lanes, lane (row mod LANES) owns at
// most one cell per walk step, and both predecessors of a cell were computed one step earlier -- by the same lane
// ((s,t-1)) and by the cyclically previous lane ((s-1,t) resp. (s-1,t-1): v_mov_b32_dpp row_ror:1).
//
// The chains themselves know nothing about bands: the parallel phases in front of them scatter the operands of every
// band cell into WAVEFRONT ORDER, [walk step][lane], with every transition that does not exist, leaves the rectangle or
// leaves the band already replaced by -inf, and slots without a cell filled with -inf.  Every lane then just computes
//     v <- logadd(dpp(v) + OX[step][lane], v + OY[step][lane])
// step after step, reading LDS at consecutive addresses two steps ahead of the dependent chain.
//
// Two chains per utterance, as in mi_wave_bidir.hip: chain A (DPP row 0) walks up from the origin, chain B (DPP row 1)
// walks down from the end cell; each stores the split ratio G of its cells.  They meet on the cut (the middle
// anti-diagonal / column), ans = logsumexp over the cut of p + q, and then each chain simply KEEPS WALKING through the
// other half in "flow" mode: it injects the cut occupancies and pushes them on with the other chain's ratios, writing
// the per-transition flows (= the occupancies px_grad / py_grad), which a last parallel phase stores band shaped.
//
// LDS (per utterance): lo[T+1] (band start per column) and three arrays of (S_n + T_n + 1) x LANES floats: OX, OY
// (operands, later the two flow outputs) and G.  LANES = 8 while r <= 7, 16 up to r = 15 (r + 1 lanes: see in_band in the kernel).
#include "ftr_common.h"
#include <type_traits>
#include <cstdlib>

namespace ftr {
namespace {

constexpr int kBandThreads = 512;   // eight waves for the parallel phases (staging, operands, store); the chains run in wave 0

__device__ __forceinline__ float dpp_row_ror1(float v) {   // lane i of each 16-lane row receives lane (i-1) mod 16
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
}
template <int N>
__device__ __forceinline__ float dpp_row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float v) {      // all-reduce over the 16 lanes of a DPP row
  v = fmaxf(v, dpp_row_ror<8>(v)); v = fmaxf(v, dpp_row_ror<4>(v));
  v = fmaxf(v, dpp_row_ror<2>(v)); v = fmaxf(v, dpp_row_ror<1>(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_row_ror<8>(v); v += dpp_row_ror<4>(v); v += dpp_row_ror<2>(v); v += dpp_row_ror<1>(v);
  return v;
}

// ---------------------------------------------------------------------------------------- band gather
// px_band[b,t,k] = px[b, s0+k, t], py_band[b,t,k] = py[b, s0+k, t] of get_rnnt_logprobs_pruned (rnnt_loss.py:942-1016)
// + the delay-penalty block (:1097-1114): logits[row, symbol] - lse, -inf where the lattice has none (s >= S, the
// boundary column of the regular type), one thread per band cell.
template <bool MOD>
__global__ void band_gather_kernel(const float* __restrict__ logits, const int32_t* __restrict__ symbols,
                                   const int32_t* __restrict__ ranges, const int32_t* __restrict__ boundary,
                                   const float* __restrict__ lse, int blank, double delay_penalty,
                                   float* __restrict__ pxb, float* __restrict__ pyb, size_t rows, int T, int S, int C, int r) {
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const size_t bt = row / r;
  const int k = (int)(row - bt * r);
  const int b = (int)(bt / T);
  const int t = (int)(bt - (size_t)b * T);
  const int s = ranges[bt * r] + k;
  const int te = boundary ? boundary[4 * b + 3] : T;
  const float l = lse[row];
  float vy = -INFINITY, vx = -INFINITY;
  if (s >= 0 && s <= S) {
    vy = logits[row * C + blank] - l;
    if (s < S) {
      vx = logits[row * C + min(max(symbols[(size_t)b * S + s], 0), C - 1)] - l;
      if (!MOD && t == te) vx = -INFINITY;
      if (delay_penalty > 0.0) vx += (float)((((double)te - 1.0) / 2.0 - (double)t) * delay_penalty);
    }
  }
  pxb[row] = vx;
  pyb[row] = vy;
}

// ---------------------------------------------------------------------------------------- are these ranges a band?
// What the band kernels assume of `ranges` inside each utterance's boundary rectangle (frames [t_begin, t_end)):
// bit 0 of flags[0] is set when some ranges[b,t,0] < ranges[b,t-1,0] (not monotone), bit 1 when some
// ranges[b,t,k] != ranges[b,t,0] + k (not contiguous).  One thread per frame; flags[0] must be zero beforehand.
__global__ void band_ranges_check_kernel(const int32_t* __restrict__ ranges, const int32_t* __restrict__ boundary,
                                         int* __restrict__ flags, int B, int T, int r) {
  const size_t bt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int bad = 0;
  if (bt < (size_t)B * T) {
    const int b = (int)(bt / T), t = (int)(bt - (size_t)b * T);
    const int tb = boundary ? max(boundary[4 * b + 1], 0) : 0, te = boundary ? min(boundary[4 * b + 3], T) : T;
    if (t >= tb && t < te) {
      const int32_t* row = ranges + bt * r;
      const int s0 = row[0];
      if (t > tb && s0 < row[-r]) bad |= 1;
      for (int k = 1; k < r; ++k) if (row[k] != s0 + k) bad |= 2;
    }
  }
  if (__any(bad != 0)) {
    int all = 0;
    for (int bit = 1; bit <= 2; bit <<= 1) if (__any((bad & bit) != 0)) all |= bit;
    if ((threadIdx.x & 63) == 0) atomicOr(flags, all);
  }
}

// ---------------------------------------------------------------------------------------- the band recursion
// Wavefront-ordered arrays, [row][lane], LANES lanes per row:
//   rows 0 .. 2U-1         only ever fetched ahead, never used
//   row 2U                 pad (G = 0): where the shorter flow walk takes its one surplus step
//   + 1 .. jm+1            chain A's steps: cell (s,t) at step (s-sb)+(t-tb) [modified: t-tb], lane (s-sb) mod LANES
//   + jm+2                 pad (OX = -inf, OY = 0: a step that leaves the value unchanged), chain A's surplus forward step
//   + jm+3 .. D+3          chain B's steps: step (se-s)+(te-t) [te-t], lane (se-s) mod LANES; the cut cells appear in both
//   2U more rows           only ever fetched ahead
// so a cell at chain A's (step j, lane l) is chain B's (step D-j, lane (S_n-1-l) mod LANES).  With the pads both chains
// run the same number of steps and no step needs a predicate.
// LDS: lo[T+1] | cutSA cutSB [32 ints] | O2[ncap] (float2: OX, OY) | G[ncap] | cutA cutB occ [48]
#ifdef FTR_BAND_STAMPS   // diagnostic build: phase times of utterance 0 (s_memrealtime, 100 MHz), read by ftr_debug_band_stamps()
static __device__ unsigned long long g_bstamp[16];
#define FTR_BSTAMP(k) do { if (b == 0 && tid == 0) g_bstamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FTR_BSTAMP(k) do {} while (0)
#endif
constexpr int kBandAhead = 4;   // steps per operand fetch group
constexpr int kRenormEvery = 4;  // the chains renormalise every kRenormEvery iterations of 2 kBandAhead steps (a power of two)
template <int LANES>
__host__ __device__ inline size_t band_lds_bytes(int T, int S) {
  const size_t ncap = ((size_t)S + T + 5 + 4 * kBandAhead) * LANES;
  return sizeof(int) * ((size_t)(T + 2) + 32) + sizeof(float) * (3 * ncap + 48) + 64;
}

template <bool MOD, int LANES>
__global__ __launch_bounds__(kBandThreads) void mi_band_kernel(
    const float* __restrict__ pxb, const float* __restrict__ pyb, const int32_t* __restrict__ ranges,
    const int32_t* __restrict__ boundary, float* __restrict__ ans, float* __restrict__ gxb, float* __restrict__ gyb,
    int B, int T, int S, int r, unsigned rinv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Bound bd = load_boundary(boundary, b, S, T);
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  const size_t cells_g = (size_t)T * r;
  float* gx_g = gxb + (size_t)b * cells_g;
  float* gy_g = gyb + (size_t)b * cells_g;
  if (Sn <= 0 || Tn <= 0 || Tn == 1) {
    // empty rectangle: ans = 0 (mi_wave_bidir.hip); a single column has no frame inside the rectangle: only the
    // one-cell lattice has a path (no transitions at all)
    for (size_t i = tid; i < cells_g; i += kBandThreads) { gx_g[i] = 0.0f; gy_g[i] = 0.0f; }
    if (tid == 0) ans[b] = (Sn <= 0 || Tn <= 0) ? 0.0f : ((Sn == 1) ? 0.0f : -INFINITY);
    return;
  }
  FTR_BSTAMP(0);
  const int te = bd.te, tb = bd.tb, sb = bd.sb, se = bd.se;
  const int D = (MOD ? 0 : (Sn - 1)) + (Tn - 1);
  const int jm = D >> 1;                                            // the cut, in chain A's steps; chain B meets it at D - jm
  const int rowA = 1 + 2 * kBandAhead, rowB = rowA + jm + 2;
  // ---- LDS carve-up
  int* lo = reinterpret_cast<int*>(smem);
  int* cutSA = lo + ((T + 2) & ~1); int* cutSB = cutSA + 16;        // lattice row of each lane's cut cell (-1: none)
  const int ncap = (S + T + 5 + 4 * kBandAhead) * LANES;            // fetches run up to two groups past either end
  float2* O2 = reinterpret_cast<float2*>(cutSB + 16);               // 8-byte aligned: an even number of ints in front
  float* G = reinterpret_cast<float*>(O2 + ncap);
  float* cutA = G + ncap; float* cutB = cutA + 16; float* occ = cutB + 16;   // 16 each, indexed by s & 15
  auto div_r = [&](int i) { return rinv ? (int)__umulhi((unsigned)i, rinv) : i; };   // i / r for 0 <= i < 2^32 / r (rinv = 0: r = 1)

  // ---- band start per lattice column: the column t_end has no frame of its own, it continues the last frame's band
  const int32_t* rg = ranges + (size_t)b * T * r;
  for (int t0 = 0; t0 <= T; t0 += 4 * kBandThreads) {               // four independent loads in flight per lane
    int v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int t = t0 + u * kBandThreads + tid; v[u] = (t <= T) ? rg[(size_t)min(t, te - 1) * r] : 0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int t = t0 + u * kBandThreads + tid; if (t <= T) lo[t] = v[u]; }
  }
  for (int i = tid; i < ncap; i += kBandThreads) O2[i] = make_float2(kNeg, kNeg);   // slots without a cell
  if (tid < 32) cutSA[tid] = -1;
  __syncthreads();
  FTR_BSTAMP(1);
  {
    // the one precondition the wavefront order rests on: the band start never decreases along t (two cells of one walk
    // step would otherwise share a slot).  A violation is answered loudly: ans = NaN, zero occupancies.
    int bad = 0;
    for (int t = tb + 1 + tid; t < te; t += kBandThreads) bad |= (lo[t] < lo[t - 1]);
    if (__syncthreads_or(bad)) {
      for (size_t i = tid; i < cells_g; i += kBandThreads) { gx_g[i] = 0.0f; gy_g[i] = 0.0f; }
      if (tid == 0) ans[b] = __builtin_nanf("");
      return;
    }
  }
  if (tid < LANES) O2[(rowA + jm + 1) * LANES + tid] = make_float2(kNeg, 0.0f);   // chain A's pad step
  // Column t_end has no frame: its cells are where the last frame's transitions lead.  For the regular type that is the
  // last frame's band again (py moves along a row; px inside column t_end is masked by fix_for_boundary); for the modified
  // type px moves up one row AND one column, so the END CELL may sit one row above that band (s_end = lo + r) and still be
  // reached -- by the last frame's top px.  (Any other cell of column t_end is a dead end and needs no slot.)
  const bool end_above = MOD && se == lo[te] + r;
  auto in_band = [&](int s, int t) { const int l = lo[t]; return (s >= l && s <= l + r - 1) || (end_above && t == te && s == se); };
  // wavefront slot of a lattice cell in chain A's part / chain B's part of the arrays
  auto slotA = [&](int s, int t) { return (rowA + (MOD ? (t - tb) : (s - sb) + (t - tb))) * LANES + ((s - sb) & (LANES - 1)); };
  auto slotB = [&](int s, int t) { return (rowB + (MOD ? (te - t) : (se - s) + (te - t))) * LANES + ((se - s) & (LANES - 1)); };

  // ---- operands, one input array at a time: the band is staged band shaped in the G region (16-byte loads, eight in
  // flight per lane) and scattered from there into wavefront order with all the masking applied:
  //   chain A's slot of a cell: OX = px(s-1, t [t-1 if modified]), OY = py(s, t-1)      (the transitions INTO the cell)
  //   chain B's slot:           OX = px(s,t), OY = py(s,t)                              (the transitions OUT of it)
  const int nfr = (te - tb) * r;                                     // frames [tb, te)
  // While an array is staged its finite entries are summed: the mean gives this utterance's shift constant (ftr_common.h,
  // Shift), which operands() subtracts from every transition it scatters.  Per-wave partial sums go through red[]
  // (the cut exchange area, unused until the chains run) in a fixed order, so all threads derive the same constant.
  float* red = cutA;
  auto stage = [&](const float* src) {
    const float* g0 = src + ((size_t)b * T + tb) * r;
    float ssum = 0.0f, scnt = 0.0f;
    for (int i0 = 0; i0 < nfr; i0 += 8 * 4 * kBandThreads) {
      f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 4 * (u * kBandThreads + tid);
        if (i + 3 < nfr) v[u] = *reinterpret_cast<const f4u*>(g0 + i);
        else { v[u] = f4{0.f, 0.f, 0.f, 0.f}; for (int e = 0; e < 4; ++e) if (i + e < nfr) v[u][e] = g0[i + e]; }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 4 * (u * kBandThreads + tid);
        for (int e = 0; e < 4; ++e) if (i + e < nfr) {
          G[i + e] = v[u][e] * kLog2e;
          if (shift_sample_ok(v[u][e])) { ssum += v[u][e]; scnt += 1.0f; }
        }
      }
    }
    ssum = wave_sum_dpp(ssum); scnt = wave_sum_dpp(scnt);
    if (lane == 0) { red[wave] = ssum; red[8 + wave] = scnt; }
  };
  auto staged_sums = [&](float& sum, float& cnt) {   // after the barrier behind stage()
    sum = 0.0f; cnt = 0.0f;
#pragma unroll
    for (int u = 0; u < kBandThreads / 64; ++u) { sum += red[u]; cnt += red[8 + u]; }
  };
  auto staged = [&](int s, int t) { return G[(t - tb) * r + (s - lo[t])]; };   // band value at lattice cell (s,t), frame t < te
  unsigned nan_acc = 0;
  auto operands = [&](auto xtag, const float c2) {
    constexpr bool ISX = decltype(xtag)::value;
    float* O = reinterpret_cast<float*>(O2) + (ISX ? 0 : 1);
    for (int i = tid; i < Tn * r; i += kBandThreads) {
      const int q = div_r(i);
      const int t = tb + q, k = i - q * r;
      const int s = lo[t] + k;
      if (s < sb || s > se) continue;
      const int dg = MOD ? (t - tb) : (s - sb) + (t - tb);
      if (dg <= jm) {
        float av = kNeg;
        if (ISX) { const int tx = MOD ? t - 1 : t; if (s - 1 >= sb && tx >= tb && tx <= te - 1 && in_band(s - 1, tx)) av = staged(s - 1, tx) - c2; }
        else { if (t - 1 >= tb && in_band(s, t - 1)) av = staged(s, t - 1) - c2; if (s == sb && t == tb) av = 0.0f; }   // origin trick
        nan_acc = max(nan_acc, __float_as_uint(av) & 0x7fffffffu);
        O[2 * slotA(s, t)] = fmaxf(av, kNeg);
        if (ISX && dg == jm) cutSA[(s - sb) & (LANES - 1)] = s;
      }
      if (dg >= jm) {
        float bv = kNeg;
        if (t <= te - 1) {
          if (ISX) { const int tnx = MOD ? t + 1 : t; if (s + 1 <= se && tnx <= te && in_band(s + 1, tnx)) bv = staged(s, t) - c2; }
          else if (in_band(s, t + 1)) bv = staged(s, t) - c2;
        }
        if (!ISX && s == se && t == te) bv = 0.0f;                   // chain B's origin is the end cell
        nan_acc = max(nan_acc, __float_as_uint(bv) & 0x7fffffffu);
        O[2 * slotB(s, t)] = fmaxf(bv, kNeg);
        if (ISX && dg == jm) cutSB[(se - s) & (LANES - 1)] = s;
      }
    }
    // the end cell outside the band of column t_end (see in_band): chain B's origin, no outgoing transitions; its walk
    // step D is behind the cut unless the rectangle is a single column (handled above)
    if (end_above && tid == 0) O[2 * slotB(se, te)] = ISX ? kNeg : 0.0f;
  };
  float sumx, cntx, sumy, cnty;
  stage(pxb);
  __syncthreads();
  FTR_BSTAMP(2);
  staged_sums(sumx, cntx);
  // cx2 depends on the px mean only and cy2 on the py mean only (shift_from_sums)
  const float cx2 = shift_from_sums<MOD>(sumx, cntx, 0.0f, 0.0f, Sn, Tn).cx2;
  operands(std::true_type{}, cx2);
  __syncthreads();
  FTR_BSTAMP(3);
  stage(pyb);
  __syncthreads();
  FTR_BSTAMP(4);
  staged_sums(sumy, cnty);
  const float cy2 = shift_from_sums<MOD>(0.0f, 0.0f, sumy, cnty, Sn, Tn).cy2;
  operands(std::false_type{}, cy2);
  const double shift_back = shift_total<MOD>(Shift{cx2, cy2}, Sn, Tn);   // what the shifts took out of ans (log2 units)
  const bool poisoned = __syncthreads_or(nan_acc > 0x7f800000u) != 0;
  FTR_BSTAMP(5);

  // ---- the two chains, in wave 0: chain A in DPP rows 0 and 2, chain B in DPP rows 1 and 3, and with LANES = 8 both
  // halves of a DPP row carry the same 8-lane chain, which makes the 16-lane rotate an 8-lane one.  All the copies compute
  // and store the same values to the same addresses, so no step carries a predicate.  The other waves wait at the barrier
  // behind this block; inside it there is a single wave, whose LDS operations execute in program order ("write, then
  // read by other lanes" needs a compiler fence only).
  if (wave == 0) {
    constexpr int U = kBandAhead;
    const int l16 = lane & 15;
    const int lg = l16 & (LANES - 1);
    const bool isB = ((lane >> 4) & 1) == 1;
    if (lane < LANES) G[(rowA - 1) * LANES + lane] = 0.0f;    // the front pad row (the staging area covered it)
    float val = (lg == 0) ? 0.0f : kNeg;                      // origin trick: the first cell starts from 0 with OY = 0
    // ---- phase 1: forward, both chains D - jm + 1 steps (chain A's last one may be the pad step).  Operands are fetched
    // a group of U steps ahead of their use, at immediate offsets from one running address.
    const int n1 = D - jm + 1;
    const int base1 = (isB ? rowB : rowA) * LANES + lg;
    auto fwd = [&](int slot, float2 o) {
      const float up = dpp_row_ror1(val);
      const float a_ = up + o.x, b_ = val + o.y;
      const float d = a_ - b_;
      const float ex = __builtin_amdgcn_exp2f(-__builtin_fabsf(d));
      val = fmaxf(a_, b_) + __builtin_amdgcn_logf(1.0f + ex);
      const float rc = __builtin_amdgcn_rcpf(1.0f + ex);
      G[slot] = (d >= 0.0f) ? rc : ex * rc;
    };
    // Renormalisation ("frames", as in mi_wave_bidir.hip): every 32 steps the chain's values are brought back to the
    // neighbourhood of zero by an integer shift, the same for all lanes of the chain (a uniform shift of an anti-diagonal
    // changes no split ratio), so that float32 keeps its resolution whatever the magnitude of the log-probabilities -- the
    // static shift above only removes what the band's MEAN transition predicts.  The maximum is taken from a copy made 2 U
    // steps earlier; on the chain it costs one subtraction, but the four dependent DPP maxima in front of it are not free on
    // a lone in-order wave (every 8 steps they cost the LDS kernel 8 % and the streaming kernel 24 %), hence every 32.
    // frame = what has been subtracted so far, added back on the cut.
    float frame = 0.0f;
    auto renorm = [&](float snap) {   // snap: this lane's value 2 U steps ago (no shift in between: the same frame)
      const float mx = row16_max(snap);
      const float st = (mx > kNegThresh) ? __builtin_rintf(mx) : 0.0f;
      val -= st; frame += st;
    };
    {
      // two register sets, each filled half an iteration (U steps) before it is used
      float2 oa[U], ob[U];
#pragma unroll
      for (int u = 0; u < U; ++u) oa[u] = O2[base1 + u * LANES];
      int i = 0;
      for (; i + 2 * U <= n1; i += 2 * U) {
        const int sl = base1 + i * LANES;
        const bool rn = ((i / (2 * U)) & (kRenormEvery - 1)) == kRenormEvery - 1;   // every 2 U kRenormEvery steps
        const float snap = val;
#pragma unroll
        for (int u = 0; u < U; ++u) ob[u] = O2[sl + (U + u) * LANES];
#pragma unroll
        for (int u = 0; u < U; ++u) fwd(sl + u * LANES, oa[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) oa[u] = O2[sl + (2 * U + u) * LANES];
#pragma unroll
        for (int u = 0; u < U; ++u) fwd(sl + (U + u) * LANES, ob[u]);
        if (rn) renorm(snap);
      }
      for (; i < n1; ++i) fwd(base1 + i * LANES, O2[base1 + i * LANES]);
    }
    FTR_BSTAMP(6);
    // ---- the cut: p + q per cut cell (keyed by s mod 16: at most one cut cell per residue), ans, occupancies
    if (lane < 48) cutA[lane] = kNeg;                         // cutA, cutB, occ
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int scut = (isB ? cutSB : cutSA)[lg];               // lattice row of this lane's cut cell
    if (scut >= 0) (isB ? cutB : cutA)[scut & 15] = val;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      // both chains' frames: chain A's lanes hold theirs in DPP rows 0 / 2, chain B's in rows 1 / 3
      const float frA = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, frame), 0));
      const float frB = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, frame), 16));
      const float v = cutA[l16] + cutB[l16];
      const float m = row16_max(v);
      const float e = exp2f(v - m);
      const float sum = row16_sum(e);
      const float total = m + log2f(sum);
      const bool dead = !(total > kNegThresh);
      if (lane < 16) occ[lane] = (dead || poisoned) ? 0.0f : e / sum;
      if (lane == 0) ans[b] = poisoned ? __builtin_nanf("") : (dead ? -INFINITY : (float)(((double)total + (double)frA + (double)frB + shift_back) * 0.6931471805599453));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float inj = (scut >= 0) ? occ[scut & 15] : 0.0f;   // the occupancy this lane injects at its cut cell
    FTR_BSTAMP(7);
    // ---- phase 2: flow.  Each chain walks on from the cut through the other chain's half: at its step j it is on the
    // cells the other chain had at step D - j (chain B's surplus step lands on the front pad row).  No masks: a transition
    // that does not exist carries exactly zero flow (its source's ratio is exactly 0 or 1, or its source got no flow), and
    // slots without a cell receive none.  Chain A leaves in each cell's CHAIN-B slot what entered the cell through its two
    // incoming transitions; chain B leaves in each cell's CHAIN-A slot what it passed towards the origin through the
    // cell's two outgoing transitions (the cut cells' own slots get values nobody reads).
    float xo = 0.0f, yo = 0.0f;
    const int n2 = D - jm + 1;
    const int top = (isB ? rowA + jm : rowB + D - jm) * LANES + ((Sn - 1 - lg) & (LANES - 1));
    auto flow = [&](int slot, float g, auto first) {
      const float xin = dpp_row_ror1(xo), yin = yo;
      float pg = xin + yin;
      if (decltype(first)::value) pg += inj;
      O2[slot] = make_float2(xin, yin);
      xo = pg * g; yo = pg - xo;
    };
    flow(top, G[top], std::true_type{});
    {
      // step i reads slot top - i * LANES, going down in memory: the 2 U rows in front of the pad row take the fetches
      // that run past the end
      float ga[U], gb[U];
      auto gslot = [&](int i) { return top - i * LANES; };
#pragma unroll
      for (int u = 0; u < U; ++u) ga[u] = G[gslot(1 + u)];
      int i = 1;
      for (; i + 2 * U <= n2; i += 2 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) gb[u] = G[gslot(i + U + u)];
#pragma unroll
        for (int u = 0; u < U; ++u) flow(top - (i + u) * LANES, ga[u], std::false_type{});
#pragma unroll
        for (int u = 0; u < U; ++u) ga[u] = G[gslot(i + 2 * U + u)];
#pragma unroll
        for (int u = 0; u < U; ++u) flow(top - (i + U + u) * LANES, gb[u], std::false_type{});
      }
      for (; i < n2; ++i) flow(top - i * LANES, G[top - i * LANES], std::false_type{});
    }
    FTR_BSTAMP(8);
  }   // wave 0
  __syncthreads();
  // ---- store the occupancies band shaped: gx_band[b,t,k] = px_grad[b, s0+k, t], gy_band likewise; zero elsewhere.
  // A transition whose source cell lies in front of the cut was handled by chain B (flow kept at the source cell), one
  // whose source lies on or behind the cut by chain A (flow kept at the destination cell).
  for (int i = tid; i < T * r; i += kBandThreads) {
    const int t = div_r(i), k = i - t * r;
    float fx = 0.0f, fy = 0.0f;
    if (t >= tb && t < te) {
      const int s = lo[t] + k;
      if (s >= sb && s <= se) {
        const int dg = MOD ? (t - tb) : (s - sb) + (t - tb);
        if (dg < jm) { const float2 f = O2[slotA(s, t)]; fx = f.x; fy = f.y; }
        else {
          const int tnx = MOD ? t + 1 : t;
          if (s + 1 <= se && tnx <= te && in_band(s + 1, tnx)) fx = O2[slotB(s + 1, tnx)].x;
          if (in_band(s, t + 1)) fy = O2[slotB(s, t + 1)].y;
        }
      }
    }
    gx_g[i] = fx;
    gy_g[i] = fy;
  }
  FTR_BSTAMP(9);
}

// ---------------------------------------------------------------------------------------- the band recursion, streaming
// The same algorithm for bands that do not fit LDS (long utterances: (S + T) LANES 12 bytes > 150 KB): the wavefront-ordered
// operand / ratio / flow arrays live in a global-memory workspace (a few hundred KB per utterance, L2 resident) and the two
// chains stream through them -- consecutive addresses, fetched kStreamSets - 1 groups of four steps (44 steps, ~1.7 us) ahead
// of their use, which hides the global latency the way two groups hide the LDS latency above.  Only lo[] (the band start per
// column) and the cut exchange stay in LDS.  Differences to the LDS kernel, all for the sake of a predicate-free loop with one
// fixed trip count for both chains: every slot is pre-filled with (OX, OY) = (-inf, 0) and G = 0 -- a step on such a slot
// leaves the lane's value unchanged and passes no flow -- and each part is followed by 2 P pad rows (P = 48 steps), so the
// loops simply run P-step blocks past their ends; the band inputs are read straight from global memory by the scatter pass.
constexpr int kStreamSets = 12;                               // register sets of kBandAhead steps each
constexpr int kStreamP = kStreamSets * kBandAhead;            // 48: block of steps per loop iteration
template <int LANES>
__host__ __device__ inline size_t band_stream_rows(int T, int S) { return (size_t)S + T + 4 + 6 * kStreamP; }
template <int LANES>
__host__ __device__ inline size_t band_stream_floats_per_utt(int T, int S) { return 3 * band_stream_rows<LANES>(T, S) * LANES; }
inline size_t band_stream_lds_bytes(int T) { return sizeof(int) * ((size_t)(T + 2) + 32) + sizeof(float) * 48 + 64; }

template <bool MOD, int LANES>
__global__ __launch_bounds__(kBandThreads) void mi_band_stream_kernel(
    const float* __restrict__ pxb, const float* __restrict__ pyb, const int32_t* __restrict__ ranges,
    const int32_t* __restrict__ boundary, float* __restrict__ ws, float* __restrict__ ans, float* __restrict__ gxb,
    float* __restrict__ gyb, int B, int T, int S, int r, unsigned rinv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Bound bd = load_boundary(boundary, b, S, T);
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  const size_t cells_g = (size_t)T * r;
  float* gx_g = gxb + (size_t)b * cells_g;
  float* gy_g = gyb + (size_t)b * cells_g;
  if (Sn <= 0 || Tn <= 0 || Tn == 1) {
    for (size_t i = tid; i < cells_g; i += kBandThreads) { gx_g[i] = 0.0f; gy_g[i] = 0.0f; }
    if (tid == 0) ans[b] = (Sn <= 0 || Tn <= 0) ? 0.0f : ((Sn == 1) ? 0.0f : -INFINITY);
    return;
  }
  const int te = bd.te, tb = bd.tb, sb = bd.sb, se = bd.se;
  const int D = (MOD ? 0 : (Sn - 1)) + (Tn - 1);
  const int jm = D >> 1;
  constexpr int P = kStreamP;
  const int rowA = 2 * P, rowB = rowA + (jm + 1) + 2 * P;          // [2P pad][A part][2P pad][B part][2P pad]
  const int nrows = rowB + (D - jm + 1) + 2 * P;
  // ---- LDS: lo[], cut rows, cut exchange;  global: O2 (float2 per slot), G
  int* lo = reinterpret_cast<int*>(smem);
  int* cutSA = lo + ((T + 2) & ~1); int* cutSB = cutSA + 16;
  float* cutA = reinterpret_cast<float*>(cutSB + 16); float* cutB = cutA + 16; float* occ = cutB + 16;
  float* wsb = ws + (size_t)b * band_stream_floats_per_utt<LANES>(T, S);
  float2* O2 = reinterpret_cast<float2*>(wsb);
  float* G = wsb + 2 * band_stream_rows<LANES>(T, S) * LANES;
  auto div_r = [&](int i) { return rinv ? (int)__umulhi((unsigned)i, rinv) : i; };

  const int32_t* rg = ranges + (size_t)b * T * r;
  for (int t0 = 0; t0 <= T; t0 += 4 * kBandThreads) {
    int v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int t = t0 + u * kBandThreads + tid; v[u] = (t <= T) ? rg[(size_t)min(t, te - 1) * r] : 0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int t = t0 + u * kBandThreads + tid; if (t <= T) lo[t] = v[u]; }
  }
  for (int i = tid; i < nrows * LANES; i += kBandThreads) { O2[i] = make_float2(kNeg, 0.0f); G[i] = 0.0f; }   // neutral slots
  if (tid < 32) cutSA[tid] = -1;
  // the utterance's shift constants (ftr_common.h, Shift) from a fixed sample of the band arrays: 2048 entries of each,
  // per-wave partial sums through the cut exchange area, summed by every thread in the same order
  const float* pxu = pxb + (size_t)b * T * r;
  const float* pyu = pyb + (size_t)b * T * r;
  {
    const unsigned nfr = (unsigned)(te - tb) * (unsigned)r;
    float sx = 0.0f, nx = 0.0f, sy = 0.0f, ny = 0.0f;
    float vx[4], vy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned i = (unsigned)tb * (unsigned)r + __umulhi((unsigned)(tid + kBandThreads * u) * 0x9E3779B1u, nfr);
      vx[u] = pxu[i]; vy[u] = pyu[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (shift_sample_ok(vx[u])) { sx += vx[u]; nx += 1.0f; }
      if (shift_sample_ok(vy[u])) { sy += vy[u]; ny += 1.0f; }
    }
    sx = wave_sum_dpp(sx); nx = wave_sum_dpp(nx); sy = wave_sum_dpp(sy); ny = wave_sum_dpp(ny);
    if (lane == 0) { cutA[wave] = sx; cutA[8 + wave] = nx; cutA[16 + wave] = sy; cutA[24 + wave] = ny; }
  }
  __syncthreads();
  Shift sh;
  {
    float sx = 0.0f, nx = 0.0f, sy = 0.0f, ny = 0.0f;
#pragma unroll
    for (int u = 0; u < kBandThreads / 64; ++u) { sx += cutA[u]; nx += cutA[8 + u]; sy += cutA[16 + u]; ny += cutA[24 + u]; }
    sh = shift_from_sums<MOD>(sx, nx, sy, ny, Sn, Tn);
  }
  const double shift_back = shift_total<MOD>(sh, Sn, Tn);
  {
    int bad = 0;
    for (int t = tb + 1 + tid; t < te; t += kBandThreads) bad |= (lo[t] < lo[t - 1]);
    if (__syncthreads_or(bad)) {     // see the LDS kernel: the band start must not decrease
      for (size_t i = tid; i < cells_g; i += kBandThreads) { gx_g[i] = 0.0f; gy_g[i] = 0.0f; }
      if (tid == 0) ans[b] = __builtin_nanf("");
      return;
    }
  }
  const bool end_above = MOD && se == lo[te] + r;              // see the LDS kernel
  auto in_band = [&](int s, int t) { const int l = lo[t]; return (s >= l && s <= l + r - 1) || (end_above && t == te && s == se); };
  auto slotA = [&](int s, int t) { return (rowA + (MOD ? (t - tb) : (s - sb) + (t - tb))) * LANES + ((s - sb) & (LANES - 1)); };
  auto slotB = [&](int s, int t) { return (rowB + (MOD ? (te - t) : (se - s) + (te - t))) * LANES + ((se - s) & (LANES - 1)); };
  // band value at lattice cell (s, t), frame t < te, log2 domain, shifted
  auto at = [&](const float* src, float c2, int s, int t) { return __builtin_fmaf(src[(size_t)t * r + (s - lo[t])], kLog2e, -c2); };
  unsigned nan_acc = 0;
  auto put = [&](int slot, float x, float y) {
    nan_acc = max(nan_acc, max(__float_as_uint(x) & 0x7fffffffu, __float_as_uint(y) & 0x7fffffffu));
    O2[slot] = make_float2(fmaxf(x, kNeg), fmaxf(y, kNeg));
  };
  // ---- operands of every band cell, both arrays in one pass, straight from global memory
  for (int i = tid; i < Tn * r; i += kBandThreads) {
    const int q = div_r(i);
    const int t = tb + q, k = i - q * r;
    const int s = lo[t] + k;
    if (s < sb || s > se) continue;
    const int dg = MOD ? (t - tb) : (s - sb) + (t - tb);
    if (dg <= jm) {        // chain A: the transitions INTO the cell
      float ax = kNeg, ay = kNeg;
      const int tx = MOD ? t - 1 : t;
      if (s - 1 >= sb && tx >= tb && tx <= te - 1 && in_band(s - 1, tx)) ax = at(pxu, sh.cx2, s - 1, tx);
      if (t - 1 >= tb && in_band(s, t - 1)) ay = at(pyu, sh.cy2, s, t - 1);
      if (s == sb && t == tb) ay = 0.0f;                               // origin trick
      put(slotA(s, t), ax, ay);
      if (dg == jm) cutSA[(s - sb) & (LANES - 1)] = s;
    }
    if (dg >= jm) {        // chain B: the transitions OUT of the cell
      float bx = kNeg, by = kNeg;
      if (t <= te - 1) {
        const int tnx = MOD ? t + 1 : t;
        if (s + 1 <= se && tnx <= te && in_band(s + 1, tnx)) bx = at(pxu, sh.cx2, s, t);
        if (in_band(s, t + 1)) by = at(pyu, sh.cy2, s, t);
      }
      if (s == se && t == te) by = 0.0f;                               // chain B's origin is the end cell
      put(slotB(s, t), bx, by);
      if (dg == jm) cutSB[(se - s) & (LANES - 1)] = s;
    }
  }
  if (end_above && tid == 0) O2[slotB(se, te)] = make_float2(kNeg, 0.0f);
  const bool poisoned = __syncthreads_or(nan_acc > 0x7f800000u) != 0;
  __threadfence();   // the slots were written by other waves of this workgroup: nothing stale in this CU's L1 from here on

  if (wave == 0) {
    constexpr int U = kBandAhead, NS = kStreamSets;
    const int l16 = lane & 15;
    const int lg = l16 & (LANES - 1);
    const bool isB = ((lane >> 4) & 1) == 1;
    float val = (lg == 0) ? 0.0f : kNeg;
    const int n1 = D - jm + 1;
    const int nrun = ((n1 + P - 1) / P) * P;                    // both chains, whole blocks: the surplus steps are neutral
    const int base1 = (isB ? rowB : rowA) * LANES + lg;
    auto fwd = [&](int slot, float2 o) {
      const float up = dpp_row_ror1(val);
      const float a_ = up + o.x, b_ = val + o.y;
      const float d = a_ - b_;
      const float ex = __builtin_amdgcn_exp2f(-__builtin_fabsf(d));
      val = fmaxf(a_, b_) + __builtin_amdgcn_logf(1.0f + ex);
      const float rc = __builtin_amdgcn_rcpf(1.0f + ex);
      G[slot] = (d >= 0.0f) ? rc : ex * rc;
    };
    // renormalisation of the chains' values twice per block of P = 48 steps, see the LDS kernel
    float frame = 0.0f;
    auto renorm = [&](float snap) {
      const float mx = row16_max(snap);
      const float st = (mx > kNegThresh) ? __builtin_rintf(mx) : 0.0f;
      val -= st; frame += st;
    };
    {
      float2 o[NS][U];
#pragma unroll
      for (int k = 0; k < NS - 1; ++k)
#pragma unroll
        for (int u = 0; u < U; ++u) o[k][u] = O2[base1 + (k * U + u) * LANES];
      static_assert((NS & 1) == 0, "the renormalisation pairs the register sets");
      for (int i = 0; i < nrun; i += P) {
        float snap = val;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          const int kf = (k + NS - 1) % NS;                     // the set NS - 1 groups ahead
          if (k == NS / 2 - 2 || k == NS - 2) snap = val;
#pragma unroll
          for (int u = 0; u < U; ++u) o[kf][u] = O2[base1 + (i + (k + NS - 1) * U + u) * LANES];
#pragma unroll
          for (int u = 0; u < U; ++u) fwd(base1 + (i + k * U + u) * LANES, o[k][u]);
          if (k == NS / 2 - 1 || k == NS - 1) renorm(snap);   // every 24 steps, from a copy 8 steps old
        }
      }
    }
    // ---- the cut (LDS), as in the LDS kernel
    if (lane < 48) cutA[lane] = kNeg;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int scut = (isB ? cutSB : cutSA)[lg];
    if (scut >= 0) (isB ? cutB : cutA)[scut & 15] = val;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      // both chains' frames: chain A's lanes hold theirs in DPP rows 0 / 2, chain B's in rows 1 / 3
      const float frA = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, frame), 0));
      const float frB = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, frame), 16));
      const float v = cutA[l16] + cutB[l16];
      const float m = row16_max(v);
      const float e = exp2f(v - m);
      const float sum = row16_sum(e);
      const float total = m + log2f(sum);
      const bool dead = !(total > kNegThresh);
      if (lane < 16) occ[lane] = (dead || poisoned) ? 0.0f : e / sum;
      if (lane == 0) ans[b] = poisoned ? __builtin_nanf("") : (dead ? -INFINITY : (float)(((double)total + (double)frA + (double)frB + shift_back) * 0.6931471805599453));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float inj = (scut >= 0) ? occ[scut & 15] : 0.0f;
    __builtin_amdgcn_s_waitcnt(0);                              // the ratios of both chains are in memory before anybody reads them
    // ---- flow: each chain walks down through the other chain's part (and on into the neutral pad rows)
    float xo = 0.0f, yo = 0.0f;
    const int n2 = D - jm + 1;
    const int nrun2 = ((n2 - 1 + P - 1) / P) * P;               // steps after the first one, whole blocks
    const int top = (isB ? rowA + jm : rowB + D - jm) * LANES + ((Sn - 1 - lg) & (LANES - 1));
    auto flow = [&](int slot, float g, auto first) {
      const float xin = dpp_row_ror1(xo), yin = yo;
      float pg = xin + yin;
      if (decltype(first)::value) pg += inj;
      O2[slot] = make_float2(xin, yin);
      xo = pg * g; yo = pg - xo;
    };
    flow(top, G[top], std::true_type{});
    {
      float g[NS][U];
#pragma unroll
      for (int k = 0; k < NS - 1; ++k)
#pragma unroll
        for (int u = 0; u < U; ++u) g[k][u] = G[top - (1 + k * U + u) * LANES];
      for (int i = 1; i < 1 + nrun2; i += P) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          const int kf = (k + NS - 1) % NS;
#pragma unroll
          for (int u = 0; u < U; ++u) g[kf][u] = G[top - (i + (k + NS - 1) * U + u) * LANES];
#pragma unroll
          for (int u = 0; u < U; ++u) flow(top - (i + k * U + u) * LANES, g[k][u], std::false_type{});
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
  }   // wave 0
  __syncthreads();
  __threadfence();   // wave 0's flows, read by everybody (this CU's L1 may hold the operand values of the same slots)
  for (int i = tid; i < T * r; i += kBandThreads) {
    const int t = div_r(i), k = i - t * r;
    float fx = 0.0f, fy = 0.0f;
    if (t >= tb && t < te) {
      const int s = lo[t] + k;
      if (s >= sb && s <= se) {
        const int dg = MOD ? (t - tb) : (s - sb) + (t - tb);
        if (dg < jm) { const float2 f = O2[slotA(s, t)]; fx = f.x; fy = f.y; }
        else {
          const int tnx = MOD ? t + 1 : t;
          if (s + 1 <= se && tnx <= te && in_band(s + 1, tnx)) fx = O2[slotB(s + 1, tnx)].x;
          if (in_band(s, t + 1)) fy = O2[slotB(s, t + 1)].y;
        }
      }
    }
    gx_g[i] = fx;
    gy_g[i] = fy;
  }
}

// ---------------------------------------------------------------------------------------- gradient w.r.t. logits
// the band_grad_kernel of pruned_logprobs.hip with the occupancies read band shaped (row = (b,t,k)); one wave per row.
template <bool VEC>
__global__ void band_grad_banded_kernel(const float* __restrict__ logits, const int32_t* __restrict__ symbols,
                                        const int32_t* __restrict__ ranges, const int32_t* __restrict__ boundary,
                                        const float* __restrict__ lse, const float* __restrict__ gxb,
                                        const float* __restrict__ gyb, const Scale scale, int blank, int modified,
                                        float* __restrict__ glogits, size_t rows, int T, int S, int C, int r) {
  const int lane = threadIdx.x & 63;
  // LAST ROWS FIRST: the caller's backward reads `glogits` front to back right after this kernel, and what a 320 MB stream
  // leaves in the 256 MB memory-side cache is what was written last (see lse_rows_reg_kernel, pruned_logprobs.hip): 121 -> 117 us
  // here and 2 us off the caller's next kernel in the c3 step (scripts/order_ab.sh)
  const size_t rowi = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (rowi >= rows) return;
  const size_t row = rows - 1 - rowi;
  const size_t bt = row / r;
  const int k = (int)(row - bt * r);
  const int b = (int)(bt / T);
  const int t = (int)(bt - (size_t)b * T);
  const int s = ranges[bt * r] + k;
  const int te = boundary ? boundary[4 * b + 3] : T;
  const float sc = scale.at(b);
  float gx = 0.0f;
  int sym = blank;
  const bool sok = s >= 0 && s <= S;
  if (sok && s < S) {
    sym = symbols[(size_t)b * S + s];
    if (modified || t != te) gx = gxb[row] * sc;
  }
  const float gy = sok ? gyb[row] * sc : 0.0f;
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
        const int cc = 4 * i + e;
        float val = -tot * __expf(v[e] - l);
        if (cc == sym) val += gx;
        if (cc == blank) val += gy;
        o[e] = val;
      }
      g4[i] = o;
    }
  } else {
    for (int cc = lane; cc < C; cc += 64) {
      float val = -tot * __expf(x[cc] - l);
      if (cc == sym) val += gx;
      if (cc == blank) val += gy;
      g[cc] = val;
    }
  }
}

}  // namespace

// 8-lane chains while the band is at most 7 rows wide (half the LDS), 16-lane chains up to 15 rows
static int band_lanes(int T, int S, int r) {
  if (r < 1 || T < 1 || S < 0) return 0;
  // r + 1 lanes: the modified type may have r + 1 cells on its last walk step (the end cell one row above the last
  // frame's band, see the kernel)
  if (r <= 7 && band_lds_bytes<8>(T, S) <= (size_t)150 * 1024) return 8;
  if (r <= 15 && band_lds_bytes<16>(T, S) <= (size_t)150 * 1024) return 16;
  return 0;
}
// 0: not supported; 1: the LDS-resident kernel; 2: the streaming kernel (needs mi_band_workspace_floats() floats of workspace)
static int band_stream_lanes(int T, int r) {
  if (r < 1 || r > 15 || band_stream_lds_bytes(T) > (size_t)150 * 1024) return 0;
  return r <= 7 ? 8 : 16;
}
int mi_band_supported(int T, int S, int r) {
  static const bool force_stream = getenv("FTR_BAND_FORCE_STREAM") != nullptr;   // test knob: every size through the streaming kernel
  if (!force_stream && band_lanes(T, S, r) != 0) return 1;
  return (T >= 1 && S >= 0 && band_stream_lanes(T, r) != 0) ? 2 : 0;
}
// the chain kernels' need (0 for the LDS-resident one)
static size_t band_chain_workspace_floats(int B, int T, int S, int r) {
  if (mi_band_supported(T, S, r) != 2) return 0;
  const size_t per = band_stream_lanes(T, r) == 8 ? band_stream_floats_per_utt<8>(T, S) : band_stream_floats_per_utt<16>(T, S);
  return per * (size_t)B;
}
// what a caller should provide: enough for the segmented route (mi_band_seg.hip) where it applies, else for the chain kernels
size_t mi_band_workspace_floats(int B, int T, int S, int r) {
  const size_t chain = band_chain_workspace_floats(B, T, S, r), seg = mi_band_seg_workspace_floats(B, T, S, r);
  return seg > chain ? seg : chain;
}

int band_ranges_check(const int32_t* ranges, const int32_t* boundary, int* flags, int B, int T, int r, hipStream_t st) {
  const int rcz = zero_words(flags, 1, st, "band_ranges_check");
  if (rcz != FTR_OK) return rcz;
  const size_t n = (size_t)B * T;
  if (n == 0) return FTR_OK;
  hipLaunchKernelGGL(band_ranges_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ranges, boundary, flags, B, T, r);
  return check_launch("band_ranges_check");
}

int band_gather(const float* logits, const int32_t* symbols, const int32_t* ranges, const int32_t* boundary,
                const float* lse, int blank, double delay_penalty, float* pxb, float* pyb, int B, int T, int S, int C,
                int r, int modified, hipStream_t st) {
  const size_t rows = (size_t)B * T * r;
  if (rows == 0) return FTR_OK;
  const unsigned blocks = (unsigned)((rows + 255) / 256);
  if (modified) hipLaunchKernelGGL(band_gather_kernel<true>, dim3(blocks), dim3(256), 0, st, logits, symbols, ranges, boundary, lse, blank, delay_penalty, pxb, pyb, rows, T, S, C, r);
  else hipLaunchKernelGGL(band_gather_kernel<false>, dim3(blocks), dim3(256), 0, st, logits, symbols, ranges, boundary, lse, blank, delay_penalty, pxb, pyb, rows, T, S, C, r);
  return check_launch("band_gather");
}

int mi_band(const float* pxb, const float* pyb, const int32_t* ranges, const int32_t* boundary, float* ws, size_t ws_floats,
            float* ans, float* gxb, float* gyb, int B, int T, int S, int r, int modified, hipStream_t st) {
  const int kind = mi_band_supported(T, S, r);
  if (!kind) { set_error("mutual_information_band: T=%d S=%d r=%d is outside the band kernels' domain (r <= 15)", T, S, r); return FTR_ERR_UNSUPPORTED; }
  // the segmented route (mi_band_seg.hip) whenever the caller's workspace is large enough for it
  if (B <= 65535 && mi_band_seg_supported(T, S, r) && ws && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 && ws_floats >= mi_band_seg_workspace_floats(B, T, S, r))
    return mi_band_seg(pxb, pyb, ranges, boundary, ws, ws_floats, ans, gxb, gyb, B, T, S, r, modified, st);
  if ((uint64_t)(T + 1) * r * r >= (1ull << 32)) { set_error("mutual_information_band: T * r too large"); return FTR_ERR_UNSUPPORTED; }
  const unsigned rinv = (r == 1) ? 0u : (unsigned)(((1ull << 32) + r - 1) / r);   // i / r == umulhi(i, rinv) while i * (r - 1) < 2^32; 0 stands for r = 1
  if (kind == 2) {
    const size_t need = band_chain_workspace_floats(B, T, S, r);
    if (!ws || ws_floats < need || (reinterpret_cast<uintptr_t>(ws) & 15) != 0) {
      set_error("mutual_information_band: this size streams through a workspace of %zu floats (16-byte aligned), got %zu", need, ws_floats);
      return FTR_ERR_INVALID_ARG;
    }
    const size_t lds = band_stream_lds_bytes(T);
    static bool big_ok = false;
    if (!big_ok) {
      const void* ks[4] = {reinterpret_cast<const void*>(mi_band_stream_kernel<true, 8>), reinterpret_cast<const void*>(mi_band_stream_kernel<false, 8>),
                           reinterpret_cast<const void*>(mi_band_stream_kernel<true, 16>), reinterpret_cast<const void*>(mi_band_stream_kernel<false, 16>)};
      for (const void* k : ks)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) {
          (void)hipGetLastError(); set_error("mutual_information_band: cannot raise the dynamic LDS limit"); return FTR_ERR_LAUNCH;
        }
      big_ok = true;
    }
#define FTR_BANDS_LAUNCH(MODV, LV) hipLaunchKernelGGL((mi_band_stream_kernel<MODV, LV>), dim3(B), dim3(kBandThreads), lds, st, \
    pxb, pyb, ranges, boundary, ws, ans, gxb, gyb, B, T, S, r, rinv)
    if (band_stream_lanes(T, r) == 8) { if (modified) FTR_BANDS_LAUNCH(true, 8); else FTR_BANDS_LAUNCH(false, 8); }
    else { if (modified) FTR_BANDS_LAUNCH(true, 16); else FTR_BANDS_LAUNCH(false, 16); }
#undef FTR_BANDS_LAUNCH
    return check_launch("mi_band_stream");
  }
  const int lanes = band_lanes(T, S, r);
  static bool big_ok = false;
  if (!big_ok) {   // 156 KB: the kernel also has a few bytes of static LDS (__syncthreads_or)
    const void* ks[4] = {reinterpret_cast<const void*>(mi_band_kernel<true, 8>), reinterpret_cast<const void*>(mi_band_kernel<false, 8>),
                         reinterpret_cast<const void*>(mi_band_kernel<true, 16>), reinterpret_cast<const void*>(mi_band_kernel<false, 16>)};
    for (const void* k : ks)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) {
        (void)hipGetLastError(); set_error("mutual_information_band: cannot raise the dynamic LDS limit"); return FTR_ERR_LAUNCH;
      }
    big_ok = true;
  }
#define FTR_BAND_LAUNCH(MODV, LV) hipLaunchKernelGGL((mi_band_kernel<MODV, LV>), dim3(B), dim3(kBandThreads), band_lds_bytes<LV>(T, S), st, \
    pxb, pyb, ranges, boundary, ans, gxb, gyb, B, T, S, r, rinv)
  if (lanes == 8) { if (modified) FTR_BAND_LAUNCH(true, 8); else FTR_BAND_LAUNCH(false, 8); }
  else { if (modified) FTR_BAND_LAUNCH(true, 16); else FTR_BAND_LAUNCH(false, 16); }
#undef FTR_BAND_LAUNCH
  return check_launch("mi_band");
}

#ifdef FTR_BAND_STAMPS
extern "C" int ftr_debug_band_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bstamp), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

int band_grad_banded(const float* logits, const int32_t* symbols, const int32_t* ranges, const int32_t* boundary,
                     int blank, const float* lse, const float* gxb, const float* gyb, Scale scale, float* glogits, int B,
                     int T, int S, int C, int r, int modified, hipStream_t st) {
  const size_t rows = (size_t)B * T * r;
  if (rows == 0) return FTR_OK;
  const int wpb = 4;
  const unsigned blocks = (unsigned)((rows + wpb - 1) / wpb);
  if ((C & 3) == 0) hipLaunchKernelGGL(band_grad_banded_kernel<true>, dim3(blocks), dim3(64 * wpb), 0, st, logits, symbols, ranges, boundary, lse, gxb, gyb, scale, blank, modified, glogits, rows, T, S, C, r);
  else hipLaunchKernelGGL(band_grad_banded_kernel<false>, dim3(blocks), dim3(64 * wpb), 0, st, logits, symbols, ranges, boundary, lse, gxb, gyb, scale, blank, modified, glogits, rows, T, S, C, r);
  return check_launch("band_grad_banded");
}

}  // namespace ftr
