// csrc/mi_wave_mono.hip -- "wavefront" mutual-information kernels, single-role variant: every wave does its own
// staging, recursion and write-out.  Used for lattices of more than 6 row bands (S+1 > 384), where the
// specialised compute/IO pair of mi_wave_duo.hip does not fit the 160 KB LDS.  The design notes below apply to both.
//
// What they compute is the recursion of the reference (tf_fast_rnnt/csrc/mutual_information.h:101-126,
// mutual_information_cuda.cu:174-422 forward, :441-760 backward); HOW is different by design:
//
//  * one workgroup per utterance, one wave64 per 64 lattice rows, lane l <-> row s.  The wave walks the
//    lattice in time-skewed order: at local step j lane l sits on column c = j - l (regular) or c = j
//    (modified), so every lane's two predecessors were produced one step earlier: its own previous
//    value (p[s,t-1]) and the neighbouring lane's previous value (p[s-1,t] / p[s-1,t-1]), fetched
//    with one full-wave DPP shift (wave_shr:1).  The whole dependent chain of a step is
//    mov_dpp, add, sub/max, v_exp_f32, add, v_log_f32, add -- no LDS, no barrier, no global memory.
//    (The reference runs this part on 32 lanes of one warp per 32x32 tile and relaunches the kernel
//    once per tile diagonal.)
//  * values are kept in the log2 domain (inputs are multiplied by log2(e) when they are staged), so
//    the hardware v_exp_f32 / v_log_f32 are used bare; -inf is represented by -1e30 inside the
//    kernel so no NaN guard sits on the chain (LogAdd's "diff - diff != 0" branch,
//    mutual_information.h:79-80, becomes unnecessary), and is turned back into -inf on the way out.
//  * px/py are fetched with coalesced 16-byte loads (4 lanes per 64-byte row segment) two chunks of
//    16 steps ahead into registers, then parked in a per-wave LDS tile laid out [quad][row] (plane
//    stride 66 x 16 B) so both the fill and the per-lane ds_read_b128 are bank-conflict free.  Tiles
//    are already skewed: lane l finds "its" four next steps at tile[q][l].
//  * the forward does not store p.  It stores, per cell, G = sigmoid(a - b): the share of the cell's
//    probability that arrived through the px edge.  That is exactly term1 of the incoming edge in the
//    reference's backward (mutual_information_cuda.cu:455-457) and 1 - G is term2.  The backward then
//    needs no exp at all: it pushes occupancy "flow" down the lattice, pg = xin + yin, xout = pg * G,
//    yout = pg - xout; px_grad = xin, py_grad = yin (eqs. 3a-3c of the reference, .cu:474-477).  One
//    lattice is written by the forward and one is read by the backward (the reference writes p and
//    p_grad and reads px, py, p again), and flow is conserved to rounding.
//  * neighbouring waves exchange their boundary row through a 64-entry LDS ring per wave pair; waves
//    run the same chunk schedule staggered by 5 chunks (regular) or 1 (modified) with one
//    __syncthreads() per chunk, which is what makes the ring race free (see RING below).
//
// Workspace ("p" in the C ABI): B*(S+1)*(T+1) floats holding G for every in-boundary cell.
#include "ftr_common.h"
#include "mi_wave_common.h"

namespace ftr {
using namespace wavecfg;
namespace {

// RING.  Wave w (the "producer") publishes, every 4 steps, the last 4 values of its lane 63 at
// ring[w+1][(j0 & 63) .. +3], j0 = its local step.  Wave w+1 (the "consumer") needs, at ITS local step
// j, the producer's value of local step j + 63 (regular; same column, one row up) or j - 1 (modified;
// previous column): in both cases ring[(j - 1) & 63].  With the consumer's chunks running STG chunks
// behind the producer's (same chunk index k executes STG slots later) every value is written at least
// one barrier before it is read and is overwritten (64 steps later) at least two barriers after it
// was read -- for STG = 5 (regular), 1 (modified); the arithmetic is in DESIGN.md "ring schedule".

// MAXW = maximum waves per workgroup of this instantiation: the launch bound 64*MAXW is what sizes the
// register budget (512 / 256 / 128 VGPRs per lane for MAXW = 4 / 8 / 16).
template <bool MOD, int MAXW>
__global__ __launch_bounds__(64 * MAXW) void mi_wave_fwd_kernel(
    const float* __restrict__ px, const float* __restrict__ py, const int32_t* __restrict__ boundary,
    float* __restrict__ ws, float* __restrict__ ans, int S, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SKEW = MOD ? 0 : 1;
  constexpr int STG = MOD ? 1 : 5;
  constexpr int NPF = Prefetch<MAXW>::N;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NW = blockDim.x >> 6;
  const Bound bd = load_boundary(boundary, b, S, T);
  const int T1 = MOD ? T : T + 1;
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  if (Sn <= 0 || Tn <= 0) { if (threadIdx.x == 0) ans[b] = 0.0f; return; }

  f4* tiles = reinterpret_cast<f4*>(smem);
  f4* tX = tiles + (size_t)(w * 2 + 0) * TILE_F4;
  f4* tY = tiles + (size_t)(w * 2 + 1) * TILE_F4;
  float* rings = reinterpret_cast<float*>(tiles + (size_t)NW * 2 * TILE_F4);
  for (int i = threadIdx.x; i < (NW + 1) * RINGN; i += blockDim.x) rings[i] = kNeg;
  __syncthreads();
  const f4* ring_in = reinterpret_cast<const f4*>(rings + w * RINGN);
  f4* ring_out = reinterpret_cast<f4*>(rings + (w + 1) * RINGN);

  const float* pxb = px + (size_t)b * S * T1;
  const float* pyb = py + (size_t)b * (S + 1) * T;
  float* wsb = ws + (size_t)b * (S + 1) * (T + 1);

  const int row0 = 64 * w;
  // staging geometry of this lane: in load/store instruction m it handles tile row 16m + (lane>>2),
  // quad (lane & 3).
  const int frow = lane >> 2, fq = lane & 3;

  const int nchunks = (Tn + 63 * SKEW + CH - 1) / CH;
  const int NWact = (Sn + 63) >> 6;
  const int Gtot = nchunks + STG * (NWact - 1);

  // where the answer appears: row Sn-1, column Tn-1
  const int wfin = (Sn - 1) >> 6, lfin = (Sn - 1) & 63;
  const int jfin = (w == wfin) ? (Tn - 1 + SKEW * lfin) : -1000;

  float pcur = (w == 0 && lane == 0) ? 0.0f : kNeg;  // origin trick: p[sb,tb] = 0 + (Y := 0)
  float ecarry = kNeg;

  f4 rx[NPF][4], ry[NPF][4];

  auto load_chunk = [&](int k, f4 (&x)[4], f4 (&y)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 16 * m + frow;
      const int r = row0 + row;
      const int c0 = CH * k + 4 * fq - SKEW * row;  // column (relative to tb) of the quad's first step
      f4 vx = {kNeg, kNeg, kNeg, kNeg}, vy = {kNeg, kNeg, kNeg, kNeg};
      if (r < Sn) {
        if (r >= 1) {  // px[s-1][t + toff], toff = -1 for modified
          const int cx = MOD ? c0 - 1 : c0;
          const ptrdiff_t base = (ptrdiff_t)(bd.sb + r - 1) * T1 + bd.tb + cx;
          if (cx >= 0 && c0 + 3 < Tn) {
            vx = *reinterpret_cast<const f4u*>(pxb + base);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (cx + e >= 0 && c0 + e < Tn) vx[e] = pxb[base + e];
          }
        }
        {  // py[s][t-1]
          const ptrdiff_t base = (ptrdiff_t)(bd.sb + r) * T + bd.tb + c0 - 1;
          if (c0 >= 1 && c0 + 3 < Tn) {
            vy = *reinterpret_cast<const f4u*>(pyb + base);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c0 + e >= 1 && c0 + e < Tn) vy[e] = pyb[base + e];
          }
        }
      }
      x[m] = vx;
      y[m] = vy;
    }
  };

  // kk = chunk being parked.  The origin cell (row s_begin, column t_begin: chunk 0, tile row 0, quad 0,
  // element 0 of wave 0) gets Y := 0 so that p = logadd(-inf, pcur(0) + 0) = 0 falls out of the recursion.
  // (Patched here, after the loads have landed anyway, never right behind the load issue.)
  auto write_tile = [&](int kk, const f4 (&x)[4], const f4 (&y)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 16 * m + frow;
      f4 xs, ys;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xs[e] = fmaxf(x[m][e] * kLog2e, kNeg);  // log2 domain; -inf (and nan) -> kNeg
        ys[e] = fmaxf(y[m][e] * kLog2e, kNeg);
      }
      if (m == 0 && kk == 0 && w == 0 && lane == 0) ys[0] = 0.0f;
      tX[fq * PLANE + row] = xs;
      tY[fq * PLANE + row] = ys;
    }
  };

  // Lane 0 has no left neighbour in the wave: its "up" value comes from the ring.  It is folded into X
  // (off the dependent chain) so the DPP shift can run with bound_ctrl (lane 0 reads 0) and fuse into
  // the add: a = v_add_f32_dpp(X', pcur).
  const float lane0 = (lane == 0) ? 1.0f : 0.0f;
  auto compute_chunk = [&](int k) {
    f4 Xn = tX[lane], Yn = tY[lane];
    f4 En = ring_in[((CH * k) & (RINGN - 1)) >> 2];  // same address in every lane (broadcast)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int j0 = CH * k + 4 * q;
      const f4 X4 = Xn, Y4 = Yn, E4 = En;
      if (q + 1 < NQ) {  // next quad's operands are fetched while this quad's chain runs
        Xn = tX[(q + 1) * PLANE + lane];
        Yn = tY[(q + 1) * PLANE + lane];
        En = ring_in[((j0 + 4) & (RINGN - 1)) >> 2];
      }
      f4 XE;  // X + (lane 0 ? value from the wave above : 0)
      XE[0] = X4[0] + lane0 * ecarry; XE[1] = X4[1] + lane0 * E4[0];
      XE[2] = X4[2] + lane0 * E4[1];  XE[3] = X4[3] + lane0 * E4[2];
      f4 G4, P4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, pcur), 0x138, 0xf, 0xf, true));
        const float a = up + XE[e];
        const float c = pcur + Y4[e];
        const float d = a - c;
        const float mx = fmaxf(a, c);
        const float ex = __builtin_amdgcn_exp2f(-__builtin_fabsf(d));
        const float u = 1.0f + ex;
        pcur = mx + __builtin_amdgcn_logf(u);
        const float rc = __builtin_amdgcn_rcpf(u);
        G4[e] = (d >= 0.0f) ? rc : ex * rc;
        P4[e] = pcur;
      }
      ecarry = E4[3];
      tX[q * PLANE + lane] = G4;  // in place: this slot of X has been consumed
      if (lane == 63) ring_out[(j0 & (RINGN - 1)) >> 2] = P4;
      if ((jfin >> 2) == (j0 >> 2)) {  // wave-uniform
        const int e = jfin & 3;
        const float v = (e == 0) ? P4[0] : (e == 1) ? P4[1] : (e == 2) ? P4[2] : P4[3];
        if (lane == lfin) ans[b] = (v <= kNegThresh) ? -INFINITY : v * kLn2;
      }
    }
  };

  auto fetch_G = [&](f4 (&gq)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) gq[m] = tX[fq * PLANE + 16 * m + frow];
  };
  auto store_G = [&](int k, const f4 (&gq)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 16 * m + frow;
      const int r = row0 + row;
      if (r < Sn) {
        const int c0 = CH * k + 4 * fq - SKEW * row;
        const f4 g = gq[m];
        const ptrdiff_t base = (ptrdiff_t)(bd.sb + r) * (T + 1) + bd.tb + c0;
        if (c0 >= 0 && c0 + 3 < Tn) {
          *reinterpret_cast<f4u*>(wsb + base) = g;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e >= 0 && c0 + e < Tn) wsb[base + e] = g[e];
        }
      }
    }
  };

  // ---- interior ("fast") chunks: every quad of every lane-row lies inside [1, Tn) in columns, so loads and
  // stores are plain 16-byte accesses with no per-element guards and no divergent control flow.  Rows
  // beyond the utterance are clamped to a valid row: what they compute never reaches a valid row (data
  // only moves from row s-1 to row s) and is never stored.
  int offX[4], offY[4], offG[4];
  bool rvalid[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int row = 16 * m + frow;
    const int r = row0 + row;
    const int cq = 4 * fq - SKEW * row;
    const int rxc = min(max(r - 1, 0), max(Sn - 2, 0));   // px row s-1 (clamped)
    const int ryc = min(r, Sn - 1);                       // py row s   (clamped)
    offX[m] = (bd.sb + rxc) * T1 + bd.tb + cq + (MOD ? -1 : 0);
    offY[m] = (bd.sb + ryc) * T + bd.tb + cq - 1;
    offG[m] = (bd.sb + r) * (T + 1) + bd.tb + cq;
    rvalid[m] = r < Sn;
  }
  auto load_fast = [&](int k, f4 (&x)[4], f4 (&y)[4]) {
    const float* px_k = pxb + CH * k;   // wave-uniform part of the address
    const float* py_k = pyb + CH * k;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      x[m] = *reinterpret_cast<const f4u*>(px_k + offX[m]);
      y[m] = *reinterpret_cast<const f4u*>(py_k + offY[m]);
    }
  };
  auto store_fast = [&](int k, const f4 (&gq)[4]) {
    float* ws_k = wsb + CH * k;
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (rvalid[m]) *reinterpret_cast<f4u*>(ws_k + offG[m]) = gq[m];
  };

  auto slot_general = [&](int k, f4 (&x)[4], f4 (&y)[4]) {
    const bool active = (k >= 0 && k < nchunks);
    f4 gq[4];
    if (active) {
      if (!MOD && k == 0) ecarry = rings[w * RINGN + 63];
      compute_chunk(k);
      fetch_G(gq);  // G quads out of the tile before it is refilled
    }
    // park the next chunk first (its loads were issued NPF slots ago), only then issue this chunk's
    // stores and the next prefetch
    if (k + 1 >= 0 && k + 1 < nchunks) write_tile(k + 1, x, y);
    if (active) store_G(k, gq);
    if (k + 1 + NPF >= 0 && k + 1 + NPF < nchunks) load_chunk(k + 1 + NPF, x, y);
    __syncthreads();
  };
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  auto slot_fast = [&](int k, f4 (&x)[4], f4 (&y)[4]) {
    f4 gq[4];
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    FTR_STAMP(t0);
    compute_chunk(k);
    FTR_STAMP(t1);
    fetch_G(gq);
    write_tile(k + 1, x, y);
    FTR_STAMP(t2);
    store_fast(k, gq);
    FTR_STAMP(t3);
    load_fast(k + 1 + NPF, x, y);
    FTR_STAMP(t4);
    __syncthreads();
    FTR_STAMP(t5);
    st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += t3 - t2; st_acc[3] += t4 - t3; st_acc[4] += t5 - t4; st_acc[5] += 1;
  };

  // Slot schedule.  Iteration `it` runs NPF slots; this wave's chunk in the first of them is
  // k0 = NPF*it - (NPF+1) - STG*w.  Every wave runs exactly NIT iterations (one barrier per slot); each wave
  // splits them into [general | fast | general] by its own k0.  Fast slot k: stores of k and loads of
  // k+1+NPF are interior  <=>  K0 <= k and k+1+NPF < K1; k >= 1 keeps the origin special cases out.
  const int K0 = MOD ? 1 : 4;                            // 16k - 63*SKEW >= 1
  const int K1 = (Tn >= CH) ? (Tn - CH) / CH + 1 : 0;    // 16k + 15 < Tn
  const int KF0 = K0, KF1 = (Sn >= 2) ? K1 - 1 - NPF : 0;
  const int NIT = (Gtot + NPF + 1 + NPF - 1) / NPF;
  const int base = -(NPF + 1) - STG * w;                 // k0 = base + NPF*it
  int it1 = (KF0 - base + NPF - 1) / NPF;                // first it with k0 >= KF0
  int it2 = (KF1 - NPF + 1 - base + NPF - 1) / NPF;      // first it with k0 + NPF - 1 >= KF1
  it1 = min(max(it1, 0), NIT);
  it2 = min(max(it2, it1), NIT);

  int it = 0;
  for (; it < it1; ++it) {
#pragma unroll
    for (int u = 0; u < NPF; ++u) slot_general(base + NPF * it + u, rx[u], ry[u]);
  }
  if (it < it2) {
    __builtin_amdgcn_s_waitcnt(kVmcnt0);  // nothing pending when the steady-state loop is entered
    for (; it < it2; ++it) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) slot_fast(base + NPF * it + u, rx[u], ry[u]);
    }
  }
  for (; it < NIT; ++it) {
#pragma unroll
    for (int u = 0; u < NPF; ++u) slot_general(base + NPF * it + u, rx[u], ry[u]);
  }
#ifdef FTR_STAMPS
  if (b == 0 && threadIdx.x == 0)
    for (int i = 0; i < 6; ++i) g_stamps[i] = st_acc[i];
#endif
}

// Backward on reversed coordinates: row index r = s_end - s (lane), column c = t_end - t.
template <bool MOD, int MAXW>
__global__ __launch_bounds__(64 * MAXW) void mi_wave_bwd_kernel(
    const int32_t* __restrict__ boundary, const float* __restrict__ ws, float* __restrict__ px_grad,
    float* __restrict__ py_grad, float* __restrict__ ans_grad, int overwrite, int S, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SKEW = MOD ? 0 : 1;
  constexpr int STG = MOD ? 1 : 5;
  constexpr int NOFF = MOD ? 1 : 0;
  constexpr int NPF = Prefetch<MAXW>::N;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NW = blockDim.x >> 6;
  const Bound bd = load_boundary(boundary, b, S, T);
  const int T1 = MOD ? T : T + 1;
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  float* pxg = px_grad + (size_t)b * S * T1;
  float* pyg = py_grad + (size_t)b * (S + 1) * T;

  // ---- zeros outside the boundary rectangle (the reference memsets everything first,
  //      tf_fast_rnnt_op.cc:93-96); the rectangle itself is fully written by the sweep below.
  {
    const bool empty = (Sn <= 0 || Tn <= 0);
    // px_grad is defined on rows [sb, se) x columns [tb, te - NOFF]
    const int xr0 = empty ? 0 : bd.sb, xr1 = empty ? 0 : bd.se;
    const int xc0 = bd.tb, xc1 = bd.te - NOFF + 1;
    for (int s = w; s < S; s += NW) {
      float* row = pxg + (size_t)s * T1;
      if (s < xr0 || s >= xr1) {
        for (int t = lane; t < T1; t += 64) row[t] = 0.0f;
      } else {
        for (int t = lane; t < xc0; t += 64) row[t] = 0.0f;
        for (int t = xc1 + lane; t < T1; t += 64) row[t] = 0.0f;
      }
    }
    // py_grad is defined on rows [sb, se] x columns [tb, te)
    const int yr0 = empty ? 0 : bd.sb, yr1 = empty ? 0 : bd.se + 1;
    for (int s = w; s < S + 1; s += NW) {
      float* row = pyg + (size_t)s * T;
      if (s < yr0 || s >= yr1) {
        for (int t = lane; t < T; t += 64) row[t] = 0.0f;
      } else {
        for (int t = lane; t < bd.tb; t += 64) row[t] = 0.0f;
        for (int t = bd.te + lane; t < T; t += 64) row[t] = 0.0f;
      }
    }
    if (empty) return;
  }

  f4* tiles = reinterpret_cast<f4*>(smem);
  f4* tG = tiles + (size_t)(w * 2 + 0) * TILE_F4;  // G in, px_grad quads out (in place)
  f4* tO = tiles + (size_t)(w * 2 + 1) * TILE_F4;  // py_grad quads out
  float* rings = reinterpret_cast<float*>(tiles + (size_t)NW * 2 * TILE_F4);
  for (int i = threadIdx.x; i < (NW + 1) * RINGN; i += blockDim.x) rings[i] = 0.0f;
  __syncthreads();
  const f4* ring_in = reinterpret_cast<const f4*>(rings + w * RINGN);
  f4* ring_out = reinterpret_cast<f4*>(rings + (w + 1) * RINGN);

  const float* wsb = ws + (size_t)b * (S + 1) * (T + 1);
  const int row0 = 64 * w;
  const int frow = lane >> 2, fq = lane & 3;
  const int nchunks = (Tn + 63 * SKEW + CH - 1) / CH;
  const int NWact = (Sn + 63) >> 6;
  const int Gtot = nchunks + STG * (NWact - 1);
  const int wfin = (Sn - 1) >> 6, lfin = (Sn - 1) & 63;
  const int jfin = (w == wfin) ? (Tn - 1 + SKEW * lfin) : -1000;  // where p_grad[sb,tb] appears

  float yprev = (w == 0 && lane == 0) ? ans_grad[b] : 0.0f;  // seeds p_grad[se,te] = ans_grad
  float xprev = 0.0f;
  float ecarry = 0.0f;
  f4 rg[NPF][4];

  auto load_chunk = [&](int k, f4 (&gq)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 16 * m + frow;
      const int r = row0 + row;
      const int c0 = CH * k + 4 * fq - SKEW * row;
      f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (r < Sn) {
        // element e is column c0+e reversed: t = te - c0 - e; memory order is the reverse of e.
        const ptrdiff_t lo = (ptrdiff_t)(bd.se - r) * (T + 1) + bd.te - c0 - 3;
        if (c0 >= 0 && c0 + 3 < Tn) {
          const f4 t4 = *reinterpret_cast<const f4u*>(wsb + lo);
          v[0] = t4[3]; v[1] = t4[2]; v[2] = t4[1]; v[3] = t4[0];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e >= 0 && c0 + e < Tn) v[e] = wsb[lo + 3 - e];
        }
      }
      gq[m] = v;
    }
  };

  auto write_tile = [&](const f4 (&gq)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) tG[fq * PLANE + 16 * m + frow] = gq[m];
  };

  auto compute_chunk = [&](int k) {
    f4 Gn = tG[lane];
    f4 En = ring_in[((CH * k) & (RINGN - 1)) >> 2];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int j0 = CH * k + 4 * q;
      const f4 G4 = Gn, E4 = En;
      if (q + 1 < NQ) {
        Gn = tG[(q + 1) * PLANE + lane];
        En = ring_in[((j0 + 4) & (RINGN - 1)) >> 2];
      }
      f4 XO4, PX4, PY4, PG4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ev = (e == 0) ? ecarry : E4[e - 1];
        const float xin = dpp_wave_shr1(ev, xprev);
        const float pg = xin + yprev;
        PX4[e] = xin;    // px_grad[s,t]  = p_grad[s+1,t(+1)] * term1(s,t)   (3b)
        PY4[e] = yprev;  // py_grad[s,t]  = p_grad[s,t+1]     * term2(s,t)   (3c)
        PG4[e] = pg;     // p_grad[s,t]                                     (3a)
        xprev = pg * G4[e];
        yprev = pg - xprev;
        XO4[e] = xprev;
      }
      ecarry = E4[3];
      tG[q * PLANE + lane] = PX4;
      tO[q * PLANE + lane] = PY4;
      if (lane == 63) ring_out[(j0 & (RINGN - 1)) >> 2] = XO4;
      if (overwrite && (jfin >> 2) == (j0 >> 2)) {
        const int e = jfin & 3;
        const float v = (e == 0) ? PG4[0] : (e == 1) ? PG4[1] : (e == 2) ? PG4[2] : PG4[3];
        if (lane == lfin) ans_grad[b] = v;
      }
    }
  };

  auto fetch_out = [&](f4 (&oq)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) oq[m] = tG[fq * PLANE + 16 * m + frow];
  };
  auto store_out = [&](int k, const f4 (&oq)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 16 * m + frow;
      const int r = row0 + row;
      if (r < Sn) {
        const int c0 = CH * k + 4 * fq - SKEW * row;
        const int s = bd.se - r;
        const f4 gx = oq[m];
        const f4 gy = tO[fq * PLANE + row];
        if (r >= 1) {  // px_grad rows are s < se; columns c in [NOFF, Tn)
          const ptrdiff_t lo = (ptrdiff_t)s * T1 + bd.te - c0 - 3;
          if (c0 >= NOFF && c0 + 3 < Tn) {
            f4 o; o[0] = gx[3]; o[1] = gx[2]; o[2] = gx[1]; o[3] = gx[0];
            *reinterpret_cast<f4u*>(pxg + lo) = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c0 + e >= NOFF && c0 + e < Tn) pxg[lo + 3 - e] = gx[e];
          }
        }
        {  // py_grad columns t < te  <=>  c >= 1
          const ptrdiff_t lo = (ptrdiff_t)s * T + bd.te - c0 - 3;
          if (c0 >= 1 && c0 + 3 < Tn) {
            f4 o; o[0] = gy[3]; o[1] = gy[2]; o[2] = gy[1]; o[3] = gy[0];
            *reinterpret_cast<f4u*>(pyg + lo) = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c0 + e >= 1 && c0 + e < Tn) pyg[lo + 3 - e] = gy[e];
          }
        }
      }
    }
  };

  // ---- interior ("fast") chunks, see the forward kernel.  Clamped rows read some valid row's G: their
  // flow is exactly zero (nothing flows past row s_begin: G[s_begin, t] == 0), so garbage G cannot matter.
  int offG[4], offPX[4], offPY[4];
  bool rvalid[4], xvalid[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int row = 16 * m + frow;
    const int r = row0 + row;
    const int cq = 4 * fq - SKEW * row;
    const int rc = min(r, Sn - 1);
    // element e of the quad is column c0+e (reversed): memory holds it at t = te - c0 - e; lowest address
    // of the quad is e = 3
    offG[m] = (bd.se - rc) * (T + 1) + bd.te - cq - 3;
    offPX[m] = (bd.se - r) * T1 + bd.te - cq - 3;
    offPY[m] = (bd.se - r) * T + bd.te - cq - 3;
    rvalid[m] = r < Sn;
    xvalid[m] = r >= 1 && r < Sn;
  }
  auto load_fast = [&](int k, f4 (&gq)[4]) {
    const float* ws_k = wsb - CH * k;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f4 t4 = *reinterpret_cast<const f4u*>(ws_k + offG[m]);
      gq[m][0] = t4[3]; gq[m][1] = t4[2]; gq[m][2] = t4[1]; gq[m][3] = t4[0];
    }
  };
  auto store_fast = [&](int k, const f4 (&oq)[4]) {
    float* px_k = pxg - CH * k;
    float* py_k = pyg - CH * k;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f4 gy = tO[fq * PLANE + 16 * m + frow];
      if (xvalid[m]) {
        f4 o; o[0] = oq[m][3]; o[1] = oq[m][2]; o[2] = oq[m][1]; o[3] = oq[m][0];
        *reinterpret_cast<f4u*>(px_k + offPX[m]) = o;
      }
      if (rvalid[m]) {
        f4 o; o[0] = gy[3]; o[1] = gy[2]; o[2] = gy[1]; o[3] = gy[0];
        *reinterpret_cast<f4u*>(py_k + offPY[m]) = o;
      }
    }
  };

  auto slot_general = [&](int k, f4 (&gq)[4]) {
    const bool active = (k >= 0 && k < nchunks);
    f4 oq[4];
    if (active) {
      if (!MOD && k == 0) ecarry = rings[w * RINGN + 63];
      compute_chunk(k);
      fetch_out(oq);  // px_grad quads out of the G tile before it is refilled
    }
    if (k + 1 >= 0 && k + 1 < nchunks) write_tile(gq);
    if (active) store_out(k, oq);
    if (k + 1 + NPF >= 0 && k + 1 + NPF < nchunks) load_chunk(k + 1 + NPF, gq);
    __syncthreads();
  };
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  auto slot_fast = [&](int k, f4 (&gq)[4]) {
    f4 oq[4];
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    FTR_STAMP(t0);
    compute_chunk(k);
    FTR_STAMP(t1);
    fetch_out(oq);
    write_tile(gq);
    FTR_STAMP(t2);
    store_fast(k, oq);
    FTR_STAMP(t3);
    load_fast(k + 1 + NPF, gq);
    FTR_STAMP(t4);
    __syncthreads();
    FTR_STAMP(t5);
    st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += t3 - t2; st_acc[3] += t4 - t3; st_acc[4] += t5 - t4; st_acc[5] += 1;
  };

  // slot schedule: identical to the forward kernel's (columns are reversed, the geometry is the same)
  const int K0 = MOD ? 1 : 4;
  const int K1 = (Tn >= CH) ? (Tn - CH) / CH + 1 : 0;
  const int KF0 = K0, KF1 = K1 - 1 - NPF;
  const int NIT = (Gtot + NPF + 1 + NPF - 1) / NPF;
  const int base = -(NPF + 1) - STG * w;
  int it1 = (KF0 - base + NPF - 1) / NPF;
  int it2 = (KF1 - NPF + 1 - base + NPF - 1) / NPF;
  it1 = min(max(it1, 0), NIT);
  it2 = min(max(it2, it1), NIT);

  int it = 0;
  for (; it < it1; ++it) {
#pragma unroll
    for (int u = 0; u < NPF; ++u) slot_general(base + NPF * it + u, rg[u]);
  }
  if (it < it2) {
    __builtin_amdgcn_s_waitcnt(kVmcnt0);
    for (; it < it2; ++it) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) slot_fast(base + NPF * it + u, rg[u]);
    }
  }
  for (; it < NIT; ++it) {
#pragma unroll
    for (int u = 0; u < NPF; ++u) slot_general(base + NPF * it + u, rg[u]);
  }
#ifdef FTR_STAMPS
  if (b == 0 && threadIdx.x == 0)
    for (int i = 0; i < 6; ++i) g_stamps[8 + i] = st_acc[i];
#endif
}

inline size_t wave_lds_bytes(int NW) {
  return (size_t)NW * 2 * TILE_F4 * sizeof(f4) + (size_t)(NW + 1) * RINGN * sizeof(float);
}

template <typename K>
int prepare_lds(K kernel, size_t lds, const char* what) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("%s: cannot reserve %zu bytes of LDS: %s", what, lds, hipGetErrorString(e));
      return FTR_ERR_LAUNCH;
    }
  }
  return FTR_OK;
}

template <bool MOD, int MAXW>
int launch_fwd(const float* px, const float* py, const int32_t* boundary, float* ws, float* ans, int B,
               int S, int T, int NW, hipStream_t st) {
  const size_t lds = wave_lds_bytes(NW);
  int rc = prepare_lds(mi_wave_fwd_kernel<MOD, MAXW>, lds, "mi_wave_fwd");
  if (rc != FTR_OK) return rc;
  hipLaunchKernelGGL((mi_wave_fwd_kernel<MOD, MAXW>), dim3(B), dim3(64 * NW), lds, st, px, py, boundary, ws, ans, S, T);
  return check_launch("mi_wave_fwd");
}
template <bool MOD, int MAXW>
int launch_bwd(const int32_t* boundary, const float* ws, float* px_grad, float* py_grad, float* ans_grad,
               int overwrite, int B, int S, int T, int NW, hipStream_t st) {
  const size_t lds = wave_lds_bytes(NW);
  int rc = prepare_lds(mi_wave_bwd_kernel<MOD, MAXW>, lds, "mi_wave_bwd");
  if (rc != FTR_OK) return rc;
  hipLaunchKernelGGL((mi_wave_bwd_kernel<MOD, MAXW>), dim3(B), dim3(64 * NW), lds, st, boundary, ws, px_grad, py_grad, ans_grad, overwrite, S, T);
  return check_launch("mi_wave_bwd");
}

}  // namespace

int mi_mono_fwd(const float* px, const float* py, const int32_t* boundary, float* ws, float* ans,
                int B, int S, int T, int modified, hipStream_t st) {
  const int NW = (S + 1 + 63) / 64;
  if (NW > 16) {
    set_error("mi_mono_fwd: S+1=%d rows exceed the 1024 rows one workgroup covers (round-1 limit)", S + 1);
    return FTR_ERR_UNSUPPORTED;
  }
#define FTR_DISPATCH(MODV)                                                                      \
  (NW <= 4 ? launch_fwd<MODV, 4>(px, py, boundary, ws, ans, B, S, T, NW, st)                   \
           : NW <= 8 ? launch_fwd<MODV, 8>(px, py, boundary, ws, ans, B, S, T, NW, st)         \
                     : launch_fwd<MODV, 16>(px, py, boundary, ws, ans, B, S, T, NW, st))
  return modified ? FTR_DISPATCH(true) : FTR_DISPATCH(false);
#undef FTR_DISPATCH
}

int mi_mono_bwd(const int32_t* boundary, const float* ws, float* px_grad, float* py_grad,
                float* ans_grad, int overwrite, int B, int S, int T, int modified, hipStream_t st) {
  const int NW = (S + 1 + 63) / 64;
  if (NW > 16) {
    set_error("mi_mono_bwd: S+1=%d rows exceed the 1024 rows one workgroup covers (round-1 limit)", S + 1);
    return FTR_ERR_UNSUPPORTED;
  }
#define FTR_DISPATCH(MODV)                                                                                              \
  (NW <= 4 ? launch_bwd<MODV, 4>(boundary, ws, px_grad, py_grad, ans_grad, overwrite, B, S, T, NW, st)                 \
           : NW <= 8 ? launch_bwd<MODV, 8>(boundary, ws, px_grad, py_grad, ans_grad, overwrite, B, S, T, NW, st)       \
                     : launch_bwd<MODV, 16>(boundary, ws, px_grad, py_grad, ans_grad, overwrite, B, S, T, NW, st))
  return modified ? FTR_DISPATCH(true) : FTR_DISPATCH(false);
#undef FTR_DISPATCH
}

}  // namespace ftr
