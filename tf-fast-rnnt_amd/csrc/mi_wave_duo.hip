// csrc/mi_wave_duo.hip -- "wavefront" mutual-information kernels for gfx950 (MI355X), the product path:
// specialised compute / IO wave pairs.
//
// What they compute is the recursion of the reference (tf_fast_rnnt/csrc/mutual_information.h:101-126,
// mutual_information_cuda.cu:174-422 forward, :441-760 backward); HOW is different by design:
//
//  * one workgroup per utterance; every band of 64 lattice rows is served by TWO waves: a COMPUTE wave that
//    runs nothing but the recursion, and an IO wave that does every byte of data movement.  Measured on
//    MI355X the recursion is bound by the issue rate and dependent latency of ONE wave (a lone wave issues
//    one VALU instruction per 4 cycles, transcendental 8), not by memory: with staging, stores and address
//    arithmetic in the same wave a step cost 195 cycles, 124 of them the chain (profiles/r01_b_*).  The
//    two waves of a pair share a SIMD, so the IO wave issues in the slots the compute wave leaves empty.
//  * COMPUTE wave: lane l <-> row s.  It walks the lattice in time-skewed order: at local step j lane l sits
//    on column c = j - l (regular) or c = j (modified), so both predecessors of a cell were produced one
//    step earlier: the lane's own previous value (p[s,t-1]) and the neighbouring lane's previous value
//    (p[s-1,t] / p[s-1,t-1]), fetched with one full-wave DPP shift (wave_shr:1).  A step is
//    mov_dpp, add, add, sub, max, v_exp_f32, add, v_log_f32, add (+ one bit-field insert that packs what the
//    IO wave needs): no barrier, no global memory, LDS only once per 4 steps.  (The reference runs this part
//    on 32 lanes of one warp per 32x32 tile and relaunches the kernel once per tile diagonal.)
//  * values are kept in the log2 domain (the IO wave multiplies inputs by log2(e) while staging them) so the
//    hardware v_exp_f32 / v_log_f32 are used bare; -inf is represented by -1e30 inside the kernel so no
//    NaN guard sits on the chain (LogAdd's "diff - diff != 0" branch, mutual_information.h:79-80), and is
//    turned back into -inf on the way out.
//  * IO wave: fetches px/py with coalesced 16-byte loads (4 lanes per 64-byte row segment, already skewed)
//    three 16-step chunks ahead into registers, parks them in an LDS tile laid out [quad][row] (plane stride
//    66 x 16 B: fill and per-lane ds_read_b128 both bank-conflict free), double buffered against the compute
//    wave; and drains the compute wave's per-cell output, turns it into G and stores it (16-byte stores).
//    Interior chunks use a branch-free steady-state loop so the compiler emits counted vmcnt waits.
//  * the forward does not store p.  It stores, per cell, G = sigmoid(a - b): the share of the cell's
//    probability that arrived through the px edge.  That is exactly term1 of the incoming edge in the
//    reference's backward (mutual_information_cuda.cu:455-457) and 1 - G is term2.  The backward then needs
//    no exp at all: it pushes occupancy "flow" down the lattice, pg = xin + yin, xout = pg * G,
//    yout = pg - xout; px_grad = xin, py_grad = yin (eqs. 3a-3c of the reference, .cu:474-477).  One lattice
//    is written by the forward and one is read by the backward (the reference writes p and p_grad and reads
//    px, py, p again), and flow is conserved to rounding.
//  * neighbouring compute waves exchange their boundary row through a 64-entry LDS ring per wave pair; all
//    waves run the same chunk schedule staggered by 5 chunks (regular) or 1 (modified) with one
//    __syncthreads() per chunk, which is what makes the ring race free (RING).
//
// Workspace ("p" in the C ABI): B*(S+1)*(T+1) floats holding G for every in-boundary cell.
// LDS per band: 6 tiles of 4224 B (forward: X, Y, OUT each double buffered; backward: G, PX, PY): up to 6
// bands (S+1 <= 384) fit the 160 KB of a CU; larger lattices use mi_wave_mono.hip.
#include "ftr_common.h"
#include "mi_wave_common.h"

namespace ftr {
using namespace wavecfg;
namespace {

constexpr int NPFD = 3;  // chunks in flight in the IO wave's registers

// --------------------------------------------------------------------------------------------- forward
template <bool MOD, int MAXB>
__global__ __launch_bounds__(128 * MAXB) void mi_duo_fwd_kernel(
    const float* __restrict__ px, const float* __restrict__ py, const int32_t* __restrict__ boundary,
    float* __restrict__ ws, float* __restrict__ ans, int S, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SKEW = MOD ? 0 : 1;
  constexpr int STG = MOD ? 1 : 5;
  constexpr int NPF = NPFD;
  constexpr int PRE = NPF + 1;  // IO pipeline warm-up slots in front of chunk 0
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NW = blockDim.x >> 7;               // bands; waves [0,NW) compute, [NW,2NW) IO
  const bool io = wid >= NW;
  const int w = io ? wid - NW : wid;
  const Bound bd = load_boundary(boundary, b, S, T);
  const int T1 = MOD ? T : T + 1;
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  if (Sn <= 0 || Tn <= 0) { if (threadIdx.x == 0) ans[b] = 0.0f; return; }

  // LDS tiles of this band, in f4 units from `lds`: X at [0,2), Y at [2,4), OUT at [4,6) tile slots; the
  // double-buffer half is picked with integer offsets so every access stays an LDS (ds_*) instruction.
  f4* lds = reinterpret_cast<f4*>(smem);
  const int tb0 = w * 6 * TILE_F4;
#define FTR_TX(k) (lds + tb0 + ((k) & 1) * TILE_F4)
#define FTR_TY(k) (lds + tb0 + (2 + ((k) & 1)) * TILE_F4)
#define FTR_TD(k) (lds + tb0 + (4 + ((k) & 1)) * TILE_F4)
  float* rings = reinterpret_cast<float*>(lds + NW * 6 * TILE_F4);
  for (int i = threadIdx.x; i < (NW + 1) * RINGN; i += blockDim.x) rings[i] = kNeg;
  __syncthreads();

  const int nchunks = (Tn + 63 * SKEW + CH - 1) / CH;
  const int NWact = (Sn + 63) >> 6;
  // slot gg: this band's compute chunk is kc = gg - PRE - STG*w; the IO wave drains kc-1, parks kc+1, loads
  // kc+1+NPF.  Last needed slot: drain of chunk nchunks-1 of the last active band.
  const int nslots = nchunks + PRE + STG * (NWact - 1) + 1;
  const int NIT = (nslots + NPF - 1) / NPF;
  const int base = -PRE - STG * w;  // kc = base + gg

  if (!io) {
    // ======================================================================= COMPUTE wave
    const f4* ring_in = reinterpret_cast<const f4*>(rings + w * RINGN);
    f4* ring_out = reinterpret_cast<f4*>(rings + (w + 1) * RINGN);
    const int wfin = (Sn - 1) >> 6, lfin = (Sn - 1) & 63;
    const int jfin = (w == wfin) ? (Tn - 1 + SKEW * lfin) : -1000;  // where ans appears
    float pcur = (w == 0 && lane == 0) ? 0.0f : kNeg;  // origin trick: p[sb,tb] = 0 + (Y := 0)
    float ecarry = kNeg;
    // lane 0 has no left neighbour: its "up" value comes from the ring, folded into X off the chain
    const float lane0 = (lane == 0) ? 1.0f : 0.0f;

    auto compute_chunk = [&](int k) {
      const f4* cX = FTR_TX(k);
      const f4* cY = FTR_TY(k);
      f4* cD = FTR_TD(k);
      f4 Xn = cX[lane], Yn = cY[lane];
      f4 En = ring_in[((CH * k) & (RINGN - 1)) >> 2];  // same address in every lane (broadcast)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int j0 = CH * k + 4 * q;
        const f4 X4 = Xn, Y4 = Yn, E4 = En;
        if (q + 1 < NQ) {  // next quad's operands are fetched while this quad's chain runs
          Xn = cX[(q + 1) * PLANE + lane];
          Yn = cY[(q + 1) * PLANE + lane];
          En = ring_in[((j0 + 4) & (RINGN - 1)) >> 2];
        }
        f4 XE;
        XE[0] = __builtin_fmaf(lane0, ecarry, X4[0]); XE[1] = __builtin_fmaf(lane0, E4[0], X4[1]);
        XE[2] = __builtin_fmaf(lane0, E4[1], X4[2]);  XE[3] = __builtin_fmaf(lane0, E4[2], X4[3]);
        f4 V4, P4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, pcur), 0x138, 0xf, 0xf, true));
          const float a = up + XE[e];
          const float c = pcur + Y4[e];
          const float d = a - c;
          const float mx = fmaxf(a, c);
          const float ex = __builtin_amdgcn_exp2f(-__builtin_fabsf(d));
          pcur = mx + __builtin_amdgcn_logf(1.0f + ex);
          V4[e] = __builtin_copysignf(ex, d);  // exp2(-|d|) with the sign of d: all the IO wave needs for G
          P4[e] = pcur;
        }
        ecarry = E4[3];
        cD[q * PLANE + lane] = V4;
        if (lane == 63) ring_out[(j0 & (RINGN - 1)) >> 2] = P4;
        if ((jfin >> 2) == (j0 >> 2)) {  // wave-uniform
          const int e = jfin & 3;
          const float v = (e == 0) ? P4[0] : (e == 1) ? P4[1] : (e == 2) ? P4[2] : P4[3];
          if (lane == lfin) ans[b] = (v <= kNegThresh) ? -INFINITY : v * kLn2;
        }
      }
    };

    unsigned long long st_acc[3] = {0, 0, 0};
    for (int gg = 0; gg < NIT * NPF; ++gg) {
      const int kc = base + gg;
      unsigned long long t0 = 0, t1 = 0, t2 = 0;
      FTR_STAMP(t0);
      if (kc >= 0 && kc < nchunks) {
        if (!MOD && kc == 0) ecarry = rings[w * RINGN + 63];
        compute_chunk(kc);
      }
      FTR_STAMP(t1);
      __syncthreads();
      FTR_STAMP(t2);
      if (kc >= 8 && kc + 8 < nchunks) { st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += 1; }
    }
#ifdef FTR_STAMPS
    if (b == 0 && threadIdx.x == 0) { g_stamps[0] = st_acc[0]; g_stamps[1] = st_acc[1]; g_stamps[2] = st_acc[2]; }
#endif
    return;
  }

  // ========================================================================= IO wave
  const float* pxb = px + (size_t)b * S * T1;
  const float* pyb = py + (size_t)b * (S + 1) * T;
  float* wsb = ws + (size_t)b * (S + 1) * (T + 1);
  const int row0 = 64 * w;
  // staging geometry of this lane: in load/store instruction m it handles tile row 16m + (lane>>2), quad (lane&3)
  const int frow = lane >> 2, fq = lane & 3;
  f4 rx[NPF][4], ry[NPF][4];

  auto load_general = [&](int k, f4 (&x)[4], f4 (&y)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 16 * m + frow;
      const int r = row0 + row;
      const int c0 = CH * k + 4 * fq - SKEW * row;  // column (relative to tb) of the quad's first step
      f4 vx = {kNeg, kNeg, kNeg, kNeg}, vy = {kNeg, kNeg, kNeg, kNeg};
      if (r < Sn) {
        if (r >= 1) {  // px[s-1][t + toff], toff = -1 for modified
          const int cx = MOD ? c0 - 1 : c0;
          const ptrdiff_t o = (ptrdiff_t)(bd.sb + r - 1) * T1 + bd.tb + cx;
          if (cx >= 0 && c0 + 3 < Tn) {
            vx = *reinterpret_cast<const f4u*>(pxb + o);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (cx + e >= 0 && c0 + e < Tn) vx[e] = pxb[o + e];
          }
        }
        {  // py[s][t-1]
          const ptrdiff_t o = (ptrdiff_t)(bd.sb + r) * T + bd.tb + c0 - 1;
          if (c0 >= 1 && c0 + 3 < Tn) {
            vy = *reinterpret_cast<const f4u*>(pyb + o);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c0 + e >= 1 && c0 + e < Tn) vy[e] = pyb[o + e];
          }
        }
      }
      x[m] = vx;
      y[m] = vy;
    }
  };
  // kk = chunk being parked.  The origin cell (row s_begin, column t_begin: chunk 0, tile row 0, quad 0,
  // element 0 of band 0) gets Y := 0 so that p = logadd(-inf, pcur(0) + 0) = 0 falls out of the recursion.
  auto park = [&](int kk, const f4 (&x)[4], const f4 (&y)[4]) {
    f4* dX = FTR_TX(kk);
    f4* dY = FTR_TY(kk);
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
      dX[fq * PLANE + row] = xs;
      dY[fq * PLANE + row] = ys;
    }
  };
  // G = sigmoid(d) from v = copysign(exp2(-|d|), d):  d >= 0 -> 1/(1+e),  d < 0 -> e/(1+e)
  auto to_G = [&](const f4& v) {
    f4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ve = v[e];  // scalar copy first: bit_cast applied to a vector-element lvalue reads element 0
      const float ea = __builtin_fabsf(ve);
      const float rc = __builtin_amdgcn_rcpf(1.0f + ea);
      g[e] = (__float_as_int(ve) < 0) ? ea * rc : rc;  // sign BIT: -0.0 (e underflowed) is "d < 0"
    }
    return g;
  };
  auto drain_general = [&](int k) {
    const f4* sD = FTR_TD(k);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 16 * m + frow;
      const int r = row0 + row;
      if (r < Sn) {
        const int c0 = CH * k + 4 * fq - SKEW * row;
        const f4 g = to_G(sD[fq * PLANE + row]);
        const ptrdiff_t o = (ptrdiff_t)(bd.sb + r) * (T + 1) + bd.tb + c0;
        if (c0 >= 0 && c0 + 3 < Tn) {
          *reinterpret_cast<f4u*>(wsb + o) = g;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e >= 0 && c0 + e < Tn) wsb[o + e] = g[e];
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
  auto drain_fast = [&](int k) {
    const f4* sD = FTR_TD(k);
    float* ws_k = wsb + CH * k;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f4 g = to_G(sD[fq * PLANE + 16 * m + frow]);
      if (rvalid[m]) *reinterpret_cast<f4u*>(ws_k + offG[m]) = g;
    }
  };

  auto slot_general = [&](int kc, f4 (&x)[4], f4 (&y)[4]) {
    if (kc + 1 >= 0 && kc + 1 < nchunks) park(kc + 1, x, y);
    if (kc - 1 >= 0 && kc - 1 < nchunks) drain_general(kc - 1);
    if (kc + 1 + NPF >= 0 && kc + 1 + NPF < nchunks) load_general(kc + 1 + NPF, x, y);
    __syncthreads();
  };
  unsigned long long st_acc[5] = {0, 0, 0, 0, 0};
  auto slot_fast = [&](int kc, f4 (&x)[4], f4 (&y)[4]) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    FTR_STAMP(t0);
    park(kc + 1, x, y);         // loads of chunk kc+1 were issued NPF slots ago
    FTR_STAMP(t1);
    drain_fast(kc - 1);
    FTR_STAMP(t2);
    load_fast(kc + 1 + NPF, x, y);
    FTR_STAMP(t3);
    __syncthreads();
    FTR_STAMP(t4);
    st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += t3 - t2; st_acc[3] += t4 - t3; st_acc[4] += 1;
  };

  // Fast slot kc: the drained chunk kc-1 and the loaded chunk kc+1+NPF are interior.
  const int K0 = MOD ? 1 : 4;                            // 16k - 63*SKEW >= 1
  const int K1 = (Tn >= CH) ? (Tn - CH) / CH + 1 : 0;    // 16k + 15 < Tn
  const int KF0 = K0 + 1, KF1 = (Sn >= 2) ? K1 - 1 - NPF : 0;   // fast slots: KF0 <= kc < KF1
  int it1 = (KF0 - base + NPF - 1) / NPF;                // first iteration whose first slot has kc >= KF0
  int it2 = (KF1 - base) / NPF;                          // first iteration whose last slot has kc >= KF1
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
  if (b == 0 && w == 0 && lane == 0)
    for (int i = 0; i < 5; ++i) g_stamps[3 + i] = st_acc[i];
#endif
}

#undef FTR_TX
#undef FTR_TY
#undef FTR_TD

// --------------------------------------------------------------------------------------------- backward
// Reversed coordinates: row index r = s_end - s (lane), column c = t_end - t.
template <bool MOD, int MAXB>
__global__ __launch_bounds__(128 * MAXB) void mi_duo_bwd_kernel(
    const int32_t* __restrict__ boundary, const float* __restrict__ ws, float* __restrict__ px_grad,
    float* __restrict__ py_grad, float* __restrict__ ans_grad, int overwrite, int S, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SKEW = MOD ? 0 : 1;
  constexpr int STG = MOD ? 1 : 5;
  constexpr int NOFF = MOD ? 1 : 0;
  constexpr int NPF = NPFD;
  constexpr int PRE = NPF + 1;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NW = blockDim.x >> 7;
  const bool io = wid >= NW;
  const int w = io ? wid - NW : wid;
  const Bound bd = load_boundary(boundary, b, S, T);
  const int T1 = MOD ? T : T + 1;
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  float* pxg = px_grad + (size_t)b * S * T1;
  float* pyg = py_grad + (size_t)b * (S + 1) * T;

  // ---- zeros outside the boundary rectangle (the reference memsets everything first,
  //      tf_fast_rnnt_op.cc:93-96); the rectangle itself is fully written by the sweep below.
  {
    const bool empty = (Sn <= 0 || Tn <= 0);
    const int nwv = 2 * NW;
    // px_grad is defined on rows [sb, se) x columns [tb, te - NOFF]
    const int xr0 = empty ? 0 : bd.sb, xr1 = empty ? 0 : bd.se;
    const int xc0 = bd.tb, xc1 = bd.te - NOFF + 1;
    for (int s = wid; s < S; s += nwv) {
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
    for (int s = wid; s < S + 1; s += nwv) {
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

  f4* lds = reinterpret_cast<f4*>(smem);
  const int tb0 = w * 6 * TILE_F4;
#define FTR_TG(k) (lds + tb0 + ((k) & 1) * TILE_F4)
#define FTR_TPX(k) (lds + tb0 + (2 + ((k) & 1)) * TILE_F4)
#define FTR_TPY(k) (lds + tb0 + (4 + ((k) & 1)) * TILE_F4)
  float* rings = reinterpret_cast<float*>(lds + NW * 6 * TILE_F4);
  for (int i = threadIdx.x; i < (NW + 1) * RINGN; i += blockDim.x) rings[i] = 0.0f;
  __syncthreads();

  const int nchunks = (Tn + 63 * SKEW + CH - 1) / CH;
  const int NWact = (Sn + 63) >> 6;
  const int nslots = nchunks + PRE + STG * (NWact - 1) + 1;
  const int NIT = (nslots + NPF - 1) / NPF;
  const int base = -PRE - STG * w;

  if (!io) {
    // ======================================================================= COMPUTE wave
    const f4* ring_in = reinterpret_cast<const f4*>(rings + w * RINGN);
    f4* ring_out = reinterpret_cast<f4*>(rings + (w + 1) * RINGN);
    const int wfin = (Sn - 1) >> 6, lfin = (Sn - 1) & 63;
    const int jfin = (w == wfin) ? (Tn - 1 + SKEW * lfin) : -1000;  // where p_grad[sb,tb] appears
    float yprev = (w == 0 && lane == 0) ? ans_grad[b] : 0.0f;  // seeds p_grad[se,te] = ans_grad
    float xprev = 0.0f;
    float ecarry = 0.0f;

    auto compute_chunk = [&](int k) {
      const f4* cG = FTR_TG(k);
      f4* cPX = FTR_TPX(k);
      f4* cPY = FTR_TPY(k);
      f4 Gn = cG[lane];
      f4 En = ring_in[((CH * k) & (RINGN - 1)) >> 2];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int j0 = CH * k + 4 * q;
        const f4 G4 = Gn, E4 = En;
        if (q + 1 < NQ) {
          Gn = cG[(q + 1) * PLANE + lane];
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
        cPX[q * PLANE + lane] = PX4;
        cPY[q * PLANE + lane] = PY4;
        if (lane == 63) ring_out[(j0 & (RINGN - 1)) >> 2] = XO4;
        if (overwrite && (jfin >> 2) == (j0 >> 2)) {
          const int e = jfin & 3;
          const float v = (e == 0) ? PG4[0] : (e == 1) ? PG4[1] : (e == 2) ? PG4[2] : PG4[3];
          if (lane == lfin) ans_grad[b] = v;
        }
      }
    };

    unsigned long long st_acc[3] = {0, 0, 0};
    for (int gg = 0; gg < NIT * NPF; ++gg) {
      const int kc = base + gg;
      unsigned long long t0 = 0, t1 = 0, t2 = 0;
      FTR_STAMP(t0);
      if (kc >= 0 && kc < nchunks) {
        if (!MOD && kc == 0) ecarry = rings[w * RINGN + 63];
        compute_chunk(kc);
      }
      FTR_STAMP(t1);
      __syncthreads();
      FTR_STAMP(t2);
      if (kc >= 8 && kc + 8 < nchunks) { st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += 1; }
    }
#ifdef FTR_STAMPS
    if (b == 0 && threadIdx.x == 0) { g_stamps[8] = st_acc[0]; g_stamps[9] = st_acc[1]; g_stamps[10] = st_acc[2]; }
#endif
    return;
  }

  // ========================================================================= IO wave
  const float* wsb = ws + (size_t)b * (S + 1) * (T + 1);
  const int row0 = 64 * w;
  const int frow = lane >> 2, fq = lane & 3;
  f4 rg[NPF][4];

  auto load_general = [&](int k, f4 (&gq)[4]) {
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
  auto park = [&](int kk, const f4 (&gq)[4]) {
    f4* dG = FTR_TG(kk);
#pragma unroll
    for (int m = 0; m < 4; ++m) dG[fq * PLANE + 16 * m + frow] = gq[m];
  };
  auto drain_general = [&](int k) {
    const f4* sX = FTR_TPX(k);
    const f4* sY = FTR_TPY(k);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 16 * m + frow;
      const int r = row0 + row;
      if (r < Sn) {
        const int c0 = CH * k + 4 * fq - SKEW * row;
        const int s = bd.se - r;
        const f4 gx = sX[fq * PLANE + row];
        const f4 gy = sY[fq * PLANE + row];
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
  auto drain_fast = [&](int k) {
    const f4* sX = FTR_TPX(k);
    const f4* sY = FTR_TPY(k);
    float* px_k = pxg - CH * k;
    float* py_k = pyg - CH * k;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f4 gx = sX[fq * PLANE + 16 * m + frow];
      const f4 gy = sY[fq * PLANE + 16 * m + frow];
      if (xvalid[m]) {
        f4 o; o[0] = gx[3]; o[1] = gx[2]; o[2] = gx[1]; o[3] = gx[0];
        *reinterpret_cast<f4u*>(px_k + offPX[m]) = o;
      }
      if (rvalid[m]) {
        f4 o; o[0] = gy[3]; o[1] = gy[2]; o[2] = gy[1]; o[3] = gy[0];
        *reinterpret_cast<f4u*>(py_k + offPY[m]) = o;
      }
    }
  };

  auto slot_general = [&](int kc, f4 (&gq)[4]) {
    if (kc + 1 >= 0 && kc + 1 < nchunks) park(kc + 1, gq);
    if (kc - 1 >= 0 && kc - 1 < nchunks) drain_general(kc - 1);
    if (kc + 1 + NPF >= 0 && kc + 1 + NPF < nchunks) load_general(kc + 1 + NPF, gq);
    __syncthreads();
  };
  unsigned long long st_acc[5] = {0, 0, 0, 0, 0};
  auto slot_fast = [&](int kc, f4 (&gq)[4]) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    FTR_STAMP(t0);
    park(kc + 1, gq);
    FTR_STAMP(t1);
    drain_fast(kc - 1);
    FTR_STAMP(t2);
    load_fast(kc + 1 + NPF, gq);
    FTR_STAMP(t3);
    __syncthreads();
    FTR_STAMP(t4);
    st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += t3 - t2; st_acc[3] += t4 - t3; st_acc[4] += 1;
  };

  const int K0 = MOD ? 1 : 4;
  const int K1 = (Tn >= CH) ? (Tn - CH) / CH + 1 : 0;
  const int KF0 = K0 + 1, KF1 = K1 - 1 - NPF;
  int it1 = (KF0 - base + NPF - 1) / NPF;
  int it2 = (KF1 - base) / NPF;
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
  if (b == 0 && w == 0 && lane == 0)
    for (int i = 0; i < 5; ++i) g_stamps[11 + i] = st_acc[i];
#endif
}

#undef FTR_TG
#undef FTR_TPX
#undef FTR_TPY

inline size_t duo_lds_bytes(int NW) {
  return (size_t)NW * 6 * TILE_F4 * sizeof(f4) + (size_t)(NW + 1) * RINGN * sizeof(float);
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

template <bool MOD, int MAXB>
int launch_fwd(const float* px, const float* py, const int32_t* boundary, float* ws, float* ans, int B,
               int S, int T, int NW, hipStream_t st) {
  const size_t lds = duo_lds_bytes(NW);
  int rc = prepare_lds(mi_duo_fwd_kernel<MOD, MAXB>, lds, "mi_wave_fwd");
  if (rc != FTR_OK) return rc;
  hipLaunchKernelGGL((mi_duo_fwd_kernel<MOD, MAXB>), dim3(B), dim3(128 * NW), lds, st, px, py, boundary, ws, ans, S, T);
  return check_launch("mi_wave_fwd");
}
template <bool MOD, int MAXB>
int launch_bwd(const int32_t* boundary, const float* ws, float* px_grad, float* py_grad, float* ans_grad,
               int overwrite, int B, int S, int T, int NW, hipStream_t st) {
  const size_t lds = duo_lds_bytes(NW);
  int rc = prepare_lds(mi_duo_bwd_kernel<MOD, MAXB>, lds, "mi_wave_bwd");
  if (rc != FTR_OK) return rc;
  hipLaunchKernelGGL((mi_duo_bwd_kernel<MOD, MAXB>), dim3(B), dim3(128 * NW), lds, st, boundary, ws, px_grad, py_grad, ans_grad, overwrite, S, T);
  return check_launch("mi_wave_bwd");
}

constexpr int kMaxDuoBands = 6;

// ---------------------------------------------------------------------------------------------
// Hardware self-test: the wavefront kernels rely on (1) wave_shr:1 DPP shifting across all 64 lanes
// with lane 0 keeping `old`, (2) 16-byte global loads/stores at 4-byte alignment.  result[0] = 1 if
// both behave as assumed.
__global__ void selftest_kernel(const float* __restrict__ in, float* __restrict__ out, int* __restrict__ result) {
  const int lane = threadIdx.x;
  const float mine = (float)(lane + 1);
  const float got = dpp_wave_shr1(-7.0f, mine);
  const bool ok1 = (lane == 0) ? (got == -7.0f) : (got == (float)lane);
  const float got0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine), 0x138, 0xf, 0xf, true));
  const bool ok3 = (lane == 0) ? (got0 == 0.0f) : (got0 == (float)lane);   // bound_ctrl form: lane 0 reads 0
  // unaligned 16B load at element offset 1 + 5*lane, store at 3 + 5*lane
  const f4 v = *reinterpret_cast<const f4u*>(in + 1 + 5 * lane);
  bool ok2 = true;
  for (int e = 0; e < 4; ++e) ok2 = ok2 && (v[e] == (float)(1 + 5 * lane + e));
  *reinterpret_cast<f4u*>(out + 3 + 5 * lane) = v;
  const unsigned long long m = __ballot(ok1 && ok2 && ok3);
  if (lane == 0) result[0] = (m == ~0ull) ? 1 : 0;
}

}  // namespace

// implemented in mi_wave_mono.hip
int mi_mono_fwd(const float* px, const float* py, const int32_t* boundary, float* ws, float* ans, int B, int S, int T, int modified, hipStream_t st);
int mi_mono_bwd(const int32_t* boundary, const float* ws, float* px_grad, float* py_grad, float* ans_grad, int overwrite, int B, int S, int T, int modified, hipStream_t st);

// force_mono: diagnostic selection of the single-role kernels for any size
int mi_wave_fwd(const float* px, const float* py, const int32_t* boundary, float* ws, float* ans,
                int B, int S, int T, int modified, int force_mono, hipStream_t st) {
  const int NW = (S + 1 + 63) / 64;
  if (force_mono || NW > kMaxDuoBands) return mi_mono_fwd(px, py, boundary, ws, ans, B, S, T, modified, st);
#define FTR_DISPATCH(MODV)                                                                     \
  (NW <= 4 ? launch_fwd<MODV, 4>(px, py, boundary, ws, ans, B, S, T, NW, st)                  \
           : launch_fwd<MODV, 6>(px, py, boundary, ws, ans, B, S, T, NW, st))
  return modified ? FTR_DISPATCH(true) : FTR_DISPATCH(false);
#undef FTR_DISPATCH
}

int mi_wave_bwd(const int32_t* boundary, const float* ws, float* px_grad, float* py_grad,
                float* ans_grad, int overwrite, int B, int S, int T, int modified, int force_mono,
                hipStream_t st) {
  const int NW = (S + 1 + 63) / 64;
  if (force_mono || NW > kMaxDuoBands) return mi_mono_bwd(boundary, ws, px_grad, py_grad, ans_grad, overwrite, B, S, T, modified, st);
#define FTR_DISPATCH(MODV)                                                                                             \
  (NW <= 4 ? launch_bwd<MODV, 4>(boundary, ws, px_grad, py_grad, ans_grad, overwrite, B, S, T, NW, st)                \
           : launch_bwd<MODV, 6>(boundary, ws, px_grad, py_grad, ans_grad, overwrite, B, S, T, NW, st))
  return modified ? FTR_DISPATCH(true) : FTR_DISPATCH(false);
#undef FTR_DISPATCH
}

int selftest(hipStream_t st, int* result_dev) {
  // scratch lives behind result_dev: [0] result int, then 512 floats in, 512 floats out
  float* in = reinterpret_cast<float*>(result_dev + 4);
  float* out = in + 512;
  float host[512];
  for (int i = 0; i < 512; ++i) host[i] = (float)i;
  if (hipMemcpyAsync(in, host, sizeof(host), hipMemcpyHostToDevice, st) != hipSuccess) {
    set_error("selftest: memcpy failed"); return FTR_ERR_LAUNCH;
  }
  if (hipStreamSynchronize(st) != hipSuccess) {  // host[] is on the stack
    set_error("selftest: sync failed"); return FTR_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, in, out, result_dev);
  return check_launch("selftest");
}

}  // namespace ftr
