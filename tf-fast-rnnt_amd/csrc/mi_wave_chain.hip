// csrc/mi_wave_chain.hip -- "wavefront" mutual-information kernels for gfx950 (MI355X), the product path:
// one workgroup per 64-row BAND of one utterance, bands chained through global memory.
//
// Why bands are spread over CUs: a CU's load path saturates at ~35 GB/s for this access shape (64-byte row
// segments at 4-byte alignment; profiles/r01_d_membench_access_shapes.log) and one band at recursion speed needs
// ~12 KiB per 16-step chunk.  With all bands of an utterance on one CU (mi_wave_duo.hip) the IO waves were the
// bottleneck and only B of the 256 CUs were busy; here B * ceil((S+1)/64) CUs work and each sees one band.
//
// A workgroup = 3 waves with fixed roles, one __syncthreads() per 16-step chunk ("slot"):
//   COMPUTE wave  the recursion only: lane l <-> row s, time-skewed walk (lane l on column j - l), predecessors
//                 from its own previous value and the neighbouring lane (DPP wave_shr:1); log2 domain, bare
//                 v_exp_f32 / v_log_f32; -1e30 stands for -inf.  See mi_wave_duo.hip / DESIGN.md section 4.
//   IO wave       px/py tiles: 16-byte global loads three chunks ahead -> registers -> LDS tile [quad][row]
//                 (double buffered); drains the compute wave's per-cell output, turns it into
//                 G = sigmoid(a - b) and stores it (forward) / stores px_grad, py_grad (backward).
//   COMM wave     the boundary row between bands.  The band above publishes lane 63's value of every step as
//                 an 8-byte granule {tag = chunk + 1, value} with a relaxed agent-scope atomic store (sc1, no
//                 fence, fire and forget); this band's COMM wave polls the 16 granules of the chunk it needs
//                 next with relaxed agent-scope atomic loads (sc1: L1 bypassed) until every tag matches, then
//                 copies the values into the LDS ring the compute wave reads.  Granules are zeroed by a memset
//                 node in front of every launch; tags are never 0.  (Recipe R2 of the CDNA guide, Guideline 16.)
// No workgroup ever waits for a LATER band, block ids are band-major (all bands 0 first), and every poll is
// bounded: if a poll times out the band poisons its ring with NaN, stops polling, and the NaN reaches ans /
// the gradients -- the launch always terminates.
//
// Workspace ("p" in the C ABI): [ G lattice: B*(S+1)*(T+1) floats | granules: B * NB * Tg * 8 bytes ].
#include "ftr_common.h"
#include "mi_wave_common.h"

namespace ftr {
using namespace wavecfg;
namespace {

constexpr int NPFC = 3;          // chunks in flight in the IO wave's registers
constexpr int kMaxSpin = 400000; // polls of ~1 us before a band gives up (never reached unless a producer died)
typedef unsigned long long u64;

// granules per (utterance, band): one per producer step, chunk aligned, one spare chunk
__host__ __device__ inline int granules_per_band(int T, int modified) {
  const int nchunks = (T + 1 + (modified ? 0 : 63) + CH - 1) / CH;
  return CH * (nchunks + 1);
}

// COMM wave helpers --------------------------------------------------------------------------------------
// publish chunk m: 16 values from the LDS out-ring -> granules of the band below
__device__ __forceinline__ void comm_publish(const float* out_ring, u64* gran_out, int m, int lane) {
  if (lane < CH) {
    const float v = out_ring[(CH * m + lane) & (RINGN - 1)];
    const u64 g = ((u64)(unsigned)(m + 1) << 32) | (u64)__float_as_uint(v);
    __hip_atomic_store(gran_out + CH * m + lane, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// import chunk m of the band above into the LDS in-ring.  `g` holds the granule this lane loaded for chunk m one
// slot earlier (the ~1.5 us sc1 round trip is hidden behind a whole slot); only if a tag does not match yet does
// the wave fall into the polling loop.  Returns false after a timeout.
__device__ __forceinline__ u64 comm_peek(const u64* gran_in, int m, int lane) {
  return __hip_atomic_load(gran_in + CH * m + (lane & (CH - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool comm_import(float* in_ring, const u64* gran_in, int m, int lane, u64 g) {
  const int idx = CH * m + (lane & (CH - 1));
  for (int spins = 0;; ++spins) {
    const bool ok = (unsigned)(g >> 32) == (unsigned)(m + 1);
    if (__all(ok)) break;                 // wave-uniform exit
    if (spins >= kMaxSpin) return false;  // wave-uniform (spins is uniform)
    __builtin_amdgcn_s_sleep(24);
    g = __hip_atomic_load(gran_in + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (lane < CH) in_ring[idx & (RINGN - 1)] = __uint_as_float((unsigned)g);
  return true;
}

// --------------------------------------------------------------------------------------------- forward
template <bool MOD>
__global__ __launch_bounds__(192) void mi_chain_fwd_kernel(
    const float* __restrict__ px, const float* __restrict__ py, const int32_t* __restrict__ boundary,
    float* __restrict__ ws, u64* __restrict__ gran, float* __restrict__ ans, int B, int NB, int Tg, int S, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SKEW = MOD ? 0 : 1;
  constexpr int NPF = NPFC;
  constexpr int LOOK = MOD ? 1 : 5;  // at slot kc the COMM wave imports the upper band's chunk kc + LOOK
  constexpr int PRE = NPF + 1;  // IO pipeline warm-up slots in front of chunk 0
  const int b = blockIdx.x % B;                  // band-major block ids: producers are dispatched first
  const int w = blockIdx.x / B;                  // band of 64 rows
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0 compute, 1 IO, 2 COMM
  const Bound bd = load_boundary(boundary, b, S, T);
  const int T1 = MOD ? T : T + 1;
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  if (Sn <= 0 || Tn <= 0) { if (w == 0 && threadIdx.x == 0) ans[b] = 0.0f; return; }
  const int NWact = (Sn + 63) >> 6;
  if (w >= NWact) return;   // bands past the utterance's last row: nobody waits for them

  // LDS tiles of this band, in f4 units from `lds`: X at [0,2), Y at [2,4), OUT at [4,6) tile slots; the
  // double-buffer half is picked with integer offsets so every access stays an LDS (ds_*) instruction.
  f4* lds = reinterpret_cast<f4*>(smem);
  const int tb0 = 0;
#define FTR_TX(k) (lds + tb0 + ((k) & 1) * TILE_F4)
#define FTR_TY(k) (lds + tb0 + (2 + ((k) & 1)) * TILE_F4)
#define FTR_TD(k) (lds + tb0 + (4 + ((k) & 1)) * TILE_F4)
  float* in_ring = reinterpret_cast<float*>(lds + 6 * TILE_F4);   // values of the band above (row row0-1)
  float* out_ring = in_ring + RINGN;                              // this band's lane 63
  for (int i = threadIdx.x; i < 2 * RINGN; i += blockDim.x) in_ring[i] = kNeg;
  __syncthreads();

  const int nchunks = (Tn + 63 * SKEW + CH - 1) / CH;
  // slot gg: the compute chunk is kc = gg - PRE; the IO wave drains kc-1, parks kc+1, loads kc+1+NPF; the COMM
  // wave publishes kc-1 and imports the upper band's chunk kc+LOOK.  Last slot: kc = nchunks (drain + publish).
  const int nslots = nchunks + PRE + 1;
  const int NIT = (nslots + NPF - 1) / NPF;
  const int base = -PRE;  // kc = base + gg

  if (wid == 0) {
    // ======================================================================= COMPUTE wave
    const f4* ring_in = reinterpret_cast<const f4*>(in_ring);
    f4* ring_out = reinterpret_cast<f4*>(out_ring);
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
        if (!MOD && kc == 0) ecarry = in_ring[63];
        compute_chunk(kc);
      }
      FTR_STAMP(t1);
      __syncthreads();
      FTR_STAMP(t2);
      if (kc >= 8 && kc + 8 < nchunks) { st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += 1; }
    }
#ifdef FTR_STAMPS
    if (b == 0 && w == 1 && threadIdx.x == 0) { g_stamps[0] = st_acc[0]; g_stamps[1] = st_acc[1]; g_stamps[2] = st_acc[2]; }
#endif
    return;
  }

  if (wid == 2) {
    // ======================================================================= COMM wave
    u64* gran_out = gran + ((size_t)b * NB + (w + 1)) * Tg;        // read by band w+1
    const u64* gran_in = gran + ((size_t)b * NB + w) * Tg;         // written by band w-1
    const bool has_up = w > 0, has_down = w + 1 < NWact;
    bool dead = false;
    u64 g_cur = 0;   // granule of chunk (kc + LOOK), loaded during the previous slot (tag 0 = not loaded)
    for (int gg = 0; gg < NIT * NPF; ++gg) {
      const int kc = base + gg;
      if (has_down && kc - 1 >= 0 && kc - 1 < nchunks) comm_publish(out_ring, gran_out, kc - 1, lane);
      const int m = kc + LOOK;
      u64 g_next = 0;
      if (has_up && !dead && m + 1 >= 0 && m + 1 < nchunks) g_next = comm_peek(gran_in, m + 1, lane);
      if (has_up && !dead && m >= 0 && m < nchunks) {
        if (!comm_import(in_ring, gran_in, m, lane, g_cur)) {   // producer never showed up: poison, stop polling
          dead = true;
          in_ring[lane] = __builtin_nanf("");
        }
      }
      g_cur = g_next;
      __syncthreads();
    }
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
  if (b == 0 && w == 1 && lane == 0)
    for (int i = 0; i < 5; ++i) g_stamps[3 + i] = st_acc[i];
#endif
}

#undef FTR_TX
#undef FTR_TY
#undef FTR_TD

// --------------------------------------------------------------------------------------------- backward
// Reversed coordinates: row index r = s_end - s (lane), column c = t_end - t.
template <bool MOD>
__global__ __launch_bounds__(192) void mi_chain_bwd_kernel(
    const int32_t* __restrict__ boundary, const float* __restrict__ ws, u64* __restrict__ gran,
    float* __restrict__ px_grad, float* __restrict__ py_grad, float* __restrict__ ans_grad, int overwrite, int B,
    int NB, int Tg, int S, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SKEW = MOD ? 0 : 1;
  constexpr int NOFF = MOD ? 1 : 0;
  constexpr int NPF = NPFC;
  constexpr int LOOK = MOD ? 1 : 5;
  constexpr int PRE = NPF + 1;
  const int b = blockIdx.x % B;
  const int w = blockIdx.x / B;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0 compute, 1 IO, 2 COMM
  const Bound bd = load_boundary(boundary, b, S, T);
  const int T1 = MOD ? T : T + 1;
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  float* pxg = px_grad + (size_t)b * S * T1;
  float* pyg = py_grad + (size_t)b * (S + 1) * T;

  // ---- zeros outside the boundary rectangle (the reference memsets everything first,
  //      tf_fast_rnnt_op.cc:93-96); the rectangle itself is fully written by the sweep below.
  {
    const bool empty = (Sn <= 0 || Tn <= 0);
    const int nwv = 3 * NB;                 // every band's three waves share the fill of this utterance
    const int fwid = 3 * w + wid;
    // px_grad is defined on rows [sb, se) x columns [tb, te - NOFF]
    const int xr0 = empty ? 0 : bd.sb, xr1 = empty ? 0 : bd.se;
    const int xc0 = bd.tb, xc1 = bd.te - NOFF + 1;
    for (int s = fwid; s < S; s += nwv) {
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
    for (int s = fwid; s < S + 1; s += nwv) {
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
  const int NWact = (Sn + 63) >> 6;
  if (w >= NWact) return;

  f4* lds = reinterpret_cast<f4*>(smem);
  const int tb0 = 0;
#define FTR_TG(k) (lds + tb0 + ((k) & 1) * TILE_F4)
#define FTR_TPX(k) (lds + tb0 + (2 + ((k) & 1)) * TILE_F4)
#define FTR_TPY(k) (lds + tb0 + (4 + ((k) & 1)) * TILE_F4)
  float* in_ring = reinterpret_cast<float*>(lds + 6 * TILE_F4);
  float* out_ring = in_ring + RINGN;
  for (int i = threadIdx.x; i < 2 * RINGN; i += blockDim.x) in_ring[i] = 0.0f;
  __syncthreads();

  const int nchunks = (Tn + 63 * SKEW + CH - 1) / CH;
  const int nslots = nchunks + PRE + 1;
  const int NIT = (nslots + NPF - 1) / NPF;
  const int base = -PRE;

  if (wid == 0) {
    // ======================================================================= COMPUTE wave
    const f4* ring_in = reinterpret_cast<const f4*>(in_ring);
    f4* ring_out = reinterpret_cast<f4*>(out_ring);
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
        if (!MOD && kc == 0) ecarry = in_ring[63];
        compute_chunk(kc);
      }
      FTR_STAMP(t1);
      __syncthreads();
      FTR_STAMP(t2);
      if (kc >= 8 && kc + 8 < nchunks) { st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += 1; }
    }
#ifdef FTR_STAMPS
    if (b == 0 && w == 1 && threadIdx.x == 0) { g_stamps[8] = st_acc[0]; g_stamps[9] = st_acc[1]; g_stamps[10] = st_acc[2]; }
#endif
    return;
  }

  if (wid == 2) {
    // ======================================================================= COMM wave
    u64* gran_out = gran + ((size_t)b * NB + (w + 1)) * Tg;
    const u64* gran_in = gran + ((size_t)b * NB + w) * Tg;
    const bool has_up = w > 0, has_down = w + 1 < NWact;
    bool dead = false;
    u64 g_cur = 0;   // granule of chunk (kc + LOOK), loaded during the previous slot (tag 0 = not loaded)
    for (int gg = 0; gg < NIT * NPF; ++gg) {
      const int kc = base + gg;
      if (has_down && kc - 1 >= 0 && kc - 1 < nchunks) comm_publish(out_ring, gran_out, kc - 1, lane);
      const int m = kc + LOOK;
      u64 g_next = 0;
      if (has_up && !dead && m + 1 >= 0 && m + 1 < nchunks) g_next = comm_peek(gran_in, m + 1, lane);
      if (has_up && !dead && m >= 0 && m < nchunks) {
        if (!comm_import(in_ring, gran_in, m, lane, g_cur)) {
          dead = true;
          in_ring[lane] = __builtin_nanf("");
        }
      }
      g_cur = g_next;
      __syncthreads();
    }
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
  if (b == 0 && w == 1 && lane == 0)
    for (int i = 0; i < 5; ++i) g_stamps[11 + i] = st_acc[i];
#endif
}

#undef FTR_TG
#undef FTR_TPX
#undef FTR_TPY

inline size_t chain_lds_bytes() { return (size_t)6 * TILE_F4 * sizeof(f4) + 2 * RINGN * sizeof(float); }

}  // namespace

// floats of workspace behind the G lattice (the lattice part is rounded up to an even count so that the
// granules are 8-byte aligned); sized for the regular variant, which needs more.
size_t mi_chain_extra_floats(int B, int S, int T) {
  const int NB = (S + 1 + 63) / 64;
  return 1 + 2 * (size_t)B * NB * granules_per_band(T, 0);
}

static u64* granule_base(float* ws, int B, int S, int T) {
  size_t L = (size_t)B * (S + 1) * (T + 1);
  L += L & 1;
  return reinterpret_cast<u64*>(ws + L);
}

int mi_chain_fwd(const float* px, const float* py, const int32_t* boundary, float* ws, float* ans, int B, int S,
                 int T, int modified, hipStream_t st) {
  const int NB = (S + 1 + 63) / 64;
  const int Tg = granules_per_band(T, modified);
  u64* gran = granule_base(ws, B, S, T);
  if ((reinterpret_cast<uintptr_t>(gran) & 7) != 0) { set_error("mi_chain_fwd: workspace must be 8-byte aligned"); return FTR_ERR_INVALID_ARG; }
  if (hipMemsetAsync(gran, 0, sizeof(u64) * (size_t)B * NB * Tg, st) != hipSuccess) { set_error("mi_chain_fwd: memset failed"); return FTR_ERR_LAUNCH; }
  const size_t lds = chain_lds_bytes();
  if (modified) hipLaunchKernelGGL(mi_chain_fwd_kernel<true>, dim3(B * NB), dim3(192), lds, st, px, py, boundary, ws, gran, ans, B, NB, Tg, S, T);
  else hipLaunchKernelGGL(mi_chain_fwd_kernel<false>, dim3(B * NB), dim3(192), lds, st, px, py, boundary, ws, gran, ans, B, NB, Tg, S, T);
  return check_launch("mi_chain_fwd");
}

int mi_chain_bwd(const int32_t* boundary, const float* ws, float* px_grad, float* py_grad, float* ans_grad,
                 int overwrite, int B, int S, int T, int modified, hipStream_t st) {
  const int NB = (S + 1 + 63) / 64;
  const int Tg = granules_per_band(T, modified);
  u64* gran = granule_base(const_cast<float*>(ws), B, S, T);   // the granule tail of the workspace is scratch
  if (hipMemsetAsync(gran, 0, sizeof(u64) * (size_t)B * NB * Tg, st) != hipSuccess) { set_error("mi_chain_bwd: memset failed"); return FTR_ERR_LAUNCH; }
  const size_t lds = chain_lds_bytes();
  if (modified) hipLaunchKernelGGL(mi_chain_bwd_kernel<true>, dim3(B * NB), dim3(192), lds, st, boundary, ws, gran, px_grad, py_grad, ans_grad, overwrite, B, NB, Tg, S, T);
  else hipLaunchKernelGGL(mi_chain_bwd_kernel<false>, dim3(B * NB), dim3(192), lds, st, boundary, ws, gran, px_grad, py_grad, ans_grad, overwrite, B, NB, Tg, S, T);
  return check_launch("mi_chain_bwd");
}

}  // namespace ftr
