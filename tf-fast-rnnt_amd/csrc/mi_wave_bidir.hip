// csrc/mi_wave_bidir.hip -- bidirectional ("meet in the middle") wavefront mutual-information kernels for gfx950,
// the product path.  One workgroup per 64-row band, bands chained through 8-byte granules, time-skewed wavefront,
// log2 domain, a 4-wave workgroup (compute / IO-in / COMM / IO-out: one wave per
// SIMD of a CU), and the serial dependency chain -- the one thing that bounds this kernel -- is
// cut in half:
//
//   forward launch, 2 * B * NB workgroups
//     dir 0 "alpha":  p(s,t) from the origin (s_begin,t_begin) up to the CUT, the anti-diagonal
//                     (s - s_begin) + (t - t_begin) = jm (regular) / the column t - t_begin = jm (modified), jm = D / 2;
//     dir 1 "beta":   q(s,t) = log-prob of reaching (s_end,t_end) from (s,t), from the end cell back to the same cut.
//                     In reversed coordinates r = s_end - s, c = t_end - t this is the SAME recursion with
//                     X(r,c) = px[s,t], Y(r,c) = py[s,t]  (reference recursion: mutual_information_cuda.cu:149-239,
//                     mirrored), so the compute and COMM waves are shared and only the IO waves' addressing differs.
//     Both store, per cell, the split ratio G = sigmoid(a - b) of the two incoming terms (alpha: what fraction of
//     p(s,t) arrived through px; beta: what fraction of q(s,t) leaves through px) in their own lattice, and the
//     values on the cut in `pmid`.
//   mid kernel, B workgroups:  ans = logsumexp over the cut of p + q;  occ = exp(p + q - ans) = the occupancy of
//     every cut cell (every path crosses the cut exactly once).
//   flow launch, 2 * B * NB workgroups (the backward pass, cf. mutual_information_cuda.cu:452-715): the occupancy
//     is injected on the cut and propagated outwards with the stored ratios -- dir 0 from the cut back to the origin
//     (writes the transitions that end on or before the cut), dir 1 from the cut forward to the end cell (writes the
//     transitions that start on or after it).  No exp/log, mass is conserved exactly, px_grad / py_grad are the
//     per-transition flows.
// Each chain is (S+T)/2 steps long instead of S+T, for the forward and for the backward pass.
//
// Walk coordinates (r, c): lane = r - 64 * band, step j = c + SKEW * r.  "FWD addressing" = walk coordinates are
// (s - s_begin, t - t_begin); "REV addressing" = (s_end - s, t_end - t).  Forward alpha and flow beta use FWD,
// forward beta and flow alpha use REV.
//
// Workspace ("p" in the C ABI), floats:  [ G_alpha lattice | G_beta lattice | pmid 2*B*(S+1) | occ B*(S+1) | seed B | shift 2*B | cut frames 2*B*NB | pad |
//                                          ctrl: pad, done[B], uflags[B] | pad |
//                                          granules of the forward launch 2*B*NB*Tg*8 bytes | granules of the flow launch | status (4) ].
// "ctrl + granules" is the HAND-OFF region: it must be all zero when a launch starts.  The launches leave it all zero
// again (every consumer clears the granules it has imported, the last workgroup of an utterance clears its counters),
// so a caller that initialised the workspace once (ftr_mutual_information_workspace_init) passes FTR_MI_WS_CLEAN and no
// memset node is needed; without the flag the launchers zero the region themselves.
//   status   (the last block, at the same address for every shape launched on one buffer) sticky: bit 0 = a band gave up waiting for its producer (the results of that launch are poisoned)
//   done[b]  bands of utterance b that have delivered their cut values (the last one runs the cut reduction)
//   uflags[b] bit 1 = a NaN was read from px / py inside the boundary rectangle of utterance b  ->  ans[b] = NaN
#include "ftr_common.h"
#include "mi_wave_common.h"
#include <type_traits>

namespace ftr {
using namespace wavecfg;

// (declared in ftr_common.h: the library's replacement for memset / memcpy nodes)
__global__ void zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    uint4* p4 = reinterpret_cast<uint4*>(p);
    const size_t n4 = n >> 2;
    for (size_t i = i0; i < n4; i += stride) p4[i] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = 4 * n4 + i0; i < n; i += stride) p[i] = 0u;
  } else {
    for (size_t i = i0; i < n; i += stride) p[i] = 0u;
  }
}
__global__ void copy_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

namespace {

// Prefetch depth of the IO-in wave (chunks in flight in registers) and the back-off of a COMM wave whose granules are
// not there yet.  Measured (profiles/r01_k_mi_prefetch_poll.log): the forward's two input streams need 4 slots of
// lead at the ~0.75 us slots the 4-wave workgroup reaches (3 left the IO-in wave waiting on memory, 76.8 -> 62.0 us
// at c3); the flow pass has one stream and is not helped; a short poll back-off (128 instead of 1536 cycles) trims the
// detection delay of every band hop.
#ifndef FTR_NPF_FWD
#define FTR_NPF_FWD 4
#endif
#ifndef FTR_NPF_FLOW
#define FTR_NPF_FLOW 4
#endif
#ifndef FTR_POLL_SLEEP
#define FTR_POLL_SLEEP 2
#endif
#ifndef FTR_MAX_SPIN
#define FTR_MAX_SPIN 400000
#endif
constexpr int kMaxSpin = FTR_MAX_SPIN; // polls of ~1 us before a band gives up (never reached unless a producer died)
typedef unsigned long long u64;

// Diagnostic build (make STAMPS=1 [STAMP_BAND=n]): per-wave busy / barrier-wait ticks of the forward kernel's slots,
// utterance 0, alpha direction, band FTR_STAMP_BAND; read back with ftr_debug_stamps():
// g_stamps[4*wid + {0,1,2}] = {busy, wait, slots} for wid 0 compute, 1 IO-in, 2 COMM, 3 IO-out.
#ifndef FTR_STAMP_BAND
#define FTR_STAMP_BAND 0
#endif
#ifdef FTR_STAMPS
#define FTR_SYNC_DECL unsigned long long st_last = 0, st_busy = 0, st_wait = 0, st_n = 0; FTR_STAMP(st_last)
#define FTR_SYNC() do { unsigned long long a_, b_; FTR_STAMP(a_); __syncthreads(); FTR_STAMP(b_); st_busy += a_ - st_last; st_wait += b_ - a_; st_last = b_; ++st_n; } while (0)
#define FTR_SYNC_REPORT(slot) do { if (!REVM && b == 0 && w == FTR_STAMP_BAND && lane == 0) { g_stamps[4 * (slot)] = st_busy; g_stamps[4 * (slot) + 1] = st_wait; g_stamps[4 * (slot) + 2] = st_n; } } while (0)
#elif defined(FTR_TRACE) && FTR_TRACE == 3
// Diagnostic build: when does each wave of the traced forward band ARRIVE at the barrier of every slot?  g_trace[256 * wid +
// 16 + n] for the waves' n-th barrier (n < 240), next to the start / end words of the FTR_TRACE == 1 layout
// (scripts/mi_trace_waves.py).
#define FTR_SYNC_DECL int tr_n = 0
#define FTR_SYNC() do { if (b == 0 && REVM == (FTR_TRACE_DIR != 0) && w == FTR_STAMP_BAND && lane == 0 && tr_n < 240) g_trace[256 * wid + 16 + tr_n] = trace_now(); ++tr_n; __syncthreads(); } while (0)
#define FTR_SYNC_REPORT(slot) do { if (b == 0 && REVM == (FTR_TRACE_DIR != 0) && w == FTR_STAMP_BAND && lane == 0 && (slot) == 1) g_trace[4] = tr_n; } while (0)
#else
#define FTR_SYNC_DECL do { } while (0)
#define FTR_SYNC() __syncthreads()
#define FTR_SYNC_REPORT(slot) do { } while (0)
#endif

__host__ __device__ inline int granules_per_band(int T, int modified) {
  const int nchunks = (T + 1 + (modified ? 0 : 63) + CH - 1) / CH;
  return CH * (nchunks + 1);
}
// floats in front of the first ratio lattice: the flow kernel's unguarded 16-byte loads may start up to 66 elements before
// a lattice row (skewed quads of edge chunks; masked after the load) -- for the first row of the first utterance that is
// in front of the lattice, and has to stay inside the workspace
constexpr size_t kLatPad = 128;
__host__ __device__ inline size_t lattice_floats(int B, int S, int T) {
  size_t L = (size_t)B * (S + 1) * (T + 1);
  return (L + 3) & ~(size_t)3;
}

// COMM wave helpers -----------------------------------------------------------------
// One granule to the band below: a relaxed agent-scope atomic store (write-through).  (A plain store, which stays in the
// XCD's L2 where an sc1 poll of a consumer on the same XCD finds it, was measured in round 3 -- all bands of a chain do share
// an XCD at B = 32 -- and is SLOWER: 58.6 / 49.5 us against 50.5 / 38.7 warm, 66.6 / 61.0 against 58.4 / 52.6 inside the step,
// profiles/r03_plainpub_experiment.log.  A plain store is visible to nobody until it has left the CU.)
__device__ __forceinline__ void publish_granule(u64* p, u64 g) {
  __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 comm_peek(const u64* gran_in, int m, int lane) {
  return __hip_atomic_load(gran_in + CH * m + (lane & (CH - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Granule = { tag, value }, tag = (chunk + 1) | (frame step of that chunk << 16): see "Frames" in the forward body (the
// flow kernel publishes a zero step).  16 bits of chunk number: T + 64 < 16 * 65535, checked by the launchers.
constexpr unsigned kTagChunkMask = 0xffffu;
constexpr int kStepMax = 32767;        // 16-bit signed frame step
// Waits (bounded) for the 16 granules of chunk m; `g` comes in as this lane's granule as peeked one slot ago and goes out
// as read with the right tag.  false: the producer never showed up (sticky status bit set).
__device__ __forceinline__ bool comm_wait(u64* gran_in, int m, int lane, u64& g, int* status) {
  const int idx = CH * m + (lane & (CH - 1));
  for (int spins = 0;; ++spins) {
    const bool ok = ((unsigned)(g >> 32) & kTagChunkMask) == (unsigned)(m + 1);
    if (__all(ok)) break;                 // wave-uniform exit
    if (spins >= kMaxSpin) {              // wave-uniform (spins is uniform): the producer never showed up
      if (lane == 0) __hip_atomic_fetch_or(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky
      return false;
    }
    __builtin_amdgcn_s_sleep(FTR_POLL_SLEEP);
    g = __hip_atomic_load(gran_in + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return true;
}
__device__ __forceinline__ void ring_put(float* in_ring, int m, int lane, float v) {
  if (lane < CH) in_ring[(CH * m + lane) & (RINGN - 1)] = v;
}
__device__ __forceinline__ bool comm_import(float* in_ring, u64* gran_in, int m, int lane, u64 g, int* status) {
  if (!comm_wait(gran_in, m, lane, g, status)) return false;
  ring_put(in_ring, m, lane, __uint_as_float((unsigned)g));
  return true;
}
// this band is the only reader of its granules: it leaves them zero for the next launch on this workspace
// (self-cleaning).  Issued AFTER the peek for the next chunk: in front of it, the next import's wait for that peek also
// waits for these write-through stores (vmcnt is in order) -- 0.3 us per slot on every band that has a band above.
__device__ __forceinline__ void comm_clear(u64* gran_in, int m, int lane) {
  if (lane < CH) __hip_atomic_store(gran_in + CH * m + lane, (u64)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// values on the cut go to the last workgroup of the utterance (possibly on another XCD): write-through stores, read
// back with agent-scope loads after the done[] counter says everybody has delivered
__device__ __forceinline__ void pmid_store(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f4 rev4(const f4 t) {
  f4 v;
  v[0] = t[3]; v[1] = t[2]; v[2] = t[1]; v[3] = t[0];
  return v;
}

// the cut: D = last walk step of the lattice, jm = D / 2 in alpha coordinates
struct Cut { int D, jm; };
template <bool MOD>
__device__ __forceinline__ Cut make_cut(int Sn, int Tn) {
  Cut c;
  c.D = (MOD ? 0 : (Sn - 1)) + (Tn - 1);
  c.jm = c.D >> 1;
  return c;
}

// input tiles: 4 buffers (steady state: chunk k+2 is being parked while chunk k+1 is read into registers and chunk k is
// computed from registers; a chained band parks chunks 0..3 before it computes chunk 0 -- see the IO-in wave); output
// tiles: 2 buffers (chunk k is written while chunk k-1 is drained).  (k) + 8
// keeps the index non-negative for the warm-up slots.
#define FTR_TX(k) (lds + (((k) + 8) & 3) * TILE_F4)
#define FTR_TY(k) (lds + (4 + (((k) + 8) & 3)) * TILE_F4)
#define FTR_TD(k) (lds + (8 + ((k) & 1)) * TILE_F4)
#define FTR_TP(k) (lds + (10 + ((k) & 1)) * TILE_F4)
constexpr int kFwdTiles = 12;
constexpr int kFlowThreads = 320;   // flow kernel: compute, IO-in, COMM (loads only), IO-out px + hand-off stores, IO-out py
constexpr int kAhead = 2;   // the IO-in wave parks chunk kc + kAhead during slot kc

// ------------------------------------------------------------------------------------------------- forward
// One direction of one band.  REVM selects the IO waves' addressing (see the header).
template <bool MOD, bool REVM>
__device__ __forceinline__ void bidir_fwd_body(unsigned char* smem, const float* __restrict__ px,
                                               const float* __restrict__ py, const Bound bd, float* __restrict__ wsb,
                                               u64* __restrict__ gran_b, float* __restrict__ pmid_b,
                                               int* __restrict__ status, int* __restrict__ uflag_b,
                                               float* __restrict__ cxy_b, int* __restrict__ phimid_b, int b, int w, int Tg,
                                               int S, int T, int jstop) {
  constexpr int SKEW = MOD ? 0 : 1;
  constexpr int NOFF = MOD ? 1 : 0;
  constexpr int NPF = FTR_NPF_FWD;
  // at slot kc the COMM wave imports the upper band's chunk kc + LOOK: the compute wave reads it from the ring at the
  // start of slot kc + 1 for its chunk kc + 1, which needs the upper chunk kc + 1 + 4 (regular: lane 0's neighbour is
  // 63 steps ahead, plus one carried element) / kc + 1 (modified)
  constexpr int LOOK = MOD ? 1 : 5;
  // IO pipeline warm-up slots in front of chunk 0: NPF + kAhead, and NPF more in a band that has a band above it (whose
  // first four chunks are parked early, see the IO-in wave)
  const bool chained = w > 0;
  const int PRE = NPF + kAhead + (chained ? NPF : 0);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0 compute, 1 IO-in, 2 COMM, 3 IO-out
  const int T1 = MOD ? T : T + 1;
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  const int NWact = (Sn + 63) >> 6;

  f4* lds = reinterpret_cast<f4*>(smem);
  float* in_ring = reinterpret_cast<float*>(lds + kFwdTiles * TILE_F4);   // values of the band above (row row0-1)
  float* shift_lds = in_ring + RINGN;   // the utterance's shift constants (cx2, cy2): COMM wave -> IO-in wave
  // Frames.  On top of the per-utterance operand shift (which removes the mean drift and the tilt of p over the lattice for
  // a model whose paths are diffuse) every band renormalises its own values as it goes: at the start of chunk k the compute
  // wave subtracts an integer step[k] from its 64 values (one subtraction per 16 steps on the chain), so that they stay
  // small whatever the model -- a sharp model's log-probabilities along its alignment are nothing like the sampled means,
  // and with the static shift alone `ans` came out of the cancellation of two numbers of several thousand.  A uniform shift
  // of a whole anti-diagonal changes no split ratio.  frame(k) = step[0] + ... + step[k]; a stored value + frame(k) = the
  // (statically shifted) log-probability.  step[k + 3] is set by the IO-out wave from the largest value at the end of chunk
  // k (the COMM wave reads the compute wave's output tile for it; it has slack in every slot): step[k+3] = rint(max + 2.5 drift) - step[k+1] - step[k+2], three chunks ahead so
  // that the COMM wave, which converts the values of the band above into this band's frame when it imports them (five
  // chunks before their use), already knows the frame they will be used in.  The steps travel to the band below in the tag
  // word of the granules, the frame of the cut values goes to the cut reduction (phimid).
  float* sring = shift_lds + 2;         // step[k & 7], float-valued integers in [-32767, 32767]; sring[8] = the band's base frame
  if (threadIdx.x < 9) sring[threadIdx.x] = 0.0f;
  for (int i = threadIdx.x; i < RINGN; i += blockDim.x) in_ring[i] = kNeg;
  __syncthreads();

  // Bands step in LOCAL walk steps (lane l of band w is on column j - SKEW * l at local step j): the cut, given in
  // global steps, is at local step jl.  A band whose rows all lie beyond the cut has nothing to compute.
  const int jl = jstop - SKEW * 64 * w;
  if (jl < 0) {
    if (wid == 0 && 64 * w + lane < Sn) pmid_store(pmid_b + 64 * w + lane, kNeg);
    if (wid == 0 && lane == 0) __hip_atomic_store(phimid_b + w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // A band never steps past its own last column (local step Tn - 1 + 63 * SKEW): if the cut lies beyond that, none
  // of the band's rows has a cell on the cut, but the band still feeds the bands below.
  const int nchunks_nat = (Tn + 63 * SKEW + CH - 1) / CH;
  if (jl >= CH * nchunks_nat && wid == 0 && 64 * w + lane < Sn) pmid_store(pmid_b + 64 * w + lane, kNeg);
  if (jl >= CH * nchunks_nat && wid == 0 && lane == 0) __hip_atomic_store(phimid_b + w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int klast = min(jl / CH + 1, nchunks_nat);   // chunks [0, klast): local steps 0 .. jl (and the rest of that chunk)
  const int klast_up = MOD ? klast : min((jl + 64) / CH + 1, nchunks_nat);   // what the band above computes (and publishes)
  const int nslots = klast + PRE + 1;
  const int NIT = (nslots + NPF - 1) / NPF;
  const int base = -PRE;  // kc = base + gg
  FTR_SYNC_DECL;

  if (wid == 0) {
    // ======================================================================= COMPUTE wave
    const f4* ring_in = reinterpret_cast<const f4*>(in_ring);
    float pcur = (w == 0 && lane == 0) ? 0.0f : kNeg;  // origin trick: p[origin] = 0 + (Y := 0)
    float ecarry = kNeg;
    const float lane0 = (lane == 0) ? 1.0f : 0.0f;

    // A lone wave issues one VALU instruction per 4 cycles (16 for v_exp/v_log) whether or not it is on the dependent
    // chain (scripts/micro/clockbench.hip: the bare chain costs 76 cycles per step at 2.4 GHz), so the step is kept
    // to the arithmetic alone: both per-cell outputs -- copysign(e, d) for the IO-out wave and p itself for the COMM
    // wave (hand-off to the band below, values on the cut) -- leave through LDS tiles with two unconditional
    // ds_write_b128 per four steps; no exec masking, no selects, no global stores in this loop.
    // The operands of a whole chunk live in registers: they are read from the tiles (12 ds_read_b128, no wait in
    // between) one slot ahead, while the previous chunk is computed, so that the dependent chain below never waits for
    // the LDS (measured: with the reads issued one quad ahead, behind that quad's tile writes in the in-order LDS queue,
    // every quad paid a full LDS round trip -- 1529 of the slot's 1729 cycles were this wave, profiles/r02_b_stamps_stage1.log).
    // (The band above's values -- four broadcast reads -- are read at the start of the slot that uses them: one slot
    // less of hand-off lag per band than reading them a slot ahead with the tiles.)
    struct Ops { f4 X[NQ], Y[NQ]; float step; };
    auto fetch = [&](int k, Ops& o) {
      const f4* cX = FTR_TX(k);
      const f4* cY = FTR_TY(k);
      o.step = sring[k & 7];   // this chunk's frame step (see "Frames"), written two slots ago, read one slot ahead like the operands
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        o.X[q] = cX[q * PLANE + lane];
        o.Y[q] = cY[q * PLANE + lane];
      }
    };
    auto compute_chunk = [&](int k, const Ops& o) {
      f4* cD = FTR_TD(k);
      f4* cP = FTR_TP(k);
      f4 E[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) E[q] = ring_in[((CH * k + 4 * q) & (RINGN - 1)) >> 2];  // same address in every lane (broadcast)
      pcur -= o.step;     // into this chunk's frame: the one operation per chunk that the renormalisation costs the chain
      ecarry -= o.step;   // (the carried element was imported for the previous chunk's frame)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f4 X4 = o.X[q], Y4 = o.Y[q], E4 = E[q];
        f4 XE;
        XE[0] = __builtin_fmaf(lane0, ecarry, X4[0]); XE[1] = __builtin_fmaf(lane0, E4[0], X4[1]);
        XE[2] = __builtin_fmaf(lane0, E4[1], X4[2]);  XE[3] = __builtin_fmaf(lane0, E4[2], X4[3]);
        const f4 DL = XE - Y4;   // d = (up - p) + (X - Y): the difference of the two lattice values first
        f4 V4, P4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pci = __builtin_bit_cast(int, pcur);
          const float up1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, pci, 0x138, 0xf, 0xf, true));
          const float up2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, pci, 0x138, 0xf, 0xf, true));
          const float d = (up1 - pcur) + DL[e];
          const float mx = fmaxf(up2 + XE[e], pcur + Y4[e]);
          const float ex = __builtin_amdgcn_exp2f(-__builtin_fabsf(d));
          pcur = mx + __builtin_amdgcn_logf(1.0f + ex);
          V4[e] = __builtin_copysignf(ex, d);  // exp2(-|d|) with the sign of d: all the IO-out wave needs for G
          P4[e] = pcur;
        }
        ecarry = E4[3];
        cD[q * PLANE + lane] = V4;
        cP[q * PLANE + lane] = P4;
      }
    };

    // slot kc: fetch chunk kc + 1 (parked during slot kc - 1), compute chunk kc from the registers fetched during
    // slot kc - 1.  Two register sets, the loop is unrolled by two so that they swap roles without copies.
    Ops oa, ob;
    auto slot = [&](int kc, const Ops& cur, Ops& nxt) {
      if (kc + 1 >= 0 && kc + 1 < klast) fetch(kc + 1, nxt);
      if (kc >= 0 && kc < klast) {
        // the element in front of chunk 0's first ring value: the band above's step 63 (regular; imported with its
        // chunk 3 by now), nothing (modified: the ring still holds its initial -inf there)
        if (kc == 0) ecarry = in_ring[RINGN - 1];
#ifndef FTR_EXP_NOCOMPUTE
        compute_chunk(kc, cur);
#endif
      }
      FTR_SYNC();
#if defined(FTR_TRACE) && FTR_TRACE == 1
      if (b == 0 && REVM == (FTR_TRACE_DIR != 0) && w == FTR_STAMP_BAND && lane == 0 && 16 + (kc - base) < kTraceN) { g_trace[16 + (kc - base)] = trace_now(); g_trace[4] = kc - base + 1; }
#endif
    };
    static_assert((NPF & 1) == 0, "the slot loop is unrolled by two");
    for (int gg = 0; gg < NIT * NPF; gg += 2) {
      slot(base + gg, oa, ob);
      slot(base + gg + 1, ob, oa);
    }
    FTR_SYNC_REPORT(0);
    return;
  }

  if (wid == 2) {
    // ======================================================================= COMM wave
    u64* gran_in = gran_b + (size_t)w * Tg;               // written by band w-1
    // This wave issues LOADS only (peeks and poll reloads).  vmcnt retires in order, so a wave that both stores and polls
    // waits, in front of every import, for its own write-through stores of the previous slot as well -- 0.2-0.3 us per
    // slot on every band with a band above, more when the memory system is busy (profiles/r02_j).  The stores of the
    // hand-off (publishing to the band below, clearing what was imported, the cut values) are the IO-out wave's.
    const bool has_up = w > 0;
    bool dead = false;
    u64 g_cur = 0;   // granule of chunk (kc + LOOK), loaded during the previous slot (tag 0 = not loaded)
    // The shift constants of this utterance (ftr_common.h, Shift): a fixed sample of 256 elements of px and of py inside
    // the rectangle (four 64-column row segments each: eight cache lines per array, not one per element; the frames below
    // take care of whatever drift a coarse mean leaves), the same positions in every band and direction, so every workgroup
    // of the utterance derives the same two numbers bit for bit.  This wave has nothing else to do in the warm-up slots: the
    // eight loads per lane go out now,
    // are reduced in slot kShiftSlot (the last warm-up slot: by then they have had three slots to arrive, and nothing queues
    // behind this wave's wait) and reach the IO-in wave through LDS one barrier later, at the top of the slot of its first park.
    constexpr int kShiftSlot = 3;
    constexpr int kSeg = 4;   // segments per array: lane l reads column col0 + l of row `row` -- two cache lines per segment
    float smp_x[kSeg], smp_y[kSeg];
    {
      const int nrx = Sn - 1, ncx = Tn - NOFF, nry = Sn, ncy = Tn - 1;
#pragma unroll
      for (int u = 0; u < kSeg; ++u) {
        const unsigned hr = (unsigned)(u + 1) * 0x9E3779B1u, hc = (unsigned)(u + 1) * 0x85EBCA77u + 0x1234567u;
        smp_x[u] = -INFINITY; smp_y[u] = -INFINITY;
        if (nrx > 0 && ncx > 0) {
          const int c0 = (int)__umulhi(hc, (unsigned)max(ncx - 63, 1));
          smp_x[u] = px[(ptrdiff_t)(bd.sb + (int)__umulhi(hr, (unsigned)nrx)) * T1 + bd.tb + min(c0 + lane, ncx - 1)];
        }
        if (ncy > 0) {
          const int c0 = (int)__umulhi(hc, (unsigned)max(ncy - 63, 1));
          smp_y[u] = py[(ptrdiff_t)(bd.sb + (int)__umulhi(hr, (unsigned)nry)) * T + bd.tb + min(c0 + lane, ncy - 1)];
        }
      }
    }
    int frame_rel = 0;   // (frame of the band above at chunk m - 1) - (this band's frame at chunk kc)
    int step_0 = 0, step_1 = 0, step_2 = 0;   // the steps scheduled for the next three chunks (see below)
    float mx_prev = -INFINITY;                // the previous chunk's maximum (in that chunk's frame)
    const bool lane_valid = 64 * w + lane < Sn;
    constexpr int kFirstUsed = MOD ? 0 : 3;   // first chunk of the band above whose values this band reads (regular: its step 63)
    for (int gg = 0; gg < NIT * NPF; ++gg) {
      const int kc = base + gg;
      const int m = kc + LOOK;
      if (gg == kShiftSlot) {
        float sx = 0.0f, nx = 0.0f, sy = 0.0f, ny = 0.0f;
#pragma unroll
        for (int u = 0; u < kSeg; ++u) {
          if (shift_sample_ok(smp_x[u])) { sx += smp_x[u]; nx += 1.0f; }
          if (shift_sample_ok(smp_y[u])) { sy += smp_y[u]; ny += 1.0f; }
        }
        Shift sh = shift_from_sums<MOD>(wave_sum_dpp(sx), wave_sum_dpp(nx), wave_sum_dpp(sy), wave_sum_dpp(ny), Sn, Tn);
#ifdef FTR_EXP_NOSTATIC   // study build: no operand shift
        sh.cx2 = 0.0f; sh.cy2 = 0.0f;
#endif
        if (lane == 0) {
          shift_lds[0] = sh.cx2; shift_lds[1] = sh.cy2;
          if (w == 0 && !REVM) { pmid_store(cxy_b, sh.cx2); pmid_store(cxy_b + 1, sh.cy2); }   // for the cut reduction
        }
      }
      // step[kc + 2] from the compute wave's values at the end of chunk kc - 1 (its output tile of the previous slot): where
      // the largest value will be three chunks from now if it keeps drifting as it did during that chunk (drift = the change
      // of the maximum plus the step that was taken out of it: absolute positions, so this is feed-forward, nothing
      // oscillates), centred over that chunk, less the two steps that are already under way.  Done here, BEFORE the poll:
      // this wave has slack in every slot (the IO-out wave, which did it first, has not), and the schedule must not wait
      // for the band above.
      {
        const int k = kc - 1;
        if (k >= 0 && k < klast) {
          const float plast = reinterpret_cast<const float*>(FTR_TP(k))[((3 * PLANE + lane) << 2) + 3];
          const float mx = wave_max_dpp((lane_valid && plast > kNegThresh) ? plast : -INFINITY);
          int step_3 = 0;
          if (mx > kNegThresh) {
            const float drift = (mx_prev > kNegThresh) ? (mx - mx_prev) + (float)step_0 : 0.0f;
            step_3 = min(max((int)__builtin_rintf(mx + 2.5f * drift) - step_1 - step_2, -kStepMax), kStepMax);
          }
#ifdef FTR_EXP_NOFRAMES   // study build: no renormalisation
          step_3 = 0;
#endif
          mx_prev = mx;
          if (lane == 0) sring[(k + 3) & 7] = (float)step_3;
          step_0 = step_1; step_1 = step_2; step_2 = step_3;      // now: step_0 = step[k + 1], step_1 = step[k + 2], step_2 = step[k + 3]
        }
      }
#ifdef FTR_EXP_NOPOLL
      if (false) {
#else
      if (has_up && !dead && m >= 0 && m < klast_up) {
#endif
        // the values are for local chunk kc + 1, whose step this wave set one slot ago (step[kc + 1] = step_1 after the shift
        // above; zero in the warm-up slots)
        if (kc + 1 >= 0 && kc + 1 < klast) frame_rel -= step_1;
        if (!comm_wait(gran_in, m, lane, g_cur, status)) {   // producer never showed up: poison, stop polling
          dead = true;
          in_ring[lane] = __builtin_nanf("");
        } else {
          // the values are log-probabilities in the producer's frame of its chunk m: the tag carries that chunk's step
          frame_rel += ((int)__builtin_amdgcn_readfirstlane((int)(g_cur >> 32))) >> 16;   // arithmetic shift: signed step
          float v = __uint_as_float((unsigned)g_cur);
          if (m == kFirstUsed) {
            // the first chunk of the band above that this band uses: this band's frame starts where those values are (by
            // now the band above may be thousands away from zero, and a frame that started at zero would spend the three
            // chunks of the schedule's lag at that magnitude).  The base frame is a virtual step in front of chunk 0.
            const float mx = wave_max_dpp((lane < CH && v > kNegThresh) ? v : -INFINITY);
            int f0 = 0;
            if (mx > kNegThresh) f0 = min(max(frame_rel + (int)__builtin_rintf(mx), -kStepMax), kStepMax);
#ifdef FTR_EXP_NOFRAMES
            f0 = 0;
#endif
            frame_rel -= f0;
            if (lane == 0) sring[8] = (float)f0;
          }
          v += (float)frame_rel;   // exact frame arithmetic (integers); -inf stays -1e30
          ring_put(in_ring, m, lane, v);
        }
      }
      // the next chunk's granules are requested now and looked at one whole slot later
      g_cur = 0;
      if (has_up && !dead && m + 1 >= 0 && m + 1 < klast_up) g_cur = comm_peek(gran_in, m + 1, lane);
      FTR_SYNC();
    }
    FTR_SYNC_REPORT(2);
    return;
  }

  // ========================================================================= IO waves
  // IO-in (wid 1): global loads three chunks ahead + parking the tiles.  IO-out (wid 3): turning the compute wave's
  // per-cell output into G and storing it.  Measured (profiles/r01_h): with one IO wave doing both, its slot (1879
  // ticks) and the compute wave's (2073) were balanced and neither could shrink alone.
  const int row0 = 64 * w;
  // staging geometry of this lane: in load/store instruction m it handles tile row 16m + (lane>>2), quad (lane&3)
  const int frow = lane >> 2, fq = lane & 3;
  f4 rx[NPF][4], ry[NPF][4];

  auto load_general = [&](int k, f4 (&x)[4], f4 (&y)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 16 * m + frow;
      const int r = row0 + row;
      const int c0 = CH * k + 4 * fq - SKEW * row;  // walk column of the quad's first step
      f4 vx = {kNeg, kNeg, kNeg, kNeg}, vy = {kNeg, kNeg, kNeg, kNeg};
      if (r < Sn) {
        if (!REVM) {
          if (r >= 1) {  // px[s-1][t + toff], toff = -1 for modified
            const int cx = MOD ? c0 - 1 : c0;
            const ptrdiff_t o = (ptrdiff_t)(bd.sb + r - 1) * T1 + bd.tb + cx;
            if (cx >= 0 && c0 + 3 < Tn) {
              vx = *reinterpret_cast<const f4u*>(px + o);
            } else if (cx + 3 >= 0 && c0 < Tn) {   // a quad that straddles the edge: rare, skipped when no lane has one
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (cx + e >= 0 && c0 + e < Tn) vx[e] = px[o + e];
            }
          }
          {  // py[s][t-1]
            const ptrdiff_t o = (ptrdiff_t)(bd.sb + r) * T + bd.tb + c0 - 1;
            if (c0 >= 1 && c0 + 3 < Tn) {
              vy = *reinterpret_cast<const f4u*>(py + o);
            } else if (c0 + 3 >= 1 && c0 < Tn) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (c0 + e >= 1 && c0 + e < Tn) vy[e] = py[o + e];
            }
          }
        } else {
          // element e is walk column c0+e, i.e. t = te - c0 - e: memory order is the reverse of e.  The registers keep
          // MEMORY order and park() swaps: reversing here would consume each load the moment it is issued (the
          // compiler then waits vmcnt(0) after every load -- no prefetch at all; seen in the ISA of the first version)
          if (r >= 1) {  // px[s][t], s = se - r <= se - 1; modified: t <= te - 1
            const ptrdiff_t lo = (ptrdiff_t)(bd.se - r) * T1 + bd.te - c0 - 3;
            if (c0 >= NOFF && c0 + 3 < Tn) {
              vx = *reinterpret_cast<const f4u*>(px + lo);      // memory order; park() reverses
            } else if (c0 + 3 >= NOFF && c0 < Tn) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (c0 + e >= NOFF && c0 + e < Tn) vx[3 - e] = px[lo + 3 - e];
            }
          }
          {  // py[s][t], t <= te - 1
            const ptrdiff_t lo = (ptrdiff_t)(bd.se - r) * T + bd.te - c0 - 3;
            if (c0 >= 1 && c0 + 3 < Tn) {
              vy = *reinterpret_cast<const f4u*>(py + lo);
            } else if (c0 + 3 >= 1 && c0 < Tn) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (c0 + e >= 1 && c0 + e < Tn) vy[3 - e] = py[lo + 3 - e];
            }
          }
        }
      }
      x[m] = vx;
      y[m] = vy;
    }
  };
  // kk = chunk being parked.  The origin cell (chunk 0, tile row 0, quad 0, element 0 of band 0) gets Y := 0 so that
  // p = logadd(-inf, pcur(0) + 0) = 0 falls out of the recursion.
  // NaN detection among the px / py values this lane stages (reported through uflags: ans = NaN): the running maximum of
  // the magnitudes' bit patterns exceeds that of infinity iff a NaN went by (two VALU instructions per value, no branches)
  unsigned nan_acc = 0;
  float cx2 = 0.0f, cy2 = 0.0f;   // the utterance's shift (log2 domain), read from LDS in slot 4 (IO-in wave)
  // v'[j] = v[j + d] (memory order): realigns a quad whose load address was clamped into the utterance's slab; the
  // elements that fall off are outside the slab, hence outside the boundary rectangle, and are masked by the caller
  auto shift4 = [](const f4 v, int d) {
    f4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = j + d;
      o[j] = (q == 0) ? v[0] : (q == 1) ? v[1] : (q == 2) ? v[2] : v[3];
    }
    return o;
  };
  const int xmaxo = S * T1 - 4, ymaxo = (S + 1) * T - 4;    // last element offset a 16-byte load may start at
  auto park = [&](int kk, const f4 (&xin)[4], const f4 (&yin)[4], auto edge_tag, int (&offXr)[4], int (&offYr)[4]) {
    constexpr bool edge = decltype(edge_tag)::value;   // compile time: the interior variant carries none of the masking
    f4* dX = FTR_TX(kk);
    f4* dY = FTR_TY(kk);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 16 * m + frow;
      f4 xm = xin[m], ym = yin[m];
      const int c0 = CH * kk + 4 * fq - SKEW * row;   // walk column of the quad's first step
      if (edge) {
        // a row group none of whose columns lies in the rectangle (start chunk k of the regular type: the groups m > k
        // have not reached column 0 yet -- three of the four in chunk 0): nothing to mask, scale or check
        const int cmax = CH * kk + 15 - SKEW * 16 * m, cmin = CH * kk - SKEW * (16 * m + 15);   // wave-uniform
        if (cmax < NOFF || cmin >= Tn) {
          const f4 nothing = {kNeg, kNeg, kNeg, kNeg};
          dX[fq * PLANE + row] = nothing;
          dY[fq * PLANE + row] = nothing;
          continue;
        }
      }
      if (edge) {   // the chunk has quads outside [1, Tn)
        const int dk = REVM ? -CH * kk : CH * kk;
        const int wx = offXr[m] + dk, wy = offYr[m] + dk;
        const int dx = wx - min(max(wx, 0), xmaxo), dy = wy - min(max(wy, 0), ymaxo);
        if (dx != 0) xm = shift4(xm, dx);
        if (dy != 0) ym = shift4(ym, dy);
      }
      f4 xs, ys;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float rx_ = xm[REVM ? 3 - e : e], ry_ = ym[REVM ? 3 - e : e];   // REV: registers are in memory order
        if (edge) {   // columns outside the rectangle: "impossible transition" (what is there is padding, not data)
          const int c = c0 + e;
          rx_ = (c >= NOFF && c < Tn) ? rx_ : -INFINITY;
          ry_ = (c >= 1 && c < Tn) ? ry_ : -INFINITY;
        }
        const float vx = __builtin_fmaf(rx_, kLog2e, -cx2), vy = __builtin_fmaf(ry_, kLog2e, -cy2);   // log2 domain, shifted (ftr_common.h, Shift)
        nan_acc = max(nan_acc, max(__float_as_uint(vx) & 0x7fffffffu, __float_as_uint(vy) & 0x7fffffffu));
        xs[e] = fmaxf(vx, kNeg);  // -inf -> kNeg (a NaN too: the chain stays finite, the utterance is flagged instead)
        ys[e] = fmaxf(vy, kNeg);
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
  // ---- interior ("fast") chunks: every quad of every lane-row lies inside [1, Tn) in columns, so loads and
  // stores are plain 16-byte accesses with no per-element guards and no divergent control flow.  Rows
  // beyond the utterance are clamped to a valid row: what they compute never reaches a valid row (data
  // only moves from row r-1 to row r) and is never stored.  Row 0's X is neutralised by the ring's -inf.
  int offX[4], offY[4], offG[4];
  bool rvalid[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int row = 16 * m + frow;
    const int r = row0 + row;
    const int cq = 4 * fq - SKEW * row;
    if (!REVM) {
      const int rxc = min(max(r - 1, 0), max(Sn - 2, 0));   // px row s-1 (clamped)
      const int ryc = min(r, Sn - 1);                       // py row s   (clamped)
      offX[m] = (bd.sb + rxc) * T1 + bd.tb + cq + (MOD ? -1 : 0);
      offY[m] = (bd.sb + ryc) * T + bd.tb + cq - 1;
      offG[m] = (bd.sb + r) * (T + 1) + bd.tb + cq;
    } else {
      const int rxc = min(max(r, 1), max(Sn - 1, 1));       // px row s = se - r, r in [1, Sn-1] (clamped)
      const int ryc = min(r, Sn - 1);
      offX[m] = (bd.se - rxc) * T1 + bd.te - cq - 3;
      offY[m] = (bd.se - ryc) * T + bd.te - cq - 3;
      offG[m] = (bd.se - r) * (T + 1) + bd.te - cq - 3;
    }
    rvalid[m] = r < Sn;
  }
  // The same eight 16-byte loads for EVERY chunk, edge chunks included: the start of each load is clamped into the
  // utterance's own slab of px / py (a quad may hang over the valid columns by up to 3 elements, which is inside the
  // slab except at its two ends); park() realigns the clamped quads and masks the columns outside the rectangle.  A
  // fixed instruction sequence keeps the waits of the prefetch pipeline counted (s_waitcnt vmcnt(n > 0)) in every
  // slot: with per-element guarded loads in the edge chunks each of those slots waited for vmcnt(0), a full memory
  // round trip, and the bands below inherited the stall (profiles/r02_c_kernel_timelines_stage2.log).
  auto load_fast = [&](int k, f4 (&x)[4], f4 (&y)[4]) {
    const int dk = REVM ? -CH * k : CH * k;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      x[m] = *reinterpret_cast<const f4u*>(px + min(max(offX[m] + dk, 0), xmaxo));   // REV: memory order, see load_general
      y[m] = *reinterpret_cast<const f4u*>(py + min(max(offY[m] + dk, 0), ymaxo));
    }
  };
  auto drain_fast = [&](int k) {
    const f4* sD = FTR_TD(k);
    float* ws_k = REVM ? wsb - CH * k : wsb + CH * k;
    f4 v[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) v[m] = sD[fq * PLANE + 16 * m + frow];
    // all LDS reads are issued before the first predicated store: left to itself the compiler sinks each read into
    // its store's exec-masked block and waits for them one at a time (seen in the ISA: 8 serial LDS round trips)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f4 g = to_G(v[m]);
      if (rvalid[m]) *reinterpret_cast<f4u*>(ws_k + offG[m]) = REVM ? rev4(g) : g;
    }
  };

  // edge chunks: the same reads and the same precomputed offsets; a quad that lies inside the rectangle is stored whole, one
  // that straddles its edge element by element (in start chunk k only the rows 16k .. 16k+15 have such a quad, so three
  // of the four row groups skip that branch).  Per-quad 64-bit addresses here (the first version) made these slots 0.95 us
  // against the 0.7 us of the steady state, in the first four slots of every band.
  auto drain_edge = [&](int k) {
    const f4* sD = FTR_TD(k);
    float* ws_k = REVM ? wsb - CH * k : wsb + CH * k;
    f4 v[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) v[m] = sD[fq * PLANE + 16 * m + frow];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (CH * k + 15 - SKEW * 16 * m < 0 || CH * k - SKEW * (16 * m + 15) >= Tn) continue;   // wave-uniform: no column of this row group lies in the rectangle
      const f4 g = to_G(v[m]);
      const int c0 = CH * k + 4 * fq - SKEW * (16 * m + frow);   // walk column of the quad's first step
      if (rvalid[m]) {
        float* dst = ws_k + offG[m];
        if (c0 >= 0 && c0 + 3 < Tn) {
          *reinterpret_cast<f4u*>(dst) = REVM ? rev4(g) : g;
        } else if (c0 + 3 >= 0 && c0 < Tn) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e >= 0 && c0 + e < Tn) dst[REVM ? 3 - e : e] = g[e];
        }
      }
    }
  };

  const int K0 = MOD ? 1 : 4;                                      // 16k - 63*SKEW >= 1
#ifdef FTR_EXP_ALLGENERIC
  const int K1 = 0;
#else
  const int K1 = min((Tn >= CH) ? (Tn - CH) / CH + 1 : 0, klast);  // 16k + 15 < Tn
#endif

  if (wid == 3) {
    // ------------------------------------------------------------------------- IO-out: G, and every store of the hand-off
    u64* gran_out = gran_b + (size_t)(w + 1) * Tg;        // read (and cleared) by band w+1
    u64* gran_in = gran_b + (size_t)w * Tg;               // written by band w-1, imported by this band's COMM wave
    // publish only what will be imported (and cleared): the band below returns at once when all its rows lie beyond
    // the cut (its jl = jl - 64 * SKEW is negative); otherwise it imports exactly the chunks [0, klast) published here
    const bool has_up = w > 0, has_down = (w + 1 < NWact) && (jl - 64 * SKEW >= 0);
    auto clear_imported = [&](int mm) {   // what the COMM wave imported one slot ago: leave it zero for the next launch
      if (has_up && mm >= 0 && mm < klast_up) comm_clear(gran_in, mm, lane);
    };
    // frame bookkeeping (see "Frames"): while chunk k is handled, frame_k = base + step[0] + ... + step[k] (the steps are set
    // by the COMM wave, three chunks ahead)
    int frame_k = 0;
    const bool lane_valid = 64 * w + lane < Sn;
    // TOUCH.  A chunk's 16 granules are one 128-byte line.  Inside a training step the memory-side cache is full of other
    // kernels' dirty lines, and a write-through store to a line that is NOT in it has to make room first -- the late
    // visibility of the hand-off that costs the chained bands 8 - 14 us per launch inside the step.  So this wave,
    // which never waits for a load, reads one granule of each of this band's INCOMING lines during its first slots (64 lines
    // per slot: the whole region of c3 in one slot, of c5 in eight), long before the band above publishes into most of them:
    // the loads take the misses, off everybody's critical path, and the stores find their lines.  (From the COMM wave the same
    // loads are harmful: vmcnt retires in order, so its next import waits behind them -- c5 +65 / +85 us.)
    u64 touched = 0;
    for (int gg = 0; gg < NIT * NPF; ++gg) {
      const int kc = base + gg;
      const int k = kc - 1;                 // the chunk the compute wave finished in the previous slot
#ifndef FTR_EXP_NOTOUCH
      if (has_up && 64 * gg + lane < klast_up) touched = __hip_atomic_load(gran_in + CH * (64 * gg + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      if (k >= 0 && k < klast) {
        const float* tp = reinterpret_cast<const float*>(FTR_TP(k));
        // chunk 0: the base frame the COMM wave chose (slots ago) counts as chunk 0's step
        const int step_k = (int)sring[k & 7] + (k == 0 ? (int)sring[8] : 0);
        frame_k += step_k;
#ifdef FTR_EXP_NOPUBLISH   // test build (tests/test_gpu_mi.py poison path): the first alpha band never publishes
        if (has_down && lane < CH && !(w == 0 && !REVM)) {
#else
        if (has_down && lane < CH) {   // lane 63's p of the 16 steps of chunk k -> granules of the band below (first: latency critical)
#endif
          const float v = tp[(((lane >> 2) * PLANE + 63) << 2) + (lane & 3)];
          const unsigned tag = (unsigned)(k + 1) | ((unsigned)step_k << 16);   // 16-bit signed step above the chunk number
          const u64 g = ((u64)tag << 32) | (u64)__float_as_uint(v);
          publish_granule(gran_out + CH * k + lane, g);
        }
        if (k == jl / CH) {   // this band's values on the cut (local step jl) and the frame they are in
          if (lane_valid) pmid_store(pmid_b + 64 * w + lane, tp[((((jl & (CH - 1)) >> 2) * PLANE + lane) << 2) + (jl & 3)]);
          if (lane == 0) __hip_atomic_store(phimid_b + w, frame_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      clear_imported(kc - 1 + LOOK);
      if (k >= 0 && k < klast) {
        if (k >= K0 && k < K1) drain_fast(k);   // wave-uniform
        else drain_edge(k);
      }
      FTR_SYNC();
    }
    clear_imported(base + NIT * NPF - 1 + LOOK);
    asm volatile("" ::"v"(touched));
    FTR_SYNC_REPORT(3);
    return;
  }

  // --------------------------------------------------------------------------- IO-in
  // degenerate slabs (fewer than 4 elements of px or py per utterance, or no px at all): guarded element loads
  const bool tiny = xmaxo < 0 || ymaxo < 0;
  // an edge chunk has quads outside [1, Tn); readfirstlane: the compiler must see the flag as wave-uniform, or it runs
  // the masking code under an exec mask in every slot (seen: the slot of this wave doubled)
  auto is_edge = [&](int k) { return __builtin_amdgcn_readfirstlane((int)!(k >= K0 && k < K1)) != 0; };
  // One loop for every slot; slot gg = kc - base uses register set gg % NPF.  Steady schedule: park the chunk this set
  // holds (loaded NPF slots ago, computed kAhead slots from now), load the chunk NPF further on.  Past the last chunk the
  // loads are still issued, with every lane's address clamped to the same 16 bytes (no bandwidth, the values are never
  // parked): a load under a condition turns every wait of the loop into vmcnt(0), and so did a separate warm-up loop with
  // a forced vmcnt(0) behind it (the first park waited for four chunks' cold misses, 2.2 us in band 0's trace).
  //
  // A chained band (one with a band above it) starts differently.  Its first four chunks are edge chunks, whose parks take
  // 1.1-1.4 us against the 0.7 us slot of the steady state (scripts/mi_trace_waves.py), and in the steady schedule they
  // sit in the slots in which the band already follows the band above at that band's steady pace: every band fell a
  // further ~2 us behind, for good.  Such a band runs NPF more warm-up slots: it loads chunks 0..3 in slots 0..3 and parks
  // them in slots 4..7, while its COMM wave still waits for the first granules (four input tiles, so that all four fit);
  // slots 8..11 park nothing (they load again what the sets already hold: the registers must stay valid) and the steady
  // schedule resumes with chunk 4 in slot 12.  Every chunk still lives in set (chunk % NPF) and is parked NPF slots after
  // its (last) load, so the compiler's counted waits hold for both schedules.
  // Same box, head -> this (us): c3 53.4 -> 48.9, B=8 46.8 -> 44.3, c5 and B=256 unchanged, c4 112-114 -> 115-116.
  static_assert(NPF == 4, "the early parks assume four register sets and four input tiles");
  constexpr int kNoChunk = 1 << 20;   // CH * kNoChunk is beyond any slab (and far from overflowing an int offset)
  auto slot_any = [&](int gg, f4 (&x)[4], f4 (&y)[4]) {
    if (gg == NPF) { cx2 = shift_lds[0]; cy2 = shift_lds[1]; }   // written by the COMM wave in slot 3; the first park is below, in this slot
    const int kp = chained ? (gg < 2 * NPF ? gg - NPF : (gg < 3 * NPF ? -1 : gg - 2 * NPF)) : gg - NPF;
    const int kl = chained && gg >= 2 * NPF ? gg - NPF : gg;
    if (kp >= 0 && kp < klast) {
      if (is_edge(kp)) park(kp, x, y, std::true_type{}, offX, offY);
      else park(kp, x, y, std::false_type{}, offX, offY);
    }
    load_fast(kl < klast ? kl : kNoChunk, x, y);
    FTR_SYNC();
  };
  if (tiny) {
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        const int kc = base + NPF * it + u;
        if (NPF * it + u == NPF) { cx2 = shift_lds[0]; cy2 = shift_lds[1]; }
        if (kc + kAhead >= 0 && kc + kAhead < klast) park(kc + kAhead, rx[u], ry[u], std::false_type{}, offX, offY);
        if (kc + kAhead + NPF >= 0 && kc + kAhead + NPF < klast) load_general(kc + kAhead + NPF, rx[u], ry[u]);
        FTR_SYNC();
      }
    }
  } else {
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) slot_any(NPF * it + u, rx[u], ry[u]);
    }
  }
  if (__any(nan_acc > 0x7f800000u) && lane == 0) __hip_atomic_fetch_or(uflag_b, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  FTR_SYNC_REPORT(1);
}

#undef FTR_TX
#undef FTR_TY
#undef FTR_TD
#undef FTR_TP

// ----------------------------------------------------------------------------------------------------- cut
// The cut reduction, run by the last workgroup of an utterance to deliver its cut values (256 threads):
// ans[b] = logsumexp over the cut of p + q (log2 domain in, natural log out); occ[b][s - s_begin] = the occupancy of the
// cut cell in lattice row s.  Alpha lane r and beta lane Sn-1-r hold the same cell.  pa / pb were written with
// write-through stores by workgroups anywhere on the chip: agent-scope loads.
__device__ __forceinline__ float pmid_load(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int phimid_load(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
// p + q of a cut cell = (alpha value + beta value) + (the two bands' frames, integers): formed in double, where the
// frames (up to ~1e5) cost no precision; everything after the maximum has been subtracted is float again.
__device__ __forceinline__ void cut_reduce(float* red, double* cache, int cache_n, const float* __restrict__ pa,
                                           const float* __restrict__ pb, const int* __restrict__ fa,
                                           const int* __restrict__ fb, float* __restrict__ ob,
                                           float* __restrict__ ans_b, int Sn, bool poisoned, double shift) {
  // one pass over the (remote) cut values: p + q goes to an LDS cache while the maximum is formed; the sum and the
  // occupancies come from the cache (rows beyond the cache, Sn > cache_n, are re-read)
  auto remote = [&](int r) {
    const int rb = Sn - 1 - r;
    return ((double)pmid_load(pa + r) + (double)pmid_load(pb + rb)) + (double)(phimid_load(fa + (r >> 6)) + phimid_load(fb + (rb >> 6)));
  };
  auto val = [&](int r) { return r < cache_n ? cache[r] : remote(r); };
  double* dred = reinterpret_cast<double*>(red);   // 4 doubles (the caller leaves 8 floats)
  double m = -1.0e300;
  for (int r = threadIdx.x; r < Sn; r += 256) {
    const double v = remote(r);
    if (r < cache_n) cache[r] = v;
    m = fmax(m, v);
  }
  m = wave_max_f64(m);
  if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmax(fmax(dred[0], dred[1]), fmax(dred[2], dred[3]));
  __syncthreads();
  float sum = 0.0f;
  for (int r = threadIdx.x; r < Sn; r += 256) sum += exp2f((float)(val(r) - m));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const double total = m + (double)log2f(sum);
  const bool dead = !(total > (double)kNegThresh);      // no path: ans = -inf, no flow
  // a NaN among the inputs of this utterance, or a band that gave up waiting: ans = NaN (loud), no flow
  // `shift`: what the per-utterance operand shift (ftr_common.h, Shift) took out of every complete path, log2 units
  if (threadIdx.x == 0) *ans_b = poisoned ? __builtin_nanf("") : (dead ? -INFINITY : (float)((total + shift) * 0.6931471805599453));
  // normalised with the very sum they add up to (not with exp2(-total)): the injected occupancies sum to 1 to rounding
  const float inv = 1.0f / sum;
  for (int r = threadIdx.x; r < Sn; r += 256)
    ob[r] = (dead || poisoned) ? 0.0f : exp2f((float)(val(r) - m)) * inv;
}

struct Ctrl {           // int offsets into the ctrl block of the workspace
  int* status; int* done; int* uflags;
};
// the status word is the hand-off region's LAST block (4 ints): the same address for every shape launched on one buffer
__device__ __forceinline__ Ctrl ctrl_of(int* ctrl, int B, int soff) {
  Ctrl c; c.status = ctrl + soff; c.done = ctrl + 4; c.uflags = ctrl + 4 + B; return c;
}

template <bool MOD>
__global__ __launch_bounds__(256) void mi_bidir_fwd_kernel(
    const float* __restrict__ px, const float* __restrict__ py, const int32_t* __restrict__ boundary,
    float* __restrict__ ws, u64* __restrict__ gran, float* __restrict__ pmid, float* __restrict__ occ, float* __restrict__ cxy,
    int* __restrict__ phimid, int* __restrict__ ctrl, int soff, float* __restrict__ ans, int B, int NB, int Tg, int S, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // band-major block ids: producers (lower band index) have lower ids.  Forward progress of a band that waits for the
  // band above relies on the dispatcher starting workgroups in id order (true on this hardware; an oversubscribed grid
  // keeps working because a band only ever waits for lower ids); if that ever fails the wait is bounded (kMaxSpin) and
  // sets the sticky status word.
  const int b2 = blockIdx.x % (2 * B);
  const int w = blockIdx.x / (2 * B);            // band of 64 walk rows
  const int dir = b2 / B, b = b2 - dir * B;
#if defined(FTR_TRACE) && (FTR_TRACE == 1 || FTR_TRACE == 3)
  if (threadIdx.x == 0) { const u64 t = trace_now(); atomicMin(&g_trace[0], t); if (b == 0 && dir == FTR_TRACE_DIR && w == FTR_STAMP_BAND) g_trace[2] = t; }
#endif
  const Bound bd = load_boundary(boundary, b, S, T);
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  if (Sn <= 0 || Tn <= 0) {                      // empty rectangle: ans = 0 (the reference never writes it)
    if (dir == 0 && w == 0 && threadIdx.x == 0) ans[b] = 0.0f;
    return;
  }
  const int NWact = (Sn + 63) >> 6;
  if (w >= NWact) return;                        // bands past the utterance's last row: nobody waits for them
  const Cut cut = make_cut<MOD>(Sn, Tn);
  const int T1 = MOD ? T : T + 1;
  const float* pxb = px + (size_t)b * S * T1;
  const float* pyb = py + (size_t)b * (S + 1) * T;
  float* wsb = ws + kLatPad + (size_t)dir * lattice_floats(B, S, T) + (size_t)b * (S + 1) * (T + 1);
  u64* gran_b = gran + ((size_t)dir * B + b) * NB * Tg;
  float* pmid_b = pmid + ((size_t)dir * B + b) * (S + 1);
  const Ctrl c = ctrl_of(ctrl, B, soff);
  int* phimid_b = phimid + ((size_t)dir * B + b) * NB;   // frame of each band's cut values (see "Frames" in the body)
  if (dir == 0) bidir_fwd_body<MOD, false>(smem, pxb, pyb, bd, wsb, gran_b, pmid_b, c.status, c.uflags + b, cxy + 2 * b, phimid_b, b, w, Tg, S, T, cut.jm);
  else bidir_fwd_body<MOD, true>(smem, pxb, pyb, bd, wsb, gran_b, pmid_b, c.status, c.uflags + b, cxy + 2 * b, phimid_b, b, w, Tg, S, T, cut.D - cut.jm);

  // ---- the last of the 2 * NWact bands of this utterance to get here runs the cut reduction
  __builtin_amdgcn_s_waitcnt(kVmcnt0);           // this wave's cut values / flags have left
  __syncthreads();
#if defined(FTR_TRACE) && (FTR_TRACE == 1 || FTR_TRACE == 3)
  if (threadIdx.x == 0) { const u64 t = trace_now(); atomicMax(&g_trace[1], t); if (b == 0 && dir == FTR_TRACE_DIR && w == FTR_STAMP_BAND) g_trace[3] = t; }
#endif
  int* sflag = reinterpret_cast<int*>(smem);
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(c.done + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sflag[0] = (old == 2 * NWact - 1);
  }
  __syncthreads();
  if (!sflag[0]) return;
  const int uf = __hip_atomic_load(c.uflags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int stt = __hip_atomic_load(c.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x == 0) {                        // counters back to zero for the next launch on this workspace
    __hip_atomic_store(c.done + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(c.uflags + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  Shift sh;   // written by the first alpha band of this utterance (write-through stores, long before its done[] increment)
  sh.cx2 = pmid_load(cxy + 2 * b); sh.cy2 = pmid_load(cxy + 2 * b + 1);
  cut_reduce(reinterpret_cast<float*>(smem) + 4, reinterpret_cast<double*>(smem) + 8, 4096, pmid + (size_t)b * (S + 1),
             pmid + ((size_t)B + b) * (S + 1), phimid + (size_t)b * NB, phimid + ((size_t)B + b) * NB, occ + (size_t)b * (S + 1),
             ans + b, Sn, (uf | stt) != 0, shift_total<MOD>(sh, Sn, Tn));
}

// ---------------------------------------------------------------------------------------------------- flow
#define FTR_TG(k) (lds + (((k) + 3) % 3) * TILE_F4)           // 3 input buffers, see the forward body
#define FTR_TPX(k) (lds + (3 + ((k) & 1)) * TILE_F4)
#define FTR_TPY(k) (lds + (5 + ((k) & 1)) * TILE_F4)
#define FTR_TXO(k) (lds + (7 + ((k) & 1)) * TILE_F4)
constexpr int kFlowTiles = 9;
static_assert(kFlowTiles <= kFwdTiles, "bidir_lds_bytes() is sized for the forward kernel");

// One direction of one band of the backward pass.  REVM = true: from the cut back to the origin (alpha half),
// REVM = false: from the cut forward to the end cell (beta half).  jinj = walk step of the cut in this direction.
template <bool MOD, bool REVM>
__device__ __forceinline__ void bidir_flow_body(unsigned char* smem, const Bound bd, const float* __restrict__ wsb,
                                                u64* __restrict__ gran_b, const float* __restrict__ occ_b,
                                                float* __restrict__ pxg, float* __restrict__ pyg,
                                                const float* __restrict__ seed, float* __restrict__ check,
                                                int* __restrict__ status, int b, int w, int Tg,
                                                int S, int T, int jinj) {
  constexpr int SKEW = MOD ? 0 : 1;
  constexpr int NOFF = MOD ? 1 : 0;
  constexpr int NPF = FTR_NPF_FLOW;
  constexpr int LOOK = MOD ? 1 : 5;      // see the forward body
  constexpr int PRE = NPF + kAhead;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0 compute, 1 IO-in, 2 COMM, 3 IO-out
  const int T1 = MOD ? T : T + 1;
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  const int NWact = (Sn + 63) >> 6;

  f4* lds = reinterpret_cast<f4*>(smem);
  float* in_ring = reinterpret_cast<float*>(lds + kFlowTiles * TILE_F4);   // tiles: G (three), PX, PY, XO (two of each)
  for (int i = threadIdx.x; i < RINGN; i += blockDim.x) in_ring[i] = 0.0f;
  __syncthreads();

  const int nchunks = (Tn + 63 * SKEW + CH - 1) / CH;
  // band-local walk step of the cut (see the forward body); negative: every cell of this band lies past the cut,
  // the band starts at its first step with nothing injected and receives its flow from the band above.
  const int jli = jinj - SKEW * 64 * w;
  const int kfirst = max(jli, 0) / CH;                // chunks [kfirst, nchunks)
  const int kfirst_up = MOD ? kfirst : max(jli + 64, 0) / CH;   // first chunk the band above computes (and publishes)
  const int nslots = (nchunks - kfirst) + PRE + 1;
  const int NIT = (nslots + NPF - 1) / NPF;
  const int base = kfirst - PRE;
#ifdef FTR_STAMP_FLOW
  FTR_SYNC_DECL;
#define FTR_FSYNC() FTR_SYNC()
#define FTR_FREPORT(slot) do { if (REVM && b == 0 && w == FTR_STAMP_BAND && lane == 0) { g_stamps[4 * (slot)] = st_busy; g_stamps[4 * (slot) + 1] = st_wait; g_stamps[4 * (slot) + 2] = st_n; } } while (0)
#else
#define FTR_FSYNC() __syncthreads()
#define FTR_FREPORT(slot) do { } while (0)
#endif

  const int wfin = (Sn - 1) >> 6, lfin = (Sn - 1) & 63;
  // REVM: the walk ends at the origin, where p_grad[s_begin,t_begin] appears (the ans_grad self check)
  const int jfin = (REVM && w == wfin) ? (Tn - 1 + SKEW * lfin) : -1000;

  if (wid == 0) {
    // ======================================================================= COMPUTE wave
    // (as in the forward body: arithmetic only; the three per-cell outputs leave through LDS tiles)
    const f4* ring_in = reinterpret_cast<const f4*>(in_ring);
    // occupancy of this lane's cut cell, scaled by the incoming gradient
    float inj = 0.0f;
    {
      const int r = 64 * w + lane;
      if (r < Sn) inj = occ_b[REVM ? (Sn - 1 - r) : r] * (seed ? seed[b] : 1.0f);   // NULL seed = ones
    }
    float yprev = 0.0f, xprev = 0.0f, ecarry = 0.0f;

    // operands of a whole chunk in registers, fetched one slot ahead (see the forward body)
    struct Ops { f4 G[NQ]; };
    auto fetch = [&](int k, Ops& o) {
      const f4* cG = FTR_TG(k);
#pragma unroll
      for (int q = 0; q < NQ; ++q) o.G[q] = cG[q * PLANE + lane];
    };
    auto compute_chunk = [&](int k, const Ops& o, auto inject_tag) {
      constexpr bool INJ = decltype(inject_tag)::value;
      f4* cPX = FTR_TPX(k);
      f4* cPY = FTR_TPY(k);
      f4* cXO = FTR_TXO(k);
      f4 E[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) E[q] = ring_in[((CH * k + 4 * q) & (RINGN - 1)) >> 2];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int j0 = CH * k + 4 * q;
        const f4 G4 = o.G[q], E4 = E[q];
        f4 XO4, PX4, PY4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ev = (e == 0) ? ecarry : E4[e - 1];
          const float xin = dpp_wave_shr1(ev, xprev);
          float pg = xin + yprev;
          if (INJ && j0 + e == jli) pg += inj;    // the cut: nothing has flowed yet, the occupancy enters here
          PX4[e] = xin;    // flow through the px transition between this cell and its walk predecessor row
          PY4[e] = yprev;  // flow through the py transition between this cell and its walk predecessor column
          // steps in front of the cut (first chunk only) lie in the other half: their ratios were never computed
          // (uninitialised memory, possibly NaN), so the zero flow there is forced rather than multiplied
          const bool pre = INJ && (j0 + e < jli);
          xprev = pre ? 0.0f : pg * G4[e];
          yprev = pre ? 0.0f : pg - xprev;
          XO4[e] = xprev;
        }
        ecarry = E4[3];
        cPX[q * PLANE + lane] = PX4;
        cPY[q * PLANE + lane] = PY4;
        cXO[q * PLANE + lane] = XO4;
      }
    };

    Ops oa, ob;
    auto slot = [&](int kc, const Ops& cur, Ops& nxt) {
      if (kc + 1 >= kfirst && kc + 1 < nchunks) fetch(kc + 1, nxt);
#ifndef FTR_EXP_FLOW_NOCOMPUTE
      if (kc >= kfirst && kc < nchunks) {
        if (kc == kfirst) {
          ecarry = in_ring[(CH * kfirst - 1) & (RINGN - 1)];   // the element in front of the first chunk
          compute_chunk(kc, cur, std::true_type{});
        } else {
          compute_chunk(kc, cur, std::false_type{});
        }
      }
#endif
      FTR_FSYNC();
#if defined(FTR_TRACE) && FTR_TRACE == 2
      if (b == 0 && REVM == (FTR_TRACE_DIR == 0) && w == FTR_STAMP_BAND && lane == 0 && 16 + (kc - base) < kTraceN) { g_trace[16 + (kc - base)] = trace_now(); g_trace[4] = kc - base + 1; }
#endif
    };
    static_assert((NPF & 1) == 0, "the slot loop is unrolled by two");
    for (int gg = 0; gg < NIT * NPF; gg += 2) {
      slot(base + gg, oa, ob);
      slot(base + gg + 1, ob, oa);
    }
    FTR_FREPORT(0);
    return;
  }

  // ========================================================================= IO waves (IO-in = wid 1, IO-out = wid 3)
  const int row0 = 64 * w;
  const int frow = lane >> 2, fq = lane & 3;
  f4 rg[NPF][4];

  auto park = [&](int kk, const f4 (&gq)[4], auto edge_tag) {
    constexpr bool edge = decltype(edge_tag)::value;
    f4* dG = FTR_TG(kk);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      f4 g = REVM ? rev4(gq[m]) : gq[m];   // REV loads stay in memory order until here
      if (edge) {   // wave-uniform: columns outside [0, Tn) hold no ratio (whatever was loaded there): no flow
        const int c0 = CH * kk + 4 * fq - SKEW * (16 * m + frow);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = (c0 + e >= 0 && c0 + e < Tn) ? g[e] : 0.0f;
      }
      dG[fq * PLANE + 16 * m + frow] = g;
    }
  };
  // Steps up to and including the cut (local walk step <= jli) belong to the other half and are never written.
  auto drain_general = [&](int k, auto xtag, auto ytag) {
    constexpr bool DOX = decltype(xtag)::value, DOY = decltype(ytag)::value;
    const f4* sX = FTR_TPX(k);
    const f4* sY = FTR_TPY(k);
    const int jq = CH * k + 4 * fq;        // walk step of element 0
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 16 * m + frow;
      const int r = row0 + row;
      if (r < Sn) {
        const int c0 = CH * k + 4 * fq - SKEW * row;
        f4 gx = {0.f, 0.f, 0.f, 0.f}, gy = gx;
        if (DOX) gx = sX[fq * PLANE + row];
        if (DOY) gy = sY[fq * PLANE + row];
        const bool whole = jq > jli;
        if (REVM) {
          const int s = bd.se - r;
          if (DOX && r >= 1) {  // px_grad[s][t]: rows s < se; walk columns c in [NOFF, Tn)
            const ptrdiff_t lo = (ptrdiff_t)s * T1 + bd.te - c0 - 3;
            if (whole && c0 >= NOFF && c0 + 3 < Tn) {
              *reinterpret_cast<f4u*>(pxg + lo) = rev4(gx);
            } else if (jq + 3 > jli && c0 + 3 >= NOFF && c0 < Tn) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (jq + e > jli && c0 + e >= NOFF && c0 + e < Tn) pxg[lo + 3 - e] = gx[e];
            }
          }
          if (DOY) {  // py_grad[s][t]: columns t < te  <=>  c >= 1
            const ptrdiff_t lo = (ptrdiff_t)s * T + bd.te - c0 - 3;
            if (whole && c0 >= 1 && c0 + 3 < Tn) {
              *reinterpret_cast<f4u*>(pyg + lo) = rev4(gy);
            } else if (jq + 3 > jli && c0 + 3 >= 1 && c0 < Tn) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (jq + e > jli && c0 + e >= 1 && c0 + e < Tn) pyg[lo + 3 - e] = gy[e];
            }
          }
        } else {
          if (DOX && r >= 1) {  // px_grad[s-1][t + toff]: the transition INTO this cell from the row below
            const int cx = MOD ? c0 - 1 : c0;
            const ptrdiff_t o = (ptrdiff_t)(bd.sb + r - 1) * T1 + bd.tb + cx;
            if (whole && cx >= 0 && c0 + 3 < Tn) {
              *reinterpret_cast<f4u*>(pxg + o) = gx;
            } else if (jq + 3 > jli && cx + 3 >= 0 && c0 < Tn) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (jq + e > jli && cx + e >= 0 && c0 + e < Tn) pxg[o + e] = gx[e];
            }
          }
          if (DOY) {  // py_grad[s][t-1]: the transition INTO this cell from the previous frame
            const ptrdiff_t o = (ptrdiff_t)(bd.sb + r) * T + bd.tb + c0 - 1;
            if (whole && c0 >= 1 && c0 + 3 < Tn) {
              *reinterpret_cast<f4u*>(pyg + o) = gy;
            } else if (jq + 3 > jli && c0 + 3 >= 1 && c0 < Tn) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (jq + e > jli && c0 + e >= 1 && c0 + e < Tn) pyg[o + e] = gy[e];
            }
          }
        }
      }
    }
  };

  // ---- interior ("fast") chunks, see the forward body.  Clamped rows read some valid row's G: their flow is
  // exactly zero (nothing is injected into them and nothing flows past the last valid row), so it cannot matter.
  int offG[4], offPX[4], offPY[4];
  bool rvalid[4], xvalid[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int row = 16 * m + frow;
    const int r = row0 + row;
    const int cq = 4 * fq - SKEW * row;
    const int rc = min(r, Sn - 1);
    if (REVM) {
      offG[m] = (bd.se - rc) * (T + 1) + bd.te - cq - 3;
      offPX[m] = (bd.se - r) * T1 + bd.te - cq - 3;
      offPY[m] = (bd.se - r) * T + bd.te - cq - 3;
    } else {
      offG[m] = (bd.sb + rc) * (T + 1) + bd.tb + cq;
      offPX[m] = (bd.sb + r - 1) * T1 + bd.tb + cq + (MOD ? -1 : 0);
      offPY[m] = (bd.sb + r) * T + bd.tb + cq - 1;
    }
    rvalid[m] = r < Sn;
    xvalid[m] = r >= 1 && r < Sn;
  }
  auto load_fast = [&](int k, f4 (&gq)[4]) {
#ifdef FTR_EXP_FLOW_NOLOAD
    return;
#endif
    const float* ws_k = REVM ? wsb - CH * k : wsb + CH * k;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      gq[m] = *reinterpret_cast<const f4u*>(ws_k + offG[m]);
    }
  };
  auto drain_fast = [&](int k, auto xtag, auto ytag) {
    constexpr bool DOX = decltype(xtag)::value, DOY = decltype(ytag)::value;
    const f4* sX = FTR_TPX(k);
    const f4* sY = FTR_TPY(k);
    float* px_k = REVM ? pxg - CH * k : pxg + CH * k;
    float* py_k = REVM ? pyg - CH * k : pyg + CH * k;
    f4 gx[4], gy[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (DOX) gx[m] = sX[fq * PLANE + 16 * m + frow];
      if (DOY) gy[m] = sY[fq * PLANE + 16 * m + frow];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all LDS reads in flight together, see the forward body
#ifdef FTR_EXP_FLOW_NOSTORE
    if ((DOX ? gx[0][0] : gy[0][0]) != 12345.678f) return;
#endif
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (DOX && xvalid[m]) *reinterpret_cast<f4u*>(px_k + offPX[m]) = REVM ? rev4(gx[m]) : gx[m];
      if (DOY && rvalid[m]) *reinterpret_cast<f4u*>(py_k + offPY[m]) = REVM ? rev4(gy[m]) : gy[m];
    }
  };

  const int K0 = max(MOD ? 1 : 4, kfirst);
#ifdef FTR_EXP_ALLGENERIC
  const int K1 = 0;
#else
  const int K1 = (Tn >= CH) ? (Tn - CH) / CH + 1 : 0;
#endif

  const int K0d = max(MOD ? 1 : 4, kfirst + 1);     // the cut's chunk is drained with the per-step mask

  if (wid == 2) {
    // ======================================================================= COMM wave
    // LOADS only, as in the forward kernel: the import of the band above's flow and the request for the next chunk's
    // granules.  Every store of the hand-off is the IO-out wave's, the py_grad stores have a wave of their own (five waves
    // per workgroup): a wave that polls must not have write-through stores of its own in flight, vmcnt retires in order.
    u64* gran_in = gran_b + (size_t)w * Tg;
    const bool has_up = w > 0;
    bool dead = false;
    u64 g_cur = 0;
    for (int gg = 0; gg < NIT * NPF; ++gg) {
      const int kc = base + gg;
      const int m = kc + LOOK;
      if (has_up && !dead && m >= kfirst_up && m < nchunks) {
        if (!comm_import(in_ring, gran_in, m, lane, g_cur, status)) {
          dead = true;
          in_ring[lane] = __builtin_nanf("");
        }
      }
      // the next chunk's granules are requested now and looked at one whole slot later
      g_cur = 0;
      if (has_up && !dead && m + 1 >= kfirst_up && m + 1 < nchunks) g_cur = comm_peek(gran_in, m + 1, lane);
      FTR_FSYNC();
    }
    FTR_FREPORT(2);
    return;
  }

  if (wid == 3) {
    // ------------------------------------------------------------------------- IO-out: px_grad and every store of the hand-off
    // (to the band below: lane 63's xout of every step, read from the XO tile; clearing what the COMM wave imported one
    // slot ago; the ans_grad self check)
    u64* gran_out = gran_b + (size_t)(w + 1) * Tg;
    u64* gran_in = gran_b + (size_t)w * Tg;
    const bool has_up = w > 0, has_down = w + 1 < NWact;
    // publish only what the band below imports (and clears): its loop visits the chunks from
    // max(kfirst_up, its kfirst - PRE + LOOK) on, with its jli = jli - 64 * SKEW
    const int kpub = max(kfirst, max(jli - 64 * SKEW, 0) / CH - PRE + LOOK);
    auto clear_imported = [&](int mm) { if (has_up && mm >= kfirst_up && mm < nchunks) comm_clear(gran_in, mm, lane); };
    u64 touched = 0;   // this band's incoming hand-off lines are read once, early (see the forward kernel's IO-out wave: TOUCH)
    for (int gg = 0; gg < NIT * NPF; ++gg) {
      const int kc = base + gg;
      const int k = kc - 1;
#ifndef FTR_EXP_NOTOUCH
      if (has_up && kfirst_up + 64 * gg + lane < nchunks) touched = __hip_atomic_load(gran_in + CH * (kfirst_up + 64 * gg + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      if (k >= kfirst && k < nchunks) {
        if (has_down && k >= kpub && lane < CH) {
          const float* txo = reinterpret_cast<const float*>(FTR_TXO(k));
          const float v = txo[(((lane >> 2) * PLANE + 63) << 2) + (lane & 3)];
          const u64 g = ((u64)(unsigned)(k + 1) << 32) | (u64)__float_as_uint(v);
          publish_granule(gran_out + CH * k + lane, g);
        }
        if (check && k == (jfin >> 4) && lane == lfin) {   // p_grad at the origin = both inflows (the self check)
          const int idx = ((((jfin & (CH - 1)) >> 2) * PLANE + lane) << 2) + (jfin & 3);
          float pg = reinterpret_cast<const float*>(FTR_TPX(k))[idx] + reinterpret_cast<const float*>(FTR_TPY(k))[idx];
          if (jfin == jli) pg += occ_b[REVM ? (Sn - 1 - (64 * w + lane)) : (64 * w + lane)] * (seed ? seed[b] : 1.0f);   // one-cell lattice
          check[b] = pg;
        }
      }
      clear_imported(kc - 1 + LOOK);
      if (k >= kfirst && k < nchunks) {
        if (k >= K0d && k < K1) drain_fast(k, std::true_type{}, std::false_type{});   // wave-uniform
        else drain_general(k, std::true_type{}, std::false_type{});
      }
      FTR_FSYNC();
    }
    clear_imported(base + NIT * NPF - 1 + LOOK);
    asm volatile("" ::"v"(touched));
    FTR_FREPORT(3);
    return;
  }

  if (wid == 4) {
    // ------------------------------------------------------------------------- IO-out: py_grad
    for (int gg = 0; gg < NIT * NPF; ++gg) {
      const int k = base + gg - 1;
      if (k >= kfirst && k < nchunks) {
        if (k >= K0d && k < K1) drain_fast(k, std::false_type{}, std::true_type{});   // wave-uniform
        else drain_general(k, std::false_type{}, std::true_type{});
      }
      FTR_FSYNC();
    }
    return;
  }

  // --------------------------------------------------------------------------- IO-in
  // The same four unguarded 16-byte loads for every chunk (the lattices have kLatPad floats in front and the cut vectors
  // behind them, so a quad hanging over a row's valid columns stays inside the workspace); park() masks the columns
  // outside [0, Tn).  A fixed instruction sequence keeps the prefetch pipeline's waits counted in every slot.
  auto is_edge = [&](int k) { return __builtin_amdgcn_readfirstlane((int)!(k >= K0 && k < K1)) != 0; };   // wave-uniform for the compiler too
  auto slot_cond = [&](int kc, f4 (&gq)[4]) {     // pipeline fill and drain: the chunk to park / to load may not exist
    const int kp = kc + kAhead, kl = kc + kAhead + NPF;
    if (kp >= kfirst && kp < nchunks) {
      if (is_edge(kp)) park(kp, gq, std::true_type{});
      else park(kp, gq, std::false_type{});
    }
    if (kl >= kfirst && kl < nchunks) load_fast(kl, gq);
    FTR_FSYNC();
  };
  auto slot_steady = [&](int kc, f4 (&gq)[4]) {
    const int kp = kc + kAhead;
    if (is_edge(kp)) park(kp, gq, std::true_type{});
    else park(kp, gq, std::false_type{});
    load_fast(kp + NPF, gq);
    FTR_FSYNC();
  };

  // steady slot kc: the parked chunk kc+kAhead and the loaded chunk kc+kAhead+NPF both exist
  const int KF0 = kfirst - kAhead, KF1 = nchunks - kAhead - NPF;
  int it1 = (KF0 - base + NPF - 1) / NPF;
  int it2 = (KF1 - base) / NPF;
  it1 = min(max(it1, 0), NIT);
  it2 = min(max(it2, it1), NIT);

  int it = 0;
  for (; it < it1; ++it) {
#pragma unroll
    for (int u = 0; u < NPF; ++u) slot_cond(base + NPF * it + u, rg[u]);
  }
  if (it < it2) {
    __builtin_amdgcn_s_waitcnt(kVmcnt0);   // known state at the loop entry: the waits inside stay counted (see the forward body)
    for (; it < it2; ++it) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) slot_steady(base + NPF * it + u, rg[u]);
    }
  }
  for (; it < NIT; ++it) {
#pragma unroll
    for (int u = 0; u < NPF; ++u) slot_cond(base + NPF * it + u, rg[u]);
  }
  FTR_FREPORT(1);
}
#undef FTR_FSYNC
#undef FTR_FREPORT

#undef FTR_TG
#undef FTR_TPX
#undef FTR_TPY
#undef FTR_TXO

template <bool MOD>
__global__ __launch_bounds__(kFlowThreads) void mi_bidir_flow_kernel(
    const int32_t* __restrict__ boundary, const float* __restrict__ ws, u64* __restrict__ gran,
    const float* __restrict__ occ, float* __restrict__ px_grad, float* __restrict__ py_grad,
    const float* __restrict__ seed, float* __restrict__ check, int* __restrict__ status, const float* __restrict__ ans,
    float* __restrict__ loss_out, int loss_code, int B, int NB, int Tg, int S, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NOFF = MOD ? 1 : 0;
  // The loss tail (-ans, -mean, -sum: what ftr_negated_reduce_f32 computes in a launch of its own) rides along when asked
  // for: `ans` was finished by the forward launch, so one wave of the last workgroup -- whose band has nothing to do in most
  // slots -- reduces it before anything else, with the additions of negated_reduce_kernel (256 virtual threads, four wave
  // sums, (s0 + s1) + (s2 + s3)): the same bits, one kernel boundary less per loss.
  if (loss_out && blockIdx.x == gridDim.x - 1 && threadIdx.x >= kFlowThreads - 64) {
    const int ln = threadIdx.x & 63;
    if (loss_code == 0) {
      for (int i = ln; i < B; i += 64) loss_out[i] = -ans[i];
    } else {
      float sv[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float acc = 0.0f;
        for (int i = 64 * v + ln; i < B; i += 256) acc += ans[i];
        sv[v] = wave_sum_dpp(acc);
      }
      const float t = (sv[0] + sv[1]) + (sv[2] + sv[3]);
      if (ln == 0) loss_out[0] = (loss_code == 1) ? -(t / (float)B) : -t;
    }
  }
  const int b2 = blockIdx.x % (2 * B);
  const int w = blockIdx.x / (2 * B);
  const int dir = b2 / B, b = b2 - dir * B;
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
#if defined(FTR_TRACE) && FTR_TRACE == 2
  if (threadIdx.x == 0) { const u64 t = trace_now(); atomicMin(&g_trace[0], t); if (b == 0 && dir == FTR_TRACE_DIR && w == FTR_STAMP_BAND) g_trace[2] = t; }
#endif
  const Bound bd = load_boundary(boundary, b, S, T);
  const int T1 = MOD ? T : T + 1;
  const int Sn = bd.se - bd.sb + 1, Tn = bd.te - bd.tb + 1;
  float* pxg = px_grad + (size_t)b * S * T1;
  float* pyg = py_grad + (size_t)b * (S + 1) * T;

  // ---- zeros outside the boundary rectangle (the reference memsets everything first,
  //      tf_fast_rnnt_op.cc:93-96); the rectangle itself is fully written by the two sweeps.
  //      dir 0 workgroups fill px_grad, dir 1 workgroups fill py_grad.
  {
    const bool empty = (Sn <= 0 || Tn <= 0);
    const int nwv = (kFlowThreads / 64) * NB;   // every band's waves share the fill of this utterance
    const int fwid = (kFlowThreads / 64) * w + wid;
    if (dir == 0) {
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
    } else {
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
    }
    if (empty) return;
  }
  if (w >= ((Sn + 63) >> 6)) return;
  const Cut cut = make_cut<MOD>(Sn, Tn);
  const float* wsb = ws + kLatPad + (size_t)dir * lattice_floats(B, S, T) + (size_t)b * (S + 1) * (T + 1);
  u64* gran_b = gran + ((size_t)dir * B + b) * NB * Tg;
  const float* occ_b = occ + (size_t)b * (S + 1);
  // dir 0 holds the alpha ratios: its flow runs in REV addressing from the cut (walk step D - jm) to the origin;
  // dir 1 holds the beta ratios: its flow runs in FWD addressing from the cut (walk step jm) to the end cell.
  // `seed` (the incoming ans_grad, never written by this launch) and `check` (p_grad at the origin, written by the band
  // that reaches it) are different buffers: no workgroup reads what another one writes
  if (dir == 0) bidir_flow_body<MOD, true>(smem, bd, wsb, gran_b, occ_b, pxg, pyg, seed, check, status, b, w, Tg, S, T, cut.D - cut.jm);
  else bidir_flow_body<MOD, false>(smem, bd, wsb, gran_b, occ_b, pxg, pyg, seed, nullptr, status, b, w, Tg, S, T, cut.jm);
#if defined(FTR_TRACE) && FTR_TRACE == 2
  __builtin_amdgcn_s_waitcnt(kVmcnt0);
  if (lane == 0) { const u64 t = trace_now(); atomicMax(&g_trace[1], t); if (wid == 0 && b == 0 && dir == FTR_TRACE_DIR && w == FTR_STAMP_BAND) g_trace[3] = t; }
#endif
}

inline size_t bidir_lds_bytes() { return (size_t)kFwdTiles * TILE_F4 * sizeof(f4) + 2 * RINGN * sizeof(float); }   // forward: 12 tiles + ring; flow: 9 tiles + ring

// The workgroup dispatcher fills a CU up to its resource limits before it moves on: with 26 KB of LDS per workgroup
// it co-locates workgroups on a few CUs of each XCD while others idle, and the co-located compute waves (one chain
// each, latency bound) share SIMDs with each other's IO waves -- the forward went 92 -> 139 us between 64 and 256
// workgroups on a 256-CU chip (profiles/r01_h).  Asking for more LDS than a fair share caps the workgroups per CU at
// ceil(total / CUs): one per CU while the grid fits the chip.
inline size_t spread_lds(size_t need, int total_wgs) {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  }
  int target = (total_wgs + ncu - 1) / ncu;
  if (target < 1) target = 1;
  const size_t cap = (size_t)160 * 1024 / (target + 1) + 1024;   // > 1/(target+1) of a CU's 160 KB
  return need > cap ? need : cap;
}
template <typename K>
inline int allow_big_lds(K kernel, const char* what) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) { set_error("%s: cannot raise the dynamic LDS limit: %s", what, hipGetErrorString(e)); return FTR_ERR_LAUNCH; }
  return FTR_OK;
}

struct BidirLayout {
  size_t lat;       // floats per ratio lattice (padded)
  size_t pmid_off, occ_off, seed_off, cxy_off, phi_off, ctrl_off, gran_off, total;   // float offsets
  size_t ctrl_floats, gran_floats;                                   // ctrl block; ONE granule region (forward or flow)
  int NB, Tg;
};
// sized for the regular variant (which needs more granules) whatever `modified` is, so that one workspace serves both
inline BidirLayout bidir_layout(int B, int S, int T) {
  BidirLayout l;
  l.lat = lattice_floats(B, S, T);
  l.pmid_off = kLatPad + 2 * l.lat;
  l.occ_off = l.pmid_off + 2 * (size_t)B * (S + 1);
  l.seed_off = l.occ_off + (size_t)B * (S + 1);
  l.cxy_off = l.seed_off + (size_t)B;                                // shift constants, two per utterance
  l.NB = (S + 1 + 63) / 64;
  l.phi_off = l.cxy_off + 2 * (size_t)B;                             // frame of every band's cut values, int per (direction, utterance, band)
  l.ctrl_off = (l.phi_off + 2 * (size_t)B * l.NB + 3) & ~(size_t)3;
  l.ctrl_floats = ((size_t)4 + 2 * (size_t)B + 3) & ~(size_t)3;     // status + pad, done[B], uflags[B]
  l.gran_off = l.ctrl_off + l.ctrl_floats;
  l.NB = (S + 1 + 63) / 64;
  l.Tg = granules_per_band(T, 0);
  l.gran_floats = 2 * (2 * (size_t)B * l.NB * l.Tg);                 // u64 granules, two directions
  l.total = l.gran_off + 2 * l.gran_floats + 4;
  return l;
}

// The hand-off region (ctrl + granules) sits at the END of the buffer the caller passes, whatever the buffer's size:
// a caller that keeps ONE buffer for many shapes (capacity = the largest data part + the largest hand-off part it has
// seen) zeroes the tail once, and every shape's hand-off region lies inside that tail and is left clean by every launch,
// while the data parts grow from the front and never reach it.  With a buffer of exactly `total` floats this is the
// layout above.  ws_floats == (size_t)-1: size unknown (the entry points without a size), taken as exact.
inline BidirLayout anchored(BidirLayout l, size_t ws_floats) {
  if (ws_floats == (size_t)-1 || ws_floats < l.total) return l;
  const size_t delta = (ws_floats - l.total) & ~(size_t)3;
  l.ctrl_off += delta; l.gran_off += delta; l.total += delta;
  return l;
}
inline size_t handoff_floats(const BidirLayout& l) { return l.total - l.ctrl_off; }
inline size_t status_off(const BidirLayout& l) { return l.total - 4; }   // the last block of the (anchored) hand-off region

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

// floats of workspace the bidirectional kernels need in total
size_t mi_bidir_workspace_floats(int B, int S, int T) { return bidir_layout(B, S, T).total; }
size_t mi_bidir_handoff_floats(int B, int S, int T) { return handoff_floats(bidir_layout(B, S, T)); }

namespace {
int check_ws(const char* what, const float* ws, size_t ws_floats, const BidirLayout& l) {
  if ((reinterpret_cast<uintptr_t>(ws) & 15) != 0) { set_error("%s: the workspace must be 16-byte aligned", what); return FTR_ERR_INVALID_ARG; }
  if (ws_floats < l.total) {
    set_error("%s: the workspace holds %zu floats, ftr_mutual_information_workspace_floats() asks for %zu (it is NOT the "
              "reference's [B,S+1,T+1] temp)", what, ws_floats, l.total);
    return FTR_ERR_INVALID_ARG;
  }
  return FTR_OK;
}
// zeroes the hand-off region (ctrl + both granule regions)
int clear_handoff(const char* what, float* ws, const BidirLayout& l, hipStream_t st) {
  return zero_words(ws + l.ctrl_off, l.total - l.ctrl_off, st, what);   // a kernel, not a memset node: see ftr_common.h
}
}  // namespace

int mi_bidir_ws_init(float* ws, size_t ws_floats, int B, int S, int T, hipStream_t st) {
  const BidirLayout l = anchored(bidir_layout(B, S, T), ws_floats);
  int rc = check_ws("mutual_information_workspace_init", ws, ws_floats, l);
  if (rc != FTR_OK) return rc;
  return clear_handoff("mutual_information_workspace_init", ws, l, st);
}

int mi_bidir_status(const float* ws, size_t ws_floats, int B, int S, int T, int* status_host, long long* dirty_host,
                    hipStream_t st) {
  const BidirLayout l = anchored(bidir_layout(B, S, T), ws_floats);
  int rc = check_ws("mutual_information_status", ws, ws_floats, l);
  if (rc != FTR_OK) return rc;
  if (hipMemcpyAsync(status_host, ws + status_off(l), sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    (void)hipGetLastError(); set_error("mutual_information_status: copy failed"); return FTR_ERR_LAUNCH;
  }
  if (dirty_host) {   // diagnostic: non-zero words of the hand-off region apart from the status word (must be 0 between launches)
    const size_t n = l.total - l.ctrl_off - 4;   // everything but the status block
    unsigned* h = static_cast<unsigned*>(malloc(n * sizeof(unsigned)));
    if (!h || hipMemcpyAsync(h, ws + l.ctrl_off, n * sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
      free(h); (void)hipGetLastError(); set_error("mutual_information_status: copy failed"); return FTR_ERR_LAUNCH;
    }
    long long cnt = 0;
    for (size_t i = 0; i < n; ++i) cnt += (h[i] != 0);
    free(h);
    *dirty_host = cnt;
  }
  return FTR_OK;
}

int mi_bidir_fwd(const float* px, const float* py, const int32_t* boundary, float* ws, size_t ws_floats, int flags,
                 float* ans, int B, int S, int T, int modified, hipStream_t st) {
  const BidirLayout l = anchored(bidir_layout(B, S, T), ws_floats);
  int rc = check_ws("mutual_information_fwd", ws, ws_floats, l);
  if (rc != FTR_OK) return rc;
  if (l.Tg / CH >= (int)kTagChunkMask) { set_error("mutual_information_fwd: T = %d is beyond the %u chunks a granule tag can number", T, kTagChunkMask); return FTR_ERR_UNSUPPORTED; }
  if (!(flags & FTR_MI_WS_CLEAN)) { rc = clear_handoff("mutual_information_fwd", ws, l, st); if (rc != FTR_OK) return rc; }
  u64* gran = reinterpret_cast<u64*>(ws + l.gran_off);
  int* ctrl = reinterpret_cast<int*>(ws + l.ctrl_off);
  const dim3 grid(2 * B * l.NB);
  const size_t lds = spread_lds(bidir_lds_bytes(), (int)grid.x);
  static bool big_ok = false;
  if (!big_ok) {
    rc = allow_big_lds(mi_bidir_fwd_kernel<true>, "mi_bidir_fwd");
    if (rc == FTR_OK) rc = allow_big_lds(mi_bidir_fwd_kernel<false>, "mi_bidir_fwd");
    if (rc != FTR_OK) return rc;
    big_ok = true;
  }
  if (modified) hipLaunchKernelGGL(mi_bidir_fwd_kernel<true>, grid, dim3(256), lds, st, px, py, boundary, ws, gran, ws + l.pmid_off, ws + l.occ_off, ws + l.cxy_off, reinterpret_cast<int*>(ws + l.phi_off), ctrl, (int)(status_off(l) - l.ctrl_off), ans, B, l.NB, l.Tg, S, T);
  else hipLaunchKernelGGL(mi_bidir_fwd_kernel<false>, grid, dim3(256), lds, st, px, py, boundary, ws, gran, ws + l.pmid_off, ws + l.occ_off, ws + l.cxy_off, reinterpret_cast<int*>(ws + l.phi_off), ctrl, (int)(status_off(l) - l.ctrl_off), ans, B, l.NB, l.Tg, S, T);
  return check_launch("mi_bidir_fwd");
}

int mi_bidir_bwd(const int32_t* boundary, const float* ws, size_t ws_floats, int flags, float* px_grad, float* py_grad,
                 float* ans_grad, int overwrite, int B, int S, int T, int modified, hipStream_t st, const float* ans,
                 float* loss_out, int loss_code) {
  const BidirLayout l = anchored(bidir_layout(B, S, T), ws_floats);
  int rc = check_ws("mutual_information_bwd", ws, ws_floats, l);
  if (rc != FTR_OK) return rc;
  float* wsm = const_cast<float*>(ws);   // seed snapshot, ctrl and the flow granules are scratch
  // no memset here: the forward launch of this workspace zeroed (or found clean) the whole hand-off region, and has
  // left ctrl and its own granule region clean again; the flow launch uses its own granule region
  (void)flags;
  u64* gran = reinterpret_cast<u64*>(wsm + l.gran_off + l.gran_floats);
  // the seed is read by every workgroup of an utterance and the self check is written by one of them: the launch reads
  // a snapshot so that it never reads what it writes
  const float* seed = ans_grad;
  float* check = nullptr;
  if (ans_grad && overwrite) {
    rc = copy_words(wsm + l.seed_off, ans_grad, (size_t)B, st, "mi_bidir_bwd: seed snapshot");
    if (rc != FTR_OK) return rc;
    seed = wsm + l.seed_off;
    check = ans_grad;
  }
  int* ctrl = reinterpret_cast<int*>(wsm + l.ctrl_off);
  const dim3 grid(2 * B * l.NB);
  const size_t lds = spread_lds(bidir_lds_bytes(), (int)grid.x);
  static bool big_ok = false;
  if (!big_ok) {
    rc = allow_big_lds(mi_bidir_flow_kernel<true>, "mi_bidir_bwd");
    if (rc == FTR_OK) rc = allow_big_lds(mi_bidir_flow_kernel<false>, "mi_bidir_bwd");
    if (rc != FTR_OK) return rc;
    big_ok = true;
  }
  if (modified) hipLaunchKernelGGL(mi_bidir_flow_kernel<true>, grid, dim3(kFlowThreads), lds, st, boundary, ws, gran, ws + l.occ_off, px_grad, py_grad, seed, check, ctrl + (status_off(l) - l.ctrl_off), ans, loss_out, loss_code, B, l.NB, l.Tg, S, T);
  else hipLaunchKernelGGL(mi_bidir_flow_kernel<false>, grid, dim3(kFlowThreads), lds, st, boundary, ws, gran, ws + l.occ_off, px_grad, py_grad, seed, check, ctrl + (status_off(l) - l.ctrl_off), ans, loss_out, loss_code, B, l.NB, l.Tg, S, T);
  return check_launch("mi_bidir_bwd");
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

// diagnostic (-DFTR_TRACE builds): reads the timeline (n <= kTraceN words) and re-arms it (min slot = ~0, the rest 0)
int debug_trace(unsigned long long* out, int n) {
  static unsigned long long init[kTraceN];
  if (n < 0 || n > kTraceN) { set_error("debug_trace: n out of range"); return FTR_ERR_INVALID_ARG; }
  if (hipDeviceSynchronize() != hipSuccess ||
      (n > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * n) != hipSuccess)) {
    (void)hipGetLastError(); set_error("debug_trace: copy failed"); return FTR_ERR_LAUNCH;
  }
  for (int i = 0; i < kTraceN; ++i) init[i] = 0;
  init[0] = ~0ull;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace), init, sizeof(init)) != hipSuccess) {
    (void)hipGetLastError(); set_error("debug_trace: reset failed"); return FTR_ERR_LAUNCH;
  }
  return FTR_OK;
}

// diagnostic (make STAMPS=1): see FTR_SYNC above
int debug_stamps(unsigned long long* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) {
    set_error("debug_stamps: hipMemcpyFromSymbol failed"); return FTR_ERR_LAUNCH;
  }
  return FTR_OK;
}

}  // namespace ftr
