// csrc/simple_fused.hip -- get_rnnt_logprobs / get_rnnt_logprobs_smoothed with the normaliser contraction INSIDE the
// kernel (gfx950, f32 MFMA): /root/reference/tf_fast_rnnt/python/tf_fast_rnnt/rnnt_loss.py:180-221 (simple),
// :1270-1365 (smoothed), :305-321 (delay penalty).
//
//   normalizers[b,s,t] = log(sum_c lm_probs[b,s,c] * am_probs[b,t,c] + tiny) + lm_max[b,s] + am_max[b,t]
//   px[b,s,t] = am[b,t,sym(b,s)] + lm[b,s,sym(b,s)] - normalizers[b,s,t]      (-inf in column T / t_end, + penalty)
//   py[b,s,t] = am[b,t,blank]    + lm[b,s,blank]    - normalizers[b,s,t]
//
// The reference runs this as a batched matmul that writes `normalizers` [B,S+1,T] to memory, followed by ~10 framework
// ops over lattices.  Here one workgroup owns 64 frames x (16 NS) symbol rows of one utterance: the [t, s] tile of the
// product is accumulated in registers with v_mfma_f32_16x16x4_f32 (arithmetic stays f32, operands from LDS, K staged 32
// columns at a time, double buffered), and the epilogue turns the accumulators straight into px and py -- the product
// itself leaves the kernel only when the caller asks for it (the backward's W still wants it).
//
// MFMA operand roles: A = am_probs tile (M = frames), B = lm_probs tile (N = symbol rows), so that an accumulator
// register quad holds four CONSECUTIVE FRAMES of one symbol row: px / py / prod rows are written 16 bytes per lane.
#include "ftr_common.h"
#include <cstdlib>

namespace ftr {
namespace {

constexpr float kTinyF = 1.401298464324817e-45f;  // tf.math.nextafter(0., 1.)  (rnnt_loss.py:181)
// frames per workgroup: 64 MB -- 4 waves x MB 16-frame MFMA blocks (wave w owns frames 16 w + 64 m .. + 15, m < MB);
// 32 columns of C staged per step (128 contiguous bytes of every tile row); LDS row stride = that + 4 floats (16-byte aligned,
// b128 fragment reads conflict free).  MB = 1 double-buffers the tile in LDS (one barrier per step); MB = 2 keeps ONE buffer
// (the tile of 128 + 16 NS rows would not leave room for two workgroups per CU otherwise: 83 KB at NS = 10) and pays a second
// barrier per step, against 16 NS MFMAs per wave and step.  MB = 2 re-reads an utterance's lm_probs rows half as often and
// feeds two MFMAs from every B fragment: it is taken where those rows do not stay in an XCD's L2 anyway (see the launcher).
template <int MB> struct FusedTile {
  static constexpr int FT = 64 * MB, FK = 32, FLD = FK + 4, NBUF = MB == 1 ? 2 : 1;
};
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kGLD = 68;   // row stride (floats) of the epilogue's gather buffer: 64 frames + 4, rows stay 16-byte aligned
template <int NS, int MB>
__host__ __device__ constexpr size_t fused_lds_bytes() {
  constexpr size_t tile = FusedTile<MB>::NBUF * (size_t)(FusedTile<MB>::FT + 16 * NS) * FusedTile<MB>::FLD;
  // epilogue: am[t, sym(s)] of 64 frames x 16 NS rows + the symbols, then the row records (8 floats) and frame records (4)
  constexpr size_t gather = (size_t)16 * NS * kGLD + 16 * NS + (size_t)16 * NS * 8 + (size_t)FusedTile<MB>::FT * 4;
  return sizeof(float) * (tile > gather ? tile : gather);
}

// grid (ceil(T1 / (64 MB)), ceil((S+1) / (16 NS)), B), 256 threads.
template <bool MOD, bool SMOOTH, int NS, int MB>
__global__ __launch_bounds__(256, 2) void simple_fused_fwd_kernel(
    const float* __restrict__ am, const float* __restrict__ lm, const int32_t* __restrict__ symbols,
    const float* __restrict__ am_probs, const float* __restrict__ lm_probs, const float* __restrict__ am_max,
    const float* __restrict__ lm_max, const int32_t* __restrict__ boundary, int blank, double delay_penalty,
    const float* __restrict__ lmonly_norm, const float* __restrict__ amonly_norm, const float* __restrict__ ulog,
    float cs, float ls, float as, float* __restrict__ px, float* __restrict__ py, float* __restrict__ prod_out, int T,
    int S, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [NBUF][ROWS][kFLD]: rows 0..FT-1 frames, then 16 NS symbol rows
  constexpr int kFT = FusedTile<MB>::FT, kFK = FusedTile<MB>::FK, kFLD = FusedTile<MB>::FLD, NBUF = FusedTile<MB>::NBUF;
  constexpr int PPR = kFK / 4;                     // 16-byte pieces per tile row and chunk
  constexpr int ROWS = kFT + 16 * NS;
  constexpr int NU = (ROWS * PPR + 255) / 256;    // 16-byte pieces per thread and chunk
  // XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round robin by linear id, and every XCD has its own L2: in
  // launch order the frame tiles that share one utterance's lm_probs rows (416 KB at c3) are spread over all eight, and each
  // L2 fetches them again (PMC: 327 MB fetched for 77 MB of operands).  Here XCD k works through a contiguous eighth of the
  // tile list instead, in four interleaved streams, so that the ~64 workgroups an XCD runs at a time share four sets of rows:
  // c3 115 -> 106 us, c5 937 -> 920.  Only while four such sets fit the L2 beside the am stream (<= 512 KB each): at c4
  // (852 KB) the same order costs 12 %, as does one stream of 64 workgroups on the same lines.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    const size_t rowset = (size_t)16 * NS * C * sizeof(float);
#ifndef FTR_EXP_FUSED_ORDER_OLD
    // Several symbol tiles per utterance (S + 1 > 16 NS): the tiles of ONE frame block run next to each other on one XCD, so
    // that the frame block's am_probs rows (the A operand) and am rows (the epilogue's gathers) come from memory once instead
    // of once per symbol tile, while the utterance's lm_probs rows -- all of them now -- stay in that L2 (<= 3 MB of its 4).
    if (gridDim.y > 1 && (total & 7u) == 0 && (size_t)(S + 1) * C * sizeof(float) <= 3 * 1024 * 1024) {
      const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const unsigned j = (lin & 7u) * (total >> 3) + (lin >> 3);
      by = j % gridDim.y; bx = (j / gridDim.y) % gridDim.x; bz = j / (gridDim.x * gridDim.y);
    }
    else
#endif
    if ((total & 31u) == 0 && rowset <= 512 * 1024) {
      const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const unsigned per = total >> 3, slot = lin >> 3;
      const unsigned j = (lin & 7u) * per + (slot & 3u) * (per >> 2) + (slot >> 2);
      bx = j % gridDim.x; by = (j / gridDim.x) % gridDim.y; bz = j / (gridDim.x * gridDim.y);
    }
    else if ((total & 15u) == 0 && rowset <= 1024 * 1024) {   // larger row sets (c4 with 128-frame tiles: 655 KB): two streams, 484 -> 476 us
      const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const unsigned per = total >> 3, slot = lin >> 3;
      const unsigned j = (lin & 7u) * per + (slot & 1u) * (per >> 1) + (slot >> 1);
      bx = j % gridDim.x; by = (j / gridDim.x) % gridDim.y; bz = j / (gridDim.x * gridDim.y);
    }
  }
  const int b = bz, t0 = bx * kFT, s0 = by * 16 * NS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T1 = MOD ? T : T + 1;
  const float* amp = am_probs + (size_t)b * T * C;
  const float* lmp = lm_probs + (size_t)b * (S + 1) * C;

  // ---- staging plan: piece e = tid + 256 u is 4 columns (e % PPR) of tile row (e / PPR)
  const float* src[NU];
  bool live[NU];
  int dst[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int e = tid + 256 * u, row = e / PPR, c4 = e % PPR;
    live[u] = false; src[u] = amp; dst[u] = row * kFLD + 4 * c4;
    if (row < kFT) { const int t = t0 + row; if (t < T) { live[u] = true; src[u] = amp + (size_t)t * C + 4 * c4; } }
    else if (row < ROWS) { const int s = s0 + row - kFT; if (s <= S) { live[u] = true; src[u] = lmp + (size_t)s * C + 4 * c4; } }
  }
  auto gload = [&](int kc, f4 (&v)[NU]) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int k = kc * kFK + 4 * ((tid + 256 * u) % PPR);
      v[u] = f4{0.f, 0.f, 0.f, 0.f};
      if (live[u] && k < C) v[u] = *reinterpret_cast<const f4*>(src[u] + kc * kFK);   // C % 4 == 0: k < C covers k + 3
    }
  };
  auto lstore = [&](int buf, const f4 (&v)[NU]) {
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (tid + 256 * u < ROWS * PPR) *reinterpret_cast<f4*>(smem + buf * ROWS * kFLD + dst[u]) = v[u];
  };

  v4f acc[MB][NS];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[m][i] = v4f{0.f, 0.f, 0.f, 0.f};
  const int nk = (C + kFK - 1) / kFK;
  // The epilogue's scalars are fetched HERE, one row (thread < 16 NS) and one frame (thread < kFT) per thread, and ride
  // through the contraction in nine registers: taken in the epilogue they are three dependent memory round trips (symbol ->
  // lm at the symbol; the frame scalars behind the gathers) during which nothing else runs -- 8 - 10 us per workgroup.
  const float* amb = am + (size_t)b * T * C;
  const float* lmb = lm + (size_t)b * (S + 1) * C;
  int r_sym = blank;
  float r_lmblank = 0.f, r_lmx = 0.f, r_lon = 0.f, r_lmsym = 0.f, r_ulog = 0.f, f_amx = 0.f, f_ablank = 0.f, f_aon = 0.f;
  const int r_s = min(s0 + tid, S);                            // rows past S: clamped loads, nothing stored
  if (tid < 16 * NS) {
    if (r_s < S) r_sym = min(max(symbols[(size_t)b * S + r_s], 0), C - 1);   // kept in bounds
    r_lmblank = lmb[(size_t)r_s * C + blank];
    r_lmx = lm_max[(size_t)b * (S + 1) + r_s];
    if (SMOOTH) r_lon = lmonly_norm[(size_t)b * (S + 1) + r_s];
  }
  if (tid < kFT && t0 + tid < T) {
    f_amx = am_max[(size_t)b * T + t0 + tid];
    f_ablank = amb[(size_t)(t0 + tid) * C + blank];
    if (SMOOTH) f_aon = amonly_norm[(size_t)b * T + t0 + tid];
  }
  f4 v[NU];
  gload(0, v);
  lstore(0, v);
  __syncthreads();
  if (tid < 16 * NS) {                                          // the symbol has arrived with the first tile
    r_lmsym = lmb[(size_t)r_s * C + r_sym];
    if (SMOOTH) r_ulog = ulog[r_sym];
  }
  const int frag = (lane & 15) * kFLD + 4 * (lane >> 4);   // this lane's row and its 4 of every 16 columns
#if defined(FTR_FUSED_EXP) && FTR_FUSED_EXP >= 2   // diagnostic: no contraction
  for (int kc = 0; kc < 0; ++kc) {
#else
  for (int kc = 0; kc < nk; ++kc) {
#endif
    if (kc + 1 < nk) gload(kc + 1, v);
    const int cur = NBUF == 2 ? (kc & 1) : 0;
    const float* A = smem + cur * ROWS * kFLD + 16 * wave * kFLD + frag;
    const float* Bm = smem + cur * ROWS * kFLD + kFT * kFLD + frag;
    // One unit = the four MFMAs of (column group g, symbol block i): 2 NS units per chunk, taken in pairs that alternate
    // between two accumulators; the B fragments run through a small register ring fetched kAheadU units ahead of their
    // use (fetching a whole chunk's fragments up front makes the compiler recycle registers and wait on every reuse).
    constexpr int NUNIT = (kFK / 16) * NS;
    constexpr int kAheadU = 4, RING = 8;
    f4 af[MB][kFK / 16], ring[RING];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int g = 0; g < kFK / 16; ++g) af[m][g] = *reinterpret_cast<const f4*>(A + 64 * m * kFLD + 16 * g);
    auto bfetch = [&](int u) { return *reinterpret_cast<const f4*>(Bm + (u % NS) * 16 * kFLD + 16 * (u / NS)); };
#pragma unroll
    for (int u = 0; u < kAheadU; ++u) if (u < NUNIT) ring[u % RING] = bfetch(u);
#pragma unroll
    for (int u = 0; u < NUNIT; u += 2) {
      if (u + kAheadU < NUNIT) ring[(u + kAheadU) % RING] = bfetch(u + kAheadU);
      if (u + kAheadU + 1 < NUNIT) ring[(u + kAheadU + 1) % RING] = bfetch(u + kAheadU + 1);
      const int g0 = u / NS, i0 = u % NS;
      const f4 b0 = ring[u % RING];
      if (u + 1 < NUNIT) {
        const int g1 = (u + 1) / NS, i1 = (u + 1) % NS;
        const f4 b1 = ring[(u + 1) % RING];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int m = 0; m < MB; ++m) {
            acc[m][i0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][g0][e], b0[e], acc[m][i0], 0, 0, 0);
            acc[m][i1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][g1][e], b1[e], acc[m][i1], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int m = 0; m < MB; ++m) acc[m][i0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][g0][e], b0[e], acc[m][i0], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the fetch-ahead order: the scheduler otherwise regroups the units and waits on fresh reads
    }
    if (NBUF == 1) __syncthreads();   // one buffer: every wave has read its fragments before the next chunk overwrites them
    if (kc + 1 < nk) lstore(NBUF == 2 ? ((kc + 1) & 1) : 0, v);
    __syncthreads();
  }

#if defined(FTR_FUSED_EXP) && FTR_FUSED_EXP == 1   // diagnostic: no epilogue (accumulators summed into one store)
  { float t = 0.f; for (int m = 0; m < MB; ++m) for (int i = 0; i < NS; ++i) t += acc[m][i][0] + acc[m][i][1] + acc[m][i][2] + acc[m][i][3]; if (t == 123.456f) py[0] = t; return; }
#endif
  // ---- epilogue: lane (n = lane & 15, q = lane >> 4) holds, per frame block m and symbol block i, frames tq .. tq+3 of row
  // s0 + 16 i + n, tq = t0 + 64 m + 16 wave + 4 q
  const int n = lane & 15;
  const int te = boundary ? boundary[4 * b + 3] : T;
  const float ulog_blank = SMOOTH ? ulog[blank] : 0.0f;
  // am[t, sym(s)] for the 64 x 16 NS cells of a frame block: taken lane by lane in accumulator layout these are 4 NS gathers
  // per lane whose 64 addresses per instruction lie in 64 different lines (4 frames x 16 symbols) -- 105 of the kernel's 540 us
  // at c4, 23 of 105 at c3 (scripts/fused_split.sh).  Instead the workgroup gathers FRAME by FRAME into the tile's LDS (free
  // now): one instruction covers 64 symbols of ONE am row (<= 32 lines, the next instruction of the frame hits the same ones),
  // every line of the am tile comes from L2 once, and each lane then reads its four consecutive frames with one ds_read_b128.
  float* asym = smem;                                             // [16 NS][kGLD]
  int* sym_l = reinterpret_cast<int*>(smem + 16 * NS * kGLD);     // [16 NS]
  float* rrec = smem + 16 * NS * kGLD + 16 * NS;                  // [16 NS][8]: lm at blank, lm_max, lm at the symbol, - | lmonly_norm, ulog at the symbol
  float* frec = rrec + 16 * NS * 8;                               // [kFT][4]: am_max, am at blank, amonly_norm
  if (tid < 16 * NS) {
    sym_l[tid] = r_sym;
    *reinterpret_cast<f4*>(rrec + 8 * tid) = f4{r_lmblank, r_lmx, r_lmsym, 0.f};
    if (SMOOTH) *reinterpret_cast<f4*>(rrec + 8 * tid + 4) = f4{r_lon, r_ulog, 0.f, 0.f};
  }
  if (tid < kFT) *reinterpret_cast<f4*>(frec + 4 * tid) = f4{f_amx, f_ablank, f_aon, 0.f};
  __syncthreads();
  float lm_blank[NS], lmx[NS], lon[NS], lm_sym[NS], ulog_sym[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const f4 ra = *reinterpret_cast<const f4*>(rrec + 8 * (16 * i + n));
    lm_blank[i] = ra[0]; lmx[i] = ra[1]; lm_sym[i] = ra[2];
    lon[i] = 0.f; ulog_sym[i] = 0.f;
    if (SMOOTH) { const f4 rb = *reinterpret_cast<const f4*>(rrec + 8 * (16 * i + n) + 4); lon[i] = rb[0]; ulog_sym[i] = rb[1]; }
  }
  constexpr int NSL = (16 * NS + 63) / 64;
  int mysym[NSL];
#pragma unroll
  for (int u = 0; u < NSL; ++u) mysym[u] = sym_l[min(lane + 64 * u, 16 * NS - 1)];
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    if (m > 0) __syncthreads();                                   // the previous block's values have been read
#if !(defined(FTR_FUSED_EXP) && (FTR_FUSED_EXP == 3 || FTR_FUSED_EXP == 5))
    {                                                             // 16 frames per wave, NSL gathers each, GB frames' worth in
      constexpr int GB = (MB == 2 && NS > 10) ? 8 : 16;                        // flight together (all 16 where the registers allow): taken four
#pragma unroll                                                    // frames at a time the phase is four memory round trips long
      for (int k0 = 0; k0 < 16; k0 += GB) {                       // (17 of a workgroup's 40 us at c5)
        float gv[GB][NSL];
#pragma unroll
        for (int k = 0; k < GB; ++k) {
          const float* row = amb + (size_t)min(t0 + 64 * m + wave + 4 * (k0 + k), T - 1) * C;
#pragma unroll
          for (int u = 0; u < NSL; ++u) gv[k][u] = row[mysym[u]];
        }
#pragma unroll
        for (int k = 0; k < GB; ++k)
#pragma unroll
          for (int u = 0; u < NSL; ++u) {
            const int sl = lane + 64 * u;
            if (sl < 16 * NS) asym[sl * kGLD + wave + 4 * (k0 + k)] = gv[k][u];
          }
      }
    }
#endif
    __syncthreads();
    const int tq = t0 + 64 * m + 16 * wave + 4 * (lane >> 4);
    if (tq >= T1) continue;                                       // (no barrier below this point inside the block)
    float amx[4], a_blank[4], aon[4], pen[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = tq + j;
      const f4 fr = *reinterpret_cast<const f4*>(frec + 4 * (64 * m + 16 * wave + 4 * (lane >> 4) + j));   // zeros past T
      amx[j] = fr[0]; a_blank[j] = fr[1]; aon[j] = fr[2];
      pen[j] = (delay_penalty > 0.0) ? (float)((((double)te - 1.0) / 2.0 - (double)t) * delay_penalty) : 0.0f;   // :305-321
    }
    f4 a_sym[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) a_sym[i] = *reinterpret_cast<const f4*>(asym + (16 * i + n) * kGLD + 16 * wave + 4 * (lane >> 4));
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int s = s0 + 16 * i + n;
      if (s > S) continue;
      f4 vy, vx, vp;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = tq + j;
        const float pr = acc[m][i][j];
        vp[j] = pr;
        const float nrm = logf(pr + kTinyF) + lmx[i] + amx[j];                                                    // :180-186
        float y = a_blank[j] + lm_blank[i] - nrm;                                                                 // :214-216
        if (SMOOTH) y = y * cs + (lm_blank[i] - lon[i]) * ls + (a_blank[j] + ulog_blank - aon[j]) * as;           // :1333-1360
        vy[j] = y;
        float x = -INFINITY;                                                  // px[:, :, T] (:193-203), fix_for_boundary (:218-219)
        if (s < S && t < T && (MOD || t != te)) {
          x = a_sym[i][j] + lm_sym[i] - nrm;                                                                      // :187-211
          if (SMOOTH) x = x * cs + (lm_sym[i] - lon[i]) * ls + (a_sym[i][j] + ulog_sym[i] - aon[j]) * as;         // :1323-1355
        }
        if (delay_penalty > 0.0) x += pen[j];
        vx[j] = x;
      }
#if defined(FTR_FUSED_EXP) && (FTR_FUSED_EXP == 4 || FTR_FUSED_EXP == 5)    // diagnostic: one store per WG instead of all of them
      if (vy[0] + vx[1] + vp[2] != 123.456f) continue;
#endif
      float* yrow = py + ((size_t)b * (S + 1) + s) * T + tq;
      if (tq + 3 < T) *reinterpret_cast<f4u*>(yrow) = vy;
      else { for (int j = 0; j < 4; ++j) if (tq + j < T) yrow[j] = vy[j]; }
      if (prod_out) {
        float* prow = prod_out + ((size_t)b * (S + 1) + s) * T + tq;
        if (tq + 3 < T) *reinterpret_cast<f4u*>(prow) = vp;
        else { for (int j = 0; j < 4; ++j) if (tq + j < T) prow[j] = vp[j]; }
      }
      if (s < S) {
        float* xrow = px + ((size_t)b * S + s) * T1 + tq;
        if (tq + 3 < T1) *reinterpret_cast<f4u*>(xrow) = vx;
        else { for (int j = 0; j < 4; ++j) if (tq + j < T1) xrow[j] = vx[j]; }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------- backward: d am
// d am[b,t,c] = am_probs[b,t,c] * sum_s W[b,s,t] lm_probs[b,s,c]  +  kdir * (sum_{s: sym(s) = c} gx[b,s,t] + [c = blank] sum_s gy[b,s,t])
//               (+ smoothed: am_probs[b,t,c] * u[c] * R[b,t],  R = -as * colsum_s(gx + gy) / (am_probs . u))
// with W = -cs (gx + gy) / (prod + tiny), gx = g_px masked where the forward wrote -inf, both times the upstream scale
// (what TF autodiff replays for rnnt_loss.py:180-221 / :1296-1365 towards am).  The unfused route runs this as W kernel ->
// library GEMM (damp [B,T,C] through memory) -> scatter / epilogue kernel.  Here a workgroup owns 64 frames x 256 columns
// (two workgroups per CU, like the forward kernel: one's epilogue and loads overlap the other's MFMAs) and walks the symbol
// rows twice with the SAME accumulators:
//   pass 1  acc[c,t] += lm_probs[s,c] * W[s,t]          (MFMA; W formed from g_px, g_py, prod while the chunk is staged)
//           acc = am_probs * (acc + u R)                (column sums for R and the blank term were taken while staging)
//   pass 2  acc[c,t] += onehot(sym(s) = c) * kdir gx[s,t]   the scatter by symbol as a second small MFMA contraction (exact
//                                                        0/1 products, deterministic, no scatter tile, no atomics)
// Every load of the staging is UNCONDITIONAL: addresses are clamped into the arrays and what was loaded out of range is
// multiplied away (a load under a condition costs an exec-masked branch each and makes the compiler wait for all of them;
// the first versions of this kernel -- 16-byte loads under row / column / frame conditions, 512 columns per workgroup at one
// workgroup per CU -- ran at a quarter of the matrix rate: 165 - 196 us at c3 against 63 + 61 for the tuned library GEMM
// plus the epilogue kernel).  Needs T % 4 == 0 and C % 4 == 0 (other shapes take the library route).
// MFMA roles: M = columns, N = frames.  Column blocks are INTERLEAVED: block i = 4 g + e holds the local columns
// 64 g + 4 m + e (m = 0..15), so that ONE 16-byte LDS read -- columns 64 g + 4 fn .. + 3 of a k row -- is the A operand of the
// four blocks of group g, and the accumulators of a group, acc[4g + 0..3][j], are four CONSECUTIVE columns
// 64 g + 16 fk + 4 j + (0..3) of frame fn: am_probs is read and d am written 16 bytes at a time.
constexpr int kBT = 64;                           // frames per workgroup
constexpr int kBS = 16;                           // symbol rows per staged chunk (four MFMA k-steps)
constexpr int kBLT = kBT + 4;                     // W tile row stride (floats)
constexpr int kBCB = 16;                          // column blocks per workgroup (256 columns)

__host__ __device__ constexpr size_t fused_bwd_lds_bytes() {
  return sizeof(float) * (2 * kBS * (16 * kBCB) + 2 * kBS * kBLT + 2 * kBS * kBT);
}

template <bool MOD>
__global__ __launch_bounds__(256, 2) void simple_fused_bwd_am_kernel(
    const float* __restrict__ gpx, const float* __restrict__ gpy, const Scale scale, const float* __restrict__ prod,
    const float* __restrict__ lm_probs, const float* __restrict__ am_probs, const int32_t* __restrict__ symbols,
    const int32_t* __restrict__ boundary, int blank, float cs, float kdir, const float* __restrict__ uvec,
    const float* __restrict__ amdot, float as, float* __restrict__ Rout, float* __restrict__ d_am, int T, int S, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // lm tile rows are CT floats apart with no padding: a fragment read is 16 bytes per lane, 16 consecutive lanes = 256
  // contiguous bytes of one k row, and rows a multiple of 256 bytes apart keep the four rows of a k-step on disjoint banks
  constexpr int NCB = kBCB, CT = 16 * NCB, LDC = CT, NG = NCB / 4;
  float* lmT = smem;                               // [2][kBS][LDC]   lm_probs rows of the chunk
  float* wT = lmT + 2 * kBS * LDC;                 // [2][kBS][kBLT]  W (pass 1) / kdir * gx (pass 2) rows of the chunk
  float* csb = wT + 2 * kBS * kBLT;                // [2][kBS][kBT]   partial column sums (x, y) per staging row class
  // XCD-aware tile order (as in the forward kernel): workgroups are dealt to the eight XCDs round robin by linear id and
  // every XCD has its own L2; in launch order the tiles that share one utterance's lm_probs rows and one frame tile's
  // g_px / g_py / prod columns land on eight different L2s.  Here XCD k works through a contiguous eighth of the tile list.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    if ((total & 7u) == 0) {
      const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const unsigned j = (lin & 7u) * (total >> 3) + (lin >> 3);
      by = j % gridDim.y; bx = (j / gridDim.y) % gridDim.x; bz = j / (gridDim.x * gridDim.y);   // column groups of a frame tile adjacent
    }
  }
  const int b = bz, t0 = bx * kBT, c0 = by * CT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T1 = MOD ? T : T + 1;
  const int te = boundary ? boundary[4 * b + 3] : T;
  const float sc = scale.at(b);
  const float* gxb = gpx + (size_t)b * S * T1;
  const float* gyb = gpy + (size_t)b * (S + 1) * T;
  const float* prb = prod + (size_t)b * (S + 1) * T;
  const float* lmb = lm_probs + (size_t)b * (S + 1) * C;
  const int nk = (S + 1 + kBS - 1) / kBS;

  // ---- staging plan.  lm tile: piece e = tid + 256 u (u < NL) = 4 columns of one of the chunk's rows (column clamped).
  constexpr int NL = (kBS * CT / 4) / 256;         // 4
  int lrow[NL], lcol[NL];
  float lmask[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int e = tid + 256 * u, c = c0 + 4 * (e % (CT / 4));
    lrow[u] = e / (CT / 4);
    lcol[u] = min(c, C - 4);
    lmask[u] = c < C ? 1.0f : 0.0f;
  }
  // W / x tile: row tid / 16 of the chunk, frames t0 + 4 (tid % 16) .. +3 (T % 4 == 0: a quad is inside [0, T) or outside)
  const int wrow = tid >> 4, wq = tid & 15, wt = t0 + 4 * wq;
  const int wtc = min(wt, T - 4);
  const float tmask = wt < T ? 1.0f : 0.0f;
  f4 xmask;                                        // g_px counts where the forward wrote a finite px: every frame but t_end (regular)
#pragma unroll
  for (int e = 0; e < 4; ++e) xmask[e] = (wt < T && (MOD || wt + e != te)) ? sc : 0.0f;
  struct Stage { f4 lv[NL]; f4 x, y, pr; };
  auto load = [&](int kc, Stage& g, bool want_all) {
    const int s = kc * kBS + wrow;
    const int sy_ = min(s, S), sx_ = min(s, S > 0 ? S - 1 : 0);
    if (want_all) {
#pragma unroll
      for (int u = 0; u < NL; ++u) g.lv[u] = *reinterpret_cast<const f4*>(lmb + (size_t)min(kc * kBS + lrow[u], S) * C + lcol[u]);
      g.y = *reinterpret_cast<const f4u*>(gyb + (size_t)sy_ * T + wtc);
      g.pr = *reinterpret_cast<const f4u*>(prb + (size_t)sy_ * T + wtc);
    }
    g.x = (S > 0) ? (f4)*reinterpret_cast<const f4u*>(gxb + (size_t)sx_ * T1 + wtc) : f4{0.f, 0.f, 0.f, 0.f};
  };
  f4 sx = {0.f, 0.f, 0.f, 0.f}, sy = {0.f, 0.f, 0.f, 0.f};   // column sums of this thread's row class

  v4f acc[NCB];
#pragma unroll
  for (int i = 0; i < NCB; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
  const int fk = lane >> 4, fn = lane & 15;        // MFMA fragment coordinates of this lane: k row, m / n index

  // =================================================================== pass 1: acc[c,t] += lm_probs[s,c] W[s,t]
  {
    auto park = [&](int kc, const Stage& g) {
      const int buf = kc & 1;
#pragma unroll
      for (int u = 0; u < NL; ++u) {
        const float m = (kc * kBS + lrow[u] <= S) ? lmask[u] : 0.0f;
        *reinterpret_cast<f4*>(lmT + (buf * kBS + lrow[u]) * LDC + 4 * ((tid + 256 * u) % (CT / 4))) = g.lv[u] * m;
      }
      const int s = kc * kBS + wrow;
      const float rowy = s <= S ? sc * tmask : 0.0f, rowx = s < S ? 1.0f : 0.0f;
      const f4 x = g.x * xmask * rowx, y = g.y * rowy;
      f4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = -cs * (x[e] + y[e]) / (g.pr[e] + kTinyF);
      sx += x; sy += y;
      *reinterpret_cast<f4*>(wT + (buf * kBS + wrow) * kBLT + 4 * wq) = w;
    };
    auto compute = [&](int kc) {
      const int buf = kc & 1;
      const float* abase = lmT + (buf * kBS + fk) * LDC + 4 * fn;
      const float* bbase = wT + (buf * kBS + fk) * kBLT + 16 * wave + fn;
      f4 af[2][NG];
      float bw[kBS / 4];
#pragma unroll
      for (int kk = 0; kk < kBS / 4; ++kk) bw[kk] = bbase[4 * kk * kBLT];
#pragma unroll
      for (int g = 0; g < NG; ++g) af[0][g] = *reinterpret_cast<const f4*>(abase + 64 * g);
#pragma unroll
      for (int kk = 0; kk < kBS / 4; ++kk) {
        if (kk + 1 < kBS / 4) {
#pragma unroll
          for (int g = 0; g < NG; ++g) af[(kk + 1) & 1][g] = *reinterpret_cast<const f4*>(abase + 4 * (kk + 1) * LDC + 64 * g);
        }
        __builtin_amdgcn_sched_barrier(0);   // the next k-step's fragments are requested BEFORE this k-step's MFMAs: left alone
                                             // the scheduler sinks every read to its first use and waits for it there
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[4 * g + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk & 1][g][e], bw[kk], acc[4 * g + e], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    Stage g;
    load(0, g, true);
    park(0, g);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
#if !(defined(FTR_FUSED_BWD_EXP) && FTR_FUSED_BWD_EXP == 3)   // study build 3: no loads in the loop (the first chunk's operands again)
      if (kc + 1 < nk) load(kc + 1, g, true);
#endif
#if !(defined(FTR_FUSED_BWD_EXP) && FTR_FUSED_BWD_EXP == 2)   // study build 2: no MFMAs in pass 1
      compute(kc);
#endif
      if (kc + 1 < nk) park(kc + 1, g);
      __syncthreads();
    }
  }
  // ---- column sums over s of x and y for this lane's frame (blank term, R), then acc = am_probs * (acc + u R)
  *reinterpret_cast<f4*>(csb + (0 * kBS + wrow) * kBT + 4 * wq) = sx;
  *reinterpret_cast<f4*>(csb + (1 * kBS + wrow) * kBT + 4 * wq) = sy;
  __syncthreads();
  const int tl = 16 * wave + fn, t = t0 + tl;
  float cx = 0.0f, cy = 0.0f;
#pragma unroll
  for (int r8 = 0; r8 < kBS; ++r8) { cx += csb[(0 * kBS + r8) * kBT + tl]; cy += csb[(1 * kBS + r8) * kBT + tl]; }
  const bool tok = t < T;
  float R = 0.0f;
  if (uvec && tok) {
    R = -as * (cx + cy) / amdot[(size_t)b * T + t];
    if (by == 0 && fk == 0) Rout[(size_t)b * T + t] = R;
  }
  const float* aprow = am_probs + ((size_t)b * T + (tok ? t : 0)) * C;
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + 64 * g + 16 * fk + 4 * j;      // acc[4g + e][j] is column c + e
      f4 ap = {0.f, 0.f, 0.f, 0.f};
      if (tok && c < C) ap = *reinterpret_cast<const f4*>(aprow + c);
      if (uvec) {
        f4 uv = {0.f, 0.f, 0.f, 0.f};
        if (c < C) uv = *reinterpret_cast<const f4*>(uvec + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e][j] = ap[e] * (acc[4 * g + e][j] + uv[e] * R);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e][j] = ap[e] * acc[4 * g + e][j];
      }
    }
  // =================================================================== pass 2: acc[c,t] += [sym(s) = c] kdir gx[s,t]
  // Little arithmetic (at most four MFMAs per k-step), so the rows are staged 128 at a time into the idle lm tile area: two
  // round trips to memory at S = 200 instead of thirteen (taken 16 rows at a time, as pass 1 is, this pass was a third of
  // the kernel: every iteration waited for its loads with nothing to overlap them with).
  {
    constexpr int kB2 = 128;                       // rows per stage: kB2 * kBT floats = the lm tile area
    static_assert(kB2 * kBT <= 2 * kBS * LDC && kB2 <= 2 * kBS * kBLT, "pass 2 reuses the tiles of pass 1");
    float* xT = lmT;                               // [kB2][kBT]
    int* sym2 = reinterpret_cast<int*>(wT);        // [kB2]
    // Only the rows whose symbol falls into this workgroup's CT columns contribute -- one row in C / CT -- so they are
    // listed first (ascending, ballot compaction over 256 rows at a time) and only those are fetched and walked: at c4
    // (four column groups) this pass was 114 of the kernel's 648 us with every group walking all S rows.
    constexpr int kCap = (2 * kBS * kBLT - kB2) * 2;                                 // list capacity (16-bit entries behind sym2)
    unsigned short* rlist = reinterpret_cast<unsigned short*>(wT + kB2);
    int* wcnt = reinterpret_cast<int*>(csb);                                         // [4]; the column sums have been read
    int count = 0;
    const bool listed = S <= 65535;
    __syncthreads();                               // pass 1 and the column sums are done with wT / csb
    if (listed) {
      for (int base = 0; base < S; base += 256) {
        const int srow = base + tid;
        const int sy = srow < S ? min(max(symbols[(size_t)b * S + srow], 0), C - 1) : -1;
        const bool mine = sy >= c0 && sy < c0 + CT;
        const unsigned long long m = __ballot(mine);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int off = count;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        const int at = off + __popcll(m & ((1ull << lane) - 1ull));
        if (mine && at < kCap) rlist[at] = (unsigned short)srow;
        count += tot;
        __syncthreads();
      }
    }
    const bool all_rows = !listed || count > kCap;  // (more listed rows than the list holds: every row, as before)
    if (all_rows) count = S;
    auto row_of = [&](int i) { return all_rows ? i : (int)rlist[i]; };
    constexpr int NX = kB2 * (kBT / 4) / 256;      // 8 quads per thread and stage
#if defined(FTR_FUSED_BWD_EXP) && FTR_FUSED_BWD_EXP == 1       // study build 1: no pass 2
    for (int r0 = 0; r0 < 0; r0 += kB2) {
#else
    for (int r0 = 0; r0 < count; r0 += kB2) {
#endif
      f4 xv[NX];
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const int row = row_of(min(r0 + wrow + 16 * u, count - 1));
        xv[u] = *reinterpret_cast<const f4u*>(gxb + (size_t)row * T1 + wtc);
      }
      const int li = r0 + tid;
      const int symr = (tid < kB2 && li < count) ? min(max(symbols[(size_t)b * S + row_of(li)], 0), C - 1) : -1;
      __syncthreads();                             // the previous stage is done with the area
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const float rowx = (r0 + wrow + 16 * u < count) ? kdir : 0.0f;
        *reinterpret_cast<f4*>(xT + (wrow + 16 * u) * kBT + 4 * wq) = xv[u] * xmask * rowx;
      }
      if (tid < kB2) sym2[tid] = symr;
      __syncthreads();
      const int nks = (min(kB2, count - r0) + 3) / 4;
      for (int kk = 0; kk < nks; ++kk) {
        const float bx = xT[(4 * kk + fk) * kBT + 16 * wave + fn];
        // this lane's A element (k row 4 kk + fk, block row fn) is 1 in block 4 g + e iff the row's symbol is local
        // column 64 g + 4 fn + e
        const int lc = sym2[4 * kk + fk] - c0;
        const int myblk = (lc >= 0 && lc < CT && ((lc & 63) >> 2) == fn) ? 4 * (lc >> 6) + (lc & 3) : -1;
        // only the (at most four) column blocks that a symbol of this k-step falls into do any work (wave-uniform mask)
        unsigned mask = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int sb = __builtin_amdgcn_readfirstlane(sym2[4 * kk + q]) - c0;
          if (sb >= 0 && sb < CT) mask |= 1u << (4 * (sb >> 6) + (sb & 3));
        }
#pragma unroll
        for (int i = 0; i < NCB; ++i)
          if (mask & (1u << i)) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32((myblk == i) ? 1.0f : 0.0f, bx, acc[i], 0, 0, 0);
      }
    }
  }
  // ---- blank column, store
  if (!tok) return;
  float* drow = d_am + ((size_t)b * T + t) * C;
  const float colb = kdir * cy;
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + 64 * g + 16 * fk + 4 * j;
      if (c >= C) continue;
      f4 v = {acc[4 * g + 0][j], acc[4 * g + 1][j], acc[4 * g + 2][j], acc[4 * g + 3][j]};
      if (blank >= c && blank < c + 4) v[blank - c] += colb;
      *reinterpret_cast<f4*>(drow + c) = v;
    }
}

}  // namespace

int simple_fused_supported(int C) { return (C % 4 == 0) ? 1 : 0; }

int simple_fused_fwd(const float* am, const float* lm, const int32_t* symbols, const float* am_probs,
                     const float* lm_probs, const float* am_max, const float* lm_max, const int32_t* boundary, int blank,
                     double delay_penalty, const float* lmonly_norm, const float* amonly_norm, const float* ulog, float cs,
                     float ls, float as, float* px, float* py, float* prod_out, int B, int T, int S, int C, int modified,
                     hipStream_t st) {
  if (!simple_fused_supported(C)) { set_error("simple_logprobs_fused_fwd: C = %d is not a multiple of 4", C); return FTR_ERR_UNSUPPORTED; }
  const int T1 = modified ? T : T + 1;
  const int blocks = (S + 1 + 15) / 16;
  const int ny = (blocks + 12) / 13;
  const int need = (blocks + ny - 1) / ny;                     // symbol blocks per workgroup, <= 13
  int ns = need <= 4 ? 4 : need <= 7 ? 7 : need <= 10 ? 10 : 13;
  // a small problem in large tiles leaves CUs with one workgroup or none (c2: 288 workgroups of 112 rows on 256 CUs, one
  // wave per SIMD): smaller symbol tiles until there are two per CU (c2 51 -> 44 us; B = 8, T = 1000, S = 200: 51 -> 34)
  {
    const size_t ftiles = (size_t)((T1 + 63) / 64) * B;
    while (ns > 4 && ftiles * ((blocks + ns - 1) / ns) < 512) ns -= 3;
  }
  if (const char* e = getenv("FTR_FUSED_NS")) { const int v = atoi(e); if (v == 4 || v == 7 || v == 10 || v == 13) ns = v; }   // A/B measurements, tests
  // 128-frame tiles where one utterance's lm_probs rows of a tile (16 ns C floats) exceed what the XCD-aware order keeps in
  // an L2 (> 512 KB: that order is off, every frame tile fetches the rows through the fabric) and there are frames for it;
  // FTR_FUSED_FT = 64 | 128 forces one (A/B measurements)
  int mb = ((size_t)16 * ns * C * sizeof(float) > 512 * 1024 && T1 > 64) ? 2 : 1;
  if (const char* e = getenv("FTR_FUSED_FT")) { const int v = atoi(e); if (v == 64) mb = 1; else if (v == 128) mb = 2; }
  const dim3 grid((T1 + 64 * mb - 1) / (64 * mb), (blocks + ns - 1) / ns, B);
  if (grid.z > 65535) { set_error("simple_logprobs_fused_fwd: B = %d > 65535", B); return FTR_ERR_UNSUPPORTED; }
  const bool smooth = lmonly_norm != nullptr;
#define FTR_FUSED_LAUNCH(MODV, SMV, NSV, MBV)                                                                              \
  do {                                                                                                                     \
    static bool raised = false;                                                                                            \
    if (!raised && fused_lds_bytes<NSV, MBV>() > 64 * 1024) {                                                              \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(simple_fused_fwd_kernel<MODV, SMV, NSV, MBV>),                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds_bytes<NSV, MBV>()) != hipSuccess) { \
        (void)hipGetLastError(); set_error("simple_logprobs_fused_fwd: cannot raise the dynamic LDS limit"); return FTR_ERR_LAUNCH; \
      }                                                                                                                    \
      raised = true;                                                                                                       \
    }                                                                                                                      \
    hipLaunchKernelGGL((simple_fused_fwd_kernel<MODV, SMV, NSV, MBV>), grid, dim3(256), (fused_lds_bytes<NSV, MBV>()), st,   \
                       am, lm, symbols, am_probs, lm_probs, am_max, lm_max, boundary, blank, delay_penalty, lmonly_norm,   \
                       amonly_norm, ulog, cs, ls, as, px, py, prod_out, T, S, C);                                         \
  } while (0)
#define FTR_FUSED_MB(MODV, SMV, NSV)                                                                                       \
  do { if (mb == 2) FTR_FUSED_LAUNCH(MODV, SMV, NSV, 2); else FTR_FUSED_LAUNCH(MODV, SMV, NSV, 1); } while (0)
#define FTR_FUSED_NS(MODV, SMV)                                                                                            \
  do {                                                                                                                     \
    if (ns == 4) FTR_FUSED_MB(MODV, SMV, 4); else if (ns == 7) FTR_FUSED_MB(MODV, SMV, 7);                                  \
    else if (ns == 10) FTR_FUSED_MB(MODV, SMV, 10); else FTR_FUSED_MB(MODV, SMV, 13);                                       \
  } while (0)
  if (modified) { if (smooth) FTR_FUSED_NS(true, true); else FTR_FUSED_NS(true, false); }
  else { if (smooth) FTR_FUSED_NS(false, true); else FTR_FUSED_NS(false, false); }
#undef FTR_FUSED_NS
#undef FTR_FUSED_MB
#undef FTR_FUSED_LAUNCH
  return check_launch("simple_logprobs_fused_fwd");
}

// 0: outside the fused backward kernel's domain (needs C % 4 == 0, T % 4 == 0, C >= 4, T >= 4): the caller then takes
// the library-GEMM route
int simple_fused_bwd_supported(int T, int C) {
  return (simple_fused_supported(C) && C >= 4 && T >= 4 && (T % 4) == 0) ? 1 : 0;
}

int simple_fused_bwd_am(const float* gpx, const float* gpy, Scale scale, const float* prod, const float* lm_probs,
                        const float* am_probs, const int32_t* symbols, const int32_t* boundary, int blank, float cs,
                        float kdir, const float* uvec, const float* amdot, float as, float* Rout, float* d_am, int B,
                        int T, int S, int C, int modified, hipStream_t st) {
  if (!simple_fused_bwd_supported(T, C)) {
    set_error("simple_logprobs_fused_bwd_am: T = %d, C = %d is outside the fused kernel's domain (T %% 4 == 0, C %% 4 == 0)", T, C);
    return FTR_ERR_UNSUPPORTED;
  }
  const dim3 grid((T + kBT - 1) / kBT, (C + 16 * kBCB - 1) / (16 * kBCB), B);
  if (grid.z > 65535) { set_error("simple_logprobs_fused_bwd_am: B = %d > 65535", B); return FTR_ERR_UNSUPPORTED; }
  if (modified) hipLaunchKernelGGL(simple_fused_bwd_am_kernel<true>, grid, dim3(256), fused_bwd_lds_bytes(), st, gpx, gpy, scale, prod, lm_probs, am_probs, symbols, boundary, blank, cs, kdir, uvec, amdot, as, Rout, d_am, T, S, C);
  else hipLaunchKernelGGL(simple_fused_bwd_am_kernel<false>, grid, dim3(256), fused_bwd_lds_bytes(), st, gpx, gpy, scale, prod, lm_probs, am_probs, symbols, boundary, blank, cs, kdir, uvec, amdot, as, Rout, d_am, T, S, C);
  return check_launch("simple_logprobs_fused_bwd_am");
}

}  // namespace ftr
