// csrc/mi_wave_common.h -- constants and helpers of the wavefront kernels (mi_wave_bidir.hip).
#pragma once
#include "ftr_common.h"

namespace ftr {
namespace wavecfg {

constexpr int CH = 16;              // steps per chunk
constexpr int NQ = CH / 4;          // quads (4 consecutive steps) per chunk
constexpr int PLANE = 66;           // float4 per [quad] plane: 64 rows + 2 pad (conflict-free fill+read)
constexpr int TILE_F4 = NQ * PLANE; // one tile = 264 float4 = 4224 B
constexpr int kVmcnt0 = 0x0F70;     // s_waitcnt immediate: vmcnt(0), expcnt/lgkmcnt untouched (gfx9 encoding)
constexpr int RINGN = 64;

// Diagnostic build only (make STAMPS=1): per-segment s_memtime sums of the steady-state slot of wave 0 of
// workgroup 0, read back through ftr_debug_stamps().  Never compiled into the product library.
static __device__ unsigned long long g_stamps[16];  // read back by debug_stamps()
#ifdef FTR_STAMPS
#define FTR_STAMP(var)                                                                            \
  do {                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");                   \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  } while (0)
#else
#define FTR_STAMP(var) do { } while (0)
#endif

// Diagnostic build only (-DFTR_TRACE=1 forward / =2 flow): a timeline in s_memrealtime ticks (100 MHz):
// g_trace[0] = earliest workgroup start, [1] = latest workgroup end (all workgroups, atomics), [2] / [3] = start / end of
// the traced workgroup (utterance 0, direction FTR_TRACE_DIR, band FTR_STAMP_BAND), [4] = slots recorded,
// [16 + k] = time at which the traced workgroup's compute wave left the barrier of slot k.  Reset + read: ftr_debug_trace().
constexpr int kTraceN = 1024;
static __device__ unsigned long long g_trace[kTraceN];
#ifdef FTR_TRACE
#ifndef FTR_TRACE_DIR
#define FTR_TRACE_DIR 0
#endif
__device__ __forceinline__ unsigned long long trace_now() { return __builtin_amdgcn_s_memrealtime(); }
#endif

__device__ __forceinline__ float dpp_wave_shr1(float old_for_lane0, float src) {
  // lane l (l >= 1) receives src of lane l-1; lane 0 keeps `old_for_lane0` (bound_ctrl = 0).
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old_for_lane0),
                                                               __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, false));
}


}  // namespace wavecfg
}  // namespace ftr
