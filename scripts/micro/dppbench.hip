// scripts/micro/dppbench.hip -- diagnostic microbenchmark (not part of the product): latency of the lane-shift used on
// the recursion's dependent chain.  One wave, a chain  p = shift(p) + c  of 4096 steps, cycles per step for:
//   0 no shift (p = p + c)            1 wave_shr:1 (full-wave shift, gfx9 DPP)       2 row_shr:1 (within rows of 16)
//   3 row_bcast:15 into rows 1..3, then row_shr:1 with the broadcast as `old` (full-wave shift from two row-level DPPs)
//   4 mode 3 fused into the arithmetic (v_add_f32_dpp writing over the plain add of the broadcast)
//   5 ds_bpermute                      6 v_exp_f32 + v_log_f32 chain (no shift)      7 v_exp only    8 v_mul chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;

template <int CTRL, int ROW_MASK, int BANK_MASK, bool BC>
__device__ __forceinline__ float dpp(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, BANK_MASK, BC));
}

template <int MODE>
__global__ void chain(float* out, u64* ticks, int steps) {
  const int lane = threadIdx.x;
  float p = 0.001f * lane;
  const float c = 0.25f;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < steps; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) p = p + c;
      if (MODE == 1) p = dpp<0x138, 0xf, 0xf, false>(-1.0f, p) + c;
      if (MODE == 2) p = dpp<0x111, 0xf, 0xf, false>(-1.0f, p) + c;
      if (MODE == 3) { const float t2 = dpp<0x142, 0xe, 0xf, false>(-1.0f, p); p = dpp<0x111, 0xf, 0xf, false>(t2, p) + c; }
      if (MODE == 4) {
        const float t2 = dpp<0x142, 0xe, 0xf, false>(-1.0f, p);
        float r = t2 + c;
        asm volatile("v_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(p), "v"(c));
        p = r;
      }
      if (MODE == 5) p = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) << 2, __builtin_bit_cast(int, p))) + c;
      if (MODE == 6) p = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-__builtin_fabsf(p))) + c;
      if (MODE == 7) p = __builtin_amdgcn_exp2f(-__builtin_fabsf(p)) + c;
      if (MODE == 8) p = p * 0.999f;
    }
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = p;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE>
void run(float* out, u64* ticks, const char* name) {
  const int steps = 4096;
  u64 h = 0; float ho[64];
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(chain<MODE>, dim3(1), dim3(64), 0, 0, out, ticks, steps); CK(hipDeviceSynchronize()); }
  CK(hipMemcpy(&h, ticks, sizeof(u64), hipMemcpyDeviceToHost));
  CK(hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost));
  printf("mode %d %-44s %6.1f cycles/step   (lane 0: %g lane 1: %g lane 16: %g lane 17: %g lane 63: %g)\n", MODE, name, (double)h / steps, ho[0], ho[1], ho[16], ho[17], ho[63]);
}

int main() {
  float* out; u64* ticks;
  CK(hipMalloc(&out, 64 * sizeof(float))); CK(hipMalloc(&ticks, sizeof(u64)));
  run<0>(out, ticks, "add");
  run<1>(out, ticks, "wave_shr:1 + add");
  run<2>(out, ticks, "row_shr:1 + add");
  run<3>(out, ticks, "row_bcast:15 -> row_shr:1(old) + add");
  run<4>(out, ticks, "row_bcast:15, add, v_add_f32_dpp row_shr:1");
  run<5>(out, ticks, "ds_bpermute + add");
  run<6>(out, ticks, "abs/exp2/add/log/add");
  run<7>(out, ticks, "abs/exp2/add");
  run<8>(out, ticks, "mul");
  return 0;
}
