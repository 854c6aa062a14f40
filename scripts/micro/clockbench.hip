// scripts/micro/clockbench.hip -- diagnostic microbenchmark (not part of the product): does the speed of ONE wave's
// dependent v_exp_f32 / v_log_f32 chain depend on how many other workgroups run the same chain elsewhere on the chip?
// Each workgroup = W waves (one chain each); grids of 16..1024 workgroups; reports the kernel time and the per-step
// cost in s_memtime ticks and in s_memrealtime (100 MHz) ticks, i.e. the effective core clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void chain(float* out, unsigned long long* ticks, int steps, int mode) {
  float p = 0.001f * threadIdx.x, q = 0.5f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  if (mode == 0) {
    for (int i = 0; i < steps; ++i) {   // the forward recursion's chain: sub, add, exp, add, log, add
      const float d = (q - p) + 0.25f;
      const float ex = __builtin_amdgcn_exp2f(-__builtin_fabsf(d));
      p = fmaxf(p, q) + __builtin_amdgcn_logf(1.0f + ex);
      q = p * 0.999f;
    }
  } else {
    for (int i = 0; i < steps; ++i) {   // the flow pass's chain: mul, sub, add
      const float x = p * 0.7f;
      q = p - x;
      p = x + q * 0.999f + 0.001f;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = p + q;
  if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  float* out; unsigned long long* ticks;
  CK(hipMalloc(&out, 4096 * 256 * sizeof(float)));
  CK(hipMalloc(&ticks, 2 * 4096 * sizeof(unsigned long long)));
  unsigned long long h[2 * 4096];
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int steps = 20000;
  for (int mode = 0; mode < 2; ++mode)
    for (int waves = 1; waves <= 4; waves *= 2)
      for (int grid = 16; grid <= 2048; grid *= 2) {
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(chain, dim3(grid), dim3(64 * waves), 0, 0, out, ticks, steps, mode);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, ticks, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
        double st = 0, sr = 0;
        for (int i = 0; i < grid; ++i) { st += h[2 * i]; sr += h[2 * i + 1]; }
        st /= grid; sr /= grid;
        printf("mode %d waves/wg %d grid %5d: kernel %8.1f us  %6.1f ns/step  memtime %6.1f ticks/step  realtime %6.2f ticks/step (100 MHz) -> memtime clock %.0f MHz\n",
               mode, waves, grid, ms * 1e3, ms * 1e6 / steps, st / steps, sr / steps, st / sr * 100.0);
      }
  return 0;
}
