// scripts/micro/slotbench.hip -- diagnostic microbenchmark (not part of the product): what does one "slot" of a
// 4-wave workgroup cost when the waves do almost nothing?  Modes:
//   0  barrier only
//   1  + every wave one ds_write_b128 / ds_read_b128 pair per slot
//   2  + wave 2 waits for a relaxed agent-scope (sc1) 8-byte load it issued one slot earlier (the COMM wave's peek)
//   3  + wave 1 waits for four plain 16-byte loads issued 3 slots earlier (the IO-in wave), wave 3 issues four 16-byte stores
//   4  mode 3 with wave 0 running a dependent 48-op VALU chain per slot
//   5  barrier replaced by LDS flag hand-shake between wave 0 and wave 1 only (2 waves)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

__global__ __launch_bounds__(256) void slots(float* buf, u64* gran, float* out, u64* ticks, int nslots, int mode, size_t stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f4* lds = reinterpret_cast<f4*>(smem);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* mybuf = buf + (size_t)blockIdx.x * stride;
  f4 acc = {0, 0, 0, 0};
  f4 pre[3][4];
  u64 g = 0;
  float p = 0.001f * lane, q = 0.5f;
  for (int u = 0; u < 3; ++u) for (int m = 0; m < 4; ++m) pre[u][m] = acc;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int k3 = 0; k3 < nslots; k3 += 3) {
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int k = k3 + u;
    if (mode >= 1) {
      lds[wid * 64 + lane] = acc;
      acc += lds[((wid + 1) & 3) * 64 + lane];
    }
    if (mode >= 2 && wid == 2) {
      acc[0] += (float)(unsigned)(g >> 32);
      g = __hip_atomic_load(gran + blockIdx.x * 64 + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (mode >= 3 && wid == 1) {
#pragma unroll
      for (int m = 0; m < 4; ++m) acc += pre[u][m];
#pragma unroll
      for (int m = 0; m < 4; ++m) pre[u][m] = *reinterpret_cast<const f4*>(mybuf + (size_t)(16 * m + (lane >> 2)) * 1004 + 16 * k + 4 * (lane & 3));
    }
    if (mode >= 3 && wid == 3) {
#pragma unroll
      for (int m = 0; m < 4; ++m) *reinterpret_cast<f4*>(mybuf + 65 * 1004 + (size_t)(16 * m + (lane >> 2)) * 1004 + 16 * k + 4 * (lane & 3)) = acc;
    }
    if (mode >= 4 && wid == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float x = p * 0.7f; q = p - x; p = x + q * 0.999f; }
    }
    __syncthreads();
  }
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + p + q + pre[0][0][0] + pre[1][1][1] + pre[2][2][2];
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
  const int nslots = 60;      // 16 * 60 = 960 columns of a 1004-float row
  const size_t stride = (size_t)140 * 1004;
  float *buf, *out; u64 *gran, *ticks;
  CK(hipMalloc(&buf, 1024 * stride * sizeof(float)));
  CK(hipMemset(buf, 0, 1024 * stride * sizeof(float)));
  CK(hipMalloc(&out, 1024 * 256 * sizeof(float)));
  CK(hipMalloc(&gran, 1024 * 64 * sizeof(u64)));
  CK(hipMemset(gran, 0, 1024 * 64 * sizeof(u64)));
  CK(hipMalloc(&ticks, 1024 * sizeof(u64)));
  u64 h[1024];
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(slots), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int lds_kb = 8; lds_kb <= 96; lds_kb += 88)
    for (int mode = 0; mode <= 4; ++mode)
      for (int grid = 64; grid <= 1024; grid *= 4) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(slots, dim3(grid), dim3(256), lds_kb * 1024, 0, buf, gran, out, ticks, nslots, mode, stride);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpy(h, ticks, sizeof(u64) * grid, hipMemcpyDeviceToHost));
        double st = 0; for (int i = 0; i < grid; ++i) st += h[i];
        printf("lds %3d KB mode %d grid %5d: kernel %7.1f us  %7.1f ns/slot  in-kernel %7.0f ticks/slot\n", lds_kb, mode, grid, ms * 1e3, ms * 1e6 / nslots, st / grid / nslots);
      }
  return 0;
}
