// scripts/micro/flagbench.hip -- diagnostic microbenchmark (not part of the product): a 4-wave workgroup pipelined
// through LDS flags instead of one s_barrier per slot.  Wave 1 ("IO-in") fills a tile slot (4 ds_write_b128) and bumps
// in_ready; wave 0 ("compute") waits for in_ready > k, reads its tile row, runs a dependent chain of `chain` VALU ops
// per step for 16 steps, writes 2 output tiles (8 ds_write_b128), bumps comp_done; waves 2 and 3 ("COMM", "IO-out")
// wait for comp_done > k, read the output tile (4 ds_read_b128) and bump their drained counters; wave 1 and wave 0
// respect ring capacities (NIN = 4, NOUT = 3).  mode 0 = the same work with one __syncthreads() per slot instead.
// Prints cycles per chunk.  Checks a checksum so that ordering bugs show.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
constexpr int NIN = 4, NOUT = 3, TILE = 264;   // f4 per tile ([4 quads][66])

// relaxed workgroup-scope atomics: no waitcnt is forced after them (a volatile access gets one: +250 cycles per chunk);
// the compiler barriers keep them ordered against the tile accesses, the LDS executes one wave's operations in order
__device__ __forceinline__ int flag_load(int* p) {
  const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
__device__ __forceinline__ void flag_store(int* p, int v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
template <typename F> __device__ __forceinline__ void wait_until(F cond) { while (!cond()) __builtin_amdgcn_s_sleep(1); }

template <int CHAIN>
__global__ __launch_bounds__(256) void pipe(float* out, u64* ticks, int nchunks, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f4* tin = reinterpret_cast<f4*>(smem);                 // NIN tiles
  f4* tout = tin + NIN * TILE;                           // NOUT * 2 tiles
  int* flags = reinterpret_cast<int*>(tout + 2 * NOUT * TILE);   // [0] in_ready [1] comp_done [2] comm [3] out
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x < 8) flags[threadIdx.x] = 0;
  __syncthreads();
  float acc = 0.0f;
  float p = 0.001f * lane;
  u64 t0 = __builtin_amdgcn_s_memtime();
  if (mode == 0) {
    for (int k = 0; k < nchunks + 2; ++k) {
      if (wid == 1 && k < nchunks) { for (int q = 0; q < 4; ++q) tin[(k & 1) * TILE + q * 66 + lane] = f4{(float)k, 1.0f, 2.0f, 3.0f}; }
      if (wid == 0 && k >= 1 && k - 1 < nchunks) {
        const int c = k - 1;
        f4 x = tin[(c & 1) * TILE + lane];
        for (int q = 0; q < 4; ++q) {
          f4 xn = (q < 3) ? tin[(c & 1) * TILE + (q + 1) * 66 + lane] : x;
          f4 o;
          for (int e = 0; e < 4; ++e) { for (int i = 0; i < CHAIN; ++i) p = p * 0.999f + x[e]; o[e] = p; }
          tout[((c & 1) * 2) * TILE + q * 66 + lane] = o;
          tout[((c & 1) * 2 + 1) * TILE + q * 66 + lane] = o;
          x = xn;
        }
      }
      if (wid >= 2 && k >= 2) {
        const int c = k - 2;
        for (int q = 0; q < 4; ++q) acc += tout[((c & 1) * 2 + (wid - 2)) * TILE + q * 66 + lane][0];
      }
      __syncthreads();
    }
  } else if (wid == 1) {
    for (int k = 0; k < nchunks; ++k) {
      wait_until([&] { return flag_load(flags + 1) > k - NIN; });
      for (int q = 0; q < 4; ++q) tin[(k % NIN) * TILE + q * 66 + lane] = f4{(float)k, 1.0f, 2.0f, 3.0f};
      flag_store(flags + 0, k + 1);
    }
  } else if (wid == 0) {
    int in_ready = 0, drained = 0;
    for (int k = 0; k < nchunks; ++k) {
      if (in_ready <= k) wait_until([&] { in_ready = flag_load(flags + 0); return in_ready > k; });
      if (drained <= k - NOUT) wait_until([&] { drained = min(flag_load(flags + 2), flag_load(flags + 3)); return drained > k - NOUT; });
      f4 x = tin[(k % NIN) * TILE + lane];
      const int in_next = flag_load(flags + 0);                       // prefetched for the next chunk
      const int dr_next = min(flag_load(flags + 2), flag_load(flags + 3));
      for (int q = 0; q < 4; ++q) {
        f4 xn = (q < 3) ? tin[(k % NIN) * TILE + (q + 1) * 66 + lane] : x;
        f4 o;
        for (int e = 0; e < 4; ++e) { for (int i = 0; i < CHAIN; ++i) p = p * 0.999f + x[e]; o[e] = p; }
        tout[((k % NOUT) * 2) * TILE + q * 66 + lane] = o;
        tout[((k % NOUT) * 2 + 1) * TILE + q * 66 + lane] = o;
        x = xn;
      }
      flag_store(flags + 1, k + 1);
      in_ready = in_next; drained = dr_next;
    }
  } else {
    for (int k = 0; k < nchunks; ++k) {
      wait_until([&] { return flag_load(flags + 1) > k; });
      for (int q = 0; q < 4; ++q) acc += tout[((k % NOUT) * 2 + (wid - 2)) * TILE + q * 66 + lane][0];
      flag_store(flags + wid, k + 1);
    }
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc + p;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int CHAIN>
void run(float* out, u64* ticks, int nchunks) {
  u64 h[256];
  float hout[256 * 256];
  const size_t lds = (NIN + 2 * NOUT) * TILE * sizeof(f4) + 64;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pipe<CHAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int mode = 0; mode <= 1; ++mode) {
    double sums[2];
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(pipe<CHAIN>, dim3(256), dim3(256), lds, 0, out, ticks, nchunks, mode);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hout, out, sizeof(hout), hipMemcpyDeviceToHost));
      double s = 0; for (int i = 0; i < 256 * 256; ++i) s += hout[i];
      sums[rep] = s;
    }
    CK(hipMemcpy(h, ticks, sizeof(u64) * 256, hipMemcpyDeviceToHost));
    double st = 0; for (int i = 0; i < 256; ++i) st += h[i];
    printf("chain %2d ops/step  %s: %7.0f cycles per 16-step chunk   checksum %.6e (%s)\n", CHAIN, mode ? "LDS flags  " : "s_barrier  ",
           st / 256 / nchunks, sums[1], sums[0] == sums[1] ? "repeatable" : "NOT REPEATABLE");
  }
}

int main() {
  float* out; u64* ticks;
  CK(hipMalloc(&out, 256 * 256 * sizeof(float)));
  CK(hipMalloc(&ticks, 256 * sizeof(u64)));
  run<1>(out, ticks, 200);
  run<3>(out, ticks, 200);
  run<6>(out, ticks, 200);
  run<9>(out, ticks, 200);
  return 0;
}
