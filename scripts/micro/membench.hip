// scripts/micro/membench.hip -- diagnostic microbenchmark (not part of the product): cost per wave-instruction
// of the access shapes the wavefront kernels could use to move a [64 rows x 16 columns] float tile between a
// row-major lattice (row stride T+1 floats, 4-byte aligned rows) and registers, with 1..8 waves per CU active.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// mode 0: dwordx4, lane -> row lane>>2, piece lane&3 (16 rows x 64 B per instruction)   [current kernels]
// mode 1: dword,   lane -> column lane of ONE row (256 B contiguous per instruction)
// mode 2: dwordx2, lane -> row lane>>3, piece lane&7 (8 rows x 64 B)
// mode 3: dwordx4, lane -> row lane>>4, piece lane&15 (4 rows x 256 B)
template <int MODE, bool STORE>
__global__ void k(float* buf, int stride, int chunks, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* base = buf + (size_t)(blockIdx.x * 8 + wave) * 64 * stride;   // 64 rows per wave
  float acc = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int c = 0; c < chunks; ++c) {
    const int col0 = 16 * c + 1;  // +1: misaligned start
    if (MODE == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float* p = base + (size_t)(16 * m + (lane >> 2)) * stride + col0 + 4 * (lane & 3);
        if (STORE) *(f4u*)p = f4u{acc, 1.f, 2.f, 3.f}; else { f4u v = *(const f4u*)p; acc += v[0] + v[3]; }
      }
    } else if (MODE == 1) {
      // one instruction per row: but a chunk has only 16 columns -> use 64 columns of one row per instr, 16 rows
      // per "chunk-equivalent" (same bytes: 16 instr x 256 B = 4 KiB = 64 rows x 16 cols)
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        float* p = base + (size_t)(4 * m + (c & 3)) * stride + 64 * (c >> 2) + 1 + lane;
        if (STORE) *p = acc; else acc += *p;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        float* p = base + (size_t)(8 * m + (lane >> 3)) * stride + col0 + 2 * (lane & 7);
        if (STORE) *(f2u*)p = f2u{acc, 1.f}; else { f2u v = *(const f2u*)p; acc += v[0] + v[1]; }
      }
    } else {
      // 4 rows x 256 B: a "chunk-equivalent" = 64 rows x 16 cols = 4 KiB = 4 instr of 1 KiB: rows 16m'..: use 64-col pieces
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float* p = base + (size_t)(16 * (c & 3) + 4 * m + (lane >> 4)) * stride + 64 * (c >> 2) + 1 + 4 * (lane & 15);
        if (STORE) *(f4u*)p = f4u{acc, 1.f, 2.f, 3.f}; else { f4u v = *(const f4u*)p; acc += v[0] + v[3]; }
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE, bool STORE>
void run(const char* name, float* buf, int stride, int waves, unsigned long long* dout, float* sink) {
  const int chunks = 60, blocks = 32;
  std::vector<unsigned long long> h(blocks * 8);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<MODE, STORE>), dim3(blocks), dim3(64 * waves), 0, 0, buf, stride, chunks, dout, sink);
    CK(hipDeviceSynchronize());
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, STORE>), dim3(blocks), dim3(64 * waves), 0, 0, buf, stride, chunks, dout, sink);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(h.data(), dout, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost));
  double avg = 0; for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) avg += h[b * 8 + w]; avg /= blocks * waves;
  printf("%-34s waves/CU %d: %8.0f memtime ticks per 4 KiB tile (64x16 floats), kernel %.1f us, %.1f GB/s per CU\n", name, waves,
         avg / chunks, ms * 1e3, (double)chunks * 4096 * waves / (ms * 1e-3) / 1e9);
}

int main() {
  const int stride = 1001;
  const size_t n = (size_t)32 * 8 * 64 * stride + 4096;
  float *buf, *sink; unsigned long long* dout;
  CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4)); CK(hipMalloc(&dout, 8 * 32 * 8)); CK(hipMalloc(&sink, 16));
  for (int waves : {1, 2, 4, 8}) {
    run<0, false>("load  x4 16rows x 64B (current)", buf, stride, waves, dout, sink);
    run<2, false>("load  x2  8rows x 64B", buf, stride, waves, dout, sink);
    run<1, false>("load  x1  1row x 256B", buf, stride, waves, dout, sink);
    run<3, false>("load  x4  4rows x 256B", buf, stride, waves, dout, sink);
    run<0, true>("store x4 16rows x 64B (current)", buf, stride, waves, dout, sink);
    run<2, true>("store x2  8rows x 64B", buf, stride, waves, dout, sink);
    run<1, true>("store x1  1row x 256B", buf, stride, waves, dout, sink);
    run<3, true>("store x4  4rows x 256B", buf, stride, waves, dout, sink);
  }
  return 0;
}
