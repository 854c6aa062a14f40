// scripts/probes/stream_probe.hip -- what separates the path's row-streaming kernels (4.4 - 4.7 TB/s on [rows x 2000 B]
// tensors) from a plain sum over k of the same tensor (5.8 TB/s)?  Variants of "read a [rows, C] f32 matrix once":
//   hipcc -O3 --offload-arch=gfx950 -o scripts/probes/stream_probe.bin scripts/probes/stream_probe.hip && scripts/probes/stream_probe.bin [rows C]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(3); } } while (0)

__device__ __forceinline__ float wsum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ float wmax(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }

// K0: thread per 16 bytes of an output row, sum over r consecutive rows (the prune gather's d am)
__global__ void k_sumk(const float* __restrict__ x, float* __restrict__ out, int C, int r, size_t total) {
  const int n4 = C >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bt = i / n4; const int c = (int)(i - bt * n4);
    f4 acc = {0, 0, 0, 0};
    for (int k = 0; k < r; ++k) acc += reinterpret_cast<const f4u*>(x + (bt * r + k) * C)[c];
    reinterpret_cast<f4u*>(out + bt * C)[c] = acc;
  }
}
// K1: one wave per row, row in registers (NQ quads per lane), MODE 0 = logsumexp, 1 = max only, 2 = plain sum
template <int NQ, int MODE>
__global__ __launch_bounds__(256) void k_row(const float* __restrict__ x, float* __restrict__ out, size_t rows, int C) {
  const int lane = threadIdx.x & 63, n4 = C >> 2;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f4u* x4 = reinterpret_cast<const f4u*>(x + row * C);
  f4 v[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { const int i = lane + 64 * q; v[q] = i < n4 ? (f4)x4[i] : f4{-1e30f, -1e30f, -1e30f, -1e30f}; }
  float m = -1e30f, s = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) m = fmaxf(fmaxf(m, fmaxf(v[q][0], v[q][1])), fmaxf(v[q][2], v[q][3]));
  if (MODE == 2) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) if (lane + 64 * q < n4) s += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
    s = wsum(s);
    if (lane == 0) out[row] = s;
    return;
  }
  m = wmax(m);
  if (MODE == 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) if (lane + 64 * q < n4) s += __expf(v[q][0] - m) + __expf(v[q][1] - m) + __expf(v[q][2] - m) + __expf(v[q][3] - m);
    s = wsum(s);
    if (lane == 0) out[row] = m + __logf(s);
  } else if (lane == 0) out[row] = m;
}
// K2: RW rows per wave at once (more loads in flight per wave), plain sum
template <int NQ, int RW>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ x, float* __restrict__ out, size_t rows, int C) {
  const int lane = threadIdx.x & 63, n4 = C >> 2;
  const size_t row0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= rows) return;
  f4 v[RW][NQ];
#pragma unroll
  for (int w = 0; w < RW; ++w) {
    const f4u* x4 = reinterpret_cast<const f4u*>(x + (row0 + w < rows ? row0 + w : rows - 1) * C);
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int i = lane + 64 * q; v[w][q] = x4[i < n4 ? i : n4 - 1]; }
  }
#pragma unroll
  for (int w = 0; w < RW; ++w) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) if (lane + 64 * q < n4) s += (v[w][q][0] + v[w][q][1]) + (v[w][q][2] + v[w][q][3]);
    s = wsum(s);
    if (lane == 0 && row0 + w < rows) out[row0 + w] = s;
  }
}
// K3: the matrix as a flat array: every lane 16 bytes, consecutive lanes consecutive addresses, grid-stride
__global__ void k_flat(const float* __restrict__ x, float* __restrict__ out, size_t n4) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += reinterpret_cast<const f4*>(x)[i];
  const float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (s == 123.456f) out[0] = s;
}
// K4: flat, one block per contiguous 16 KB, no grid stride (a short-lived wave per KB like K0)
__global__ void k_flat_short(const float* __restrict__ x, float* __restrict__ out, size_t n4) {
  const size_t base = (size_t)blockIdx.x * 1024;   // 1024 quads = 16 KB per block of 256 threads: 4 loads per thread
  f4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < 4; ++u) { const size_t i = base + threadIdx.x + 256 * u; if (i < n4) acc += reinterpret_cast<const f4*>(x)[i]; }
  const float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (s == 123.456f) out[0] = s;
}

int main(int argc, char** argv) {
  const size_t rows = argc > 1 ? strtoull(argv[1], 0, 10) : 160000;
  const int C = argc > 2 ? atoi(argv[2]) : 500;
  const int r = 5;
  const size_t n = rows * C;
  float *x, *out;
  HIP_OK(hipMalloc(&x, n * 4)); HIP_OK(hipMalloc(&out, n * 4 / r + 4096));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
  HIP_OK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    HIP_OK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
    float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    HIP_OK(hipGetLastError());
    const double us = 1e3 * ms / reps;
    printf("%-58s %8.1f us  %6.2f TB/s (read only)\n", name, us, n * 4 / us / 1e6);
  };
  printf("rows %zu x C %d = %.1f MB\n", rows, C, n * 4 / 1e6);
  const size_t tot = rows / r * (C / 4);
  timeit("K0 sum over k=5 rows, thread per quad (+ 1/5 written)", [&] { hipLaunchKernelGGL(k_sumk, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, x, out, C, r, tot); });
  const unsigned nb = (unsigned)((rows + 3) / 4);
  if (C <= 512) {
    timeit("K1 wave per row, logsumexp", [&] { hipLaunchKernelGGL((k_row<2, 0>), dim3(nb), dim3(256), 0, 0, x, out, rows, C); });
    timeit("K1 wave per row, max only", [&] { hipLaunchKernelGGL((k_row<2, 1>), dim3(nb), dim3(256), 0, 0, x, out, rows, C); });
    timeit("K1 wave per row, plain sum", [&] { hipLaunchKernelGGL((k_row<2, 2>), dim3(nb), dim3(256), 0, 0, x, out, rows, C); });
    timeit("K2 wave per 2 rows, plain sum", [&] { hipLaunchKernelGGL((k_rows<2, 2>), dim3((nb + 1) / 2), dim3(256), 0, 0, x, out, rows, C); });
    timeit("K2 wave per 4 rows, plain sum", [&] { hipLaunchKernelGGL((k_rows<2, 4>), dim3((nb + 3) / 4), dim3(256), 0, 0, x, out, rows, C); });
    timeit("K2 wave per 8 rows, plain sum", [&] { hipLaunchKernelGGL((k_rows<2, 8>), dim3((nb + 7) / 8), dim3(256), 0, 0, x, out, rows, C); });
  } else {
    timeit("K1 wave per row, logsumexp", [&] { hipLaunchKernelGGL((k_row<4, 0>), dim3(nb), dim3(256), 0, 0, x, out, rows, C); });
    timeit("K1 wave per row, plain sum", [&] { hipLaunchKernelGGL((k_row<4, 2>), dim3(nb), dim3(256), 0, 0, x, out, rows, C); });
    timeit("K2 wave per 2 rows, plain sum", [&] { hipLaunchKernelGGL((k_rows<4, 2>), dim3((nb + 1) / 2), dim3(256), 0, 0, x, out, rows, C); });
    timeit("K2 wave per 4 rows, plain sum", [&] { hipLaunchKernelGGL((k_rows<4, 4>), dim3((nb + 3) / 4), dim3(256), 0, 0, x, out, rows, C); });
  }
  const size_t n4 = n / 4;
  timeit("K3 flat, grid-stride, 2048 blocks x 256", [&] { hipLaunchKernelGGL(k_flat, dim3(2048), dim3(256), 0, 0, x, out, n4); });
  timeit("K3 flat, grid-stride, 8192 blocks x 256", [&] { hipLaunchKernelGGL(k_flat, dim3(8192), dim3(256), 0, 0, x, out, n4); });
  timeit("K4 flat, 16 KB per block, no stride", [&] { hipLaunchKernelGGL(k_flat_short, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, x, out, n4); });
  return 0;
}
