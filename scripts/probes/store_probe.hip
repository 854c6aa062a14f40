// scripts/probes/store_probe.hip -- does the fused forward's store pattern (a wave instruction = 16 rows x 64 bytes, rows T
// floats apart) reach the write rate of row-contiguous stores (4 rows x 256 bytes)?  Three [R, T] f32 outputs like px / py /
// prod, one workgroup per 64 frames x 208 rows, 256 threads.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/probes/store_probe.bin scripts/probes/store_probe.hip && scripts/probes/store_probe.bin [R T]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(3); } } while (0)

// MODE 0: accumulator layout (lane n = lane & 15 -> row 16 i + n, q = lane >> 4 -> frames 16 w + 4 q .. + 3)
// MODE 1: row layout (lane & 15 -> frames 4 (lane & 15) .. + 3 of the tile, lane >> 4 -> row 4 k + (lane >> 4), wave w takes k = w, w + 4, ..)
// MODE 2: row layout, one row per instruction, 4 bytes per lane (for rows that are only 4-byte aligned)
template <int MODE>
__global__ __launch_bounds__(256) void k_store(float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2, int R, int T, int ld0) {
  const int t0 = blockIdx.x * 64, r0 = blockIdx.y * 208;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const f4 v = {1.f * blockIdx.x, 2.f, 3.f, 4.f * lane};
  if (MODE == 0) {
    const int tq = t0 + 16 * w + 4 * (lane >> 4);
    if (tq + 3 >= T) return;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      const int r = r0 + 16 * i + (lane & 15);
      if (r >= R) continue;
      *reinterpret_cast<f4u*>(o1 + (size_t)r * T + tq) = v;
      *reinterpret_cast<f4u*>(o2 + (size_t)r * T + tq) = v;
      *reinterpret_cast<f4u*>(o0 + (size_t)r * ld0 + tq) = v;
    }
  } else if (MODE == 1) {
    const int tq = t0 + 4 * (lane & 15);
    if (tq + 3 >= T) return;
    for (int o = 0; o < 3; ++o) {
      float* dst = o == 0 ? o1 : (o == 1 ? o2 : o0);
      const int ld = o == 2 ? ld0 : T;
#pragma unroll
      for (int k = 0; k < 13; ++k) {
        const int r = r0 + 16 * k + 4 * w + (lane >> 4);
        if (r >= R) continue;
        *reinterpret_cast<f4u*>(dst + (size_t)r * ld + tq) = v;
      }
    }
  } else {
    const int t = t0 + lane;
    if (t >= T) return;
    for (int o = 0; o < 3; ++o) {
      float* dst = o == 0 ? o1 : (o == 1 ? o2 : o0);
      const int ld = o == 2 ? ld0 : T;
#pragma unroll 13
      for (int k = 0; k < 52; ++k) {
        const int r = r0 + 4 * k + w;
        if (r >= R) continue;
        dst[(size_t)r * ld + t] = v[0];
      }
    }
  }
}

int main(int argc, char** argv) {
  const int R = argc > 2 ? atoi(argv[1]) : 8008, T = argc > 2 ? atoi(argv[2]) : 8000;
  float *o0, *o1, *o2;
  HIP_OK(hipMalloc(&o0, (size_t)R * (T + 1) * 4)); HIP_OK(hipMalloc(&o1, (size_t)R * T * 4)); HIP_OK(hipMalloc(&o2, (size_t)R * T * 4));
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  const dim3 grid((T + 63) / 64, (R + 207) / 208);
  const double mb = 3.0 * R * (double)T * 4 / 1e6;
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    HIP_OK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) launch();
    HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
    float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, 100.f * ms, mb / (100.0 * ms));
  };
  printf("R=%d T=%d: %.0f MB per launch, grid %u x %u\n", R, T, mb, grid.x, grid.y);
  run("accumulator layout, aligned rows", [&] { hipLaunchKernelGGL(k_store<0>, grid, dim3(256), 0, 0, o0, o1, o2, R, T, T); });
  run("accumulator layout, px rows T+1", [&] { hipLaunchKernelGGL(k_store<0>, grid, dim3(256), 0, 0, o0, o1, o2, R, T, T + 1); });
  run("row layout 4 x 256 B, aligned rows", [&] { hipLaunchKernelGGL(k_store<1>, grid, dim3(256), 0, 0, o0, o1, o2, R, T, T); });
  run("row layout 4 x 256 B, px rows T+1", [&] { hipLaunchKernelGGL(k_store<1>, grid, dim3(256), 0, 0, o0, o1, o2, R, T, T + 1); });
  run("row layout 1 x 256 B dwords, px rows T+1", [&] { hipLaunchKernelGGL(k_store<2>, grid, dim3(256), 0, 0, o0, o1, o2, R, T, T + 1); });
  return 0;
}
