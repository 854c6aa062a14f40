// scripts/probes/bf16x3_gemm_probe.hip -- what would the builders' backward contractions cost as bf16 x 3 split products
// (hi*hi + hi*lo + lo*hi, bf16 operands, f32 accumulation on the bf16 matrix pipe) instead of f32 MFMA GEMMs, and how exact
// are they?  rocBLAS strided-batched gemm_ex, bf16 in / f32 out, best of the library's candidates.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/probes/bf16x3_gemm_probe.bin scripts/probes/bf16x3_gemm_probe.hip -lrocblas
//   scripts/probes/bf16x3_gemm_probe.bin [B T S1 C]
#define ROCBLAS_BETA_FEATURES_API 1
#define ROCBLAS_NO_DEPRECATED_WARNINGS 1
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(3); } } while (0)
#define RB_OK(x) do { rocblas_status s_ = (x); if (s_ != rocblas_status_success) { fprintf(stderr, "rocBLAS status %d at line %d\n", (int)s_, __LINE__); exit(4); } } while (0)
typedef unsigned short bf16;
__device__ __forceinline__ bf16 to_bf16(float x) {   // round to nearest even
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16)(u >> 16);
}
__device__ __forceinline__ float from_bf16(bf16 h) { return __uint_as_float((unsigned)h << 16); }
__global__ void split_kernel(const float* __restrict__ x, bf16* __restrict__ hi, bf16* __restrict__ lo, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    const bf16 h = to_bf16(v);
    hi[i] = h;
    lo[i] = to_bf16(v - from_bf16(h));
  }
}
// stacked along the leading dimension: out[b][j * rows + r][c] = part_j(x[b][r][c]), parts given by `sel` (0 hi, 1 lo)
__global__ void stack_kernel(const float* __restrict__ x, bf16* __restrict__ out, int rows, int cols, int s0, int s1, int s2, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t per = (size_t)rows * cols, b = i / per, rc = i - b * per;
    const float v = x[i];
    const bf16 h = to_bf16(v), l = to_bf16(v - from_bf16(h));
    bf16* o = out + b * 3 * per + rc;
    o[0] = s0 ? l : h; o[per] = s1 ? l : h; o[2 * per] = s2 ? l : h;
  }
}
__global__ void fill_kernel(float* x, size_t n, unsigned seed, float lo, float hi) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    x[i] = lo + (hi - lo) * (h >> 8) * (1.0f / 16777216.0f);
  }
}
struct G { rocblas_operation ta, tb; int m, n, k, lda, ldb, ldc, batch; long long sa, sb, sc; };
static rocblas_handle h;
static hipEvent_t e0, e1;
static rocblas_status launch(const G& g, const void* A, const void* B, float* C, rocblas_datatype in, float beta, rocblas_gemm_algo algo, int sol) {
  const float one = 1.0f;
  return rocblas_gemm_strided_batched_ex(h, g.ta, g.tb, g.m, g.n, g.k, &one, A, in, g.lda, g.sa, B, in, g.ldb, g.sb, &beta, C, rocblas_datatype_f32_r, g.ldc, g.sc, C, rocblas_datatype_f32_r, g.ldc, g.sc, g.batch, rocblas_datatype_f32_r, algo, sol, rocblas_gemm_flags_none);
}
static float run(const G& g, const void* A, const void* B, float* C, rocblas_datatype in, float beta, rocblas_gemm_algo algo, int sol, int reps) {
  auto call = [&] { return launch(g, A, B, C, in, beta, algo, sol); };
  if (call() != rocblas_status_success) return -1.f;
  HIP_OK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) if (call() != rocblas_status_success) return -1.f;
  HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  return 1e3f * ms / reps;
}
static float best(const G& g, const void* A, const void* B, float* C, rocblas_datatype in, float beta, int* which, int* ncand) {
  const float one = 1.0f;
  rocblas_int n = 0;
  auto q = [&](rocblas_int* l, rocblas_int* sz) { return rocblas_gemm_strided_batched_ex_get_solutions(h, g.ta, g.tb, g.m, g.n, g.k, &one, A, in, g.lda, g.sa, B, in, g.ldb, g.sb, &beta, C, rocblas_datatype_f32_r, g.ldc, g.sc, C, rocblas_datatype_f32_r, g.ldc, g.sc, g.batch, rocblas_datatype_f32_r, rocblas_gemm_algo_solution_index, rocblas_gemm_flags_none, l, sz); };
  float b = run(g, A, B, C, in, beta, rocblas_gemm_algo_standard, 0, 5);
  *which = 0; *ncand = 0;
  if (q(nullptr, &n) != rocblas_status_success || n <= 0) return b;
  std::vector<rocblas_int> s((size_t)n);
  if (q(s.data(), &n) != rocblas_status_success) return b;
  *ncand = n;
  for (int i = 0; i < n; ++i) {
    const float t = run(g, A, B, C, in, beta, rocblas_gemm_algo_solution_index, s[(size_t)i], 2);
    if (t > 0.f && t < 0.97f * b) { const float t2 = run(g, A, B, C, in, beta, rocblas_gemm_algo_solution_index, s[(size_t)i], 5); if (t2 > 0.f && t2 < b) { b = t2; *which = s[(size_t)i]; } }
  }
  return b;
}
static double maxrel(const float* a, const float* b, size_t n) {
  std::vector<float> ha(n), hb(n);
  HIP_OK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  double mx = 0, worst = 0;
  for (size_t i = 0; i < n; ++i) mx = fmax(mx, fabs((double)ha[i]));
  for (size_t i = 0; i < n; ++i) worst = fmax(worst, fabs((double)ha[i] - hb[i]));
  return worst / mx;
}
int main(int argc, char** argv) {
  const int B = argc > 4 ? atoi(argv[1]) : 32, T = argc > 4 ? atoi(argv[2]) : 1000, S1 = argc > 4 ? atoi(argv[3]) : 201, C = argc > 4 ? atoi(argv[4]) : 500;
  RB_OK(rocblas_create_handle(&h)); HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  const size_t nW = (size_t)B * S1 * T, nA = (size_t)B * T * C, nL = (size_t)B * S1 * C;
  float *W, *A, *L, *damp, *damp2, *dlmp, *dlmp2;
  bf16 *Wst, *Lst, *Ahi, *Alo;
  HIP_OK(hipMalloc(&W, nW * 4)); HIP_OK(hipMalloc(&A, nA * 4)); HIP_OK(hipMalloc(&L, nL * 4));
  HIP_OK(hipMalloc(&damp, nA * 4)); HIP_OK(hipMalloc(&damp2, nA * 4)); HIP_OK(hipMalloc(&dlmp, nL * 4)); HIP_OK(hipMalloc(&dlmp2, nL * 4));
  HIP_OK(hipMalloc(&Wst, nW * 6)); HIP_OK(hipMalloc(&Lst, nL * 6)); HIP_OK(hipMalloc(&Ahi, nA * 2)); HIP_OK(hipMalloc(&Alo, nA * 2));
  fill_kernel<<<2048, 256>>>(W, nW, 1u, -3.0f, 0.0f);      // W <= 0 (occupancies over products)
  fill_kernel<<<2048, 256>>>(A, nA, 2u, 0.0f, 1.0f);       // probabilities relative to the row maximum
  fill_kernel<<<2048, 256>>>(L, nL, 3u, 0.0f, 1.0f);
  stack_kernel<<<2048, 256>>>(W, Wst, S1, T, 0, 0, 1, nW);   // [W_hi; W_hi; W_lo]
  stack_kernel<<<2048, 256>>>(L, Lst, S1, C, 0, 1, 0, nL);   // [L_hi; L_lo; L_hi]
  split_kernel<<<2048, 256>>>(A, Ahi, Alo, nA);
  HIP_OK(hipDeviceSynchronize());
  int which, nc;
  // kind 2: damp[T,C] = W^T . L  (column major: out^T = L^T . W ... as normalizer_gemm.hip: op N on L (C x S1, ld C), op T on W (T x S1 -> ld T))
  G k2{rocblas_operation_none, rocblas_operation_transpose, C, T, S1, C, T, C, B, (long long)S1 * C, (long long)S1 * T, (long long)T * C};
  float t = best(k2, L, W, damp, rocblas_datatype_f32_r, 0.f, &which, &nc);
  printf("kind 2 (d am product) f32: %.1f us (%.1f TFLOP/s), %d candidates\n", t, 2.0 * B * S1 * T * C / t / 1e6, nc);
  G k2s = k2; k2s.k = 3 * S1; k2s.sa = 3LL * S1 * C; k2s.sb = 3LL * S1 * T;
  t = best(k2s, Lst, Wst, damp2, rocblas_datatype_bf16_r, 0.f, &which, &nc);
  printf("kind 2 bf16 x 3, K stacked (%d): %.1f us, %d candidates, max |diff| / max |f32| = %.2e\n", 3 * S1, t, nc, t > 0 ? maxrel(damp, damp2, nA) : -1.0);
  // kind 1: dlmp[S1,C] = W . A : op N on A (C x T, ld C), op N on W (T x S1, ld T)
  G k1{rocblas_operation_none, rocblas_operation_none, C, S1, T, C, T, C, B, (long long)T * C, (long long)S1 * T, (long long)S1 * C};
  t = best(k1, A, W, dlmp, rocblas_datatype_f32_r, 0.f, &which, &nc);
  printf("kind 1 (d lm product) f32: %.1f us (%.1f TFLOP/s), %d candidates\n", t, 2.0 * B * S1 * T * C / t / 1e6, nc);
  G k1b = k1; k1b.sb = 3LL * S1 * T;
  const bf16 *Whi = Wst, *Wlo = Wst + 2 * (size_t)S1 * T;
  float t1 = best(k1b, Ahi, Whi, dlmp2, rocblas_datatype_bf16_r, 0.f, &which, &nc);
  const int sol = which;
  // the same kernel for the three products: hi*hi (beta 0), lo_A*hi_W, hi_A*lo_W (beta 1)
  auto three = [&] {
    const rocblas_gemm_algo al = sol ? rocblas_gemm_algo_solution_index : rocblas_gemm_algo_standard;
    RB_OK(launch(k1b, Ahi, Whi, dlmp2, rocblas_datatype_bf16_r, 0.f, al, sol));
    RB_OK(launch(k1b, Alo, Whi, dlmp2, rocblas_datatype_bf16_r, 1.f, al, sol));
    RB_OK(launch(k1b, Ahi, Wlo, dlmp2, rocblas_datatype_bf16_r, 1.f, al, sol));
  };
  three(); HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) three(); HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  printf("kind 1 bf16 x 3, three launches: one %.1f us, all three %.1f us, %d candidates, max |diff| / max |f32| = %.2e\n", t1, 1e3f * ms / 5, nc, maxrel(dlmp, dlmp2, nL));
  return 0;
}
