// Speed-of-light probe for the lattice recursion's dependent chain (one wave64 per workgroup, lane = lattice row,
// operands in registers, nothing but the chain in the loop).  Three formulations of  p[s,t] = logadd(p[s-1,t] + X, p[s,t-1] + Y):
//   0  log2 domain as shipped in mi_wave_bidir.hip (dpp, sub, add, exp2, add, log2, add on the chain; max and the
//      copysign output beside it)
//   1  scaled linear domain, per-lane block exponent fixed over a 16-step slot:  L <- fma(dpp(L), EX, L * EY)
//      (EX = 2^(X + Rup - R), EY = 2^Y prepared off the chain)
//   2  as 1 plus what the compute wave would still have to emit per step for the other waves (the up-term Lup * EX
//      next to L, so that the occupancy ratio and log2 L can be formed elsewhere) and a frexp renormalisation per slot
//   3  the SAFE linear form: exponents aligned per step (X, Y split into integer exponent and mantissa, R <- max(R_up + XI, R + YI),
//      two ldexp): no overflow, no flush above 2^-126 relative, no transcendental
// Build: hipcc --offload-arch=gfx950 -O3 -o chain_probe chain_probe.hip ; run: ./chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void chain_kernel(const float* __restrict__ ops, float* __restrict__ out, long long* __restrict__ cyc, int slots) {
  const int lane = threadIdx.x;
  float X[16], Y[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { X[i] = ops[i * 64 + lane]; Y[i] = ops[(16 + i) * 64 + lane]; }
  float v = ops[32 * 64 + lane], acc = 0.0f;
  int R = 0;
  int XI[16], YI[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { XI[i] = MODE == 3 ? (int)ops[i * 64 + lane] - 1 : 0; YI[i] = MODE == 3 ? (int)ops[(16 + i) * 64 + lane] - 1 : 0; }
  const long long t0 = clock64();
  for (int k = 0; k < slots; ++k) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int vi = __builtin_bit_cast(int, v);
      const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x138, 0xf, 0xf, true));
      if (MODE == 0) {
        const float d = (up - v) + (X[i] - Y[i]);
        const float mx = fmaxf(up + X[i], v + Y[i]);
        const float ex = __builtin_amdgcn_exp2f(-__builtin_fabsf(d));
        v = mx + __builtin_amdgcn_logf(1.0f + ex);
        acc += __builtin_copysignf(ex, d);
      } else if (MODE == 1) {
        v = __builtin_fmaf(up, X[i], v * Y[i]);
        acc += v;
      } else if (MODE == 3) {
        const int upR = __builtin_amdgcn_update_dpp(0, R, 0x138, 0xf, 0xf, true);
        const int e1 = upR + XI[i], e2 = R + YI[i];
        const int Rn = e1 > e2 ? e1 : e2;
        const float t1 = __builtin_amdgcn_ldexpf(up * X[i], e1 - Rn);
        const float t2 = __builtin_amdgcn_ldexpf(v * Y[i], e2 - Rn);
        v = t1 + t2; R = Rn;
        acc += t1; acc += v;
      } else {
        const float t1 = up * X[i];
        v = __builtin_fmaf(v, Y[i], t1);
        acc += t1; acc += v;     // stand-ins for the two tile writes (up-term and L)
      }
    }
    if (MODE >= 2) {  // per-slot renormalisation: L -> mantissa, exponent into the block exponent
      R += __builtin_amdgcn_frexp_expf(v);
      v = __builtin_amdgcn_frexp_mantf(v);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = clock64();
  out[blockIdx.x * 64 + lane] = v + acc + (float)R;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int WG = 128, slots = 4096;
  std::vector<float> h(33 * 64);
  for (int mode = 0; mode < 4; ++mode) {
    for (int i = 0; i < 32 * 64; ++i) h[i] = mode == 0 ? -1.0f : 0.5f;    // log2 domain: X = Y = -1;  linear: EX = EY = 1/2
    for (int i = 0; i < 64; ++i) h[32 * 64 + i] = mode == 0 ? 0.0f : 1.0f;
    float *ops, *out; long long* cyc;
    hipMalloc(&ops, h.size() * 4); hipMalloc(&out, WG * 64 * 4); hipMalloc(&cyc, WG * 8);
    hipMemcpy(ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) chain_kernel<0><<<WG, 64>>>(ops, out, cyc, slots);
      if (mode == 1) chain_kernel<1><<<WG, 64>>>(ops, out, cyc, slots);
      if (mode == 2) chain_kernel<2><<<WG, 64>>>(ops, out, cyc, slots);
      if (mode == 3) chain_kernel<3><<<WG, 64>>>(ops, out, cyc, slots);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    float o; hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost);
    printf("mode %d: %.1f ns/step  (%.2f us per 16-step slot), clock64 ticks/step %.2f, out %g\n", mode, best * 1e6 / (slots * 16.0), best * 1e3 / slots, (double)c0 / (slots * 16.0), o);
    hipFree(ops); hipFree(out); hipFree(cyc);
  }
  return 0;
}
