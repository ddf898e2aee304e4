// Exhaustive check of the short correctly-rounded sequences of include/go2sim_detmath.h (dm_rsqrt_cr, dm_rcp_cr) on the device:
//   every positive float a in [2^-100, 2^100]:  dm_rsqrt_cr(a) == (float)(1.0 / sqrt((double)a))     (the host definition)
//   every float x with |x| in [2^-100, 2^100]:  dm_rcp_cr(x)   == 1.0f / x                            (IEEE division)
//   every non-negative float a:                  dm_sqrt_cr(a)  == sqrtf(a)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/exhaustive/cr_math_probe.hip -o gpurun_out/cr_math_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include "go2sim_detmath.h"

__global__ void k_rsqrt(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* first_bad) {
  uint64_t i = (uint64_t)lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < hi; i += stride) {
    const float a = dm_bits2f((uint32_t)i);
    const float got = dm_rsqrt_cr(a), got2 = dm_rsqrt_cr_inrange(a);          // the guarded form and the branch-free form for clamped arguments
    const float ref = (float)(1.0 / sqrt((double)a));
    if (dm_f2bits(got) != dm_f2bits(ref) || dm_f2bits(got2) != dm_f2bits(ref)) { if (atomicAdd(bad, 1ull) == 0ull) *first_bad = (uint32_t)i; }
  }
}
__global__ void k_sqrt(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* first_bad) {
  uint64_t i = (uint64_t)lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < hi; i += stride) {
    const float a = dm_bits2f((uint32_t)i);
    const float got = dm_sqrt_cr(a);
    const float ref = sqrtf(a);
    if (dm_f2bits(got) != dm_f2bits(ref)) { if (atomicAdd(bad, 1ull) == 0ull) *first_bad = (uint32_t)i; }
  }
}
__global__ void k_rcp(uint32_t lo, uint32_t hi, uint32_t sign, unsigned long long* bad, uint32_t* first_bad) {
  uint64_t i = (uint64_t)lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < hi; i += stride) {
    const float x = dm_bits2f((uint32_t)i | sign);
    const float got = dm_rcp_cr(x);
    const float ref = 1.0f / x;
    if (dm_f2bits(got) != dm_f2bits(ref)) { if (atomicAdd(bad, 1ull) == 0ull) *first_bad = (uint32_t)i | sign; }
  }
}
__global__ void k_dump(uint32_t lo, uint32_t n, float* out_rsqrt, float* out_rcp, float* out_sqrt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float a = dm_bits2f(lo + i); out_rsqrt[i] = dm_rsqrt_cr(a); out_rcp[i] = dm_rcp_cr(a); out_sqrt[i] = dm_sqrt_cr(a); }
}

int main() {
  unsigned long long* bad; uint32_t* fb;
  (void)hipMalloc(&bad, 16); (void)hipMalloc(&fb, 8);
  const uint32_t lo = (uint32_t)(127 - 100) << 23, hi = (uint32_t)(127 + 100) << 23;
  unsigned long long hb[2]; uint32_t hf[2];
  (void)hipMemset(bad, 0, 16); (void)hipMemset(fb, 0, 8);
  hipLaunchKernelGGL(k_rsqrt, dim3(4096), dim3(256), 0, 0, lo, hi, bad, fb);
  hipLaunchKernelGGL(k_rcp, dim3(4096), dim3(256), 0, 0, lo, hi, 0u, bad + 1, fb + 1);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, fb, 8, hipMemcpyDeviceToHost);
  printf("rsqrt_cr: %llu mismatches over %u inputs (first 0x%08x)\n", hb[0], hi - lo, hf[0]);
  const unsigned long long bad_rsqrt = hb[0];
  printf("rcp_cr (+): %llu mismatches over %u inputs (first 0x%08x)\n", hb[1], hi - lo, hf[1]);
  (void)hipMemset(bad, 0, 16);
  hipLaunchKernelGGL(k_rcp, dim3(4096), dim3(256), 0, 0, lo, hi, 0x80000000u, bad + 1, fb + 1);
  hipLaunchKernelGGL(k_sqrt, dim3(4096), dim3(256), 0, 0, 0u, 0x7f800001u, bad, fb);       // every non-negative float, zero / denormals / infinity included
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, fb, 8, hipMemcpyDeviceToHost);
  printf("rcp_cr (-): %llu mismatches (first 0x%08x)\n", hb[1], hf[1]);
  printf("sqrt_cr: %llu mismatches over all non-negative floats (first 0x%08x)\n", hb[0], hf[0]);
  const unsigned long long bad_sqrt = hb[0];
  // two binades against the HOST definitions (ties the device's float64 reference to the CPU's)
  const uint32_t n = 2u << 23, lo2 = 127u << 23;
  float *d1, *d2, *d3; (void)hipMalloc(&d1, (size_t)n * 4); (void)hipMalloc(&d2, (size_t)n * 4); (void)hipMalloc(&d3, (size_t)n * 4);
  hipLaunchKernelGGL(k_dump, dim3((n + 255) / 256), dim3(256), 0, 0, lo2, n, d1, d2, d3);
  std::vector<float> h1(n), h2(n), h3(n);
  (void)hipMemcpy(h1.data(), d1, (size_t)n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h2.data(), d2, (size_t)n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h3.data(), d3, (size_t)n * 4, hipMemcpyDeviceToHost);
  unsigned long long b1 = 0, b2 = 0, b3 = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const float a = dm_bits2f(lo2 + i);
    if (dm_f2bits(h1[i]) != dm_f2bits(dm_rsqrt_cr(a))) ++b1;
    if (dm_f2bits(h2[i]) != dm_f2bits(dm_rcp_cr(a))) ++b2;
    if (dm_f2bits(h3[i]) != dm_f2bits(dm_sqrt_cr(a))) ++b3;
  }
  printf("device vs HOST definitions over two binades [1, 4): rsqrt %llu, rcp %llu, sqrt %llu mismatches of %u\n", b1, b2, b3, n);
  return (bad_rsqrt || hb[1] || bad_sqrt || b1 || b2 || b3) ? 1 : 0;
}
