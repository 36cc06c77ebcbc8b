// Exhaustive check on the device: for EVERY binary32 y, is v_rcp_f32 + one Newton step the correctly rounded 1/y?
// Prints the number of mismatches against the IEEE quotient 1.0f / y per exponent class, for the candidates of a
// cheaper reciprocal in the numeric contract.  hipcc --offload-arch=gfx950 -O2 -ffp-contract=off
// -fhip-fp32-correctly-rounded-divide-sqrt scripts/microbench/rcp_exhaustive.hip -o scripts/microbench/rcp_exhaustive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ inline float newton1(float y) {
  float r = __builtin_amdgcn_rcpf(y);
  float e = __builtin_fmaf(-y, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ inline float newton1_fixup(float y) { return __builtin_amdgcn_div_fixupf(newton1(y), y, 1.0f); }
__device__ inline float newton2(float y) {
  float r = newton1(y);
  float e = __builtin_fmaf(-y, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}

// bad[variant][exponent field 0..255]
__global__ void check(unsigned long long *bad) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint64_t u = tid; u < (1ull << 32); u += stride) {
    float y;
    uint32_t b = (uint32_t)u;
    memcpy(&y, &b, 4);
    const float ref = 1.0f / y;
    uint32_t rb;
    memcpy(&rb, &ref, 4);
    const float c[3] = {newton1(y), newton1_fixup(y), newton2(y)};
    for (int v = 0; v < 3; v++) {
      uint32_t cb;
      memcpy(&cb, &c[v], 4);
      const bool bothNaN = (ref != ref) && (c[v] != c[v]);
      if (cb != rb && !bothNaN) atomicAdd(&bad[v * 256 + ((b >> 23) & 0xff)], 1ull);
    }
  }
}

int main() {
  unsigned long long *d, h[3 * 256];
  hipMalloc(&d, sizeof(h));
  hipMemset(d, 0, sizeof(h));
  check<<<4096, 256>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[3] = {"rcp + 1 Newton step", "rcp + 1 Newton step + v_div_fixup", "rcp + 2 Newton steps"};
  for (int v = 0; v < 3; v++) {
    unsigned long long tot = 0;
    for (int e = 0; e < 256; e++) tot += h[v * 256 + e];
    printf("%s: %llu of 2^32 inputs differ from 1.0f / y; by exponent field:", names[v], tot);
    for (int e = 0; e < 256; e++)
      if (h[v * 256 + e]) printf(" [%d]=%llu", e, h[v * 256 + e]);
    printf("\n");
  }
  return 0;
}
