// Exhaustive check on the device: which short sequences give the correctly rounded sqrt(x) for every x of the normal range?
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
//       scripts/microbench/sqrt_exhaustive.hip -o scripts/microbench/sqrt_exhaustive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ inline float f(uint32_t b) { float y; memcpy(&y, &b, 4); return y; }
__device__ inline uint32_t u(float y) { uint32_t b; memcpy(&b, &y, 4); return b; }

// A: rsq, s = x·r, one correction with the exact residual
__device__ inline float seqA(float x) {
  float r = __builtin_amdgcn_rsqf(x);
  float s = x * r;
  float e = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(e, 0.5f * r, s);
}
// B: v_sqrt seed, correction with h = 0.5·rsq(x)
__device__ inline float seqB(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  float r = __builtin_amdgcn_rsqf(x);
  float e = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(e, 0.5f * r, s);
}
// C: A with two corrections
__device__ inline float seqC(float x) {
  float r = __builtin_amdgcn_rsqf(x);
  float s = x * r, h = 0.5f * r;
  float e = __builtin_fmaf(-s, s, x);
  s = __builtin_fmaf(e, h, s);
  e = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(e, h, s);
}
// D: Goldschmidt-style: refine h too
__device__ inline float seqD(float x) {
  float r = __builtin_amdgcn_rsqf(x);
  float s = x * r, h = 0.5f * r;
  float t = __builtin_fmaf(-h, s, 0.5f);
  s = __builtin_fmaf(s, t, s);
  h = __builtin_fmaf(h, t, h);
  float e = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(e, h, s);
}

__global__ void check(unsigned long long *bad) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  unsigned long long b[4] = {0, 0, 0, 0};
  for (uint64_t i = 0x00800000ull + tid; i < 0x7f800000ull; i += stride) {  // every positive normal x
    const float x = f((uint32_t)i), ref = sqrtf(x);
    const float c[4] = {seqA(x), seqB(x), seqC(x), seqD(x)};
    for (int v = 0; v < 4; v++)
      if (u(c[v]) != u(ref)) b[v]++;
  }
  for (int v = 0; v < 4; v++)
    if (b[v]) atomicAdd(&bad[v], b[v]);
}

int main() {
  unsigned long long *d, h[4];
  (void)hipMalloc(&d, sizeof(h));
  (void)hipMemset(d, 0, sizeof(h));
  check<<<4096, 256>>>(d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[4] = {"A rsq, x*r, 1 correction (5 instr)", "B v_sqrt + rsq, 1 correction (5 instr, 2 trans)", "C rsq, 2 corrections (7)",
                          "D rsq, Goldschmidt step + correction (8)"};
  for (int v = 0; v < 4; v++) printf("%s: %llu of 2^31-2^24 positive normal inputs differ from sqrtf(x)\n", names[v], h[v]);
  return 0;
}
