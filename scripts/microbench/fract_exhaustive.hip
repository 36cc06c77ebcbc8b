// Exhaustive check on the device: is v_fract_f32(x) == min(x − floor(x), 1 − 2^-24) for every binary32 x?
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off scripts/microbench/fract_exhaustive.hip -o scripts/microbench/fract_exhaustive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
__device__ inline float f(uint32_t b) { float y; memcpy(&y, &b, 4); return y; }
__device__ inline uint32_t u(float y) { uint32_t b; memcpy(&b, &y, 4); return b; }
__global__ void check(unsigned long long *bad, uint32_t *ex) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  unsigned long long b0 = 0, b1 = 0;
  for (uint64_t i = tid; i < (1ull << 32); i += stride) {
    const float x = f((uint32_t)i);
    const float hw = __builtin_amdgcn_fractf(x);
    const float d = x - __builtin_floorf(x);
    const float ref = (d >= 1.0f) ? 0.99999994f : d;  // NaN stays NaN
    const bool same = (u(hw) == u(ref)) || (hw != hw && ref != ref);
    if (!same) { b0++; if (atomicAdd(&ex[0], 1u) < 8) { uint32_t k = atomicAdd(&ex[1], 1u); if (k < 8) { ex[2 + 3 * k] = (uint32_t)i; ex[3 + 3 * k] = u(hw); ex[4 + 3 * k] = u(ref); } } }
    if (!(u(hw) == u(d) || (hw != hw && d != d))) b1++;
  }
  if (b0) atomicAdd(&bad[0], b0);
  if (b1) atomicAdd(&bad[1], b1);
}
int main() {
  unsigned long long *d, h[2]; uint32_t *e, he[32];
  (void)hipMalloc(&d, sizeof(h)); (void)hipMemset(d, 0, sizeof(h));
  (void)hipMalloc(&e, sizeof(he)); (void)hipMemset(e, 0, sizeof(he));
  check<<<4096, 256>>>(d, e);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipMemcpy(he, e, sizeof(he), hipMemcpyDeviceToHost);
  printf("v_fract_f32 vs min(x - floor(x), 1 - 2^-24): %llu of 2^32 differ; vs plain x - floor(x): %llu differ\n", h[0], h[1]);
  for (uint32_t k = 0; k < he[1] && k < 8; k++) printf("  x=%08x hw=%08x ref=%08x\n", he[2 + 3 * k], he[3 + 3 * k], he[4 + 3 * k]);
  return 0;
}
