// VALU issue-rate microbenchmark for gfx950: plain v_fma_f32 vs packed v_pk_fma_f32, by waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 4096;  // loop trips, 16 independent chains each

__global__ void k_fma(float *out, float a, float b) {
  float v[16];
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 0.001f + i;
  for (int n = 0; n < N; n++) {
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = __builtin_fmaf(v[i], a, b);
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pk(float *out, float a, float b) {
  f2 v[16];
  for (int i = 0; i < 16; i++) v[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
  f2 A = f2{a, a * 1.0001f}, B = f2{b, b * 0.9999f};
  for (int n = 0; n < N; n++) {
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = __builtin_elementwise_fma(v[i], A, B);
  }
  f2 s = f2{0, 0};
  for (int i = 0; i < 16; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
// EXEC-mask sensitivity: the same fma loop with only some lanes of every wave alive.  mode 0 all 64 lanes, 1 lanes 0-31,
// 2 even lanes, 3 lanes 0-15.  If a pass whose 32 lanes are all masked off were skipped, mode 1 would run ~2x faster.
__global__ void k_fma_masked(float *out, float a, float b, int mode) {
  const int lane = threadIdx.x & 63;
  if (mode == 1 && lane >= 32) return;
  if (mode == 2 && (lane & 1)) return;
  if (mode == 3 && lane >= 16) return;
  float v[16];
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 0.001f + i;
  for (int n = 0; n < N; n++) {
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = __builtin_fmaf(v[i], a, b);
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  float *out; CHECK(hipMalloc(&out, sizeof(float) * cus * 8 * 1024));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("CUs %d clock %d kHz\n", cus, p.clockRate);
  for (int wps = 1; wps <= 8; wps *= 2) {       // waves per SIMD
    for (int kind = 0; kind < 2; kind++) {
      dim3 grid(cus), block(64 * 4 * wps);       // one block per CU, 4*wps waves
      for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        if (kind == 0) hipLaunchKernelGGL(k_fma, grid, block, 0, 0, out, 0.999f, 0.001f);
        else hipLaunchKernelGGL(k_pk, grid, block, 0, 0, out, 0.999f, 0.001f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      }
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      double instr_per_simd = (double)N * 16 * wps;           // wave-instructions issued on each SIMD
      double cyc = ms * 1e-3 * 2.4e9;
      double flops = (double)cus * 4 * wps * 64 * N * 16 * 2 * (kind ? 2 : 1);
      printf("%s waves/SIMD %d: %.3f ms  -> %.2f cycles@2.4GHz per wave-instr per SIMD, %.1f TFLOP/s\n",
             kind ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, cyc / instr_per_simd, flops / (ms * 1e-3) / 1e12);
    }
  }
  for (int mode = 0; mode < 4; mode++) {
    const int wps = 4;
    dim3 grid(cus), block(64 * 4 * wps);
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_fma_masked, grid, block, 0, 0, out, 0.999f, 0.001f, mode);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("exec-mask mode %d (0 all, 1 low half, 2 even lanes, 3 low quarter), 4 waves/SIMD: %.3f ms\n", mode, ms);
  }
  return 0;
}
