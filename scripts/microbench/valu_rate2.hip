// VALU issue-rate microbenchmark for gfx950, round 3 (replaces valu_rate.hip, whose 8-waves row never launched: 2048-thread
// blocks, launch error unchecked).  Every launch is checked, blocks are 256 threads (one wave per SIMD), W blocks per CU give
// W waves per SIMD, kernels run >= 5 ms, and the cycles are the chip's own: every wave stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) around its loop, so the result does not depend on an assumed clock.
//
//   cycles per wave-instruction per SIMD = (mean s_memtime span of a wave) / (instructions the W waves of its SIMD issue)
//
// Streams: fma (16 independent chains), fma1 (ONE dependent chain), muladd (v_mul / v_add alternating), cmpsel (v_cmp +
// v_cndmask pairs), trans (v_rcp_f32, independent), mix = the class mix of one Mandelbulb iteration (profiles/r02_b_isa_budget.md:
// 53 fma, 29 mul/add, 50 compare / select / integer, 8 + 4 division-fixup / transcendental, 4 convert per 148).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o valu_rate2 valu_rate2.hip (the product's flags:
// no packed-f32 vectorisation) ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { K_FMA = 0, K_FMA1, K_MULADD, K_CMPSEL, K_TRANS, K_MIX, K_COUNT };
static const char *kNames[] = {"v_fma_f32 x16 independent", "v_fma_f32 one dependent chain", "v_mul_f32 / v_add_f32", "v_cmp + v_cndmask",
                               "v_rcp_f32 independent", "bulb-iteration class mix"};
// wave-instructions of one loop trip, per kind (checked against the disassembly: see the table printed by --counts)
static const int kPerTrip[] = {64, 64, 64, 64, 64, 83};  // VALU only; the cmpsel stream also carries 31 s_nop (one per v_cmp → v_cndmask pair: the VCC hazard)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *stamps, float a, float b, int trips) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 0.001f + i * 0.37f + 1.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int n = 0; n < trips; n++) {
    if (KIND == K_FMA) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = __builtin_fmaf(v[i], a, b);
    } else if (KIND == K_FMA1) {
#pragma unroll
      for (int r = 0; r < 64; r++) v[0] = __builtin_fmaf(v[0], a, b);
    } else if (KIND == K_MULADD) {
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) { v[i] = v[i] * a; v[i] = v[i] + b; }
    } else if (KIND == K_CMPSEL) {
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = (v[i] > a) ? b : v[i];  // v_cmp (→ vcc) + v_cndmask
    } else if (KIND == K_TRANS) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = __builtin_amdgcn_rcpf(v[i]);
    } else {
      // half of an iteration's 148: 26 fma, 14 mul/add, 12 compare+select pairs (24), 4 integer, 2 transcendental, 2 convert, 2 fixup-like fma
#pragma unroll
      for (int i = 0; i < 13; i++) { v[i] = __builtin_fmaf(v[i], a, b); v[(i + 1) & 15] = __builtin_fmaf(v[(i + 1) & 15], v[i], a); }
#pragma unroll
      for (int i = 0; i < 7; i++) { v[i] = v[i] * a; v[i + 7] = v[i + 7] + v[i]; }
#pragma unroll
      for (int i = 0; i < 12; i++) v[i] = (v[i] > a) ? b : v[i];
#pragma unroll
      for (int i = 0; i < 4; i++) v[i] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, v[i]) & 0x7fffffffu) ^ (__builtin_bit_cast(unsigned, v[i + 4]) << 30));
      v[12] = __builtin_amdgcn_rcpf(v[12]);
      v[13] = __builtin_amdgcn_sqrtf(v[13]);
      v[14] = __builtin_rintf(v[14]);
      v[15] = (float)(int)v[15];
      v[0] = __builtin_fmaf(-v[12], v[13], v[0]);
      v[1] = __builtin_fmaf(v[14], v[15], v[1]);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

template <int KIND>
int run(int cus, int wps, float *out, unsigned long long *dStamps, std::vector<unsigned long long> &h, hipEvent_t e0, hipEvent_t e1) {
  const int blocks = cus * wps;  // 256-thread blocks: one wave on each SIMD of a CU; wps blocks per CU = wps waves per SIMD
  int trips = 20000;
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, dStamps, 0.999f, 0.001f, trips);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 0 && ms < 5.0f) trips = (int)(trips * 6.0f / (ms > 0.01f ? ms : 0.01f));  // aim at >= 5 ms
  }
  CHECK(hipMemcpy(h.data(), dStamps, sizeof(unsigned long long) * 2 * blocks * 4, hipMemcpyDeviceToHost));
  double cyc = 0, ticks = 0;
  for (int w = 0; w < blocks * 4; w++) { cyc += (double)h[2 * w]; ticks += (double)h[2 * w + 1]; }
  const double waves = blocks * 4.0, meanCyc = cyc / waves, mhz = 100.0 * cyc / ticks;
  const double instrPerSimd = (double)trips * kPerTrip[KIND] * wps;
  printf("| %-30s | %d | %8.3f | %7.0f | %6.2f | %6.2f |\n", kNames[KIND], wps, ms, mhz, meanCyc / instrPerSimd,
         ms * 1e-3 * mhz * 1e6 / instrPerSimd);
  return 0;
}

int main(int argc, char **argv) {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float *out;
  unsigned long long *dStamps;
  CHECK(hipMalloc(&out, sizeof(float) * cus * 8 * 256));
  CHECK(hipMalloc(&dStamps, sizeof(unsigned long long) * 2 * cus * 8 * 4));
  std::vector<unsigned long long> h(2 * cus * 8 * 4);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("CUs %d, nominal clock %d kHz\n", cus, p.clockRate);
  printf("| stream | waves/SIMD | kernel ms | shader MHz (s_memtime / s_memrealtime) | cycles per wave-instr per SIMD (wave spans) | same from the kernel time |\n|---|---|---|---|---|---|\n");
  for (int wps = 1; wps <= 8; wps = (wps < 4 ? wps * 2 : wps + 1)) {
    if (run<K_FMA>(cus, wps, out, dStamps, h, e0, e1)) return 1;
    if (run<K_FMA1>(cus, wps, out, dStamps, h, e0, e1)) return 1;
    if (run<K_MULADD>(cus, wps, out, dStamps, h, e0, e1)) return 1;
    if (run<K_CMPSEL>(cus, wps, out, dStamps, h, e0, e1)) return 1;
    if (run<K_TRANS>(cus, wps, out, dStamps, h, e0, e1)) return 1;
    if (run<K_MIX>(cus, wps, out, dStamps, h, e0, e1)) return 1;
  }
  return 0;
}
