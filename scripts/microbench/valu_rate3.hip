// VALU issue rates of the integer / bit-field / select instructions a select-free spelling of the Mandelbulb iteration's quadrant
// and sign logic would use (round 3, follow-up of valu_rate2.hip).  Every stream is 64 inline-asm instructions per trip on 16
// independent registers, so the compiler cannot re-spell them; launches checked; cycles from s_memtime stamps of every wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate3 valu_rate3.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { K_BFI = 0, K_BFE, K_LSHLADD, K_ANDOR, K_ASHR, K_XOR, K_CNDMASK, K_CMP, K_CMPVCC_CND, K_SUBABS, K_MINABS, K_ADD3, K_FMA, K_BITOP3, K_MIN2, K_MINABS2, K_MAXABS, K_MINE64, K_BITOP3S, K_FMAS, K_MAX2, K_MUL2, K_COUNT };
static const char *kNames[] = {"v_bfi_b32 v,v,v", "v_bfe_i32 v,0,1", "v_lshl_add_u32 v,31,v", "v_and_or_b32 v,v,v", "v_ashrrev_i32 31,v (VOP2)",
                               "v_xor_b32 v,v (VOP2)", "v_cndmask_b32_e64 v,v,s[..] (mask fixed)", "v_cmp_gt_f32_e64 s[..],v,v (no reader)",
                               "v_cmp_gt_f32 vcc + v_cndmask vcc pairs", "v_sub_f32 |v|,|v| (VOP3)", "v_min_f32 1.0,|v| (VOP3)", "v_add3_u32 v,v,v",
                               "v_fma_f32 v,v,v", "v_bitop3_b32 v,v,v", "v_min_f32 v,v (VOP2)", "v_min_f32 v,|v| (VOP3)", "v_max_f32 |v|,|v| (VOP3)",
                               "v_min_f32_e64 v,v (VOP3, no modifier)", "v_bitop3_b32 v,v,s (mask in an SGPR)", "v_fma_f32 v,s,v (SGPR operand)", "v_max_f32 v,v (VOP2)", "v_mul_f32 v,v (VOP2)"};

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned long long *stamps, unsigned a, unsigned b, int trips) {
  unsigned v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = __builtin_bit_cast(unsigned, threadIdx.x * 0.001f + i * 0.37f + 1.0f) + (a & 1u);  // normal floats: no denormal / NaN operands in the float rows
  unsigned long long m = ((unsigned long long)a << 32) | (b * 0x9e3779b9u);
  unsigned long long acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int n = 0; n < trips; n++) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (KIND == K_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (KIND == K_BFE) asm volatile("v_bfe_i32 %0, %0, 0, 1" : "+v"(v[i]));
        else if (KIND == K_LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 31, %1" : "+v"(v[i]) : "v"(b));
        else if (KIND == K_ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (KIND == K_ASHR) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(v[i]));
        else if (KIND == K_XOR) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
        else if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "s"(m));
        else if (KIND == K_CMP) { unsigned long long o; asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(o) : "v"(v[i]), "v"(a)); if (i == 15 && r == 3) acc ^= o; }
        else if (KIND == K_CMPVCC_CND) { if (i & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : ); else asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(a) : "vcc"); }
        else if (KIND == K_SUBABS) asm volatile("v_sub_f32 %0, |%0|, |%1|" : "+v"(v[i]) : "v"(a));
        else if (KIND == K_MINABS) asm volatile("v_min_f32 %0, 1.0, |%0|" : "+v"(v[i]));
        else if (KIND == K_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (KIND == K_BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(v[i]) : "v"(a), "v"(b));
        else if (KIND == K_MIN2) asm volatile("v_min_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
        else if (KIND == K_MINABS2) asm volatile("v_min_f32 %0, %0, |%1|" : "+v"(v[i]) : "v"(a));
        else if (KIND == K_MAXABS) asm volatile("v_max_f32 %0, |%0|, |%1|" : "+v"(v[i]) : "v"(a));
        else if (KIND == K_MINE64) asm volatile("v_min_f32_e64 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        else if (KIND == K_BITOP3S) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(v[i]) : "v"(a), "s"(b));
        else if (KIND == K_FMAS) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(a), "v"(b));
        else if (KIND == K_MAX2) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[i]) : "v"(b));
        else asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  unsigned s = (unsigned)acc;
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
int run(int cus, int wps, unsigned *out, unsigned long long *dStamps, std::vector<unsigned long long> &h, hipEvent_t e0, hipEvent_t e1) {
  const int blocks = cus * wps;
  int trips = 20000;
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, dStamps, 0x3f7fbe77u, 0x3a83126fu, trips);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 0 && ms < 5.0f) trips = (int)(trips * 6.0f / (ms > 0.01f ? ms : 0.01f));
  }
  CHECK(hipMemcpy(h.data(), dStamps, sizeof(unsigned long long) * 2 * blocks * 4, hipMemcpyDeviceToHost));
  double cyc = 0, ticks = 0;
  for (int w = 0; w < blocks * 4; w++) { cyc += (double)h[2 * w]; ticks += (double)h[2 * w + 1]; }
  const double waves = blocks * 4.0, meanCyc = cyc / waves, mhz = 100.0 * cyc / ticks;
  const double instrPerSimd = (double)trips * 64 * wps;
  printf("| %-42s | %d | %8.3f | %5.0f | %6.2f | %6.2f |\n", kNames[KIND], wps, ms, mhz, meanCyc / instrPerSimd, ms * 1e-3 * mhz * 1e6 / instrPerSimd);
  return 0;
}

template <int KIND>
int sweep(int cus, unsigned *out, unsigned long long *dStamps, std::vector<unsigned long long> &h, hipEvent_t e0, hipEvent_t e1) {
  for (int wps : {2, 5, 8})
    if (run<KIND>(cus, wps, out, dStamps, h, e0, e1)) return 1;
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  unsigned *out;
  unsigned long long *dStamps;
  CHECK(hipMalloc(&out, sizeof(unsigned) * cus * 8 * 256));
  CHECK(hipMalloc(&dStamps, sizeof(unsigned long long) * 2 * cus * 8 * 4));
  std::vector<unsigned long long> h(2 * cus * 8 * 4);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("| stream | waves/SIMD | kernel ms | shader MHz | cycles per wave-instruction per SIMD (mean wave span) | same from the kernel time |\n|---|---|---|---|---|---|\n");
  if (sweep<K_FMA>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_BFI>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_BITOP3>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_BFE>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_LSHLADD>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_ANDOR>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_ADD3>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_ASHR>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_XOR>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_SUBABS>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_MINABS>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_CNDMASK>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_CMP>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_CMPVCC_CND>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_MIN2>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_MINE64>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_MINABS2>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_MAXABS>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_BITOP3S>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_FMAS>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_MAX2>(cus, out, dStamps, h, e0, e1)) return 1;
  if (sweep<K_MUL2>(cus, out, dStamps, h, e0, e1)) return 1;
  return 0;
}
