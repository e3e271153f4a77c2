// Probe (development aid): issue rate of vector instructions on gfx950 with W wavefronts per SIMD.
// Every kernel runs ITER iterations of 64 independent instructions of one kind (8 chains); the host reports
// cycles per wave-instruction per SIMD at the clock the run implies (s_memtime inside the kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define ITER 4000
#define REP8(X) X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITER; i++) {
    if (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (KIND == 1) { REP8(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n v_fma_f64 %6, %6, %6, %6\n v_fma_f64 %7, %7, %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));) }
    if (KIND == 2) { REP8(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %4\n v_xor_b32 %4, %4, %5\n v_xor_b32 %5, %5, %6\n v_xor_b32 %6, %6, %7\n v_xor_b32 %7, %7, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));) }
    if (KIND == 3) { REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 9\n v_alignbit_b32 %1, %1, %2, 9\n v_alignbit_b32 %2, %2, %3, 9\n v_alignbit_b32 %3, %3, %4, 9\n v_alignbit_b32 %4, %4, %5, 9\n v_alignbit_b32 %5, %5, %6, 9\n v_alignbit_b32 %6, %6, %7, 9\n v_alignbit_b32 %7, %7, %0, 9" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));) }
    if (KIND == 4) { REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (KIND == 5) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));) }
    if (KIND == 6) { REP8(asm volatile("v_pk_sub_i16 %0, %0, %1\n v_pk_sub_i16 %1, %1, %2\n v_pk_sub_i16 %2, %2, %3\n v_pk_sub_i16 %3, %3, %4\n v_pk_sub_i16 %4, %4, %5\n v_pk_sub_i16 %5, %5, %6\n v_pk_sub_i16 %6, %6, %7\n v_pk_sub_i16 %7, %7, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));) }
    if (KIND == 7) { REP8(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %4\n v_add_f64 %4, %4, %5\n v_add_f64 %5, %5, %6\n v_add_f64 %6, %6, %7\n v_add_f64 %7, %7, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));) }
    if (KIND == 8) { REP8(asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n v_lshlrev_b64 %3, 3, %3\n v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 3, %5\n v_lshlrev_b64 %6, 3, %6\n v_lshlrev_b64 %7, 3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));) }
    if (KIND == 9) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %4\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));) }
    if (KIND == 10) { REP8(asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %9\n v_cvt_f64_f32 %6, %10\n v_cvt_f64_f32 %7, %11" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
    if (KIND == 11) { REP8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));) }
    if (KIND == 12) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
static void run(const char *name, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd; // 256-thread blocks: 4 waves, one per SIMD of a CU
  float *out; unsigned long long *cyc;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&cyc, (size_t)blocks * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<KIND><<<blocks, 256>>>(out, cyc, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<KIND><<<blocks, 256>>>(out, cyc, 1.0f);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long *h = (unsigned long long *)malloc((size_t)blocks * 8);
  (void)hipMemcpy(h, cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (int b = 0; b < blocks; b++) mean += (double)h[b]; mean /= blocks;
  const double n = (double)ITER * 64; // wave-instructions per wave
  printf("%-14s %d waves/SIMD: %.2f in-kernel cycles per instruction per wave -> %.2f SIMD cycles per wave-instruction; wall %.3f ms -> %.2f GHz-equivalent\n", name, waves_per_simd, mean / n, mean / n / waves_per_simd, ms, mean / (ms * 1e6));
  (void)hipFree(out); (void)hipFree(cyc); free(h);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", w); run<1>("v_fma_f64", w); run<2>("v_xor_b32", w); run<3>("v_alignbit_b32", w); run<4>("v_max3_f32", w);
    run<5>("v_pk_fma_f32", w); run<6>("v_pk_sub_i16", w); run<7>("v_add_f64", w); run<8>("v_lshlrev_b64", w); run<9>("v_mul_lo_u32", w);
    run<10>("v_cvt_f64_f32", w); run<11>("v_mov_dpp", w); run<12>("cmp+cndmask", w);
  }
  return 0;
}
