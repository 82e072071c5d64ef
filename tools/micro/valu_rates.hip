// Issue rate of common VALU instructions on gfx950: 4 waves per SIMD, 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
  float a[8]; float2_ p[8]; unsigned u[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 1e-3f + i; p[i] = {a[i], a[i] + 1.f}; u[i] = threadIdx.x * 7u + i; }
  const float b = 1.0001f, c = 0.5f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
        if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (OP == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (OP == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 8) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
        if (OP == 10) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
        if (OP == 11) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 12) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
        if (OP == 13) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
        if (OP == 14) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 15) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(u[i]));
        if (OP == 16) asm volatile("s_nop 0");
        if (OP == 32) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(b) : "vcc");
        if (OP == 33) asm volatile("v_cmp_lt_f32 s[20:21], %1, %2\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(b) : "s20", "s21");
        if (OP == 34) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_add_f32 %0, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(b) : "vcc");
        if (OP == 35) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(b) : "vcc");
        if (OP == 18) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b));
        if (OP == 19) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 20) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 21) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 22) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
        if (OP == 23) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
        if (OP == 24) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 25) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
        if (OP == 26) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
        if (OP == 27) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 28) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        if (OP == 29) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(c));
        if (OP == 30) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a[i]));
        if (OP == 31) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc");
        if (OP == 36) asm volatile("v_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
        if (OP == 37) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 38) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 39) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
        if (OP == 40) asm volatile("v_mul_f32 %0, %0, %1\n v_min_f32 %2, %2, %3" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(b), "v"(c));
        if (OP == 41) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_add_f32 %2, %2, %3" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(b), "v"(c));
        if (OP == 17) asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]" : : : "s20", "s21", "scc");   // scc: the loop branch reads it
      }
    }
  }
  float s = 0; unsigned t = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { s += a[i] + p[i].x + p[i].y; t += u[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)t;
}
template <int OP> int run(const char *name, float *d) {
  hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
  const int iters = 8000;
  float ms = 1e30f;
  for (int rep = 0; rep < 4; rep++) {          // first launch warms up; keep the fastest
    HIPCHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, iters);
    HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
    float t; HIPCHECK(hipEventElapsedTime(&t, e0, e1));
    if (t < ms) ms = t;
  }
  const double per_simd = (double)iters * 64 * 4;      // wave-instructions per SIMD (4 waves each)
  printf("%-16s %.3f ms -> %.2f ns-cycles@2.4GHz per wave-instruction per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
  return 0;
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float *d; HIPCHECK(hipMalloc(&d, 256 * 1024 * 4));
  run<0>("v_fma_f32", d); run<1>("v_mul_f32", d); run<2>("v_add_f32", d); run<3>("v_pk_fma_f32", d); run<4>("v_pk_mul_f32", d);
  run<5>("v_pk_add_f32", d); run<6>("v_and_b32", d); run<7>("v_add_u32", d); run<8>("v_min_f32", d); run<9>("v_cndmask_b32", d);
  run<10>("v_cmp_lt_f32", d); run<11>("v_lshl_or_b32", d); run<12>("v_cvt_f32_u32", d); run<13>("v_fract_f32", d); run<14>("v_rcp_f32", d);
  run<15>("v_bfe_u32", d); run<16>("s_nop 0", d); run<17>("s_and_b64", d);
  run<18>("v_cndmask e64 sgpr", d); run<19>("v_max_f32", d); run<20>("v_sub_f32", d); run<21>("v_mov_b32", d); run<22>("v_or3_b32", d);
  run<23>("v_lshlrev_b32", d); run<24>("v_min_u32", d); run<25>("v_floor_f32", d); run<26>("v_cvt_u32_f32", d); run<27>("v_lshl_add_u32", d);
  run<28>("v_sqrt_f32", d); run<29>("v_fma_f32 sgpr", d); run<30>("v_mul_f32 const", d); run<31>("v_addc_co_u32", d); run<0>("v_fma_f32 again", d);
  run<36>("v_swap_b32", d); run<37>("v_max3_f32", d); run<38>("v_med3_f32", d); run<39>("v_rsq_f32", d); run<40>("mul + min (2 instr)", d); run<41>("cndmask + add (2 instr)", d);
  run<32>("cmp vcc+cndmask (2 instr)", d); run<33>("cmp sgpr+cndmask64 (2)", d); run<34>("cmp,add,cndmask (3)", d); run<35>("cmp,nop1,cndmask", d);
  return 0;
}
