// Which VALU instructions share an issue pipe on gfx950, and do a full-rate and a half-rate instruction overlap?
// 4 waves per SIMD (1024-thread blocks, one per CU), every instruction on its own register chain (8 chains per class), no VCC traffic
// unless the row says so.  Rows: cycles (at 2.4 GHz) per group per SIMD-wave-slot, i.e. time / (iterations x 64 groups x 4 waves).
// If two classes ran on separate pipes a group {a x F, b x H} would take max(a tF, b tH); on one pipe a tF + b tH.
#include <hip/hip_runtime.h>
#include <cstdio>
#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

#define F(i)  asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c))
#define FM(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(b))
#define H(i)  asm volatile("v_max_f32 %0, %0, %1" : "+v"(h[i]) : "v"(b))
#define T(i)  asm volatile("v_rcp_f32 %0, %0" : "+v"(h[i]))
#define S(i)  asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc")
typedef float f2_ __attribute__((ext_vector_type(2)));
#define PK(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[i]) : "v"(pk[((i) + 1) & 7]))

template <int OP>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float fs) {
  float f[8], h[8]; unsigned u[8]; f2_ pk[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { f[i] = threadIdx.x * 1e-3f + i; h[i] = f[i] + 0.5f; u[i] = threadIdx.x * 7u + i; pk[i] = {1.0f + f[i] * 1e-6f, 1.0f - f[i] * 1e-6f}; }
  const float b = 1.0001f, c = 0.5f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) { F(i); }
        if (OP == 1) { H(i); }
        if (OP == 2) { F(i); H(i); }
        if (OP == 3) { F(i); FM(i); H(i); }                         // 2 F + 1 H
        if (OP == 4) { F(i); FM(i); F((i + 3) & 7); H(i); }         // 3 F + 1 H
        if (OP == 5) { F(i); H(i); H((i + 3) & 7); }                // 1 F + 2 H
        if (OP == 6) { T(i); }
        if (OP == 7) { T(i); F(i); }
        if (OP == 8) { T(i); F(i); FM(i); F((i + 3) & 7); }         // 1 T + 3 F
        if (OP == 9) { T(i); H((i + 3) & 7); }                      // 1 T + 1 H
        if (OP == 10) { S(i); }
        if (OP == 11) { S(i); F(i); }
        if (OP == 12) { S(i); H(i); }
        if (OP == 60) { PK(i); }
        if (OP == 61) { PK(i); F(i); }
        if (OP == 62) { PK(i); F(i); FM(i); }
        if (OP == 63) { PK(i); F(i); PK((i + 3) & 7); FM(i); }        // 2 PK + 2 F: six flops
        if (OP == 64) { F(i); FM(i); F((i + 1) & 7); FM((i + 1) & 7); F((i + 2) & 7); FM((i + 2) & 7); }   // 6 F: the same six flops
        if (OP == 65) { PK(i); H(i); }
        // ---- which half-rate kinds hide a full-rate instruction the way v_max does ----
        if (OP == 70) { F(i); asm volatile("v_add_f32 %0, s20, %0" : "+v"(h[i]) :: "s20"); }            // 1 F + 1 add with an SGPR operand
        if (OP == 71) { F(i); FM(i); asm volatile("v_add_f32 %0, s20, %0" : "+v"(h[i]) :: "s20"); }     // 2 F + 1 ...
        if (OP == 72) asm volatile("v_mul_f32 %0, 0x3f8020c5, %0" : "+v"(f[i]));                         // 32-bit literal operand
        if (OP == 73) { F(i); asm volatile("v_mul_f32 %0, 0x3f8020c5, %0" : "+v"(h[i])); }
        if (OP == 74) { F(i); asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(h[i]) : "v"(u[i])); }
        if (OP == 75) { F(i); asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" :: "v"(h[i]), "v"(b) : "s20", "s21"); }
        if (OP == 76) { F(i); asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); }
        if (OP == 77) { F(i); H(i); FM(i); H((i + 3) & 7); F((i + 5) & 7); H((i + 6) & 7); }             // 3 F + 3 H alternating
        if (OP == 78) { F(i); FM(i); H(i); F((i + 3) & 7); FM((i + 3) & 7); H((i + 3) & 7); F((i + 5) & 7); FM((i + 5) & 7); H((i + 6) & 7); }   // 6 F + 3 H
        if (OP == 79) { F(i); FM(i); F((i + 3) & 7); FM((i + 3) & 7); F((i + 5) & 7); FM((i + 5) & 7); asm volatile("v_add_f32 %0, s20, %0" : "+v"(h[i]) :: "s20"); asm volatile("v_add_f32 %0, s20, %0" : "+v"(h[(i + 3) & 7]) :: "s20"); asm volatile("v_add_f32 %0, s20, %0" : "+v"(h[(i + 6) & 7]) :: "s20"); }   // 6 F + 3 SGPR-operand adds
        if (OP == 80) { F(i); FM(i); F((i + 3) & 7); FM((i + 3) & 7); F((i + 5) & 7); FM((i + 5) & 7); F((i + 1) & 7); FM((i + 2) & 7); F((i + 6) & 7); }   // 9 F: the same nine flops
        if (OP == 13) { F(i); F((i + 1) & 7); F((i + 2) & 7); F((i + 3) & 7); H(i); H((i + 4) & 7); }   // 4 F + 2 H, grouped
        // ---- classes of single instructions ----
        if (OP == 20) asm volatile("v_add_f32 %0, s20, %0" : "+v"(f[i]) :: "s20");                     // VOP2, SGPR src0
        if (OP == 21) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 22) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 23) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 25) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
        if (OP == 26) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "v"(b), "v"(c));
        if (OP == 27) asm volatile("v_rndne_f32 %0, %0" : "+v"(f[i]));
        if (OP == 28) asm volatile("v_trunc_f32 %0, %0" : "+v"(f[i]));
        if (OP == 29) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(f[i]));
        if (OP == 30) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" :: "v"(f[i]), "v"(b) : "s20", "s21");
        if (OP == 31) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(u[i]), "v"(u[(i + 1) & 7]) : "vcc");
        if (OP == 32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
        if (OP == 33) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
        if (OP == 34) asm volatile("v_mul_f32 %0, s20, %0" : "+v"(f[i]) :: "s20");
        if (OP == 35) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(f[i]) : "v"(b));
        if (OP == 36) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 37) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(u[i]));
        if (OP == 38) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(u[i]));
        if (OP == 39) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(u[i]));
        if (OP == 40) asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 41) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(f[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 42) asm volatile("v_subrev_f32 %0, s20, %0" : "+v"(f[i]) :: "s20");
        if (OP == 43) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(b) : );            // vcc never written in the loop
        if (OP == 44) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(f[i]), "+v"(h[i]) : "v"(f[(i + 1) & 7]), "v"(b) : "vcc");   // 1 cmp + 2 cndmask
        if (OP == 45) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(b), "v"(c));
        if (OP == 46) asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %0, %0, %2" : "+v"(f[i]) : "v"(b), "v"(c));
        if (OP == 47) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == 48) asm volatile("v_mov_b32 %0, s20" : "=v"(f[i]) :: "s20");
        if (OP == 49) asm volatile("v_add_u32 %0, s20, %0" : "+v"(u[i]) :: "s20");
        if (OP == 50) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(*(unsigned long long *)&u[i & 6]) : "v"(*(unsigned long long *)&u[(i + 2) & 6]));
        if (OP == 51) asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(b));
        if (OP == 52) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(b), "v"(c));
        if (OP == 53) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
        if (OP == 54) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[i]) : "v"(u[i]));
        if (OP == 55) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
        if (OP == 56) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 57) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
        if (OP == 58) asm volatile("v_cmp_class_f32 vcc, %0, %1" :: "v"(f[i]), "v"(u[i]) : "vcc");
        if (OP == 59) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc");
      }
    }
  }
  float s = fs; unsigned t = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { s += f[i] + h[i] + pk[i].x + pk[i].y; t += u[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)t;
}
template <int OP> int run(const char *name, float *d) {
  hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
  const int iters = 4000;
  float ms = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    HIPCHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, iters, 0.0f);
    HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
    float t; HIPCHECK(hipEventElapsedTime(&t, e0, e1));
    if (t < ms) ms = t;
  }
  printf("%-34s %8.3f ms -> %6.2f cycles@2.4GHz per group per wave-slot\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * 4));
  return 0;
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float *d; HIPCHECK(hipMalloc(&d, 256 * 1024 * 4));
  run<0>("F  (v_add_f32)", d); run<1>("H  (v_max_f32)", d); run<2>("1F + 1H", d); run<3>("2F + 1H", d); run<4>("3F + 1H", d); run<5>("1F + 2H", d); run<13>("4F + 2H grouped", d);
  run<60>("PK (v_pk_mul_f32)", d); run<61>("1PK + 1F", d); run<62>("1PK + 2F", d); run<63>("2PK + 2F (six flops)", d); run<64>("6F (the same six flops)", d); run<65>("1PK + 1H", d);
  run<6>("T  (v_rcp_f32)", d); run<7>("1T + 1F", d); run<8>("1T + 3F", d); run<9>("1T + 1H", d);
  run<10>("S  (s_add_u32)", d); run<11>("1S + 1F", d); run<12>("1S + 1H", d);
  run<70>("1F + 1 sgpr-operand add", d); run<71>("2F + 1 sgpr-operand add", d); run<72>("v_mul_f32 literal", d); run<73>("1F + 1 literal mul", d);
  run<74>("1F + 1 cvt", d); run<75>("1F + 1 cmp", d); run<76>("1F + 1 lshl_or", d); run<77>("3F + 3H alternating", d); run<78>("6F + 3H", d);
  run<79>("6F + 3 sgpr-operand adds", d); run<80>("9F", d);
  run<20>("v_add_f32 sgpr src0 (VOP2)", d); run<34>("v_mul_f32 sgpr src0", d); run<42>("v_subrev_f32 sgpr", d); run<49>("v_add_u32 sgpr", d); run<48>("v_mov_b32 sgpr", d); run<47>("v_mov_b32", d);
  run<21>("v_or_b32", d); run<22>("v_xor_b32", d); run<23>("v_sub_u32", d); run<24>("v_mul_u32_u24", d); run<25>("v_mad_u32_u24", d); run<36>("v_mul_lo_u32", d);
  run<26>("v_fmac_f32", d); run<35>("v_fma_f32 inline const", d); run<27>("v_rndne_f32", d); run<28>("v_trunc_f32", d); run<29>("v_cvt_i32_f32", d); run<39>("v_cvt_f32_i32", d); run<54>("v_cvt_f32_ubyte0", d);
  run<30>("v_cmp_lt_f32 -> sgpr pair", d); run<31>("v_cmp_lt_u32 vcc", d); run<58>("v_cmp_class_f32", d); run<43>("v_cndmask_b32 vcc (vcc const)", d); run<44>("1 cmp + 2 cndmask", d);
  run<32>("v_add3_u32", d); run<33>("v_and_or_b32", d); run<37>("v_lshrrev_b32", d); run<38>("v_ashrrev_i32", d); run<40>("v_max_u32", d); run<41>("v_ldexp_f32", d);
  run<45>("v_max3_f32", d); run<46>("2 x v_max_f32 (dependent)", d); run<52>("v_med3_f32", d); run<50>("v_lshl_add_u64", d); run<51>("sub + mul dependent (2 instr)", d);
  run<53>("v_exp_f32", d); run<55>("v_perm_b32", d); run<56>("v_alignbit_b32", d); run<57>("v_bfi_b32", d); run<59>("v_sub_co_u32", d);
  return 0;
}
