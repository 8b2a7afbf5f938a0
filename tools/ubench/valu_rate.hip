// VALU issue-rate microbenchmark (gfx950): cycles per wave64 instruction per SIMD for the integer / float ops the
// front-end kernels are made of.  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) k(int iters, unsigned* out, unsigned seed) {
  unsigned a[8];
  float f[8];
  double d[4];
  for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 2654435761u + i + seed; f[i] = (float)a[i] * 1e-9f; }
  for (int i = 0; i < 4; i++) d[i] = (double)a[i] * 1e-9;
  const unsigned m = seed | 3u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {   // 8 independent chains x 16 = 128 ops per iteration
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 1) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 3) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 5) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
        if (OP == 6) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
        if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i+1)&7]));
        if (OP == 8) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 9) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 10) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 11) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == 12) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == 13) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
        if (OP == 14) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
        if (OP == 15) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
        if (OP == 16) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == 17) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == 18) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == 19) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (OP == 20) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[i]) : "v"(a[i]));
        if (OP == 21) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
        if (OP == 22) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
        if (OP == 23) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 24) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 25) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(a[i]));
        if (OP == 26) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 27) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 28) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 29) asm volatile("v_cmp_gt_u32 s[4:5], %0, %1" : : "v"(a[i]), "v"(m) : "s4", "s5");
        if (OP == 30) asm volatile("v_cmp_gt_f32 s[4:5], %0, %1" : : "v"(f[i]), "v"(f[(i+1)&7]) : "s4", "s5");
        if (OP == 31) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
        if (OP == 32) asm volatile("v_cndmask_b32 %0, %0, %1, s[6:7]" : "+v"(a[i]) : "v"(m));
        if (OP == 33) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i&3]) : "v"(d[(i + 1) & 3]));
        if (OP == 34) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i&3]) : "v"(d[(i + 1) & 3]));
        if (OP == 35) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 3]) : "v"(d[(i + 1) & 3]));
        if (OP == 36) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i & 3]) : "v"(d[(i + 1) & 3]));
        if (OP == 37) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i & 3]) : "v"(d[(i + 1) & 3]));
        if (OP == 38) asm volatile("v_readlane_b32 s8, %0, 3" : : "v"(a[i]) : "s8");
        if (OP == 39) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
        if (OP == 40) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(m));
        if (OP == 41) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 42) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0" : "+v"(a[i]) : "v"(m));
        if (OP == 43) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
        if (OP == 44) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
        if (OP == 45) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(d[i&3]) : "v"(m), "v"(a[i]) : "s10","s11");
        if (OP == 46) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 47) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m));
      }
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + (unsigned)f[i];
  for (int i = 0; i < 4; i++) s += (unsigned)d[i];
  if (s == 0x12345u) out[0] = s;
}

template <int OP>
void run(const char* name, int wavesPerSimd) {
  unsigned* out;
  hipMalloc(&out, 4);
  const int iters = 2000;
  const int blocks = 256 * wavesPerSimd;   // 256 CUs x (waves per SIMD) blocks of 4 waves
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, 10, out, 1u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, out, 1u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instrPerSimd = (double)iters * 128 * wavesPerSimd;   // wave-instructions issued by one SIMD
  const double cyc = ms * 1e-3 * 2.4e9 / instrPerSimd;
  printf("%-22s waves/SIMD %d  %.3f ms  %.2f cycles per wave64 instruction (at 2.4 GHz)\n", name, wavesPerSimd, ms, cyc);
  hipFree(out);
}

int main() {
  const int w = 8;
  run<0>("v_add_u32", w);
  run<1>("v_sub_u32", w);
  run<2>("v_and_b32", w);
  run<3>("v_or_b32", w);
  run<4>("v_xor_b32", w);
  run<5>("v_lshlrev_b32", w);
  run<6>("v_lshrrev_b32", w);
  run<7>("v_mov_b32", w);
  run<8>("v_max_i32", w);
  run<9>("v_max_u32", w);
  run<10>("v_min3_i32", w);
  run<11>("v_max_f32", w);
  run<12>("v_min_f32", w);
  run<13>("v_min3_f32", w);
  run<14>("v_max3_f32", w);
  run<15>("v_med3_f32", w);
  run<16>("v_sub_f32", w);
  run<17>("v_mul_f32", w);
  run<18>("v_fma_f32", w);
  run<19>("v_mac_f32/fmac", w);
  run<20>("v_cvt_f32_ubyte1", w);
  run<21>("v_cvt_f32_u32", w);
  run<22>("v_cvt_u32_f32", w);
  run<23>("v_add3_u32", w);
  run<24>("v_and_or_b32", w);
  run<25>("v_bfe_u32", w);
  run<26>("v_bcnt_u32_b32", w);
  run<27>("v_mul_u32_u24", w);
  run<28>("v_mul_lo_u32", w);
  run<29>("v_cmp_gt_u32->s[4:5]", w);
  run<30>("v_cmp_gt_f32->s[4:5]", w);
  run<31>("v_cmp_gt_u32 vcc e32", w);
  run<32>("v_cndmask s[6:7]", w);
  run<33>("v_pk_add_f32", w);
  run<34>("v_pk_fma_f32", w);
  run<35>("v_add_f64", w);
  run<36>("v_fma_f64", w);
  run<37>("v_mul_f64", w);
  run<38>("v_readlane", w);
  run<39>("v_mbcnt_lo", w);
  run<40>("v_alignbyte", w);
  run<41>("v_perm_b32", w);
  run<42>("v_sub_u32_sdwa", w);
  run<43>("v_rcp_f32", w);
  run<44>("v_sqrt_f32", w);
  run<45>("v_mad_u64_u32", w);
  run<46>("v_lshl_add_u32", w);
  run<47>("v_add_u32 dpp shr1", w);
  return 0;
}
