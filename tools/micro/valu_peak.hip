// VALU issue calibration for the traversal kernels' roofline (gfx950): shader cycles one SIMD spends per wave64 instruction of
// each class.  Every statement is 8 instructions with 8 DIFFERENT destination registers and sources no neighbour writes (no
// RAW / WAW stalls); 8 waves per SIMD (2048 blocks of 256 threads, all resident: 64 VGPRs, no LDS) issue nothing else.  The
// stream is inline asm (the compiler can neither pack, fold nor drop it).  Cycles come from s_memtime around the loop of the
// block (all 8 waves of a SIMD run the same loop side by side) and are cross-checked by wall time x the clock measured with
// s_memrealtime (100 MHz); stamps go to a buffer nothing else reads.  Run under rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
// SQ_BUSY_CYCLES GRBM_GUI_ACTIVE to calibrate those counters' units on a known instruction stream (one kernel per class).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_peak.hip -o tools/micro/valu_peak.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256, 8) void k_class(unsigned long long* stamps, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    float d0 = lane, d1 = lane + 1.f, d2 = lane + 2.f, d3 = lane + 3.f, d4 = lane + 4.f, d5 = lane + 5.f, d6 = lane + 6.f, d7 = lane + 7.f;
    const float s0 = 1.0001f + lane, s1 = 0.5f;
    double e0 = lane, e1 = lane + 1., e2 = lane + 2., e3 = lane + 3., e4 = lane + 4., e5 = lane + 5., e6 = lane + 6., e7 = lane + 7.;
    const double t0d = 1.0001 + lane, t1d = 0.5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %8, %9, %8\nv_fma_f32 %1, %8, %9, %8\nv_fma_f32 %2, %8, %9, %8\nv_fma_f32 %3, %8, %9, %8\nv_fma_f32 %4, %8, %9, %8\nv_fma_f32 %5, %8, %9, %8\nv_fma_f32 %6, %8, %9, %8\nv_fma_f32 %7, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 1) asm volatile("v_mul_f32_e32 %0, %8, %9\nv_mul_f32_e32 %1, %8, %9\nv_mul_f32_e32 %2, %8, %9\nv_mul_f32_e32 %3, %8, %9\nv_mul_f32_e32 %4, %8, %9\nv_mul_f32_e32 %5, %8, %9\nv_mul_f32_e32 %6, %8, %9\nv_mul_f32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 2) asm volatile("v_add_f32_e32 %0, %8, %9\nv_add_f32_e32 %1, %8, %9\nv_add_f32_e32 %2, %8, %9\nv_add_f32_e32 %3, %8, %9\nv_add_f32_e32 %4, %8, %9\nv_add_f32_e32 %5, %8, %9\nv_add_f32_e32 %6, %8, %9\nv_add_f32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 3) asm volatile("v_mov_b32_e32 %0, %8\nv_mov_b32_e32 %1, %8\nv_mov_b32_e32 %2, %8\nv_mov_b32_e32 %3, %8\nv_mov_b32_e32 %4, %8\nv_mov_b32_e32 %5, %8\nv_mov_b32_e32 %6, %8\nv_mov_b32_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 4) asm volatile("v_add_u32_e32 %0, %8, %9\nv_add_u32_e32 %1, %8, %9\nv_add_u32_e32 %2, %8, %9\nv_add_u32_e32 %3, %8, %9\nv_add_u32_e32 %4, %8, %9\nv_add_u32_e32 %5, %8, %9\nv_add_u32_e32 %6, %8, %9\nv_add_u32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 5) asm volatile("v_lshlrev_b32_e32 %0, 3, %8\nv_lshlrev_b32_e32 %1, 3, %8\nv_lshlrev_b32_e32 %2, 3, %8\nv_lshlrev_b32_e32 %3, 3, %8\nv_lshlrev_b32_e32 %4, 3, %8\nv_lshlrev_b32_e32 %5, 3, %8\nv_lshlrev_b32_e32 %6, 3, %8\nv_lshlrev_b32_e32 %7, 3, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 6) asm volatile("v_and_b32_e32 %0, %8, %9\nv_and_b32_e32 %1, %8, %9\nv_and_b32_e32 %2, %8, %9\nv_and_b32_e32 %3, %8, %9\nv_and_b32_e32 %4, %8, %9\nv_and_b32_e32 %5, %8, %9\nv_and_b32_e32 %6, %8, %9\nv_and_b32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 7) asm volatile("v_lshl_add_u32 %0, %8, 2, %9\nv_lshl_add_u32 %1, %8, 2, %9\nv_lshl_add_u32 %2, %8, 2, %9\nv_lshl_add_u32 %3, %8, 2, %9\nv_lshl_add_u32 %4, %8, 2, %9\nv_lshl_add_u32 %5, %8, 2, %9\nv_lshl_add_u32 %6, %8, 2, %9\nv_lshl_add_u32 %7, %8, 2, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 8) asm volatile("v_cvt_f32_ubyte1_e32 %0, %8\nv_cvt_f32_ubyte1_e32 %1, %8\nv_cvt_f32_ubyte1_e32 %2, %8\nv_cvt_f32_ubyte1_e32 %3, %8\nv_cvt_f32_ubyte1_e32 %4, %8\nv_cvt_f32_ubyte1_e32 %5, %8\nv_cvt_f32_ubyte1_e32 %6, %8\nv_cvt_f32_ubyte1_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 9) asm volatile("v_cvt_f32_u32_e32 %0, %8\nv_cvt_f32_u32_e32 %1, %8\nv_cvt_f32_u32_e32 %2, %8\nv_cvt_f32_u32_e32 %3, %8\nv_cvt_f32_u32_e32 %4, %8\nv_cvt_f32_u32_e32 %5, %8\nv_cvt_f32_u32_e32 %6, %8\nv_cvt_f32_u32_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 10) asm volatile("v_max_f32_e32 %0, %8, %9\nv_max_f32_e32 %1, %8, %9\nv_max_f32_e32 %2, %8, %9\nv_max_f32_e32 %3, %8, %9\nv_max_f32_e32 %4, %8, %9\nv_max_f32_e32 %5, %8, %9\nv_max_f32_e32 %6, %8, %9\nv_max_f32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 11) asm volatile("v_max3_f32 %0, %8, %9, %8\nv_max3_f32 %1, %8, %9, %8\nv_max3_f32 %2, %8, %9, %8\nv_max3_f32 %3, %8, %9, %8\nv_max3_f32 %4, %8, %9, %8\nv_max3_f32 %5, %8, %9, %8\nv_max3_f32 %6, %8, %9, %8\nv_max3_f32 %7, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 12) asm volatile("v_cndmask_b32_e64 %0, %8, %9, s[6:7]\nv_cndmask_b32_e64 %1, %8, %9, s[6:7]\nv_cndmask_b32_e64 %2, %8, %9, s[6:7]\nv_cndmask_b32_e64 %3, %8, %9, s[6:7]\nv_cndmask_b32_e64 %4, %8, %9, s[6:7]\nv_cndmask_b32_e64 %5, %8, %9, s[6:7]\nv_cndmask_b32_e64 %6, %8, %9, s[6:7]\nv_cndmask_b32_e64 %7, %8, %9, s[6:7]" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1) : "vcc", "s6", "s7", "s8", "s9");
        if (KIND == 13) asm volatile("v_cmp_lt_f32_e64 s[8:9], %8, %9\nv_cmp_lt_f32_e64 s[8:9], %8, %9\nv_cmp_lt_f32_e64 s[8:9], %8, %9\nv_cmp_lt_f32_e64 s[8:9], %8, %9\nv_cmp_lt_f32_e64 s[8:9], %8, %9\nv_cmp_lt_f32_e64 s[8:9], %8, %9\nv_cmp_lt_f32_e64 s[8:9], %8, %9\nv_cmp_lt_f32_e64 s[8:9], %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1) : "vcc", "s6", "s7", "s8", "s9");
        if (KIND == 14) asm volatile("v_perm_b32 %0, %8, %9, %8\nv_perm_b32 %1, %8, %9, %8\nv_perm_b32 %2, %8, %9, %8\nv_perm_b32 %3, %8, %9, %8\nv_perm_b32 %4, %8, %9, %8\nv_perm_b32 %5, %8, %9, %8\nv_perm_b32 %6, %8, %9, %8\nv_perm_b32 %7, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 15) asm volatile("v_bfe_u32 %0, %8, 8, 8\nv_bfe_u32 %1, %8, 8, 8\nv_bfe_u32 %2, %8, 8, 8\nv_bfe_u32 %3, %8, 8, 8\nv_bfe_u32 %4, %8, 8, 8\nv_bfe_u32 %5, %8, 8, 8\nv_bfe_u32 %6, %8, 8, 8\nv_bfe_u32 %7, %8, 8, 8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 16) asm volatile("v_mul_lo_u32 %0, %8, %9\nv_mul_lo_u32 %1, %8, %9\nv_mul_lo_u32 %2, %8, %9\nv_mul_lo_u32 %3, %8, %9\nv_mul_lo_u32 %4, %8, %9\nv_mul_lo_u32 %5, %8, %9\nv_mul_lo_u32 %6, %8, %9\nv_mul_lo_u32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 17) asm volatile("v_rcp_f32_e32 %0, %8\nv_rcp_f32_e32 %1, %8\nv_rcp_f32_e32 %2, %8\nv_rcp_f32_e32 %3, %8\nv_rcp_f32_e32 %4, %8\nv_rcp_f32_e32 %5, %8\nv_rcp_f32_e32 %6, %8\nv_rcp_f32_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 18) asm volatile("v_sqrt_f32_e32 %0, %8\nv_sqrt_f32_e32 %1, %8\nv_sqrt_f32_e32 %2, %8\nv_sqrt_f32_e32 %3, %8\nv_sqrt_f32_e32 %4, %8\nv_sqrt_f32_e32 %5, %8\nv_sqrt_f32_e32 %6, %8\nv_sqrt_f32_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 19) asm volatile("v_div_scale_f32 %0, vcc, %8, %9, %8\nv_div_scale_f32 %1, vcc, %8, %9, %8\nv_div_scale_f32 %2, vcc, %8, %9, %8\nv_div_scale_f32 %3, vcc, %8, %9, %8\nv_div_scale_f32 %4, vcc, %8, %9, %8\nv_div_scale_f32 %5, vcc, %8, %9, %8\nv_div_scale_f32 %6, vcc, %8, %9, %8\nv_div_scale_f32 %7, vcc, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1) : "vcc", "s6", "s7", "s8", "s9");
        if (KIND == 20) asm volatile("v_div_fmas_f32 %0, %8, %9, %8\nv_div_fmas_f32 %1, %8, %9, %8\nv_div_fmas_f32 %2, %8, %9, %8\nv_div_fmas_f32 %3, %8, %9, %8\nv_div_fmas_f32 %4, %8, %9, %8\nv_div_fmas_f32 %5, %8, %9, %8\nv_div_fmas_f32 %6, %8, %9, %8\nv_div_fmas_f32 %7, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 21) asm volatile("v_div_fixup_f32 %0, %8, %9, %8\nv_div_fixup_f32 %1, %8, %9, %8\nv_div_fixup_f32 %2, %8, %9, %8\nv_div_fixup_f32 %3, %8, %9, %8\nv_div_fixup_f32 %4, %8, %9, %8\nv_div_fixup_f32 %5, %8, %9, %8\nv_div_fixup_f32 %6, %8, %9, %8\nv_div_fixup_f32 %7, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 22) asm volatile("v_fma_f64 %0, %8, %9, %8\nv_fma_f64 %1, %8, %9, %8\nv_fma_f64 %2, %8, %9, %8\nv_fma_f64 %3, %8, %9, %8\nv_fma_f64 %4, %8, %9, %8\nv_fma_f64 %5, %8, %9, %8\nv_fma_f64 %6, %8, %9, %8\nv_fma_f64 %7, %8, %9, %8" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(t0d), "v"(t1d));
        if (KIND == 23) asm volatile("v_add_f64 %0, %8, %9\nv_add_f64 %1, %8, %9\nv_add_f64 %2, %8, %9\nv_add_f64 %3, %8, %9\nv_add_f64 %4, %8, %9\nv_add_f64 %5, %8, %9\nv_add_f64 %6, %8, %9\nv_add_f64 %7, %8, %9" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(t0d), "v"(t1d));
        if (KIND == 24) asm volatile("v_pk_fma_f32 %0, %8, %9, %8\nv_pk_fma_f32 %1, %8, %9, %8\nv_pk_fma_f32 %2, %8, %9, %8\nv_pk_fma_f32 %3, %8, %9, %8\nv_pk_fma_f32 %4, %8, %9, %8\nv_pk_fma_f32 %5, %8, %9, %8\nv_pk_fma_f32 %6, %8, %9, %8\nv_pk_fma_f32 %7, %8, %9, %8" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(t0d), "v"(t1d));
        if (KIND == 25) asm volatile("v_rcp_f64_e32 %0, %8\nv_rcp_f64_e32 %1, %8\nv_rcp_f64_e32 %2, %8\nv_rcp_f64_e32 %3, %8\nv_rcp_f64_e32 %4, %8\nv_rcp_f64_e32 %5, %8\nv_rcp_f64_e32 %6, %8\nv_rcp_f64_e32 %7, %8" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(t0d), "v"(t1d));
        if (KIND == 200) asm volatile("v_xor_b32_e32 %0, %8, %9\nv_xor_b32_e32 %1, %8, %9\nv_xor_b32_e32 %2, %8, %9\nv_xor_b32_e32 %3, %8, %9\nv_xor_b32_e32 %4, %8, %9\nv_xor_b32_e32 %5, %8, %9\nv_xor_b32_e32 %6, %8, %9\nv_xor_b32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 201) asm volatile("v_or_b32_e32 %0, %8, %9\nv_or_b32_e32 %1, %8, %9\nv_or_b32_e32 %2, %8, %9\nv_or_b32_e32 %3, %8, %9\nv_or_b32_e32 %4, %8, %9\nv_or_b32_e32 %5, %8, %9\nv_or_b32_e32 %6, %8, %9\nv_or_b32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 202) asm volatile("v_not_b32_e32 %0, %8\nv_not_b32_e32 %1, %8\nv_not_b32_e32 %2, %8\nv_not_b32_e32 %3, %8\nv_not_b32_e32 %4, %8\nv_not_b32_e32 %5, %8\nv_not_b32_e32 %6, %8\nv_not_b32_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 203) asm volatile("v_bfi_b32 %0, %8, %9, %8\nv_bfi_b32 %1, %8, %9, %8\nv_bfi_b32 %2, %8, %9, %8\nv_bfi_b32 %3, %8, %9, %8\nv_bfi_b32 %4, %8, %9, %8\nv_bfi_b32 %5, %8, %9, %8\nv_bfi_b32 %6, %8, %9, %8\nv_bfi_b32 %7, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 204) asm volatile("v_sub_u32_e32 %0, %8, %9\nv_sub_u32_e32 %1, %8, %9\nv_sub_u32_e32 %2, %8, %9\nv_sub_u32_e32 %3, %8, %9\nv_sub_u32_e32 %4, %8, %9\nv_sub_u32_e32 %5, %8, %9\nv_sub_u32_e32 %6, %8, %9\nv_sub_u32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 205) asm volatile("v_lshrrev_b32_e32 %0, 8, %8\nv_lshrrev_b32_e32 %1, 8, %8\nv_lshrrev_b32_e32 %2, 8, %8\nv_lshrrev_b32_e32 %3, 8, %8\nv_lshrrev_b32_e32 %4, 8, %8\nv_lshrrev_b32_e32 %5, 8, %8\nv_lshrrev_b32_e32 %6, 8, %8\nv_lshrrev_b32_e32 %7, 8, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 206) asm volatile("v_ashrrev_i32_e32 %0, 8, %8\nv_ashrrev_i32_e32 %1, 8, %8\nv_ashrrev_i32_e32 %2, 8, %8\nv_ashrrev_i32_e32 %3, 8, %8\nv_ashrrev_i32_e32 %4, 8, %8\nv_ashrrev_i32_e32 %5, 8, %8\nv_ashrrev_i32_e32 %6, 8, %8\nv_ashrrev_i32_e32 %7, 8, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 207) asm volatile("v_min_u32_e32 %0, %8, %9\nv_min_u32_e32 %1, %8, %9\nv_min_u32_e32 %2, %8, %9\nv_min_u32_e32 %3, %8, %9\nv_min_u32_e32 %4, %8, %9\nv_min_u32_e32 %5, %8, %9\nv_min_u32_e32 %6, %8, %9\nv_min_u32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 208) asm volatile("v_max_i32_e32 %0, %8, %9\nv_max_i32_e32 %1, %8, %9\nv_max_i32_e32 %2, %8, %9\nv_max_i32_e32 %3, %8, %9\nv_max_i32_e32 %4, %8, %9\nv_max_i32_e32 %5, %8, %9\nv_max_i32_e32 %6, %8, %9\nv_max_i32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 209) asm volatile("v_med3_f32 %0, %8, %9, %8\nv_med3_f32 %1, %8, %9, %8\nv_med3_f32 %2, %8, %9, %8\nv_med3_f32 %3, %8, %9, %8\nv_med3_f32 %4, %8, %9, %8\nv_med3_f32 %5, %8, %9, %8\nv_med3_f32 %6, %8, %9, %8\nv_med3_f32 %7, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 210) asm volatile("v_add3_u32 %0, %8, %9, %8\nv_add3_u32 %1, %8, %9, %8\nv_add3_u32 %2, %8, %9, %8\nv_add3_u32 %3, %8, %9, %8\nv_add3_u32 %4, %8, %9, %8\nv_add3_u32 %5, %8, %9, %8\nv_add3_u32 %6, %8, %9, %8\nv_add3_u32 %7, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 211) asm volatile("v_mad_u32_u24 %0, %8, %9, %8\nv_mad_u32_u24 %1, %8, %9, %8\nv_mad_u32_u24 %2, %8, %9, %8\nv_mad_u32_u24 %3, %8, %9, %8\nv_mad_u32_u24 %4, %8, %9, %8\nv_mad_u32_u24 %5, %8, %9, %8\nv_mad_u32_u24 %6, %8, %9, %8\nv_mad_u32_u24 %7, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 212) asm volatile("v_mul_u32_u24_e32 %0, %8, %9\nv_mul_u32_u24_e32 %1, %8, %9\nv_mul_u32_u24_e32 %2, %8, %9\nv_mul_u32_u24_e32 %3, %8, %9\nv_mul_u32_u24_e32 %4, %8, %9\nv_mul_u32_u24_e32 %5, %8, %9\nv_mul_u32_u24_e32 %6, %8, %9\nv_mul_u32_u24_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 213) asm volatile("v_fmac_f32_e32 %0, %8, %9\nv_fmac_f32_e32 %1, %8, %9\nv_fmac_f32_e32 %2, %8, %9\nv_fmac_f32_e32 %3, %8, %9\nv_fmac_f32_e32 %4, %8, %9\nv_fmac_f32_e32 %5, %8, %9\nv_fmac_f32_e32 %6, %8, %9\nv_fmac_f32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 214) asm volatile("v_sub_f32_e32 %0, %8, %9\nv_sub_f32_e32 %1, %8, %9\nv_sub_f32_e32 %2, %8, %9\nv_sub_f32_e32 %3, %8, %9\nv_sub_f32_e32 %4, %8, %9\nv_sub_f32_e32 %5, %8, %9\nv_sub_f32_e32 %6, %8, %9\nv_sub_f32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 215) asm volatile("v_fract_f32_e32 %0, %8\nv_fract_f32_e32 %1, %8\nv_fract_f32_e32 %2, %8\nv_fract_f32_e32 %3, %8\nv_fract_f32_e32 %4, %8\nv_fract_f32_e32 %5, %8\nv_fract_f32_e32 %6, %8\nv_fract_f32_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 216) asm volatile("v_floor_f32_e32 %0, %8\nv_floor_f32_e32 %1, %8\nv_floor_f32_e32 %2, %8\nv_floor_f32_e32 %3, %8\nv_floor_f32_e32 %4, %8\nv_floor_f32_e32 %5, %8\nv_floor_f32_e32 %6, %8\nv_floor_f32_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 217) asm volatile("v_trunc_f32_e32 %0, %8\nv_trunc_f32_e32 %1, %8\nv_trunc_f32_e32 %2, %8\nv_trunc_f32_e32 %3, %8\nv_trunc_f32_e32 %4, %8\nv_trunc_f32_e32 %5, %8\nv_trunc_f32_e32 %6, %8\nv_trunc_f32_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 218) asm volatile("v_rndne_f32_e32 %0, %8\nv_rndne_f32_e32 %1, %8\nv_rndne_f32_e32 %2, %8\nv_rndne_f32_e32 %3, %8\nv_rndne_f32_e32 %4, %8\nv_rndne_f32_e32 %5, %8\nv_rndne_f32_e32 %6, %8\nv_rndne_f32_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 219) asm volatile("v_cvt_u32_f32_e32 %0, %8\nv_cvt_u32_f32_e32 %1, %8\nv_cvt_u32_f32_e32 %2, %8\nv_cvt_u32_f32_e32 %3, %8\nv_cvt_u32_f32_e32 %4, %8\nv_cvt_u32_f32_e32 %5, %8\nv_cvt_u32_f32_e32 %6, %8\nv_cvt_u32_f32_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 220) asm volatile("v_cvt_f32_i32_e32 %0, %8\nv_cvt_f32_i32_e32 %1, %8\nv_cvt_f32_i32_e32 %2, %8\nv_cvt_f32_i32_e32 %3, %8\nv_cvt_f32_i32_e32 %4, %8\nv_cvt_f32_i32_e32 %5, %8\nv_cvt_f32_i32_e32 %6, %8\nv_cvt_f32_i32_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 221) asm volatile("v_ldexp_f32 %0, %8, 3\nv_ldexp_f32 %1, %8, 3\nv_ldexp_f32 %2, %8, 3\nv_ldexp_f32 %3, %8, 3\nv_ldexp_f32 %4, %8, 3\nv_ldexp_f32 %5, %8, 3\nv_ldexp_f32 %6, %8, 3\nv_ldexp_f32 %7, %8, 3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 222) asm volatile("v_mul_hi_u32 %0, %8, %9\nv_mul_hi_u32 %1, %8, %9\nv_mul_hi_u32 %2, %8, %9\nv_mul_hi_u32 %3, %8, %9\nv_mul_hi_u32 %4, %8, %9\nv_mul_hi_u32 %5, %8, %9\nv_mul_hi_u32 %6, %8, %9\nv_mul_hi_u32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 223) asm volatile("v_cndmask_b32_e32 %0, %8, %9, vcc\nv_cndmask_b32_e32 %1, %8, %9, vcc\nv_cndmask_b32_e32 %2, %8, %9, vcc\nv_cndmask_b32_e32 %3, %8, %9, vcc\nv_cndmask_b32_e32 %4, %8, %9, vcc\nv_cndmask_b32_e32 %5, %8, %9, vcc\nv_cndmask_b32_e32 %6, %8, %9, vcc\nv_cndmask_b32_e32 %7, %8, %9, vcc" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1) : "vcc");
        if (KIND == 224) asm volatile("v_alignbit_b32 %0, %8, %9, 8\nv_alignbit_b32 %1, %8, %9, 8\nv_alignbit_b32 %2, %8, %9, 8\nv_alignbit_b32 %3, %8, %9, 8\nv_alignbit_b32 %4, %8, %9, 8\nv_alignbit_b32 %5, %8, %9, 8\nv_alignbit_b32 %6, %8, %9, 8\nv_alignbit_b32 %7, %8, %9, 8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 225) asm volatile("v_and_or_b32 %0, %8, %9, %8\nv_and_or_b32 %1, %8, %9, %8\nv_and_or_b32 %2, %8, %9, %8\nv_and_or_b32 %3, %8, %9, %8\nv_and_or_b32 %4, %8, %9, %8\nv_and_or_b32 %5, %8, %9, %8\nv_and_or_b32 %6, %8, %9, %8\nv_and_or_b32 %7, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 226) asm volatile("v_lshl_or_b32 %0, %8, 8, %9\nv_lshl_or_b32 %1, %8, 8, %9\nv_lshl_or_b32 %2, %8, 8, %9\nv_lshl_or_b32 %3, %8, 8, %9\nv_lshl_or_b32 %4, %8, 8, %9\nv_lshl_or_b32 %5, %8, 8, %9\nv_lshl_or_b32 %6, %8, 8, %9\nv_lshl_or_b32 %7, %8, 8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 227) asm volatile("v_min_f32_e32 %0, %8, %9\nv_min_f32_e32 %1, %8, %9\nv_min_f32_e32 %2, %8, %9\nv_min_f32_e32 %3, %8, %9\nv_min_f32_e32 %4, %8, %9\nv_min_f32_e32 %5, %8, %9\nv_min_f32_e32 %6, %8, %9\nv_min_f32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 228) asm volatile("v_fma_f32 %0, |%8|, -%9, %8\nv_fma_f32 %1, |%8|, -%9, %8\nv_fma_f32 %2, |%8|, -%9, %8\nv_fma_f32 %3, |%8|, -%9, %8\nv_fma_f32 %4, |%8|, -%9, %8\nv_fma_f32 %5, |%8|, -%9, %8\nv_fma_f32 %6, |%8|, -%9, %8\nv_fma_f32 %7, |%8|, -%9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 229) asm volatile("v_add_f32_e64 %0, |%8|, %9\nv_add_f32_e64 %1, |%8|, %9\nv_add_f32_e64 %2, |%8|, %9\nv_add_f32_e64 %3, |%8|, %9\nv_add_f32_e64 %4, |%8|, %9\nv_add_f32_e64 %5, |%8|, %9\nv_add_f32_e64 %6, |%8|, %9\nv_add_f32_e64 %7, |%8|, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 230) asm volatile("v_mul_f32_e64 %0, %8, %9 mul:2\nv_mul_f32_e64 %1, %8, %9 mul:2\nv_mul_f32_e64 %2, %8, %9 mul:2\nv_mul_f32_e64 %3, %8, %9 mul:2\nv_mul_f32_e64 %4, %8, %9 mul:2\nv_mul_f32_e64 %5, %8, %9 mul:2\nv_mul_f32_e64 %6, %8, %9 mul:2\nv_mul_f32_e64 %7, %8, %9 mul:2" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 231) asm volatile("v_cvt_f32_ubyte0_e32 %0, %8\nv_cvt_f32_ubyte0_e32 %1, %8\nv_cvt_f32_ubyte0_e32 %2, %8\nv_cvt_f32_ubyte0_e32 %3, %8\nv_cvt_f32_ubyte0_e32 %4, %8\nv_cvt_f32_ubyte0_e32 %5, %8\nv_cvt_f32_ubyte0_e32 %6, %8\nv_cvt_f32_ubyte0_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 232) asm volatile("v_cvt_off_f32_i4_e32 %0, %8\nv_cvt_off_f32_i4_e32 %1, %8\nv_cvt_off_f32_i4_e32 %2, %8\nv_cvt_off_f32_i4_e32 %3, %8\nv_cvt_off_f32_i4_e32 %4, %8\nv_cvt_off_f32_i4_e32 %5, %8\nv_cvt_off_f32_i4_e32 %6, %8\nv_cvt_off_f32_i4_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 233) asm volatile("v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        // mixed streams: does a 4-cycle class overlap with fp32 fma issued beside it?  KIND 100: alternating inside every wave;
        // KIND 101: odd waves issue only fma, even waves only cvt (the same number of instructions per wave either way)
        if (KIND == 100) asm volatile("v_fma_f32 %0, %8, %9, %8\nv_cvt_f32_ubyte1_e32 %1, %8\nv_fma_f32 %2, %8, %9, %8\nv_cvt_f32_ubyte1_e32 %3, %8\nv_fma_f32 %4, %8, %9, %8\nv_cvt_f32_ubyte1_e32 %5, %8\nv_fma_f32 %6, %8, %9, %8\nv_cvt_f32_ubyte1_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 101) {
            if ((threadIdx.x >> 6) & 1) asm volatile("v_fma_f32 %0, %8, %9, %8\nv_fma_f32 %1, %8, %9, %8\nv_fma_f32 %2, %8, %9, %8\nv_fma_f32 %3, %8, %9, %8\nv_fma_f32 %4, %8, %9, %8\nv_fma_f32 %5, %8, %9, %8\nv_fma_f32 %6, %8, %9, %8\nv_fma_f32 %7, %8, %9, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
            else asm volatile("v_cvt_f32_ubyte1_e32 %0, %8\nv_cvt_f32_ubyte1_e32 %1, %8\nv_cvt_f32_ubyte1_e32 %2, %8\nv_cvt_f32_ubyte1_e32 %3, %8\nv_cvt_f32_ubyte1_e32 %4, %8\nv_cvt_f32_ubyte1_e32 %5, %8\nv_cvt_f32_ubyte1_e32 %6, %8\nv_cvt_f32_ubyte1_e32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        }
        if (KIND == 102) asm volatile("v_fma_f32 %0, %8, %9, %8\nv_cndmask_b32_e64 %1, %8, %9, s[6:7]\nv_fma_f32 %2, %8, %9, %8\nv_max_f32_e32 %3, %8, %9\nv_fma_f32 %4, %8, %9, %8\nv_cndmask_b32_e64 %5, %8, %9, s[6:7]\nv_fma_f32 %6, %8, %9, %8\nv_max_f32_e32 %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1) : "s6", "s7");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + (float)(e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7);
}

template <int KIND>
int run(const char* name) {
    const int blocks = 256 * 8, iters = 50000; // 8 waves per SIMD
    unsigned long long* d; float* s;
    CHK(hipMalloc(&d, blocks * 2 * sizeof(unsigned long long)));
    CHK(hipMalloc(&s, blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k_class<KIND><<<blocks, 256>>>(d, s, iters); // warm-up (clock ramp)
    CHK(hipEventRecord(e0));
    k_class<KIND><<<blocks, 256>>>(d, s, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 2);
    CHK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> cyc, mhz;
    for (int b = 0; b < blocks; b++) { cyc.push_back((double)h[2 * b]); mhz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
    const double per_simd = (double)iters * 8 * 8;       // instructions the 8 waves of one SIMD issue
    printf("%-20s %8.3f ms  clock %5.0f MHz  => %.2f cycles per wave64 instruction per SIMD by block stamps (median; min %.2f), %.2f by wall time x clock\n",
           name, ms, mhz[blocks / 2], cyc[blocks / 2] / per_simd, cyc[0] / per_simd, ms * 1e-3 * mhz[blocks / 2] * 1e6 / per_simd);
    CHK(hipFree(d)); CHK(hipFree(s));
    return 0;
}
int main() {
    if (run<0>("v_fma_f32")) return 1;
    if (run<1>("v_mul_f32")) return 1;
    if (run<2>("v_add_f32")) return 1;
    if (run<3>("v_mov_b32")) return 1;
    if (run<4>("v_add_u32")) return 1;
    if (run<5>("v_lshlrev_b32")) return 1;
    if (run<6>("v_and_b32")) return 1;
    if (run<7>("v_lshl_add_u32")) return 1;
    if (run<8>("v_cvt_f32_ubyte1")) return 1;
    if (run<9>("v_cvt_f32_u32")) return 1;
    if (run<10>("v_max_f32")) return 1;
    if (run<11>("v_max3_f32")) return 1;
    if (run<12>("v_cndmask_b32_e64")) return 1;
    if (run<13>("v_cmp_lt_f32_e64")) return 1;
    if (run<14>("v_perm_b32")) return 1;
    if (run<15>("v_bfe_u32")) return 1;
    if (run<16>("v_mul_lo_u32")) return 1;
    if (run<17>("v_rcp_f32")) return 1;
    if (run<18>("v_sqrt_f32")) return 1;
    if (run<19>("v_div_scale_f32")) return 1;
    if (run<20>("v_div_fmas_f32")) return 1;
    if (run<21>("v_div_fixup_f32")) return 1;
    if (run<22>("v_fma_f64")) return 1;
    if (run<23>("v_add_f64")) return 1;
    if (run<24>("v_pk_fma_f32")) return 1;
    if (run<25>("v_rcp_f64")) return 1;
    if (run<200>("v_xor_b32")) return 1;
    if (run<201>("v_or_b32")) return 1;
    if (run<202>("v_not_b32")) return 1;
    if (run<203>("v_bfi_b32")) return 1;
    if (run<204>("v_sub_u32")) return 1;
    if (run<205>("v_lshrrev_b32")) return 1;
    if (run<206>("v_ashrrev_i32")) return 1;
    if (run<207>("v_min_u32")) return 1;
    if (run<208>("v_max_i32")) return 1;
    if (run<209>("v_med3_f32")) return 1;
    if (run<210>("v_add3_u32")) return 1;
    if (run<211>("v_mad_u32_u24")) return 1;
    if (run<212>("v_mul_u32_u24")) return 1;
    if (run<213>("v_fmac_f32")) return 1;
    if (run<214>("v_sub_f32")) return 1;
    if (run<215>("v_fract_f32")) return 1;
    if (run<216>("v_floor_f32")) return 1;
    if (run<217>("v_trunc_f32")) return 1;
    if (run<218>("v_rndne_f32")) return 1;
    if (run<219>("v_cvt_u32_f32")) return 1;
    if (run<220>("v_cvt_f32_i32")) return 1;
    if (run<221>("v_ldexp_f32")) return 1;
    if (run<222>("v_mul_hi_u32")) return 1;
    if (run<223>("v_cndmask_b32_e32")) return 1;
    if (run<224>("v_alignbit_b32")) return 1;
    if (run<225>("v_and_or_b32")) return 1;
    if (run<226>("v_lshl_or_b32")) return 1;
    if (run<227>("v_min_f32")) return 1;
    if (run<228>("v_fma_f32 |abs| -neg")) return 1;
    if (run<229>("v_add_f32 sdwa? e64 abs")) return 1;
    if (run<230>("v_mul_f32 omod")) return 1;
    if (run<231>("v_cvt_f32_ubyte0")) return 1;
    if (run<232>("v_cvt_off_f32_i4")) return 1;
    if (run<233>("v_mov_b32 dpp quad")) return 1;
    if (run<100>("4 fma + 4 cvt / wave")) return 1;
    if (run<101>("fma waves | cvt waves")) return 1;
    if (run<102>("4 fma + 4 sel/max")) return 1;
    return 0;
}
