// VALU issue calibration for the traversal kernels' roofline (gfx950): how many shader cycles one SIMD spends per wave64
// instruction of each class when 4 waves per SIMD issue nothing else.  The stream is inline asm (the compiler can neither pack,
// fold nor drop it); the clock comes from s_memtime (shader cycles) against s_memrealtime (100 MHz), stamped around the
// loop; only stamp deltas leave the kernel, into a buffer nothing else reads.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_peak.hip -o tools/micro/valu_peak.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define OP8(S) asm volatile(S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S : "+v"(a) , "+v"(b) : "v"(m), "v"(c))
template <int KIND>
__global__ __launch_bounds__(256, 8) void k(unsigned long long* stamps, float* sink, int iters, int n_active) {
    const int lane = threadIdx.x & 63;
    float a = lane, b = lane + 1.f;
    const float m = 1.0001f, c = 0.5f;
    double da = lane, db = lane + 1.0;
    const double dm = 1.0001, dc = 0.5;
    unsigned long long sm = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (lane < n_active) {
        for (int i = 0; i < iters; i++) {
            // 8 instructions per statement, alternating two independent destination registers
            if (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %2, %3\nv_fma_f32 %1, %1, %2, %3\nv_fma_f32 %0, %0, %2, %3\nv_fma_f32 %1, %1, %2, %3\nv_fma_f32 %0, %0, %2, %3\nv_fma_f32 %1, %1, %2, %3\nv_fma_f32 %0, %0, %2, %3\nv_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(m), "v"(c)); }
            if (KIND == 1) { asm volatile("v_cvt_f32_ubyte0_e32 %0, %2\nv_cvt_f32_ubyte1_e32 %1, %2\nv_cvt_f32_ubyte2_e32 %0, %2\nv_cvt_f32_ubyte3_e32 %1, %2\nv_cvt_f32_ubyte0_e32 %0, %2\nv_cvt_f32_ubyte1_e32 %1, %2\nv_cvt_f32_ubyte2_e32 %0, %2\nv_cvt_f32_ubyte3_e32 %1, %2" : "+v"(a), "+v"(b) : "v"(m), "v"(c)); }
            if (KIND == 2) { asm volatile("v_max3_f32 %0, %0, %2, %3\nv_min3_f32 %1, %1, %2, %3\nv_max3_f32 %0, %0, %2, %3\nv_min3_f32 %1, %1, %2, %3\nv_max3_f32 %0, %0, %2, %3\nv_min3_f32 %1, %1, %2, %3\nv_max3_f32 %0, %0, %2, %3\nv_min3_f32 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(m), "v"(c)); }
            if (KIND == 3) { asm volatile("v_cndmask_b32_e32 %0, %2, %3, vcc\nv_cndmask_b32_e32 %1, %2, %3, vcc\nv_cndmask_b32_e32 %0, %2, %3, vcc\nv_cndmask_b32_e32 %1, %2, %3, vcc\nv_cndmask_b32_e32 %0, %2, %3, vcc\nv_cndmask_b32_e32 %1, %2, %3, vcc\nv_cndmask_b32_e32 %0, %2, %3, vcc\nv_cndmask_b32_e32 %1, %2, %3, vcc" : "+v"(a), "+v"(b) : "v"(m), "v"(c) : "vcc"); }
            if (KIND == 4) { asm volatile("v_fma_f64 %0, %0, %2, %3\nv_fma_f64 %1, %1, %2, %3\nv_fma_f64 %0, %0, %2, %3\nv_fma_f64 %1, %1, %2, %3\nv_fma_f64 %0, %0, %2, %3\nv_fma_f64 %1, %1, %2, %3\nv_fma_f64 %0, %0, %2, %3\nv_fma_f64 %1, %1, %2, %3" : "+v"(da), "+v"(db) : "v"(dm), "v"(dc)); }
            if (KIND == 5) { asm volatile("v_rcp_f32_e32 %0, %0\nv_rcp_f32_e32 %1, %1\nv_rcp_f32_e32 %0, %0\nv_rcp_f32_e32 %1, %1\nv_rcp_f32_e32 %0, %0\nv_rcp_f32_e32 %1, %1\nv_rcp_f32_e32 %0, %0\nv_rcp_f32_e32 %1, %1" : "+v"(a), "+v"(b)); }
            if (KIND == 6) { asm volatile("v_mul_lo_u32 %0, %0, %2\nv_mul_lo_u32 %1, %1, %2\nv_mul_lo_u32 %0, %0, %2\nv_mul_lo_u32 %1, %1, %2\nv_mul_lo_u32 %0, %0, %2\nv_mul_lo_u32 %1, %1, %2\nv_mul_lo_u32 %0, %0, %2\nv_mul_lo_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(m)); }
            if (KIND == 7) { asm volatile("v_pk_fma_f32 %0, %0, %2, %2\nv_pk_fma_f32 %1, %1, %2, %2\nv_pk_fma_f32 %0, %0, %2, %2\nv_pk_fma_f32 %1, %1, %2, %2\nv_pk_fma_f32 %0, %0, %2, %2\nv_pk_fma_f32 %1, %1, %2, %2\nv_pk_fma_f32 %0, %0, %2, %2\nv_pk_fma_f32 %1, %1, %2, %2" : "+v"(da), "+v"(db) : "v"(dm)); }
            if (KIND == 8) { asm volatile("v_rcp_f64_e32 %0, %0\nv_rcp_f64_e32 %1, %1\nv_rcp_f64_e32 %0, %0\nv_rcp_f64_e32 %1, %1\nv_rcp_f64_e32 %0, %0\nv_rcp_f64_e32 %1, %1\nv_rcp_f64_e32 %0, %0\nv_rcp_f64_e32 %1, %1" : "+v"(da), "+v"(db)); }
            if (KIND == 10) { asm volatile("v_perm_b32 %0, %2, %3, %4\nv_perm_b32 %1, %2, %3, %4\nv_perm_b32 %0, %2, %3, %4\nv_perm_b32 %1, %2, %3, %4\nv_perm_b32 %0, %2, %3, %4\nv_perm_b32 %1, %2, %3, %4\nv_perm_b32 %0, %2, %3, %4\nv_perm_b32 %1, %2, %3, %4" : "+v"(a), "+v"(b) : "v"(m), "v"(c), "s"(0x07060100)); }
            if (KIND == 11) { asm volatile("v_and_or_b32 %0, %2, %3, %4\nv_and_or_b32 %1, %2, %3, %4\nv_and_or_b32 %0, %2, %3, %4\nv_and_or_b32 %1, %2, %3, %4\nv_and_or_b32 %0, %2, %3, %4\nv_and_or_b32 %1, %2, %3, %4\nv_and_or_b32 %0, %2, %3, %4\nv_and_or_b32 %1, %2, %3, %4" : "+v"(a), "+v"(b) : "v"(m), "v"(c), "s"(0x3f800000)); }
            if (KIND == 12) { asm volatile("v_bfe_u32 %0, %2, 8, 8\nv_bfe_u32 %1, %2, 16, 8\nv_bfe_u32 %0, %2, 8, 8\nv_bfe_u32 %1, %2, 16, 8\nv_bfe_u32 %0, %2, 8, 8\nv_bfe_u32 %1, %2, 16, 8\nv_bfe_u32 %0, %2, 8, 8\nv_bfe_u32 %1, %2, 16, 8" : "+v"(a), "+v"(b) : "v"(m)); }
            if (KIND == 13) { asm volatile("v_cvt_f32_u32_e32 %0, %2\nv_cvt_f32_u32_e32 %1, %2\nv_cvt_f32_u32_e32 %0, %2\nv_cvt_f32_u32_e32 %1, %2\nv_cvt_f32_u32_e32 %0, %2\nv_cvt_f32_u32_e32 %1, %2\nv_cvt_f32_u32_e32 %0, %2\nv_cvt_f32_u32_e32 %1, %2" : "+v"(a), "+v"(b) : "v"(m)); }
            if (KIND == 14) { asm volatile("v_cvt_f32_u32_sdwa %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\nv_cvt_f32_u32_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\nv_cvt_f32_u32_sdwa %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\nv_cvt_f32_u32_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\nv_cvt_f32_u32_sdwa %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\nv_cvt_f32_u32_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\nv_cvt_f32_u32_sdwa %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\nv_cvt_f32_u32_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "+v"(a), "+v"(b) : "v"(m)); }
            if (KIND == 15) { asm volatile("v_fma_mix_f32 %0, %2, %3, %0 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %1, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\nv_fma_mix_f32 %0, %2, %3, %0 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %1, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\nv_fma_mix_f32 %0, %2, %3, %0 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %1, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\nv_fma_mix_f32 %0, %2, %3, %0 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %1, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a), "+v"(b) : "v"(m), "v"(c)); }
            if (KIND == 16) { asm volatile("v_max_f32_e32 %0, %0, %2\nv_min_f32_e32 %1, %1, %2\nv_max_f32_e32 %0, %0, %2\nv_min_f32_e32 %1, %1, %2\nv_max_f32_e32 %0, %0, %2\nv_min_f32_e32 %1, %1, %2\nv_max_f32_e32 %0, %0, %2\nv_min_f32_e32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(m)); }
            if (KIND == 17) { asm volatile("v_cndmask_b32_e64 %0, %2, %3, %4\nv_cndmask_b32_e64 %1, %2, %3, %4\nv_cndmask_b32_e64 %0, %2, %3, %4\nv_cndmask_b32_e64 %1, %2, %3, %4\nv_cndmask_b32_e64 %0, %2, %3, %4\nv_cndmask_b32_e64 %1, %2, %3, %4\nv_cndmask_b32_e64 %0, %2, %3, %4\nv_cndmask_b32_e64 %1, %2, %3, %4" : "+v"(a), "+v"(b) : "v"(m), "v"(c), "s"(0x5555555555555555ull)); }
            if (KIND == 18) { asm volatile("v_cmp_lt_f32_e64 %2, %0, %1\nv_cmp_lt_f32_e64 %2, %1, %0\nv_cmp_lt_f32_e64 %2, %0, %1\nv_cmp_lt_f32_e64 %2, %1, %0\nv_cmp_lt_f32_e64 %2, %0, %1\nv_cmp_lt_f32_e64 %2, %1, %0\nv_cmp_lt_f32_e64 %2, %0, %1\nv_cmp_lt_f32_e64 %2, %1, %0" : "+v"(a), "+v"(b), "=s"(sm)); }
            if (KIND == 19) { asm volatile("v_mul_f32_e32 %0, %0, %2\nv_add_f32_e32 %1, %1, %2\nv_mul_f32_e32 %0, %0, %2\nv_add_f32_e32 %1, %1, %2\nv_mul_f32_e32 %0, %0, %2\nv_add_f32_e32 %1, %1, %2\nv_mul_f32_e32 %0, %0, %2\nv_add_f32_e32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(m)); }
            if (KIND == 20) { asm volatile("v_mov_b32_e32 %0, %2\nv_mov_b32_e32 %1, %2\nv_mov_b32_e32 %0, %2\nv_mov_b32_e32 %1, %2\nv_mov_b32_e32 %0, %2\nv_mov_b32_e32 %1, %2\nv_mov_b32_e32 %0, %2\nv_mov_b32_e32 %1, %2" : "+v"(a), "+v"(b) : "v"(m)); }
            if (KIND == 21) { asm volatile("v_add_u32_e32 %0, %0, %2\nv_add_u32_e32 %1, %1, %2\nv_add_u32_e32 %0, %0, %2\nv_add_u32_e32 %1, %1, %2\nv_add_u32_e32 %0, %0, %2\nv_add_u32_e32 %1, %1, %2\nv_add_u32_e32 %0, %0, %2\nv_add_u32_e32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(m)); }
            if (KIND == 22) { asm volatile("v_lshlrev_b32_e32 %0, 3, %2\nv_lshlrev_b32_e32 %1, 5, %2\nv_lshlrev_b32_e32 %0, 3, %2\nv_lshlrev_b32_e32 %1, 5, %2\nv_lshlrev_b32_e32 %0, 3, %2\nv_lshlrev_b32_e32 %1, 5, %2\nv_lshlrev_b32_e32 %0, 3, %2\nv_lshlrev_b32_e32 %1, 5, %2" : "+v"(a), "+v"(b) : "v"(m)); }
            if (KIND == 23) { asm volatile("v_add_f64 %0, %0, %2\nv_mul_f64 %1, %1, %2\nv_add_f64 %0, %0, %2\nv_mul_f64 %1, %1, %2\nv_add_f64 %0, %0, %2\nv_mul_f64 %1, %1, %2\nv_add_f64 %0, %0, %2\nv_mul_f64 %1, %1, %2" : "+v"(da), "+v"(db) : "v"(dm)); }
            if (KIND == 24) { asm volatile("v_div_scale_f32 %0, vcc, %0, %2, %0\nv_div_fixup_f32 %1, %1, %2, %3\nv_div_scale_f32 %0, vcc, %0, %2, %0\nv_div_fixup_f32 %1, %1, %2, %3\nv_div_scale_f32 %0, vcc, %0, %2, %0\nv_div_fixup_f32 %1, %1, %2, %3\nv_div_scale_f32 %0, vcc, %0, %2, %0\nv_div_fixup_f32 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(m), "v"(c) : "vcc"); }
            if (KIND == 9) { asm volatile("v_sqrt_f32_e32 %0, %0\nv_sqrt_f32_e32 %1, %1\nv_sqrt_f32_e32 %0, %0\nv_sqrt_f32_e32 %1, %1\nv_sqrt_f32_e32 %0, %0\nv_sqrt_f32_e32 %1, %1\nv_sqrt_f32_e32 %0, %0\nv_sqrt_f32_e32 %1, %1" : "+v"(a), "+v"(b)); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + (float)(da + db) + (float)(sm & 1);
}

template <int KIND>
int run(const char* name, int n_active) {
    const int blocks = 256 * 4, iters = 100000; // 4 waves per SIMD, all resident at once
    unsigned long long* d; float* s;
    CHK(hipMalloc(&d, blocks * 2 * sizeof(unsigned long long)));
    CHK(hipMalloc(&s, blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<KIND><<<blocks, 256>>>(d, s, iters, n_active); // warm-up (clock ramp)
    CHK(hipEventRecord(e0));
    k<KIND><<<blocks, 256>>>(d, s, iters, n_active);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 2);
    CHK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> cyc, mhz;
    for (int b = 0; b < blocks; b++) { cyc.push_back((double)h[2 * b]); mhz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
    const double per_wave = (double)iters * 8;           // instructions per wave
    // 4 waves share a SIMD for the whole kernel (1024 blocks of 4 waves, 4 blocks per CU): cycles per instruction per SIMD
    const double cpi = cyc[blocks / 2] / (per_wave * 4.0);
    printf("%-22s lanes %2d: %8.3f ms  clock %6.0f MHz  block cycles %.3e  => %.2f cycles per wave64 instruction per SIMD (wall-clock check: %.2f)\n",
           name, n_active, ms, mhz[blocks / 2], cyc[blocks / 2], cpi, ms * 1e-3 * mhz[blocks / 2] * 1e6 / (per_wave * 4.0));
    CHK(hipFree(d)); CHK(hipFree(s));
    return 0;
}
int main() {
    if (run<0>("v_fma_f32", 64)) return 1;
    run<0>("v_fma_f32", 32); run<0>("v_fma_f32", 16);
    run<1>("v_cvt_f32_ubyteN", 64); run<2>("v_max3/min3_f32", 64); run<3>("v_cndmask_b32", 64);
    run<10>("v_perm_b32", 64); run<11>("v_and_or_b32", 64); run<12>("v_bfe_u32", 64); run<13>("v_cvt_f32_u32", 64); run<14>("v_cvt_f32_u32_sdwa", 64);
    run<15>("v_fma_mix_f32", 64); run<16>("v_max/min_f32", 64); run<17>("v_cndmask_b32_e64", 64); run<18>("v_cmp_lt_f32_e64", 64); run<19>("v_mul/add_f32", 64);
    run<20>("v_mov_b32", 64); run<21>("v_add_u32", 64); run<22>("v_lshlrev_b32", 64); run<23>("v_add/mul_f64", 64); run<24>("v_div_scale/fixup_f32", 64);
    run<4>("v_fma_f64", 64); run<5>("v_rcp_f32", 64); run<9>("v_sqrt_f32", 64); run<6>("v_mul_lo_u32", 64); run<7>("v_pk_fma_f32", 64); run<8>("v_rcp_f64", 64);
    return 0;
}
