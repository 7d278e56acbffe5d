// Which issue class are the mixed-precision instructions in (gfx950)?  Companion of valu_peak.hip (same harness, same units):
// if v_fma_mix_f32 -- an fp32 fma that takes fp16 operands as they are -- issued like v_fma_f32 (2.4 cycles per wave64
// instruction), the 24 byte->float conversions of a QBVH4 node visit (v_cvt_f32_ubyteN: 4.3 cycles, the class that bounds the
// walkers) could be folded into the 24 plane fmas by storing the quantised planes as fp16.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_mix.hip -o tools/micro/valu_mix.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define REP8(op, tail) op " %0, " tail "\n" op " %1, " tail "\n" op " %2, " tail "\n" op " %3, " tail "\n" op " %4, " tail "\n" op " %5, " tail "\n" op " %6, " tail "\n" op " %7, " tail
#define OUTS "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7)

template <int KIND>
__global__ __launch_bounds__(256, 8) void k_class(unsigned long long* stamps, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    float d0 = lane, d1 = lane + 1.f, d2 = lane + 2.f, d3 = lane + 3.f, d4 = lane + 4.f, d5 = lane + 5.f, d6 = lane + 6.f, d7 = lane + 7.f;
    const float s0 = 1.0001f + lane, s1 = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) asm volatile(REP8("v_fma_f32", "%8, %9, %8") : OUTS : "v"(s0), "v"(s1));
        if (KIND == 1) asm volatile(REP8("v_fma_mix_f32", "%8, %9, %8 op_sel_hi:[1,0,0]") : OUTS : "v"(s0), "v"(s1));
        if (KIND == 2) asm volatile(REP8("v_fma_mix_f32", "%8, %9, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]") : OUTS : "v"(s0), "v"(s1));
        if (KIND == 3) asm volatile(REP8("v_cvt_f32_f16_e32", "%8") : OUTS : "v"(s0), "v"(s1));
        if (KIND == 4) asm volatile(REP8("v_pk_fma_f16", "%8, %9, %8") : OUTS : "v"(s0), "v"(s1));
        if (KIND == 5) asm volatile(REP8("v_dot2_f32_f16", "%8, %9, %8") : OUTS : "v"(s0), "v"(s1));
        if (KIND == 6) asm volatile(REP8("v_cvt_f32_ubyte1_e32", "%8") : OUTS : "v"(s0), "v"(s1));
        if (KIND == 7) asm volatile(REP8("v_fma_mix_f32", "%8, %9, %8 op_sel_hi:[1,1,0]") : OUTS : "v"(s0), "v"(s1));
        if (KIND == 8) asm volatile(REP8("v_fma_f32", "%8, %9, %8 clamp") : OUTS : "v"(s0), "v"(s1));
        // mixed: 4 fma_mix + 4 max (the slow class): do they overlap like fp32 fma + cvt do?
        if (KIND == 9) asm volatile("v_fma_mix_f32 %0, %8, %9, %8 op_sel_hi:[1,0,0]\nv_max_f32_e32 %1, %8, %9\nv_fma_mix_f32 %2, %8, %9, %8 op_sel_hi:[1,0,0]\nv_max_f32_e32 %3, %8, %9\nv_fma_mix_f32 %4, %8, %9, %8 op_sel_hi:[1,0,0]\nv_max_f32_e32 %5, %8, %9\nv_fma_mix_f32 %6, %8, %9, %8 op_sel_hi:[1,0,0]\nv_max_f32_e32 %7, %8, %9" : OUTS : "v"(s0), "v"(s1));
        if (KIND == 10) asm volatile("v_fma_f32 %0, %8, %9, %8\nv_max_f32_e32 %1, %8, %9\nv_fma_f32 %2, %8, %9, %8\nv_max_f32_e32 %3, %8, %9\nv_fma_f32 %4, %8, %9, %8\nv_max_f32_e32 %5, %8, %9\nv_fma_f32 %6, %8, %9, %8\nv_max_f32_e32 %7, %8, %9" : OUTS : "v"(s0), "v"(s1));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
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
    const double per_simd = (double)iters * 8 * 8;
    printf("%-44s %8.3f ms  clock %5.0f MHz  => %.2f cycles per wave64 instruction per SIMD (block stamps, median), %.2f by wall time x clock\n",
           name, ms, mhz[blocks / 2], cyc[blocks / 2] / per_simd, ms * 1e-3 * mhz[blocks / 2] * 1e6 / per_simd);
    CHK(hipFree(d)); CHK(hipFree(s));
    return 0;
}
int main() {
    if (run<0>("v_fma_f32")) return 1;
    if (run<1>("v_fma_mix_f32 (src0 = f16 low half)")) return 1;
    if (run<2>("v_fma_mix_f32 (src0 = f16 high half)")) return 1;
    if (run<7>("v_fma_mix_f32 (src0, src1 = f16)")) return 1;
    if (run<3>("v_cvt_f32_f16")) return 1;
    if (run<4>("v_pk_fma_f16")) return 1;
    if (run<5>("v_dot2_f32_f16")) return 1;
    if (run<6>("v_cvt_f32_ubyte1")) return 1;
    if (run<8>("v_fma_f32 clamp")) return 1;
    if (run<9>("4 x v_fma_mix_f32 + 4 x v_max_f32")) return 1;
    if (run<10>("4 x v_fma_f32 + 4 x v_max_f32")) return 1;
    return 0;
}
