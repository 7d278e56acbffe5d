// Can the byte -> float step of the QBVH4 plane decode leave the slow VALU class?  gfx950 has SDWA on VOP2 fp32 ops: with
// src0_sel:BYTE_n the operand is the zero-extended byte taken as raw float bits, i.e. the DENORMAL b * 2^-149 (fp32 denormals
// are enabled in every kernel here: .amdhsa_float_denorm_mode_32 3).  v_mul_f32_sdwa by 2^127 then yields b * 2^-22 exactly.
// This program (1) checks that value for every byte in every position, (2) times the instruction against v_cvt_f32_ubyteN and
// a plain v_mul_f32 (same method as valu_peak.hip: 8 independent destinations, 8 waves per SIMD, s_memtime around the loop),
// with denormal and with ordinary operands, and (3) does the same for v_mul_f32 with the clamp output modifier.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/sdwa_byte.hip -o tools/micro/sdwa_byte.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_check(const unsigned* w, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned x = w[i];
    const float K = 1.7014118346046923e38f; // 2^127
    float f0, f1, f2, f3;
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(f0) : "v"(x), "v"(K));
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(f1) : "v"(x), "v"(K));
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(f2) : "v"(x), "v"(K));
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(f3) : "v"(x), "v"(K));
    out[4 * i + 0] = f0; out[4 * i + 1] = f1; out[4 * i + 2] = f2; out[4 * i + 3] = f3;
}

#define R8(I) I(0) "\n" I(1) "\n" I(2) "\n" I(3) "\n" I(4) "\n" I(5) "\n" I(6) "\n" I(7)
#define SDWA(n) "v_mul_f32_sdwa %" #n ", %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
#define CVT(n) "v_cvt_f32_ubyte1_e32 %" #n ", %8"
#define MUL(n) "v_mul_f32_e32 %" #n ", %8, %9"
#define MULC(n) "v_mul_f32_e64 %" #n ", %8, %9 clamp"
#define MIX(n) "v_mul_f32_sdwa %" #n ", %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_max_f32_e32 %" #n ", %8, %9"
#define MIXC(n) "v_cvt_f32_ubyte1_e32 %" #n ", %8\nv_max_f32_e32 %" #n ", %8, %9"

template <int KIND>
__global__ __launch_bounds__(256, 8) void k_class(unsigned long long* stamps, float* sink, int iters, unsigned src_bits, float k) {
    const int lane = threadIdx.x & 63;
    float d0 = lane, d1 = lane + 1.f, d2 = lane + 2.f, d3 = lane + 3.f, d4 = lane + 4.f, d5 = lane + 5.f, d6 = lane + 6.f, d7 = lane + 7.f;
    const float s0 = __uint_as_float(src_bits + (unsigned)lane * 0x0101u), s1 = k;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) asm volatile(R8(SDWA) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 1) asm volatile(R8(CVT) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 2) asm volatile(R8(MUL) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 3) asm volatile(R8(MULC) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 4) asm volatile(R8(MIX) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
        if (KIND == 5) asm volatile(R8(MIXC) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 12345.678f) sink[0] = d0;
}

template <int KIND>
static int run(const char* name, unsigned src_bits, float k, int per_iter) {
    const int blocks = 2048, iters = 20000;
    unsigned long long* stamps; float* sink;
    CHK(hipMalloc(&stamps, blocks * 4 * sizeof(unsigned long long))); CHK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k_class<KIND><<<blocks, 256>>>(stamps, sink, 100, src_bits, k);
    CHK(hipEventRecord(e0));
    k_class<KIND><<<blocks, 256>>>(stamps, sink, iters, src_bits, k);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    // 2048 blocks x 4 waves = 8192 waves over 1024 SIMDs = 8 per SIMD, one round; instructions per SIMD = 8 waves x iters x per_iter
    const double inst_per_simd = 8.0 * iters * per_iter;
    const double clk_ghz = 2.4; // nominal; valu_peak.hip measures the clock -- here only the RATIOS between the rows matter
    printf("%-44s %8.3f ms  %.2f cycles per wave instruction at %.1f GHz\n", name, ms, ms * 1e6 * clk_ghz / inst_per_simd, clk_ghz);
    (void)hipFree(stamps); (void)hipFree(sink);
    return 0;
}

int main() {
    // (1) values
    std::vector<unsigned> w(256);
    for (unsigned b = 0; b < 256; b++) w[b] = b | ((255u - b) << 8) | (((b * 7u) & 255u) << 16) | (((b * 13u + 5u) & 255u) << 24);
    unsigned* dw; float* dout;
    CHK(hipMalloc(&dw, 256 * 4)); CHK(hipMalloc(&dout, 256 * 16));
    CHK(hipMemcpy(dw, w.data(), 256 * 4, hipMemcpyHostToDevice));
    k_check<<<1, 256>>>(dw, dout, 256);
    std::vector<float> out(1024);
    CHK(hipMemcpy(out.data(), dout, 256 * 16, hipMemcpyDeviceToHost));
    int bad = 0;
    for (unsigned i = 0; i < 256; i++)
        for (int c = 0; c < 4; c++) {
            const float want = (float)((w[i] >> (8 * c)) & 255u) * 2.384185791015625e-07f; // 2^-22
            if (std::memcmp(&want, &out[4 * i + c], 4) != 0) { if (bad < 5) printf("MISMATCH w=%08x byte %d: got %g want %g\n", w[i], c, out[4 * i + c], want); bad++; }
        }
    printf("v_mul_f32_sdwa BYTE_n x 2^127 == byte * 2^-22 for all 256 values x 4 positions: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    // (2) rates
    run<0>("v_mul_f32_sdwa BYTE_1 (denormal operand)", 0x00003700u, 1.7014118346046923e38f, 8);
    run<0>("v_mul_f32_sdwa BYTE_1 (zero byte)", 0x00000000u, 1.7014118346046923e38f, 8);
    run<1>("v_cvt_f32_ubyte1", 0x00003700u, 1.f, 8);
    run<2>("v_mul_f32 (ordinary operands)", 0x3f800000u, 1.5f, 8);
    run<2>("v_mul_f32 (denormal x 2^127)", 0x00000037u, 1.7014118346046923e38f, 8);
    run<3>("v_mul_f32 clamp", 0x3f800000u, 1.5f, 8);
    run<4>("sdwa mul + v_max_f32 interleaved (per instr)", 0x00003700u, 1.7014118346046923e38f, 16);
    run<5>("cvt_ubyte + v_max_f32 interleaved (per instr)", 0x00003700u, 1.f, 16);
    return bad != 0;
}
