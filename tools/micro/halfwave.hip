// Does gfx950 skip the idle half of a wave64 VALU instruction?  (SIMD-32: a wave issues over two passes of 32 lanes.)
// A VALU-bound loop runs with lanes [0, N) active for several N; if time(N <= 32) ~ 0.5 x time(64) the hardware skips
// an all-idle half and lane COMPACTION into one half is worth something; if the times are equal it is not.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/halfwave.hip -o gpurun_out/halfwave && gpurun_out/halfwave
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float* out, int n_active, int iters, int pattern) {
    const int lane = threadIdx.x & 63;
    bool on = pattern == 0 ? lane < n_active : (pattern == 1 ? (lane & 1) == 0 && (lane >> 1) < n_active : lane >= 64 - n_active);
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const float m = 1.0001f, c = 0.5f;
    if (on) {
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    float* d;
    const int blocks = 256 * 8, iters = 20000;
    hipMalloc(&d, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int ns[] = {64, 48, 33, 32, 16, 1};
    for (int pattern = 0; pattern < 3; pattern++)
        for (int n : ns) {
            if (pattern == 1 && n > 32) continue;
            k<<<blocks, 256>>>(d, n, 100, pattern);
            hipEventRecord(e0);
            k<<<blocks, 256>>>(d, n, iters, pattern);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double inst = (double)blocks * 4 * iters * 8;
            printf("pattern %d (%s) active %2d: %.3f ms  %.2f cycles/wave-instr/SIMD at 2.4 GHz\n", pattern,
                   pattern == 0 ? "low lanes" : (pattern == 1 ? "even lanes" : "high lanes"), n, ms, ms * 1e-3 * 2.4e9 * 1024 / inst);
        }
    return 0;
}
