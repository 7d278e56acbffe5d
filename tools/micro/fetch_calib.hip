// What do FETCH_SIZE / TCC_EA0_RDREQ read for the access patterns of the path-tracing kernels?
// MI355X_MICROARCH.md calibrates FETCH_SIZE for ONE pattern (wide coalesced streams: it reports half the bytes) and says
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The shading kernel's
// reads are gathers (a triangle record, texels, table entries per lane), so this program issues gathers whose fabric traffic
// is known by construction and lets the counters be read against them (tools/pmc_fetch_calib.sh):
//   stream16   every lane 16 contiguous bytes, the whole table once                (the guide's case: FETCH_SIZE = bytes / 2)
//   stream4    every lane 4 contiguous bytes, the whole table once
//   gather4    every lane ONE dword at offset 0 of its own random, distinct 128-byte line (N lines touched once)
//   gather16   the same with a 16-byte load
//   second64   gather4, then -- after the first load has returned -- the dword at offset 64 of the SAME line: if the first miss
//              filled the whole 128-byte line this one hits in L2 and the request count stays N, if fills are 64 bytes it doubles
//   second32   the same at offset 32 (32-byte sectors?)
//   record128  every lane reads its own whole random 128-byte line as 8 x 16 bytes (a TriShade record)
//   pair0 / pair64 / pair32   TWO launches over the same 131072 lines (1/4 of the L2s' capacity), the same lane -> line mapping
//              and so the same XCD: the first reads the dword at offset 0, the second the dword at offset 64 (or 32).  L1 does
//              not survive a launch boundary, L2 does: if the second launch causes (almost) no memory-side requests, the first
//              one's misses filled whole 128-byte lines -- every request is a 128-byte one, whatever the lane asked for
//   evict64 / evict32 / evict0   ONE wave per workgroup, 8 workgroups (one per XCD): 64 lines at offset 0, then 16 x 64 OTHER lines
//              (1024 lines: four times the 32 KiB L1, a thirtieth of the XCD's L2), then the FIRST 64 lines again at offset 64
//              (32, 0): the L1 has lost them, the L2 has not.  Requests per round = 64 x 17 if the first miss filled the
//              whole 128-byte line, 64 x 18 if a fill is 64 (32) bytes; evict0 is the control (must read 17)
// (second64 / second32 use volatile loads, which bypass the caches altogether: they read two requests per line whatever the
// fill size -- kept as the control that the counters do see a second request; pair* are separate launches, and the L2 is
// invalidated between launches: also two.  evict* is the one that decides.)
// Every line is touched at most once per kernel and the table (4 GiB) is far larger than L2 + Infinity Cache, so nothing is
// re-used across lanes; requests per touched line = TCC_EA0_RDREQ / N tells the fill granularity, and with it what one
// request is worth in bytes.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_calib.hip -o tools/micro/fetch_calib.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// line index of access i: a bijection on [0, n_lines) (n_lines a power of two, odd multiplier), far apart for neighbouring i
__device__ __forceinline__ uint64_t line_of(uint64_t i, uint64_t n_lines) { return (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & (n_lines - 1); }

__global__ void stream16(const float4* __restrict__ t, uint64_t n16, float* out) {
    float acc = 0.f;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const float4 v = t[i]; acc += v.x + v.w; }
    if (acc == 12345.f) out[0] = acc;
}
__global__ void stream4(const float* __restrict__ t, uint64_t n4, float* out) {
    float acc = 0.f;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) acc += t[i];
    if (acc == 12345.f) out[0] = acc;
}
__global__ void pair(const char* __restrict__ t, uint64_t n_lines, uint64_t n, int off, float* out) {
    float acc = 0.f;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        acc += *reinterpret_cast<const float*>(t + line_of(i, n_lines) * 128ull + off);
    if (acc == 12345.f) out[0] = acc;
}
__global__ __launch_bounds__(64) void evict(const char* __restrict__ t, uint64_t n_lines, int rounds, int off, float* out) {
    float acc = 0.f;
    for (int r = 0; r < rounds; r++) {
        const uint64_t base = ((uint64_t)(blockIdx.x * rounds + r) * 17ull) * 64ull + threadIdx.x; // 17 x 64 fresh lines per round
        const char* pa = t + line_of(base, n_lines) * 128ull;
        float a = *reinterpret_cast<const float*>(pa);
        for (int k = 1; k <= 16; k++) { // dependent chain: each filler load's address waits for the value before it
            const uint64_t i = base + (uint64_t)k * 64ull + (a == 12345.f ? 1 : 0);
            a += *reinterpret_cast<const float*>(t + line_of(i, n_lines) * 128ull);
        }
        acc += a + *reinterpret_cast<const float*>(pa + off + (a == 12345.f ? 4 : 0));
    }
    if (acc == 12345.f) out[0] = acc;
}
template <int MODE> // 0 gather4, 1 gather16, 2 second64, 3 second32, 4 record128
__global__ void gather(const char* __restrict__ t, uint64_t n_lines, uint64_t n, float* out) {
    float acc = 0.f;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const char* p = t + line_of(i, n_lines) * 128ull;
        if (MODE == 0) acc += *reinterpret_cast<const float*>(p);
        else if (MODE == 1) { const float4 v = *reinterpret_cast<const float4*>(p); acc += v.x + v.w; }
        else if (MODE == 2 || MODE == 3) {
            const float a = *reinterpret_cast<const volatile float*>(p);
            // the second address depends on the first value (always +0 here): it cannot be issued before the first load is back
            const int off = (MODE == 2 ? 64 : 32) + (a == 12345.f ? 4 : 0);
            __builtin_amdgcn_s_sleep(64);
            acc += a + *reinterpret_cast<const volatile float*>(p + off);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) { const float4 v = reinterpret_cast<const float4*>(p)[k]; acc += v.x + v.w; }
        }
    }
    if (acc == 12345.f) out[0] = acc;
}

int main() {
    const uint64_t bytes = 4ull << 30, n_lines = bytes / 128, n = 16ull << 20; // 16 M accesses over 32 M lines
    char* t; float* out;
    CHECK(hipMalloc(&t, bytes)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(t, 0, bytes));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 16, block = 256;
    auto timed = [&](const char* name, double useful, double lines, auto launch) {
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-10s %8.3f ms  useful bytes %.4g  lines touched %.4g  (useful GB/s %.1f)\n", name, ms, useful, lines, useful / ms / 1e6);
    };
    // a 1 GiB stream between the kernels pushes the previous kernel's lines out of L2 and most of the Infinity Cache
    const uint64_t sb = 1ull << 30;
    timed("stream16", (double)sb, sb / 128.0, [&] { stream16<<<grid, block>>>(reinterpret_cast<const float4*>(t), sb / 16, out); });
    timed("stream4", (double)sb, sb / 128.0, [&] { stream4<<<grid, block>>>(reinterpret_cast<const float*>(t + sb), sb / 4, out); });
    timed("gather4", 4.0 * n, (double)n, [&] { gather<0><<<grid, block>>>(t, n_lines, n, out); });
    timed("stream16", (double)sb, sb / 128.0, [&] { stream16<<<grid, block>>>(reinterpret_cast<const float4*>(t + 2 * sb), sb / 16, out); });
    timed("gather16", 16.0 * n, (double)n, [&] { gather<1><<<grid, block>>>(t, n_lines, n, out); });
    timed("stream16", (double)sb, sb / 128.0, [&] { stream16<<<grid, block>>>(reinterpret_cast<const float4*>(t + 3 * sb), sb / 16, out); });
    timed("second64", 8.0 * n, (double)n, [&] { gather<2><<<grid, block>>>(t, n_lines, n, out); });
    timed("stream16", (double)sb, sb / 128.0, [&] { stream16<<<grid, block>>>(reinterpret_cast<const float4*>(t), sb / 16, out); });
    timed("second32", 8.0 * n, (double)n, [&] { gather<3><<<grid, block>>>(t, n_lines, n, out); });
    timed("stream16", (double)sb, sb / 128.0, [&] { stream16<<<grid, block>>>(reinterpret_cast<const float4*>(t + sb), sb / 16, out); });
    timed("record128", 128.0 * n, (double)n, [&] { gather<4><<<grid, block>>>(t, n_lines, n, out); });
    const uint64_t np = 131072;
    for (int off : {64, 32}) {
        timed("stream16", (double)sb, sb / 128.0, [&] { stream16<<<grid, block>>>(reinterpret_cast<const float4*>(t + 2 * sb), sb / 16, out); });
        timed("pair0", 4.0 * np, (double)np, [&] { pair<<<512, 256>>>(t, n_lines, np, 0, out); });
        timed(off == 64 ? "pair64" : "pair32", 4.0 * np, (double)np, [&] { pair<<<512, 256>>>(t, n_lines, np, off, out); });
    }
    for (int off : {64, 32, 0}) {
        timed("stream16", (double)sb, sb / 128.0, [&] { stream16<<<grid, block>>>(reinterpret_cast<const float4*>(t + 3 * sb), sb / 16, out); });
        timed(off == 64 ? "evict64" : (off == 32 ? "evict32" : "evict0"), 4.0 * 8 * 200 * 64 * 18, 8.0 * 200 * 64 * 17, [&] { evict<<<8, 64>>>(t, n_lines, 200, off, out); });
    }
    return 0;
}
