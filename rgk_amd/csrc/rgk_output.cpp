// Output path: normalise the accumulator the way RenderFrame does after every round and write
// the half-float RGBA OpenEXR file.  Host code, no device work: the reference does this once per round on the
// host as well (src/render_driver.cpp:229-247), O(pixels).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rgk.h"

extern "C" int rgk_internal_fail(int code, const char* msg); // rgk_host.cpp: sets rgk_last_error

// float -> half with round-to-nearest-even, denormals, overflow to infinity, NaN kept: the value half(float)
// of OpenEXR's half class produces.
extern "C" uint16_t rgk_float_to_half(float v) {
    uint32_t f;
    std::memcpy(&f, &v, 4);
    const uint32_t sign = (f >> 16) & 0x8000u;
    const uint32_t a = f & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (a > 0x7f800000u ? (0x0200u | ((a >> 13) & 0x3ffu)) : 0u)); // inf / NaN
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);   // rounds to >= 65520: overflow to infinity
    if (a < 0x33000001u) return (uint16_t)sign;                 // < 2^-25 (or exactly 2^-25, a tie to even 0): zero
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }          // half denormal: value = m * 2^(e-23), unit 2^-24
    else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    uint32_t h = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;           // may carry into the exponent: still correct
    return (uint16_t)(sign | (base + h));
}

extern "C" int rgk_output_normalize(const float* accum_rgb, const uint32_t* accum_count, uint32_t xres, uint32_t yres, float output_scale,
                                    float* out_rgb, float* scale_used) {
    if (!accum_rgb || !accum_count || !out_rgb || xres == 0 || yres == 0) return rgk_internal_fail(RGK_ERR_INVALID, "rgk_output_normalize: null or empty argument");
    const size_t P = (size_t)xres * yres;
    float val = output_scale;
    if (val <= 0.0f) { // texture.cpp:381-391
        float m = 0.0f;
        for (size_t p = 0; p < P; p++) {
            if (accum_count[p] == 0) continue;
            const float n = (float)accum_count[p];
            m = std::max(m, accum_rgb[3 * p] / n);
            m = std::max(m, accum_rgb[3 * p + 1] / n);
            m = std::max(m, accum_rgb[3 * p + 2] / n);
        }
        val = 1.0f / m;
    }
    for (size_t p = 0; p < P; p++) { // out.data *= val (:393-398), then GetPixel = data / count (:349-354)
        const uint32_t c = accum_count[p];
        for (int k = 0; k < 3; k++) out_rgb[3 * p + k] = c ? (accum_rgb[3 * p + k] * val) / (float)c : 0.0f;
    }
    if (scale_used) *scale_used = val;
    return RGK_OK;
}

namespace {
struct Out {
    std::vector<uint8_t> b;
    void raw(const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; b.insert(b.end(), q, q + n); }
    void str(const char* s) { raw(s, std::strlen(s) + 1); }
    void i32(int32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    void f32(float v) { raw(&v, 4); }
    void attr(const char* name, const char* type, int32_t size) { str(name); str(type); i32(size); }
};
} // namespace

extern "C" int rgk_output_write_exr(const char* path, uint32_t xres, uint32_t yres, const float* rgb) {
    if (!path || !rgb || xres == 0 || yres == 0) return rgk_internal_fail(RGK_ERR_INVALID, "rgk_output_write_exr: null or empty argument");
    Out o;
    o.i32(20000630); // magic
    o.i32(2);        // version 2, single-part scan-line
    // channels: A, B, G, R (alphabetical), HALF, linear, sampling 1 x 1
    o.attr("channels", "chlist", 4 * (2 + 4 + 4 + 4 + 4) + 1);
    for (const char* c : {"A", "B", "G", "R"}) { o.str(c); o.i32(1); o.b.push_back(0); o.b.push_back(0); o.b.push_back(0); o.b.push_back(0); o.i32(1); o.i32(1); }
    o.b.push_back(0);
    o.attr("compression", "compression", 1); o.b.push_back(0); // NO_COMPRESSION
    o.attr("dataWindow", "box2i", 16); o.i32(0); o.i32(0); o.i32((int32_t)xres - 1); o.i32((int32_t)yres - 1);
    o.attr("displayWindow", "box2i", 16); o.i32(0); o.i32(0); o.i32((int32_t)xres - 1); o.i32((int32_t)yres - 1);
    o.attr("lineOrder", "lineOrder", 1); o.b.push_back(0); // INCREASING_Y
    o.attr("pixelAspectRatio", "float", 4); o.f32(1.0f);
    o.attr("screenWindowCenter", "v2f", 8); o.f32(0.0f); o.f32(0.0f);
    o.attr("screenWindowWidth", "float", 4); o.f32(1.0f);
    o.b.push_back(0); // end of header
    const size_t line_bytes = (size_t)xres * 4 * 2, table_at = o.b.size();
    const uint64_t first = table_at + (uint64_t)8 * yres;
    for (uint32_t y = 0; y < yres; y++) o.u64(first + (uint64_t)y * (8 + line_bytes));
    std::vector<uint16_t> line((size_t)xres * 4);
    const uint16_t one = rgk_float_to_half(1.0f);
    for (uint32_t y = 0; y < yres; y++) {
        const float* src = rgb + (size_t)y * xres * 3;
        for (uint32_t x = 0; x < xres; x++) {
            line[x] = one;                                            // A
            line[(size_t)xres + x] = rgk_float_to_half(src[3 * x + 2]);     // B
            line[(size_t)2 * xres + x] = rgk_float_to_half(src[3 * x + 1]); // G
            line[(size_t)3 * xres + x] = rgk_float_to_half(src[3 * x]);     // R
        }
        o.i32((int32_t)y);
        o.i32((int32_t)line_bytes);
        o.raw(line.data(), line_bytes);
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) return rgk_internal_fail(RGK_ERR_INVALID, "rgk_output_write_exr: cannot open the output file");
    const size_t w = std::fwrite(o.b.data(), 1, o.b.size(), f);
    const int rc = std::fclose(f);
    if (w != o.b.size() || rc != 0) return rgk_internal_fail(RGK_ERR_DEVICE, "rgk_output_write_exr: short write");
    return RGK_OK;
}
