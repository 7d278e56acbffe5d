// Device-side math for the gfx950 path-tracing kernels.
//
// Each function states the reference function whose RESULT it reproduces.  The code is
// written for the GPU (registers, float4 records, no virtual dispatch), not transcribed:
// but every float operation that feeds a result is kept in the reference's evaluation
// order and the library is built with -ffp-contract=off; sin / cos / acos / asin / atan2 are the pinned
// definitions of include/rgk_libm.h on both sides -- so GPU and CPU agree to the last bit, see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.h"
#include "../../include/rgk_libm.h" // sin / cos / acos / asin / atan2: pinned definitions shared with the oracle (bit-identical on CPU and GPU)

#define RGK_PI_F 3.14159265358979323846264338327950288f

// ------------------------------------------------------------------ scene-table loads
// Pointers that reach a kernel inside a by-value struct (DevScene, PassParams) are generic to the compiler: it
// emits flat_load with a 64-bit address pair per access.  The scene tables are global memory and far below 4 GiB
// each, so they are read as (uniform base in SGPRs) + (ONE 32-bit byte offset in a VGPR), address space 1:
// `global_load ... v_off, s[base]`.  The shade kernels are register-bound; this is where their spills came from.
#define RGK_GLOBAL __attribute__((address_space(1)))
typedef float rgk_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ const RGK_GLOBAL char* gld_addr(const void* base, uint32_t byte_off) {
    const RGK_GLOBAL char* p = (const RGK_GLOBAL char*)base + byte_off;
    return p;
}
__device__ __forceinline__ uint32_t gld_u32(const void* base, uint32_t byte_off) { return *reinterpret_cast<const RGK_GLOBAL uint32_t*>(gld_addr(base, byte_off)); }
__device__ __forceinline__ uint32_t gld_u16(const void* base, uint32_t byte_off) { return *reinterpret_cast<const RGK_GLOBAL uint16_t*>(gld_addr(base, byte_off)); }
__device__ __forceinline__ float gld_f32(const void* base, uint32_t byte_off) { return *reinterpret_cast<const RGK_GLOBAL float*>(gld_addr(base, byte_off)); }
__device__ __forceinline__ float4 gld_f4(const void* base, uint32_t byte_off) {
    const rgk_f4v v = *reinterpret_cast<const RGK_GLOBAL rgk_f4v*>(gld_addr(base, byte_off));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <typename T>
__device__ __forceinline__ T gld_rec(const void* base, uint32_t byte_off) { // a whole record (a multiple of 4 bytes)
    static_assert(sizeof(T) % 4 == 0, "record size");
    T out;
    uint32_t* o = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (uint32_t k = 0; k < sizeof(T) / 4; k++) o[k] = gld_u32(base, byte_off + 4u * k);
    return out;
}
// per-slot path state can exceed 4 GiB: 64-bit offset, still address space 1
__device__ __forceinline__ float4 gld_f4_wide(const void* base, size_t index) {
    const rgk_f4v v = reinterpret_cast<const RGK_GLOBAL rgk_f4v*>((const RGK_GLOBAL char*)base)[index];
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gst_f4_wide(void* base, size_t index, float4 v) {
    const rgk_f4v w = {v.x, v.y, v.z, v.w};
    reinterpret_cast<RGK_GLOBAL rgk_f4v*>((RGK_GLOBAL char*)base)[index] = w;
}

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot3(f3 a, f3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
__device__ __forceinline__ f3 cross3(f3 x, f3 y) {
    return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
__device__ __forceinline__ float len3(f3 v) { return sqrtf(dot3(v, v)); }
__device__ __forceinline__ f3 norm3(f3 v) { return v * (1.0f / sqrtf(dot3(v, v))); }
__device__ __forceinline__ float comp(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
__device__ __forceinline__ float max3c(f3 v) { return fmaxf(fmaxf(v.x, v.y), v.z); }
__device__ __forceinline__ bool is_zero3(f3 v) { return v.x == 0.f && v.y == 0.f && v.z == 0.f; }
__device__ __forceinline__ float glm_angle(f3 a, f3 b) { return rgk_acosf(fminf(fmaxf(dot3(a, b), -1.0f), 1.0f)); }

struct quatf {
    float w, x, y, z;
};
// q * v, glm semantics
__device__ __forceinline__ f3 qrot(quatf q, f3 v) {
    f3 qv = mk3(q.x, q.y, q.z);
    f3 uv = cross3(qv, v);
    f3 uuv = cross3(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.0f;
}
__device__ __forceinline__ quatf qinverse(quatf q) {
    float d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    quatf r; r.w = q.w / d; r.x = -q.x / d; r.y = -q.y / d; r.z = -q.z / d;
    return r;
}
__device__ __forceinline__ quatf angle_axis(float a, f3 axis) {
    const rgk_sincos sc_ = rgk_sincosf_v(a * 0.5f);
    const float s = sc_.s, c = sc_.c;
    quatf r; r.w = c; r.x = axis.x * s; r.y = axis.y * s; r.z = axis.z * s;
    return r;
}
// RotationBetweenVectors, reference src/glm.cpp:3-33
__device__ __forceinline__ quatf rotation_between(f3 start, f3 dest) {
    start = norm3(start);
    dest = norm3(dest);
    float cosTheta = dot3(start, dest);
    if (cosTheta < -1 + 0.001f) {
        f3 axis = cross3(mk3(0.0f, 1.0f, 0.0f), start);
        if ((double)len3(axis) < 0.01) axis = cross3(mk3(1.0f, 0.0f, 0.0f), start);
        axis = norm3(axis);
        return angle_axis(RGK_PI_F, axis);
    }
    f3 axis = cross3(start, dest);
    float s = sqrtf((1 + cosTheta) * 2);
    float invs = 1 / s;
    quatf r; r.w = s * 0.5f; r.x = axis.x * invs; r.y = axis.y * invs; r.z = axis.z * invs;
    return r;
}
// RotationFromY, reference src/glm.cpp:35-59
__device__ __forceinline__ quatf rotation_from_y(f3 dest) {
    dest = norm3(dest);
    float cosTheta = dest.y;
    if (cosTheta < -1 + 0.00001f) return angle_axis(RGK_PI_F, mk3(1.0f, 0.0f, 0.0f));
    f3 axis = cross3(mk3(0.0f, 1.0f, 0.0f), dest);
    float s = sqrtf((1 + cosTheta) * 2);
    float invs = 1 / s;
    quatf r; r.w = s * 0.5f; r.x = axis.x * invs; r.y = axis.y * invs; r.z = axis.z * invs;
    return r;
}

// ------------------------------------------------------------------ sampler (K0)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

// HS::Halton_sampler::sample(dim, index) after init_faure(): reference
// external/halton_sampler.h:627-889 (dispatch), :1418-.. (halton2, halton3, ...).
// Digit walk with a per-base Faure permutation and a multiply-high division instead of
// the header's 256 generated functions and grouped-digit tables; bit-identical results.
__device__ __forceinline__ float halton_raw(const DevScene& sc, uint32_t hdim, uint32_t index) {
    if (hdim == 0) {
        uint32_t u = 0x3f800000u | (__brev(index) >> 9);
        return __uint_as_float(u) - 1.f;
    }
    const DevHaltonDim hd = gld_rec<DevHaltonDim>(sc.hdims, hdim * (uint32_t)sizeof(DevHaltonDim));
    uint32_t acc = 0, j = 0;
    while (index != 0 && j < hd.digits) {
        uint32_t t = __umulhi(hd.magic, index);
        uint32_t q = (t + ((index - t) >> 1)) >> hd.shift;
        uint32_t r = index - q * hd.base;
        acc = acc * hd.base + gld_u16(sc.hperm, (hd.perm_off + r) << 1);
        index = q;
        j++;
    }
    for (; j < hd.digits; j++) acc *= hd.base; // sigma(0) == 0 for Faure permutations
    return (float)acc * hd.scale;
}
__device__ __forceinline__ uint32_t hdim_key(uint32_t hdim) { return mix32(hdim * 0x9e3779b9u + 0x85ebca6bu); }
__device__ __forceinline__ float halton_cp(const DevScene& sc, uint32_t seed, uint32_t index, uint32_t hdim) {
    if (hdim >= 192) return u01(mix32(mix32(seed ^ hdim_key(hdim)) + index * 0xc2b2ae35u));
    float u = halton_raw(sc, hdim, index) + u01(mix32(seed ^ hdim_key(hdim)));
    if (u >= 1.0f) u -= 1.0f;
    return u;
}
// Sampler::Get2D / Get1D for logical dimension k of sample `index` (reference
// src/sampler.cpp:26-36: separate 1-D and 2-D counters; 64 table dimensions each)
__device__ __forceinline__ float2 sample2d(const DevScene& sc, uint32_t seed, uint32_t index, uint32_t k) {
    uint32_t d = k < 64 ? 3 * k : 192 + 3 * (k - 64);
    float x = halton_cp(sc, seed, index, d);
    float y = halton_cp(sc, seed, index, d + 1);
    return make_float2(x, y);
}
__device__ __forceinline__ float sample1d(const DevScene& sc, uint32_t seed, uint32_t index, uint32_t k) {
    return halton_cp(sc, seed, index, k < 64 ? 3 * k + 2 : 192 + 3 * (k - 64) + 2);
}
// The raw Halton value depends on (dimension, sample index) only, not on the pixel: a round
// tabulates halton_raw once into htab[hdim * S + index] (192 x multisample floats, L2-resident)
// and every path reads it back -- same bits, no digit walk on the hot path.
struct SamplerTab {
    const float* htab;
    uint32_t S;
};
__device__ __forceinline__ float halton_cp_t(const SamplerTab& tb, uint32_t seed, uint32_t index, uint32_t hdim) {
    if (hdim >= 192) return u01(mix32(mix32(seed ^ hdim_key(hdim)) + index * 0xc2b2ae35u));
    float u = tb.htab[hdim * tb.S + index] + u01(mix32(seed ^ hdim_key(hdim)));
    if (u >= 1.0f) u -= 1.0f;
    return u;
}
__device__ __forceinline__ float2 sample2d_t(const SamplerTab& tb, uint32_t seed, uint32_t index, uint32_t k) {
    uint32_t d = k < 64 ? 3 * k : 192 + 3 * (k - 64);
    float x = halton_cp_t(tb, seed, index, d);
    float y = halton_cp_t(tb, seed, index, d + 1);
    return make_float2(x, y);
}
__device__ __forceinline__ float sample1d_t(const SamplerTab& tb, uint32_t seed, uint32_t index, uint32_t k) {
    return halton_cp_t(tb, seed, index, k < 64 ? 3 * k + 2 : 192 + 3 * (k - 64) + 2);
}

// ------------------------------------------------------------------ random_utils.hpp:12-73
__device__ __forceinline__ float2 disc_uniform(float2 s) {
    float r = sqrtf(s.x);
    float a = (float)((double)(s.y * 2.0f) * 3.14159265358979323846);
    const rgk_sincos sc_ = rgk_sincosf_v(a);
    const float sn = sc_.s, cs = sc_.c;
    return make_float2(r * sn, r * cs);
}
__device__ __forceinline__ f3 hemisphere_cosine_z(float2 s) {
    float2 p = disc_uniform(s);
    float z = sqrtf(fmaxf(0.00001f, 1 - p.x * p.x - p.y * p.y));
    return mk3(p.x, p.y, z);
}
__device__ __forceinline__ f3 hemisphere_cosine_y(float2 s) {
    float2 p = disc_uniform(s);
    float y = sqrtf(fmaxf(0.00001f, 1 - p.x * p.x - p.y * p.y));
    return mk3(p.x, y, p.y);
}
__device__ __forceinline__ f3 sphere_uniform(float2 s) {
    float z = s.x * 2.0f - 1.0f;
    float a = (float)((double)s.y * 6.283185);
    float r = sqrtf(1 - z * z);
    const rgk_sincos sc_ = rgk_sincosf_v(a);
    const float sn = sc_.s, cs = sc_.c;
    return mk3(r * cs, r * sn, z);
}
__device__ __forceinline__ bool decide_and_rescale(float& sample, float probability) {
    if (probability == 0.0f) return false;
    if (probability == 1.0f) return true;
    if (sample < probability) { sample /= probability; return true; }
    sample = (sample - probability) / (1.0f - probability);
    return false;
}

// ------------------------------------------------------------------ textures (a14)
__device__ __forceinline__ float glm_repeat(float x) { return x - floorf(x); }
__device__ __forceinline__ uint32_t tex_kind(const TexRef t) { return t.kind; }
// Texel index = tex_row(y) + tex_col(x).  Float texels (16 bytes each) lie row by row.  Byte texels (4 bytes each, RGK_TEX_RGB8) lie
// in tiles of 8 x 4 texels = ONE 128-byte line -- the unit every miss moves (FETCH_SIZE calibration): the 2 x 2
// footprint of GetPixelInterpolated and the three taps of the bump slopes then sit in one line two times in three instead of
// always straddling two rows.  Tiles row by row, the image padded up to whole tiles (rgk_host.cpp lays them out the same way).
__device__ __forceinline__ uint32_t tex_row(const TexRef t, int y) {
    const uint32_t w = t.a & 0xffffu;
    if (RGK_TEX_TILED && t.kind == RGK_TEX_RGB8) return ((uint32_t)y >> 2) * (((w + 7u) >> 3) << 5) + (((uint32_t)y & 3u) << 3);
    return (uint32_t)y * w;
}
__device__ __forceinline__ uint32_t tex_col(const TexRef t, int x) {
    if (RGK_TEX_TILED && t.kind == RGK_TEX_RGB8) return (((uint32_t)x >> 3) << 5) + ((uint32_t)x & 7u);
    return (uint32_t)x;
}
// The first byte->float texel tables of the scene are copied into dynamic LDS by every kernel that shades
// (lut_lds_fill at kernel entry; launch with RGK_LDS_LUT_FLOATS * 4 bytes of dynamic LDS): the three dependent
// table loads per texel were 7 % of the shade kernel.
#define RGK_LDS_LUT_FLOATS 512u
#define RGK_LDS_MATERIALS 64u // the first material records follow the tables (the hot loop's third dependent load)
#define RGK_LDS_SHADE_BYTES ((RGK_LDS_LUT_FLOATS + RGK_LDS_MATERIALS * (uint32_t)(sizeof(DevMaterial) / 4)) * 4u)
extern __shared__ float rgk_lut_lds[];
__device__ __forceinline__ void lut_lds_fill(const DevScene& sc) {
    for (uint32_t k = threadIdx.x; k < RGK_LDS_LUT_FLOATS; k += blockDim.x) rgk_lut_lds[k] = k < sc.n_lut_floats ? gld_f32(sc.luts, k << 2) : 0.f;
    const uint32_t nm = (sc.n_materials < RGK_LDS_MATERIALS ? sc.n_materials : RGK_LDS_MATERIALS) * (uint32_t)(sizeof(DevMaterial) / 4);
    for (uint32_t k = threadIdx.x; k < nm; k += blockDim.x) rgk_lut_lds[RGK_LDS_LUT_FLOATS + k] = gld_f32(sc.materials, k << 2);
    __syncthreads();
}
__device__ __forceinline__ DevMaterial mat_load(const DevScene& sc, uint32_t id) {
    if (id < RGK_LDS_MATERIALS) {
        DevMaterial m;
        const float* src = rgk_lut_lds + RGK_LDS_LUT_FLOATS + id * (uint32_t)(sizeof(DevMaterial) / 4);
        float* dst = reinterpret_cast<float*>(&m);
#pragma unroll
        for (uint32_t k = 0; k < sizeof(DevMaterial) / 4; k++) dst[k] = src[k];
        return m;
    }
    return gld_rec<DevMaterial>(sc.materials, id * (uint32_t)sizeof(DevMaterial));
}
__device__ __forceinline__ f3 texel_at(const DevScene& sc, const TexRef t, uint32_t idx) { // idx = tex_row + tex_col
    if (tex_kind(t) == RGK_TEX_RGB8) { // the bytes the 8-bit loaders keep + the table that makes them the reference's floats
        const uint32_t w = gld_u32(sc.texels8, (t.b + idx) << 2);
        if (t.c + 256u <= RGK_LDS_LUT_FLOATS) { // the table sits in the workgroup's LDS (lut_lds_fill): 3 ds_read instead of 3 global loads
            const float* lut = rgk_lut_lds + t.c;
            return mk3(lut[w & 0xffu], lut[(w >> 8) & 0xffu], lut[(w >> 16) & 0xffu]);
        }
        const uint32_t lut = t.c << 2;
        return mk3(gld_f32(sc.luts, lut + ((w & 0xffu) << 2)), gld_f32(sc.luts, lut + ((w >> 6) & 0x3fcu)), gld_f32(sc.luts, lut + ((w >> 14) & 0x3fcu)));
    }
    const float4 v = gld_f4(sc.texels, (t.b + idx) << 4);
    return mk3(v.x, v.y, v.z);
}
// N texels of one texture with their loads IN FLIGHT TOGETHER: one decision about the texel format and one about where the table
// lives, then N loads, then N decodes.  Texel by texel (texel_at in a row) every fetch sits behind its own format branches, so the
// compiler waits for each word before it asks for the next: the four taps of GetPixelInterpolated and the three of the bump
// slopes were seven round trips to memory one after the other, per texture.
template <int N>
__device__ __forceinline__ void texels_at(const DevScene& sc, const TexRef t, const uint32_t (&idx)[N], f3 (&out)[N]) {
    if (tex_kind(t) == RGK_TEX_RGB8) {
        uint32_t w[N];
#pragma unroll
        for (int k = 0; k < N; k++) w[k] = gld_u32(sc.texels8, (t.b + idx[k]) << 2);
        if (t.c + 256u <= RGK_LDS_LUT_FLOATS) {
            const float* lut = rgk_lut_lds + t.c;
#pragma unroll
            for (int k = 0; k < N; k++) out[k] = mk3(lut[w[k] & 0xffu], lut[(w[k] >> 8) & 0xffu], lut[(w[k] >> 16) & 0xffu]);
        } else {
            const uint32_t lut = t.c << 2;
#pragma unroll
            for (int k = 0; k < N; k++)
                out[k] = mk3(gld_f32(sc.luts, lut + ((w[k] & 0xffu) << 2)), gld_f32(sc.luts, lut + ((w[k] >> 6) & 0x3fcu)), gld_f32(sc.luts, lut + ((w[k] >> 14) & 0x3fcu)));
        }
    } else {
        float4 v[N];
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = gld_f4(sc.texels, (t.b + idx[k]) << 4);
#pragma unroll
        for (int k = 0; k < N; k++) out[k] = mk3(v[k].x, v[k].y, v[k].z);
    }
}
// ReadableTexture::GetPixelInterpolated, reference src/texture.cpp:35-77 (FileTexture) and
// src/texture.hpp:64-80 (Solid / Empty): the four taps and the two weights ...
struct TexFoot {
    uint32_t idx[4]; // r0 + k0, r0 + k1, r1 + k0, r1 + k1
    float fx, fy;
};
__device__ __forceinline__ TexFoot tex_foot(const TexRef t, float2 uv) {
    const int xsize = (int)(t.a & 0xffffu), ysize = (int)(t.a >> 16);
    float x = glm_repeat(uv.x) * xsize - 0.5f;
    float y = glm_repeat(uv.y) * ysize - 0.5f;
    float ix0f = truncf(x), iy0f = truncf(y);
    float fx = x - ix0f, fy = y - iy0f; // == std::modf fractional part (exact)
    int ix0 = (int)ix0f, iy0 = (int)iy0f;
    int ix1 = (ix0 != xsize - 1) ? ix0 + 1 : ix0;
    int iy1 = (iy0 != ysize - 1) ? iy0 + 1 : iy0;
    if (ix0 == -1) ix0 = 0;
    if (iy0 == -1) iy0 = 0;
    const uint32_t r0 = tex_row(t, iy0), r1 = tex_row(t, iy1), k0 = tex_col(t, ix0), k1 = tex_col(t, ix1);
    TexFoot f;
    f.idx[0] = r0 + k0; f.idx[1] = r0 + k1; f.idx[2] = r1 + k0; f.idx[3] = r1 + k1;
    f.fx = fx; f.fy = fy;
    return f;
}
// ... and the blend
__device__ __forceinline__ f3 tex_blend(const TexFoot& f, const f3 (&c)[4]) {
    const float fy = 1.0f - f.fy;
    const float fx = 1.0f - f.fx;
    f3 c0s = fx * c[0] + (1.0f - fx) * c[1];
    f3 c1s = fx * c[2] + (1.0f - fx) * c[3];
    return fy * c0s + (1.0f - fy) * c1s;
}
__device__ __forceinline__ f3 tex_get(const DevScene& sc, const TexRef t, float2 uv) {
    if (tex_kind(t) == RGK_TEXREF_NONE) return mk3(0.f, 0.f, 0.f);
    if (tex_kind(t) == RGK_TEX_SOLID) return mk3(__uint_as_float(t.a), __uint_as_float(t.b), __uint_as_float(t.c));
    const TexFoot f = tex_foot(t, uv);
    f3 c[4];
    texels_at<4>(sc, t, f.idx, c);
    return tex_blend(f, c);
}
// (A material's colour and diffuse maps fetched as ONE batch of eight taps: measured, nothing alone and slower together with other
// hoisted loads -- the kernel is at its 128 registers.)
// (Asking for the first texel of the colour and diffuse maps EARLY -- as soon as uv is known, before the bump map's round trip,
// the word kept alive until after the real fetch -- was measured again in round 3 on top of the tiled layout: first vertices 31.2 ->
// 32.2 ms per round, later vertices 24.9 -> 26.5.  The shading kernels do not wait for texels; not kept.)
// GetSlopeRight / GetSlopeBottom, reference src/texture.cpp:79-102
__device__ __forceinline__ void tex_slopes(const DevScene& sc, const TexRef t, float2 uv, float& right, float& bottom) {
    right = 0.f; bottom = 0.f;
    if (tex_kind(t) != RGK_TEX_RGB32F && tex_kind(t) != RGK_TEX_RGB8) return;
    const int xsize = (int)(t.a & 0xffffu), ysize = (int)(t.a >> 16);
    int x = (int)(glm_repeat(uv.x) * xsize - 0.5f);
    int y = (int)(glm_repeat(uv.y) * ysize - 0.5f);
    int x2 = (x != xsize - 1) ? x + 1 : x;
    int y2 = (y != ysize - 1) ? y + 1 : y;
    if (x == -1) x = 0;
    if (y == -1) y = 0;
    const uint32_t r0 = tex_row(t, y), k0 = tex_col(t, x);
    const uint32_t idx[3] = {r0 + k0, r0 + tex_col(t, x2), tex_row(t, y2) + k0};
    f3 c[3];
    texels_at<3>(sc, t, idx, c);
    const f3 here = c[0], tr = c[1], tb = c[2];
    float a = (here.x + here.y + here.z) / 3;
    right = a - (tr.x + tr.y + tr.z) / 3;
    bottom = a - (tb.x + tb.y + tb.z) / 3;
}

// ------------------------------------------------------------------ LTC (a12)
// The fitted matrix has 5 live entries: M = [m0 0 m6; 0 m4 0; m2 0 m8] (rows), see
// reference src/LTC/ltc.hpp:4-12 (column-major mat33 -> glm::mat3).
struct LtcM {
    float m0, m2, m4, m6, m8, amp;
};
// LTC::get_bilinear, reference src/LTC/ltc.cpp:20-57.  Table entry = {m0,m2,m4,m6}{amp,-,-,-}.
__device__ __forceinline__ LtcM ltc_bilinear(const void* ltc, uint32_t tab, float theta, float alpha) { // tab: byte offset of the table in `ltc`
    float t = fmaxf(0.0f, fminf(1.0f, theta / (0.5f * 3.14159f)));
    float a = fmaxf(0.0f, fminf(1.0f, sqrtf(alpha)));
    if (t >= 1.0f) t = 0.999f;
    if (a >= 1.0f) a = 0.999f;
    const int s = 63;
    int t1 = (int)floorf(t * s), t2 = t1 + 1;
    int a1 = (int)floorf(a * s), a2 = a1 + 1;
    const uint32_t o11 = tab + ((uint32_t)(a1 + t1 * 64) << 5), o12 = tab + ((uint32_t)(a2 + t1 * 64) << 5);
    const uint32_t o21 = tab + ((uint32_t)(a1 + t2 * 64) << 5), o22 = tab + ((uint32_t)(a2 + t2 * 64) << 5);
    const float4 m11 = gld_f4(ltc, o11), m12 = gld_f4(ltc, o12), m21 = gld_f4(ltc, o21), m22 = gld_f4(ltc, o22);
    const float p11 = gld_f32(ltc, o11 + 16u), p12 = gld_f32(ltc, o12 + 16u), p21 = gld_f32(ltc, o21 + 16u), p22 = gld_f32(ltc, o22 + 16u);
    float dt1 = t * s - t1, dt2 = t2 - t * s, da1 = a * s - a1, da2 = a2 - a * s;
#define RGK_BIL(e11, e12, e21, e22) ((e11) * dt2 * da2 + (e12) * dt2 * da1 + (e21) * dt1 * da2 + (e22) * dt1 * da1)
    LtcM r;
    r.m0 = RGK_BIL(m11.x, m12.x, m21.x, m22.x); r.m2 = RGK_BIL(m11.y, m12.y, m21.y, m22.y);
    r.m4 = RGK_BIL(m11.z, m12.z, m21.z, m22.z); r.m6 = RGK_BIL(m11.w, m12.w, m21.w, m22.w);
    r.amp = RGK_BIL(p11, p12, p21, p22);
#undef RGK_BIL
    r.m8 = dt2 * da2 + dt2 * da1 + dt1 * da2 + dt1 * da1; // the constant-1 entry, interpolated like the rest
    return r;
}
__device__ __forceinline__ float ltc_theta(f3 B) { return glm_angle(B, mk3(0.f, 0.f, 1.f)); }

// LTC::GetPDF(ltc, N=+Z, Vr=A, Vi=B, alpha), reference src/LTC/ltc.cpp:59-87, given the table
// entry M interpolated at theta = angle(B, N).  The closed forms are glm::inverse / determinant /
// mat*vec with the structural zeros of M and of the N = +Z frame folded in: every surviving
// product and sum keeps its place and order in glm's cofactor expressions (x*0 = +-0 and
// x + +-0 = x for finite x), so the value is bit-identical to the general formulas of the oracle.
__device__ __forceinline__ float ltc_pdf_M(const LtcM& M, f3 A, f3 B) {
    // rotate = mat3((B.x,B.y,0), (-B.y,B.x,0), (0,0,1)); A3 = inverse(rotate) * A
    float det = B.x * B.x + B.y * B.y;
    float ood = 1.0f / det;
    float bxo = B.x * ood, byo = B.y * ood;
    f3 A3 = mk3(bxo * A.x + byo * A.y, (-byo) * A.x + bxo * A.y, (det * ood) * A.z);
    // invM * A3 with m[0]=(m0,0,m2) m[1]=(0,m4,0) m[2]=(m6,0,m8)
    float a00 = M.m0, a02 = M.m2, a11 = M.m4, a20 = M.m6, a22 = M.m8;
    float detM = a00 * (a11 * a22) - a20 * (a11 * a02);
    float oodM = 1.0f / detM;
    float j00 = (a11 * a22) * oodM, j20 = (-(a20 * a11)) * oodM;
    float j11 = (a00 * a22 - a20 * a02) * oodM;
    float j02 = (-(a11 * a02)) * oodM, j22 = (a00 * a11) * oodM;
    f3 p = norm3(mk3(j00 * A3.x + j20 * A3.z, j11 * A3.y, j02 * A3.x + j22 * A3.z));
    f3 L = mk3(a00 * p.x + a20 * p.z, a11 * p.y, a02 * p.x + a22 * p.z); // M * p
    float l = len3(L);
    float Jacobian = detM / (l * l * l);
    float D = 1.0f / 3.14159f * fmaxf(0.0f, p.z);
    return M.amp * D / Jacobian;
}
// LTC::GetRandom(ltc, N=+Z, Vi, roughness, rand_hscos), reference src/LTC/ltc.cpp:113-143, given
// M interpolated at max(theta, pi/4)
__device__ __forceinline__ f3 ltc_random_M(const LtcM& M, f3 Vi, f3 rnd) {
    f3 s = mk3(M.m0 * rnd.x + M.m6 * rnd.z, M.m4 * rnd.y, M.m2 * rnd.x + M.m8 * rnd.z);
    if (s.z < 0.0001f) s.z = 0.0001f;
    // rotate * s, rotate = mat3((Vi.x,Vi.y,0), (-Vi.y,Vi.x,0), (0,0,1))
    f3 r = mk3(Vi.x * s.x + (-Vi.y) * s.y, Vi.y * s.x + Vi.x * s.y, s.z);
    return norm3(r);
}
__device__ __forceinline__ float ltc_pdf(const void* ltc, uint32_t tab, f3 A, f3 B, float alpha) {
    return ltc_pdf_M(ltc_bilinear(ltc, tab, ltc_theta(B), alpha), A, B);
}
__device__ __forceinline__ f3 ltc_random(const void* ltc, uint32_t tab, f3 Vi, float roughness, f3 rnd) {
    return ltc_random_M(ltc_bilinear(ltc, tab, fmaxf(ltc_theta(Vi), RGK_PI_F / 4.0f), roughness), Vi, rnd);
}

// ------------------------------------------------------------------ BxDFs (a11)
// FresnellDielectric, reference src/bxdf/bxdf.cpp:332-355
__device__ __forceinline__ void fresnel_dielectric(float eta, float cosTheta, float& R, float& cosT) {
    if (cosTheta < 0.0f) { eta = 1.0f / eta; cosTheta = -cosTheta; }
    float sinThetaTSq = eta * eta * (1.0f - cosTheta * cosTheta);
    if (sinThetaTSq > 1.0f) { R = 1.0f; cosT = 0.0f; return; }
    float cosThetaTrans = sqrtf(fmaxf(1.0f - sinThetaTSq, 0.0f));
    float Rs = (eta * cosTheta - cosThetaTrans) / (eta * cosTheta + cosThetaTrans);
    float Rp = (eta * cosThetaTrans - cosTheta) / (eta * cosThetaTrans + cosTheta);
    R = 0.5f * (Rs * Rs + Rp * Rp);
    cosT = cosThetaTrans;
}

// BxDF::value of a non-mix material, reference src/bxdf/bxdf.cpp:192-423, bxdf.hpp:107-159
__device__ __forceinline__ f3 bxdf_value_leaf(const DevScene& sc, const DevMaterial& m, f3 Vi, f3 Vr, float2 uv) {
    const f3 zero = mk3(0.f, 0.f, 0.f);
    switch (m.kind) {
    case RGK_BXDF_DIFFUSE:
        if (Vi.z <= 0 || Vr.z <= 0) return zero;
        return tex_get(sc, m.t_diffuse, uv) / RGK_PI_F;
    case RGK_BXDF_MIRROR: {
        f3 refl = mk3(-Vi.x, -Vi.y, Vi.z);
        return (fabsf(dot3(refl, Vr) - 1) < 0.0001f) ? tex_get(sc, m.t_color, uv) : zero;
    }
    case RGK_BXDF_DIELECTRIC: {
        float eta = (Vi.z < 0) ? m.ior : (float)(1.0 / (double)m.ior);
        float R, cosT;
        fresnel_dielectric(eta, Vi.z, R, cosT);
        f3 c = tex_get(sc, m.t_color, uv);
        if (Vi.z * Vr.z > 0) {
            f3 refl = mk3(-Vi.x, -Vi.y, Vi.z);
            return (fabsf(dot3(Vr, refl) - 1) < 0.001f) ? mk3(R, R, R) * c : zero;
        }
        f3 refr = mk3(-Vi.x * eta, -Vi.y * eta, (Vi.z > 0) ? -cosT : cosT);
        float T = 1.0f - R;
        return (fabsf(dot3(Vr, refr) - 1) < 0.001f) ? mk3(T, T, T) * c : zero;
    }
    case RGK_BXDF_TRANSPARENT: {
        f3 inv = mk3(-Vi.x, -Vi.y, -Vi.z);
        return (fabsf(dot3(inv, Vr) - 1) < 0.0001f) ? mk3(1.f, 1.f, 1.f) : zero;
    }
    case RGK_BXDF_LTC_BECKMANN:
    case RGK_BXDF_LTC_GGX: {
        if (Vi.z <= 0 || Vr.z <= 0) return zero;
        const uint32_t tab = (m.kind == RGK_BXDF_LTC_GGX) ? 0u : RGK_LTC_TABLE_BYTES; // GGX table, then Beckmann, in one buffer
        const f3 spec = tex_get(sc, m.t_color, uv);
        if (is_zero3(spec)) return zero; // Q6: a black lobe is not evaluated (the reference multiplies 0 by the pdf)
        return spec * ltc_pdf(sc.ltc, tab, Vi, Vr, m.roughness);
    }
    case RGK_BXDF_LTC_BECKMANN_DIFFUSE:
    case RGK_BXDF_LTC_GGX_DIFFUSE: {
        if (Vi.z <= 0 || Vr.z <= 0) return zero;
        const uint32_t tab = (m.kind == RGK_BXDF_LTC_GGX_DIFFUSE) ? 0u : RGK_LTC_TABLE_BYTES; // GGX table, then Beckmann, in one buffer
        f3 diff = tex_get(sc, m.t_diffuse, uv);
        f3 spec = tex_get(sc, m.t_color, uv);
        if (is_zero3(spec)) return diff / RGK_PI_F; // Q6
        return spec * ltc_pdf(sc.ltc, tab, Vi, Vr, m.roughness) + diff / RGK_PI_F;
    }
    default: return zero;
    }
}
// BxDFMix::value recurses once per level (reference bxdf.cpp:235-239); the device walks an
// explicit stack so nested mixes up to 4 deep are evaluated without recursion.
__device__ __forceinline__ f3 bxdf_value(const DevScene& sc, int mat, f3 Vi, f3 Vr, float2 uv) {
    const DevMaterial m = gld_rec<DevMaterial>(sc.materials, mat * (uint32_t)sizeof(DevMaterial));
    if (m.kind != RGK_BXDF_MIX) return bxdf_value_leaf(sc, m, Vi, Vr, uv);
    // s1*amt1 + s2*(1-amt1) with one nested level on either side
    f3 s[2];
    const int ch[2] = {m.mix_m1, m.mix_m2};
    for (int k = 0; k < 2; k++) {
        const DevMaterial c = gld_rec<DevMaterial>(sc.materials, ch[k] * (uint32_t)sizeof(DevMaterial));
        if (c.kind != RGK_BXDF_MIX) s[k] = bxdf_value_leaf(sc, c, Vi, Vr, uv);
        else {
            const DevMaterial c1 = gld_rec<DevMaterial>(sc.materials, c.mix_m1 * (uint32_t)sizeof(DevMaterial));
            const DevMaterial c2 = gld_rec<DevMaterial>(sc.materials, c.mix_m2 * (uint32_t)sizeof(DevMaterial));
            f3 v1 = (c1.kind == RGK_BXDF_MIX) ? mk3(0.f, 0.f, 0.f) : bxdf_value_leaf(sc, c1, Vi, Vr, uv);
            f3 v2 = (c2.kind == RGK_BXDF_MIX) ? mk3(0.f, 0.f, 0.f) : bxdf_value_leaf(sc, c2, Vi, Vr, uv);
            s[k] = v1 * c.amount + v2 * (1.0f - c.amount);
        }
    }
    return s[0] * m.amount + s[1] * (1.0f - m.amount);
}

// BxDF::sample, reference src/bxdf/bxdf.cpp:197-204,241-249,272-276,378-408,419-423, bxdf.hpp:115-159
__device__ __forceinline__ void bxdf_sample(const DevScene& sc, int mat, f3 Vi, float2 uv, float2 u, f3& dir, f3& weight, bool& may_leak) {
    DevMaterial m = gld_rec<DevMaterial>(sc.materials, mat * (uint32_t)sizeof(DevMaterial));
    for (int lvl = 0; lvl < 8 && m.kind == RGK_BXDF_MIX; lvl++) // BxDFMix::sample descends one side
        m = gld_rec<DevMaterial>(sc.materials, (uint32_t)(decide_and_rescale(u.x, m.amount) ? m.mix_m1 : m.mix_m2) * (uint32_t)sizeof(DevMaterial));
    may_leak = false;
    const f3 zero = mk3(0.f, 0.f, 0.f);
    switch (m.kind) {
    case RGK_BXDF_DIFFUSE:
        if (Vi.z <= 0) { dir = mk3(0.f, 1.f, 0.f); weight = zero; return; }
        dir = hemisphere_cosine_z(u);
        weight = tex_get(sc, m.t_diffuse, uv);
        return;
    case RGK_BXDF_MIRROR:
        dir = mk3(-Vi.x, -Vi.y, Vi.z);
        weight = tex_get(sc, m.t_color, uv);
        return;
    case RGK_BXDF_DIELECTRIC: {
        float eta = (Vi.z < 0) ? m.ior : (float)(1.0 / (double)m.ior);
        float R, cosT;
        fresnel_dielectric(eta, fabsf(Vi.z), R, cosT);
        weight = tex_get(sc, m.t_color, uv);
        if (decide_and_rescale(u.x, R)) { dir = mk3(-Vi.x, -Vi.y, Vi.z); return; }
        cosT = fabsf(cosT);
        dir = mk3(-Vi.x * eta, -Vi.y * eta, (Vi.z > 0) ? -cosT : cosT);
        may_leak = true;
        return;
    }
    case RGK_BXDF_TRANSPARENT:
        dir = mk3(-Vi.x, -Vi.y, -Vi.z);
        weight = mk3(1.f, 1.f, 1.f);
        may_leak = true;
        return;
    case RGK_BXDF_LTC_BECKMANN:
    case RGK_BXDF_LTC_GGX: {
        const uint32_t tab = (m.kind == RGK_BXDF_LTC_GGX) ? 0u : RGK_LTC_TABLE_BYTES; // GGX table, then Beckmann, in one buffer
        f3 v = ltc_random(sc.ltc, tab, Vi, m.roughness, hemisphere_cosine_z(u));
        dir = v;
        weight = (v.z <= 0) ? zero : tex_get(sc, m.t_color, uv);
        return;
    }
    case RGK_BXDF_LTC_BECKMANN_DIFFUSE:
    case RGK_BXDF_LTC_GGX_DIFFUSE: {
        const uint32_t tab = (m.kind == RGK_BXDF_LTC_GGX_DIFFUSE) ? 0u : RGK_LTC_TABLE_BYTES; // GGX table, then Beckmann, in one buffer
        f3 diff = tex_get(sc, m.t_diffuse, uv);
        f3 spec = tex_get(sc, m.t_color, uv);
        float dp = diff.x + diff.y + diff.z, sp = spec.x + spec.y + spec.z;
        float prob = dp / (dp + sp + 0.0001f);
        if (decide_and_rescale(u.x, prob)) {
            if (Vi.z <= 0) { dir = mk3(0.f, 1.f, 0.f); weight = zero; return; }
            dir = hemisphere_cosine_z(u);
            weight = diff;
            return;
        }
        f3 v = ltc_random(sc.ltc, tab, Vi, m.roughness, hemisphere_cosine_z(u));
        dir = v;
        weight = (v.z <= 0) ? zero : spec;
        return;
    }
    default:
        dir = mk3(0.f, 1.f, 0.f);
        weight = zero;
    }
}

// Every function that takes the scene is force-inlined -- the generic BxDF route (mirror, dielectric, transparent,
// mix) included -- so a kernel's by-value DevScene never needs an address: its pointers stay in SGPRs instead of being
// copied to scratch at kernel entry and re-read from there (which is what ONE out-of-line `const DevScene&` call
// used to cost).  (DevScene::self, a device-resident copy of the record, is kept for an out-of-line variant.)
#define RGK_SLOW_ATTR __forceinline__ // measured: as an out-of-line call it costs the shade kernel 25 % (190 vs 153 ms per two rounds)
__device__ RGK_SLOW_ATTR f3 bxdf_value_slow(const DevScene& sc, int mat, f3 Vi, f3 Vr, float2 uv) { return bxdf_value(sc, mat, Vi, Vr, uv); }
__device__ RGK_SLOW_ATTR void bxdf_sample_slow(const DevScene& sc, int mat, f3 Vi, float2 uv, float2 u, f3& dir, f3& weight, bool& may_leak) {
    bxdf_sample(sc, mat, Vi, uv, u, dir, weight, may_leak);
}

// Per-vertex material evaluation with everything `sample` and `value` share fetched once:
// texture colours at uv, and the LTC table entries (value interpolates at theta, sample at
// max(theta, pi/4): the same entry whenever theta >= pi/4).  Same values as bxdf_sample /
// bxdf_value; mirrors, dielectrics, transparents and mixes take the generic route.
struct MatPrep {
    bool fast;
    bool lobe; // the LTC lobe can contribute: its colour is not exactly black (Q6: the reference multiplies a black colour by the pdf;
               // here the lobe is then neither looked up nor evaluated, and a sample of it is the zero weight it would have been)
    f3 diffc, colorc;
    LtcM Mv, Ms;
};
__device__ __forceinline__ bool mat_is_fast(uint32_t k) { return k == RGK_BXDF_DIFFUSE || k >= RGK_BXDF_LTC_BECKMANN; } // MatPrep::fast
__device__ __forceinline__ void mat_prepare(const DevScene& sc, const DevMaterial& m, float2 uv, f3 VrL, bool need_sample, MatPrep& e) {
    // (The two colours are formed as VALUES and stored once, at the end.  Stored from where they are fetched -- e.colorc = tex_get(..);
    // e.diffc = tex_get(..) -- the compiler merges the two fetches' tails into one block that stores through a SELECTED pointer
    // before this function is inlined, and the kernel then keeps both colours in scratch memory: 24 bytes and 17 scratch
    // instructions per vertex in every shading kernel.)
    bool fast = false, lobe = false;
    f3 diffc = mk3(0.f, 0.f, 0.f), colorc = diffc;
    const uint32_t k = m.kind;
    if (k == RGK_BXDF_DIFFUSE) {
        fast = true;
        diffc = tex_get(sc, m.t_diffuse, uv);
    } else if (k >= RGK_BXDF_LTC_BECKMANN) {
        fast = true;
        colorc = tex_get(sc, m.t_color, uv);
        if (k >= RGK_BXDF_LTC_BECKMANN_DIFFUSE) diffc = tex_get(sc, m.t_diffuse, uv);
        lobe = !is_zero3(colorc);
        if (lobe) {
            const uint32_t tab = (k == RGK_BXDF_LTC_GGX || k == RGK_BXDF_LTC_GGX_DIFFUSE) ? 0u : RGK_LTC_TABLE_BYTES; // GGX table, then Beckmann, in one buffer
            const float theta = ltc_theta(VrL);
            e.Mv = ltc_bilinear(sc.ltc, tab, theta, m.roughness);
            e.Ms = (theta >= RGK_PI_F / 4.0f || !need_sample) ? e.Mv : ltc_bilinear(sc.ltc, tab, RGK_PI_F / 4.0f, m.roughness);
        }
    }
    e.fast = fast; e.lobe = lobe;
    e.diffc = diffc; e.colorc = colorc;
}
// GENERIC = false: the caller guarantees a fast-route material (the generic route is compiled out)
template <bool GENERIC = true>
__device__ __forceinline__ void mat_sample(const DevScene& sc, int mat, const DevMaterial& m, const MatPrep& e, f3 VrL, float2 uv, float2 u,
                                  f3& dir, f3& weight, bool& may_leak) {
    if (GENERIC && !e.fast) { bxdf_sample_slow(sc, mat, VrL, uv, u, dir, weight, may_leak); return; }
    may_leak = false;
    const f3 zero = mk3(0.f, 0.f, 0.f);
    bool lobe = m.kind != RGK_BXDF_DIFFUSE; // LTC lobe, unless the diffuse branch is chosen below
    if (m.kind >= RGK_BXDF_LTC_BECKMANN_DIFFUSE) {
        float dp = e.diffc.x + e.diffc.y + e.diffc.z, sp = e.colorc.x + e.colorc.y + e.colorc.z;
        float prob = dp / (dp + sp + 0.0001f);
        if (decide_and_rescale(u.x, prob)) lobe = false;
    }
    if (!lobe) {
        if (VrL.z <= 0) { dir = mk3(0.f, 1.f, 0.f); weight = zero; return; }
        dir = hemisphere_cosine_z(u);
        weight = e.diffc;
        return;
    }
    if (!e.lobe) { dir = mk3(0.f, 0.f, 1.f); weight = zero; return; } // a black lobe: weight 0 ends the path, its direction is never read
    f3 v = ltc_random_M(e.Ms, VrL, hemisphere_cosine_z(u));
    dir = v;
    weight = (v.z <= 0) ? zero : e.colorc;
}
template <bool GENERIC = true>
__device__ __forceinline__ f3 mat_value(const DevScene& sc, int mat, const DevMaterial& m, const MatPrep& e, f3 ViL, f3 VrL, float2 uv) {
    if (GENERIC && !e.fast) return bxdf_value_slow(sc, mat, ViL, VrL, uv);
    if (ViL.z <= 0 || VrL.z <= 0) return mk3(0.f, 0.f, 0.f);
    if (m.kind == RGK_BXDF_DIFFUSE) return e.diffc / RGK_PI_F;
    if (!e.lobe) return (m.kind >= RGK_BXDF_LTC_BECKMANN_DIFFUSE) ? e.diffc / RGK_PI_F : mk3(0.f, 0.f, 0.f);
    float pdf = ltc_pdf_M(e.Mv, ViL, VrL);
    if (m.kind >= RGK_BXDF_LTC_BECKMANN_DIFFUSE) return e.colorc * pdf + e.diffc / RGK_PI_F;
    return e.colorc * pdf;
}

// bxdf_value(mat, Vi, Vr) for ANY pair of directions at a vertex whose MatPrep exists (the bidirectional connections
// evaluate the vertex's material once per light vertex, path_tracer.cpp:463-480): the texture colours fetched for
// the vertex are reused, only the LTC entry -- interpolated at angle(Vr, N) -- is looked up again.  Same values as
// bxdf_value_leaf; materials on the generic route go there.
template <bool GENERIC = true>
__device__ __forceinline__ f3 mat_value_at(const DevScene& sc, int mat, const DevMaterial& m, const MatPrep& e, f3 Vi, f3 Vr, float2 uv) {
    if (GENERIC && !e.fast) return bxdf_value_slow(sc, mat, Vi, Vr, uv);
    if (Vi.z <= 0 || Vr.z <= 0) return mk3(0.f, 0.f, 0.f);
    if (m.kind == RGK_BXDF_DIFFUSE) return e.diffc / RGK_PI_F;
    if (!e.lobe) return (m.kind >= RGK_BXDF_LTC_BECKMANN_DIFFUSE) ? e.diffc / RGK_PI_F : mk3(0.f, 0.f, 0.f);
    const uint32_t tab = (m.kind == RGK_BXDF_LTC_GGX || m.kind == RGK_BXDF_LTC_GGX_DIFFUSE) ? 0u : RGK_LTC_TABLE_BYTES;
    const float pdf = ltc_pdf(sc.ltc, tab, Vi, Vr, m.roughness);
    if (m.kind >= RGK_BXDF_LTC_BECKMANN_DIFFUSE) return e.colorc * pdf + e.diffc / RGK_PI_F;
    return e.colorc * pdf;
}

// bxdf_value of a material KNOWN to take the fast route (diffuse, LTC) given the two texture colours at its uv (the
// light vertex of a bidirectional connection stores them, rgk_bdpt.h): bxdf_value_leaf's cases for those kinds.
__device__ __forceinline__ f3 bxdf_value_fastkind(const DevScene& sc, const DevMaterial& m, float4 diffc, float4 colorc, f3 Vi, f3 Vr) {
    if (Vi.z <= 0 || Vr.z <= 0) return mk3(0.f, 0.f, 0.f);
    const f3 diff = mk3(diffc.x, diffc.y, diffc.z), spec = mk3(colorc.x, colorc.y, colorc.z);
    if (m.kind == RGK_BXDF_DIFFUSE) return diff / RGK_PI_F;
    if (is_zero3(spec)) return (m.kind >= RGK_BXDF_LTC_BECKMANN_DIFFUSE) ? diff / RGK_PI_F : mk3(0.f, 0.f, 0.f); // Q6
    const uint32_t tab = (m.kind == RGK_BXDF_LTC_GGX || m.kind == RGK_BXDF_LTC_GGX_DIFFUSE) ? 0u : RGK_LTC_TABLE_BYTES;
    if (m.kind >= RGK_BXDF_LTC_BECKMANN_DIFFUSE) return spec * ltc_pdf(sc.ltc, tab, Vi, Vr, m.roughness) + diff / RGK_PI_F;
    return spec * ltc_pdf(sc.ltc, tab, Vi, Vr, m.roughness);
}

// ------------------------------------------------------------------ lights / sky (a10, a15)
struct DLight {
    f3 pos, color, normal;
    float intensity, size;
    int type;  // 0 FULL_SPHERE, 1 HEMISPHERE, -1 none (Q15: zero contribution)
    int index; // point-light index, or index into areal_tris
};
__device__ __forceinline__ float light_dir_factor(const DLight& l, f3 v) {
    return l.type == 0 ? 1.0f : fmaxf(0.0f, dot3(v, l.normal));
}
// Scene::GetRandomLight + ArealLight::GetRandomLight + Triangle::GetRandomPoint, reference
// src/scene.cpp:686-745, src/primitives.cpp:61-73
__device__ __forceinline__ DLight random_light(const DevScene& sc, float2 choice, float light_sample, float2 tri_sample) {
    DLight L;
    L.type = -1;
    L.pos = L.color = L.normal = mk3(0.f, 0.f, 0.f);
    L.intensity = 0.f; L.size = 0.f; L.index = 0;
    float total_power = sc.total_point_power + sc.total_areal_power;
    if (total_power <= 0.0f) return L;
    float q = choice.x * total_power;
    if (q < sc.total_point_power) {
        for (uint32_t i = 0; i < sc.n_pointlights; i++) {
            const DevPointLight pl = gld_rec<DevPointLight>(sc.pointlights, i * (uint32_t)sizeof(DevPointLight));
            q -= pl.intensity * 4.0f * RGK_PI_F;
            if (q <= 0.0f) {
                L.type = 0;
                L.pos = mk3(pl.pos[0], pl.pos[1], pl.pos[2]);
                L.color = mk3(pl.color[0], pl.color[1], pl.color[2]);
                L.intensity = pl.intensity; L.size = pl.size; L.index = (int)i;
                return L;
            }
        }
        return L;
    }
    q = choice.y * sc.total_areal_power;
    for (uint32_t i = 0; i < sc.n_areal; i++) {
        const DevArealLight al = gld_rec<DevArealLight>(sc.areal, i * (uint32_t)sizeof(DevArealLight));
        q -= al.power;
        if (q <= 0.0f) {
            float p = light_sample * al.total_area;
            for (uint32_t j = 0; j < al.count; j++) {
                const DevArealTri att = gld_rec<DevArealTri>(sc.areal_tris, (al.first + j) * (uint32_t)sizeof(DevArealTri));
                const DevArealTri* at = &att;
                p -= at->area;
                if (p <= 0.0f) {
                    float2 r = tri_sample;
                    f3 a = mk3(at->a[0], at->a[1], at->a[2]);
                    f3 c = mk3(at->b[0], at->b[1], at->b[2]);
                    f3 b = mk3(at->c[0], at->c[1], at->c[2]);
                    f3 Va = a - c, Vb = b - c;
                    if (r.x + r.y > 1.0f) { r.x = 1.0f - r.x; r.y = 1.0f - r.y; }
                    L.type = 1;
                    L.pos = c + r.x * Va + r.y * Vb;
                    L.color = mk3(al.emission[0], al.emission[1], al.emission[2]);
                    L.intensity = 1.0f;
                    L.normal = mk3(at->normal_a[0], at->normal_a[1], at->normal_a[2]);
                    L.index = (int)(al.first + j);
                    return L;
                }
            }
            return L;
        }
    }
    return L;
}
// The path's light (TracePath picks ONE light per path, path_tracer.cpp:315-322,339-343) is sampled
// once in k_raygen and kept per slot as {pos.xyz, code}: code = point-light index, or
// 0x80000000 | areal-triangle index, or RGK_LIGHT_NONE.  Colour / normal / intensity are re-read
// from the (tiny, cached) light tables.
#define RGK_LIGHT_NONE 0x7fffffffu
__device__ __forceinline__ uint32_t light_code(const DevScene& sc, float2 choice, float light_sample, float2 tri_sample, f3& pos) {
    DLight L = random_light(sc, choice, light_sample, tri_sample);
    pos = L.pos;
    if (L.type < 0) return RGK_LIGHT_NONE;
    if (L.type == 0) {
        pos = L.pos + L.size * sphere_uniform(tri_sample); // path_tracer.cpp:339-342 (areal_sample reused)
        return (uint32_t)L.index;
    }
    return 0x80000000u | (uint32_t)L.index;
}
__device__ __forceinline__ DLight light_from_code(const DevScene& sc, f3 pos, uint32_t code) {
    DLight L;
    L.pos = pos;
    L.normal = L.color = mk3(0.f, 0.f, 0.f);
    L.intensity = 0.f; L.size = 0.f; L.index = 0;
    if (code == RGK_LIGHT_NONE) { L.type = -1; return L; }
    if (code & 0x80000000u) {
        const DevArealTri att = gld_rec<DevArealTri>(sc.areal_tris, (code & 0x7fffffffu) * (uint32_t)sizeof(DevArealTri));
        const DevArealLight all = gld_rec<DevArealLight>(sc.areal, att.light * (uint32_t)sizeof(DevArealLight));
        const DevArealTri* at = &att;
        const DevArealLight* al = &all;
        L.type = 1;
        L.color = mk3(al->emission[0], al->emission[1], al->emission[2]);
        L.intensity = 1.0f;
        L.normal = mk3(at->normal_a[0], at->normal_a[1], at->normal_a[2]);
        return L;
    }
    const DevPointLight plv = gld_rec<DevPointLight>(sc.pointlights, code * (uint32_t)sizeof(DevPointLight));
    const DevPointLight* pl = &plv;
    L.type = 0;
    L.color = mk3(pl->color[0], pl->color[1], pl->color[2]);
    L.intensity = pl->intensity; L.size = pl->size;
    return L;
}
// Scene::GetSkyboxRay, reference src/scene.cpp:748-763
__device__ __forceinline__ f3 skybox(const DevScene& sc, f3 direction) {
    if (sc.sky_mode == 0) return mk3(sc.sky_color[0], sc.sky_color[1], sc.sky_color[2]) * mk3(sc.sky_intensity, sc.sky_intensity, sc.sky_intensity);
    float alpha = rgk_asinf(direction.y);
    float beta = -rgk_atan2f(direction.x, direction.z);
    beta += sc.sky_rotate * 0.0174533f;
    float x = beta / (2.0f * RGK_PI_F) + 0.5f;
    float y = alpha / RGK_PI_F + 0.5f;
    f3 c = tex_get(sc, sc.sky_tex, make_float2(x, y));
    return c * mk3(sc.sky_intensity, sc.sky_intensity, sc.sky_intensity);
}

// ------------------------------------------------------------------ path vertex (shared by the BDPT kernels)
// The geometric half of one GeneratePath iteration, reference src/path_tracer.cpp:152-235: hit point,
// interpolated + normalised face normal (with the NaN fallbacks), uv, bump-tilted shading normal and
// the global->local frame.  `ok == false` is the reference's `return path` (vertex dropped, path ends).
struct Vertex {
    bool ok;
    f3 pos, faceN, lightN, Vr, VrL;
    float2 uv;
    uint32_t mat_id;
    DevMaterial mat;
    quatf g2l;
};
__device__ __forceinline__ f3 clamp3(f3 v, float c) { return mk3(v.x > c ? c : v.x, v.y > c ? c : v.y, v.z > c ? c : v.z); }

__device__ __forceinline__ void surface_point(const DevScene& sc, float bumpmap_scale, f3 o, f3 d, float4 h, Vertex& v) {
    const int tri = __float_as_int(h.w);
    const uint32_t tsr = (uint32_t)tri * (uint32_t)sizeof(TriShade);
    const float4 g0 = gld_f4(sc.tri_shade, tsr), g1 = gld_f4(sc.tri_shade, tsr + 16u), g2 = gld_f4(sc.tri_shade, tsr + 32u);
    v.mat_id = gld_u32(sc.tri_shade, tsr + 96u);
    v.mat = mat_load(sc, v.mat_id);
    const float al = h.y, be = h.z;
    const float ia = 1.0f - al - be, ib = al, ic = be; // Intersection::a,b,c scene_intersect.cpp:280-283
    v.Vr = -d;
    v.pos = o + h.x * d;
    f3 nA_ = mk3(g0.x, g0.y, g0.z), nB_ = mk3(g1.x, g1.y, g1.z), nC_ = mk3(g2.x, g2.y, g2.z);
    f3 faceN = ia * nA_ + ib * nB_ + ic * nC_;
    v.ok = true;
    if (faceN.x != faceN.x) { // NaN fallbacks, path_tracer.cpp:157-171
        faceN = nA_;
        if (faceN.x != faceN.x) { faceN = nB_; if (faceN.x != faceN.x) { faceN = nC_; if (faceN.x != faceN.x) v.ok = false; } }
    }
    if (v.ok && len3(faceN) <= 0.0f) v.ok = false; // path_tracer.cpp:175
    v.uv = make_float2(0.f, 0.f);
    v.faceN = v.lightN = faceN;
    if (!v.ok) return;
    faceN = norm3(faceN);
    const float4 g3 = gld_f4(sc.tri_shade, tsr + 48u), g4 = gld_f4(sc.tri_shade, tsr + 64u), g5 = gld_f4(sc.tri_shade, tsr + 80u);
    if (sc.has_texcoords) {
        v.uv.x = ia * g0.w + ib * g2.w + ic * g4.w;
        v.uv.y = ia * g1.w + ib * g3.w + ic * g5.w;
    }
    f3 lightN = faceN;
    if (tex_kind(v.mat.t_bump) != RGK_TEXREF_NONE) { // bump, path_tracer.cpp:204-231
        float right, bottom;
        tex_slopes(sc, v.mat.t_bump, v.uv, right, bottom);
        f3 tangent = ia * mk3(g3.x, g3.y, g3.z) + ib * mk3(g4.x, g4.y, g4.z) + ic * mk3(g5.x, g5.y, g5.z);
        if (!(tangent.x * tangent.x + tangent.y * tangent.y + tangent.z * tangent.z < 0.001f)) {
            tangent = norm3(tangent);
            f3 bitangent = norm3(cross3(faceN, tangent));
            f3 tangent2 = cross3(bitangent, faceN);
            lightN = norm3(faceN + (tangent2 * right + bitangent * bottom) * bumpmap_scale);
            if (lightN.x != lightN.x) lightN = faceN;
        }
    }
    v.faceN = faceN;
    v.lightN = lightN;
    v.g2l = rotation_between(lightN, mk3(0.f, 0.f, 1.f)); // SystemTransform(lightN, +Z), src/glm.hpp:21-24
    v.VrL = qrot(v.g2l, v.Vr);
}

// Camera::GetCoordsFromDirection, reference src/camera.cpp:48-83 (Q16: coordinates clamped to the frame)
__device__ __forceinline__ bool coords_from_direction(const DevCamera& cam, f3 dir, int& x, int& y) {
    const f3 N = mk3(cam.direction[0], cam.direction[1], cam.direction[2]);
    const f3 origin = mk3(cam.origin[0], cam.origin[1], cam.origin[2]);
    const f3 V = mk3(cam.viewscreen[0], cam.viewscreen[1], cam.viewscreen[2]);
    const f3 v1 = mk3(cam.viewscreen_x[0], cam.viewscreen_x[1], cam.viewscreen_x[2]);
    const f3 v2 = mk3(cam.viewscreen_y[0], cam.viewscreen_y[1], cam.viewscreen_y[2]);
    float q = dot3(dir, N);
    if ((double)q < 0.0001) return false;
    float t = dot3(V - origin, N) / q;
    if (t <= 0) return false;
    f3 p = origin + dir * t;
    f3 vp = p - V;
    float plen = len3(vp);
    float v1_cast_len = plen * (dot3(norm3(vp), norm3(v1)));
    float v2_cast_len = plen * (dot3(norm3(vp), norm3(v2)));
    float x_ratio = v1_cast_len / len3(v1);
    float y_ratio = v2_cast_len / len3(v2);
    if (x_ratio < 0.0f || x_ratio > 1.0f || y_ratio < 0.0f || y_ratio > 1.0f) return false;
    x = (int)(cam.xsize * x_ratio);
    y = (int)(cam.ysize * y_ratio);
    if (x > cam.xsize - 1) x = cam.xsize - 1;
    if (y > cam.ysize - 1) y = cam.ysize - 1;
    return true;
}
