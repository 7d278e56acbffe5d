// Launch interface between the host runtime (rgk_host.cpp) and the kernels (rgk_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.h"

#ifndef RGK_TRACE_BLOCK
#define RGK_TRACE_BLOCK 256
#endif
#ifndef RGK_SHADE_WAVES
#define RGK_SHADE_WAVES 4 // waves per SIMD the shade kernel is compiled for (128 VGPRs)
#endif
#ifndef RGK_SHADE_BLOCK
#define RGK_SHADE_BLOCK 512
#endif
#ifndef RGK_LIGHT_BLOCK
#define RGK_LIGHT_BLOCK RGK_SHADE_BLOCK // the light sub-path's kernels (rgk_bdpt.h)
#endif
#ifndef RGK_SHADE_BLOCK_LATER
#define RGK_SHADE_BLOCK_LATER 256 // k_shade at bounce >= 1 (see there)
#endif
#define RGK_MAX_DEPTH 62
#define RGK_LV_FLOAT4 6 // float4 per stored light vertex: {pos,mat}{lightN,u}{Vr,v}{light_from_source,valid}{diffuse colour}{specular colour}

// device counter block (uint32), zeroed per pass by k_init_counters
#define RGK_CNT_QUEUE 0     // [b]  rays in bounce b's queue        (b = 0..depth)
#define RGK_CNT_SHADOW 64   // [b]  shadow rays issued at bounce b
#define RGK_CNT_FETCH_T 128 // [b]  work-fetch cursor of k_trace_closest at bounce b
#define RGK_CNT_FETCH_S 192 // [b]  work-fetch cursor of k_trace_shadow at bounce b
#define RGK_CNT_GENERIC 256 // [b]  vertices of bounce b left to the generic-BxDF shade launch
#define RGK_CNT_HITS 320    // [k]  light sub-path: rays of bounce k that hit something (k_list_hits)
#define RGK_CNT_SRAYS 384   // [b]  bidirectional rounds: shadow rays traced at bounce b (RGK_CNT_SHADOW counts the vertices queued)
#define RGK_CNT_CONN 448    // [b]  bidirectional rounds: camera vertices of bounce b with connections (k_connect; = entries of the vertex queue)
#define RGK_CNT_FETCH_J 512 // [b]  work-fetch cursor of k_trace_shadow_jobs at bounce b
#define RGK_CNT_TOTAL 576

// what k_trace_shadow does with a visible ray's payload
#define RGK_SHADOW_ADD 0   // tot[slot] += radiance                     (uni-directional path)
#define RGK_SHADOW_SPLAT 2 // atomicAdd(accum_rgb[pixel], radiance)     (BDPT: light-tracing side effect)

#ifndef RGK_ENTRY_K
#define RGK_ENTRY_K 6 // entry nodes per pixel group (unused ones hold the traversal's stack sentinel)
#endif
#ifndef RGK_ENTRY_SHIFT
#define RGK_ENTRY_SHIFT 3 // log2 of the pixels per group: consecutive pixels of the round's list (8x8 blocks, row-major inside: a row of 8)
// (Sponza proxy, ms per round: off 146.6; K, pixels = 4, 64: 138.8; 8, 64: 137.5; 4, 16: 139.4; 4, 8: 139.1; 6, 8: 136.3; 8, 8: 136.2)
#endif
#define RGK_ENTRY_PIX (1u << RGK_ENTRY_SHIFT)
// One pass = pixels [j0, j0+npix) of the round's pixel list x samples [s0, s0+ns).
// Path slot = ((srel >> g) * npix + j) << g | (srel & (2^g - 1)) with srel = s - s0, j = pixel - j0, g = gshift: 2^g consecutive
// samples of a pixel sit side by side, so a wave of 64 slots is 64 >> g neighbouring pixels x 2^g samples -- rays that differ by
// sub-pixel jitter walk the same nodes and shade the same triangle and texels (g = 0: one sample of 64 pixels, round 1's
// order).  ns is a multiple of 2^g (the host falls back to g = 0 otherwise).
struct PassParams {
    uint32_t j0, npix, s0, ns;
    uint32_t gshift;
    uint32_t beam;            // bounce 0 of a pinhole camera with gshift == 3: one lane walks the tree for the 8 samples of a pixel (k_trace_camera_beam)
    uint32_t multisample, depth, xres, yres;
    float clamp, russian, bumpmap_scale;
    uint32_t reverse;
    const uint32_t* pix_xy;   // x | y << 16, per pixel of the round
    const uint32_t* pix_seed; // PathTracer::samplerSeed for that pixel (a2)
    const float* htab;        // halton_raw(hdim, s) for hdim < 192, s < multisample: htab[hdim * multisample + s]
    float4* light;            // per slot: the path's light {pos.xyz, code}, written by the first vertex of a path that goes on
    const float4* lbox;       // ... and the box {lo, hi} per group inside which a shadow ray must END to use them
    const int* lentry;        // first-vertex shadow rays of a single-light scene: entry nodes per pixel group (k_entry_points_light), or null
    const float* entry_cap;   // ... and per group the distance up to which that list is complete (+inf: all the way)
    const int* entry;         // camera rays: RGK_ENTRY_K node refs per group of RGK_ENTRY_PIX consecutive pixels of the round's list (k_entry_points), or null
    // bidirectional state (reverse > 0), null otherwise; per slot with stride `batch`
    float4* lstart;           // light_at_path_start.rgb
    float4* lv;               // light vertices: lv[(k*RGK_LV_FLOAT4 + c) * batch + slot], c: see RGK_LV_FLOAT4
    uint32_t* hitlist;        // light sub-path: queue indices of the rays that hit something (k_list_hits)
    uint32_t* lvmask;         // per slot: bit k = light vertex k exists, bit 7 = one of them has a generic-route material
    float4* conn;             // camera vertices with connections: conn[c * batch + i] for queue index i, c < 6 (k_shade<BDPT> -> k_connect)
    uint32_t* connlist;       // ... and the queue indices that have one
    uint32_t batch;
    uint32_t* generic;        // queue indices of the vertices whose material takes the generic BxDF route (k_shade)
};

// slot <-> (pixel of the pass, sample of the pass)
__device__ __forceinline__ void slot_decode(const PassParams& pp, uint32_t slot, uint32_t& j, uint32_t& srel) {
    const uint32_t r = slot >> pp.gshift, sb = r / pp.npix;
    j = r - sb * pp.npix;
    srel = (sb << pp.gshift) | (slot & ((1u << pp.gshift) - 1u));
}
__device__ __forceinline__ uint32_t slot_of(const PassParams& pp, uint32_t j, uint32_t srel) {
    return ((((srel >> pp.gshift) * pp.npix) + j) << pp.gshift) | (srel & ((1u << pp.gshift) - 1u));
}

void rgk_launch_entry_points(hipStream_t st, const DevScene& sc, const DevCamera& cam, uint32_t xres, uint32_t yres, const uint32_t* pix_xy, uint32_t n_pixels,
                             uint32_t g_first, uint32_t g_count, const uint32_t* trange, int* entries, float* cap);
void rgk_launch_group_trange(hipStream_t st, const PassParams& pp, const float4* hit, uint32_t* trange);
void rgk_launch_light_entry_points(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PassParams& pp, uint32_t n_pixels_round, const uint32_t* trange, int* entries, float4* lbox);
void rgk_launch_stage_mark(hipStream_t st, uint32_t* host_word, uint32_t v); // progress: the device writes v to pinned host memory
void rgk_launch_init_counters(hipStream_t st, uint32_t* counters, uint32_t n0);
void rgk_launch_build_pixel_list(hipStream_t st, const rgk_tile* tiles, const uint32_t* tile_off, uint32_t n_tiles, uint32_t* pix_xy, uint32_t* pix_seed);
void rgk_launch_build_halton_table(hipStream_t st, const DevScene& sc, uint32_t S, float* htab);
// traversal-stack configuration of a scene: entries its tree can need, how many of them live in LDS, overflow area
struct RgkTraceCfg {
    int stack, lds;
    int* ovf;
};
int rgk_trace_grid(int lds_entries);
// upper bounds on the queue lengths the following launches consume (grids shrink accordingly); 0xffffffff = unknown
void rgk_launch_set_bound(uint32_t items, uint32_t shadow_items); // workgroups of a persistent trace launch (LDS-limited residency x 256 CUs)
void rgk_launch_trace_camera(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PassParams& pp, const RgkTraceCfg& tc, bool count_stats, float4* hit,
                             const uint32_t* count_ptr, uint32_t* fetch, unsigned long long* stats);
void rgk_launch_trace_closest(hipStream_t st, const DevScene& sc, const RgkTraceCfg& tc, bool count_stats, const float4* rayA, const float4* rayB,
                              const float2* nearfar, float4* hit, const uint32_t* count_ptr, uint32_t* fetch, unsigned long long* stats);
void rgk_launch_trace_shadow(hipStream_t st, const DevScene& sc, const RgkTraceCfg& tc, bool count_stats, const float4* shA, const float4* shB,
                             const float4* shC, float4* tot, uint8_t* vis_out, int mode, float* splat_rgb, const uint32_t* count_ptr,
                             uint32_t* fetch, unsigned long long* stats);
void rgk_launch_trace_shadow_first(hipStream_t st, const DevScene& sc, const PassParams& pp, const RgkTraceCfg& tc, bool count_stats, const float4* shA, const float4* shB,
                                   const float4* shC, float4* tot, const uint32_t* count_ptr, uint32_t* fetch, unsigned long long* stats);
void rgk_launch_raygen_light(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PassParams& pp, float4* rayA, float4* rayB,
                             float4* thr, uint32_t* counters);
void rgk_launch_shade_light(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PassParams& pp, uint32_t k, const float4* rayA,
                            const float4* rayB, const float4* hit, float4* thr, float4* nextA, float4* nextB, float4* shA, float4* shB,
                            float4* shC, uint32_t* counters);
void rgk_launch_list_hits(hipStream_t st, const float4* hit, const uint32_t* count_ptr, uint32_t* list, uint32_t* list_count);
void rgk_launch_connect(hipStream_t st, const DevScene& sc, const PassParams& pp, uint32_t bounce, float4* jobs, float4* rads, uint32_t* counters);
void rgk_launch_trace_shadow_jobs(hipStream_t st, const DevScene& sc, const PassParams& pp, const RgkTraceCfg& tc, bool count_stats, const float4* jobs, const float4* rads,
                                  float4* tot, const uint32_t* count_ptr, uint32_t* fetch, unsigned long long* stats);
void rgk_launch_shade(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PassParams& pp, uint32_t bounce, const float4* rayA,
                      const float4* rayB, const float4* hit, float4* thr, float4* tot, float4* nextA, float4* nextB, float4* shA,
                      float4* shB, float4* shC, uint32_t* counters, bool bdpt = false);
void rgk_launch_resolve(hipStream_t st, const PassParams& pp, const float4* tot, float4* pixsum, float* accum_rgb, uint32_t* accum_count);
void rgk_launch_pack_rays(hipStream_t st, uint32_t n, const float* rays, const int32_t* ignore, float4* rayA, float4* rayB, float2* nearfar);
void rgk_launch_pack_visibility(hipStream_t st, const DevScene& sc, uint32_t n, const float* a, const float* b, float4* shA, float4* shB,
                                float4* shC);
void rgk_launch_unpack_hits(hipStream_t st, uint32_t n, const float4* hit, rgk_hit* out);
void rgk_launch_sampler_eval(hipStream_t st, const DevScene& sc, uint32_t n, const uint32_t* seed, const uint32_t* index,
                             const uint32_t* dim, int is2d, float* out);

void rgk_launch_bxdf_value(hipStream_t st, const DevScene& sc, uint32_t n, uint32_t route, const uint32_t* mat, const float* Vi, const float* Vr, const float* uv, float* out);
void rgk_launch_bxdf_sample(hipStream_t st, const DevScene& sc, uint32_t n, uint32_t route, const uint32_t* mat, const float* Vi, const float* uv, const float* u,
                            float* out_dir, float* out_w, uint8_t* leak);
void rgk_launch_texture_sample(hipStream_t st, const DevScene& sc, uint32_t n, const TexRef* refs, const int32_t* tex, const float* uv, float* rgb, float* sr, float* sb);
void rgk_launch_libm_eval(hipStream_t st, int fn, uint32_t n, const float* a, const float* b, float* out);
