// Bidirectional half of PathTracer::TracePath (reverse > 0): the light sub-path of fixed length
// `reverse`, its light-tracing splats, and the connections of every camera-path vertex to every
// light-path vertex.  Reference src/path_tracer.cpp:336-349 (sub-path), :359-398 (phase 2,
// side effects), :463-480 (connections), :485-496 (emission, clamp, accumulate).
//
// Pass structure on the GPU (the light sub-path does not depend on the camera path once its
// sampler dimensions are pinned):
//   k_raygen_light -> [k_trace_closest -> k_list_hits -> k_shade_light -> k_trace_shadow(splat)] x reverse
//   [k_trace_camera | k_trace_closest -> k_shade<BDPT> -> k_connect -> k_trace_shadow(add) + k_trace_shadow_jobs] x depth
// Light vertices live per slot in pp.lv, which of them exist in pp.lvmask.  The camera path is the unidirectional pipeline
// (rgk_kernels.hip k_shade: no ray-generation kernel, camera rays made where they are traced and shaded) -- a vertex whose
// slot has no light vertex, and that does not emit, IS a unidirectional vertex: its NEE ray goes to the plain shadow queue.
// The others leave a record; k_connect evaluates their connections and queues ONE entry per vertex (the vertex, its
// contribution, one radiance per ray: the 1 + reverse shadow rays all end at the vertex and start at points the slot already
// holds), and k_trace_shadow_jobs traces the rays one after the other and forms clamp(NEE + connections + emission) *
// contribution in the reference's order itself (rgk_trace.h, JOB).  Round 2 shaded every camera vertex in a kernel of its own
// that looped over the light vertices (128 VGPRs + 152 bytes of scratch, 2.5 x the time of a unidirectional vertex), queued
// 48 bytes and a 16-byte result cell per RAY, and added the cells up in another pass (k_finish_vertex, 12 % of a round).
// Most light rays of a light far outside the geometry miss the scene: k_list_hits compacts the indices of those that hit,
// so the light-vertex shading runs over full waves (it had 8 % of its lanes busy).
#pragma once
#include <hip/hip_runtime.h>
#include "rgk_device.h"
#include "rgk_kernels.h"

// Workgroup-level append: returns this lane's position in the queue behind `counter` (valid only
// when flag).  Two barriers; every thread of the workgroup must call it the same number of times.
__device__ __forceinline__ uint32_t block_append(bool flag, uint32_t* counter, uint32_t* s_cnt, uint32_t* s_base) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) s_cnt[w] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < RGK_LIGHT_BLOCK / 64; k++) { uint32_t a = s_cnt[k]; s_cnt[k] = t; t += a; }
        *s_base = t ? atomicAdd(counter, t) : 0u;
    }
    __syncthreads();
    const uint32_t p = *s_base + s_cnt[w] + __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    return p;
}

__device__ __forceinline__ float4* lv_ptr(const PassParams& pp, uint32_t k, uint32_t c, uint32_t slot) {
    return pp.lv + ((size_t)(k * RGK_LV_FLOAT4 + c) * pp.batch + slot);
}

// ---- light ray, path_tracer.cpp:336-349,359-363
// A light ray that cannot touch the scene's box can never make a light vertex, and with a light far outside the geometry
// that is most of them (configs[3]: nine in ten): only the others are queued (same clip, same decision as the traversal
// kernel's own, rgk_trace.h clip_to_scene) -- the first traversal launch, the hit list and the ray records shrink accordingly.
// The reference counts every light ray it traces (raycount++, :126); the host adds the culled ones back (RGK_CNT_CULLED).
__global__ __launch_bounds__(RGK_LIGHT_BLOCK) void k_raygen_light(const DevScene sc, const DevCamera cam, const PassParams pp, float4* __restrict__ rayA,
                                                                   float4* __restrict__ rayB, float4* __restrict__ thr, uint32_t* __restrict__ counters) {
    const uint32_t n = pp.npix * pp.ns;
    const SamplerTab tb = {pp.htab, pp.multisample};
    // The queued rays are STAGED in LDS and written out a full workgroup's worth at a time: one reservation per 512 QUEUED rays.
    // Reserving per 512 slots -- nine in ten of which queue nothing on configs[3] -- meant 2 M returning atomics on one word per
    // round, and at ~88 per microsecond chip-wide those atomics WERE the kernel's 24 ms.  (The per-slot records -- light, start
    // colour, throughput, the vertex mask -- are indexed by slot, not by queue position, and are written where they are made.)
    __shared__ uint32_t s_cnt[RGK_LIGHT_BLOCK / 64];
    __shared__ uint32_t s_base;
    __shared__ float4 st_a[2 * RGK_LIGHT_BLOCK], st_b[2 * RGK_LIGHT_BLOCK];
    uint32_t fill = 0; // workgroup-uniform: rays staged so far (< RGK_LIGHT_BLOCK at the top of an iteration)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * RGK_LIGHT_BLOCK; base < n; base += gridDim.x * RGK_LIGHT_BLOCK) {
        const uint32_t slot = base + threadIdx.x;
        bool queue = false;
        float4 a = make_float4(0, 0, 0, 0), b = a, li = a, ls = a;
        if (slot < n) {
            uint32_t srel, j; slot_decode(pp, slot, j, srel);
            const uint32_t seed = pp.pix_seed[pp.j0 + j], s = pp.s0 + srel;
            const uint32_t base2d = (cam.lens_size != 0.0f) ? 2u : 1u;
            const float2 areal_s = sample2d_t(tb, seed, s, base2d), lightdir_s = sample2d_t(tb, seed, s, base2d + 1u);
            f3 lpos;
            const uint32_t lcode = light_code(sc, sample2d_t(tb, seed, s, base2d + 2u), sample1d_t(tb, seed, s, 0u), areal_s, lpos);
            li = make_float4(lpos.x, lpos.y, lpos.z, __uint_as_float(lcode));
            DLight L = light_from_code(sc, lpos, lcode);
            f3 normal = L.normal, ldir = mk3(0.f, 1.f, 0.f);
            if (L.type == 0) {
                const f3 dir = sphere_uniform(areal_s);
                normal = (L.size > 0.0f) ? dir : mk3(0.f, 0.f, 0.f); // Q15 (defined)
                ldir = qrot(rotation_from_y(norm3(dir)), hemisphere_cosine_y(lightdir_s));
            } else if (L.type == 1) {
                ldir = qrot(rotation_from_y(L.normal), hemisphere_cosine_y(lightdir_s));
            }
            L.normal = normal;
            const f3 o = lpos + sc.epsilon * normal * 100.0f;
            const f3 d = norm3(ldir);
            const float k = (L.type < 0) ? 0.0f : L.intensity * light_dir_factor(L, ldir);
            const f3 start = L.color * mk3(k, k, k);
            ls = make_float4(start.x, start.y, start.z, 0.f);
            a = make_float4(o.x, o.y, o.z, d.x);
            b = make_float4(d.y, d.z, __uint_as_float(0xffffffffu), __uint_as_float(slot));
            pp.lvmask[slot] = 0u; // no light vertex yet
            const bool nan_ray = (o.x != o.x) | (o.y != o.y) | (o.z != o.z) | (d.x != d.x) | (d.y != d.y) | (d.z != d.z);
            float t0, t1;
            queue = !nan_ray && clip_to_scene(sc, o, d, 0.0f, 10000.0f, t0, t1); // Ray::near / Ray::far defaults, src/ray.hpp:25-26
        }
        const unsigned long long m = __ballot(queue);
        if (lane == 0) s_cnt[w] = __popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < RGK_LIGHT_BLOCK / 64; k++) { const uint32_t c = s_cnt[k]; before += k < w ? c : 0u; total += c; }
        if (queue) {
            const uint32_t at = fill + before + __popcll(m & ((1ull << lane) - 1ull));
            st_a[at] = a; st_b[at] = b;
            pp.light[slot] = li; pp.lstart[slot] = ls;
            thr[slot] = make_float4(1.f, 1.f, 1.f, __uint_as_float(1u << 16));
        }
        fill += total;
        __syncthreads(); // the staged rays are visible, s_cnt may be written again
        if (fill >= RGK_LIGHT_BLOCK) {
            if (threadIdx.x == 0) s_base = atomicAdd(&counters[RGK_CNT_QUEUE], fill);
            __syncthreads();
            const uint32_t q0 = s_base;
            for (uint32_t k = threadIdx.x; k < fill; k += RGK_LIGHT_BLOCK) { rayA[q0 + k] = st_a[k]; rayB[q0 + k] = st_b[k]; }
            fill = 0;
            __syncthreads();
        }
    }
    if (fill) {
        if (threadIdx.x == 0) s_base = atomicAdd(&counters[RGK_CNT_QUEUE], fill);
        __syncthreads();
        const uint32_t q0 = s_base;
        for (uint32_t k = threadIdx.x; k < fill; k += RGK_LIGHT_BLOCK) { rayA[q0 + k] = st_a[k]; rayB[q0 + k] = st_b[k]; }
    }
}

// The sampling half of a GeneratePath iteration (path_tracer.cpp:239-300) shared by both sub-paths.
struct Step {
    bool go;
    f3 cum, no, nd;
};
template <bool GENERIC>
__device__ __forceinline__ void path_step(const DevScene& sc, const SamplerTab& tb, const Vertex& v, const MatPrep& mp, uint32_t seed, uint32_t s,
                                 uint32_t dim2d, uint32_t n, uint32_t depth, float russian, f3 cum, uint32_t& c1, Step& st) {
    st.go = false; st.cum = cum; st.no = st.nd = mk3(0.f, 0.f, 0.f);
    if (!(n < depth)) return; // last vertex: nothing sampled here can be observed
    const quatf l2g = qinverse(v.g2l);
    float2 u = sample2d_t(tb, seed, s, dim2d);
    f3 dirL, weight; bool may_leak;
    mat_sample<GENERIC>(sc, (int)v.mat_id, v.mat, mp, v.VrL, v.uv, u, dirL, weight, may_leak);
    const bool inside = dirL.z < 0;
    const f3 dir = qrot(l2g, dirL);
    uint32_t n_eff = n;
    if (!(dot3(dir, v.faceN) * dot3(v.Vr, v.faceN) > 0) && !may_leak) n_eff += 10000u;
    const bool no_russian = (v.mat.flags & RGK_MAT_NO_RUSSIAN) != 0;
    const float rc = (!no_russian && russian > 0.0f && n_eff > 1u) ? 1.0f / russian : 1.0f;
    cum = cum * rc;
    cum = cum * weight;
    st.cum = cum;
    bool go = !(max3c(cum) < 0.001f);
    if (go && !no_russian && russian >= 0.0f) {
        float r = sample1d_t(tb, seed, s, c1);
        c1++;
        if (r > russian) go = false;
    }
    if (go && !(n_eff < depth)) go = false;
    if (go) {
        st.no = v.pos + v.faceN * sc.epsilon * 10.0f * (inside ? -1.0f : 1.0f);
        st.nd = norm3(norm3(dir));
    }
    st.go = go;
}

// Queue indices of the rays that hit something, in queue order inside a workgroup's chunk (the order only decides which lanes
// shade which vertex): ballot + prefix popcount per wave, waves added through LDS, one atomic per workgroup.
__global__ __launch_bounds__(RGK_LIGHT_BLOCK) void k_list_hits(const float4* __restrict__ hit, const uint32_t* __restrict__ count_ptr,
                                                                uint32_t* __restrict__ list, uint32_t* __restrict__ list_count) {
    const uint32_t count = *count_ptr;
    __shared__ uint32_t s_cnt[RGK_LIGHT_BLOCK / 64];
    __shared__ uint32_t s_base;
    for (uint32_t base = blockIdx.x * RGK_LIGHT_BLOCK; base < count; base += gridDim.x * RGK_LIGHT_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const bool h = i < count && __float_as_int(hit[i].w) >= 0;
        const uint32_t p = block_append(h, list_count, s_cnt, &s_base);
        if (h) list[p] = i;
    }
}

// GENERIC = false / true: as in k_shade -- the first launch shades every vertex whose materials all take the fast BxDF
// route and lists the others (queue indices in pp.generic), the second walks that list with the full BxDF code.
// ---- light sub-path vertex k: store it, splat it to the camera, continue (russian = -1: no roulette)
template <bool GENERIC>
__global__ __launch_bounds__(RGK_LIGHT_BLOCK, 4) void k_shade_light(const DevScene sc, const DevCamera cam, const PassParams pp, const uint32_t k,
                                                                     const float4* __restrict__ rayA, const float4* __restrict__ rayB,
                                                                     const float4* __restrict__ hit, float4* __restrict__ thr,
                                                                     float4* __restrict__ nextA, float4* __restrict__ nextB, float4* __restrict__ shA,
                                                                     float4* __restrict__ shB, float4* __restrict__ shC, uint32_t* __restrict__ counters) {
    const uint32_t count = counters[(GENERIC ? RGK_CNT_GENERIC : RGK_CNT_HITS) + k]; // (the fast launch walks k_list_hits's list)
    const float eps = sc.epsilon;
    const SamplerTab tb = {pp.htab, pp.multisample};
    __shared__ uint32_t s_cnt[RGK_LIGHT_BLOCK / 64];
    __shared__ uint32_t s_base;
    lut_lds_fill(sc);
    for (uint32_t base = blockIdx.x * RGK_LIGHT_BLOCK; base < count; base += gridDim.x * RGK_LIGHT_BLOCK) {
        const bool valid = base + threadIdx.x < count;
        const uint32_t i = !valid ? 0u : (GENERIC ? pp.generic[base + threadIdx.x] : pp.hitlist[base + threadIdx.x]);
        bool cont = false, splat = false, defer = false;
        float4 nA = make_float4(0, 0, 0, 0), nB = nA, sA = nA, sB = nA, sC = nA;
        if (valid) {
            const float4 a = rayA[i], b = rayB[i], h = hit[i];
            const uint32_t slot = __float_as_uint(b.w);
            const f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
            const float4 st4 = thr[slot];
            f3 cum = mk3(st4.x, st4.y, st4.z);
            const uint32_t n = k + 1u;
            uint32_t c1 = 0;
            if (__float_as_int(h.w) >= 0) {
                Vertex v;
                surface_point(sc, pp.bumpmap_scale, o, d, h, v);
                defer = !GENERIC && v.ok && !mat_is_fast(v.mat.kind);
                if (v.ok && !defer) {
                    uint32_t srel, j; slot_decode(pp, slot, j, srel);
                    const uint32_t seed = pp.pix_seed[pp.j0 + j], s = pp.s0 + srel;
                    const uint32_t base2d = (cam.lens_size != 0.0f) ? 2u : 1u;
                    MatPrep mp;
                    mat_prepare(sc, v.mat, v.uv, v.VrL, n < pp.reverse, mp);
                    const f3 contribution = cum;
                    const float4 ls = pp.lstart[slot];
                    const f3 light_here = contribution * mk3(ls.x, ls.y, ls.z); // p.contribution * light_at_path_start, :370
                    *lv_ptr(pp, k, 0, slot) = make_float4(v.pos.x, v.pos.y, v.pos.z, __uint_as_float(v.mat_id));
                    *lv_ptr(pp, k, 1, slot) = make_float4(v.lightN.x, v.lightN.y, v.lightN.z, v.uv.x);
                    *lv_ptr(pp, k, 2, slot) = make_float4(v.Vr.x, v.Vr.y, v.Vr.z, v.uv.y);
                    *lv_ptr(pp, k, 3, slot) = make_float4(light_here.x, light_here.y, light_here.z, 1.0f);
                    // the vertex's texture colours: every camera vertex that connects to it would fetch them again
                    *lv_ptr(pp, k, 4, slot) = make_float4(mp.diffc.x, mp.diffc.y, mp.diffc.z, 0.f);
                    *lv_ptr(pp, k, 5, slot) = make_float4(mp.colorc.x, mp.colorc.y, mp.colorc.z, 0.f);
                    // bit k: light vertex k exists; bit 7: one of them has a material on the generic BxDF route (one vertex per slot and launch: no race)
                    pp.lvmask[slot] |= (1u << k) | (GENERIC ? 0x80u : 0u);
                    // phase 2: connect to the camera, :377-397.  camerapos = r.origin of this sample.
                    f3 campos = mk3(cam.origin[0], cam.origin[1], cam.origin[2]);
                    if (cam.lens_size != 0.0f) {
                        const float2 dsc = disc_uniform(sample2d_t(tb, seed, s, 1));
                        campos = campos + (dsc.x * cam.lens_size) * mk3(cam.left[0], cam.left[1], cam.left[2]) +
                                 (dsc.y * cam.lens_size) * mk3(cam.up[0], cam.up[1], cam.up[2]);
                    }
                    {
                        const f3 direction = norm3(v.pos - campos);
                        f3 q = light_here * mat_value_at<GENERIC>(sc, (int)v.mat_id, v.mat, mp, v.VrL, qrot(v.g2l, -direction), v.uv);
                        const f3 dd = v.pos - campos;
                        const float G = fmaxf(0.0f, dot3(v.lightN, -direction)) / dot3(dd, dd);
                        int x2, y2;
                        if (G >= 0.00001f && !(q.x != q.x) && coords_from_direction(cam, direction, x2, y2)) {
                            q = q * mk3(G, G, G);
                            const f3 diff = campos - v.pos; // Ray(p.pos, camerapos, 20 eps)
                            const f3 sd = norm3(diff);
                            splat = true;
                            sA = make_float4(v.pos.x, v.pos.y, v.pos.z, sd.x);
                            sB = make_float4(sd.y, sd.z, len3(diff) - eps * 20.0f, __uint_as_float((uint32_t)y2 * pp.xres + (uint32_t)x2));
                            sC = make_float4(q.x, q.y, q.z, 0.0f + eps * 20.0f);
                        }
                    }
                    Step st;
                    path_step<GENERIC>(sc, tb, v, mp, seed, s, base2d + 3u + pp.depth + (n - 1u), n, pp.reverse, -1.0f, cum, c1, st);
                    if (st.go) {
                        cont = true;
                        nA = make_float4(st.no.x, st.no.y, st.no.z, st.nd.x);
                        nB = make_float4(st.nd.y, st.nd.z, h.w, __uint_as_float(slot));
                        thr[slot] = make_float4(st.cum.x, st.cum.y, st.cum.z, st4.w);
                    }
                }
            }
        }
        const uint32_t pn = block_append(cont, &counters[RGK_CNT_QUEUE + k + 1], s_cnt, &s_base);
        if (cont) { nextA[pn] = nA; nextB[pn] = nB; }
        const uint32_t ps = block_append(splat, &counters[RGK_CNT_SHADOW + k], s_cnt, &s_base);
        if (splat) { shA[ps] = sA; shB[ps] = sB; shC[ps] = sC; }
        if (!GENERIC) {
            const uint32_t pd = block_append(defer, &counters[RGK_CNT_GENERIC + k], s_cnt, &s_base);
            if (defer) pp.generic[pd] = i;
        }
    }
}

// ---- connections of a camera-path vertex (path_tracer.cpp:463-480) from the record k_shade<BDPT> left: its material towards
// every light vertex of its slot x the light vertex's material towards it x G x the light arriving there.  One thread per
// record; the records are rare (slots whose light sub-path hit the scene), so this kernel may be as heavy as it likes.
// jobs / rads: the vertex queue of k_trace_shadow_jobs (rgk_trace.h JOB): jobs[t] = {p, slot}, jobs[batch + t] = {contribution, mask},
// jobs[2 batch + t] = {emission} (when mask bit 8 is set), jobs[3 batch + t] = {where the NEE ray starts}, rads[q batch + t] =
// radiance of ray q (q = 0 NEE, q = 1..reverse the light vertices) for the bits q of mask.  A ray whose radiance is exactly
// zero is not traced (it could only add zero).
__global__ __launch_bounds__(256) void k_connect(const DevScene sc, const PassParams pp, const uint32_t bounce, float4* __restrict__ jobs,
                                                  float4* __restrict__ rads, uint32_t* __restrict__ counters) {
    const uint32_t count = counters[RGK_CNT_CONN + bounce];
    lut_lds_fill(sc);
    const size_t bs = pp.batch;
    uint32_t n_rays = 0; // shadow rays this thread queued (counters[RGK_CNT_SRAYS + bounce]: the queue itself counts vertices)
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
        const uint32_t i = pp.connlist[t];
        const float4 c0 = pp.conn[i], c1 = pp.conn[bs + i], c2 = pp.conn[2 * bs + i], c3 = pp.conn[3 * bs + i], c4 = pp.conn[4 * bs + i], c5 = pp.conn[5 * bs + i];
        const f3 pos = mk3(c0.x, c0.y, c0.z), lightN = mk3(c1.x, c1.y, c1.z), VrL = mk3(c3.x, c3.y, c3.z);
        const uint32_t slot = __float_as_uint(c0.w), mat_id = __float_as_uint(c1.w);
        quatf g2l; g2l.x = c2.x; g2l.y = c2.y; g2l.z = c2.z; g2l.w = c2.w;
        const float2 uv = make_float2(c3.w, c4.w);
        const uint32_t lvm = pp.lvmask[slot];
        const DevMaterial mat = mat_load(sc, mat_id);
        // the fast route needs every material involved to be a diffuse / LTC one; otherwise the full BxDF code (the same values)
        const bool fast = mat_is_fast(mat.kind) && !(lvm & 0x80u);
        MatPrep mp;
        mp.fast = false;
        if (fast) mat_prepare(sc, mat, uv, VrL, false, mp);
        uint32_t mask = 0;
        if (c5.x != 0.f || c5.y != 0.f || c5.z != 0.f) {
            mask |= 1u;
            rads[t] = make_float4(c5.x, c5.y, c5.z, 0.f);
            jobs[3 * bs + t] = pp.light[slot]; // Ray(light.pos, p.pos): the stored position IS Light::pos
        }
        for (uint32_t q = 1; q <= pp.reverse; q++) {
            if (!(lvm & (1u << (q - 1u)))) continue;
            const float4 l0 = *lv_ptr(pp, q - 1, 0, slot), l1 = *lv_ptr(pp, q - 1, 1, slot), l2 = *lv_ptr(pp, q - 1, 2, slot), l3 = *lv_ptr(pp, q - 1, 3, slot);
            const f3 lpos = mk3(l0.x, l0.y, l0.z), lN = mk3(l1.x, l1.y, l1.z), lVr = mk3(l2.x, l2.y, l2.z);
            const float2 luv = make_float2(l1.w, l2.w);
            const f3 light_to_p = norm3(pos - lpos);
            const f3 p_to_light = -light_to_p;
            const quatf lg2l = rotation_between(lN, mk3(0.f, 0.f, 1.f));
            f3 f_light, f_point;
            if (fast) {
                f_light = bxdf_value_fastkind(sc, mat_load(sc, __float_as_uint(l0.w)), *lv_ptr(pp, q - 1, 4, slot), *lv_ptr(pp, q - 1, 5, slot), qrot(lg2l, light_to_p), qrot(lg2l, lVr));
                f_point = mat_value_at<false>(sc, (int)mat_id, mat, mp, VrL, qrot(g2l, p_to_light), uv);
            } else {
                f_light = bxdf_value_slow(sc, (int)__float_as_uint(l0.w), qrot(lg2l, light_to_p), qrot(lg2l, lVr), luv);
                f_point = bxdf_value_slow(sc, (int)mat_id, VrL, qrot(g2l, p_to_light), uv);
            }
            const f3 dd = pos - lpos;
            const float G = fabsf(dot3(lightN, p_to_light)) / dot3(dd, dd);
            const f3 rad = mk3(l3.x, l3.y, l3.z) * (f_light * f_point * G);
            if (rad.x != 0.f || rad.y != 0.f || rad.z != 0.f) {
                mask |= 1u << q;
                rads[(size_t)q * bs + t] = make_float4(rad.x, rad.y, rad.z, 0.f);
            }
        }
        if (__float_as_uint(c5.w) & 1u) { // front-facing and emitting
            mask |= 0x100u;
            jobs[2 * bs + t] = make_float4(mat.emission[0], mat.emission[1], mat.emission[2], 0.f);
        }
        jobs[t] = c0;
        jobs[bs + t] = make_float4(c4.x, c4.y, c4.z, __uint_as_float(mask));
        n_rays += (uint32_t)__popc(mask & 0xffu);
    }
    for (int ofs = 32; ofs > 0; ofs >>= 1) n_rays += __shfl_xor(n_rays, ofs);
    if ((threadIdx.x & 63) == 0 && n_rays) atomicAdd(&counters[RGK_CNT_SRAYS + bounce], n_rays);
}
