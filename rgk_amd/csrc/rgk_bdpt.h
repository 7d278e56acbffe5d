// Bidirectional half of PathTracer::TracePath (reverse > 0): the light sub-path of fixed length
// `reverse`, its light-tracing splats, and the connections of every camera-path vertex to every
// light-path vertex.  Reference src/path_tracer.cpp:336-349 (sub-path), :359-398 (phase 2,
// side effects), :463-480 (connections), :485-496 (emission, clamp, accumulate).
//
// Pass structure on the GPU (the light sub-path does not depend on the camera path once its
// sampler dimensions are pinned):
//   k_raygen_light -> [k_trace_closest -> k_shade_light -> k_trace_shadow(splat)] x reverse
//   k_raygen<BDPT> -> [k_trace_closest -> k_shade_bdpt -> k_trace_shadow(cell) -> k_finish_vertex] x depth
// Light vertices live per slot in pp.lv; each camera vertex issues 1 + reverse shadow rays whose
// radiance lands in pp.term[q][slot]; k_finish_vertex then forms clamp(NEE + connections + emission)
// * contribution in the reference's order.
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
        for (int k = 0; k < RGK_SHADE_BLOCK / 64; k++) { uint32_t a = s_cnt[k]; s_cnt[k] = t; t += a; }
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
__global__ __launch_bounds__(256) void k_raygen_light(const DevScene sc, const DevCamera cam, const PassParams pp, float4* __restrict__ rayA,
                                                       float4* __restrict__ rayB, float4* __restrict__ thr) {
    const uint32_t n = pp.npix * pp.ns;
    const SamplerTab tb = {pp.htab, pp.multisample};
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += gridDim.x * blockDim.x) {
        uint32_t srel, j; slot_decode(pp, slot, j, srel);
        const uint32_t seed = pp.pix_seed[pp.j0 + j], s = pp.s0 + srel;
        const uint32_t base2d = (cam.lens_size != 0.0f) ? 2u : 1u;
        const float2 areal_s = sample2d_t(tb, seed, s, base2d), lightdir_s = sample2d_t(tb, seed, s, base2d + 1u);
        f3 lpos;
        const uint32_t lcode = light_code(sc, sample2d_t(tb, seed, s, base2d + 2u), sample1d_t(tb, seed, s, 0u), areal_s, lpos);
        pp.light[slot] = make_float4(lpos.x, lpos.y, lpos.z, __uint_as_float(lcode));
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
        pp.lstart[slot] = make_float4(start.x, start.y, start.z, 0.f);
        rayA[slot] = make_float4(o.x, o.y, o.z, d.x);
        rayB[slot] = make_float4(d.y, d.z, __uint_as_float(0xffffffffu), __uint_as_float(slot));
        thr[slot] = make_float4(1.f, 1.f, 1.f, __uint_as_float(1u << 16));
        for (uint32_t q = 0; q < pp.reverse; q++) *lv_ptr(pp, q, 3, slot) = make_float4(0.f, 0.f, 0.f, 0.f); // no vertex yet
    }
}

// ---- camera rays only (the light sample was cached by k_raygen_light)
__global__ __launch_bounds__(256) void k_raygen_camera(const DevScene sc, const DevCamera cam, const PassParams pp, float4* __restrict__ rayA,
                                                        float4* __restrict__ rayB, float4* __restrict__ thr, float4* __restrict__ tot) {
    const uint32_t n = pp.npix * pp.ns;
    const SamplerTab tb = {pp.htab, pp.multisample};
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += gridDim.x * blockDim.x) {
        uint32_t srel, j; slot_decode(pp, slot, j, srel);
        const uint32_t pix = pp.pix_xy[pp.j0 + j], seed = pp.pix_seed[pp.j0 + j], s = pp.s0 + srel;
        const float2 jit = sample2d_t(tb, seed, s, 0);
        float2 lens = make_float2(0.f, 0.f);
        if (cam.lens_size != 0.0f) lens = sample2d_t(tb, seed, s, 1);
        f3 o, d;
        camera_ray(cam, (int)(pix & 0xffff), (int)(pix >> 16), (int)pp.xres, (int)pp.yres, jit, lens, o, d);
        rayA[slot] = make_float4(o.x, o.y, o.z, d.x);
        rayB[slot] = make_float4(d.y, d.z, __uint_as_float(0xffffffffu), __uint_as_float(slot));
        thr[slot] = make_float4(1.f, 1.f, 1.f, __uint_as_float(1u << 16));
        tot[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
        pp.vfin[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
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

// GENERIC = false / true: as in k_shade -- the first launch shades every vertex whose materials all take the fast BxDF
// route and lists the others (queue indices in pp.generic), the second walks that list with the full BxDF code.
// ---- light sub-path vertex k: store it, splat it to the camera, continue (russian = -1: no roulette)
template <bool GENERIC>
__global__ __launch_bounds__(RGK_SHADE_BLOCK, 4) void k_shade_light(const DevScene sc, const DevCamera cam, const PassParams pp, const uint32_t k,
                                                                     const float4* __restrict__ rayA, const float4* __restrict__ rayB,
                                                                     const float4* __restrict__ hit, float4* __restrict__ thr,
                                                                     float4* __restrict__ nextA, float4* __restrict__ nextB, float4* __restrict__ shA,
                                                                     float4* __restrict__ shB, float4* __restrict__ shC, uint32_t* __restrict__ counters) {
    const uint32_t count = counters[(GENERIC ? RGK_CNT_GENERIC : RGK_CNT_QUEUE) + k];
    const float eps = sc.epsilon;
    const SamplerTab tb = {pp.htab, pp.multisample};
    __shared__ uint32_t s_cnt[RGK_SHADE_BLOCK / 64];
    __shared__ uint32_t s_base;
    lut_lds_fill(sc);
    for (uint32_t base = blockIdx.x * RGK_SHADE_BLOCK; base < count; base += gridDim.x * RGK_SHADE_BLOCK) {
        const bool valid = base + threadIdx.x < count;
        const uint32_t i = !valid ? 0u : (GENERIC ? pp.generic[base + threadIdx.x] : base + threadIdx.x);
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

// ---- camera-path vertex with connections
template <bool GENERIC>
__global__ __launch_bounds__(RGK_SHADE_BLOCK, 4) void k_shade_bdpt(const DevScene sc, const DevCamera cam, const PassParams pp, const uint32_t bounce,
                                                                    const float4* __restrict__ rayA, const float4* __restrict__ rayB,
                                                                    const float4* __restrict__ hit, float4* __restrict__ thr, float4* __restrict__ tot,
                                                                    float4* __restrict__ nextA, float4* __restrict__ nextB, float4* __restrict__ shA,
                                                                    float4* __restrict__ shB, float4* __restrict__ shC, uint32_t* __restrict__ counters) {
    const uint32_t count = counters[(GENERIC ? RGK_CNT_GENERIC : RGK_CNT_QUEUE) + bounce];
    const float eps = sc.epsilon;
    const SamplerTab tb = {pp.htab, pp.multisample};
    __shared__ uint32_t s_cnt[RGK_SHADE_BLOCK / 64];
    __shared__ uint32_t s_base;
    lut_lds_fill(sc);
    for (uint32_t base = blockIdx.x * RGK_SHADE_BLOCK; base < count; base += gridDim.x * RGK_SHADE_BLOCK) {
        const bool valid = base + threadIdx.x < count;
        const uint32_t i = !valid ? 0u : (GENERIC ? pp.generic[base + threadIdx.x] : base + threadIdx.x);
        bool cont = false, have = false, defer = false;
        float4 nA = make_float4(0, 0, 0, 0), nB = nA;
        Vertex v;
        v.ok = false;
        uint32_t slot = 0;
        MatPrep mp;
        mp.fast = false;
        if (valid) {
            const float4 a = rayA[i], b = rayB[i], h = hit[i];
            slot = __float_as_uint(b.w);
            const f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
            const float4 st4 = thr[slot];
            f3 cum = mk3(st4.x, st4.y, st4.z);
            const uint32_t bits = __float_as_uint(st4.w);
            const uint32_t n = (bits & 0xffffu) + 1u;
            uint32_t c1 = bits >> 16;
            if (__float_as_int(h.w) < 0) {
                const f3 add = cum * skybox(sc, -d);
                float4 t = tot[slot];
                t.x = t.x + add.x; t.y = t.y + add.y; t.z = t.z + add.z;
                tot[slot] = t;
            } else {
                surface_point(sc, pp.bumpmap_scale, o, d, h, v);
                if (!GENERIC && v.ok) { // every material this vertex will evaluate: its own and those of the path's light vertices
                    defer = !mat_is_fast(v.mat.kind);
                    for (uint32_t q = 0; q < pp.reverse; q++)
                        if (lv_ptr(pp, q, 3, slot)->w != 0.0f && !mat_is_fast(mat_load(sc, __float_as_uint(lv_ptr(pp, q, 0, slot)->w)).kind)) defer = true;
                }
                if (v.ok && !defer) {
                    have = true;
                    uint32_t srel, j; slot_decode(pp, slot, j, srel);
                    const uint32_t seed = pp.pix_seed[pp.j0 + j], s = pp.s0 + srel;
                    const uint32_t base2d = (cam.lens_size != 0.0f) ? 2u : 1u;
                    mat_prepare(sc, v.mat, v.uv, v.VrL, n < pp.depth, mp);
                    f3 e_front = mk3(0.f, 0.f, 0.f);
                    if (dot3(v.faceN, v.Vr) > 0) e_front = mk3(v.mat.emission[0], v.mat.emission[1], v.mat.emission[2]);
                    pp.vfin[slot] = make_float4(cum.x, cum.y, cum.z, 1.0f); // contribution of this vertex
                    pp.vemit[slot] = make_float4(e_front.x, e_front.y, e_front.z, 0.f);
                    Step st;
                    path_step<GENERIC>(sc, tb, v, mp, seed, s, base2d + 3u + (n - 1u), n, pp.depth, pp.russian, cum, c1, st);
                    if (st.go) {
                        cont = true;
                        nA = make_float4(st.no.x, st.no.y, st.no.z, st.nd.x);
                        nB = make_float4(st.nd.y, st.nd.z, h.w, __uint_as_float(slot));
                        thr[slot] = make_float4(st.cum.x, st.cum.y, st.cum.z, __uint_as_float((n & 0xffffu) | (c1 << 16)));
                    }
                }
            }
        }
        const uint32_t pn = block_append(cont, &counters[RGK_CNT_QUEUE + bounce + 1], s_cnt, &s_base);
        if (cont) { nextA[pn] = nA; nextB[pn] = nB; }
        if (!GENERIC) {
            const uint32_t pd = block_append(defer, &counters[RGK_CNT_GENERIC + bounce], s_cnt, &s_base);
            if (defer) pp.generic[pd] = i;
        }
        // ---- q = 0: NEE to the path's light (:427-460); q = 1..reverse: light vertex q-1 (:463-480)
        for (uint32_t q = 0; q <= pp.reverse; q++) {
            bool shadow = false;
            float4 sA = make_float4(0, 0, 0, 0), sB = sA, sC = sA;
            if (have) {
                f3 from = mk3(0.f, 0.f, 0.f), rad = from;
                bool candidate = false;
                if (q == 0) {
                    const float4 li = pp.light[slot];
                    const DLight L = light_from_code(sc, mk3(li.x, li.y, li.z), __float_as_uint(li.w));
                    if (L.type >= 0) {
                        candidate = true;
                        from = L.pos;
                        const f3 dd = v.pos - L.pos;
                        const f3 Vi = norm3(L.pos - v.pos);
                        const f3 f = mat_value<GENERIC>(sc, (int)v.mat_id, v.mat, mp, qrot(v.g2l, Vi), v.VrL, v.uv);
                        const float G = fabsf(dot3(v.lightN, Vi)) / dot3(dd, dd);
                        const float kk = L.intensity * light_dir_factor(L, -Vi);
                        rad = (L.color * mk3(kk, kk, kk)) * (f * G);
                    }
                } else {
                    const float4 l3 = *lv_ptr(pp, q - 1, 3, slot);
                    if (l3.w != 0.0f) {
                        candidate = true;
                        const float4 l0 = *lv_ptr(pp, q - 1, 0, slot), l1 = *lv_ptr(pp, q - 1, 1, slot), l2 = *lv_ptr(pp, q - 1, 2, slot);
                        const f3 lpos = mk3(l0.x, l0.y, l0.z), lN = mk3(l1.x, l1.y, l1.z), lVr = mk3(l2.x, l2.y, l2.z);
                        const float2 luv = make_float2(l1.w, l2.w);
                        from = lpos;
                        const f3 light_to_p = norm3(v.pos - lpos);
                        const f3 p_to_light = -light_to_p;
                        const quatf lg2l = rotation_between(lN, mk3(0.f, 0.f, 1.f));
                        const f3 f_light = GENERIC ? bxdf_value_slow(sc, (int)__float_as_uint(l0.w), qrot(lg2l, light_to_p), qrot(lg2l, lVr), luv)
                                                   : bxdf_value_fastkind(sc, mat_load(sc, __float_as_uint(l0.w)), *lv_ptr(pp, q - 1, 4, slot), *lv_ptr(pp, q - 1, 5, slot),
                                                                         qrot(lg2l, light_to_p), qrot(lg2l, lVr));
                        const f3 f_point = mat_value_at<GENERIC>(sc, (int)v.mat_id, v.mat, mp, v.VrL, qrot(v.g2l, p_to_light), v.uv);
                        const f3 dd = v.pos - lpos;
                        const float G = fabsf(dot3(v.lightN, p_to_light)) / dot3(dd, dd);
                        rad = mk3(l3.x, l3.y, l3.z) * (f_light * f_point * G);
                    }
                }
                pp.term[(size_t)q * pp.batch + slot] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (candidate && (rad.x != 0.f || rad.y != 0.f || rad.z != 0.f)) {
                    const f3 diff = v.pos - from; // Ray(from, p.pos, 20 eps)
                    const f3 sd = norm3(diff);
                    shadow = true;
                    sA = make_float4(from.x, from.y, from.z, sd.x);
                    sB = make_float4(sd.y, sd.z, len3(diff) - eps * 20.0f, __uint_as_float(q * pp.batch + slot));
                    sC = make_float4(rad.x, rad.y, rad.z, 0.0f + eps * 20.0f);
                }
            }
            const uint32_t ps = block_append(shadow, &counters[RGK_CNT_SHADOW + bounce], s_cnt, &s_base);
            if (shadow) { shA[ps] = sA; shB[ps] = sB; shC[ps] = sC; }
        }
    }
}

// ---- total_here = NEE + connections (+ emission), clamp, path_total += total_here * contribution (:422-496)
__global__ __launch_bounds__(256) void k_finish_vertex(const PassParams pp, const uint32_t bounce, const float4* __restrict__ rayB,
                                                        float4* __restrict__ tot, const uint32_t* __restrict__ counters) {
    const uint32_t count = counters[RGK_CNT_QUEUE + bounce];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint32_t slot = __float_as_uint(rayB[i].w);
        const float4 fin = pp.vfin[slot];
        if (fin.w == 0.0f) continue;
        f3 total = mk3(0.f, 0.f, 0.f);
        for (uint32_t q = 0; q <= pp.reverse; q++) {
            const float4 t = pp.term[(size_t)q * pp.batch + slot];
            total = total + mk3(t.x, t.y, t.z);
        }
        const float4 e = pp.vemit[slot];
        if (e.x != 0.f || e.y != 0.f || e.z != 0.f) total = total + mk3(e.x, e.y, e.z);
        total = clamp3(total, pp.clamp);
        const f3 add = total * mk3(fin.x, fin.y, fin.z);
        float4 t = tot[slot];
        t.x = t.x + add.x; t.y = t.y + add.y; t.z = t.z + add.z;
        tot[slot] = t;
        pp.vfin[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
