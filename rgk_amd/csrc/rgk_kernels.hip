// Hand-written HIP kernels for gfx950 (MI355X): the wavefront path-tracing pipeline.
//
//   k_trace_camera  K1+K2  RenderPixel ray generation + FindIntersectKdOtherThan  reference src/path_tracer.cpp:53-61, src/camera.cpp:32-46 (rgk_trace.h)
//   k_trace_closest K2  FindIntersectKdOtherThan         reference src/scene_intersect.cpp:211-327 + src/primitives.cpp:75-166 (rgk_trace.h)
//   k_shade         K3/K4/K6/K7  GeneratePath body + NEE + compaction  reference src/path_tracer.cpp:134-300,427-460,485-496
//   k_trace_shadow[_first]  K5  Scene::Visibility + accumulate   reference src/scene.cpp:670-673, src/path_tracer.cpp:431,455-457 (rgk_trace.h)
//   k_resolve[_tiled]  K9  clamp / NaN scrub / AddPixel  reference src/path_tracer.cpp:502-507, src/tracer.cpp:18, src/texture.cpp:342-347
//   k_build_pixel_list, k_build_halton_table             Tracer::Render pixel order + seeds; halton_raw per round
//   k_entry_points, k_group_trange, k_entry_points_light  where a pixel group's camera rays / first shadow rays start in the tree
//                                                         (no reference counterpart: work shared by the rays of a group)
//
// Design: one path per slot, SoA-of-float4 queues so every lane moves 16 B
// per load; persistent waves pull 64 rays at a time from a device-side counter; a
// per-lane traversal stack lives in LDS ([depth][lane] -> conflict-free) and idle lanes are
// refilled with fresh rays (rgk_trace.h); surviving paths are compacted into the next
// bounce's queue with a wave ballot + prefix popcount and one atomic per workgroup.
// No MFMA: there is no dense contraction on this path.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "rgk_device.h"
#include "rgk_kernels.h"

#ifndef RGK_TRANGE_ATOMIC
#define RGK_TRANGE_ATOMIC 0
#endif
#include "rgk_trace.h" // K2 / K5: persistent traversal with lane refill

#ifndef RGK_SKIP_DEAD_NEE
#define RGK_SKIP_DEAD_NEE 1
#endif

// ------------------------------------------------------------------ K1: ray generation (camera_ray, camera_ray_of_slot: rgk_trace.h)
// (Unidirectional rounds have no ray-generation kernel: bounce 0 derives the camera ray from the slot number in the traversal
// kernel and again in the shading kernel, and the first vertex samples the path's light and starts its sum -- see
// k_trace_camera and k_shade<GENERIC, FIRST = true>.  The bidirectional kernels keep theirs, rgk_bdpt.h.)

// ------------------------------------------------------------------ K3+K4+K6+K7: shade

// FIRST = true: bounce 0.  Queue index == path slot, the ray is the slot's camera ray (not stored anywhere), the path state is
// implicit {1, 1, 1 | n = 0, 1-D counter = 1}, the path's ONE light is sampled here (TracePath: areal_sample, (lightdir_sample),
// GetRandomLight(Get2D, Get1D, areal_sample) -- path_tracer.cpp:315-322) and kept per slot for the later bounces of paths that
// go on, and the slot's radiance sum STARTS here (written, not read-modified: nothing zeroes it beforehand).
// BDPT = true: a camera-path vertex of a bidirectional round (reverse > 0).  The same vertex code; but where the slot's light
// sub-path has vertices to connect to (pp.lvmask) or the vertex emits, its total is clamp(NEE + connections + emission) -- the
// clamp spans all of them (path_tracer.cpp:422-496) -- so instead of its own shadow ray the vertex leaves a RECORD
// (pp.conn[c * batch + i], i = its queue index, listed in pp.connlist) for k_connect (rgk_bdpt.h), which evaluates the
// connections and queues the vertex for k_trace_shadow_jobs.  Most slots have no light vertex (a light far outside the
// geometry: its first ray must hit the scene at all), and those vertices are shaded exactly like a unidirectional one.
template <bool GENERIC, bool FIRST, bool BDPT = false>
__global__ __launch_bounds__((FIRST ? RGK_SHADE_BLOCK : RGK_SHADE_BLOCK_LATER), RGK_SHADE_WAVES) void k_shade(const DevScene sc, const DevCamera cam, const PassParams pp, const uint32_t bounce,
                                                            const float4* __restrict__ rayA, const float4* __restrict__ rayB,
                                                            const float4* __restrict__ hit, float4* __restrict__ thr, float4* __restrict__ tot,
                                                            float4* __restrict__ nextA, float4* __restrict__ nextB, float4* __restrict__ shA,
                                                            float4* __restrict__ shB, float4* __restrict__ shC, uint32_t* __restrict__ counters) {
    // GENERIC = false: every vertex whose material takes the fast route (diffuse, LTC); the others (mirror, dielectric,
    // transparent, mix) are only LISTED here and shaded by the GENERIC = true launch that follows, which walks that
    // list with the full BxDF code.  Keeping the rare, big generic route out of this kernel takes its scratch from
    // 136 to 24 bytes per thread -- and the kernel is bound by the bytes it moves.
    const uint32_t count = counters[(GENERIC ? RGK_CNT_GENERIC : RGK_CNT_QUEUE) + bounce];
    const int lane = threadIdx.x & 63;
    const float eps = sc.epsilon;
    const SamplerTab tb = {pp.htab, pp.multisample};
    // workgroup size: 512 for the first vertices, 256 for the later ones -- their lanes diverge (0.59 of them busy), and with four
    // waves instead of eight behind each of the compaction's two barriers the launch is 5.6 % shorter (21.35 -> 20.15 ms; the first
    // vertices, whose waves run alike, lose 1.7 % that way and keep 512)
    constexpr uint32_t BLK = FIRST ? RGK_SHADE_BLOCK : RGK_SHADE_BLOCK_LATER;
    __shared__ uint32_t s_cnt[4][BLK / 64];
    __shared__ uint32_t s_base[4];
    lut_lds_fill(sc);
    // workgroup-uniform trip count (the compaction below synchronises the workgroup)
    for (uint32_t base = blockIdx.x * BLK; base < count; base += gridDim.x * BLK) {
        const bool valid = base + threadIdx.x < count;
        const uint32_t i = !valid ? 0u : (GENERIC ? pp.generic[base + threadIdx.x] : base + threadIdx.x);
        bool cont = false, shadow = false, defer = false, conn = false;
        float4 nA = make_float4(0, 0, 0, 0), nB = nA, sA = nA, sB = nA, sC = nA;
        if (valid) {
            const float4 h = hit[i];
            uint32_t slot;
            f3 o, d;
            if (FIRST) { slot = i; camera_ray_of_slot(cam, pp, slot, o, d); }
            else {
                const float4 a = rayA[i], b = rayB[i];
                slot = __float_as_uint(b.w);
                o = mk3(a.x, a.y, a.z); d = mk3(a.w, b.x, b.y);
            }
            const float4 st = FIRST ? make_float4(1.f, 1.f, 1.f, __uint_as_float(1u << 16)) : thr[slot];
            const float4 li_slot = FIRST ? make_float4(0.f, 0.f, 0.f, 0.f) : pp.light[slot]; // (asked for with the path state, not where it is first used: later vertices 21.75 -> 21.3 ms)
            f3 tot0 = mk3(0.f, 0.f, 0.f); // FIRST, fast launch: what this vertex adds to the (so far empty) sum of its slot
            float4 li_keep = make_float4(0.f, 0.f, 0.f, 0.f);
            f3 cum = mk3(st.x, st.y, st.z);
            uint32_t bits = __float_as_uint(st.w);
            uint32_t n = (bits & 0xffffu) + 1u; // n++ at loop top, reference path_tracer.cpp:123
            uint32_t c1 = bits >> 16;
            const int tri = __float_as_int(h.w);
            uint32_t srel, j; slot_decode(pp, slot, j, srel);
            const uint32_t seed = pp.pix_seed[pp.j0 + j];
            const uint32_t s = pp.s0 + srel;
            const uint32_t base2d = (cam.lens_size != 0.0f) ? 2u : 1u;
            const f3 Vr = -d;
            if (tri < 0) {
                // sky vertex: path_total += contribution * sky, reference path_tracer.cpp:137-146,409-415
                f3 sky = skybox(sc, Vr);
                f3 add = cum * sky;
                if (FIRST && !GENERIC) tot0 = mk3(0.f + add.x, 0.f + add.y, 0.f + add.z);
                else {
                    float4 t = tot[slot];
                    t.x = t.x + add.x; t.y = t.y + add.y; t.z = t.z + add.z;
                    tot[slot] = t;
                }
            } else {
                // one 128-byte line per triangle: {nA,uvA.x}{nB,uvA.y}{nC,uvB.x}{tA,uvB.y}{tB,uvC.x}{tC,uvC.y}{mat}
                const uint32_t tsr = (uint32_t)tri * (uint32_t)sizeof(TriShade);
                const float4 g0 = gld_f4(sc.tri_shade, tsr), g1 = gld_f4(sc.tri_shade, tsr + 16u), g2 = gld_f4(sc.tri_shade, tsr + 32u);
                const uint32_t mat_id = gld_u32(sc.tri_shade, tsr + 96u);
                const DevMaterial mat = mat_load(sc, mat_id);
                defer = !GENERIC && !mat_is_fast(mat.kind);
                if (!defer) {
                const float al = h.y, be = h.z;
                const float ia = 1.0f - al - be, ib = al, ic = be; // Intersection::a,b,c scene_intersect.cpp:280-283
                f3 pos = o + h.x * d;
                f3 nA_ = mk3(g0.x, g0.y, g0.z), nB_ = mk3(g1.x, g1.y, g1.z), nC_ = mk3(g2.x, g2.y, g2.z);
                f3 faceN = ia * nA_ + ib * nB_ + ic * nC_;
                bool ok = true;
                if (faceN.x != faceN.x) { // NaN fallbacks, path_tracer.cpp:157-171
                    faceN = nA_;
                    if (faceN.x != faceN.x) { faceN = nB_; if (faceN.x != faceN.x) { faceN = nC_; if (faceN.x != faceN.x) ok = false; } }
                }
                if (ok && len3(faceN) <= 0.0f) ok = false; // path_tracer.cpp:175
                if (ok) {
                    faceN = norm3(faceN);
                    // (asking for the whole 128-byte record at once, before the normal has been checked: measured, nothing -- the line is
                    // in L1 by now)
                    const float4 g3 = gld_f4(sc.tri_shade, tsr + 48u), g4 = gld_f4(sc.tri_shade, tsr + 64u), g5 = gld_f4(sc.tri_shade, tsr + 80u);
                    float2 uv = make_float2(0.f, 0.f);
                    if (sc.has_texcoords) {
                        uv.x = ia * g0.w + ib * g2.w + ic * g4.w;
                        uv.y = ia * g1.w + ib * g3.w + ic * g5.w;
                    }
                    const bool bumped = tex_kind(mat.t_bump) != RGK_TEXREF_NONE;
                    f3 lightN = faceN;
                    if (bumped) { // bump, path_tracer.cpp:204-231
                        float right, bottom;
                        tex_slopes(sc, mat.t_bump, uv, right, bottom);
                        f3 tangent = ia * mk3(g3.x, g3.y, g3.z) + ib * mk3(g4.x, g4.y, g4.z) + ic * mk3(g5.x, g5.y, g5.z);
                        if (!(tangent.x * tangent.x + tangent.y * tangent.y + tangent.z * tangent.z < 0.001f)) {
                            tangent = norm3(tangent);
                            f3 bitangent = norm3(cross3(faceN, tangent));
                            f3 tangent2 = cross3(bitangent, faceN);
                            lightN = norm3(faceN + (tangent2 * right + bitangent * bottom) * pp.bumpmap_scale);
                            if (lightN.x != lightN.x) lightN = faceN;
                        }
                    }
                    // SystemTransform(lightN, +Z), reference src/glm.hpp:21-24
                    const quatf g2l = rotation_between(lightN, mk3(0.f, 0.f, 1.f));
                    const f3 VrL = qrot(g2l, Vr);
                    // The vertex at n == depth is the path's last (while(n < depth__), :122): its sampled direction,
                    // transfer coefficient and roulette draw can never be observed, so they are not computed.
                    const bool last = !(n < pp.depth);
                    const bool no_russian = (mat.flags & RGK_MAT_NO_RUSSIAN) != 0;
                    // `ends`: nothing follows this vertex (it is the last by depth).  Then the sampled direction and its weight can never
                    // be observed and are not computed, and the material is needed for ONE thing, the BxDF value towards the path's
                    // light -- which for the kinds shaded here is exactly 0 when the light or the viewer is below the (bumped)
                    // surface's horizon (bxdf_value_fastkind).  Then neither texels nor tables are fetched: the vertex adds its
                    // emission, if any, and nothing else -- the same zero the long way round gives (0 * G needs G finite: the
                    // vertex is not AT the light).  (Drawing the roulette number first, so that a path it ends counts as ending
                    // here too, is exact as well -- the draw depends on nothing computed here -- but measured slower, 57.9 vs
                    // 55.9 ms of shading per round: the lanes that could stop early wait for their wave anyway.)
                    const bool ends = last;
                    float4 li; // the path's light {pos, code}: sampled by the first vertex, carried in pp.light after it
                    if (FIRST) {
                        f3 lpos;
                        const uint32_t lcode = light_code(sc, sample2d_t(tb, seed, s, base2d + 2u), sample1d_t(tb, seed, s, 0u), sample2d_t(tb, seed, s, base2d), lpos);
                        li = make_float4(lpos.x, lpos.y, lpos.z, __uint_as_float(lcode));
                    } else li = li_slot;
                    bool nee_dead = false;
                    if (RGK_SKIP_DEAD_NEE && ends && !GENERIC) {
                        const DLight L0 = light_from_code(sc, mk3(li.x, li.y, li.z), __float_as_uint(li.w));
                        if (L0.type < 0) nee_dead = true;
                        else {
                            const f3 df = pos - L0.pos;
                            nee_dead = dot3(df, df) > 0.f && (qrot(g2l, norm3(L0.pos - pos)).z <= 0.f || VrL.z <= 0.f);
                        }
                    }
                    MatPrep mp;
                    if (!nee_dead) mat_prepare(sc, mat, uv, VrL, !ends, mp);
                    else { mp.fast = true; mp.lobe = false; mp.diffc = mp.colorc = mk3(0.f, 0.f, 0.f); }
                    const f3 contribution = cum; // excludes this vertex's own coefficients, :135
                    bool inside = false;
                    f3 dir = mk3(0.f, 0.f, 0.f);
                    uint32_t n_eff = n;
                    if (!ends) {
                        // BxDF sample, path_tracer.cpp:243-250
                        const quatf l2g = qinverse(g2l);
                        float2 u = sample2d_t(tb, seed, s, base2d + 3u + (n - 1u));
                        f3 dirL, weight; bool may_leak;
                        mat_sample<GENERIC>(sc, (int)mat_id, mat, mp, VrL, uv, u, dirL, weight, may_leak);
                        inside = dirL.z < 0;
                        dir = qrot(l2g, dirL);
                        if (!(dot3(dir, faceN) * dot3(Vr, faceN) > 0) && !may_leak) n_eff += 10000u; // leak, :252-260
                        const float rc = (!no_russian && pp.russian > 0.0f && n_eff > 1u) ? 1.0f / pp.russian : 1.0f;
                        cum = cum * rc;
                        cum = cum * weight;
                    }

                    // ---- phase 3 for this vertex: NEE to the path's light, :427-460,485-496
                    {
                        li_keep = li;
                        const DLight L = light_from_code(sc, mk3(li.x, li.y, li.z), __float_as_uint(li.w));
                        f3 e_front = mk3(0.f, 0.f, 0.f);
                        if (dot3(faceN, Vr) > 0) e_front = mk3(mat.emission[0], mat.emission[1], mat.emission[2]);
                        const bool has_e = (e_front.x != 0.f) || (e_front.y != 0.f) || (e_front.z != 0.f);
                        // bidirectional: light vertices to connect to (bits 0..6), or an emitting vertex -> the record route
                        if (BDPT) conn = ((pp.lvmask[slot] & 0x7fu) != 0u) || has_e;
                        f3 B = mk3(0.f, 0.f, 0.f);
                        if (has_e && !conn) {
                            B = clamp3(mk3(0.f, 0.f, 0.f) + e_front, pp.clamp) * contribution;
                            if (FIRST && !GENERIC) tot0 = mk3(0.f + B.x, 0.f + B.y, 0.f + B.z);
                            else {
                                float4 t = tot[slot];
                                t.x = t.x + B.x; t.y = t.y + B.y; t.z = t.z + B.z;
                                tot[slot] = t;
                            }
                        }
                        f3 out = mk3(0.f, 0.f, 0.f); // NEE radiance before visibility, clamp and contribution
                        if (L.type >= 0 && !nee_dead) {
                            const f3 diff = pos - L.pos; // Ray(light.pos, p.pos, 20 eps), src/ray.hpp:15-22
                            const f3 sd = norm3(diff);
                            const float slen = len3(diff);
                            const f3 Vi = norm3(L.pos - pos);
                            const f3 f = mat_value<GENERIC>(sc, (int)mat_id, mat, mp, qrot(g2l, Vi), VrL, uv);
                            const float G = fabsf(dot3(lightN, Vi)) / dot3(diff, diff);
                            const float k = L.intensity * light_dir_factor(L, -Vi);
                            const f3 inc = L.color * mk3(k, k, k);
                            out = inc * (f * G);
                            if (!conn) {
                                f3 A = clamp3((mk3(0.f, 0.f, 0.f) + out) + e_front, pp.clamp) * contribution;
                                f3 rad = has_e ? A - B : A;
                                if (rad.x != 0.f || rad.y != 0.f || rad.z != 0.f) {
                                    shadow = true;
                                    sA = make_float4(L.pos.x, L.pos.y, L.pos.z, sd.x);
                                    sB = make_float4(sd.y, sd.z, slen - eps * 20.0f, __uint_as_float(slot));
                                    sC = make_float4(rad.x, rad.y, rad.z, 0.0f + eps * 20.0f);
                                }
                            }
                        }
                        if (BDPT && conn) { // everything k_connect needs to evaluate this vertex's material towards a light vertex
                            const size_t bs = pp.batch;
                            pp.conn[i] = make_float4(pos.x, pos.y, pos.z, __uint_as_float(slot));
                            pp.conn[bs + i] = make_float4(lightN.x, lightN.y, lightN.z, __uint_as_float(mat_id));
                            pp.conn[2 * bs + i] = make_float4(g2l.x, g2l.y, g2l.z, g2l.w);
                            pp.conn[3 * bs + i] = make_float4(VrL.x, VrL.y, VrL.z, uv.x);
                            pp.conn[4 * bs + i] = make_float4(contribution.x, contribution.y, contribution.z, uv.y);
                            pp.conn[5 * bs + i] = make_float4(out.x, out.y, out.z, __uint_as_float(has_e ? 1u : 0u));
                        }
                    }
                    // ---- continuation, path_tracer.cpp:275-300
                    bool go = !ends && !(max3c(cum) < 0.001f);
                    if (go && !no_russian && pp.russian >= 0.0f) {
                        float r = sample1d_t(tb, seed, s, c1);
                        c1++;
                        if (r > pp.russian) go = false;
                    }
                    if (go && n_eff > pp.depth) go = false;
                    if (go && !(n_eff < pp.depth)) go = false; // while(n < depth__)
                    if (go) {
                        const float sgn = inside ? -1.0f : 1.0f;
                        f3 no = pos + faceN * eps * 10.0f * sgn;
                        f3 nd = norm3(norm3(dir)); // normalised by the caller and again by Ray's ctor
                        cont = true;
                        nA = make_float4(no.x, no.y, no.z, nd.x);
                        nB = make_float4(nd.y, nd.z, __int_as_float(tri), __uint_as_float(slot));
                        thr[slot] = make_float4(cum.x, cum.y, cum.z, __uint_as_float((n & 0xffffu) | (c1 << 16)));
                    }
                    // the path's light: later vertices read it (and k_connect, for the NEE ray of a vertex on the record route)
                    if (FIRST && (go || conn)) pp.light[slot] = li_keep;
                }
                } // !defer
            }
            if (FIRST && !GENERIC) tot[slot] = make_float4(tot0.x, tot0.y, tot0.z, 0.f); // every slot of the pass passes here exactly once
        }
        // ---- K7: compaction into the next queues.
        {
            // Wave ballot + prefix popcount inside the wave, the waves of the workgroup add up through LDS, and ONE
            // atomic per workgroup and queue reserves the range.
            const unsigned long long m = __ballot(cont), ms = __ballot(shadow), md = __ballot(defer), mc = BDPT ? __ballot(conn) : 0ull;
            const int w = threadIdx.x >> 6;
            if (lane == 0) { s_cnt[0][w] = __popcll(m); s_cnt[1][w] = __popcll(ms); s_cnt[2][w] = __popcll(md); if (BDPT) s_cnt[3][w] = __popcll(mc); }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
                for (int k = 0; k < (int)(BLK / 64); k++) {
                    uint32_t a = s_cnt[0][k], b = s_cnt[1][k], c = s_cnt[2][k];
                    s_cnt[0][k] = t0; s_cnt[1][k] = t1; s_cnt[2][k] = t2;
                    t0 += a; t1 += b; t2 += c;
                    if (BDPT) { uint32_t e = s_cnt[3][k]; s_cnt[3][k] = t3; t3 += e; }
                }
                s_base[0] = t0 ? atomicAdd(&counters[RGK_CNT_QUEUE + bounce + 1], t0) : 0u;
                s_base[1] = t1 ? atomicAdd(&counters[RGK_CNT_SHADOW + bounce], t1) : 0u;
                s_base[2] = t2 ? atomicAdd(&counters[RGK_CNT_GENERIC + bounce], t2) : 0u;
                if (BDPT) s_base[3] = t3 ? atomicAdd(&counters[RGK_CNT_CONN + bounce], t3) : 0u;
            }
            __syncthreads();
            if (!GENERIC && defer) pp.generic[s_base[2] + s_cnt[2][w] + __popcll(md & ((1ull << lane) - 1ull))] = i;
            if (BDPT && conn) pp.connlist[s_base[3] + s_cnt[3][w] + __popcll(mc & ((1ull << lane) - 1ull))] = i;
            if (cont) {
                uint32_t p = s_base[0] + s_cnt[0][w] + __popcll(m & ((1ull << lane) - 1ull));
                nextA[p] = nA; nextB[p] = nB;
            }
            if (shadow) {
                uint32_t p = s_base[1] + s_cnt[1][w] + __popcll(ms & ((1ull << lane) - 1ull));
                shA[p] = sA; shB[p] = sB; shC[p] = sC;
            }
            __syncthreads(); // s_cnt / s_base are rewritten by the next iteration
        }
    }
}

#include "rgk_bdpt.h" // K8: light sub-path, splats, connections (reverse > 0)

// ------------------------------------------------------------------ K9: resolve
// final clamp + NaN/negative scrub (path_tracer.cpp:502-507), per-pixel sum over the pass's
// samples in sample order (RenderPixel :64), then AddPixel when the pixel's last sample is in.
__global__ __launch_bounds__(256) void k_resolve(const PassParams pp, const float4* __restrict__ tot, float4* __restrict__ pixsum,
                                                  float* __restrict__ accum_rgb, uint32_t* __restrict__ accum_count) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < pp.npix; j += gridDim.x * blockDim.x) {
        float4 acc = (pp.s0 == 0) ? make_float4(0.f, 0.f, 0.f, 0.f) : pixsum[pp.j0 + j];
        for (uint32_t srel = 0; srel < pp.ns; srel++) {
            float4 t = tot[slot_of(pp, j, srel)];
            f3 v = clamp3(mk3(t.x, t.y, t.z), pp.clamp);
            if (v.x != v.x || v.x < 0.0f) v.x = 0.0f;
            if (v.y != v.y || v.y < 0.0f) v.y = 0.0f;
            if (v.z != v.z || v.z < 0.0f) v.z = 0.0f;
            acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z;
        }
        if (pp.s0 + pp.ns >= pp.multisample) {
            uint32_t pix = pp.pix_xy[pp.j0 + j];
            size_t p = (size_t)(pix >> 16) * pp.xres + (pix & 0xffff);
            accum_rgb[3 * p + 0] += acc.x;
            accum_rgb[3 * p + 1] += acc.y;
            accum_rgb[3 * p + 2] += acc.z;
            accum_count[p] += pp.multisample;
        } else {
            pixsum[pp.j0 + j] = acc;
        }
    }
}

// The same for gshift > 0, where the 2^g samples of a pixel are contiguous (a thread walking its pixel's samples would stride
// 2^g * 16 bytes against its neighbours): one wave per PT pixels stages each sample block through LDS with coalesced loads, then
// lane p sums pixel p's samples in sample order -- the same additions in the same order as above.
__global__ __launch_bounds__(64) void k_resolve_tiled(const PassParams pp, const float4* __restrict__ tot, float4* __restrict__ pixsum,
                                                       float* __restrict__ accum_rgb, uint32_t* __restrict__ accum_count, const uint32_t PT) {
    extern __shared__ float4 tile[]; // [PT pixels][G + 1], PT <= 64
    const uint32_t G = 1u << pp.gshift, lane = threadIdx.x;
    for (uint32_t jb = blockIdx.x * PT; jb < pp.npix; jb += gridDim.x * PT) {
        const uint32_t np = min(PT, pp.npix - jb), j = jb + lane;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < np && pp.s0 != 0) acc = pixsum[pp.j0 + j];
        for (uint32_t sb = 0; sb < (pp.ns >> pp.gshift); sb++) {
            const float4* __restrict__ src = tot + ((size_t)(sb * pp.npix + jb) << pp.gshift); // np * G consecutive slots
            for (uint32_t k = 0; k * 64u < np * G; k++) {
                const uint32_t idx = k * 64u + lane;
                if (idx < np * G) tile[(idx >> pp.gshift) * (G + 1u) + (idx & (G - 1u))] = src[idx];
            }
            __syncthreads();
            if (lane < np)
                for (uint32_t g = 0; g < G; g++) {
                    const float4 t = tile[lane * (G + 1u) + g];
                    f3 v = clamp3(mk3(t.x, t.y, t.z), pp.clamp);
                    if (v.x != v.x || v.x < 0.0f) v.x = 0.0f;
                    if (v.y != v.y || v.y < 0.0f) v.y = 0.0f;
                    if (v.z != v.z || v.z < 0.0f) v.z = 0.0f;
                    acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z;
                }
            __syncthreads();
        }
        if (lane < np) {
            if (pp.s0 + pp.ns >= pp.multisample) {
                uint32_t pix = pp.pix_xy[pp.j0 + j];
                size_t p = (size_t)(pix >> 16) * pp.xres + (pix & 0xffff);
                accum_rgb[3 * p + 0] += acc.x;
                accum_rgb[3 * p + 1] += acc.y;
                accum_rgb[3 * p + 2] += acc.z;
                accum_count[p] += pp.multisample;
            } else {
                pixsum[pp.j0 + j] = acc;
            }
        }
    }
}

// The round's pixel list in Tracer::Render order with the per-pixel seeds (a1, a2), built on the device from the tile list
// (2040 tiles at 1080p: a few KB up instead of 16 MB of host-built lists every round).  The seed of a pixel is fixed by its
// row-major rank k inside the task (RenderPixel is called in that order, tracer.cpp:8-9, and bumps the seed first,
// path_tracer.cpp:47).  The ORDER in which pixels occupy path slots is free: 8x8 blocks, so the 64 lanes of a wave start as
// a compact bundle of camera rays instead of two 32-pixel row segments (coherent traversal and shading).
__global__ __launch_bounds__(256) void k_build_pixel_list(const rgk_tile* __restrict__ tiles, const uint32_t* __restrict__ tile_off, uint32_t n_tiles,
                                                           uint32_t* __restrict__ pix_xy, uint32_t* __restrict__ pix_seed) {
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const rgk_tile tl = tiles[t];
        const uint32_t tw = tl.x1 - tl.x0, th = tl.y1 - tl.y0, base = tile_off[t];
        for (uint32_t q = threadIdx.x; q < tw * th; q += blockDim.x) {
            // slot q of the tile in 8x8-block order -> (x, y) inside the tile
            const uint32_t nbr = (th + 7u) >> 3, nbc = (tw + 7u) >> 3;
            uint32_t r = q / (8u * tw);
            if (r > nbr - 1u) r = nbr - 1u;
            const uint32_t rem = q - r * 8u * tw;
            const uint32_t bh = (th - 8u * r) < 8u ? (th - 8u * r) : 8u;
            uint32_t c = rem / (bh * 8u);
            if (c > nbc - 1u) c = nbc - 1u;
            const uint32_t rem2 = rem - c * bh * 8u;
            const uint32_t bw = (tw - 8u * c) < 8u ? (tw - 8u * c) : 8u;
            const uint32_t y = 8u * r + rem2 / bw, x = 8u * c + rem2 % bw; // (Z order inside the block: no difference measured)
            const uint32_t k = y * tw + x;
            pix_xy[base + q] = (tl.x0 + x) | ((tl.y0 + y) << 16);
            pix_seed[base + q] = tl.seed + (k + 1u) * 0x42424242u;
        }
    }
}

// halton_raw tabulated once per round: htab[hdim * S + s]
__global__ void k_build_halton_table(const DevScene sc, uint32_t S, float* __restrict__ htab) {
    const uint32_t n = 192u * S;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t hdim = i / S, s = i - hdim * S;
        htab[i] = halton_raw(sc, hdim, s);
    }
}

__global__ void k_init_counters(uint32_t* counters, uint32_t n0) {
    for (uint32_t i = threadIdx.x; i < RGK_CNT_TOTAL; i += blockDim.x) counters[i] = (i == RGK_CNT_QUEUE) ? n0 : 0u;
}

// rays given as 8 floats {o, d, near, far} (rgk_trace_closest API) -> queue layout
__global__ void k_pack_rays(uint32_t n, const float* __restrict__ rays, const int32_t* __restrict__ ignore, float4* rayA, float4* rayB, float2* nearfar) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* r = rays + 8 * (size_t)i;
        rayA[i] = make_float4(r[0], r[1], r[2], r[3]);
        rayB[i] = make_float4(r[4], r[5], __int_as_float(ignore ? ignore[i] : -1), __uint_as_float(i));
        nearfar[i] = make_float2(r[6], r[7]);
    }
}
// Ray(a, b, 20 eps) for Scene::Visibility(a, b)
__global__ void k_pack_visibility(const DevScene sc, uint32_t n, const float* __restrict__ a, const float* __restrict__ b, float4* shA, float4* shB, float4* shC) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        f3 pa = mk3(a[3 * i], a[3 * i + 1], a[3 * i + 2]), pb = mk3(b[3 * i], b[3 * i + 1], b[3 * i + 2]);
        f3 diff = pb - pa;
        f3 d = norm3(diff);
        float e = sc.epsilon * 20.0f;
        shA[i] = make_float4(pa.x, pa.y, pa.z, d.x);
        shB[i] = make_float4(d.y, d.z, len3(diff) - e, __uint_as_float(i));
        shC[i] = make_float4(0.f, 0.f, 0.f, 0.0f + e);
    }
}
__global__ void k_unpack_hits(uint32_t n, const float4* __restrict__ hit, rgk_hit* out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 h = hit[i];
        rgk_hit r;
        r.t = h.x; r.tri = __float_as_int(h.w);
        r.a = 1.0f - h.y - h.z; r.b = h.y; r.c = h.z;
        if (r.tri < 0) { r.a = 0.f; r.b = 0.f; r.c = 0.f; }
        out[i] = r;
    }
}
__global__ void k_sampler_eval(const DevScene sc, uint32_t n, const uint32_t* seed, const uint32_t* index, const uint32_t* dim, int is2d, float* out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (is2d) { float2 v = sample2d(sc, seed[i], index[i], dim[i]); out[2 * i] = v.x; out[2 * i + 1] = v.y; }
        else { out[2 * i] = sample1d(sc, seed[i], index[i], dim[i]); out[2 * i + 1] = 0.f; }
    }
}

// ------------------------------------------------------------------ unit-level seams: BxDF::value / sample, texture lookups
// route 0: the generic code (bxdf_value / bxdf_sample: every material kind, mixes included);
// route 1: what k_shade runs for diffuse / LTC materials (mat_prepare + mat_value / mat_sample with the shared fetches), generic otherwise.
__global__ __launch_bounds__(256) void k_bxdf_value(const DevScene sc, uint32_t n, uint32_t route, const uint32_t* __restrict__ mat, const float* __restrict__ Vi,
                                                    const float* __restrict__ Vr, const float* __restrict__ uv, float* __restrict__ out) {
    lut_lds_fill(sc);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const f3 vi = mk3(Vi[3 * i], Vi[3 * i + 1], Vi[3 * i + 2]), vr = mk3(Vr[3 * i], Vr[3 * i + 1], Vr[3 * i + 2]);
        const float2 t = make_float2(uv[2 * i], uv[2 * i + 1]);
        f3 v;
        const DevMaterial m = mat_load(sc, mat[i]);
        if (route == 1u && mat_is_fast(m.kind)) {
            MatPrep mp;
            mat_prepare(sc, m, t, vr, false, mp);
            v = mat_value<false>(sc, (int)mat[i], m, mp, vi, vr, t);
        } else v = bxdf_value(sc, (int)mat[i], vi, vr, t);
        out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
    }
}
__global__ __launch_bounds__(256) void k_bxdf_sample(const DevScene sc, uint32_t n, uint32_t route, const uint32_t* __restrict__ mat, const float* __restrict__ Vi,
                                                     const float* __restrict__ uv, const float* __restrict__ u, float* __restrict__ out_dir,
                                                     float* __restrict__ out_w, uint8_t* __restrict__ leak) {
    lut_lds_fill(sc);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const f3 vi = mk3(Vi[3 * i], Vi[3 * i + 1], Vi[3 * i + 2]);
        const float2 t = make_float2(uv[2 * i], uv[2 * i + 1]), uu = make_float2(u[2 * i], u[2 * i + 1]);
        f3 d, w;
        bool ml;
        const DevMaterial m = mat_load(sc, mat[i]);
        if (route == 1u && mat_is_fast(m.kind)) {
            MatPrep mp;
            mat_prepare(sc, m, t, vi, true, mp);
            mat_sample<false>(sc, (int)mat[i], m, mp, vi, t, uu, d, w, ml);
        } else bxdf_sample(sc, (int)mat[i], vi, t, uu, d, w, ml);
        out_dir[3 * i] = d.x; out_dir[3 * i + 1] = d.y; out_dir[3 * i + 2] = d.z;
        out_w[3 * i] = w.x; out_w[3 * i + 1] = w.y; out_w[3 * i + 2] = w.z;
        leak[i] = ml ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void k_texture_sample(const DevScene sc, uint32_t n, const TexRef* __restrict__ refs, const int32_t* __restrict__ tex,
                                                        const float* __restrict__ uv, float* __restrict__ rgb, float* __restrict__ sr, float* __restrict__ sb) {
    lut_lds_fill(sc);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        TexRef t;
        t.kind = RGK_TEXREF_NONE; t.a = t.b = t.c = 0;
        if (tex[i] >= 0) t = refs[tex[i]];
        const float2 p = make_float2(uv[2 * i], uv[2 * i + 1]);
        const f3 c = tex_get(sc, t, p);
        float r, b;
        tex_slopes(sc, t, p, r, b);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
        sr[i] = r; sb[i] = b;
    }
}
// the pinned transcendental functions (include/rgk_libm.h) evaluated on the device: fn 0 sin, 1 cos, 2 acos, 3 asin, 4 atan2(a, b)
__global__ void k_libm_eval(int fn, uint32_t n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float r = 0.f;
        if (fn == 0) r = rgk_sinf(a[i]);
        else if (fn == 1) r = rgk_cosf(a[i]);
        else if (fn == 2) r = rgk_acosf(a[i]);
        else if (fn == 3) r = rgk_asinf(a[i]);
        else r = rgk_atan2f(a[i], b[i]);
        out[i] = r;
    }
}
void rgk_launch_libm_eval(hipStream_t st, int fn, uint32_t n, const float* a, const float* b, float* out) { k_libm_eval<<<(n + 255) / 256, 256, 0, st>>>(fn, n, a, b, out); }
void rgk_launch_bxdf_value(hipStream_t st, const DevScene& sc, uint32_t n, uint32_t route, const uint32_t* mat, const float* Vi, const float* Vr, const float* uv, float* out) {
    k_bxdf_value<<<(n + 255) / 256, 256, RGK_LDS_SHADE_BYTES, st>>>(sc, n, route, mat, Vi, Vr, uv, out);
}
void rgk_launch_bxdf_sample(hipStream_t st, const DevScene& sc, uint32_t n, uint32_t route, const uint32_t* mat, const float* Vi, const float* uv, const float* u,
                            float* out_dir, float* out_w, uint8_t* leak) {
    k_bxdf_sample<<<(n + 255) / 256, 256, RGK_LDS_SHADE_BYTES, st>>>(sc, n, route, mat, Vi, uv, u, out_dir, out_w, leak);
}
void rgk_launch_texture_sample(hipStream_t st, const DevScene& sc, uint32_t n, const TexRef* refs, const int32_t* tex, const float* uv, float* rgb, float* sr, float* sb) {
    k_texture_sample<<<(n + 255) / 256, 256, RGK_LDS_SHADE_BYTES, st>>>(sc, n, refs, tex, uv, rgb, sr, sb);
}

// ------------------------------------------------------------------ launch wrappers (host)
// Upper bound on the length of the queues the next launches will consume (the host reads a queue counter back every
// few bounces of a deep path loop): a 40-bounce round ends in dozens of launches over a few hundred rays, and a
// full persistent grid of 3000 waves then costs more in work-fetch atomics and LDS fills than the rays themselves.
static thread_local uint32_t g_bound = 0xffffffffu, g_bound_shadow = 0xffffffffu;
void rgk_launch_set_bound(uint32_t items, uint32_t shadow_items) { g_bound = items; g_bound_shadow = shadow_items; }
static inline int bounded_grid(int full, uint32_t items, uint32_t per_block) {
    const uint64_t need = ((uint64_t)items + per_block - 1) / per_block;
    return (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)full, need));
}

int rgk_trace_grid(int lds_entries) {
    // LDS-limited residency: entries*256*4 B per block out of 160 KiB, 256 CUs
    int per_cu = (160 * 1024) / (lds_entries * RGK_TRACE_BLOCK * 4);
    static const int cap = [] { const char* e = std::getenv("RGK_TRACE_PER_CU"); return e ? std::atoi(e) : 8; }(); // experiments: leave room for a second stream
    if (per_cu > cap) per_cu = cap;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    return 256 * per_cu;
}
// (stack need, LDS entries) variants.  Default 256/16: 16 entries per lane in LDS, the rest -- reached only by the deep
// part of a walk -- per lane in global memory.  That keeps 8 workgroups per CU resident whatever the tree depth
// (occupancy was LDS-bound: 5 per CU with 32 entries, 3 with 48), and more waves are what the L1-latency-bound half
// of the kernel wanted: Sponza trace launch 40.3 -> 33.8 ms, shadow 25 -> 20 ms per round (32 / 24 / 16 / 12 / 8
// entries: 40.3 / 37.0 / 35.1 / 35.5 / 35.2 ms at 7 waves per SIMD; 16 entries at 8 waves: 34.2).
// RGK_STACK_LDS=32 selects 256/32, RGK_STACK_OVF=0 the all-LDS 32/32 (shallow trees only).
#define RGK_TRACE_DISPATCH(K, BOUND, ...)                                                            \
    {                                                                                                \
        const int grid = bounded_grid(rgk_trace_grid(tc.lds), BOUND, RGK_TRACE_BLOCK);               \
        if (tc.stack <= 32 && tc.lds == 32) { if (count_stats) K<true, 32, 32><<<grid, RGK_TRACE_BLOCK, 0, st>>>(__VA_ARGS__); else K<false, 32, 32><<<grid, RGK_TRACE_BLOCK, 0, st>>>(__VA_ARGS__); } \
        else if (tc.lds == 32) { if (count_stats) K<true, 256, 32><<<grid, RGK_TRACE_BLOCK, 0, st>>>(__VA_ARGS__); else K<false, 256, 32><<<grid, RGK_TRACE_BLOCK, 0, st>>>(__VA_ARGS__); } \
        else { if (count_stats) K<true, 256, 16><<<grid, RGK_TRACE_BLOCK, 0, st>>>(__VA_ARGS__); else K<false, 256, 16><<<grid, RGK_TRACE_BLOCK, 0, st>>>(__VA_ARGS__); } \
    }


// ------------------------------------------------------------------ entry points of the camera rays
// All camera rays through a group of RGK_ENTRY_PIX consecutive pixels of the round's list (a row of an 8x8 block, or whatever a
// ragged tile leaves) lie in the pyramid from the eye through the group's pixel rectangle.  Descend from the root while the children that pyramid
// can touch -- box entirely outside one of the four side planes = cannot be touched -- are few: what remains are at most
// RGK_ENTRY_K inner nodes below which every triangle lies that any of those rays can hit.  The traversal kernel starts there
// (all samples of the group's pixels share one descent) instead of at the root: the same triangles are tested, the top levels
// are not walked once per ray.  Conservative by construction (boxes padded, rectangle widened); a lens camera keeps the root.
// The descent both entry-point kernels share.  The region is {x : dot(pl[k], x - apex) >= off[k] for all k < np} (unit normals
// pointing inward); a child box entirely on the outer side of one plane cannot be touched by any ray of the region.
__device__ __forceinline__ void entry_descent(const DevScene& sc, const f3 apex, const f3* pl, const float* off, const int np, const float pad, const f3 axis, int* list_out) {
    int list[RGK_ENTRY_K];
    int cnt = 1;
    list[0] = 0;
    const QNode* __restrict__ nodes = sc.nodes;
    for (int iter = 0; iter < 64; iter++) {
        bool changed = false;
        for (int li = 0; li < cnt && !changed; li++) {
            const QNode q = nodes[list[li]];
            int tc[4], nt = 0;
            bool leaf = false;
            for (int ch = 0; ch < 4; ch++) {
                if (q.qlo[0][ch] > q.qhi[0][ch]) continue; // unused slot
                const float s3[3] = {q.sx, q.sy, q.sz};
                float lo[3], hi[3];
                for (int a = 0; a < 3; a++) { lo[a] = q.p[a] + (float)q.qlo[a][ch] * s3[a] - pad - comp(apex, a); hi[a] = q.p[a] + (float)q.qhi[a][ch] * s3[a] + pad - comp(apex, a); }
                bool outside = false;
                for (int k = 0; k < np && !outside; k++) { // the box corner farthest INSIDE plane k is still outside: the whole box is
                    const float d = (pl[k].x > 0.f ? hi[0] : lo[0]) * pl[k].x + (pl[k].y > 0.f ? hi[1] : lo[1]) * pl[k].y + (pl[k].z > 0.f ? hi[2] : lo[2]) * pl[k].z;
                    const float ext = fabsf(hi[0]) + fabsf(lo[0]) + fabsf(hi[1]) + fabsf(lo[1]) + fabsf(hi[2]) + fabsf(lo[2]);
                    outside = d < off[k] - 1e-5f * ext; // (rounding of the dot product: a few ulps of its largest term)
                }
                if (outside) continue;
                tc[nt++] = q.child[ch];
                if (q.child[ch] < 0) leaf = true;
            }
            if (leaf) continue;                      // a touched child is a leaf: this node stays an entry
            if (cnt - 1 + nt > RGK_ENTRY_K) continue; // no room to open it
            for (int k = li; k + 1 < cnt; k++) list[k] = list[k + 1];
            cnt--;
            for (int k = 0; k < nt; k++) list[cnt++] = tc[k];
            changed = true;
        }
        if (!changed) break;
    }
    // nearest first: by the distance of the node's box centre along the region's axis (any order is correct)
    float key[RGK_ENTRY_K];
    for (int k = 0; k < cnt; k++) { const QNode q = nodes[list[k]]; key[k] = (q.p[0] + 127.f * q.sx - apex.x) * axis.x + (q.p[1] + 127.f * q.sy - apex.y) * axis.y + (q.p[2] + 127.f * q.sz - apex.z) * axis.z; }
    for (int a = 1; a < cnt; a++) for (int b = a; b > 0 && key[b] < key[b - 1]; b--) { const float t = key[b]; key[b] = key[b - 1]; key[b - 1] = t; const int u = list[b]; list[b] = list[b - 1]; list[b - 1] = u; }
    for (int k = 0; k < RGK_ENTRY_K; k++) list_out[k] = k < cnt ? list[k] : STACK_SENTINEL;
}
// pixel rectangle of a group (widened by a twentieth of a pixel) as view-screen fractions, and its four corner directions
__device__ __forceinline__ void group_corners(const DevCamera& cam, uint32_t xres, uint32_t yres, const uint32_t* __restrict__ pix_xy, uint32_t n_pixels, uint32_t g, f3* c) {
    uint32_t x0 = 0xffffu, y0 = 0xffffu, x1 = 0, y1 = 0;
    for (uint32_t j = g * RGK_ENTRY_PIX; j < min(g * RGK_ENTRY_PIX + RGK_ENTRY_PIX, n_pixels); j++) {
        const uint32_t pix = pix_xy[j], x = pix & 0xffffu, y = pix >> 16;
        x0 = min(x0, x); x1 = max(x1, x); y0 = min(y0, y); y1 = max(y1, y);
    }
    const float fx0 = ((float)x0 - 0.05f) / (float)xres, fx1 = ((float)x1 + 1.05f) / (float)xres;
    const float fy0 = ((float)y0 - 0.05f) / (float)yres, fy1 = ((float)y1 + 1.05f) / (float)yres;
    const f3 vs = mk3(cam.viewscreen[0], cam.viewscreen[1], cam.viewscreen[2]) - mk3(cam.origin[0], cam.origin[1], cam.origin[2]);
    const f3 vx = mk3(cam.viewscreen_x[0], cam.viewscreen_x[1], cam.viewscreen_x[2]), vy = mk3(cam.viewscreen_y[0], cam.viewscreen_y[1], cam.viewscreen_y[2]);
    c[0] = vs + fx0 * vx + fy0 * vy; c[1] = vs + fx1 * vx + fy0 * vy; c[2] = vs + fx1 * vx + fy1 * vy; c[3] = vs + fx0 * vx + fy1 * vy;
}
// `trange` (null in a frame's first round): the nearest / farthest first hit per group seen in a finished pass of this frame.
// With it the pyramid is CAPPED behind the farthest one: the list then covers only what a ray can hit up to that distance
// (`cap`), the ray is traced with its far end pulled in to the cap, and if it then finds nothing it is traced again from the
// root -- a hit within the cap is the nearest hit, because everything left out lies beyond it.
__global__ __launch_bounds__(64) void k_entry_points(const DevScene sc, const DevCamera cam, const uint32_t xres, const uint32_t yres,
                                                      const uint32_t* __restrict__ pix_xy, const uint32_t n_pixels, const uint32_t g_first, const uint32_t g_count,
                                                      const uint32_t* __restrict__ trange, int* __restrict__ entries, float* __restrict__ cap) {
    const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= g_count) return;
    const uint32_t g = g_first + gi;
    int* e = entries + (size_t)g * RGK_ENTRY_K;
    cap[g] = __builtin_inff();
    if (cam.lens_size != 0.0f) { e[0] = 0; for (int k = 1; k < RGK_ENTRY_K; k++) e[k] = STACK_SENTINEL; return; }
    f3 c[4];
    group_corners(cam, xres, yres, pix_xy, n_pixels, g, c);
    const f3 cm = c[0] + c[1] + c[2] + c[3];
    f3 pl[5];
    float off[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 4; k++) {
        f3 n = cross3(c[k], c[(k + 1) & 3]);
        if (dot3(n, cm) < 0.f) n = -n;
        pl[k] = n * (1.0f / fmaxf(len3(n), 1e-30f)); // unit inward normal of side plane k (through the eye)
    }
    int np = 4;
    const float pad = 8.0f * sc.epsilon;
    if (trange) {
        const float tmax = __uint_as_float(trange[2 * g + 1]);
        if (tmax > 0.f && tmax < 1e30f) { // a point at distance t along a ray has axial coordinate <= t
            const float far = tmax * 1.02f + pad;
            pl[4] = cm * (-1.0f / fmaxf(len3(cm), 1e-30f));
            off[4] = -(far + pad);
            np = 5;
            cap[g] = far;
        }
    }
    int list[RGK_ENTRY_K];
    entry_descent(sc, mk3(cam.origin[0], cam.origin[1], cam.origin[2]), pl, off, np, pad, cm, list);
    for (int k = 0; k < RGK_ENTRY_K; k++) e[k] = list[k];
}

// ------------------------------------------------------------------ entry points of the first vertex's shadow rays
// A scene lit by ONE point or sphere light sends every shadow ray FROM that light (Ray(light.pos, p.pos, 20 eps), src/ray.hpp:15-22)
// to a first hit of the group's camera rays, and those first hits lie in the slice of the group's view pyramid between the
// nearest and the farthest of them.  k_group_trange collects that distance range per pixel group from the hit records of a
// pass; k_entry_points_light bounds the slice by a box, spans the pyramid from the light over that box (capped behind it) and
// descends the tree exactly like k_entry_points.  A sphere light's start points lie within `size` of its centre: the pyramid's
// apex moves back so that it holds the light's box as well.
__global__ __launch_bounds__(256) void k_group_trange(const PassParams pp, const float4* __restrict__ hit, const uint32_t n_slots, uint32_t* __restrict__ trange) {
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < ((n_slots + 63u) & ~63u); slot += gridDim.x * blockDim.x) {
        const bool valid = slot < n_slots;
        uint32_t srel = 0, j = 0;
        if (valid) slot_decode(pp, slot, j, srel);
        const uint32_t g = valid ? (pp.j0 + j) >> RGK_ENTRY_SHIFT : 0xffffffffu;
        float tmin = __builtin_inff(), tmax = 0.f;
        if (valid) { const float4 h = hit[slot]; if (__float_as_int(h.w) >= 0) tmin = tmax = h.x; }
        const uint32_t g0 = __builtin_amdgcn_readfirstlane(g);
        if (__builtin_amdgcn_ballot_w64(g != g0) == 0ull) { // the whole wave is one group (the usual case: 8 pixels x 8 samples)
            for (int o = 32; o > 0; o >>= 1) { tmin = fminf(tmin, __shfl_xor(tmin, o)); tmax = fmaxf(tmax, __shfl_xor(tmax, o)); }
            if ((threadIdx.x & 63) == 0 && g0 != 0xffffffffu && tmax > 0.f) { atomicMin(&trange[2 * g0], __float_as_uint(tmin)); atomicMax(&trange[2 * g0 + 1], __float_as_uint(tmax)); }
        } else if (valid && tmax > 0.f) { atomicMin(&trange[2 * g], __float_as_uint(tmin)); atomicMax(&trange[2 * g + 1], __float_as_uint(tmax)); }
    }
}
// The same ranges with ONE WAVE PER GROUP and no atomics: lane = (pixel of the group, sample of a block of 8), the wave walks the
// pass's sample blocks (with 8 samples side by side in the slot order, 64 consecutive slots = 1 KB per step) and writes the
// group's two words once.  The atomic form above sends 2 atomics per 64 slots at words that sit 16 groups to a 128-byte line,
// and consecutive waves ARE consecutive groups: 1.6 ms per 530 M slots at 256 spp, but 21 ms per 212 M at 512 spp (configs[3]:
// four such passes in a frame's first round).  Distances compare as their bit patterns, as the atomics did.
__global__ __launch_bounds__(64) void k_group_trange_wave(const PassParams pp, const float4* __restrict__ hit, const uint32_t g_first, const uint32_t groups,
                                                           uint32_t* __restrict__ trange) {
    const uint32_t gi = blockIdx.x;
    if (gi >= groups) return;
    const uint32_t g = g_first + gi, lane = threadIdx.x, p = lane >> 3, k = lane & 7u;
    const long long jj = (long long)((unsigned long long)g << RGK_ENTRY_SHIFT) + (long long)p - (long long)pp.j0; // the lane's pixel within the pass
    const bool pix_ok = jj >= 0 && jj < (long long)pp.npix;
    uint32_t tmin = 0x7f800000u, tmax = 0u;
#pragma unroll 4
    for (uint32_t s0 = 0; s0 < pp.ns; s0 += 8u) {
        const uint32_t srel = s0 + k;
        if (pix_ok && srel < pp.ns) {
            const float4 h = hit[slot_of(pp, (uint32_t)jj, srel)];
            if (__float_as_int(h.w) >= 0 && h.x > 0.f) { const uint32_t t = __float_as_uint(h.x); tmin = min(tmin, t); tmax = max(tmax, t); }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { tmin = min(tmin, (uint32_t)__shfl_xor((int)tmin, o)); tmax = max(tmax, (uint32_t)__shfl_xor((int)tmax, o)); }
    if (lane == 0) { trange[2 * (size_t)g] = tmin; trange[2 * (size_t)g + 1] = tmax; }
}
__global__ void k_init_trange(uint32_t* trange, uint32_t groups) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += gridDim.x * blockDim.x) { trange[2 * i] = 0x7f800000u; trange[2 * i + 1] = 0u; }
}
__global__ __launch_bounds__(64) void k_entry_points_light(const DevScene sc, const DevCamera cam, const uint32_t xres, const uint32_t yres,
                                                            const uint32_t* __restrict__ pix_xy, const uint32_t n_pixels, const uint32_t g_first, const uint32_t g_count,
                                                            const uint32_t* __restrict__ trange, int* __restrict__ entries, float4* __restrict__ lbox) {
    const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= g_count) return;
    const uint32_t g = g_first + gi;
    int* e = entries + (size_t)g * RGK_ENTRY_K;
    e[0] = 0;
    for (int k = 1; k < RGK_ENTRY_K; k++) e[k] = STACK_SENTINEL;
    // the box the entries below are good for; a shadow ray that ends outside it starts at the root (the range comes from the
    // first hits of one pass of the frame's first round, widened -- later rounds jitter differently).  Fallbacks: the root, good
    // for every ray.
    lbox[2 * (size_t)g] = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), 0.f);
    lbox[2 * (size_t)g + 1] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
    const float tmin = __uint_as_float(trange[2 * g]), tmax = __uint_as_float(trange[2 * g + 1]);
    if (!(tmax > 0.f) || !(tmin <= tmax) || cam.lens_size != 0.0f) return; // no first hit in this group (no shadow ray will ask), or a lens camera: the root
    const DevPointLight L = sc.pointlights[0];
    const f3 lp = mk3(L.pos[0], L.pos[1], L.pos[2]), eye = mk3(cam.origin[0], cam.origin[1], cam.origin[2]);
    f3 c[4];
    group_corners(cam, xres, yres, pix_xy, n_pixels, g, c);
    f3 axis = c[0] + c[1] + c[2] + c[3];
    axis = axis * (1.0f / fmaxf(len3(axis), 1e-30f));
    // the first hits: eye + t d with |d| = 1 inside the pyramid and tmin <= t <= tmax, so their axial coordinate lies between
    // tmin * (the smallest corner cosine) and tmax: the slice of the pyramid between those two planes, a polytope with 8 corners
    float cmin = 1.f;
    for (int k = 0; k < 4; k++) cmin = fminf(cmin, dot3(c[k], axis) / fmaxf(len3(c[k]), 1e-30f));
    if (!(cmin > 0.1f)) return; // (a group spread over more than ~80 degrees: the root)
    const float a_near = tmin * cmin * 0.98f, a_far = tmax * 1.02f; // (widened: the range is that of ONE pass of ONE round; the lists serve the frame)
    float blo[3] = {1e30f, 1e30f, 1e30f}, bhi[3] = {-1e30f, -1e30f, -1e30f};
    for (int k = 0; k < 4; k++) {
        const float ca = dot3(c[k], axis);
        const f3 vn = eye + c[k] * (a_near / ca), vf = eye + c[k] * (a_far / ca);
        for (int a = 0; a < 3; a++) { blo[a] = fminf(blo[a], fminf(comp(vn, a), comp(vf, a))); bhi[a] = fmaxf(bhi[a], fmaxf(comp(vn, a), comp(vf, a))); }
    }
    const float pad = 8.0f * sc.epsilon;
    for (int a = 0; a < 3; a++) { const float m = 1e-5f * (fabsf(blo[a]) + fabsf(bhi[a])) + pad; blo[a] -= m; bhi[a] += m; }
    // the pyramid over that box.  A point light is its apex.  A sphere light's rays start anywhere within `size` of its centre:
    // the apex moves back along the axis to where the lines from the light's rim to the box's rim meet (behind the light when
    // the box is the larger of the two, far behind -- an almost parallel shaft -- when it is not), and the side planes are
    // taken over the corners of BOTH boxes, so the region contains every segment between them.
    const f3 ctr = mk3(0.5f * (blo[0] + bhi[0]), 0.5f * (blo[1] + bhi[1]), 0.5f * (blo[2] + bhi[2]));
    f3 w = ctr - lp;
    const float wl = len3(w);
    if (!(wl > 0.f)) return;
    w = w * (1.0f / wl);
    const float rl = L.size * 1.001f + (L.size > 0.f ? pad : 0.f);
    const float hb = 0.5f * sqrtf((bhi[0] - blo[0]) * (bhi[0] - blo[0]) + (bhi[1] - blo[1]) * (bhi[1] - blo[1]) + (bhi[2] - blo[2]) * (bhi[2] - blo[2]));
    float back = 0.f;
    if (rl > 0.f) back = fmaxf(2.0f * rl, hb > 1.05f * rl ? wl * rl / (hb - rl) : 100.0f * wl);
    const f3 apex = lp - w * back;
    const f3 t0 = fabsf(w.x) > 0.9f ? mk3(0.f, 1.f, 0.f) : mk3(1.f, 0.f, 0.f);
    f3 u = cross3(w, t0); u = u * (1.0f / len3(u));
    const f3 v = cross3(w, u);
    float amin = 1e30f, amax = -1e30f, bmin = 1e30f, bmax = -1e30f, wmax = 0.f, wmin = 1e30f;
    for (int k = 0; k < (rl > 0.f ? 16 : 8); k++) {
        const f3 q = k < 8 ? mk3((k & 1) ? bhi[0] : blo[0], (k & 2) ? bhi[1] : blo[1], (k & 4) ? bhi[2] : blo[2])
                           : mk3(lp.x + ((k & 1) ? rl : -rl), lp.y + ((k & 2) ? rl : -rl), lp.z + ((k & 4) ? rl : -rl));
        const f3 r = q - apex;
        const float rw = dot3(r, w);
        wmin = fminf(wmin, rw);
        if (k < 8) wmax = fmaxf(wmax, rw);
        if (rw > 0.f) { const float ra = dot3(r, u) / rw, rb = dot3(r, v) / rw; amin = fminf(amin, ra); amax = fmaxf(amax, ra); bmin = fminf(bmin, rb); bmax = fmaxf(bmax, rb); }
    }
    if (!(wmin > 0.02f * wmax)) return; // the light sits beside or inside the box: no useful pyramid, the root
    const float am = 1e-4f * (1.f + fabsf(amin) + fabsf(amax)), bm = 1e-4f * (1.f + fabsf(bmin) + fabsf(bmax));
    amin -= am; amax += am; bmin -= bm; bmax += bm;
    f3 pl[5];
    float off[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    pl[0] = u - amin * w; pl[1] = amax * w - u; pl[2] = v - bmin * w; pl[3] = bmax * w - v;
    for (int k = 0; k < 4; k++) pl[k] = pl[k] * (1.0f / len3(pl[k]));
    pl[4] = -w; off[4] = -(wmax * 1.001f + pad); // behind the box: dot(x - apex, w) <= wmax
    int list[RGK_ENTRY_K];
    entry_descent(sc, apex, pl, off, 5, pad, w, list);
    for (int k = 0; k < RGK_ENTRY_K; k++) e[k] = list[k];
    lbox[2 * (size_t)g] = make_float4(blo[0], blo[1], blo[2], 0.f);
    lbox[2 * (size_t)g + 1] = make_float4(bhi[0], bhi[1], bhi[2], 0.f);
}
void rgk_launch_entry_points(hipStream_t st, const DevScene& sc, const DevCamera& cam, uint32_t xres, uint32_t yres, const uint32_t* pix_xy, uint32_t n_pixels,
                             uint32_t g_first, uint32_t g_count, const uint32_t* trange, int* entries, float* cap) {
    k_entry_points<<<(g_count + 63u) / 64u, 64, 0, st>>>(sc, cam, xres, yres, pix_xy, n_pixels, g_first, g_count, trange, entries, cap);
}
// nearest / farthest first hit per pixel group of a finished bounce-0 trace of this pass
void rgk_launch_group_trange(hipStream_t st, const PassParams& pp, const float4* hit, uint32_t* trange) {
    const uint32_t g_first = pp.j0 >> RGK_ENTRY_SHIFT, g_last = (pp.j0 + pp.npix + RGK_ENTRY_PIX - 1u) >> RGK_ENTRY_SHIFT, groups = g_last - g_first;
#if RGK_TRANGE_ATOMIC
    k_init_trange<<<(groups + 255u) / 256u, 256, 0, st>>>(trange + 2 * (size_t)g_first, groups);
    const uint32_t n = pp.npix * pp.ns; // every slot of this pass (done once per frame and pixel range, so 16 bytes per path do not matter)
    uint32_t blocks = (n + 255u) / 256u;
    if (blocks > 256u * 64u) blocks = 256u * 64u;
    k_group_trange<<<blocks, 256, 0, st>>>(pp, hit, n, trange);
#else
    k_group_trange_wave<<<groups, 64, 0, st>>>(pp, hit, g_first, groups, trange);
#endif
}
void rgk_launch_light_entry_points(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PassParams& pp, uint32_t n_pixels_round, const uint32_t* trange, int* entries, float4* lbox) {
    const uint32_t g_first = pp.j0 >> RGK_ENTRY_SHIFT, g_last = (pp.j0 + pp.npix + RGK_ENTRY_PIX - 1u) >> RGK_ENTRY_SHIFT, groups = g_last - g_first;
    k_entry_points_light<<<(groups + 63u) / 64u, 64, 0, st>>>(sc, cam, pp.xres, pp.yres, pp.pix_xy, n_pixels_round, g_first, groups, trange, entries, lbox);
}

__global__ void k_stage_mark(uint32_t* host_word, uint32_t v) { *(volatile uint32_t*)host_word = v; __threadfence_system(); }
void rgk_launch_stage_mark(hipStream_t st, uint32_t* host_word, uint32_t v) { k_stage_mark<<<1, 1, 0, st>>>(host_word, v); }
void rgk_launch_init_counters(hipStream_t st, uint32_t* counters, uint32_t n0) { k_init_counters<<<1, 256, 0, st>>>(counters, n0); }
void rgk_launch_build_pixel_list(hipStream_t st, const rgk_tile* tiles, const uint32_t* tile_off, uint32_t n_tiles, uint32_t* pix_xy, uint32_t* pix_seed) {
    k_build_pixel_list<<<n_tiles < 4096u ? n_tiles : 4096u, 256, 0, st>>>(tiles, tile_off, n_tiles, pix_xy, pix_seed);
}
void rgk_launch_build_halton_table(hipStream_t st, const DevScene& sc, uint32_t S, float* htab) {
    uint32_t n = 192u * S;
    k_build_halton_table<<<(n + 255) / 256, 256, 0, st>>>(sc, S, htab);
}

// bounce 0 of a unidirectional pass: camera rays generated in the traversal kernel (no queue)
void rgk_launch_trace_camera(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PassParams& pp, const RgkTraceCfg& tc, bool count_stats, float4* hit,
                             const uint32_t* count_ptr, uint32_t* fetch, unsigned long long* stats) {
    if (pp.beam && pp.gshift == 3 && cam.lens_size == 0.0f && tc.lds < tc.stack) {
        // 8 stack entries per lane in LDS (the rest in the overflow area): with the bundle's 32 floats per lane that makes 40 KB per
        // workgroup, four workgroups per CU
        const int grid = bounded_grid(rgk_trace_grid(tc.lds), (g_bound >> 3) + 1u, RGK_TRACE_BLOCK);
        if (count_stats) k_trace_camera_beam<true, 256, 8><<<grid, RGK_TRACE_BLOCK, 0, st>>>(sc, cam, pp, hit, count_ptr, fetch, stats, tc.ovf);
        else k_trace_camera_beam<false, 256, 8><<<grid, RGK_TRACE_BLOCK, 0, st>>>(sc, cam, pp, hit, count_ptr, fetch, stats, tc.ovf);
        return;
    }
    RGK_TRACE_DISPATCH(k_trace_camera, g_bound, sc, cam, pp, hit, count_ptr, fetch, stats, tc.ovf)
}

void rgk_launch_trace_closest(hipStream_t st, const DevScene& sc, const RgkTraceCfg& tc, bool count_stats, const float4* rayA, const float4* rayB,
                              const float2* nearfar, float4* hit, const uint32_t* count_ptr, uint32_t* fetch, unsigned long long* stats) {
    RGK_TRACE_DISPATCH(k_trace_closest, g_bound, sc, rayA, rayB, nearfar, hit, count_ptr, fetch, stats, tc.ovf)
}

void rgk_launch_trace_shadow(hipStream_t st, const DevScene& sc, const RgkTraceCfg& tc, bool count_stats, const float4* shA, const float4* shB,
                             const float4* shC, float4* tot, uint8_t* vis_out, int mode, float* splat_rgb, const uint32_t* count_ptr,
                             uint32_t* fetch, unsigned long long* stats) {
    RGK_TRACE_DISPATCH(k_trace_shadow, g_bound_shadow, sc, shA, shB, shC, tot, vis_out, mode, splat_rgb, count_ptr, fetch, stats, tc.ovf)
}

void rgk_launch_trace_shadow_first(hipStream_t st, const DevScene& sc, const PassParams& pp, const RgkTraceCfg& tc, bool count_stats, const float4* shA, const float4* shB,
                                   const float4* shC, float4* tot, const uint32_t* count_ptr, uint32_t* fetch, unsigned long long* stats) {
    RGK_TRACE_DISPATCH(k_trace_shadow_first, g_bound_shadow, sc, pp, shA, shB, shC, tot, count_ptr, fetch, stats, tc.ovf)
}

void rgk_launch_shade(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PassParams& pp, uint32_t bounce, const float4* rayA,
                      const float4* rayB, const float4* hit, float4* thr, float4* tot, float4* nextA, float4* nextB, float4* shA,
                      float4* shB, float4* shC, uint32_t* counters, bool bdpt) {
    const int blk = bounce == 0 ? RGK_SHADE_BLOCK : RGK_SHADE_BLOCK_LATER;
    const int g1 = bounded_grid(256 * 4 * 512 / blk, g_bound, blk), g2 = bounded_grid(256 * 2 * 512 / blk, g_bound, blk);
    // the second launch shades the vertices the first one listed (materials on the generic BxDF route); it returns at once when there are none
    if (bdpt && bounce == 0) {
        k_shade<false, true, true><<<g1, blk, RGK_LDS_SHADE_BYTES, st>>>(sc, cam, pp, bounce, rayA, rayB, hit, thr, tot, nextA, nextB, shA, shB, shC, counters);
        k_shade<true, true, true><<<g2, blk, RGK_LDS_SHADE_BYTES, st>>>(sc, cam, pp, bounce, rayA, rayB, hit, thr, tot, nextA, nextB, shA, shB, shC, counters);
    } else if (bdpt) {
        k_shade<false, false, true><<<g1, blk, RGK_LDS_SHADE_BYTES, st>>>(sc, cam, pp, bounce, rayA, rayB, hit, thr, tot, nextA, nextB, shA, shB, shC, counters);
        k_shade<true, false, true><<<g2, blk, RGK_LDS_SHADE_BYTES, st>>>(sc, cam, pp, bounce, rayA, rayB, hit, thr, tot, nextA, nextB, shA, shB, shC, counters);
    } else if (bounce == 0) {
        k_shade<false, true><<<g1, blk, RGK_LDS_SHADE_BYTES, st>>>(sc, cam, pp, bounce, rayA, rayB, hit, thr, tot, nextA, nextB, shA, shB, shC, counters);
        k_shade<true, true><<<g2, blk, RGK_LDS_SHADE_BYTES, st>>>(sc, cam, pp, bounce, rayA, rayB, hit, thr, tot, nextA, nextB, shA, shB, shC, counters);
    } else {
        k_shade<false, false><<<g1, blk, RGK_LDS_SHADE_BYTES, st>>>(sc, cam, pp, bounce, rayA, rayB, hit, thr, tot, nextA, nextB, shA, shB, shC, counters);
        k_shade<true, false><<<g2, blk, RGK_LDS_SHADE_BYTES, st>>>(sc, cam, pp, bounce, rayA, rayB, hit, thr, tot, nextA, nextB, shA, shB, shC, counters);
    }
}

void rgk_launch_resolve(hipStream_t st, const PassParams& pp, const float4* tot, float4* pixsum, float* accum_rgb, uint32_t* accum_count) {
    int grid = (int)((pp.npix + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    if (pp.gshift == 0) { k_resolve<<<grid, 256, 0, st>>>(pp, tot, pixsum, accum_rgb, accum_count); return; }
    const uint32_t G = 1u << pp.gshift;
    const uint32_t PT = G <= 8 ? 64u : 512u / G; // pixels per tile: ~9 KB of LDS per wave (more waves per CU matter more here than full lanes in the short summing phase)
    int tiles = (int)((pp.npix + PT - 1) / PT);
    k_resolve_tiled<<<tiles > 256 * 64 ? 256 * 64 : tiles, 64, PT * (G + 1) * sizeof(float4), st>>>(pp, tot, pixsum, accum_rgb, accum_count, PT);
}

void rgk_launch_pack_rays(hipStream_t st, uint32_t n, const float* rays, const int32_t* ignore, float4* rayA, float4* rayB, float2* nearfar) {
    k_pack_rays<<<(n + 255) / 256, 256, 0, st>>>(n, rays, ignore, rayA, rayB, nearfar);
}
void rgk_launch_pack_visibility(hipStream_t st, const DevScene& sc, uint32_t n, const float* a, const float* b, float4* shA, float4* shB, float4* shC) {
    k_pack_visibility<<<(n + 255) / 256, 256, 0, st>>>(sc, n, a, b, shA, shB, shC);
}
void rgk_launch_unpack_hits(hipStream_t st, uint32_t n, const float4* hit, rgk_hit* out) { k_unpack_hits<<<(n + 255) / 256, 256, 0, st>>>(n, hit, out); }
void rgk_launch_sampler_eval(hipStream_t st, const DevScene& sc, uint32_t n, const uint32_t* seed, const uint32_t* index, const uint32_t* dim,
                             int is2d, float* out) {
    k_sampler_eval<<<(n + 255) / 256, 256, 0, st>>>(sc, n, seed, index, dim, is2d, out);
}

static inline int slot_grid(const PassParams& pp) {
    uint32_t n = pp.npix * pp.ns;
    int grid = (int)((n + 255) / 256);
    return grid > 256 * 16 ? 256 * 16 : grid;
}
void rgk_launch_raygen_light(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PassParams& pp, float4* rayA, float4* rayB,
                             float4* thr, uint32_t* counters) {
    const uint32_t n = pp.npix * pp.ns;
    k_raygen_light<<<bounded_grid(256 * 4 * 512 / RGK_LIGHT_BLOCK, n, RGK_LIGHT_BLOCK), RGK_LIGHT_BLOCK, 0, st>>>(sc, cam, pp, rayA, rayB, thr, counters);
}
void rgk_launch_shade_light(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PassParams& pp, uint32_t k, const float4* rayA,
                            const float4* rayB, const float4* hit, float4* thr, float4* nextA, float4* nextB, float4* shA, float4* shB,
                            float4* shC, uint32_t* counters) {
    k_shade_light<false><<<bounded_grid(256 * 4 * 512 / RGK_LIGHT_BLOCK, g_bound, RGK_LIGHT_BLOCK), RGK_LIGHT_BLOCK, RGK_LDS_SHADE_BYTES, st>>>(sc, cam, pp, k, rayA, rayB, hit, thr, nextA, nextB, shA, shB, shC, counters);
    k_shade_light<true><<<bounded_grid(256 * 2 * 512 / RGK_LIGHT_BLOCK, g_bound, RGK_LIGHT_BLOCK), RGK_LIGHT_BLOCK, RGK_LDS_SHADE_BYTES, st>>>(sc, cam, pp, k, rayA, rayB, hit, thr, nextA, nextB, shA, shB, shC, counters);
}
void rgk_launch_connect(hipStream_t st, const DevScene& sc, const PassParams& pp, uint32_t bounce, float4* jobs, float4* rads, uint32_t* counters) {
    k_connect<<<bounded_grid(256 * 8, g_bound, 256), 256, RGK_LDS_SHADE_BYTES, st>>>(sc, pp, bounce, jobs, rads, counters);
}
void rgk_launch_list_hits(hipStream_t st, const float4* hit, const uint32_t* count_ptr, uint32_t* list, uint32_t* list_count) {
    k_list_hits<<<bounded_grid(256 * 4 * 512 / RGK_LIGHT_BLOCK, g_bound, RGK_LIGHT_BLOCK), RGK_LIGHT_BLOCK, 0, st>>>(hit, count_ptr, list, list_count);
}
void rgk_launch_trace_shadow_jobs(hipStream_t st, const DevScene& sc, const PassParams& pp, const RgkTraceCfg& tc, bool count_stats, const float4* jobs, const float4* rads,
                                  float4* tot, const uint32_t* count_ptr, uint32_t* fetch, unsigned long long* stats) {
    RGK_TRACE_DISPATCH(k_trace_shadow_jobs, g_bound_shadow, sc, pp, jobs, rads, tot, count_ptr, fetch, stats, tc.ovf)
}
