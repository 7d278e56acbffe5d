// Ray traversal kernels (K2 closest hit, K5 shadow / visibility) for gfx950.
//
// Result semantics are the reference's kd traversal (src/scene_intersect.cpp:4-116,211-327):
// nearest accepted hit with t in [t0 - eps, t1 + eps] where [t0, t1] is the ray's [near, far]
// clipped to the epsilon-padded scene box (:223-232,272), exact ties to the higher triangle id (see the leaf loop), the
// `ignore` triangle is skipped; the triangle test is Triangle::TestIntersection
// (src/primitives.cpp:75-166) with its fp64 plane solve.  The accelerator is the build's own:
// a 4-wide BVH with 8-bit quantised child boxes, one 64-byte line per node (device_types.h).
//
// Execution model: persistent waves with per-lane ray REFILL.  The kernel is bound by the VALU -- precisely by the pipe its
// non-fp32 instructions share (0.92-0.97 of that ceiling) -- and rays of one 64-ray batch finish at very
// different times.  So a wave keeps a cursor into its
// chunk of the ray queue and, whenever at most RGK_REFILL_BELOW of its lanes still hold a ray, hands the idle ones fresh rays
// (ballot + prefix popcount); a lane that drains its stack stores its hit and goes idle.
// The first LDSN entries of the per-lane traversal stack live in LDS as [entry][lane] (bank = lane: conflict-free), the
// deeper ones per lane in global memory: with the whole stack in LDS the kernel was LDS-bound at 5 waves per SIMD.
//
// Round-1 counters after the refill (profiles/r01_trace_pmc.txt): the vector L1 is ~90 % occupied too (TA busy 67 % + 32 %
// pending-line stalls; L1 hit rate 93 %) -- the kernel is co-limited, which is why two one-sided trades lost: 128-byte nodes with full-precision planes (85 VALU per node instead of 150 thanks
// to packed fma and no decode, but 7 loads per visit instead of 4: 30.1 vs 25.9 ms) and keeping the top 96..512
// nodes in LDS (fewer L1 lookups, but a divergent LDS / global choice per visit: 23.7..24.1 vs 23.0 ms).
#pragma once
#include <hip/hip_runtime.h>
#include "rgk_device.h"
#include "rgk_kernels.h"

#define STACK_SENTINEL 0x7fffffff
#ifndef RGK_CHUNK_MAX
#define RGK_CHUNK_MAX 2048u
#endif
#ifndef RGK_TRACE_WAVES
#define RGK_TRACE_WAVES 8 // waves per SIMD the 16-LDS-entry traversal kernels are compiled for (64 VGPRs); the 32-entry ones are LDS-bound at 5
#endif
#ifndef RGK_JOB_FINISH_ATOMIC
#define RGK_JOB_FINISH_ATOMIC 0 // the vertex total into the slot sum as three float atomics instead of a read-modify-write
#endif
#ifndef RGK_JOB_REFILL_BELOW
#define RGK_JOB_REFILL_BELOW 24 // the vertex queue's walker: lanes between two rays of their vertex wait for the refill too
#endif
// (per launch kind, swept again at the end of round 3 on Sponza: camera 8 / 16 / 24: 15.9 / 15.6 / 15.6 ms; any-hit 12 / 16 / 24 / 32: 8.9 / 8.7 /
// 8.4 / 8.4; closest-hit 24 ... 48: flat.  Cornell's short rays would take 48 for the closest-hit launches, 36.4 -> 34.5 ms, and 24 for the shadow ones.)
#ifndef RGK_REFILL_BELOW_CAMERA
#define RGK_REFILL_BELOW_CAMERA 24
#endif
#ifndef RGK_REFILL_BELOW_ANY
#define RGK_REFILL_BELOW_ANY 24
#endif
#ifndef RGK_REFILL_BELOW
#define RGK_REFILL_BELOW 24 // refill a wave when at most this many lanes still hold a ray (swept 8..60 with the majority walk: 24 best)
#endif

// ------------------------------------------------------------------ camera rays (K1)
// Camera::GetPixelRay / GetPixelRayLens, reference src/camera.cpp:26-46; Ray ctor src/ray.hpp:10-13
__device__ __forceinline__ void camera_ray(const DevCamera& cam, int x, int y, int xres, int yres, float2 off, float2 lens, f3& o, f3& d) {
    float fx = (x + off.x) / (float)(xres), fy = (y + off.y) / (float)(yres);
    f3 vs = mk3(cam.viewscreen[0], cam.viewscreen[1], cam.viewscreen[2]);
    f3 vx = mk3(cam.viewscreen_x[0], cam.viewscreen_x[1], cam.viewscreen_x[2]);
    f3 vy = mk3(cam.viewscreen_y[0], cam.viewscreen_y[1], cam.viewscreen_y[2]);
    f3 p = vs + fx * vx + fy * vy;
    o = mk3(cam.origin[0], cam.origin[1], cam.origin[2]);
    if (cam.lens_size != 0.0f) {
        float2 dsc = disc_uniform(lens);
        float lx = dsc.x * cam.lens_size, ly = dsc.y * cam.lens_size;
        o = o + lx * mk3(cam.left[0], cam.left[1], cam.left[2]) + ly * mk3(cam.up[0], cam.up[1], cam.up[2]);
    }
    d = norm3(p - o);
}
// The camera ray of path slot `slot` of the current pass (RenderPixel, reference src/path_tracer.cpp:53-61): pixel and sample from
// the slot, pixel jitter = 2-D dimension 0, lens = 2-D dimension 1.  A few dozen instructions and two cached table reads, so
// the bounce-0 traversal and the bounce-0 shading each derive it from the slot number instead of one kernel writing 32 bytes
// per path for the other two to read back (k_raygen was 4 % of a Sponza round, all of it HBM traffic).
__device__ __forceinline__ void camera_ray_of_slot(const DevCamera& cam, const PassParams& pp, uint32_t slot, f3& o, f3& d, uint32_t* j_out = nullptr) {
    uint32_t srel, j; slot_decode(pp, slot, j, srel);
    if (j_out) *j_out = pp.j0 + j;
    const uint32_t pix = pp.pix_xy[pp.j0 + j], seed = pp.pix_seed[pp.j0 + j];
    const uint32_t s = pp.s0 + srel;
    const SamplerTab tb = {pp.htab, pp.multisample};
    const float2 jit = sample2d_t(tb, seed, s, 0);
    float2 lens = make_float2(0.f, 0.f);
    if (cam.lens_size != 0.0f) lens = sample2d_t(tb, seed, s, 1);
    camera_ray(cam, (int)(pix & 0xffff), (int)(pix >> 16), (int)pp.xres, (int)pp.yres, jit, lens, o, d);
}

// Triangle::TestIntersection, reference src/primitives.cpp:75-166.  r0..r2 = TriIsect.
__device__ __forceinline__ bool tri_test(const float4 r0, const float4 r1, const float4 r2, const f3 o, const f3 d,
                                         const float eps, float& t, float& alpha, float& beta) {
#if defined(RGK_DIAG_NO_TRI) /* timing diagnosis only (wrong images): what the walkers cost without their triangle tests */
    t = r0.x + r1.x + r2.x + o.x + d.x; alpha = beta = 0.f;
    return t == 12345.678f;
#endif
    f3 n = mk3(r0.x, r0.y, r0.z);
    double dotv = (double)dot3(d, n);
    if (dotv != dotv) return false;
    if (dotv < (double)eps && dotv > (double)(-eps)) return false;
    double dot2 = (double)dot3(o, n);
    t = (float)(-((double)r0.w + dot2) / dotv);
    uint32_t axes = __float_as_uint(r2.z);
    int i1 = axes & 3, i2 = (axes >> 2) & 3;
    float px = comp(o, i1) + comp(d, i1) * t;
    float py = comp(o, i2) + comp(d, i2) * t;
    float q0x = px - r1.x, q0y = py - r1.y;
    float q1x = r1.z, q1y = r1.w, q2x = r2.x, q2y = r2.y;
    // the reference's two cases (q1x ~ 0 or not) as ONE pair of divisions with selected operands: the same operations on the
    // same values, but a wave whose lanes disagree about the case no longer runs four IEEE divisions
    if (q1x > -eps && q1x < eps) {
        beta = q0x / q2x;
        if (beta < 0 || beta > 1) return false;
        alpha = (q0y - beta * q2y) / q1y;
    } else {
        beta = (q0y * q1x - q0x * q1y) / (q2y * q1x - q2x * q1y);
        if (beta < 0 || beta > 1) return false;
        alpha = (q0x - beta * q2x) / q1x;
    }
    if (alpha < 0 || (double)(alpha + beta) > 1.0) return false;
    return true;
}

// (A conservative fp32 look at a (ray, triangle) pair BEFORE this test -- plane distance through the hardware reciprocal, the hit
// point's three edge functions without any division, every comparison with a margin far above the rounding that separates it
// from the exact test, NaN-safe -- was built and measured in round 3: every parity test stayed bit-identical, and it lost
// everywhere.  In the per-ray walkers its ~20 extra live registers turn the 64-VGPR / 8-wave kernels' 12-44 bytes of scratch
// into 100-160: Sponza 1080p x 256, ms per round 126.8 -> 195.0 at 8 waves per SIMD, 153.8 at 7, 143.3 at 6.  In the bundle
// walker, which has registers to spare, the camera launch went 17.4 -> 18.8 ms: the exact test's own early outs are about as
// cheap as the look that would avoid it.  Running every test TWICE costs the camera launch +6.2 ms of 16.1, the bounce launch
// +8.8 of 36.2, the shadow launches +2.4 of 16.9 -- triangle tests are 14 % of a round; the node loads and box tests, doubled the
// same way, +4.7, +17.1 and +6.2.)
// Scene-bbox clip of [near, far], reference src/scene_intersect.cpp:223-232
__device__ __forceinline__ bool clip_to_scene(const DevScene& sc, f3 o, f3 d, float tnear, float tfar, float& t0, float& t1) {
    t0 = tnear; t1 = tfar;
    for (int i = 0; i < 3; ++i) {
        float invRayDir = 1.f / comp(d, i);
        float tN = (sc.bb_min[i] - comp(o, i)) * invRayDir;
        float tF = (sc.bb_max[i] - comp(o, i)) * invRayDir;
        if (tN > tF) { float s = tN; tN = tF; tF = s; }
        t0 = tN > t0 ? tN : t0;
        t1 = tF < t1 ? tF : t1;
        if (t0 > t1) return false;
    }
    return true;
}

// A wave takes `chunk` consecutive rays per visit to the device-wide cursor.  One returning atomic
// on a single word sustains only ~88 dequeues/us chip-wide (MI355X guide, price list "dequeue"):
// at 64 rays per dequeue that alone capped the kernel at ~5.6 G rays/s, so the chunk grows with
// the queue.  The last piece a wave takes is the launch's tail, so a wave's share comes in `pieces` pieces: 16 for the
// closest-hit launches (one rank's eighth of a frame: 10.0 -> 9.5 ms per round against 4 pieces), 4 for the shadow launches,
// whose rays are short enough that 16 pieces ran into the dequeue rate (2.5 -> 3.8 ms).
__device__ __forceinline__ uint32_t fetch_chunk(uint32_t count, uint32_t nwaves, uint32_t pieces) {
    uint32_t c = (count / (nwaves * pieces)) & ~63u;
    return c < 64u ? 64u : (c > RGK_CHUNK_MAX ? RGK_CHUNK_MAX : c);
}

__device__ __forceinline__ float cvt_ubyte(uint32_t w, int c) { return (float)((w >> (8 * c)) & 0xffu); }

// ANY = false: closest hit, results to hit[i] = {t, alpha, beta, tri}.
// ANY = true : Scene::Visibility; vis_out[i] = visible, or (path mode) tot[slot] += radiance if visible.
//   q0 = {o.xyz, d.x}; q1 = {d.y, d.z, ignore | far, slot}; q2 = {radiance.rgb, near} (shadow only)
// STACK: entries the tree can need (host bound); LDSN <= STACK of them live in LDS, the rest -- reached only on the
// deepest walks of a deep tree -- in a per-lane global overflow area, so that a deep tree does not halve the occupancy.
// RAYGEN (closest hit, bounce 0): there is no ray queue -- ray i is the camera ray of path slot i, made here.
// JOB (any hit, bidirectional rounds): a queue entry is not a ray but a camera-path VERTEX with its 1 + reverse shadow rays --
// NEE to the path's light and the connections to the path's light vertices (path_tracer.cpp:427-480).  They all END at the
// vertex and START at points the slot already holds (pp.light, pp.lv), so the entry carries only the vertex, its contribution,
// and one radiance per ray: q0[i] = {p.xyz, slot}, q0[batch + i] = {contribution.rgb, mask}, q0[2 batch + i] = {emission.rgb}
// (read only when mask bit 8 says it is non-zero), q0[3 batch + i] = {where the NEE ray starts}, q1[q batch + i] = {radiance of
// ray q} for the bits q set in mask.  The lane traces the flagged rays one after the other, adds the radiance of the visible
// ones in the reference's order (NEE, light vertex 0, 1, ..., then the emission), clamps and adds total * contribution to the
// slot's sum (:485-496) -- what took a term cell per ray and a pass of its own (k_finish_vertex, 12 % of a round) before.
// Most vertices have the NEE ray only (a light sub-path needs its first ray to hit the scene), so everything that ray needs sits
// behind the queue index alone: one round of loads per refill, as for a plain shadow ray.  Between rays a lane keeps the sum,
// the mask and the queue index; the vertex itself is read again for the (rare) later rays and for the total.
__device__ __forceinline__ int lds_pop(const int* p) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
    return v;
}
template <bool ANY, bool COUNT, int STACK, int LDSN, bool RAYGEN = false, bool JOB = false>
__device__ __forceinline__ void trace_persistent(const DevScene& sc, const float4* __restrict__ q0, const float4* __restrict__ q1,
                                                 const float4* __restrict__ q2, const float2* __restrict__ nearfar,
                                                 float4* __restrict__ hit, float4* __restrict__ tot, uint8_t* __restrict__ vis_out,
                                                 const int mode, float* __restrict__ splat_rgb,
                                                 const uint32_t count, uint32_t* __restrict__ fetch, int* __restrict__ stack, int* __restrict__ ovf, const uint32_t ostride,
                                                 uint32_t& n_nodes, uint32_t& n_tris, const DevCamera* cam = nullptr, const PassParams* pp = nullptr,
                                                 unsigned long long* __restrict__ util = nullptr) {
    const int lane = threadIdx.x & 63;
    uint32_t u_node_it = 0, u_leaf_it = 0, u_outer_it = 0, u_refill = 0; // COUNT: wave-level iteration counts (lane occupancy per phase)
    const int stride = RGK_TRACE_BLOCK;
    const float eps = sc.epsilon;
    const int walk_q = (int)sc.walk_q;
    const float4* __restrict__ nodes = reinterpret_cast<const float4*>(sc.nodes);
    const float4* __restrict__ tris = reinterpret_cast<const float4*>(sc.tris);
    const uint32_t chunk = fetch_chunk(count, gridDim.x * (RGK_TRACE_BLOCK / 64), ANY ? 4u : 16u);
    // A wave's FIRST slice is dealt statically (wave w takes [w * chunk, (w + 1) * chunk)), the later ones through the device-wide
    // cursor, which therefore counts from n_waves * chunk.  With the cursor alone every one of a launch's ~8000 waves started with
    // a returning atomic on one word (~88 per microsecond chip-wide): ~90 us per launch whatever its work -- for the hundreds of
    // short launches of a deep path loop, most of their time.  A wave whose static slice lies beyond the queue leaves at once.
    const uint32_t n_waves = gridDim.x * (RGK_TRACE_BLOCK / 64), wave_id = blockIdx.x * (RGK_TRACE_BLOCK / 64) + (threadIdx.x >> 6);
    uint32_t w_next = min(wave_id * chunk, count), w_end = min(w_next + chunk, count); // wave-uniform: this wave's slice of the queue
    bool exhausted = false;         // wave-uniform: the device cursor has passed `count`
    // per-lane ray state
    bool active = false;
    uint32_t idx = 0, ignore = 0xffffffffu, slot = 0;
    f3 o = mk3(0.f, 0.f, 0.f), d = o, inv = o, rad = o;
    float tlo = 0.f, thi = 0.f, best_t = 0.f, best_a = 0.f, best_b = 0.f;
    int best_tri = -1, cur = STACK_SENTINEL, sp = 0;
    // JOB: the sum over the vertex's visible rays so far, the rays still to trace (mask bits 0..7; bit 8: emissive)
    f3 jsum = o;
    uint32_t jmask = 0;
    bool seg_pending = false;

#define RGK_PUT(x) { if (LDSN >= STACK || sp < LDSN) stack[sp * stride] = (x); else ovf[(size_t)(sp - LDSN) * ostride] = (x); }
// JOB: Ray(from, p, 20 eps) (src/ray.hpp:15-22) into the lane's traversal state; a ray that cannot touch the scene is visible at once
#define RGK_JOB_RAY(from_, p_)                                                                                                         \
    {                                                                                                                                  \
        o = (from_);                                                                                                                   \
        const f3 diff_ = (p_) - o;                                                                                                     \
        d = norm3(diff_);                                                                                                              \
        const float tf_ = len3(diff_) - eps * 20.0f, tn_ = 0.0f + eps * 20.0f;                                                         \
        best_t = __builtin_inff(); best_tri = -1;                                                                                      \
        sp = 0;                                                                                                                        \
        float t0_, t1_;                                                                                                                \
        const bool nan_ = (o.x != o.x) | (o.y != o.y) | (o.z != o.z) | (d.x != d.x) | (d.y != d.y) | (d.z != d.z);                     \
        if (!nan_ && clip_to_scene(sc, o, d, tn_, tf_, t0_, t1_)) {                                                                    \
            tlo = t0_ - eps; thi = t1_ + eps;                                                                                          \
            inv = mk3(fminf(fmaxf(1.f / d.x, -1e30f), 1e30f), fminf(fmaxf(1.f / d.y, -1e30f), 1e30f), fminf(fmaxf(1.f / d.z, -1e30f), 1e30f)); \
            cur = 0;                                                                                                                   \
            seg_pending = false;                                                                                                       \
        } else jsum = jsum + rad;                                                                                                      \
    }
// (The LDS part is read by an explicit ds_read_b32.  Written as a plain conditional the compiler folds the two sources into ONE
// flat_load of a selected pointer -- also with the sides forced to values, also through a volatile pointer: then a system-coherent
// flat load -- and every pop, nearly all of which come from LDS, takes the flat path: a vector-memory AND an LDS operation, waited
// for as both.  The low half of a generic pointer into LDS is its LDS address.)
#if defined(RGK_POP_FLAT)
#define RGK_POP() ((LDSN >= STACK || sp < LDSN) ? stack[sp * stride] : ovf[(size_t)(sp - LDSN) * ostride])
#else
#define RGK_POP() ((LDSN >= STACK || sp < LDSN) ? lds_pop(stack + sp * stride) : ovf[(size_t)(sp - LDSN) * ostride])
#endif
    for (;;) {
        // ------------------------------------------------ refill idle lanes
        // (JOB: a lane whose ray is done but whose vertex has more waits like an idle lane -- `act` counts the lanes with a ray in
        // flight -- and gets its next ray when the wave refills: setting a ray up means dependent loads, and doing that whenever
        // any one lane finishes a ray stalls the other 63 each time)
        const unsigned long long occ = __ballot(active);
        unsigned long long act = JOB ? __ballot(active && !seg_pending) : occ;
        const int nact = __popcll(act);
        if (COUNT) u_outer_it++;
        const bool refill_now = nact <= (JOB ? RGK_JOB_REFILL_BELOW : RAYGEN ? RGK_REFILL_BELOW_CAMERA : ANY ? RGK_REFILL_BELOW_ANY : RGK_REFILL_BELOW);
        if (refill_now && !(exhausted && w_next >= w_end)) {
            if (COUNT) u_refill++;
            if (w_next >= w_end) {
                uint32_t base = 0;
                if (n_waves * chunk >= count) base = count; // (the static slices covered the queue: no cursor traffic at all)
                else if (lane == 0) base = atomicAdd(fetch, chunk) + n_waves * chunk;
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= count) exhausted = true;
                else { w_next = base; w_end = min(base + chunk, count); }
            }
            const uint32_t avail = (w_end > w_next) ? (w_end - w_next) : 0u;
            if (avail) {
                const unsigned long long idle = ~occ;
                const uint32_t rank = __popcll(idle & ((1ull << lane) - 1ull));
                if (!active && rank < avail) {
                    idx = w_next + rank;
                    float tn = 0.0f, tf = 10000.0f; // Ray::near / Ray::far defaults, src/ray.hpp:25-26
                    uint32_t pixel_j = 0;
                    if (JOB) { // a vertex: its NEE ray at once (everything it needs sits behind idx), later rays one by one below
                        const float4 a = q0[idx], f0 = q0[3 * (size_t)pp->batch + idx], r0 = q1[idx];
                        jmask = __float_as_uint(q0[(size_t)pp->batch + idx].w);
                        jsum = mk3(0.f, 0.f, 0.f);
                        ignore = 0xffffffffu;
                        cur = STACK_SENTINEL; sp = 0;
                        seg_pending = true; active = true;
                        if (jmask & 1u) {
                            jmask &= ~1u;
                            rad = mk3(r0.x, r0.y, r0.z);
                            RGK_JOB_RAY(mk3(f0.x, f0.y, f0.z), mk3(a.x, a.y, a.z))
                        }
                    } else
                    if (RAYGEN) {
                        camera_ray_of_slot(*cam, *pp, idx, o, d, &pixel_j);
                        ignore = 0xffffffffu;
                    } else {
                        const float4 a = q0[idx], b = q1[idx];
                        o = mk3(a.x, a.y, a.z); d = mk3(a.w, b.x, b.y);
                        if (ANY) {
                            const float4 c = q2[idx];
                            rad = mk3(c.x, c.y, c.z); tn = c.w; tf = b.z;
                            slot = __float_as_uint(b.w);
                            ignore = 0xffffffffu;
                        } else {
                            ignore = __float_as_uint(b.z);
                            if (nearfar) { float2 nf = nearfar[idx]; tn = nf.x; tf = nf.y; }
                        }
                    }
                    if (!JOB) {
                    best_t = __builtin_inff(); best_tri = -1; best_a = 0.f; best_b = 0.f;
                    float t0, t1;
                    sp = 0;
                    // A ray with a NaN coordinate (a connection between coincident points normalises a zero vector) can never
                    // record a hit -- every plane solve and barycentric test on it compares false -- but its NaN slab
                    // distances drop out of the box tests, so it would walk the WHOLE tree first: on the 1 M-triangle
                    // scene a handful of such rays kept every late-bounce shadow launch alive for 5-9 ms.  Same answer, at once.
                    const bool nan_ray = (o.x != o.x) | (o.y != o.y) | (o.z != o.z) | (d.x != d.x) | (d.y != d.y) | (d.z != d.z);
                    if (!nan_ray && clip_to_scene(sc, o, d, tn, tf, t0, t1)) {
                        tlo = t0 - eps; thi = t1 + eps;
                        // 1/d for the box tests only, clamped to +-1e30: with an infinite reciprocal (a ray lying exactly in an
                        // axis-aligned wall: connections between two points of one wall) q*inf + (-inf) is NaN, the axis
                        // drops out of every slab test and the ray visits every box along its way -- thousands of nodes
                        // per ray kept the late-bounce shadow launches of the 1 M-triangle scene alive for 5-9 ms.  A huge
                        // finite reciprocal keeps the slab test exact enough (boxes are eps-padded); results never depend on it.
                        inv = mk3(fminf(fmaxf(1.f / d.x, -1e30f), 1e30f), fminf(fmaxf(1.f / d.y, -1e30f), 1e30f), fminf(fmaxf(1.f / d.z, -1e30f), 1e30f));
                        cur = 0;
                        if (ANY && pp && pp->lentry) {
                            // first-vertex shadow rays of a single-light scene start at the entry nodes of their pixel group
                            // (k_entry_points_light); `slot` is the path slot the radiance goes to
                            uint32_t srel, j;
                            slot_decode(*pp, slot, j, srel);
                            const size_t grp = (size_t)((pp->j0 + j) >> RGK_ENTRY_SHIFT);
                            const float4 blo = pp->lbox[2 * grp], bhi = pp->lbox[2 * grp + 1];
                            const f3 end = o + tf * d; // where the ray stops (20 eps short of the shaded point)
                            if (end.x >= blo.x && end.x <= bhi.x && end.y >= blo.y && end.y <= bhi.y && end.z >= blo.z && end.z <= bhi.z) {
                                const int* e = pp->lentry + grp * RGK_ENTRY_K;
                                cur = e[0];
#pragma unroll
                                for (int k = RGK_ENTRY_K - 1; k >= 1; k--) { const int r = e[k]; if (r != STACK_SENTINEL) { RGK_PUT(r) sp++; } }
                            }
                        }
                        if (RAYGEN && pp->entry) {
                            // the walk starts at the entry nodes of this pixel's group (k_entry_points): the nodes below which
                            // everything lies that ANY camera ray through the group's pixels can touch, nearest first -- up to
                            // the group's cap distance if it has one: then the ray's far end is pulled in to the cap, and a ray
                            // that finds nothing that way is traced again from the root (see where rays retire)
                            const float capd = pp->entry_cap[pixel_j >> RGK_ENTRY_SHIFT];
                            if (capd < thi) { thi = capd; ignore = 0xfffffffeu; } // (no triangle has that id: "capped, first attempt")
                            const int* e = pp->entry + (size_t)(pixel_j >> RGK_ENTRY_SHIFT) * RGK_ENTRY_K;
                            cur = e[0];
#pragma unroll
                            for (int k = RGK_ENTRY_K - 1; k >= 1; k--) { const int r = e[k]; if (r != STACK_SENTINEL) { RGK_PUT(r) sp++; } }
                        }
                    } else cur = STACK_SENTINEL; // misses the scene box: reported below as a miss
                    active = true;
                    } // !JOB
                }
                const uint32_t taken = min(avail, (uint32_t)(64 - __popcll(occ)));
                w_next += taken;
            }
            act = __ballot(active);
        }
        if (JOB && refill_now) {
            // lanes between two rays of their vertex: the next flagged ray, or -- none left -- the vertex's total into the slot sum
            while (active && seg_pending) {
                const float4 a = q0[idx]; // {p, slot}
                const uint32_t jslot = __float_as_uint(a.w);
                const uint32_t m = jmask & 0xffu;
                if (m == 0u) {
                    f3 total = jsum;
                    if (jmask & 0x100u) { const float4 e = q0[2 * (size_t)pp->batch + idx]; total = total + mk3(e.x, e.y, e.z); }
                    total = clamp3(total, pp->clamp);
                    const float4 c = q0[(size_t)pp->batch + idx];
                    const f3 add = total * mk3(c.x, c.y, c.z);
#if RGK_JOB_FINISH_ATOMIC
                    // (one vertex per slot and bounce, so each component sees ONE addition per launch -- the same bits as the
                    // read-modify-write -- but nothing to wait for at a point where the whole wave waits)
                    float* tp = reinterpret_cast<float*>(tot + jslot);
                    unsafeAtomicAdd(tp, add.x); unsafeAtomicAdd(tp + 1, add.y); unsafeAtomicAdd(tp + 2, add.z);
#else
                    float4 t = tot[jslot]; // one vertex per slot and bounce: no race
                    t.x = t.x + add.x; t.y = t.y + add.y; t.z = t.z + add.z;
                    tot[jslot] = t;
#endif
                    active = false; seg_pending = false;
                    break;
                }
                const uint32_t q = (uint32_t)__builtin_ctz(m);
                jmask &= ~(1u << q);
                const float4 r = q1[(size_t)q * pp->batch + idx];
                rad = mk3(r.x, r.y, r.z);
                const float4 fr = (q == 0u) ? q0[3 * (size_t)pp->batch + idx] : pp->lv[(size_t)((q - 1u) * RGK_LV_FLOAT4) * pp->batch + jslot];
                RGK_JOB_RAY(mk3(fr.x, fr.y, fr.z), mk3(a.x, a.y, a.z))
            }
            act = __ballot(active);
        }
        if (act == 0) {
            if (exhausted && w_next >= w_end) break;
            continue;
        }
        // ------------------------------------------------ inner nodes, while the lanes standing at an inner node have
        // the majority (x walk_q / 4) over the lanes waiting at a leaf.  Round-1 counters: the classic while-while
        // form (walk until every lane sits at a leaf, then run every leaf to its end) left half the lanes of each
        // VALU instruction idle; leaving the walk once the leaves have the majority cut 10 % off the kernel.
        // (One scheduled step per outer iteration instead of two tight loops was slower: 25.3 vs 23.3 ms.)
        // (Postponing the first leaf and walking on -- "speculative traversal" -- was measured: 8 % more
        // nodes, 30 % more triangle tests, 20 % slower.  Not kept.)
        for (;;) {
            // (a lane without a ray always holds cur == SENTINEL, so `cur` alone tells the three states apart; the ballots are
            // compares straight into a lane mask)
            const bool walking = (uint32_t)cur < (uint32_t)STACK_SENTINEL;
            const int n_walk = __popcll(__builtin_amdgcn_ballot_w64(walking));
            if (n_walk == 0) break;
            if (n_walk * 4 < __popcll(__builtin_amdgcn_ballot_w64(cur < 0)) * walk_q) break; // the leaves have the majority
            if (COUNT) u_node_it++;
            if (!walking) continue;
            float te[4];
            int ref[4];
            if (COUNT) n_nodes++;
#if defined(RGK_DIAG_NODE_TWICE) /* timing diagnosis only: the node's loads and box tests twice */
            for (int rep_ = 0; rep_ < 2; rep_++) { asm volatile("" : "+v"(cur));
#endif
            const float limit = ANY ? thi : fminf(thi, best_t);
            // one 64-byte QNode: {p.xyz, sx} {child[4]} {qlo.x qlo.y qlo.z qhi.x} {qhi.y qhi.z sy sz}
            // (the SGPR-base + 32-bit-offset load form of rgk_device.h gld_* was measured here: 23.8 vs 23.0 ms, not kept)
            const float4 n0 = nodes[4 * cur + 0], n1 = nodes[4 * cur + 1], n2 = nodes[4 * cur + 2], n3 = nodes[4 * cur + 3];
            const float sx = n0.w, sy = n3.z, sz = n3.w; // per-axis quantisation step (a power of two)
            // plane distance along the ray: t = (p + q*s - o) / d = q * (s/d) + (p - o)/d, one fma per plane.
            // The split loses a few ulps of |(p-o)/d| to cancellation; the child boxes carry an absolute
            // pad of eps = 1e-5 * scene diagonal, two orders above float resolution at scene scale, so the
            // test stays conservative.  (Box tests only steer the walk: they never produce a result.)
            const float kx = sx * inv.x, ky = sy * inv.y, kz = sz * inv.z;
            const float cx = (n0.x - o.x) * inv.x, cy = (n0.y - o.y) * inv.y, cz = (n0.z - o.z) * inv.z;
            // near / far plane words by the sign of 1/d: no per-plane min/max needed; an unused slot
            // (qlo = 255, qhi = 0) then has near > far on every axis and can never be entered
            const bool ngx = (__float_as_uint(inv.x) >> 31) != 0, ngy = (__float_as_uint(inv.y) >> 31) != 0, ngz = (__float_as_uint(inv.z) >> 31) != 0;
            const uint32_t lx = __float_as_uint(n2.x), ly = __float_as_uint(n2.y), lz = __float_as_uint(n2.z);
            const uint32_t hx = __float_as_uint(n2.w), hy = __float_as_uint(n3.x), hz = __float_as_uint(n3.y);
            const uint32_t nx = ngx ? hx : lx, fx = ngx ? lx : hx;
            const uint32_t ny = ngy ? hy : ly, fy = ngy ? ly : hy;
            const uint32_t nz = ngz ? hz : lz, fz = ngz ? lz : hz;
            ref[0] = __float_as_int(n1.x); ref[1] = __float_as_int(n1.y); ref[2] = __float_as_int(n1.z); ref[3] = __float_as_int(n1.w);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                // NaNs (0 * inf on an axis-parallel ray) drop out of fmax/fmin, which only widens the
                // interval (conservative).
                const float t0x = __builtin_fmaf(cvt_ubyte(nx, c), kx, cx), t1x = __builtin_fmaf(cvt_ubyte(fx, c), kx, cx);
                const float t0y = __builtin_fmaf(cvt_ubyte(ny, c), ky, cy), t1y = __builtin_fmaf(cvt_ubyte(fy, c), ky, cy);
                const float t0z = __builtin_fmaf(cvt_ubyte(nz, c), kz, cz), t1z = __builtin_fmaf(cvt_ubyte(fz, c), kz, cz);
                const float tn = fmaxf(fmaxf(fmaxf(t0x, t0y), t0z), tlo);
                const float tf = fminf(fminf(fminf(t1x, t1y), t1z), limit);
                const bool h = tn <= tf;
                te[c] = h ? tn : __builtin_inff();
                if (!h) ref[c] = STACK_SENTINEL;
            }
#if defined(RGK_DIAG_NODE_TWICE)
            asm volatile("" :: "v"(te[0]), "v"(te[1]), "v"(te[2]), "v"(te[3]), "v"(ref[0]), "v"(ref[1]), "v"(ref[2]), "v"(ref[3])); }
#endif
            if (!ANY) {
                // sort the four (entry distance, ref) pairs ascending: 5-comparator network
#define RGK_CSWAP(a, b) { const bool sw = te[b] < te[a]; const float tt = sw ? te[b] : te[a], tu = sw ? te[a] : te[b]; \
                          const int rr = sw ? ref[b] : ref[a], ru = sw ? ref[a] : ref[b]; te[a] = tt; te[b] = tu; ref[a] = rr; ref[b] = ru; }
                RGK_CSWAP(0, 1) RGK_CSWAP(2, 3) RGK_CSWAP(0, 2) RGK_CSWAP(1, 3) RGK_CSWAP(1, 2)
#undef RGK_CSWAP
                // misses carry te = inf / ref = SENTINEL and sort to the back; push far -> near.
                // Branch-free: always write the next free entry, advance only for a real child (the host
                // sized STACK above the deepest push sequence the tree can produce, rgk_host.cpp QbvhBuilder).
                if (LDSN >= STACK || sp + 3 <= LDSN) { // all three possible entries fit the LDS part: one test instead of three
                    stack[sp * stride] = ref[3]; sp += (ref[3] != STACK_SENTINEL);
                    stack[sp * stride] = ref[2]; sp += (ref[2] != STACK_SENTINEL);
                    stack[sp * stride] = ref[1]; sp += (ref[1] != STACK_SENTINEL);
                } else {
                    RGK_PUT(ref[3]) sp += (ref[3] != STACK_SENTINEL);
                    RGK_PUT(ref[2]) sp += (ref[2] != STACK_SENTINEL);
                    RGK_PUT(ref[1]) sp += (ref[1] != STACK_SENTINEL);
                }
                cur = ref[0];
            } else {
                // any-hit: the answer does not depend on the order, the time does -- walking on with the NEAREST entered child
                // finds an occluder sooner than taking the children in slot order (shadow launches -16 %; a full sort: same)
                int bi = 0;
                float bt = te[0];
                if (te[1] < bt) { bt = te[1]; bi = 1; }
                if (te[2] < bt) { bt = te[2]; bi = 2; }
                if (te[3] < bt) { bt = te[3]; bi = 3; }
                cur = bi == 0 ? ref[0] : (bi == 1 ? ref[1] : (bi == 2 ? ref[2] : ref[3])); // SENTINEL when no child is entered
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (c != bi && ref[c] != STACK_SENTINEL && sp < STACK) { RGK_PUT(ref[c]) sp++; }
            }
            if (cur == STACK_SENTINEL && sp > 0) { sp--; cur = RGK_POP(); }
        }
        // ------------------------------------------------ leaf: every triangle of it.  (One triangle per scheduled step,
        // leaving when the walkers regain the majority, was measured slower: 24.2 vs 23.3 ms.)
        if (COUNT) { // wave iterations of the triangle loop below = the longest leaf among the lanes standing at one
            const uint32_t mine = (active && cur < 0) ? ((~(uint32_t)cur) & 15u) + 1u : 0u;
            for (uint32_t c = 1; c <= 16; c++) if (__ballot(mine >= c)) u_leaf_it++;
        }
        if (active && cur < 0) {
            const uint32_t code = ~(uint32_t)cur;
            const uint32_t first = code >> 4, cnt = (code & 15u) + 1u;
            bool done = false;
            for (uint32_t k = 0; k < cnt && !done; k++) {
                const float4 r0 = tris[3 * (first + k) + 0], r1 = tris[3 * (first + k) + 1], r2 = tris[3 * (first + k) + 2];
                const uint32_t tid = __float_as_uint(r2.w);
                if (tid == ignore) continue;
                if (COUNT) n_tris++;
                float t, al, be;
                bool hit_ = tri_test(r0, r1, r2, o, d, eps, t, al, be);
#if defined(RGK_DIAG_TRI_TWICE) /* timing diagnosis only: the same test a second time (the optimiser cannot see that it is the same) */
                { f3 o2 = o; asm volatile("" : "+v"(o2.x)); float t2, a2, b2; const bool h2 = tri_test(r0, r1, r2, o2, d, eps, t2, a2, b2); if (h2 != hit_) { t = t2; hit_ = h2; } }
#endif
                if (hit_) {
                    if (t < tlo || t > thi) continue;
                    // Exact ties go to the HIGHER triangle id, so that the result does not depend on the order in which this walker
                    // happens to reach the leaves (it changes with every change to the tree builder).  The reference takes the first
                    // triangle of its kd leaf's list (strict <, src/scene_intersect.cpp:272-284), and that list is in the order of
                    // an unstable sort of box events (src/scene.cpp:459-468): no rule of its own.  Coplanar overlapping surfaces do
                    // tie -- the Cornell lights lie IN the ceiling -- and there "higher id" is what the reference's build yields
                    // (300 k random rays, four different trees: test_closest_hit_cornell_bit_exact).
                    if (t < best_t || (!ANY && t == best_t && tid > (uint32_t)best_tri && best_tri >= 0)) { best_t = t; best_tri = (int)tid; best_a = al; best_b = be; if (ANY) done = true; }
                }
            }
            if (ANY && done) { cur = STACK_SENTINEL; sp = 0; }
            else if (sp > 0) { sp--; cur = RGK_POP(); }
            else cur = STACK_SENTINEL;
        }
        // ------------------------------------------------ retire finished rays
        if (RAYGEN && active && cur == STACK_SENTINEL && ignore == 0xfffffffeu && best_tri < 0) {
            // nothing within the group's cap: this ray goes where the frame's earlier rays of its group did not -- again, from the root
            float t0, t1;
            (void)clip_to_scene(sc, o, d, 0.0f, 10000.0f, t0, t1);
            thi = t1 + eps;
            ignore = 0xffffffffu;
            cur = 0; sp = 0;
        }
        if (JOB) {
            if (active && !seg_pending && cur == STACK_SENTINEL) { // this ray is done: the vertex's next one when the wave refills
                if (best_tri < 0) jsum = jsum + rad;
                seg_pending = true;
            }
        } else
        if (active && cur == STACK_SENTINEL) {
            if (!ANY) hit[idx] = make_float4(best_t, best_a, best_b, __int_as_float(best_tri));
            else if (vis_out) vis_out[idx] = best_tri < 0;
            else if (best_tri < 0) {
                if (mode == RGK_SHADOW_ADD) {
                    float4 t = tot[slot]; // one path per slot, one shadow ray per path and bounce: no race
                    t.x = t.x + rad.x; t.y = t.y + rad.y; t.z = t.z + rad.z;
                    tot[slot] = t;
                } else { // RGK_SHADOW_SPLAT: light-tracing side effect, AddPixel(x2, y2, r, 0) tracer.cpp:20-26; `slot` = pixel
                    atomicAdd(&splat_rgb[3 * (size_t)slot + 0], rad.x);
                    atomicAdd(&splat_rgb[3 * (size_t)slot + 1], rad.y);
                    atomicAdd(&splat_rgb[3 * (size_t)slot + 2], rad.z);
                }
            }
            active = false;
        }
    }
    if (COUNT && util && lane == 0) {
        atomicAdd(&util[0], (unsigned long long)u_node_it); atomicAdd(&util[1], (unsigned long long)u_leaf_it);
        atomicAdd(&util[2], (unsigned long long)u_outer_it); atomicAdd(&util[3], (unsigned long long)u_refill);
    }
}

// ------------------------------------------------------------------ K2: closest hit
template <bool COUNT, int STACK, int LDSN>
__global__ __launch_bounds__(RGK_TRACE_BLOCK, (LDSN <= 16 ? RGK_TRACE_WAVES : 5)) void k_trace_closest(const DevScene sc, const float4* __restrict__ rayA,
                                                                    const float4* __restrict__ rayB, const float2* __restrict__ nearfar,
                                                                    float4* __restrict__ hit, const uint32_t* __restrict__ count_ptr,
                                                                    uint32_t* __restrict__ fetch, unsigned long long* __restrict__ stats, int* __restrict__ ovf) {
    __shared__ int lds_stack[LDSN * RGK_TRACE_BLOCK];
    uint32_t n_nodes = 0, n_tris = 0;
    trace_persistent<false, COUNT, STACK, LDSN>(sc, rayA, rayB, nullptr, nearfar, hit, nullptr, nullptr, 0, nullptr, *count_ptr, fetch,
                                          lds_stack + threadIdx.x, ovf + (blockIdx.x * RGK_TRACE_BLOCK + threadIdx.x), gridDim.x * RGK_TRACE_BLOCK, n_nodes, n_tris, nullptr, nullptr, stats + 4);
    if (COUNT) {
        atomicAdd(&stats[0], (unsigned long long)n_nodes);
        atomicAdd(&stats[1], (unsigned long long)n_tris);
    }
}

// K1 + K2 at bounce 0: the camera rays of a pass, generated where they are traced (no ray queue; hit[slot] out)
template <bool COUNT, int STACK, int LDSN>
__global__ __launch_bounds__(RGK_TRACE_BLOCK, (LDSN <= 16 ? RGK_TRACE_WAVES : 5)) void k_trace_camera(const DevScene sc, const DevCamera cam, const PassParams pp,
                                                                    float4* __restrict__ hit, const uint32_t* __restrict__ count_ptr,
                                                                    uint32_t* __restrict__ fetch, unsigned long long* __restrict__ stats, int* __restrict__ ovf) {
    __shared__ int lds_stack[LDSN * RGK_TRACE_BLOCK];
    uint32_t n_nodes = 0, n_tris = 0;
    trace_persistent<false, COUNT, STACK, LDSN, true>(sc, nullptr, nullptr, nullptr, nullptr, hit, nullptr, nullptr, 0, nullptr, *count_ptr, fetch,
                                          lds_stack + threadIdx.x, ovf + (blockIdx.x * RGK_TRACE_BLOCK + threadIdx.x), gridDim.x * RGK_TRACE_BLOCK, n_nodes, n_tris, &cam, &pp, stats + 4);
    if (COUNT) {
        atomicAdd(&stats[0], (unsigned long long)n_nodes);
        atomicAdd(&stats[1], (unsigned long long)n_tris);
    }
}

// ------------------------------------------------------------------ K1 + K2 at bounce 0, one lane per PIXEL: the beam walk
// The 8 samples of a pixel that sit side by side in the slot order (PassParams::gshift = 3) are rays from ONE origin (a pinhole
// camera) through one pixel: they walk the same nodes.  Here one lane walks the tree ONCE for the eight of them -- a node is
// entered when the bundle can enter it: per axis the rays' reciprocal directions as an interval, the slab distances as interval
// products (conservative: widened by a few ulps, boxes are epsilon-padded on top) -- and at a leaf every triangle is tested
// against each of the 8 rays with the same tri_test, the same acceptance window (its own clip to the scene box) and the same tie
// rule as k_trace_camera.  A ray's result is the nearest accepted hit among the triangles of every leaf ITS OWN walk would have
// reached (the bundle reaches a superset, and a hit beyond the ray's own pruning distance loses to the nearer one anyway): the
// same hit, bit for bit.  Node work per ray drops eight-fold; the 64 lanes of a wave are an 8 x 8 pixel block.
// Capped entry lists (k_entry_points): rays that find nothing within the group's cap are walked again, together, from the root.
#ifndef RGK_BEAM_WAVES
#define RGK_BEAM_WAVES 4
#endif
template <bool COUNT, int STACK, int LDSN>
__global__ __launch_bounds__(RGK_TRACE_BLOCK, RGK_BEAM_WAVES) void k_trace_camera_beam(const DevScene sc, const DevCamera cam, const PassParams pp,
                                                                    float4* hit, const uint32_t* __restrict__ count_ptr,
                                                                    uint32_t* __restrict__ fetch, unsigned long long* __restrict__ stats, int* __restrict__ ovf_base) {
    __shared__ int lds_stack[LDSN * RGK_TRACE_BLOCK];
    __shared__ float lds_ray[32 * RGK_TRACE_BLOCK]; // per lane, [entry][lane]: the 8 directions (24 floats) and the 8 nearest accepted distances
    int* __restrict__ stack = lds_stack + threadIdx.x;
    float* __restrict__ ray = lds_ray + threadIdx.x;
#define RGK_D(k_) mk3(ray[(3 * (k_) + 0) * RGK_TRACE_BLOCK], ray[(3 * (k_) + 1) * RGK_TRACE_BLOCK], ray[(3 * (k_) + 2) * RGK_TRACE_BLOCK])
#define RGK_BEST(k_) ray[(24 + (k_)) * RGK_TRACE_BLOCK]
    int* __restrict__ ovf = ovf_base + (blockIdx.x * RGK_TRACE_BLOCK + threadIdx.x);
    const uint32_t ostride = gridDim.x * RGK_TRACE_BLOCK;
    const int stride = RGK_TRACE_BLOCK;
    const int lane = threadIdx.x & 63;
    const float eps = sc.epsilon;
    const int walk_q = (int)sc.walk_q;
    const float4* __restrict__ nodes = reinterpret_cast<const float4*>(sc.nodes);
    const float4* __restrict__ tris = reinterpret_cast<const float4*>(sc.tris);
    const SamplerTab tb = {pp.htab, pp.multisample};
    const uint32_t count = (*count_ptr) >> 3; // bundles: 8 slots each
    const uint32_t chunk = fetch_chunk(count, gridDim.x * (RGK_TRACE_BLOCK / 64), 16u);
    const uint32_t n_waves = gridDim.x * (RGK_TRACE_BLOCK / 64), wave_id = blockIdx.x * (RGK_TRACE_BLOCK / 64) + (threadIdx.x >> 6);
    uint32_t w_next = min(wave_id * chunk, count), w_end = min(w_next + chunk, count); // first slice dealt statically (see trace_persistent)
    bool exhausted = false;
    uint32_t n_nodes = 0, n_tris = 0;
    // per-lane bundle state
    bool active = false, capped = false;
    uint32_t idx = 0, valid = 0; // valid: bit k = sample k still looks for its hit in this walk
    f3 o = mk3(0.f, 0.f, 0.f), imin = o, imax = o;
    // (a sample's whole hit record {t, alpha, beta, triangle} lives in hit[] from the start: written as "miss" when the bundle is
    // set up, overwritten whenever a nearer hit is accepted; directions and nearest distances sit in LDS, indexed by the sample)
    uint32_t axis_mode = 0; // 2 bits per axis: 0 all directions positive, 1 all negative, 2 mixed (the axis cannot exclude a box)
    float bmax = 0.f, btlo = 0.f, capd = 0.f;
    int cur = STACK_SENTINEL, sp = 0;

    // the bundle's direction intervals and pruning bounds over the samples in `valid`
#define RGK_BEAM_BOUNDS()                                                                                                              \
    {                                                                                                                                  \
        f3 dmin = mk3(__builtin_inff(), __builtin_inff(), __builtin_inff()), dmax = -dmin;                                             \
        btlo = __builtin_inff();                                                                                                       \
        for (int k = 0; k < 8; k++) if (valid & (1u << k)) {                                                                          \
            const f3 dk_ = RGK_D(k);                                                                                                   \
            dmin = mk3(fminf(dmin.x, dk_.x), fminf(dmin.y, dk_.y), fminf(dmin.z, dk_.z));                                              \
            dmax = mk3(fmaxf(dmax.x, dk_.x), fmaxf(dmax.y, dk_.y), fmaxf(dmax.z, dk_.z));                                              \
            float t0_, t1_; (void)clip_to_scene(sc, o, dk_, 0.0f, 10000.0f, t0_, t1_); btlo = fminf(btlo, t0_ - eps);                  \
        }                                                                                                                              \
        axis_mode = 0;                                                                                                                 \
        float lo_[3] = {dmin.x, dmin.y, dmin.z}, hi_[3] = {dmax.x, dmax.y, dmax.z}, im_[3], ix_[3];                                    \
        _Pragma("unroll") for (int a = 0; a < 3; a++) {                                                                                \
            const uint32_t m = lo_[a] > 0.f ? 0u : (hi_[a] < 0.f ? 1u : 2u);                                                           \
            axis_mode |= m << (2 * a);                                                                                                 \
            float a0 = 1.f / hi_[a], a1 = 1.f / lo_[a]; /* 1/d over [lo, hi] of one sign: [1/hi, 1/lo] */                              \
            a0 -= fabsf(a0) * 4e-7f; a1 += fabsf(a1) * 4e-7f;                                                                          \
            im_[a] = m == 2u ? 0.f : fminf(fmaxf(a0, -1e30f), 1e30f); ix_[a] = m == 2u ? 0.f : fminf(fmaxf(a1, -1e30f), 1e30f);       \
        }                                                                                                                              \
        imin = mk3(im_[0], im_[1], im_[2]); imax = mk3(ix_[0], ix_[1], ix_[2]);                                                        \
    }

    for (;;) {
        // ------------------------------------------------ refill idle lanes with fresh bundles
        unsigned long long act = __ballot(active);
        const int nact = __popcll(act);
        if (nact <= RGK_REFILL_BELOW && !(exhausted && w_next >= w_end)) {
            if (w_next >= w_end) {
                uint32_t base = 0;
                if (n_waves * chunk >= count) base = count;
                else if (lane == 0) base = atomicAdd(fetch, chunk) + n_waves * chunk;
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= count) exhausted = true;
                else { w_next = base; w_end = min(base + chunk, count); }
            }
            const uint32_t avail = (w_end > w_next) ? (w_end - w_next) : 0u;
            if (avail) {
                const uint32_t rank = __popcll(~act & ((1ull << lane) - 1ull));
                if (!active && rank < avail) {
                    idx = w_next + rank;
                    const uint32_t sb = idx / pp.npix, j = idx - sb * pp.npix; // slot = idx << 3 | k  (slot_decode with gshift = 3)
                    const uint32_t pix = pp.pix_xy[pp.j0 + j], seed = pp.pix_seed[pp.j0 + j];
                    valid = 0;
                    for (int k = 0; k < 8; k++) {
                        const float2 jit = sample2d_t(tb, seed, pp.s0 + (sb << 3) + (uint32_t)k, 0);
                        f3 dk;
                        camera_ray(cam, (int)(pix & 0xffff), (int)(pix >> 16), (int)pp.xres, (int)pp.yres, jit, make_float2(0.f, 0.f), o, dk);
                        ray[(3 * k + 0) * RGK_TRACE_BLOCK] = dk.x; ray[(3 * k + 1) * RGK_TRACE_BLOCK] = dk.y; ray[(3 * k + 2) * RGK_TRACE_BLOCK] = dk.z;
                        RGK_BEST(k) = __builtin_inff();
                        hit[((size_t)idx << 3) + k] = make_float4(__builtin_inff(), 0.f, 0.f, __int_as_float(-1));
                        const bool nan_ray = (o.x != o.x) | (o.y != o.y) | (o.z != o.z) | (dk.x != dk.x) | (dk.y != dk.y) | (dk.z != dk.z);
                        float t0, t1;
                        if (!nan_ray && clip_to_scene(sc, o, dk, 0.0f, 10000.0f, t0, t1)) valid |= 1u << k;
                    }
                    sp = 0;
                    cur = STACK_SENTINEL;
                    capped = false;
                    capd = __builtin_inff();
                    bmax = __builtin_inff();
                    if (valid) {
                        RGK_BEAM_BOUNDS()
                        cur = 0;
                        if (pp.entry) {
                            const size_t grp = (size_t)((pp.j0 + j) >> RGK_ENTRY_SHIFT);
                            capd = pp.entry_cap[grp];
                            capped = capd < __builtin_inff();
                            const int* e = pp.entry + grp * RGK_ENTRY_K;
                            cur = e[0];
#pragma unroll
                            for (int k = RGK_ENTRY_K - 1; k >= 1; k--) { const int r = e[k]; if (r != STACK_SENTINEL) { RGK_PUT(r) sp++; } }
                        }
                    }
                    active = true;
                }
                w_next += min(avail, (uint32_t)(64 - nact));
            }
            act = __ballot(active);
        }
        if (act == 0) {
            if (exhausted && w_next >= w_end) break;
            continue;
        }
        // ------------------------------------------------ inner nodes (same scheduling as trace_persistent)
        for (;;) {
            const bool walking = (uint32_t)cur < (uint32_t)STACK_SENTINEL;
            const int n_walk = __popcll(__builtin_amdgcn_ballot_w64(walking));
            if (n_walk == 0) break;
            if (n_walk * 4 < __popcll(__builtin_amdgcn_ballot_w64(cur < 0)) * walk_q) break;
            if (!walking) continue;
            if (COUNT) n_nodes++;
            const float limit = fminf(bmax, capd);
            const float4 n0 = nodes[4 * cur + 0], n1 = nodes[4 * cur + 1], n2 = nodes[4 * cur + 2], n3 = nodes[4 * cur + 3];
            const float sx = n0.w, sy = n3.z, sz = n3.w;
            const float px = n0.x - o.x, py = n0.y - o.y, pz = n0.z - o.z;
            const uint32_t mx = axis_mode & 3u, my = (axis_mode >> 2) & 3u, mz = (axis_mode >> 4) & 3u;
            const uint32_t lx = __float_as_uint(n2.x), ly = __float_as_uint(n2.y), lz = __float_as_uint(n2.z);
            const uint32_t hx = __float_as_uint(n2.w), hy = __float_as_uint(n3.x), hz = __float_as_uint(n3.y);
            const uint32_t nx = mx == 1u ? hx : lx, fx = mx == 1u ? lx : hx; // near / far plane words by the bundle's direction sign
            const uint32_t ny = my == 1u ? hy : ly, fy = my == 1u ? ly : hy;
            const uint32_t nz = mz == 1u ? hz : lz, fz = mz == 1u ? lz : hz;
            float te[4];
            int ref[4];
            ref[0] = __float_as_int(n1.x); ref[1] = __float_as_int(n1.y); ref[2] = __float_as_int(n1.z); ref[3] = __float_as_int(n1.w);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                // plane - origin, then its distance over the bundle: the interval product (plane - o) * [imin, imax]
                const float rnx = __builtin_fmaf(cvt_ubyte(nx, c), sx, px), rfx = __builtin_fmaf(cvt_ubyte(fx, c), sx, px);
                const float rny = __builtin_fmaf(cvt_ubyte(ny, c), sy, py), rfy = __builtin_fmaf(cvt_ubyte(fy, c), sy, py);
                const float rnz = __builtin_fmaf(cvt_ubyte(nz, c), sz, pz), rfz = __builtin_fmaf(cvt_ubyte(fz, c), sz, pz);
                float t0x = fminf(rnx * imin.x, rnx * imax.x), t1x = fmaxf(rfx * imin.x, rfx * imax.x);
                float t0y = fminf(rny * imin.y, rny * imax.y), t1y = fmaxf(rfy * imin.y, rfy * imax.y);
                float t0z = fminf(rnz * imin.z, rnz * imax.z), t1z = fmaxf(rfz * imin.z, rfz * imax.z);
                if (mx == 2u) { t0x = -__builtin_inff(); t1x = __builtin_inff(); }
                if (my == 2u) { t0y = -__builtin_inff(); t1y = __builtin_inff(); }
                if (mz == 2u) { t0z = -__builtin_inff(); t1z = __builtin_inff(); }
                const float tn = fmaxf(fmaxf(fmaxf(t0x, t0y), t0z), btlo);
                const float tf = fminf(fminf(fminf(t1x, t1y), t1z), limit);
                const bool h = (ref[c] != STACK_SENTINEL) && (tn <= tf); // (an unused slot carries the sentinel as its child code)
                te[c] = h ? tn : __builtin_inff();
                if (!h) ref[c] = STACK_SENTINEL;
            }
#define RGK_CSWAP(a, b) { const bool sw = te[b] < te[a]; const float tt = sw ? te[b] : te[a], tu = sw ? te[a] : te[b]; \
                          const int rr = sw ? ref[b] : ref[a], ru = sw ? ref[a] : ref[b]; te[a] = tt; te[b] = tu; ref[a] = rr; ref[b] = ru; }
            RGK_CSWAP(0, 1) RGK_CSWAP(2, 3) RGK_CSWAP(0, 2) RGK_CSWAP(1, 3) RGK_CSWAP(1, 2)
#undef RGK_CSWAP
            RGK_PUT(ref[3]) sp += (ref[3] != STACK_SENTINEL);
            RGK_PUT(ref[2]) sp += (ref[2] != STACK_SENTINEL);
            RGK_PUT(ref[1]) sp += (ref[1] != STACK_SENTINEL);
            cur = ref[0];
            if (cur == STACK_SENTINEL && sp > 0) { sp--; cur = RGK_POP(); }
        }
        // ------------------------------------------------ leaf: every triangle of it against every sample of the bundle
        if (active && cur < 0) {
            const uint32_t code = ~(uint32_t)cur;
            const uint32_t first = code >> 4, cnt = (code & 15u) + 1u;
            for (uint32_t k = 0; k < cnt; k++) {
                const float4 r0 = tris[3 * (first + k) + 0], r1 = tris[3 * (first + k) + 1], r2 = tris[3 * (first + k) + 2];
                const int tid = (int)__float_as_uint(r2.w);
                if (COUNT) n_tris += (uint32_t)__popc(valid);
                for (int q = 0; q < 8; q++) {
                    if (!(valid & (1u << q))) continue;
                    const f3 dq = RGK_D(q);
                    float t, al, be;
                    if (!tri_test(r0, r1, r2, o, dq, eps, t, al, be)) continue;
                    // the ray's own acceptance window: its [near, far] clipped to the scene box (and to the group's cap)
                    float t0, t1;
                    (void)clip_to_scene(sc, o, dq, 0.0f, 10000.0f, t0, t1);
                    const float thi = capped ? fminf(t1 + eps, capd) : t1 + eps;
                    if (t < t0 - eps || t > thi) continue;
                    const float bq = RGK_BEST(q);
                    bool take = t < bq;
                    if (t == bq) take = tid > __float_as_int(hit[((size_t)idx << 3) + q].w); // an exact tie: the higher triangle id (its own earlier write, same lane)
                    if (take) { RGK_BEST(q) = t; hit[((size_t)idx << 3) + q] = make_float4(t, al, be, __int_as_float(tid)); }
                }
            }
            bmax = 0.f; // nothing beyond the farthest sample's nearest hit can matter to any of them
            for (int q = 0; q < 8; q++) if (valid & (1u << q)) bmax = fmaxf(bmax, RGK_BEST(q));
            if (sp > 0) { sp--; cur = RGK_POP(); }
            else cur = STACK_SENTINEL;
        }
        // ------------------------------------------------ retire: walk again from the root for samples the capped list left empty-handed
        if (active && cur == STACK_SENTINEL) {
            uint32_t again = 0;
            if (capped) {
                for (int q = 0; q < 8; q++) if ((valid & (1u << q)) && RGK_BEST(q) == __builtin_inff()) again |= 1u << q;
            }
            if (again) {
                valid = again; capped = false; capd = __builtin_inff(); bmax = __builtin_inff();
                RGK_BEAM_BOUNDS()
                cur = 0; sp = 0;
            } else active = false; // (the records are in hit[] already)
        }
    }
#undef RGK_BEAM_BOUNDS
#undef RGK_D
#undef RGK_BEST
    if (COUNT) {
        atomicAdd(&stats[0], (unsigned long long)n_nodes);
        atomicAdd(&stats[1], (unsigned long long)n_tris);
    }
}

// ------------------------------------------------------------------ K5: shadow rays + accumulate
// shA = (o.xyz, d.x)  shB = (d.y, d.z, far, slot)  shC = (radiance.rgb, near)
template <bool COUNT, int STACK, int LDSN>
__global__ __launch_bounds__(RGK_TRACE_BLOCK, (LDSN <= 16 ? RGK_TRACE_WAVES : 5)) void k_trace_shadow(const DevScene sc, const float4* __restrict__ shA,
                                                                   const float4* __restrict__ shB, const float4* __restrict__ shC,
                                                                   float4* __restrict__ tot, uint8_t* __restrict__ vis_out,
                                                                   const int mode, float* __restrict__ splat_rgb,
                                                                   const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ fetch,
                                                                   unsigned long long* __restrict__ stats, int* __restrict__ ovf) {
    __shared__ int lds_stack[LDSN * RGK_TRACE_BLOCK];
    uint32_t n_nodes = 0, n_tris = 0;
    trace_persistent<true, COUNT, STACK, LDSN>(sc, shA, shB, shC, nullptr, nullptr, tot, vis_out, mode, splat_rgb, *count_ptr, fetch,
                                         lds_stack + threadIdx.x, ovf + (blockIdx.x * RGK_TRACE_BLOCK + threadIdx.x), gridDim.x * RGK_TRACE_BLOCK, n_nodes, n_tris);
    if (COUNT) {
        atomicAdd(&stats[2], (unsigned long long)n_nodes);
        atomicAdd(&stats[3], (unsigned long long)n_tris);
    }
}

// K5 for bidirectional rounds: one queue entry per camera-path vertex (JOB above).  jobs = {vertex}{contribution, mask}{emission},
// rads = one radiance per ray (k_connect wrote both, and counted the rays).
#ifndef RGK_JOB_FINISH_ATOMIC
#define RGK_JOB_FINISH_ATOMIC 0
#endif
#ifndef RGK_JOB_WAVES
#define RGK_JOB_WAVES 6 // the vertex state costs ~12 VGPRs over a plain shadow ray: 8 waves per SIMD would spill 44 bytes
#endif
template <bool COUNT, int STACK, int LDSN>
__global__ __launch_bounds__(RGK_TRACE_BLOCK, (LDSN <= 16 ? RGK_JOB_WAVES : 5)) void k_trace_shadow_jobs(const DevScene sc, const PassParams pp, const float4* __restrict__ jobs,
                                                                   const float4* __restrict__ rads, float4* __restrict__ tot,
                                                                   const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ fetch,
                                                                   unsigned long long* __restrict__ stats, int* __restrict__ ovf) {
    __shared__ int lds_stack[LDSN * RGK_TRACE_BLOCK];
    uint32_t n_nodes = 0, n_tris = 0;
    trace_persistent<true, COUNT, STACK, LDSN, false, true>(sc, jobs, rads, nullptr, nullptr, nullptr, tot, nullptr, RGK_SHADOW_ADD, nullptr, *count_ptr, fetch,
                                         lds_stack + threadIdx.x, ovf + (blockIdx.x * RGK_TRACE_BLOCK + threadIdx.x), gridDim.x * RGK_TRACE_BLOCK, n_nodes, n_tris, nullptr, &pp);
    if (COUNT) {
        atomicAdd(&stats[2], (unsigned long long)n_nodes);
        atomicAdd(&stats[3], (unsigned long long)n_tris);
    }
}

// K5 for the first vertex of a pass in a single-light scene: the same walker, started at the pixel group's light-side entry nodes
template <bool COUNT, int STACK, int LDSN>
__global__ __launch_bounds__(RGK_TRACE_BLOCK, (LDSN <= 16 ? RGK_TRACE_WAVES : 5)) void k_trace_shadow_first(const DevScene sc, const PassParams pp, const float4* __restrict__ shA,
                                                                   const float4* __restrict__ shB, const float4* __restrict__ shC,
                                                                   float4* __restrict__ tot, const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ fetch,
                                                                   unsigned long long* __restrict__ stats, int* __restrict__ ovf) {
    __shared__ int lds_stack[LDSN * RGK_TRACE_BLOCK];
    uint32_t n_nodes = 0, n_tris = 0;
    trace_persistent<true, COUNT, STACK, LDSN>(sc, shA, shB, shC, nullptr, nullptr, tot, nullptr, RGK_SHADOW_ADD, nullptr, *count_ptr, fetch,
                                         lds_stack + threadIdx.x, ovf + (blockIdx.x * RGK_TRACE_BLOCK + threadIdx.x), gridDim.x * RGK_TRACE_BLOCK, n_nodes, n_tris, nullptr, &pp);
    if (COUNT) {
        atomicAdd(&stats[2], (unsigned long long)n_nodes);
        atomicAdd(&stats[3], (unsigned long long)n_tris);
    }
}
