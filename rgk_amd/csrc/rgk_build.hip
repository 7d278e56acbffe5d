// Accelerator build on the device (replaces the reference's CPU kd build, src/scene.cpp:401-657, as the host
// builder of rgk_host.cpp does -- results are compared, never the structure).
//
//   1. Morton key per reference box (the centroid inside the scene box at as many bits per axis as the 64-bit key leaves beside
//      the reference index -- 14 at a million references; the index below them: unique keys)
//   2. radix sort of the 64-bit keys (hipcub)
//   3. LBVH hierarchy, one thread per internal node (Karras 2012: direction, range by exponential + binary search over the
//      longest common prefix, split)
//   4. bottom-up refit with one arrival counter per internal node: boxes (epsilon-padded like the host builder's) and subtree
//      sizes -- and, in the same pass, ONE quality step: the thread that completes a node may swap one of its children with a
//      grandchild on the other side (a tree rotation, Kensler 2008) when that shrinks the surface of the child node in between;
//      the subtree below is complete and nobody else is in it, so the step needs no locks.  A rotated node's leaves are no longer
//      one range of the sorted order: it is flagged and never becomes a leaf itself (its intact sub-subtrees do)
//   5. top-down collapse into the quantised 4-wide nodes the traversal kernels read, one level per launch: a node's children are
//      its binary children, the one with the largest surface opened again and again until there are four; subtrees of at most
//      `max_leaf` references become leaves over their (contiguous) range of the sorted order; child boxes are quantised to 8 bits
//      per plane relative to the node's box, rounded outward and re-checked with the decode the kernels use
//   6. the per-triangle intersection records gathered into leaf (= sorted) order
//
// Quality: an LBVH has no surface-area heuristic -- it builds in milliseconds and costs node visits at render time; the host's
// binned-SAH builder stays the default (rgk_scene_desc.build_flags selects).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>

#include "device_types.h"
#include "rgk_build.h"

namespace {

__device__ __forceinline__ unsigned long long expand21(unsigned long long v) { // 21 bits -> every third bit
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x001f00000000ffffull;
    v = (v | (v << 16)) & 0x001f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// key = Morton code of the box centre (`mbits` bits per axis) above the reference index (`ibits` bits): unique keys.  mbits is
// what the 64 bits leave beside the index -- 14 per axis at a million references.  (Round 2 used 10 per axis whatever the size: a
// dense mesh that occupies a tenth of the scene box -- the statue of configs[3] -- then has ~100 cells per axis for a million
// triangles, thousands of triangles share a code, and below that level the "hierarchy" is the order of their indices.)
__global__ void k_morton(const RgkBuildPrim* __restrict__ prims, uint32_t n, float3 smin, float3 sinv, int mbits, int ibits, unsigned long long* __restrict__ keys) {
    const float cells = (float)(1u << mbits), top = cells - 1.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const RgkBuildPrim p = prims[i];
        const float cx = 0.5f * (p.bmin[0] + p.bmax[0]), cy = 0.5f * (p.bmin[1] + p.bmax[1]), cz = 0.5f * (p.bmin[2] + p.bmax[2]);
        const uint32_t x = (uint32_t)fminf(fmaxf((cx - smin.x) * sinv.x * cells, 0.f), top);
        const uint32_t y = (uint32_t)fminf(fmaxf((cy - smin.y) * sinv.y * cells, 0.f), top);
        const uint32_t z = (uint32_t)fminf(fmaxf((cz - smin.z) * sinv.z * cells, 0.f), top);
        const unsigned long long code = (expand21(x) << 2) | (expand21(y) << 1) | expand21(z);
        keys[i] = (code << ibits) | i;
    }
}
// after the hierarchy is built only the reference index of a key is needed
__global__ void k_strip_keys(unsigned long long* __restrict__ keys, uint32_t n, unsigned long long mask) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) keys[i] &= mask;
}

// length of the common prefix of keys i and j (unique 64-bit keys), -1 outside the array
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// children: >= 0 internal node index, < 0: ~(position in sorted order)
__global__ void k_hierarchy(const unsigned long long* __restrict__ keys, int n, int* __restrict__ left, int* __restrict__ right,
                            int* __restrict__ parent, int* __restrict__ leaf_parent, uint32_t* __restrict__ first) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
        const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
        const int dmin = delta(keys, n, i, i - d);
        int lmax = 2;
        while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int dnode = delta(keys, n, i, j);
        int s = 0;
        for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
            if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
            if (t <= 1) break;
        }
        const int gamma = i + s * d + min(d, 0);
        const int lo = min(i, j), hi = max(i, j);
        const int lc = (lo == gamma) ? ~gamma : gamma;
        const int rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
        left[i] = lc; right[i] = rc;
        first[i] = (uint32_t)lo;
        if (lc >= 0) parent[lc] = i; else leaf_parent[~lc] = i;
        if (rc >= 0) parent[rc] = i; else leaf_parent[~rc] = i;
        if (i == 0) parent[0] = -1;
    }
}

struct BBox { float mn[3], mx[3]; };
__device__ __forceinline__ float area(const BBox& b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return (dx < 0 || dy < 0 || dz < 0) ? 0.f : 2.f * (dx * dy + dy * dz + dz * dx);
}

// box of a child code: internal -> node box, leaf -> the sorted reference's box padded by `pad`
__device__ __forceinline__ BBox child_box(int code, const BBox* __restrict__ nbox, const RgkBuildPrim* __restrict__ prims,
                                          const unsigned long long* __restrict__ keys, float pad) {
    if (code >= 0) return nbox[code];
    const RgkBuildPrim p = prims[(uint32_t)(keys[~code] & 0xffffffffull)];
    BBox b;
    for (int a = 0; a < 3; a++) { b.mn[a] = p.bmin[a] - pad; b.mx[a] = p.bmax[a] + pad; }
    return b;
}

__global__ void k_refit(const RgkBuildPrim* __restrict__ prims, const unsigned long long* __restrict__ keys, int n, float pad,
                        const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ parent,
                        const int* __restrict__ leaf_parent, int* __restrict__ arrived, BBox* __restrict__ nbox, uint32_t* __restrict__ count) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        int p = leaf_parent[k];
        while (p >= 0) {
            __threadfence();                               // what this thread wrote below is visible before it announces itself
            if (atomicAdd(&arrived[p], 1) == 0) break;     // the other child is not done: its thread will take the node
            __threadfence();                               // second to arrive: see the sibling's box
            const int lc = left[p], rc = right[p];
            const BBox a = child_box(lc, nbox, prims, keys, pad), b = child_box(rc, nbox, prims, keys, pad);
            BBox u;
            for (int x = 0; x < 3; x++) { u.mn[x] = fminf(a.mn[x], b.mn[x]); u.mx[x] = fmaxf(a.mx[x], b.mx[x]); }
            nbox[p] = u;
            count[p] = (lc >= 0 ? count[lc] : 1u) + (rc >= 0 ? count[rc] : 1u);
            p = parent[p];
        }
    }
}

#ifndef RGK_PLOC_RADIUS
#define RGK_PLOC_RADIUS 6 // places of the current order on either side in which a cluster looks for its partner (2..8: the same; 16 and more: worse)
#endif
#define RGK_ROTATED 0x80000000u // count[]: the node's leaves are not one range of the sorted order any more
__device__ __forceinline__ BBox unite(const BBox& a, const BBox& b) {
    BBox u;
    for (int x = 0; x < 3; x++) { u.mn[x] = fminf(a.mn[x], b.mn[x]); u.mx[x] = fmaxf(a.mx[x], b.mx[x]); }
    return u;
}
// The same pass with the rotation step (see the header).  count[] carries RGK_ROTATED in its top bit.
__global__ void k_refit_rotate(const RgkBuildPrim* __restrict__ prims, const unsigned long long* __restrict__ keys, int n, float pad,
                               int* __restrict__ left, int* __restrict__ right, int* __restrict__ parent,
                               int* __restrict__ leaf_parent, int* __restrict__ arrived, BBox* __restrict__ nbox, uint32_t* __restrict__ count) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        int p = leaf_parent[k];
        while (p >= 0) {
            __threadfence();
            if (atomicAdd(&arrived[p], 1) == 0) break;
            __threadfence();
            int lc = left[p], rc = right[p];
            BBox a = child_box(lc, nbox, prims, keys, pad), b = child_box(rc, nbox, prims, keys, pad);
            nbox[p] = unite(a, b);
            const uint32_t cl = lc >= 0 ? (count[lc] & ~RGK_ROTATED) : 1u, cr = rc >= 0 ? (count[rc] & ~RGK_ROTATED) : 1u;
            count[p] = (cl + cr) | (count[p] & RGK_ROTATED); // (the flag of an earlier pass stays: a rotated node's leaves never become one range again)
            // rotations: swap `a` (left) with a child of the right node, or `b` (right) with a child of the left node -- the node in
            // between then bounds {the swapped-in child, its remaining child}; take the swap that shrinks its surface most
            float best = 0.f;
            int which = -1;
            BBox nb{};
            if (rc >= 0) {
                const int rl = left[rc], rr = right[rc];
                const BBox brl = child_box(rl, nbox, prims, keys, pad), brr = child_box(rr, nbox, prims, keys, pad);
                const float ar = area(b);
                const BBox u0 = unite(a, brr), u1 = unite(brl, a); // left <-> rl : right node = {left, rr};  left <-> rr : right node = {rl, left}
                if (ar - area(u0) > best) { best = ar - area(u0); which = 0; nb = u0; }
                if (ar - area(u1) > best) { best = ar - area(u1); which = 1; nb = u1; }
            }
            if (lc >= 0) {
                const int ll = left[lc], lr = right[lc];
                const BBox bll = child_box(ll, nbox, prims, keys, pad), blr = child_box(lr, nbox, prims, keys, pad);
                const float al = area(a);
                const BBox u2 = unite(b, blr), u3 = unite(bll, b); // right <-> ll : left node = {right, lr};  right <-> lr : left node = {ll, right}
                if (al - area(u2) > best) { best = al - area(u2); which = 2; nb = u2; }
                if (al - area(u3) > best) { best = al - area(u3); which = 3; nb = u3; }
            }
            if (which >= 0) {
                auto cnt = [&](int c) { return c >= 0 ? (count[c] & ~RGK_ROTATED) : 1u; };
                auto adopt = [&](int child, int by) { if (child >= 0) parent[child] = by; else leaf_parent[~child] = by; }; // (the next pass climbs these)
                if (which == 0) { const int rl = left[rc]; left[p] = rl; left[rc] = lc; count[rc] = (cnt(lc) + cnt(right[rc])) | RGK_ROTATED; nbox[rc] = nb; adopt(rl, p); adopt(lc, rc); }
                else if (which == 1) { const int rr = right[rc]; left[p] = rr; right[rc] = lc; count[rc] = (cnt(left[rc]) + cnt(lc)) | RGK_ROTATED; nbox[rc] = nb; adopt(rr, p); adopt(lc, rc); }
                else if (which == 2) { const int ll = left[lc]; right[p] = ll; left[lc] = rc; count[lc] = (cnt(rc) + cnt(right[lc])) | RGK_ROTATED; nbox[lc] = nb; adopt(ll, p); adopt(rc, lc); }
                else { const int lr = right[lc]; right[p] = lr; right[lc] = rc; count[lc] = (cnt(left[lc]) + cnt(rc)) | RGK_ROTATED; nbox[lc] = nb; adopt(lr, p); adopt(rc, lc); }
                // (p itself still covers exactly its old range of the sorted order: only the node in between lost that)
            }
            p = parent[p];
        }
    }
}

// ---- PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) instead of the Karras hierarchy: the tree is built
// bottom-up from the Morton-ordered references by merging, round after round, every pair of clusters that are each other's
// best partner -- the one within `radius` places of the current order whose union with it has the smallest surface.  That is
// a surface-area criterion at every merge, which the prefix hierarchy does not have.  cid: a cluster's child code (>= 0 inner
// node, < 0 ~(position of a reference in the sorted order)); cbox: its box; cnum: references below it.
__device__ __forceinline__ bool ploc_better(float a, int lo, int hi, float ba, int blo, int bhi) { // one total order for both partners
    return a < ba || (a == ba && (lo < blo || (lo == blo && hi < bhi)));
}
__global__ void k_ploc_init(const RgkBuildPrim* __restrict__ prims, const unsigned long long* __restrict__ keys, uint32_t n, float pad,
                            int* __restrict__ cid, BBox* __restrict__ cbox, uint32_t* __restrict__ cnum) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        cid[k] = ~(int)k; cnum[k] = 1u;
        cbox[k] = child_box(~(int)k, nullptr, prims, keys, pad);
    }
}
__global__ void k_ploc_nn(const BBox* __restrict__ cbox, int m, int radius, int* __restrict__ nn) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const BBox b = cbox[i];
        float best = __builtin_inff();
        int bj = -1;
        const int j0 = max(0, i - radius), j1 = min(m - 1, i + radius);
        for (int j = j0; j <= j1; j++) {
            if (j == i) continue;
            const float a = area(unite(b, cbox[j]));
            if (bj < 0 || ploc_better(a, min(i, j), max(i, j), best, min(i, bj), max(i, bj))) { best = a; bj = j; }
        }
        nn[i] = bj;
    }
}
// inner nodes are numbered from n - 2 downwards in the order they are made, so the last one -- the root -- is node 0
__global__ void k_ploc_merge(const int* __restrict__ cid, const BBox* __restrict__ cbox, const uint32_t* __restrict__ cnum, const int* __restrict__ nn, int m, int n,
                             uint32_t* __restrict__ n_made, int* __restrict__ left, int* __restrict__ right, int* __restrict__ parent, int* __restrict__ leaf_parent,
                             BBox* __restrict__ nbox, uint32_t* __restrict__ count, int* __restrict__ cid2, BBox* __restrict__ cbox2, uint32_t* __restrict__ cnum2,
                             uint32_t* __restrict__ keep) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int j = nn[i];
        const bool mutual = j >= 0 && nn[j] == i;
        if (mutual && i > j) { keep[i] = 0u; continue; }
        keep[i] = 1u;
        if (!mutual) { cid2[i] = cid[i]; cbox2[i] = cbox[i]; cnum2[i] = cnum[i]; continue; }
        const int id = n - 2 - (int)atomicAdd(n_made, 1u);
        const int a = cid[i], b = cid[j];
        const BBox u = unite(cbox[i], cbox[j]);
        left[id] = a; right[id] = b;
        if (a >= 0) parent[a] = id; else leaf_parent[~a] = id;
        if (b >= 0) parent[b] = id; else leaf_parent[~b] = id;
        nbox[id] = u; count[id] = cnum[i] + cnum[j];
        if (id == 0) parent[0] = -1;
        cid2[i] = id; cbox2[i] = u; cnum2[i] = cnum[i] + cnum[j];
    }
}
__global__ void k_ploc_compact(const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos, int m, const int* __restrict__ cid2, const BBox* __restrict__ cbox2,
                               const uint32_t* __restrict__ cnum2, int* __restrict__ cid, BBox* __restrict__ cbox, uint32_t* __restrict__ cnum) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
        if (keep[i]) { const uint32_t p = pos[i]; cid[p] = cid2[i]; cbox[p] = cbox2[i]; cnum[p] = cnum2[i]; }
}
// A subtree of that tree is not a range of the Morton order, and the collapse makes leaves of ranges: the references are laid
// out again in the order of the tree's own leaves (left subtree first).  The place of code c = the references to the left of it =
// the sum of its left siblings' sizes on the way up.
__device__ __forceinline__ uint32_t ploc_offset(int c, int p, const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ parent,
                                                const uint32_t* __restrict__ count) {
    uint32_t o = 0;
    while (p >= 0) {
        if (right[p] == c) { const int l = left[p]; o += l >= 0 ? count[l] : 1u; }
        c = p; p = parent[p];
    }
    return o;
}
__global__ void k_ploc_places(int n, const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ parent, const int* __restrict__ leaf_parent,
                              const uint32_t* __restrict__ count, const unsigned long long* __restrict__ keys, uint32_t* __restrict__ first,
                              uint32_t* __restrict__ place, unsigned long long* __restrict__ keys_out, int* __restrict__ leaf_parent_out) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t at = ploc_offset(~k, leaf_parent[k], left, right, parent, count);
        place[k] = at; keys_out[at] = keys[k]; leaf_parent_out[at] = leaf_parent[k];
        if (k < n - 1) first[k] = ploc_offset(k, parent[k], left, right, parent, count);
    }
}
__global__ void k_ploc_rename(int n, int* __restrict__ left, int* __restrict__ right, const uint32_t* __restrict__ place) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
        const int l = left[i], r = right[i];
        if (l < 0) left[i] = ~(int)place[~l];
        if (r < 0) right[i] = ~(int)place[~r];
    }
}

// quantise one axis of up to four child boxes against the node box: the smallest power-of-two step whose outward-rounded codes
// fit 8 bits, verified with the kernels' decode fma(q, step, p) -- the host builder's loop (rgk_host.cpp QbvhBuilder)
__device__ void quantise_axis(const BBox* ch, int nch, int a, float p, float ext, float& step, uint8_t* qlo, uint8_t* qhi) {
    int e = -126;
    if (ext > 0.f) { (void)frexpf(ext / 255.0f, &e); }
    for (;; e++) {
        if (e < -126) e = -126;
        const float scale = ldexpf(1.0f, e);
        bool ok = true;
        uint8_t lo[4], hi[4];
        for (int i = 0; i < nch && ok; i++) {
            float fl = floorf((ch[i].mn[a] - p) / scale), fh = ceilf((ch[i].mx[a] - p) / scale);
            if (fl < 0.f) fl = 0.f;
            while (fl > 0.f && __builtin_fmaf(fl, scale, p) > ch[i].mn[a]) fl -= 1.f;
            while (fh <= 255.f && __builtin_fmaf(fh, scale, p) < ch[i].mx[a]) fh += 1.f;
            if (fh > 255.f || fl > 255.f) { ok = false; break; }
            lo[i] = (uint8_t)fl; hi[i] = (uint8_t)fh;
        }
        if (!ok && e < 127) continue;
        step = scale;
        for (int i = 0; i < 4; i++) { qlo[i] = i < nch ? lo[i] : 255; qhi[i] = i < nch ? hi[i] : 0; }
        return;
    }
}

// one level of the collapse: frontier entries {binary node, output index} -> QNodes + the next frontier
__global__ void k_collapse(const int2* __restrict__ frontier, uint32_t n_front, int2* __restrict__ next, uint32_t* __restrict__ n_next,
                           uint32_t* __restrict__ n_out, uint32_t max_leaf, float pad, const RgkBuildPrim* __restrict__ prims,
                           const unsigned long long* __restrict__ keys, const int* __restrict__ left, const int* __restrict__ right,
                           const BBox* __restrict__ nbox, const uint32_t* __restrict__ count, const uint32_t* __restrict__ first,
                           QNode* __restrict__ out) {
    for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < n_front; f += gridDim.x * blockDim.x) {
        const int b = frontier[f].x, oi = frontier[f].y;
        int ref[4];
        BBox box[4];
        int nch = 2;
        ref[0] = left[b]; ref[1] = right[b];
        box[0] = child_box(ref[0], nbox, prims, keys, pad); box[1] = child_box(ref[1], nbox, prims, keys, pad);
        while (nch < 4) { // open the expandable child with the largest surface
            int best = -1;
            float best_area = -1.f;
            for (int i = 0; i < nch; i++)
                if (ref[i] >= 0 && ((count[ref[i]] & ~RGK_ROTATED) > max_leaf || (count[ref[i]] & RGK_ROTATED))) { const float ar = area(box[i]); if (ar > best_area) { best_area = ar; best = i; } }
            if (best < 0) break;
            const int o = ref[best];
            for (int i = best; i + 1 < nch; i++) { ref[i] = ref[i + 1]; box[i] = box[i + 1]; } // erase, then append both children (host order)
            nch--;
            ref[nch] = left[o]; box[nch] = child_box(ref[nch], nbox, prims, keys, pad); nch++;
            ref[nch] = right[o]; box[nch] = child_box(ref[nch], nbox, prims, keys, pad); nch++;
        }
        BBox nb;
        for (int a = 0; a < 3; a++) { nb.mn[a] = box[0].mn[a]; nb.mx[a] = box[0].mx[a]; }
        for (int i = 1; i < nch; i++)
            for (int a = 0; a < 3; a++) { nb.mn[a] = fminf(nb.mn[a], box[i].mn[a]); nb.mx[a] = fmaxf(nb.mx[a], box[i].mx[a]); }
        QNode q;
        for (int a = 0; a < 3; a++) q.p[a] = nb.mn[a];
        quantise_axis(box, nch, 0, q.p[0], nb.mx[0] - nb.mn[0], q.sx, q.qlo[0], q.qhi[0]);
        quantise_axis(box, nch, 1, q.p[1], nb.mx[1] - nb.mn[1], q.sy, q.qlo[1], q.qhi[1]);
        quantise_axis(box, nch, 2, q.p[2], nb.mx[2] - nb.mn[2], q.sz, q.qlo[2], q.qhi[2]);
        for (int i = 0; i < 4; i++) {
            int code = RGK_QNODE_EMPTY;
            if (i < nch) {
                const int r = ref[i];
                if (r < 0) code = (int)~((((uint32_t)~r) << 4) | 0u);                               // one reference
                else if (count[r] <= max_leaf) code = (int)~((first[r] << 4) | (count[r] - 1u));      // a small, unrotated subtree: its range of the sorted order
                else {
                    const uint32_t idx = atomicAdd(n_out, 1u);
                    const uint32_t slot = atomicAdd(n_next, 1u);
                    next[slot] = make_int2(r, (int)idx);
                    code = (int)idx;
                }
            }
            q.child[i] = code;
        }
        out[oi] = q;
    }
}

__global__ void k_gather_recs(const unsigned long long* __restrict__ keys, uint32_t n, const RgkBuildPrim* __restrict__ prims,
                              const TriIsect* __restrict__ recs, TriIsect* __restrict__ leaf_recs, float4* __restrict__ leaf_pb) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const RgkBuildPrim p = prims[(uint32_t)(keys[k] & 0xffffffffull)];
        leaf_recs[k] = recs[p.tri];
        leaf_pb[k] = make_float4(p.pb[0], p.pb[1], p.pb[2], p.pb[3]);
    }
}

// ------------------------------------------------------------------ refit: moved vertices, same triangles, same tree
// The intersection record of one reference from the triangle's (new) vertices -- operation for operation what rgk_scene_create
// computes on the host (Triangle::CalculatePlane, src/primitives.cpp:24-36, and the per-triangle differences TestIntersection
// would recompute): same IEEE operations in the same order, no contraction, so a refitted scene holds the bits a fresh one would.
__global__ void k_refit_recs(const float* __restrict__ vertices, const uint32_t* __restrict__ idx, const float4* __restrict__ leaf_pb, uint32_t n_refs,
                             TriIsect* __restrict__ recs, BBox* __restrict__ refbox) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_refs; k += gridDim.x * blockDim.x) {
        const uint32_t tri = recs[k].tri;
        float v[3][3];
        for (int c = 0; c < 3; c++) { const uint32_t vi = idx[3 * tri + c]; for (int a = 0; a < 3; a++) v[c][a] = vertices[3 * (size_t)vi + a]; }
        const float d0[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]}, d1[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
        // crossv(d1, d0), normv, -dotv(n, v0): rgk_host.cpp's helpers spelled out
        const float cx = d1[1] * d0[2] - d0[1] * d1[2], cy = d1[2] * d0[0] - d0[2] * d1[0], cz = d1[0] * d0[1] - d0[0] * d1[1];
        const float tx = cx * cx, ty = cy * cy, tz = cz * cz;
        const float inv = 1.0f / __builtin_sqrtf(tx + ty + tz);
        const float nx = cx * inv, ny = cy * inv, nz = cz * inv;
        const float ux = nx * v[0][0], uy = ny * v[0][1], uz = nz * v[0][2];
        TriIsect r;
        r.n[0] = nx; r.n[1] = ny; r.n[2] = nz; r.d = -(ux + uy + uz);
        int i1, i2;
        const float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
        if (ax > ay && ax > az) { i1 = 1; i2 = 2; }
        else if (ay > az) { i1 = 0; i2 = 2; }
        else { i1 = 0; i2 = 1; }
        r.v0a = v[0][i1]; r.v0b = v[0][i2];
        r.q1x = v[1][i1] - v[0][i1]; r.q1y = v[1][i2] - v[0][i2];
        r.q2x = v[2][i1] - v[0][i1]; r.q2y = v[2][i2] - v[0][i2];
        r.axes = (uint32_t)i1 | ((uint32_t)i2 << 2);
        r.tri = tri;
        recs[k] = r;
        BBox b; // the whole triangle's box ...
        for (int a = 0; a < 3; a++) { b.mn[a] = fminf(v[0][a], fminf(v[1][a], v[2][a])); b.mx[a] = fmaxf(v[0][a], fmaxf(v[1][a], v[2][a])); }
        // ... or, for one of the pieces a large triangle was split into at build time, the box of that piece: a triangle moves
        // affinely, so the piece is still {v0 + b (v1 - v0) + c (v2 - v0)} over its old parameter box -- a parallelogram whose four
        // corners bound it (widened by a few ulps of the coordinates: the corners are rounded)
        const float4 pb = leaf_pb[k];
        if (!(pb.x == 0.f && pb.y == 1.f && pb.z == 0.f && pb.w == 1.f))
            for (int a = 0; a < 3; a++) {
                const float e1 = v[1][a] - v[0][a], e2 = v[2][a] - v[0][a];
                const float c00 = v[0][a] + pb.x * e1 + pb.z * e2, c10 = v[0][a] + pb.y * e1 + pb.z * e2, c01 = v[0][a] + pb.x * e1 + pb.w * e2, c11 = v[0][a] + pb.y * e1 + pb.w * e2;
                const float m = 4e-6f * (fabsf(v[0][a]) + fabsf(e1) + fabsf(e2));
                b.mn[a] = fmaxf(b.mn[a], fminf(fminf(c00, c10), fminf(c01, c11)) - m);
                b.mx[a] = fminf(b.mx[a], fmaxf(fmaxf(c00, c10), fmaxf(c01, c11)) + m);
            }
        refbox[k] = b;
    }
}
// the three normals / tangents of every triangle's shading record from the (new) per-vertex arrays; uvs and material stay
__global__ void k_refit_shade(const float* __restrict__ normals, const float* __restrict__ tangents, const uint32_t* __restrict__ idx, uint32_t n_tris,
                              TriShade* __restrict__ shade) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_tris; t += gridDim.x * blockDim.x)
        for (int c = 0; c < 3; c++) {
            const uint32_t vi = idx[3 * t + c];
            for (int a = 0; a < 3; a++) {
                if (normals) shade[t].q[c][a] = normals[3 * (size_t)vi + a];
                if (tangents) shade[t].q[3 + c][a] = tangents[3 * (size_t)vi + a];
            }
        }
}
// who is whose parent in the 4-wide tree, and how many inner children each node waits for
__global__ void k_qbvh_parents(const QNode* __restrict__ nodes, uint32_t n_nodes, int* __restrict__ parent, int* __restrict__ n_inner) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += gridDim.x * blockDim.x) {
        int inner = 0;
        for (int c = 0; c < 4; c++) { const int ch = nodes[i].child[c]; if (ch >= 0 && ch != RGK_QNODE_EMPTY) { parent[ch] = (int)i; inner++; } }
        n_inner[i] = inner;
        if (i == 0) parent[0] = -1;
    }
}
// bottom-up: a node whose inner children are all done recomputes its child boxes (leaves: the references' boxes; inner: the
// child's own new box), its own box, and the 8-bit codes -- then reports to its parent; the last child to report takes the parent
__global__ void k_qbvh_refit(QNode* __restrict__ nodes, uint32_t n_nodes, const int* __restrict__ parent, const int* __restrict__ n_inner,
                             int* __restrict__ arrived, const BBox* __restrict__ refbox, BBox* __restrict__ nbox, float pad) {
    for (uint32_t start = blockIdx.x * blockDim.x + threadIdx.x; start < n_nodes; start += gridDim.x * blockDim.x) {
        if (n_inner[start] != 0) continue;
        int i = (int)start;
        for (;;) {
            QNode q = nodes[i];
            BBox ch[4];
            int slot[4], nch = 0;
            for (int c = 0; c < 4; c++) {
                const int code = q.child[c];
                if (code == RGK_QNODE_EMPTY) continue;
                BBox b;
                if (code >= 0) b = nbox[code];
                else {
                    const uint32_t u = ~(uint32_t)code, first = u >> 4, cnt = (u & 15u) + 1u;
                    b = refbox[first];
                    for (uint32_t k = 1; k < cnt; k++) b = unite(b, refbox[first + k]);
                    for (int a = 0; a < 3; a++) { b.mn[a] -= pad; b.mx[a] += pad; }
                }
                ch[nch] = b; slot[nch] = c; nch++;
            }
            BBox nb = ch[0];
            for (int k = 1; k < nch; k++) nb = unite(nb, ch[k]);
            nbox[i] = nb;
            uint8_t qlo[3][4], qhi[3][4];
            float step[3];
            for (int a = 0; a < 3; a++) quantise_axis(ch, nch, a, nb.mn[a], nb.mx[a] - nb.mn[a], step[a], qlo[a], qhi[a]);
            for (int a = 0; a < 3; a++) {
                q.p[a] = nb.mn[a];
                for (int c = 0; c < 4; c++) { q.qlo[a][c] = 255; q.qhi[a][c] = 0; }
                for (int k = 0; k < nch; k++) { q.qlo[a][slot[k]] = qlo[a][k]; q.qhi[a][slot[k]] = qhi[a][k]; }
            }
            q.sx = step[0]; q.sy = step[1]; q.sz = step[2];
            nodes[i] = q;
            const int p = parent[i];
            if (p < 0) break;
            __threadfence();
            if (atomicAdd(&arrived[p], 1) + 1 < n_inner[p]) break; // a sibling is not done yet: its thread takes the parent
            __threadfence();
            i = p;
        }
    }
}

template <typename T>
struct Tmp { // device scratch freed on scope exit
    T* p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
    ~Tmp() { if (p) (void)hipFree(p); }
};

} // namespace

#define BCHK_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { *err = hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? -3 : -2; } } while (0)
int rgk_refit_bvh4_device(hipStream_t st, uint32_t n_refs, uint32_t n_nodes, uint32_t n_tris, const float* d_vertices, const float* d_normals, const float* d_tangents,
                          const uint32_t* d_idx, const float4* d_leaf_pb, float pad, TriIsect* d_leaf_recs, QNode* d_nodes, TriShade* d_shade, const char** err) {
    *err = "";
    Tmp<BBox> refbox, nbox;
    Tmp<int> parent, n_inner, arrived;
    BCHK_(refbox.alloc(n_refs)); BCHK_(nbox.alloc(n_nodes)); BCHK_(parent.alloc(n_nodes)); BCHK_(n_inner.alloc(n_nodes)); BCHK_(arrived.alloc(n_nodes));
    BCHK_(hipMemsetAsync(arrived.p, 0, (size_t)n_nodes * sizeof(int), st));
    const int gr = (int)std::min<uint32_t>((n_refs + 255) / 256, 256 * 16), gn = (int)std::min<uint32_t>((n_nodes + 255) / 256, 256 * 16);
    k_refit_recs<<<gr, 256, 0, st>>>(d_vertices, d_idx, d_leaf_pb, n_refs, d_leaf_recs, refbox.p);
    if (d_normals || d_tangents) k_refit_shade<<<(int)std::min<uint32_t>((n_tris + 255) / 256, 256 * 16), 256, 0, st>>>(d_normals, d_tangents, d_idx, n_tris, d_shade);
    k_qbvh_parents<<<gn, 256, 0, st>>>(d_nodes, n_nodes, parent.p, n_inner.p);
    k_qbvh_refit<<<gn, 256, 0, st>>>(d_nodes, n_nodes, parent.p, n_inner.p, arrived.p, refbox.p, nbox.p, pad);
    BCHK_(hipStreamSynchronize(st));
    BCHK_(hipGetLastError());
    return 0;
}

#define BCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { *err = hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? -3 : -2; } } while (0)

int rgk_build_bvh4_device(hipStream_t st, const RgkBuildPrim* h_prims, uint32_t n, const float smin[3], const float smax[3], float pad,
                          uint32_t max_leaf, int rotate, const TriIsect* d_recs, QNode* d_nodes, TriIsect* d_leaf_recs, float4* d_leaf_pb, uint32_t* n_nodes,
                          uint32_t* n_levels, const char** err) {
    *err = "";
    if (n < 2 || n <= max_leaf) { *err = "too few references for the device build"; return -5; }
    Tmp<RgkBuildPrim> prims;
    Tmp<unsigned long long> keys, keys_sorted;
    Tmp<int> left, right, parent, leaf_parent, arrived;
    Tmp<uint32_t> first, count, ctr;
    Tmp<BBox> nbox;
    Tmp<int2> fa, fb;
    BCHK(prims.alloc(n)); BCHK(keys.alloc(n)); BCHK(keys_sorted.alloc(n));
    BCHK(left.alloc(n)); BCHK(right.alloc(n)); BCHK(parent.alloc(n)); BCHK(leaf_parent.alloc(n)); BCHK(arrived.alloc(n));
    BCHK(first.alloc(n)); BCHK(count.alloc(n)); BCHK(ctr.alloc(4)); BCHK(nbox.alloc(n)); BCHK(fa.alloc(n)); BCHK(fb.alloc(n));
    BCHK(hipMemcpyAsync(prims.p, h_prims, (size_t)n * sizeof(RgkBuildPrim), hipMemcpyHostToDevice, st));
    const int grid = (int)std::min<uint32_t>((n + 255) / 256, 256 * 16);
    float3 mn = make_float3(smin[0], smin[1], smin[2]);
    float3 inv = make_float3(smax[0] > smin[0] ? 1.f / (smax[0] - smin[0]) : 0.f, smax[1] > smin[1] ? 1.f / (smax[1] - smin[1]) : 0.f,
                             smax[2] > smin[2] ? 1.f / (smax[2] - smin[2]) : 0.f);
    int ibits = 1;
    while (ibits < 32 && (1ull << ibits) < (unsigned long long)n) ibits++;
    int mbits = std::min(21, (64 - ibits) / 3);
    if (const char* e = std::getenv("RGK_LBVH_MORTON_BITS")) mbits = std::max(1, std::min(mbits, std::atoi(e))); // experiments (10 = round 2's keys)
    k_morton<<<grid, 256, 0, st>>>(prims.p, n, mn, inv, mbits, ibits, keys.p);
    {
        size_t tmp_bytes = 0;
        BCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, keys.p, keys_sorted.p, (int)n, 0, ibits + 3 * mbits, st));
        Tmp<unsigned char> tmp;
        BCHK(tmp.alloc(tmp_bytes));
        BCHK(hipcub::DeviceRadixSort::SortKeys(tmp.p, tmp_bytes, keys.p, keys_sorted.p, (int)n, 0, ibits + 3 * mbits, st));
        BCHK(hipStreamSynchronize(st)); // tmp goes out of scope
    }
    BCHK(hipMemsetAsync(arrived.p, 0, (size_t)n * sizeof(int), st));
    int ploc_radius = RGK_PLOC_RADIUS;
    if (const char* e = std::getenv("RGK_LBVH_PLOC")) ploc_radius = std::max(0, std::min(256, std::atoi(e))); // 0: the Karras hierarchy
    if (ploc_radius && !std::getenv("RGK_LBVH_ROTATE")) rotate = 0; // (rotations on top of the clustered tree: measured, nothing -- tools/gpu_lbvh_morton_sweep.py)
    unsigned long long* order = keys_sorted.p; // the reference behind every leaf position (low bits of the key)
    Tmp<unsigned long long> keys_tree;
    Tmp<int> leaf_parent_tree;
    if (!ploc_radius) {
        k_hierarchy<<<grid, 256, 0, st>>>(keys_sorted.p, (int)n, left.p, right.p, parent.p, leaf_parent.p, first.p);
        k_strip_keys<<<grid, 256, 0, st>>>(keys_sorted.p, n, (1ull << ibits) - 1ull);
        BCHK(hipMemsetAsync(count.p, 0, (size_t)n * sizeof(uint32_t), st));
    } else {
        k_strip_keys<<<grid, 256, 0, st>>>(keys_sorted.p, n, (1ull << ibits) - 1ull);
        Tmp<int> cid, cid2, nn;
        Tmp<BBox> cbox, cbox2;
        Tmp<uint32_t> cnum, cnum2, keep, pos, place;
        BCHK(cid.alloc(n)); BCHK(cid2.alloc(n)); BCHK(nn.alloc(n)); BCHK(cbox.alloc(n)); BCHK(cbox2.alloc(n));
        BCHK(cnum.alloc(n)); BCHK(cnum2.alloc(n)); BCHK(keep.alloc(n)); BCHK(pos.alloc(n)); BCHK(place.alloc(n));
        BCHK(keys_tree.alloc(n)); BCHK(leaf_parent_tree.alloc(n));
        size_t scan_bytes = 0;
        BCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, keep.p, pos.p, (int)n, st));
        Tmp<unsigned char> scan_tmp;
        BCHK(scan_tmp.alloc(scan_bytes));
        BCHK(hipMemsetAsync(ctr.p + 2, 0, sizeof(uint32_t), st)); // ctr[2] = inner nodes made so far
        k_ploc_init<<<grid, 256, 0, st>>>(prims.p, keys_sorted.p, n, pad, cid.p, cbox.p, cnum.p);
        uint32_t m = n;
        for (int round = 0; m > 1; round++) {
            if (round > 4096) { *err = "device BVH: clustering does not converge"; return -5; }
            const int g = (int)std::min<uint32_t>((m + 255) / 256, 256 * 16);
            k_ploc_nn<<<g, 256, 0, st>>>(cbox.p, (int)m, ploc_radius, nn.p);
            k_ploc_merge<<<g, 256, 0, st>>>(cid.p, cbox.p, cnum.p, nn.p, (int)m, (int)n, ctr.p + 2, left.p, right.p, parent.p, leaf_parent.p, nbox.p, count.p,
                                            cid2.p, cbox2.p, cnum2.p, keep.p);
            BCHK(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, scan_bytes, keep.p, pos.p, (int)m, st));
            k_ploc_compact<<<g, 256, 0, st>>>(keep.p, pos.p, (int)m, cid2.p, cbox2.p, cnum2.p, cid.p, cbox.p, cnum.p);
            uint32_t made = 0;
            BCHK(hipMemcpyAsync(&made, ctr.p + 2, sizeof(made), hipMemcpyDeviceToHost, st));
            BCHK(hipStreamSynchronize(st));
            if (n - made >= m) { *err = "device BVH: a clustering round merged nothing"; return -5; }
            m = n - made;
        }
        k_ploc_places<<<grid, 256, 0, st>>>((int)n, left.p, right.p, parent.p, leaf_parent.p, count.p, keys_sorted.p, first.p, place.p, keys_tree.p, leaf_parent_tree.p);
        k_ploc_rename<<<grid, 256, 0, st>>>((int)n, left.p, right.p, place.p);
        BCHK(hipMemcpyAsync(leaf_parent.p, leaf_parent_tree.p, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, st));
        order = keys_tree.p;
        BCHK(hipStreamSynchronize(st)); // the round's temporaries go out of scope
    }
    for (int pass = 0; pass < rotate; pass++) { // each pass: the whole tree bottom-up, one rotation per node at most
        if (pass) BCHK(hipMemsetAsync(arrived.p, 0, (size_t)n * sizeof(int), st));
        k_refit_rotate<<<grid, 256, 0, st>>>(prims.p, order, (int)n, pad, left.p, right.p, parent.p, leaf_parent.p, arrived.p, nbox.p, count.p);
    }
    if (!rotate) k_refit<<<grid, 256, 0, st>>>(prims.p, order, (int)n, pad, left.p, right.p, parent.p, leaf_parent.p, arrived.p, nbox.p, count.p);
    k_gather_recs<<<grid, 256, 0, st>>>(order, n, prims.p, d_recs, d_leaf_recs, d_leaf_pb);
    // collapse, level by level.  ctr[0] = nodes allocated, ctr[1] = next frontier length
    uint32_t h[2] = {1u, 0u};
    BCHK(hipMemcpyAsync(ctr.p, h, sizeof(h), hipMemcpyHostToDevice, st));
    const int2 root = make_int2(0, 0);
    BCHK(hipMemcpyAsync(fa.p, &root, sizeof(root), hipMemcpyHostToDevice, st));
    uint32_t n_front = 1, levels = 0;
    int2 *cur = fa.p, *nxt = fb.p;
    while (n_front) {
        levels++;
        if (levels > 200) { *err = "device BVH deeper than 200 levels"; return -5; }
        k_collapse<<<(int)std::min<uint32_t>((n_front + 63) / 64, 4096), 64, 0, st>>>(cur, n_front, nxt, ctr.p + 1, ctr.p, max_leaf, pad, prims.p, order,
                                                                                     left.p, right.p, nbox.p, count.p, first.p, d_nodes);
        BCHK(hipMemcpyAsync(h, ctr.p, sizeof(h), hipMemcpyDeviceToHost, st));
        BCHK(hipStreamSynchronize(st));
        n_front = h[1];
        const uint32_t zero = 0;
        BCHK(hipMemcpyAsync(ctr.p + 1, &zero, sizeof(zero), hipMemcpyHostToDevice, st));
        std::swap(cur, nxt);
    }
    BCHK(hipStreamSynchronize(st));
    BCHK(hipGetLastError());
    *n_nodes = h[0];
    *n_levels = levels;
    return 0;
}
