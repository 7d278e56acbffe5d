// Accelerator build on the device (replaces the reference's CPU kd build, src/scene.cpp:401-657, as the host
// builder of rgk_host.cpp does -- results are compared, never the structure).
//
//   1. Morton key per reference box (30 bits of the centroid inside the scene box, the reference index below them: unique keys)
//   2. radix sort of the 64-bit keys (hipcub)
//   3. LBVH hierarchy, one thread per internal node (Karras 2012: direction, range by exponential + binary search over the
//      longest common prefix, split)
//   4. bottom-up refit with one arrival counter per internal node: boxes (epsilon-padded like the host builder's) and subtree sizes
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

#include "device_types.h"
#include "rgk_build.h"

namespace {

__device__ __forceinline__ uint32_t expand10(uint32_t v) { // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void k_morton(const RgkBuildPrim* __restrict__ prims, uint32_t n, float3 smin, float3 sinv, unsigned long long* __restrict__ keys) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const RgkBuildPrim p = prims[i];
        const float cx = 0.5f * (p.bmin[0] + p.bmax[0]), cy = 0.5f * (p.bmin[1] + p.bmax[1]), cz = 0.5f * (p.bmin[2] + p.bmax[2]);
        const uint32_t x = (uint32_t)fminf(fmaxf((cx - smin.x) * sinv.x * 1024.f, 0.f), 1023.f);
        const uint32_t y = (uint32_t)fminf(fmaxf((cy - smin.y) * sinv.y * 1024.f, 0.f), 1023.f);
        const uint32_t z = (uint32_t)fminf(fmaxf((cz - smin.z) * sinv.z * 1024.f, 0.f), 1023.f);
        const uint32_t code = (expand10(x) << 2) | (expand10(y) << 1) | expand10(z);
        keys[i] = ((unsigned long long)code << 32) | i;
    }
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
                if (ref[i] >= 0 && count[ref[i]] > max_leaf) { const float ar = area(box[i]); if (ar > best_area) { best_area = ar; best = i; } }
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
                else if (count[r] <= max_leaf) code = (int)~((first[r] << 4) | (count[r] - 1u));      // a small subtree: its range of the sorted order
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
                              const TriIsect* __restrict__ recs, TriIsect* __restrict__ leaf_recs) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
        leaf_recs[k] = recs[prims[(uint32_t)(keys[k] & 0xffffffffull)].tri];
}

template <typename T>
struct Tmp { // device scratch freed on scope exit
    T* p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
    ~Tmp() { if (p) (void)hipFree(p); }
};

} // namespace

#define BCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { *err = hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? -3 : -2; } } while (0)

int rgk_build_bvh4_device(hipStream_t st, const RgkBuildPrim* h_prims, uint32_t n, const float smin[3], const float smax[3], float pad,
                          uint32_t max_leaf, const TriIsect* d_recs, QNode* d_nodes, TriIsect* d_leaf_recs, uint32_t* n_nodes,
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
    k_morton<<<grid, 256, 0, st>>>(prims.p, n, mn, inv, keys.p);
    {
        size_t tmp_bytes = 0;
        BCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, keys.p, keys_sorted.p, (int)n, 0, 62, st));
        Tmp<unsigned char> tmp;
        BCHK(tmp.alloc(tmp_bytes));
        BCHK(hipcub::DeviceRadixSort::SortKeys(tmp.p, tmp_bytes, keys.p, keys_sorted.p, (int)n, 0, 62, st));
        BCHK(hipStreamSynchronize(st)); // tmp goes out of scope
    }
    BCHK(hipMemsetAsync(arrived.p, 0, (size_t)n * sizeof(int), st));
    k_hierarchy<<<grid, 256, 0, st>>>(keys_sorted.p, (int)n, left.p, right.p, parent.p, leaf_parent.p, first.p);
    k_refit<<<grid, 256, 0, st>>>(prims.p, keys_sorted.p, (int)n, pad, left.p, right.p, parent.p, leaf_parent.p, arrived.p, nbox.p, count.p);
    k_gather_recs<<<grid, 256, 0, st>>>(keys_sorted.p, n, prims.p, d_recs, d_leaf_recs);
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
        k_collapse<<<(int)std::min<uint32_t>((n_front + 63) / 64, 4096), 64, 0, st>>>(cur, n_front, nxt, ctr.p + 1, ctr.p, max_leaf, pad, prims.p, keys_sorted.p,
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
