// Host runtime behind the C ABI of include/rgk.h.
//
//   * scene commit: the OUTPUTS of Scene::Commit the hot path reads (reference
//     src/scene.cpp:294-400): triangle planes, areal-light tables sorted by area,
//     light powers, epsilon = 1e-5 * bbox diagonal, epsilon-padded bbox;
//   * the build's own accelerator (binned-SAH BVH2 over pre-split references, collapsed to a
//     quantised 4-wide BVH with one 64-byte line per node) -- the reference's
//     kd-tree construction (scene.cpp:431-657) is out of scope, only its nearest-hit
//     semantics are kept;
//   * the round driver: tiles -> per-pixel seeds (a2) -> passes of paths resident in
//     HBM -> raygen / trace / shade / shadow / resolve launches on one HIP stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <iterator>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>
#include <functional>
#include <random>

#include "../../include/rgk.h"
#include "device_types.h"
#include "rgk_kernels.h"
#include "rgk_build.h"

namespace {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(x)                                                                                     \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess)                                                                         \
            return fail(e_ == hipErrorOutOfMemory ? RGK_ERR_OOM : RGK_ERR_DEVICE, "%s: %s (%s:%d)", #x, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                   \
    } while (0)

struct V3 {
    float x, y, z;
};
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 crossv(V3 x, V3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline float dotv(V3 a, V3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
inline V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 normv(V3 v) { return scale(v, 1.0f / std::sqrt(dotv(v, v))); }
inline float comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

// ------------------------------------------------------------------ BVH build
struct Prim {
    float bmin[3], bmax[3], c[3];
    uint32_t tri;
    uint32_t ref;  // this reference's number (prims are shuffled by the builders; the leaf order lists these)
    float pb[4];   // the piece of the triangle this reference stands for, as a box in the triangle's own coordinates: P = v0 + b (v1 - v0) + c (v2 - v0)
                   // with b in [pb[0], pb[1]], c in [pb[2], pb[3]] -- (0, 1, 0, 1): the whole triangle.  A moved triangle maps its pieces
                   // affinely, so rgk_scene_refit re-boxes a reference from these four numbers instead of from the whole triangle
};
struct Box {
    float mn[3], mx[3];
    void reset() { for (int i = 0; i < 3; i++) { mn[i] = std::numeric_limits<float>::infinity(); mx[i] = -mn[i]; } }
    void grow(const float* a, const float* b) { for (int i = 0; i < 3; i++) { mn[i] = std::min(mn[i], a[i]); mx[i] = std::max(mx[i], b[i]); } }
    void grow(const Box& o) { grow(o.mn, o.mx); }
    float area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.f;
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};

// Early split clipping of large triangles (reference-splitting before the build): a triangle whose box is longer
// than `lmax` on some axis enters the build as several references, one per piece of the triangle clipped at the
// box midpoint, each with the tight box of its piece.  The pieces tile the triangle, so every hit point lies in
// the (eps-padded) box of a reference; the leaf records are whole triangles, so a hit is what it was -- the tree
// just stops dragging wall- and floor-sized boxes through its upper levels.
struct RefSplitter {
    struct P3 { double x[3]; };
    float lmax;
    size_t budget; // extra references still allowed
    std::vector<Prim>* out;
    double tv[3][3]; // the triangle being split (for the pieces' parameter boxes)
    void param_box(const std::vector<P3>& poly, float pb[4]) const {
        double e1[3], e2[3], a11 = 0, a12 = 0, a22 = 0;
        for (int k = 0; k < 3; k++) { e1[k] = tv[1][k] - tv[0][k]; e2[k] = tv[2][k] - tv[0][k]; a11 += e1[k] * e1[k]; a12 += e1[k] * e2[k]; a22 += e2[k] * e2[k]; }
        const double det = a11 * a22 - a12 * a12;
        double b0 = 1, b1 = 0, c0 = 1, c1 = 0;
        if (!(det > 0)) { pb[0] = 0.f; pb[1] = 1.f; pb[2] = 0.f; pb[3] = 1.f; return; }
        for (const P3& v : poly) {
            double r1 = 0, r2 = 0;
            for (int k = 0; k < 3; k++) { const double w = v.x[k] - tv[0][k]; r1 += e1[k] * w; r2 += e2[k] * w; }
            const double b = (a22 * r1 - a12 * r2) / det, c = (a11 * r2 - a12 * r1) / det;
            b0 = std::min(b0, b); b1 = std::max(b1, b); c0 = std::min(c0, c); c1 = std::max(c1, c);
        }
        const double pad = 1e-5; // (the solve's rounding; the pieces overlap by this much)
        pb[0] = (float)std::max(0.0, b0 - pad); pb[1] = (float)std::min(1.0, b1 + pad); pb[2] = (float)std::max(0.0, c0 - pad); pb[3] = (float)std::min(1.0, c1 + pad);
    }
    static void clip(const std::vector<P3>& in, int ax, double plane, bool keep_below, std::vector<P3>& res) {
        res.clear();
        const size_t n = in.size();
        for (size_t i = 0; i < n; i++) {
            const P3 &a = in[i], &b = in[(i + 1) % n];
            const bool ia = keep_below ? a.x[ax] <= plane : a.x[ax] >= plane, ib = keep_below ? b.x[ax] <= plane : b.x[ax] >= plane;
            if (ia) res.push_back(a);
            if (ia != ib) {
                const double t = (plane - a.x[ax]) / (b.x[ax] - a.x[ax]);
                P3 m;
                for (int k = 0; k < 3; k++) m.x[k] = a.x[k] + t * (b.x[k] - a.x[k]);
                m.x[ax] = plane;
                res.push_back(m);
            }
        }
    }
    void emit(const std::vector<P3>& poly, const float* bmin, const float* bmax, uint32_t tri, int depth) {
        // tight box of the piece: polygon bounds (outward-rounded to float) within the parent's box
        Prim p;
        for (int a = 0; a < 3; a++) {
            double lo = std::numeric_limits<double>::infinity(), hi = -lo;
            for (const P3& v : poly) { lo = std::min(lo, v.x[a]); hi = std::max(hi, v.x[a]); }
            float fl = (float)lo, fh = (float)hi;
            if ((double)fl > lo) fl = std::nextafterf(fl, -std::numeric_limits<float>::infinity());
            if ((double)fh < hi) fh = std::nextafterf(fh, std::numeric_limits<float>::infinity());
            p.bmin[a] = std::max(fl, bmin[a]); p.bmax[a] = std::min(fh, bmax[a]);
            if (p.bmin[a] > p.bmax[a]) p.bmin[a] = p.bmax[a] = 0.5f * (bmin[a] + bmax[a]);
        }
        int ax = 0;
        for (int a = 1; a < 3; a++) if (p.bmax[a] - p.bmin[a] > p.bmax[ax] - p.bmin[ax]) ax = a;
        if (!(p.bmax[ax] - p.bmin[ax] > lmax) || depth >= 12 || budget == 0 || poly.size() < 3) {
            for (int a = 0; a < 3; a++) p.c[a] = 0.5f * (p.bmin[a] + p.bmax[a]);
            p.tri = tri;
            p.ref = (uint32_t)out->size();
            if (depth == 0) { p.pb[0] = 0.f; p.pb[1] = 1.f; p.pb[2] = 0.f; p.pb[3] = 1.f; } else param_box(poly, p.pb);
            out->push_back(p);
            return;
        }
        budget--;
        const double mid = 0.5 * ((double)p.bmin[ax] + (double)p.bmax[ax]);
        std::vector<P3> lo, hi;
        clip(poly, ax, mid, true, lo);
        clip(poly, ax, mid, false, hi);
        float cmax[3] = {p.bmax[0], p.bmax[1], p.bmax[2]}, cmin[3] = {p.bmin[0], p.bmin[1], p.bmin[2]};
        cmax[ax] = std::nextafterf((float)mid, std::numeric_limits<float>::infinity());
        cmin[ax] = std::nextafterf((float)mid, -std::numeric_limits<float>::infinity());
        if (lo.size() >= 3) emit(lo, p.bmin, cmax, tri, depth + 1);
        if (hi.size() >= 3) emit(hi, cmin, p.bmax, tri, depth + 1);
    }
};

struct BvhBuilder {
    std::vector<Prim>& prims;
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> order; // reference numbers (Prim::ref) in leaf order
    uint32_t max_depth = 0;
    float pad;
    static constexpr int NBINS = 16;
    int MAX_LEAF = 4;                      // leaf encoding allows up to 16
    float C_TRAV = 1.0f, C_ISECT = 1.0f;   // SAH: one node step vs one triangle test (swept on MI355X: 1.0 best)
    BvhBuilder(std::vector<Prim>& p, float pad_) : prims(p), pad(pad_) {
        if (const char* e = getenv("RGK_BVH_MAXLEAF")) MAX_LEAF = std::min(16, std::max(1, atoi(e)));
        if (const char* e = getenv("RGK_BVH_CISECT")) C_ISECT = (float)atof(e);
    }

    int make_leaf(size_t b, size_t e) {
        uint32_t first = order.size();
        for (size_t i = b; i < e; i++) order.push_back(prims[i].ref);
        uint32_t cnt = (uint32_t)(e - b);
        return (int)~((first << 4) | (cnt - 1));
    }
    // returns the child code for prims[b,e) and its (padded) box
    int build(size_t b, size_t e, uint32_t depth, Box& box) {
        max_depth = std::max(max_depth, depth);
        box.reset();
        Box cb;
        cb.reset();
        for (size_t i = b; i < e; i++) { box.grow(prims[i].bmin, prims[i].bmax); cb.grow(prims[i].c, prims[i].c); }
        size_t n = e - b;
        size_t mid = 0;
        bool leaf = (n == 1);
        if (!leaf) {
            float best = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_bin = -1;
            float parent_area = box.area();
            for (int ax = 0; ax < 3; ax++) {
                float lo = cb.mn[ax], hi = cb.mx[ax];
                if (!(hi > lo)) continue;
                Box bb[NBINS];
                uint32_t cnt[NBINS] = {0};
                for (auto& x : bb) x.reset();
                float k = NBINS / (hi - lo);
                for (size_t i = b; i < e; i++) {
                    int bi = std::min(NBINS - 1, std::max(0, (int)((prims[i].c[ax] - lo) * k)));
                    cnt[bi]++;
                    bb[bi].grow(prims[i].bmin, prims[i].bmax);
                }
                float ra[NBINS];
                uint32_t rc[NBINS];
                Box acc;
                acc.reset();
                uint32_t c = 0;
                for (int i = NBINS - 1; i > 0; i--) { acc.grow(bb[i]); c += cnt[i]; ra[i] = acc.area(); rc[i] = c; }
                acc.reset();
                c = 0;
                for (int i = 0; i < NBINS - 1; i++) {
                    acc.grow(bb[i]);
                    c += cnt[i];
                    if (c == 0 || rc[i + 1] == 0) continue;
                    float cost = C_TRAV + C_ISECT * (acc.area() * c + ra[i + 1] * rc[i + 1]) / std::max(parent_area, 1e-30f);
                    if (cost < best) { best = cost; best_axis = ax; best_bin = i; }
                }
            }
            if (best_axis >= 0 && (n > (size_t)MAX_LEAF || best < C_ISECT * n)) {
                float lo = cb.mn[best_axis], hi = cb.mx[best_axis];
                float k = NBINS / (hi - lo);
                auto it = std::partition(prims.begin() + b, prims.begin() + e, [&](const Prim& p) {
                    int bi = std::min(NBINS - 1, std::max(0, (int)((p.c[best_axis] - lo) * k)));
                    return bi <= best_bin;
                });
                mid = it - prims.begin();
                if (mid == b || mid == e) best_axis = -1;
            } else if (best_axis >= 0) {
                leaf = true; // SAH prefers a leaf and it fits
                best_axis = 0;
            }
            if (!leaf && best_axis < 0) {
                if (n <= (size_t)MAX_LEAF) leaf = true;
                else { // coincident centroids: median split by index
                    int ax = 0;
                    float ex = -1;
                    for (int a = 0; a < 3; a++) if (box.mx[a] - box.mn[a] > ex) { ex = box.mx[a] - box.mn[a]; ax = a; }
                    mid = b + n / 2;
                    std::nth_element(prims.begin() + b, prims.begin() + mid, prims.begin() + e,
                                     [ax](const Prim& p, const Prim& q) { return p.c[ax] < q.c[ax]; });
                }
            }
        }
        for (int i = 0; i < 3; i++) { box.mn[i] -= pad; box.mx[i] += pad; }
        if (leaf) return make_leaf(b, e);
        int idx = (int)nodes.size();
        nodes.emplace_back();
        Box lb, rb;
        int l = build(b, mid, depth + 1, lb);
        int r = build(mid, e, depth + 1, rb);
        BvhNode& nd = nodes[idx];
        for (int i = 0; i < 3; i++) { nd.lmin[i] = lb.mn[i]; nd.lmax[i] = lb.mx[i]; nd.rmin[i] = rb.mn[i]; nd.rmax[i] = rb.mx[i]; }
        nd.left = l; nd.right = r; nd.pad[0] = nd.pad[1] = 0;
        return idx;
    }
};

// ------------------------------------------------------------------ BVH2 optimisation by reinsertion
// The binned top-down build decides every split with local information; afterwards single subtrees are taken out and put back
// where the surface-area cost of the whole tree grows least (insertion-based optimisation, Bittner, Hapala, Havran 2013, in its
// simplest form: the largest nodes first, branch-and-bound search from the root).  Same triangles, same leaves, so the same
// hits; on the Sponza proxy 8 rounds over half of the nodes cut the surface-area cost by 4 % and the node visits per ray by
// 4 % (diffuse bounce rays) to 9 % (camera rays) -- tools/probe_wide_bvh.py measures it on the CPU.
static void optimise_bvh2(std::vector<BvhNode>& nodes, std::vector<uint32_t>& order, int rounds, float frac) {
    const int NI = (int)nodes.size();
    if (NI < 8 || rounds <= 0) return;
    std::vector<Box> box; std::vector<int> l, r, par, leaf_code;
    box.reserve(2 * NI + 1); l.assign(NI, -1); r.assign(NI, -1);
    box.resize(NI);
    auto side_box = [](const BvhNode& n, bool left) { Box b; for (int a = 0; a < 3; a++) { b.mn[a] = left ? n.lmin[a] : n.rmin[a]; b.mx[a] = left ? n.lmax[a] : n.rmax[a]; } return b; };
    for (int i = 0; i < NI; i++) {
        for (int sd = 0; sd < 2; sd++) {
            const int code = sd == 0 ? nodes[i].left : nodes[i].right;
            const Box b = side_box(nodes[i], sd == 0);
            int id;
            if (code >= 0) { id = code; box[id] = b; }
            else { id = (int)box.size(); box.push_back(b); l.push_back(-1); r.push_back(-1); leaf_code.resize(box.size(), 0); leaf_code[id] = code; }
            (sd == 0 ? l[i] : r[i]) = id;
        }
    }
    leaf_code.resize(box.size(), 0);
    const int N = (int)box.size();
    par.assign(N, -1);
    for (int i = 0; i < NI; i++) { par[l[i]] = i; par[r[i]] = i; }
    box[0] = box[l[0]]; box[0].grow(box[r[0]]);
    auto refit_up = [&](int n) { while (n >= 0) { Box b = box[l[n]]; b.grow(box[r[n]]); box[n] = b; n = par[n]; } };
    std::mt19937 rng(7);
    const auto heap_cmp = [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first > b.first; };
    std::vector<std::pair<float, int>> pq;
    for (int it = 0; it < rounds; it++) {
        std::vector<int> cand;
        const auto larger = [&](int a, int b) { const float x = box[a].area(), y = box[b].area(); return x > y || (x == y && a < b); };
        if (!(it & 1)) { // the largest nodes (they cost the most); bounded work per round: a 1 M-triangle tree moves its largest nodes only
            for (int i = 1; i < N; i++) if (par[i] > 0) cand.push_back(i);
            const size_t take = std::min<size_t>((size_t)(cand.size() * frac), 65536);
            std::nth_element(cand.begin(), cand.begin() + take, cand.end(), larger);
            cand.resize(take);
            std::sort(cand.begin(), cand.end(), larger);
        } else { // every other round: any nodes
            const size_t take = std::min<size_t>((size_t)(N * frac), 65536);
            for (size_t k = 0; k < take; k++) { const int i = (int)(rng() % (uint32_t)N); if (par[i] > 0) cand.push_back(i); }
        }
        for (int n : cand) {
            const int p = par[n];
            if (p <= 0) continue;
            const int g = par[p];
            const int sib = l[p] == n ? r[p] : l[p];
            (l[g] == p ? l[g] : r[g]) = sib; // n and its parent leave the tree: the sibling moves up
            par[sib] = g;
            refit_up(g);
            const Box nb = box[n];
            const float na = nb.area();
            float best = std::numeric_limits<float>::infinity();
            int bx = sib;
            pq.clear(); pq.push_back({0.f, 0});
            while (!pq.empty()) {
                std::pop_heap(pq.begin(), pq.end(), heap_cmp);
                const float ind = pq.back().first; const int x = pq.back().second;
                pq.pop_back();
                if (ind + na >= best) break;
                Box u = box[x]; u.grow(nb);
                const float total = ind + u.area();
                if (total < best && par[x] >= 0) { best = total; bx = x; } // (not above the root: node 0 stays the root)
                const float child_ind = total - box[x].area();
                if (l[x] >= 0 && child_ind + na < best) {
                    pq.push_back({child_ind, l[x]}); std::push_heap(pq.begin(), pq.end(), heap_cmp);
                    pq.push_back({child_ind, r[x]}); std::push_heap(pq.begin(), pq.end(), heap_cmp);
                }
            }
            const int xp = par[bx]; // p becomes the parent of (bx, n) where bx was
            (l[xp] == bx ? l[xp] : r[xp]) = p;
            par[p] = xp; l[p] = bx; r[p] = n; par[bx] = p; par[n] = p;
            refit_up(p);
        }
    }
    // back to the builder's form: inner nodes in depth-first order from node 0, leaves re-listed in that order
    std::vector<BvhNode> out; out.reserve(NI);
    std::vector<uint32_t> new_order; new_order.reserve(order.size());
    std::function<int(int)> emit = [&](int n) -> int {
        if (l[n] < 0) {
            const uint32_t code = ~(uint32_t)leaf_code[n], first = code >> 4, cnt = (code & 15u) + 1u;
            const uint32_t nf = (uint32_t)new_order.size();
            for (uint32_t k = 0; k < cnt; k++) new_order.push_back(order[first + k]);
            return (int)~((nf << 4) | (cnt - 1));
        }
        const int idx = (int)out.size();
        out.emplace_back();
        const int a = emit(l[n]), b = emit(r[n]);
        BvhNode& nd = out[idx];
        for (int k = 0; k < 3; k++) { nd.lmin[k] = box[l[n]].mn[k]; nd.lmax[k] = box[l[n]].mx[k]; nd.rmin[k] = box[r[n]].mn[k]; nd.rmax[k] = box[r[n]].mx[k]; }
        nd.left = a; nd.right = b; nd.pad[0] = nd.pad[1] = 0;
        return idx;
    };
    emit(0);
    nodes.swap(out);
    order.swap(new_order);
}

// ------------------------------------------------------------------ BVH2 -> quantised BVH4
// Collapse the binary tree (always open the inner child with the largest surface until four
// children) and quantise each child box to 8 bits per plane relative to the node's box, rounding
// outward and re-checking the decode in float exactly as the kernel evaluates it.
struct QbvhBuilder {
    const std::vector<BvhNode>& bn;
    std::vector<QNode> out;
    uint32_t max_stack = 0, max_depth = 0;
    explicit QbvhBuilder(const std::vector<BvhNode>& b) : bn(b) {}
    struct Child { int ref; Box box; };

    static bool valid(const Box& b) { return b.mn[0] <= b.mx[0] && b.mn[1] <= b.mx[1] && b.mn[2] <= b.mx[2]; }
    void children_of(int node, Child& l, Child& r) const {
        const BvhNode& n = bn[node];
        l.ref = n.left; r.ref = n.right;
        for (int a = 0; a < 3; a++) { l.box.mn[a] = n.lmin[a]; l.box.mx[a] = n.lmax[a]; r.box.mn[a] = n.rmin[a]; r.box.mx[a] = n.rmax[a]; }
    }
    // `stack_before`: entries a traversal may already hold when it reaches this node
    int collapse(int node, uint32_t depth, uint32_t stack_before) {
        std::vector<Child> ch(2);
        children_of(node, ch[0], ch[1]);
        ch.erase(std::remove_if(ch.begin(), ch.end(), [](const Child& c) { return !valid(c.box); }), ch.end());
        while (ch.size() < 4) {
            int best = -1;
            float best_area = -1.f;
            for (size_t i = 0; i < ch.size(); i++)
                if (ch[i].ref >= 0 && ch[i].box.area() > best_area) { best_area = ch[i].box.area(); best = (int)i; }
            if (best < 0) break;
            Child a, b;
            children_of(ch[best].ref, a, b);
            ch.erase(ch.begin() + best);
            if (valid(a.box)) ch.push_back(a);
            if (valid(b.box)) ch.push_back(b);
        }
        int idx = (int)out.size();
        out.emplace_back();
        max_depth = std::max(max_depth, depth);
        const uint32_t pushed = (uint32_t)ch.size() - 1;
        max_stack = std::max(max_stack, stack_before + pushed);
        Box nb;
        nb.reset();
        for (auto& c : ch) nb.grow(c.box);
        QNode q;
        std::memset(&q, 0, sizeof(q));
        for (int a = 0; a < 3; a++) {
            q.p[a] = nb.mn[a];
            float ext = nb.mx[a] - nb.mn[a];
            int e = 0;
            if (ext > 0.f) { std::frexp(ext / 255.0f, &e); } else e = -126;
            for (;; e++) { // find the smallest exponent whose outward-rounded codes all fit and verify
                if (e < -126) e = -126;
                const float scale = std::ldexp(1.0f, e);
                bool ok = true;
                uint8_t lo[4], hi[4];
                for (size_t i = 0; i < ch.size() && ok; i++) {
                    float fl = std::floor((ch[i].box.mn[a] - q.p[a]) / scale), fh = std::ceil((ch[i].box.mx[a] - q.p[a]) / scale);
                    if (fl < 0.f) fl = 0.f;
                    while (fl > 0.f && std::fmaf(fl, scale, q.p[a]) > ch[i].box.mn[a]) fl -= 1.f;
                    while (fh <= 255.f && std::fmaf(fh, scale, q.p[a]) < ch[i].box.mx[a]) fh += 1.f;
                    if (fh > 255.f || fl > 255.f) { ok = false; break; }
                    lo[i] = (uint8_t)fl; hi[i] = (uint8_t)fh;
                }
                if (!ok) continue;
                (a == 0 ? q.sx : (a == 1 ? q.sy : q.sz)) = scale;
                for (size_t i = 0; i < 4; i++) { q.qlo[a][i] = i < ch.size() ? lo[i] : 255; q.qhi[a][i] = i < ch.size() ? hi[i] : 0; }
                break;
            }
        }
        for (size_t i = 0; i < 4; i++) q.child[i] = RGK_QNODE_EMPTY;
        out[idx] = q;
        for (size_t i = 0; i < ch.size(); i++) {
            int ref = ch[i].ref;
            if (ref >= 0) ref = collapse(ref, depth + 1, stack_before + pushed);
            out[idx].child[i] = ref;
        }
        return idx;
    }
};

// ------------------------------------------------------------------ scene object
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int alloc(size_t count) {
        if (count <= n && p) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e != hipSuccess) return fail(RGK_ERR_OOM, "hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
        n = count;
        // RGK_POISON=1: every fresh device buffer is filled with 0xFF bytes (NaN as float, huge as index), so that any read of
        // memory the pipeline did not write first shows in the results instead of hiding behind zero-filled fresh pages
        static const bool poison = std::getenv("RGK_POISON") != nullptr;
        if (poison) { // (the fill runs on the null stream, the scene's stream is non-blocking: wait for it, or it lands on top of real data)
            (void)hipMemset(p, 0xFF, count * sizeof(T));
            (void)hipDeviceSynchronize();
        }
        return 0;
    }
    int upload(const std::vector<T>& v) {
        int rc = alloc(v.size());
        if (rc) return rc;
        if (!v.empty()) {
            hipError_t e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
            if (e != hipSuccess) return fail(RGK_ERR_DEVICE, "hipMemcpy H2D: %s", hipGetErrorString(e));
        }
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

} // namespace

extern "C" __attribute__((visibility("hidden"))) int rgk_internal_fail(int code, const char* msg) { return fail(code, "%s", msg); }

// Tuning switches of one scene (nothing here changes a result).  Filled ONCE, in rgk_scene_create, from the environment
// (RGK_ENTRY_POINTS, RGK_ENTRY_CAP, RGK_LIGHT_ENTRY, RGK_SAMPLE_GROUP, RGK_BATCH_PATHS, RGK_WORKSPACE_GB, RGK_DEBUG_BVH,
// RGK_DEBUG_UTIL); afterwards only rgk_scene_set_tuning changes them -- a round never reads the environment (round 2 did, per
// round: process-global state under a host that may render from two threads).
struct RgkTuning {
    bool entry_points = true; // camera rays start at their pixel group's entry nodes (k_entry_points)
    bool entry_cap = true;    // ... capped behind the group's first hits from a frame's second round on
    bool light_entry = true;  // first-vertex shadow rays of a single-light scene start at light-side entry nodes
    int sample_group = -1;    // log2 of the samples of a pixel side by side in the slot order; -1: the compiled default
    size_t batch_paths = 0;   // paths per pass; 0: sized from the memory that is free
    double workspace_gb = 0;  // ... or from this many GB; 0: 96 (160 for bidirectional rounds), at most 60 % of what is free
    int beam = 1;             // pinhole cameras: bounce 0 walks the tree once per pixel for 8 samples (k_trace_camera_beam) -- 1: while the
                              // entry lists are uncapped (a frame's first round), 2: always, 0: never
    bool two_lanes = false;   // experiment (measured: no gain, see render_round): the two halves of the pixel list as two passes on two streams
    bool debug_bvh = false, debug_util = false;
};

struct rgk_scene {
    int device = 0;
    RgkTuning tune;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // second lane of a round (two halves of the pixel list side by side: one's launch tails under the other's launches)
    hipEvent_t ev_prelude = nullptr; // the round's lists and tables are in place (stream2 waits for it)
    rgk_scene_info info{};
    DevScene dev{};
    RgkTraceCfg tcfg{32, 32, nullptr};
    DevBuf<int> ovf; // traversal-stack overflow area (deep trees), one per lane
    size_t ovf_lane = 0;
    // scene data
    DevBuf<QNode> nodes;
    DevBuf<TriIsect> tris;
    DevBuf<TriShade> tri_shade;
    DevBuf<DevMaterial> materials;
    DevBuf<float4> texels;
    DevBuf<uint32_t> texels8;
    DevBuf<float> luts;
    DevBuf<DevPointLight> pointlights;
    DevBuf<DevArealLight> areal;
    DevBuf<DevArealTri> areal_tris;
    DevBuf<float4> ltc;
    DevBuf<DevScene> self;    // device-resident copy of `dev` (DevScene::self)
    DevBuf<uint32_t> generic; // queue indices left to the generic-BxDF shade launch
    DevBuf<TexRef> texrefs;   // one TexRef per descriptor texture (rgk_texture_sample)
    // what rgk_scene_refit needs of the descriptor after rgk_scene_create has returned
    DevBuf<uint32_t> d_idx;   // tri_indices on the device
    DevBuf<float4> leaf_pb;   // per leaf reference: its piece of the triangle in the triangle's own coordinates (Prim::pb)
    std::vector<uint32_t> h_idx, h_tri_mat, h_areal_off, h_areal_tris;
    std::vector<rgk_material> h_mats;
    std::vector<float> h_normals;
    uint32_t n_vertices = 0, n_triangles = 0, n_refs = 0, n_nodes = 0;
    uint32_t n_textures = 0, n_materials = 0;
    DevBuf<DevHaltonDim> hdims;
    DevBuf<uint16_t> hperm;
    // workspace
    size_t batch = 0;
    DevBuf<float4> rayA[2], rayB[2], hit, thr, tot, shA, shB, shC, pixsum, light;
    DevBuf<float4> lstart, lv; // bidirectional state (reverse > 0)
    DevBuf<uint32_t> hitlist;  // ... light sub-path: queue indices of the rays that hit (k_list_hits)
    DevBuf<uint32_t> lvmask, connlist; // ... which light vertices a slot has; queue indices of the camera vertices with connections
    DevBuf<float4> conn, jobs, rads;   // ... their records (k_shade<BDPT> -> k_connect) and the vertex queue of k_trace_shadow_jobs
    uint32_t batch_reverse = 0;
    DevBuf<float> htab;
    DevBuf<float2> nearfar;
    DevBuf<uint32_t> counters, pix_xy, pix_seed, tile_buf;
    DevBuf<int> entry; // RGK_ENTRY_K traversal entry nodes per group of RGK_ENTRY_PIX pixels of the round's list
    DevBuf<float> entry_cap;  // ... and up to which distance each list is complete (k_entry_points)
    size_t entry_capped = 0;  // pixels of the round's list whose lists have been rebuilt with this frame's first-hit distances
    DevBuf<int> lentry;       // the same for the first vertex's shadow rays (single-light scenes), rebuilt per pass
    DevBuf<uint32_t> trange;  // per pixel group: nearest / farthest first hit of a block of samples (float bits)
    DevBuf<float4> lbox;      // per pixel group: the box its light-side entry nodes are good for
    size_t lentry_done = 0;   // how many pixels of the round's list the light-side entries of this frame cover so far
    uint64_t entry_key = 0; // camera + tile geometry they were made for
    size_t entry_n = 0;
    DevBuf<unsigned long long> stats;
    DevBuf<float> scratch_f;
    DevBuf<uint32_t> scratch_u;
    std::vector<hipEvent_t> events;
    uint32_t* h_counters = nullptr; // pinned
    // progress, read by rgk_scene_get_progress from any thread
    std::atomic<uint32_t> prog_stages{0}, prog_rounds{0}, prog_busy{0};
    uint32_t* h_stage = nullptr; // pinned word the DEVICE writes (k_stage_mark): stages of the running round that are done
    std::atomic<uint64_t> prog_pixels{0}, prog_paths{0};
    ~rgk_scene() {
        (void)hipSetDevice(device);
        for (auto e : events) (void)hipEventDestroy(e);
        if (h_counters) (void)hipHostFree(h_counters);
        if (h_stage) (void)hipHostFree(h_stage);
        nodes.release(); tris.release(); tri_shade.release(); materials.release(); texels8.release(); luts.release();
        texels.release(); pointlights.release(); areal.release(); areal_tris.release(); ltc.release(); self.release(); ovf.release();
        hdims.release(); hperm.release(); texrefs.release(); d_idx.release(); leaf_pb.release();
        for (int i = 0; i < 2; i++) { rayA[i].release(); rayB[i].release(); }
        hit.release(); thr.release(); tot.release(); shA.release(); shB.release(); shC.release(); pixsum.release();
        light.release(); generic.release(); htab.release(); lstart.release(); lv.release(); hitlist.release(); lvmask.release(); connlist.release(); conn.release(); jobs.release(); rads.release();
        nearfar.release(); counters.release(); pix_xy.release(); pix_seed.release(); tile_buf.release(); stats.release(); entry.release(); entry_cap.release(); lentry.release(); trange.release(); lbox.release();
        scratch_f.release(); scratch_u.release();
        if (ev_prelude) (void)hipEventDestroy(ev_prelude);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

int ensure_workspace(rgk_scene* s, size_t paths, uint32_t reverse = 0) {
    if (paths <= s->batch && reverse <= s->batch_reverse) return 0;
    if (s->batch) { paths = std::max(paths, s->batch); reverse = std::max(reverse, s->batch_reverse); }
    int rc = 0;
    for (int i = 0; i < 2 && !rc; i++) { rc = s->rayA[i].alloc(paths); if (!rc) rc = s->rayB[i].alloc(paths); }
    if (!rc) rc = s->hit.alloc(paths);
    if (!rc) rc = s->thr.alloc(paths);
    if (!rc) rc = s->tot.alloc(paths);
    // plain shadow queue: one ray {shA, shB, shC} per path and bounce
    if (!rc) rc = s->shA.alloc(paths);
    if (!rc) rc = s->shB.alloc(paths);
    if (!rc) rc = s->shC.alloc(paths);
    if (reverse) {
        if (!rc) rc = s->lstart.alloc(paths);
        if (!rc) rc = s->lv.alloc(paths * RGK_LV_FLOAT4 * reverse);
        if (!rc) rc = s->hitlist.alloc(paths);
        if (!rc) rc = s->lvmask.alloc(paths);
        if (!rc) rc = s->connlist.alloc(paths);
        if (!rc) rc = s->conn.alloc(paths * 6);                  // the record a camera vertex with connections leaves
        if (!rc) rc = s->jobs.alloc(paths * 4);                  // vertex queue: {vertex}{contribution, mask}{emission}{NEE ray start}
        if (!rc) rc = s->rads.alloc(paths * ((size_t)reverse + 1)); // ... and one radiance per ray
    }
    if (!rc) rc = s->light.alloc(paths);
    if (!rc) rc = s->generic.alloc(paths);
    if (!rc) rc = s->counters.alloc(4 * RGK_CNT_TOTAL); // per lane: [0] camera phase, [1] light sub-path phase
    if (!rc) rc = s->stats.alloc(8);
    if (rc) { s->batch = 0; s->batch_reverse = 0; return rc; } // some buffers are gone: the next call starts over
    if (!s->h_counters) HIPCHK(hipHostMalloc((void**)&s->h_counters, 4 * RGK_CNT_TOTAL * sizeof(uint32_t)));
    if (!s->h_stage) { HIPCHK(hipHostMalloc((void**)&s->h_stage, 2 * sizeof(uint32_t))); s->h_stage[0] = s->h_stage[1] = 0; }
    s->batch = paths;
    s->batch_reverse = reverse;
    return 0;
}

void build_halton(std::vector<DevHaltonDim>& dims, std::vector<uint16_t>& perm) {
    // Faure permutations: the standard recursive construction the reference uses
    // (external/halton_sampler.h:574-604), one permutation per prime base.
    const unsigned max_base = 1619u;
    std::vector<std::vector<uint16_t>> perms(max_base + 1);
    for (unsigned k = 1; k <= 3; ++k) { perms[k].resize(k); for (unsigned i = 0; i < k; ++i) perms[k][i] = i; }
    for (unsigned base = 4; base <= max_base; ++base) {
        perms[base].resize(base);
        unsigned b = base / 2;
        if (base & 1) {
            for (unsigned i = 0; i + 1 < base; ++i) {
                uint16_t v = perms[base - 1][i];
                perms[base][i + (i >= b)] = v + (v >= b);
            }
            perms[base][b] = b;
        } else {
            for (unsigned i = 0; i < b; ++i) { perms[base][i] = 2 * perms[b][i]; perms[base][b + i] = 2 * perms[b][i] + 1; }
        }
    }
    for (unsigned p = 2; dims.size() < 256; p++) {
        bool prime = true;
        for (unsigned d = 2; d * d <= p; d++) if (p % d == 0) { prime = false; break; }
        if (!prime) continue;
        DevHaltonDim hd{};
        hd.base = p;
        uint64_t bk = p; unsigned k = 1;
        while (bk * p <= 500) { bk *= p; k++; }  // digits per table lookup in the reference
        uint64_t tot = bk; unsigned G = 1;
        while (tot * bk < (1ull << 32)) { tot *= bk; G++; } // lookups per sample
        hd.digits = k * G;
        hd.scale = float(0x1.fffffcp-1 / (double)tot);
        hd.perm_off = (uint32_t)perm.size();
        if (p > 2) { // exact u32 division by p: q = (t + ((n - t) >> 1)) >> (l - 1), t = mulhi(m, n)
            unsigned l = 0;
            while ((1u << l) < p) l++;
            hd.magic = (uint32_t)(((1ull << 32) * ((1ull << l) - p)) / p + 1);
            hd.shift = l - 1;
        }
        perm.insert(perm.end(), perms[p].begin(), perms[p].end());
        dims.push_back(hd);
    }
}

int validate_desc(const rgk_scene_desc* d) {
    if (!d) return fail(RGK_ERR_INVALID, "null scene descriptor");
    if (d->n_triangles == 0 || d->n_vertices == 0) return fail(RGK_ERR_INVALID, "scene has no geometry");
    if (!d->vertices || !d->normals || !d->tangents || !d->tri_indices || !d->tri_material)
        return fail(RGK_ERR_INVALID, "null geometry pointer");
    if (d->n_materials == 0 || !d->materials) return fail(RGK_ERR_INVALID, "scene has no materials");
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        for (int k = 0; k < 3; k++)
            if (d->tri_indices[3 * i + k] >= d->n_vertices) return fail(RGK_ERR_INVALID, "triangle %u: vertex index out of range", i);
        if (d->tri_material[i] >= d->n_materials) return fail(RGK_ERR_INVALID, "triangle %u: material index out of range", i);
    }
    bool ggx = false, bek = false;
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rgk_material& m = d->materials[i];
        if (m.kind > RGK_BXDF_LTC_GGX_DIFFUSE) return fail(RGK_ERR_INVALID, "material %u: unknown bxdf kind %u", i, m.kind);
        const int32_t t[3] = {m.tex_diffuse, m.tex_color, m.tex_bump};
        for (int k = 0; k < 3; k++)
            if (t[k] >= (int32_t)d->n_textures) return fail(RGK_ERR_INVALID, "material %u: texture index out of range", i);
        if (m.kind == RGK_BXDF_MIX && (m.mix_m1 < 0 || m.mix_m2 < 0 || m.mix_m1 >= (int32_t)d->n_materials || m.mix_m2 >= (int32_t)d->n_materials))
            return fail(RGK_ERR_INVALID, "material %u: mix children out of range", i);
        if (m.kind == RGK_BXDF_LTC_GGX || m.kind == RGK_BXDF_LTC_GGX_DIFFUSE) ggx = true;
        if (m.kind == RGK_BXDF_LTC_BECKMANN || m.kind == RGK_BXDF_LTC_BECKMANN_DIFFUSE) bek = true;
    }
    {   // BxDFMix recurses (bxdf.cpp:235-249); the kernels evaluate a mix of mixes of leaves (two levels) without recursion.
        // Anything deeper, or a mix that reaches itself, is refused here rather than rendered wrong.
        std::vector<int> depth(d->n_materials, -1); // -1 unvisited, -2 on the current walk
        struct Walk {
            const rgk_scene_desc* d; std::vector<int>& depth;
            int go(uint32_t i) {
                if (d->materials[i].kind != RGK_BXDF_MIX) return depth[i] = 0;
                if (depth[i] == -2) return -1; // cycle
                if (depth[i] >= 0) return depth[i];
                depth[i] = -2;
                const int a = go((uint32_t)d->materials[i].mix_m1), b = go((uint32_t)d->materials[i].mix_m2);
                if (a < 0 || b < 0) return -1;
                return depth[i] = 1 + std::max(a, b);
            }
        } walk{d, depth};
        for (uint32_t i = 0; i < d->n_materials; i++) {
            const int k = walk.go(i);
            if (k < 0) return fail(RGK_ERR_INVALID, "material %u: mix materials form a cycle", i);
            if (k > 2) return fail(RGK_ERR_UNSUPPORTED, "material %u: mix nested %d levels deep (at most 2 are evaluated)", i, k);
        }
    }
    if (ggx && !d->ltc_ggx) return fail(RGK_ERR_INVALID, "LTC GGX material without ltc_ggx table");
    if (bek && !d->ltc_beckmann) return fail(RGK_ERR_INVALID, "LTC Beckmann material without ltc_beckmann table");
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rgk_texture& t = d->textures[i];
        if (t.kind == RGK_TEX_RGB32F && (!t.texels || t.width == 0 || t.height == 0)) return fail(RGK_ERR_INVALID, "texture %u: empty image", i);
        if (t.kind == RGK_TEX_RGB8 && (!t.texels8 || !t.lut || t.width == 0 || t.height == 0)) return fail(RGK_ERR_INVALID, "texture %u: empty 8-bit image", i);
        if (t.kind > RGK_TEX_RGB8) return fail(RGK_ERR_INVALID, "texture %u: unknown kind", i);
    }
    for (uint32_t i = 0; i < d->n_areal_lights; i++)
        for (uint32_t j = d->areal_offsets[i]; j < d->areal_offsets[i + 1]; j++)
            if (d->areal_tris[j] >= d->n_triangles) return fail(RGK_ERR_INVALID, "areal light %u: triangle out of range", i);
    if (d->sky_mode == RGK_SKY_ENVMAP && (d->sky_texture < 0 || d->sky_texture >= (int32_t)d->n_textures))
        return fail(RGK_ERR_INVALID, "sky envmap texture out of range");
    return 0;
}

} // namespace

namespace {
// host arrays -> device scratch, one call
template <typename T>
int up(DevBuf<T>& b, const T* src, size_t count) {
    int rc = b.alloc(count);
    if (rc) return rc;
    if (hipMemcpy(b.p, src, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return fail(RGK_ERR_DEVICE, "hipMemcpy H2D failed");
    return 0;
}
template <typename T>
int down(T* dst, const DevBuf<T>& b, size_t count) {
    if (hipMemcpy(dst, b.p, count * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) return fail(RGK_ERR_DEVICE, "hipMemcpy D2H failed");
    return 0;
}
} // namespace

extern "C" {

const char* rgk_last_error(void) { return g_err.c_str(); }

int rgk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// RGK_DEBUG_DESC=1: one line per table of the descriptor with an FNV-1a digest of its bytes, on stderr -- lets a host binding
// be checked against a known-good one ("did my flattening hand over the same scene?") without a debugger.
static void debug_desc(const rgk_scene_desc* d) {
    auto h = [](const void* p, size_t n) { uint64_t x = 1469598103934665603ull; const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; p && i < n; i++) { x ^= b[i]; x *= 1099511628211ull; } return (unsigned long long)x; };
    std::fprintf(stderr, "[rgk desc] vertices %u %016llx normals %016llx tangents %016llx texcoords %016llx\n", d->n_vertices, h(d->vertices, 12ull * d->n_vertices),
                 h(d->normals, 12ull * d->n_vertices), h(d->tangents, 12ull * d->n_vertices), h(d->texcoords, 8ull * d->n_vertices));
    std::fprintf(stderr, "[rgk desc] triangles %u idx %016llx mat %016llx\n", d->n_triangles, h(d->tri_indices, 12ull * d->n_triangles), h(d->tri_material, 4ull * d->n_triangles));
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rgk_material& m = d->materials[i];
        std::fprintf(stderr, "[rgk desc] material %u kind %u flags %u emission %g %g %g rough %.9g ior %.9g amount %.9g tex %d %d %d mix %d %d\n", i, m.kind, m.flags, m.emission[0], m.emission[1],
                     m.emission[2], m.roughness, m.ior, m.amount, m.tex_diffuse, m.tex_color, m.tex_bump, m.mix_m1, m.mix_m2);
    }
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rgk_texture& t = d->textures[i];
        const size_t n = (size_t)t.width * t.height;
        std::fprintf(stderr, "[rgk desc] texture %u kind %u %ux%u color %.9g %.9g %.9g texels %016llx\n", i, t.kind, t.width, t.height, t.color[0], t.color[1], t.color[2],
                     t.kind == RGK_TEX_RGB32F ? h(t.texels, 12 * n) : (t.kind == RGK_TEX_RGB8 ? h(t.texels8, 3 * n) ^ h(t.lut, 1024) : 0ull));
    }
    std::fprintf(stderr, "[rgk desc] pointlights %u %016llx areal %u offsets %016llx tris %016llx\n", d->n_pointlights, h(d->pointlights, sizeof(rgk_pointlight) * (size_t)d->n_pointlights),
                 d->n_areal_lights, h(d->areal_offsets, 4ull * (d->n_areal_lights + 1)), h(d->areal_tris, d->n_areal_lights ? 4ull * d->areal_offsets[d->n_areal_lights] : 0));
    std::fprintf(stderr, "[rgk desc] sky mode %u color %.9g %.9g %.9g intensity %.9g rotate %.9g tex %d ltc %016llx %016llx\n", d->sky_mode, d->sky_color[0], d->sky_color[1], d->sky_color[2],
                 d->sky_intensity, d->sky_rotate, d->sky_texture, h(d->ltc_ggx, 4096 * 20), h(d->ltc_beckmann, 4096 * 20));
}

#ifndef RGK_SAMPLE_GROUP_DEFAULT
#define RGK_SAMPLE_GROUP_DEFAULT 3 // 8 samples of a pixel side by side: swept 0..6 on the Sponza proxy (154.2, -, 149.6, 148.4, 148.1, 150.3, 150.0 ms per round)
#endif
// Scene::Commit's areal-light tables (src/scene.cpp:323-344): per emissive object its triangles sorted by area (descending), the
// total area, power = area * (r + g + b).  Used by rgk_scene_create and, for moved vertices, by rgk_scene_refit.
static void build_areal_tables(const float* vertices, const float* normals, const uint32_t* tri_indices, const uint32_t* tri_material, const rgk_material* materials,
                               uint32_t n_areal, const uint32_t* areal_offsets, const uint32_t* areal_tris, std::vector<DevArealLight>& als,
                               std::vector<DevArealTri>& ats, float& total_areal) {
    auto vert = [&](uint32_t i) { return V3{vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]}; };
    als.clear(); ats.clear(); total_areal = 0.f;
    for (uint32_t i = 0; i < n_areal; i++) {
        uint32_t b = areal_offsets[i], e = areal_offsets[i + 1];
        if (e <= b) continue;
        std::vector<std::pair<float, uint32_t>> twa;
        float total_area = 0.f;
        for (uint32_t j = b; j < e; j++) {
            uint32_t t = areal_tris[j];
            V3 A = vert(tri_indices[3 * t]), B = vert(tri_indices[3 * t + 1]), C = vert(tri_indices[3 * t + 2]);
            V3 c = crossv(sub(A, B), sub(C, B)); // Triangle::GetArea primitives.cpp:38-45
            float area = 0.5f * std::sqrt(dotv(c, c));
            twa.push_back({area, t});
            total_area += area;
        }
        const rgk_material& m0 = materials[tri_material[twa[0].second]];
        std::sort(twa.rbegin(), twa.rend()); // descending by (area, index)
        DevArealLight al{};
        al.total_area = total_area;
        for (int k = 0; k < 3; k++) al.emission[k] = m0.emission[k];
        al.power = total_area * (m0.emission[0] + m0.emission[1] + m0.emission[2]);
        al.first = (uint32_t)ats.size();
        al.count = (uint32_t)twa.size();
        for (auto& p : twa) {
            DevArealTri at{};
            at.area = p.first; at.tri = p.second; at.light = (uint32_t)als.size();
            uint32_t ia = tri_indices[3 * p.second], ib = tri_indices[3 * p.second + 1], ic = tri_indices[3 * p.second + 2];
            for (int k = 0; k < 3; k++) {
                at.a[k] = vertices[3 * ia + k]; at.b[k] = vertices[3 * ib + k]; at.c[k] = vertices[3 * ic + k];
                at.normal_a[k] = normals[3 * ia + k];
            }
            ats.push_back(at);
        }
        total_areal += al.power;
        als.push_back(al);
    }
}

#ifndef RGK_BUILD_AUTO_DEVICE_REFS
#define RGK_BUILD_AUTO_DEVICE_REFS 500000
#endif
#ifndef RGK_LBVH_ROTATE_PASSES
#define RGK_LBVH_ROTATE_PASSES 4
#endif
int rgk_scene_create(const rgk_scene_desc* d, int device, rgk_scene** out) {
    if (!out) return fail(RGK_ERR_INVALID, "null output pointer");
    *out = nullptr;
    int rc = validate_desc(d);
    if (rc) return rc;
    if (std::getenv("RGK_DEBUG_DESC")) debug_desc(d);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RGK_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(RGK_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIPCHK(hipSetDevice(device));
    rgk_scene* s = new rgk_scene;
    s->device = device;
    struct Guard { rgk_scene* s; ~Guard() { delete s; } } guard{s};
    HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&s->ev_prelude, hipEventDisableTiming));
    {
        auto off = [](const char* name) { const char* e = std::getenv(name); return e && e[0] == '0'; };
        RgkTuning& t = s->tune;
        t.entry_points = !off("RGK_ENTRY_POINTS"); t.entry_cap = !off("RGK_ENTRY_CAP"); t.light_entry = !off("RGK_LIGHT_ENTRY");
        if (const char* e = std::getenv("RGK_SAMPLE_GROUP")) t.sample_group = std::min(6, std::max(0, std::atoi(e)));
        if (const char* e = std::getenv("RGK_BATCH_PATHS")) t.batch_paths = std::max<size_t>(1024, strtoull(e, nullptr, 10));
        if (const char* e = std::getenv("RGK_WORKSPACE_GB")) t.workspace_gb = atof(e);
        if (const char* e = std::getenv("RGK_BEAM")) t.beam = std::min(2, std::max(0, std::atoi(e)));
        { const char* e = std::getenv("RGK_TWO_LANES"); t.two_lanes = e && e[0] == '1'; }
        t.debug_bvh = std::getenv("RGK_DEBUG_BVH") != nullptr; t.debug_util = std::getenv("RGK_DEBUG_UTIL") != nullptr;
    }

    const uint32_t nt = d->n_triangles;
    auto vert = [&](uint32_t i) { return V3{d->vertices[3 * i], d->vertices[3 * i + 1], d->vertices[3 * i + 2]}; };

    // ---- Commit: bounds, epsilon (scene.cpp:364-395)
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -mn[a]; }
    for (uint32_t i = 0; i < nt; i++)
        for (int k = 0; k < 3; k++) {
            V3 v = vert(d->tri_indices[3 * i + k]);
            for (int a = 0; a < 3; a++) { float c = comp(v, a); if (c < mn[a]) mn[a] = c; if (c > mx[a]) mx[a] = c; }
        }
    float xs = mx[0] - mn[0], ys = mx[1] - mn[1], zs = mx[2] - mn[2];
    float diameter = std::sqrt(xs * xs + ys * ys + zs * zs);
    float eps = 0.00001f * diameter;
    if (!(eps == eps) || !(diameter < std::numeric_limits<float>::infinity())) return fail(RGK_ERR_INVALID, "non-finite vertex coordinates");
    DevScene& ds = s->dev;
    ds.epsilon = eps;
    for (int a = 0; a < 3; a++) { ds.bb_min[a] = mn[a] - eps; ds.bb_max[a] = mx[a] + eps; }

    // ---- planes + intersection records (primitives.cpp:24-36, 75-166)
    std::vector<TriIsect> recs(nt);
    std::vector<Prim> prims;
    prims.reserve(nt);
    RefSplitter splitter;
    {   // RGK_BVH_SPLIT = longest box side, as a fraction of the scene diagonal, above which a triangle is pre-split
        const char* e = std::getenv("RGK_BVH_SPLIT");
        const float f = e ? (float)std::atof(e) : 0.1f; // swept on the Sponza proxy: 0.08..0.15 best (-7 % node visits); finer splits deepen the tree
        splitter.lmax = f > 0.f ? f * (eps * 1e5f) : 0.f; // eps = 1e-5 * diagonal
        splitter.budget = (size_t)nt; // at most 2x references
        splitter.out = &prims;
    }
    for (uint32_t i = 0; i < nt; i++) {
        V3 v0 = vert(d->tri_indices[3 * i]), v1 = vert(d->tri_indices[3 * i + 1]), v2 = vert(d->tri_indices[3 * i + 2]);
        V3 d0 = sub(v1, v0), d1 = sub(v2, v0);
        V3 n = normv(crossv(d1, d0));
        float dd = -dotv(n, v0);
        TriIsect& r = recs[i];
        r.n[0] = n.x; r.n[1] = n.y; r.n[2] = n.z; r.d = dd;
        int i1, i2;
        float ax = std::fabs(n.x), ay = std::fabs(n.y), az = std::fabs(n.z);
        if (ax > ay && ax > az) { i1 = 1; i2 = 2; }
        else if (ay > az) { i1 = 0; i2 = 2; }
        else { i1 = 0; i2 = 1; }
        r.v0a = comp(v0, i1); r.v0b = comp(v0, i2);
        r.q1x = comp(v1, i1) - comp(v0, i1); r.q1y = comp(v1, i2) - comp(v0, i2);
        r.q2x = comp(v2, i1) - comp(v0, i1); r.q2y = comp(v2, i2) - comp(v0, i2);
        r.axes = (uint32_t)i1 | ((uint32_t)i2 << 2);
        r.tri = i;
        if (n.x == n.x && n.y == n.y && n.z == n.z) { // a NaN plane can never be hit (primitives.cpp:90)
            Prim p;
            for (int a = 0; a < 3; a++) {
                p.bmin[a] = std::min(comp(v0, a), std::min(comp(v1, a), comp(v2, a)));
                p.bmax[a] = std::max(comp(v0, a), std::max(comp(v1, a), comp(v2, a)));
                p.c[a] = 0.5f * (p.bmin[a] + p.bmax[a]);
            }
            p.tri = i;
            p.pb[0] = 0.f; p.pb[1] = 1.f; p.pb[2] = 0.f; p.pb[3] = 1.f;
            if (splitter.lmax > 0.f) {
                std::vector<RefSplitter::P3> poly(3);
                for (int a = 0; a < 3; a++) { poly[0].x[a] = comp(v0, a); poly[1].x[a] = comp(v1, a); poly[2].x[a] = comp(v2, a); }
                for (int c = 0; c < 3; c++) for (int a = 0; a < 3; a++) splitter.tv[c][a] = poly[c].x[a];
                splitter.emit(poly, p.bmin, p.bmax, i, 0);
            } else {
                p.ref = (uint32_t)prims.size();
                prims.push_back(p);
            }
        }
    }
    // ---- accelerator
    if (prims.empty()) return fail(RGK_ERR_INVALID, "every triangle is degenerate");
    if (prims.size() >= (1u << 25)) return fail(RGK_ERR_UNSUPPORTED, "too many triangles (32-bit byte offsets into the triangle tables: < 2^25)");
    std::vector<QNode> qnodes;
    std::vector<TriIsect> leaf_recs;
    std::vector<float4> leaf_pb; // per leaf reference: its piece of the triangle in the triangle's own coordinates (Prim::pb), for rgk_scene_refit
    std::vector<uint32_t> ref_tri(prims.size());
    std::vector<float4> ref_pb(prims.size());
    for (const Prim& p : prims) { ref_tri[p.ref] = p.tri; ref_pb[p.ref] = make_float4(p.pb[0], p.pb[1], p.pb[2], p.pb[3]); }
    uint32_t max_depth = 0, max_stack = 0, n_nodes = 0, n_refs = (uint32_t)prims.size();
    bool on_device = false;
    // leaves of the device build: at most 2 references (Morton-adjacent triangles make loose leaves: with 4, a ray tests twice the
    // triangles the host tree makes it test -- measured with 1 / 2 / 3 / 4 on the 1.05 M-triangle scene, closest-hit + shadow ms of
    // a round: 7.69 / 7.48 / 7.59 / 8.10, host SAH 7.36: tools/gpu_lbvh_rotate_sweep.py)
    int MAX_LEAF_DEV = 2;
    if (const char* e = std::getenv("RGK_BVH_MAXLEAF_DEV")) MAX_LEAF_DEV = std::min(16, std::max(1, std::atoi(e)));
    // which builder: asked for explicitly, or (RGK_BUILD_AUTO) by size -- from half a million references on, the host's SAH build
    // takes seconds (1.5 s at 1.05 M) where the device build takes 0.15 s and traces within 2 % of it
    const bool want_device = (d->build_flags & RGK_BUILD_DEVICE) || (!(d->build_flags & RGK_BUILD_HOST_SAH) && prims.size() >= (size_t)RGK_BUILD_AUTO_DEVICE_REFS);
    if (want_device && prims.size() > (size_t)MAX_LEAF_DEV) {
        // LBVH on the GPU (rgk_build.hip): the references go up, nodes and leaf-ordered records stay on the device
        std::vector<RgkBuildPrim> bp(prims.size());
        for (size_t i = 0; i < prims.size(); i++) {
            for (int a = 0; a < 3; a++) { bp[i].bmin[a] = prims[i].bmin[a]; bp[i].bmax[a] = prims[i].bmax[a]; }
            bp[i].tri = prims[i].tri;
            for (int a = 0; a < 4; a++) bp[i].pb[a] = prims[i].pb[a];
        }
        DevBuf<TriIsect> d_recs;
        if ((rc = d_recs.upload(recs)) || (rc = s->nodes.alloc(prims.size())) || (rc = s->tris.alloc(prims.size())) || (rc = s->leaf_pb.alloc(prims.size()))) { d_recs.release(); return rc; }
        uint32_t levels = 0;
        const char* err = "";
        const char* rot = std::getenv("RGK_LBVH_ROTATE"); // passes of the rotation step; 0: the plain LBVH (for comparisons)
        rc = rgk_build_bvh4_device(s->stream, bp.data(), n_refs, mn, mx, eps, (uint32_t)MAX_LEAF_DEV, rot ? std::max(0, std::min(32, std::atoi(rot))) : RGK_LBVH_ROTATE_PASSES, d_recs.p, s->nodes.p, s->tris.p, s->leaf_pb.p, &n_nodes, &levels, &err);
        d_recs.release();
        if (rc) return fail(rc, "device BVH build: %s", err);
        max_depth = levels;
        max_stack = 3 * levels; // three pushes per level at most
        on_device = true;
    } else {
        std::vector<BvhNode> nodes;
        BvhBuilder bb(prims, eps);
        Box rootbox;
        bb.nodes.reserve(prims.size());
        bb.nodes.emplace_back(); // node 0 = root, filled below if the whole scene is one leaf
        int code;
        if (prims.size() <= (size_t)bb.MAX_LEAF) {
            code = bb.build(0, prims.size(), 1, rootbox);
            BvhNode& r = bb.nodes[0];
            for (int a = 0; a < 3; a++) {
                r.lmin[a] = rootbox.mn[a]; r.lmax[a] = rootbox.mx[a];
                r.rmin[a] = std::numeric_limits<float>::infinity(); r.rmax[a] = -std::numeric_limits<float>::infinity();
            }
            r.left = code; r.right = code; r.pad[0] = r.pad[1] = 0;
        } else {
            bb.nodes.pop_back();
            code = bb.build(0, prims.size(), 0, rootbox);
            if (code != 0) return fail(RGK_ERR_DEVICE, "internal: BVH root is not node 0");
            const char* e = std::getenv("RGK_BVH_OPT"); // reinsertion rounds (0 = off)
            optimise_bvh2(bb.nodes, bb.order, e ? std::atoi(e) : 8, 0.5f);
        }
        nodes.swap(bb.nodes);
        leaf_recs.reserve(bb.order.size());
        leaf_pb.reserve(bb.order.size());
        for (uint32_t r : bb.order) { leaf_recs.push_back(recs[ref_tri[r]]); leaf_pb.push_back(ref_pb[r]); }
        QbvhBuilder qb(nodes);
        qb.out.reserve(nodes.size() / 2 + 1);
        if (qb.collapse(0, 0, 0) != 0) return fail(RGK_ERR_DEVICE, "internal: QBVH root is not node 0");
        qnodes.swap(qb.out);
        max_depth = qb.max_depth; max_stack = qb.max_stack; n_nodes = (uint32_t)qnodes.size(); n_refs = (uint32_t)leaf_recs.size();
    }
    if (max_stack + 1 + RGK_ENTRY_K > 256) return fail(RGK_ERR_UNSUPPORTED, "BVH needs %u traversal-stack entries (max 256)", max_stack + 1);
    {   // traversal stack: 16 entries per lane in LDS + per-lane overflow in global memory (rgk_kernels.hip RGK_TRACE_DISPATCH)
        const int need = (int)max_stack + 1 + RGK_ENTRY_K; // (+ the entry nodes a camera ray starts with)
        const char* e = std::getenv("RGK_STACK_OVF");
        const char* l = std::getenv("RGK_STACK_LDS");
        if (e && e[0] == '0' && need <= 32) { s->tcfg.stack = 32; s->tcfg.lds = 32; }
        else { s->tcfg.stack = 256; s->tcfg.lds = (l && std::atoi(l) == 32) ? 32 : 16; }
        s->tcfg.ovf = nullptr;
        if (s->tcfg.lds < s->tcfg.stack) {
            const size_t per_lane = (size_t)std::max(need - std::min(s->tcfg.lds, 8), 1); // (k_trace_camera_beam keeps 8 entries in LDS)
            s->ovf_lane = (size_t)rgk_trace_grid(s->tcfg.lds) * RGK_TRACE_BLOCK * per_lane;
            if ((rc = s->ovf.alloc(2 * s->ovf_lane))) return rc; // one area per lane of a round: two traversal launches may be in flight
            s->tcfg.ovf = s->ovf.p;
        }
    }

    // ---- shading arrays
    std::vector<TriShade> tsh(nt);
    for (uint32_t i = 0; i < nt; i++) {
        TriShade& t = tsh[i];
        std::memset(&t, 0, sizeof(t));
        const uint32_t v[3] = {d->tri_indices[3 * i], d->tri_indices[3 * i + 1], d->tri_indices[3 * i + 2]};
        float uvs[6];
        for (int k = 0; k < 3; k++) {
            for (int a = 0; a < 3; a++) { t.q[k][a] = d->normals[3 * v[k] + a]; t.q[3 + k][a] = d->tangents[3 * v[k] + a]; }
            uvs[2 * k] = d->texcoords ? d->texcoords[2 * v[k]] : 0.f;
            uvs[2 * k + 1] = d->texcoords ? d->texcoords[2 * v[k] + 1] : 0.f;
        }
        for (int k = 0; k < 6; k++) t.q[k][3] = uvs[k]; // uvA.x uvA.y uvB.x uvB.y uvC.x uvC.y
        t.mat = d->tri_material[i];
    }
    // textures: image texels into one float4 pool (float textures) or one dword pool + byte -> float tables (8-bit ones);
    // a TexRef per (material, slot)
    std::vector<float4> pool;
    std::vector<uint32_t> pool8;
    std::vector<float> luts;
    std::vector<TexRef> trefs(d->n_textures);
    // A float texture whose channel values are at most 256 distinct floats is what a loader leaves that decodes an 8-bit
    // file to floats and keeps only those (the reference: Color(byte / 255).gammaDecode(2.2) per channel,
    // src/texture.cpp:203,252-254, every FileTexture it holds).  Such a texture is stored as bytes + the table of its values --
    // bit-identical by construction (the table holds the very floats), a quarter of the texel traffic, and the table sits in LDS.
    // Textures share a table while the union of their value sets fits 256 entries (one table for all of Sponza's 17 images).
    struct Palette { std::vector<uint32_t> vals; bool fixed; uint32_t lut_off; }; // sorted bit patterns; fixed: a caller-supplied table
    std::vector<Palette> palettes;
    std::vector<int> tex_palette(d->n_textures, -1);
    uint32_t n_float_tex = 0, n_palettized = 0;
    auto bits_of = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    for (uint32_t i = 0; i < d->n_textures; i++) { // caller-supplied tables first: a float texture whose values all occur in one shares it
        const rgk_texture& t = d->textures[i];
        if (t.kind != RGK_TEX_RGB8) continue;
        bool have = false;
        for (const Palette& p : palettes) if (std::memcmp(&luts[p.lut_off], t.lut, 256 * sizeof(float)) == 0) { have = true; break; }
        if (have) continue;
        Palette p; p.fixed = true; p.lut_off = (uint32_t)luts.size();
        luts.insert(luts.end(), t.lut, t.lut + 256);
        for (int k = 0; k < 256; k++) p.vals.push_back(bits_of(t.lut[k]));
        std::sort(p.vals.begin(), p.vals.end());
        p.vals.erase(std::unique(p.vals.begin(), p.vals.end()), p.vals.end());
        palettes.push_back(std::move(p));
    }
    if (!(d->build_flags & RGK_BUILD_KEEP_FLOAT_TEXTURES))
        for (uint32_t i = 0; i < d->n_textures; i++) {
            const rgk_texture& t = d->textures[i];
            if (t.kind != RGK_TEX_RGB32F || t.width > 65535 || t.height > 65535) continue;
            // distinct channel values, giving up at the 257th (open addressing, 1024 slots)
            std::vector<uint32_t> slots(1024, 0u);
            std::vector<uint8_t> used(1024, 0);
            std::vector<uint32_t> vals;
            const size_t n = (size_t)3 * t.width * t.height;
            bool ok = true;
            for (size_t k = 0; k < n && ok; k++) {
                const uint32_t u = bits_of(t.texels[k]);
                uint32_t h = (u * 2654435761u) >> 22;
                while (used[h] && slots[h] != u) h = (h + 1) & 1023u;
                if (!used[h]) { used[h] = 1; slots[h] = u; vals.push_back(u); if (vals.size() > 256) ok = false; }
            }
            if (!ok) continue;
            std::sort(vals.begin(), vals.end());
            int pick = -1;
            for (size_t p = 0; p < palettes.size() && pick < 0; p++) // all of it already in a table?
                if (std::includes(palettes[p].vals.begin(), palettes[p].vals.end(), vals.begin(), vals.end())) pick = (int)p;
            for (size_t p = 0; p < palettes.size() && pick < 0; p++) { // a table of this scene's that can take the new values?
                if (palettes[p].fixed) continue;
                std::vector<uint32_t> u;
                std::set_union(palettes[p].vals.begin(), palettes[p].vals.end(), vals.begin(), vals.end(), std::back_inserter(u));
                if (u.size() <= 256) { palettes[p].vals.swap(u); pick = (int)p; }
            }
            if (pick < 0) { Palette p; p.fixed = false; p.lut_off = 0; p.vals = vals; palettes.push_back(std::move(p)); pick = (int)palettes.size() - 1; }
            tex_palette[i] = pick;
        }
    for (Palette& p : palettes) // this scene's own tables: the sorted values, padded with zeros
        if (!p.fixed) {
            p.lut_off = (uint32_t)luts.size();
            for (size_t k = 0; k < 256; k++) { float f = 0.f; if (k < p.vals.size()) std::memcpy(&f, &p.vals[k], 4); luts.push_back(f); }
        }
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rgk_texture& t = d->textures[i];
        TexRef& o = trefs[i];
        o.kind = t.kind; o.a = o.b = o.c = 0;
        if (t.kind == RGK_TEX_SOLID) {
            std::memcpy(&o.a, &t.color[0], 4); std::memcpy(&o.b, &t.color[1], 4); std::memcpy(&o.c, &t.color[2], 4);
        } else if (t.kind == RGK_TEX_RGB8 || tex_palette[i] >= 0) {
            if (t.width > 65535 || t.height > 65535) return fail(RGK_ERR_UNSUPPORTED, "texture %u larger than 65535 texels on a side", i);
            const size_t n = (size_t)t.width * t.height;
            // byte texels lie in tiles of 8 x 4 (one 128-byte line; rgk_device.h tex_row / tex_col), the image padded up to whole tiles
            const size_t tiles_x = ((size_t)t.width + 7) / 8, tiles_y = ((size_t)t.height + 3) / 4, n_padded = RGK_TEX_TILED ? tiles_x * tiles_y * 32 : n;
            if (pool8.size() + n_padded >= (1ull << 30)) return fail(RGK_ERR_UNSUPPORTED, "8-bit texel pool exceeds 2^30 texels"); // 32-bit byte offsets
            o.kind = RGK_TEX_RGB8;
            o.a = t.width | (t.height << 16);
            while (pool8.size() % 32) pool8.push_back(0u); // a tile = a line: the pool itself is 128-byte aligned
            o.b = (uint32_t)pool8.size();
            const size_t pool_at = pool8.size();
            pool8.resize(pool_at + n_padded, 0u);
            auto put = [&](size_t k, uint32_t w) { // texel k = y * width + x  ->  its place in the tiled order
                const size_t x = k % t.width, y = k / t.width;
                pool8[pool_at + (RGK_TEX_TILED ? ((y >> 2) * tiles_x + (x >> 3)) * 32 + ((y & 3) << 3) + (x & 7) : k)] = w;
            };
            if (t.kind == RGK_TEX_RGB8) {
                for (const Palette& p : palettes) if (p.fixed && std::memcmp(&luts[p.lut_off], t.lut, 256 * sizeof(float)) == 0) { o.c = p.lut_off; break; }
                for (size_t k = 0; k < n; k++)
                    put(k, (uint32_t)t.texels8[3 * k] | ((uint32_t)t.texels8[3 * k + 1] << 8) | ((uint32_t)t.texels8[3 * k + 2] << 16));
            } else { // a float texture with few distinct values: its texels as indices into the table (the first entry holding the value)
                n_float_tex++; n_palettized++;
                const Palette& p = palettes[(size_t)tex_palette[i]];
                o.c = p.lut_off;
                std::vector<std::pair<uint32_t, uint8_t>> idx; // (bit pattern, table index), sorted by pattern
                for (int k = 255; k >= 0; k--) idx.push_back({bits_of(luts[p.lut_off + (size_t)k]), (uint8_t)k});
                std::stable_sort(idx.begin(), idx.end(), [](const std::pair<uint32_t, uint8_t>& a, const std::pair<uint32_t, uint8_t>& b) { return a.first < b.first || (a.first == b.first && a.second < b.second); });
                auto index_of = [&](float f) -> uint32_t {
                    const uint32_t u = bits_of(f);
                    auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(u, (uint8_t)0));
                    return it->second; // present by construction
                };
                for (size_t k = 0; k < n; k++)
                    put(k, index_of(t.texels[3 * k]) | (index_of(t.texels[3 * k + 1]) << 8) | (index_of(t.texels[3 * k + 2]) << 16));
            }
        } else {
            n_float_tex++;
            if (t.width > 65535 || t.height > 65535) return fail(RGK_ERR_UNSUPPORTED, "texture %u larger than 65535 texels on a side", i);
            const size_t n = (size_t)t.width * t.height;
            if (pool.size() + n >= (1ull << 28)) return fail(RGK_ERR_UNSUPPORTED, "float texel pool exceeds 2^28 texels"); // 32-bit byte offsets
            o.a = t.width | (t.height << 16);
            o.b = (uint32_t)pool.size();
            pool.reserve(pool.size() + n);
            for (size_t k = 0; k < n; k++) pool.push_back(make_float4(t.texels[3 * k], t.texels[3 * k + 1], t.texels[3 * k + 2], 0.f));
        }
    }
    auto tref = [&](int32_t id) { TexRef r; r.kind = RGK_TEXREF_NONE; r.a = r.b = r.c = 0; return id < 0 ? r : trefs[id]; };
    std::vector<DevMaterial> mats(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rgk_material& m = d->materials[i];
        DevMaterial& o = mats[i];
        std::memset(&o, 0, sizeof(o));
        o.kind = m.kind; o.flags = m.flags;
        for (int k = 0; k < 3; k++) o.emission[k] = m.emission[k];
        o.roughness = m.roughness; o.ior = m.ior; o.amount = m.amount;
        o.t_diffuse = tref(m.tex_diffuse); o.t_color = tref(m.tex_color); o.t_bump = tref(m.tex_bump);
        o.mix_m1 = m.mix_m1; o.mix_m2 = m.mix_m2;
    }
    // ---- lights (scene.cpp:323-344)
    std::vector<DevPointLight> pls(d->n_pointlights);
    float total_point = 0.f;
    const float PI_F = 3.14159265358979323846264338327950288f;
    for (uint32_t i = 0; i < d->n_pointlights; i++) {
        const rgk_pointlight& l = d->pointlights[i];
        DevPointLight& o = pls[i];
        for (int k = 0; k < 3; k++) { o.pos[k] = l.pos[k]; o.color[k] = l.color[k]; }
        o.intensity = l.intensity; o.size = l.size;
        total_point += l.intensity * 4.0f * PI_F;
    }
    std::vector<DevArealLight> als;
    std::vector<DevArealTri> ats;
    float total_areal = 0.f;
    build_areal_tables(d->vertices, d->normals, d->tri_indices, d->tri_material, d->materials, d->n_areal_lights, d->areal_offsets, d->areal_tris, als, ats, total_areal);
    std::vector<DevHaltonDim> hd;
    std::vector<uint16_t> hp;
    build_halton(hd, hp);

    // ---- upload
    s->n_textures = d->n_textures; s->n_materials = d->n_materials;
    s->n_vertices = d->n_vertices; s->n_triangles = nt; s->n_refs = n_refs; s->n_nodes = n_nodes;
    s->h_idx.assign(d->tri_indices, d->tri_indices + 3 * (size_t)nt);
    s->h_tri_mat.assign(d->tri_material, d->tri_material + nt);
    s->h_mats.assign(d->materials, d->materials + d->n_materials);
    s->h_normals.assign(d->normals, d->normals + 3 * (size_t)d->n_vertices);
    if (d->n_areal_lights) {
        s->h_areal_off.assign(d->areal_offsets, d->areal_offsets + d->n_areal_lights + 1);
        s->h_areal_tris.assign(d->areal_tris, d->areal_tris + d->areal_offsets[d->n_areal_lights]);
    }
    if ((rc = s->d_idx.upload(s->h_idx))) return rc;
    if ((rc = s->texrefs.upload(trefs))) return rc;
    if (!on_device && ((rc = s->nodes.upload(qnodes)) || (rc = s->tris.upload(leaf_recs)) || (rc = s->leaf_pb.upload(leaf_pb)))) return rc;
    if ((rc = s->tri_shade.upload(tsh)) ||
        (rc = s->materials.upload(mats)) || (rc = s->texels.upload(pool)) || (rc = s->texels8.upload(pool8)) ||
        (rc = s->luts.upload(luts)) || (rc = s->pointlights.upload(pls)) || (rc = s->areal.upload(als)) ||
        (rc = s->areal_tris.upload(ats)) || (rc = s->hdims.upload(hd)) || (rc = s->hperm.upload(hp)))
        return rc;
    { // both LTC tables in one buffer, {m0,m2,m4,m6}{amp,0,0,0} per entry (two 16-byte loads): GGX, then Beckmann
        std::vector<float4> t(2 * 2 * 4096, make_float4(0.f, 0.f, 0.f, 0.f));
        const float* src[2] = {d->ltc_ggx, d->ltc_beckmann};
        for (int w = 0; w < 2; w++)
            for (int k = 0; src[w] && k < 4096; k++) {
                t[(size_t)w * 8192 + 2 * k] = make_float4(src[w][5 * k], src[w][5 * k + 1], src[w][5 * k + 2], src[w][5 * k + 3]);
                t[(size_t)w * 8192 + 2 * k + 1] = make_float4(src[w][5 * k + 4], 0.f, 0.f, 0.f);
            }
        if ((rc = s->ltc.upload(t))) return rc;
    }
    ds.nodes = s->nodes.p;
    { const char* e = std::getenv("RGK_WALK_Q"); ds.walk_q = e ? (uint32_t)std::atoi(e) : 3u; }
    if (s->tune.debug_bvh) std::fprintf(stderr, "[rgk] bvh4 (%s) nodes %u max_stack %u max_depth %u refs %u of %u triangles\n", on_device ? "device LBVH" : "host SAH", n_nodes, max_stack, max_depth, n_refs, nt);
    ds.tris = s->tris.p; ds.tri_shade = s->tri_shade.p;
    ds.materials = s->materials.p; ds.texels = s->texels.p; ds.texels8 = s->texels8.p; ds.luts = s->luts.p; ds.n_lut_floats = (uint32_t)luts.size(); ds.n_materials = (uint32_t)mats.size();
    ds.pointlights = s->pointlights.p; ds.areal = s->areal.p; ds.areal_tris = s->areal_tris.p;
    ds.ltc = s->ltc.p; ds.hdims = s->hdims.p; ds.hperm = s->hperm.p;
    ds.n_pointlights = (uint32_t)pls.size(); ds.n_areal = (uint32_t)als.size();
    ds.total_point_power = total_point; ds.total_areal_power = total_areal;
    ds.has_texcoords = d->texcoords ? 1u : 0u;
    ds.sky_mode = d->sky_mode;
    for (int k = 0; k < 3; k++) ds.sky_color[k] = d->sky_color[k];
    ds.sky_intensity = d->sky_intensity; ds.sky_rotate = d->sky_rotate; ds.sky_tex = tref(d->sky_mode == RGK_SKY_ENVMAP ? d->sky_texture : -1);
    if ((rc = s->self.alloc(1))) return rc;
    ds.self = s->self.p;
    if (hipMemcpy(s->self.p, &ds, sizeof(DevScene), hipMemcpyHostToDevice) != hipSuccess) return fail(RGK_ERR_DEVICE, "hipMemcpy(DevScene)");

    rgk_scene_info& inf = s->info;
    inf.epsilon = eps;
    for (int a = 0; a < 3; a++) { inf.bbox_min[a] = ds.bb_min[a]; inf.bbox_max[a] = ds.bb_max[a]; }
    inf.total_areal_power = total_areal; inf.total_point_power = total_point;
    inf.n_nodes = n_nodes; inf.node_bytes = RGK_NODE_BYTES; inf.tri_bytes = RGK_TRI_BYTES;
    inf.max_depth = max_depth; inf.n_leaf_refs = n_refs;
    inf.n_float_textures = n_float_tex; inf.n_palettized_textures = n_palettized;
    guard.s = nullptr;
    *out = s;
    return RGK_OK;
}

int rgk_scene_refit(rgk_scene* s, const float* vertices, const float* normals, const float* tangents) {
    if (!s || !vertices) return fail(RGK_ERR_INVALID, "null argument");
    if (s->prog_busy.load()) return fail(RGK_ERR_INVALID, "rgk_scene_refit while a round is in flight on this scene");
    HIPCHK(hipSetDevice(s->device));
    const uint32_t nt = s->n_triangles, nv = s->n_vertices;
    // ---- Commit's scalars for the new positions: bounds, epsilon (scene.cpp:364-395)
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -mn[a]; }
    for (size_t k = 0; k < 3 * (size_t)nt; k++) {
        const float* v = vertices + 3 * (size_t)s->h_idx[k];
        for (int a = 0; a < 3; a++) { if (v[a] < mn[a]) mn[a] = v[a]; if (v[a] > mx[a]) mx[a] = v[a]; }
    }
    const float xs = mx[0] - mn[0], ys = mx[1] - mn[1], zs = mx[2] - mn[2];
    const float diameter = std::sqrt(xs * xs + ys * ys + zs * zs);
    const float eps = 0.00001f * diameter;
    if (!(eps == eps) || !(diameter < std::numeric_limits<float>::infinity())) return fail(RGK_ERR_INVALID, "non-finite vertex coordinates");
    // ---- records, shading normals / tangents, and the tree's boxes: on the device
    int rc;
    if ((rc = up(s->scratch_f, vertices, 3 * (size_t)nv))) return rc;
    DevBuf<float> d_n, d_t;
    struct Rel { DevBuf<float>&a, &b; ~Rel() { a.release(); b.release(); } } rel{d_n, d_t};
    if (normals && (rc = up(d_n, normals, 3 * (size_t)nv))) return rc;
    if (tangents && (rc = up(d_t, tangents, 3 * (size_t)nv))) return rc;
    const char* err = "";
    rc = rgk_refit_bvh4_device(s->stream, s->n_refs, s->n_nodes, nt, s->scratch_f.p, normals ? d_n.p : nullptr, tangents ? d_t.p : nullptr, s->d_idx.p, s->leaf_pb.p, eps,
                               s->tris.p, s->nodes.p, s->tri_shade.p, &err);
    if (rc) return fail(rc, "refit: %s", err);
    // ---- areal-light tables (areas, positions, vertex-A normals change with the vertices)
    if (normals) s->h_normals.assign(normals, normals + 3 * (size_t)nv);
    std::vector<DevArealLight> als;
    std::vector<DevArealTri> ats;
    float total_areal = 0.f;
    build_areal_tables(vertices, s->h_normals.data(), s->h_idx.data(), s->h_tri_mat.data(), s->h_mats.data(), (uint32_t)(s->h_areal_off.empty() ? 0 : s->h_areal_off.size() - 1),
                       s->h_areal_off.data(), s->h_areal_tris.data(), als, ats, total_areal);
    if ((rc = s->areal.upload(als)) || (rc = s->areal_tris.upload(ats))) return rc;
    DevScene& ds = s->dev;
    ds.areal = s->areal.p; ds.areal_tris = s->areal_tris.p; ds.n_areal = (uint32_t)als.size(); ds.total_areal_power = total_areal;
    ds.epsilon = eps;
    for (int a = 0; a < 3; a++) { ds.bb_min[a] = mn[a] - eps; ds.bb_max[a] = mx[a] + eps; }
    if (hipMemcpy(s->self.p, &ds, sizeof(DevScene), hipMemcpyHostToDevice) != hipSuccess) return fail(RGK_ERR_DEVICE, "hipMemcpy(DevScene)");
    s->info.epsilon = eps; s->info.total_areal_power = total_areal;
    for (int a = 0; a < 3; a++) { s->info.bbox_min[a] = ds.bb_min[a]; s->info.bbox_max[a] = ds.bb_max[a]; }
    s->entry_key = 0; s->entry_n = 0; s->entry_capped = 0; s->lentry_done = 0; // per-frame lists were made for the old boxes
    return RGK_OK;
}

void rgk_scene_destroy(rgk_scene* s) { delete s; }

int rgk_scene_get_info(const rgk_scene* s, rgk_scene_info* out) {
    if (!s || !out) return fail(RGK_ERR_INVALID, "null argument");
    *out = s->info;
    return RGK_OK;
}

int rgk_scene_get_progress(const rgk_scene* s, rgk_progress* out) {
    if (!s || !out) return fail(RGK_ERR_INVALID, "null argument");
    out->stage = s->h_stage ? ((volatile const uint32_t*)s->h_stage)[0] + ((volatile const uint32_t*)s->h_stage)[1] : 0u; out->stages = s->prog_stages.load(); out->rounds = s->prog_rounds.load(); out->busy = s->prog_busy.load();
    out->round_pixels = s->prog_pixels.load(); out->round_paths = s->prog_paths.load();
    if (out->stage > out->stages) out->stage = out->stages;
    return RGK_OK;
}

int rgk_scene_set_tuning(rgk_scene* s, const char* key, double value) {
    if (!s || !key) return fail(RGK_ERR_INVALID, "null argument");
    if (s->prog_busy.load()) return fail(RGK_ERR_INVALID, "rgk_scene_set_tuning while a round is in flight on this scene");
    RgkTuning& t = s->tune;
    const std::string k(key);
    if (k == "entry_points") t.entry_points = value != 0;
    else if (k == "entry_cap") t.entry_cap = value != 0;
    else if (k == "light_entry") t.light_entry = value != 0;
    else if (k == "sample_group") t.sample_group = value < 0 ? -1 : (int)std::min(6.0, value);
    else if (k == "batch_paths") t.batch_paths = value <= 0 ? 0 : std::max<size_t>(1024, (size_t)value);
    else if (k == "workspace_gb") t.workspace_gb = value <= 0 ? 0.0 : value;
    else if (k == "two_lanes") t.two_lanes = value != 0;
    else if (k == "beam") t.beam = (int)std::min(2.0, std::max(0.0, value));
    else return fail(RGK_ERR_INVALID, "unknown tuning key '%s'", key);
    // per-frame lists were made under the old switches: the next round rebuilds them
    s->entry_key = 0; s->entry_n = 0; s->entry_capped = 0; s->lentry_done = 0;
    return RGK_OK;
}

int rgk_generate_task_list(uint32_t tile_size, uint32_t xres, uint32_t yres, float mid_x, float mid_y, uint32_t seedstart,
                           uint32_t seedcount_base, rgk_tile* tiles, uint32_t* n_tiles) {
    if (!n_tiles || tile_size == 0) return fail(RGK_ERR_INVALID, "bad argument");
    struct T { rgk_tile t; float d; };
    std::vector<T> v;
    for (uint32_t yp = 0; yp < yres; yp += tile_size)
        for (uint32_t xp = 0; xp < xres; xp += tile_size) {
            T t;
            t.t.x0 = xp; t.t.x1 = std::min(xres, xp + tile_size);
            t.t.y0 = yp; t.t.y1 = std::min(yres, yp + tile_size);
            t.t.seed = 0;
            float mx = (t.t.x0 + t.t.x1) / 2.0f, my = (t.t.y0 + t.t.y1) / 2.0f; // RenderTask::midpoint tracer.hpp:18
            float dx = mid_x - mx, dy = mid_y - my;
            t.d = std::sqrt(dx * dx + dy * dy);
            v.push_back(t);
        }
    std::stable_sort(v.begin(), v.end(), [](const T& a, const T& b) { return a.d < b.d; });
    if (tiles) {
        if (*n_tiles < v.size()) return fail(RGK_ERR_INVALID, "tile buffer too small (%u < %zu)", *n_tiles, v.size());
        for (size_t i = 0; i < v.size(); i++) { tiles[i] = v[i].t; tiles[i].seed = seedstart + seedcount_base + (uint32_t)i; }
    }
    *n_tiles = (uint32_t)v.size();
    return RGK_OK;
}

int rgk_camera_init(rgk_camera* o, const float pos[3], const float lookat[3], const float upv[3], float yview, float xview, int32_t xsize,
                    int32_t ysize, float focus_plane, float lens_size) {
    // Camera::Camera, reference src/camera.cpp:7-24
    if (!o || !pos || !lookat || !upv) return fail(RGK_ERR_INVALID, "null argument");
    V3 origin{pos[0], pos[1], pos[2]}, la{lookat[0], lookat[1], lookat[2]}, up{upv[0], upv[1], upv[2]};
    V3 direction = normv(sub(la, origin));
    V3 left = normv(crossv(up, direction));
    up = normv(crossv(left, direction));
    V3 vx = scale(scale(left, -xview), focus_plane);
    V3 vy = scale(scale(up, yview), focus_plane);
    V3 a = {origin.x + direction.x * focus_plane, origin.y + direction.y * focus_plane, origin.z + direction.z * focus_plane};
    V3 hy = scale(vy, 0.5f), hx = scale(vx, 0.5f);
    V3 vs = sub(sub(a, hy), hx);
    auto put = [](float* dst, V3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; };
    put(o->origin, origin); put(o->direction, direction); put(o->cameraup, up); put(o->cameraleft, left);
    put(o->viewscreen, vs); put(o->viewscreen_x, vx); put(o->viewscreen_y, vy);
    o->lens_size = lens_size; o->xsize = xsize; o->ysize = ysize;
    return RGK_OK;
}

static void make_camera(const rgk_camera* c, DevCamera& o) { // the members RenderRound's `const Camera&` carries, as they are
    for (int k = 0; k < 3; k++) {
        o.origin[k] = c->origin[k]; o.direction[k] = c->direction[k]; o.up[k] = c->cameraup[k]; o.left[k] = c->cameraleft[k];
        o.viewscreen[k] = c->viewscreen[k]; o.viewscreen_x[k] = c->viewscreen_x[k]; o.viewscreen_y[k] = c->viewscreen_y[k];
    }
    o.lens_size = c->lens_size; o.xsize = c->xsize; o.ysize = c->ysize;
}

// Paths resident per pass.  The path state is sized for the machine, not for a cache: by default
// 96 GB of the 288 GB HBM3E (measured on Sponza 1080p x 256 spp: 2^25 paths/pass 2235 Mpaths/s, 2^27 2398,
// 2^28 2440 -- fewer, longer launches and shorter tails; 48 -> 96 GB: +1.4 % there, +6 % on the bidirectional
// configuration whose paths carry 3.5x the state).  RGK_WORKSPACE_GB / RGK_BATCH_PATHS override.
static size_t batch_paths(const RgkTuning& tune, uint32_t reverse) {
    if (tune.batch_paths) return tune.batch_paths;
    // rays 2 x 32, hit 16, state 16, sum 16, light 16, shadow queue 48, generic list 4; bidirectional: + light start 16, light
    // vertices, hit list 4, and the vertex queue's 2 + (1 + reverse) more float4 than a ray's 3
    const size_t per_path = 180 + (reverse ? 16 + 16 * RGK_LV_FLOAT4 * (size_t)reverse + 12 + 16 * (6 + 4 + (size_t)reverse + 1) : 0);
    const bool g = tune.workspace_gb > 0;
    double gb = g ? tune.workspace_gb : (reverse ? 160.0 : 96.0); // bidirectional paths carry 3.5x the state: 765 -> 781 Mpaths/s
    if (!g) { // a shared or smaller card: never plan for more than 60 % of what is free right now
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) gb = std::min(gb, 0.6 * (double)free_b / 1e9);
    }
    size_t b = (size_t)(gb * 1e9 / (double)per_path);
    return std::min<size_t>(std::max<size_t>(b, 1024), (size_t)1 << 30);
}

int rgk_render_round_device(rgk_scene* s, const rgk_camera* camera, const rgk_params* prm, const rgk_tile* tiles, uint32_t n_tiles,
                            float* d_accum_rgb, uint32_t* d_accum_count, rgk_counters* counters) {
    if (!s || !camera || !prm || (!tiles && n_tiles) || !d_accum_rgb || !d_accum_count) return fail(RGK_ERR_INVALID, "null argument");
    if (prm->xres == 0 || prm->yres == 0 || prm->xres > 65535 || prm->yres > 65535) return fail(RGK_ERR_INVALID, "resolution out of range");
    if (prm->multisample == 0) return fail(RGK_ERR_INVALID, "multisample must be >= 1");
    if (prm->depth > RGK_MAX_DEPTH) return fail(RGK_ERR_UNSUPPORTED, "recursion depth %u > %d", prm->depth, RGK_MAX_DEPTH);
    if (prm->reverse > 7) return fail(RGK_ERR_UNSUPPORTED, "reverse %u > 7 light sub-path vertices", prm->reverse);
    if (prm->sampler != RGK_SAMPLER_HALTON) return fail(RGK_ERR_UNSUPPORTED, "the HIP path implements the Halton sampler only");
    HIPCHK(hipSetDevice(s->device));
    if (counters) std::memset(counters, 0, sizeof(*counters));
    // ---- pixel list in Tracer::Render order, per-pixel seeds (a1, a2): built on the device from the tile list
    std::vector<uint32_t> toff(n_tiles + 1, 0u);
    for (uint32_t i = 0; i < n_tiles; i++) {
        const rgk_tile& t = tiles[i];
        if (t.x1 > prm->xres || t.y1 > prm->yres || t.x0 > t.x1 || t.y0 > t.y1) return fail(RGK_ERR_INVALID, "tile %u outside the frame", i);
        const uint64_t n = (uint64_t)toff[i] + (uint64_t)(t.x1 - t.x0) * (t.y1 - t.y0);
        if (n >= (1ull << 31)) return fail(RGK_ERR_UNSUPPORTED, "more than 2^31 pixels in one round");
        toff[i + 1] = (uint32_t)n;
    }
    const size_t P = toff[n_tiles];
    if (P == 0) return RGK_OK;
    int rc;
    {
        hipStream_t st0 = s->stream;
        if ((rc = s->pix_xy.alloc(P)) || (rc = s->pix_seed.alloc(P)) || (rc = s->tile_buf.alloc((size_t)n_tiles * 5 + n_tiles + 1))) return rc;
        // (both sources outlive the copies: `toff` lives to the end of this call, which synchronises the stream before returning)
        static_assert(sizeof(rgk_tile) == 5 * sizeof(uint32_t), "rgk_tile layout");
        HIPCHK(hipMemcpyAsync(s->tile_buf.p, tiles, (size_t)n_tiles * sizeof(rgk_tile), hipMemcpyHostToDevice, st0));
        HIPCHK(hipMemcpyAsync(s->tile_buf.p + (size_t)n_tiles * 5, toff.data(), (n_tiles + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st0));
        rgk_launch_build_pixel_list(st0, reinterpret_cast<const rgk_tile*>(s->tile_buf.p), s->tile_buf.p + (size_t)n_tiles * 5, n_tiles, s->pix_xy.p, s->pix_seed.p);
        if (s->tune.entry_points) { // (off: every camera ray starts at the root)
            // the entry nodes depend on the camera and on which pixels the list holds in which order -- not on the seeds: a frame's
            // rounds share them (0.7 ms per round at 1080p otherwise)
            uint64_t key = 1469598103934665603ull;
            auto mix = [&key](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { key ^= b[i]; key *= 1099511628211ull; } };
            mix(camera, sizeof(*camera)); mix(&prm->xres, sizeof(prm->xres)); mix(&prm->yres, sizeof(prm->yres));
            for (uint32_t t = 0; t < n_tiles; t++) mix(&tiles[t], 4 * sizeof(uint32_t)); // x0, x1, y0, y1 (the seed is the fifth word)
            const size_t n_entry = ((size_t)P + RGK_ENTRY_PIX - 1) / RGK_ENTRY_PIX * RGK_ENTRY_K;
            if (!(s->entry.p && s->entry_n == n_entry && s->entry_key == key)) {
                if ((rc = s->entry.alloc(n_entry)) || (rc = s->entry_cap.alloc(n_entry / RGK_ENTRY_K + 1)) || (rc = s->trange.alloc((n_entry / RGK_ENTRY_K + 1) * 2))) return rc;
                DevCamera cam0;
                make_camera(camera, cam0);
                rgk_launch_entry_points(st0, s->dev, cam0, prm->xres, prm->yres, s->pix_xy.p, (uint32_t)P, 0u, (uint32_t)(n_entry / RGK_ENTRY_K), nullptr, s->entry.p, s->entry_cap.p);
                s->entry_key = key; s->entry_n = n_entry;
                s->entry_capped = 0; s->lentry_done = 0; // a new frame: capped / light-side lists are rebuilt as its first passes finish
            }
        } else { s->entry.release(); s->entry_cap.release(); }
    }
    // no light at all: TracePath builds no light sub-path (`reverse > 0 && valid light`), same as reverse == 0
    const uint32_t R = (s->dev.total_point_power + s->dev.total_areal_power > 0.0f) ? prm->reverse : 0u;
    // paths per pass: what the card has room for now (an existing workspace counts as room); halved on an allocation failure
    size_t B = batch_paths(s->tune, R);
    if (!s->tune.batch_paths && s->batch_reverse >= R) B = std::max(B, s->batch); // (an explicit batch size is taken literally)
    // Two lanes (an experiment, off by default: RGK_TWO_LANES=1 / rgk_scene_set_tuning "two_lanes"): a unidirectional round of
    // shallow depth runs the two halves of its pixel list as two passes side by side on two streams, each in its own half of the
    // workspace -- the idea being that every launch ends in a tail of waves still finishing which the other lane's launches could
    // fill.  Measured inside one process, same bits: Sponza 1080p x 256 126.7 vs 126.9 ms per round (nothing), Cornell 1024 x 256
    // 119.3 vs 110.4 (8 % WORSE: ten bounces of short launches, each now competing for the card) -- the launches fill the machine
    // one at a time, as round 2 found with two processes and with two host threads.
    const bool count_stats = (prm->flags & RGK_FLAG_COUNT_TRAVERSAL) != 0;
    const bool track = prm->depth > 12; // measured: depth 10 loses 4 % to the read-backs, depth 40 gains 4 %
    const bool two = s->tune.two_lanes && R == 0 && !track && !count_stats && (uint64_t)P * prm->multisample >= (1ull << 22) && P >= 4096;
    size_t npix_pass;
    uint32_t ns_pass;
    size_t lane_cap = 0; // paths per lane = offset of the second lane in every workspace array
    for (;;) {
        const size_t Bl = two ? B / 2 : B; // per lane
        npix_pass = std::min(P, Bl);
        if (two) npix_pass = std::min(npix_pass, ((P + 1) / 2 + 1023) & ~(size_t)1023); // at least two passes: halves of the list, whole 1024-pixel tiles
        const uint32_t ns_max = (uint32_t)std::max<size_t>(1, std::min<size_t>(prm->multisample, Bl / npix_pass));
        const uint32_t n_sample_passes = (prm->multisample + ns_max - 1) / ns_max;
        ns_pass = (prm->multisample + n_sample_passes - 1) / n_sample_passes; // equal-sized passes
        lane_cap = npix_pass * ns_pass;
        rc = ensure_workspace(s, two ? 2 * lane_cap : lane_cap, R);
        if (rc != RGK_ERR_OOM || B <= ((size_t)1 << 20)) break;
        (void)hipGetLastError();
        B /= 2;
    }
    if (rc) return rc;
    if ((rc = s->pixsum.alloc(P))) return rc;

    DevCamera cam;
    make_camera(camera, cam);
    hipStream_t st = s->stream;
    const bool timing = (prm->flags & RGK_FLAG_TIME_KERNELS) != 0;
    HIPCHK(hipMemsetAsync(s->stats.p, 0, 8 * sizeof(unsigned long long), st));
    struct Ev { int cls; hipEvent_t a, b; };
    std::vector<Ev> evs;
    size_t ev_used = 0;
    auto ev_get = [&](hipEvent_t& e) -> int {
        if (ev_used == s->events.size()) { hipEvent_t n; HIPCHK(hipEventCreate(&n)); s->events.push_back(n); }
        e = s->events[ev_used++];
        return 0;
    };
    // counting mode: the traversal counters (stats[0..3]: closest nodes / triangles, shadow nodes / triangles) are read back after
    // every traversal launch, so that each kernel gets its own share (stream-ordered copies; nobody times a counting round)
    std::vector<std::pair<int, std::array<unsigned long long, 4>>> snaps;
    if (count_stats) snaps.reserve(4096);
#define TIMED(kid_, call)                                                \
    do {                                                                 \
        if (timing) {                                                    \
            Ev ev; ev.cls = (kid_);                                      \
            if ((rc = ev_get(ev.a)) || (rc = ev_get(ev.b))) return rc;   \
            HIPCHK(hipEventRecord(ev.a, st));                            \
            call;                                                        \
            HIPCHK(hipEventRecord(ev.b, st));                            \
            evs.push_back(ev);                                           \
        } else { call; }                                                 \
        if (count_stats && is_trace_kernel(kid_)) {                      \
            snaps.emplace_back((int)(kid_), std::array<unsigned long long, 4>{}); \
            HIPCHK(hipMemcpyAsync(snaps.back().second.data(), s->stats.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st)); \
            HIPCHK(hipStreamSynchronize(st));                            \
        }                                                                \
    } while (0)
    auto is_trace_kernel = [](int k) { return k == RGK_K_TRACE_CAMERA || k == RGK_K_TRACE_CLOSEST || k == RGK_K_SHADOW_FIRST || k == RGK_K_SHADOW ||
                                              k == RGK_K_SHADOW_JOBS || k == RGK_K_LIGHT_TRACE || k == RGK_K_LIGHT_SPLAT; };
    uint64_t units[RGK_K_COUNT] = {};

    uint64_t path_rays = 0, shadow_rays = 0;
    // progress: one stage per bounce per pass; a one-thread kernel queued behind each bounce writes the stage number into pinned
    // host memory when the DEVICE gets there (a host function in the stream did the same but stalls the stream for a host
    // round trip per mark)
    {
        const uint32_t n_pix_passes = (uint32_t)((P + npix_pass - 1) / npix_pass), n_s_passes = (prm->multisample + ns_pass - 1) / ns_pass;
        ((volatile uint32_t*)s->h_stage)[0] = 0; ((volatile uint32_t*)s->h_stage)[1] = 0; s->prog_stages = n_pix_passes * n_s_passes * std::max(1u, prm->depth);
        s->prog_pixels = P; s->prog_paths = (uint64_t)P * prm->multisample; s->prog_busy = 1;
    }
    struct Done { rgk_scene* s; ~Done() { ((volatile uint32_t*)s->h_stage)[0] = s->prog_stages.load(); ((volatile uint32_t*)s->h_stage)[1] = 0; s->prog_busy = 0; } } done_guard{s};
    uint32_t stage_targets[2] = {0, 0}; // per lane: what its stage word must read once everything queued on it so far has run
    int lane = 0;
    auto stage_mark = [&](uint32_t upto) -> int { // queued: "this lane's stages up to `upto` are done" (monotonic: bounces that never ran count too)
        rgk_launch_stage_mark(st, s->h_stage + lane, upto);
        return 0;
    };
    PassParams pp{};
    pp.multisample = prm->multisample; pp.depth = prm->depth; pp.xres = prm->xres; pp.yres = prm->yres;
    pp.clamp = prm->clamp; pp.russian = prm->russian; pp.bumpmap_scale = prm->bumpmap_scale; pp.reverse = R;
    pp.lstart = s->lstart.p; pp.lv = s->lv.p; pp.hitlist = s->hitlist.p; pp.lvmask = s->lvmask.p; pp.conn = s->conn.p; pp.connlist = s->connlist.p;
    pp.batch = (uint32_t)s->batch;
    pp.pix_xy = s->pix_xy.p; pp.pix_seed = s->pix_seed.p;
    pp.entry = s->entry.p; // (null when switched off; only the unidirectional bounce-0 launch reads it)
    pp.entry_cap = s->entry_cap.p;
    const bool cap_entries = s->entry.p != nullptr && s->tune.entry_cap;
    pp.lentry = nullptr;
    // one point / sphere light and nothing else that emits: every first-vertex shadow ray starts there (k_entry_points_light)
    const bool light_entry = s->entry.p && s->tune.light_entry && s->dev.n_pointlights == 1 && s->dev.n_areal == 0;
    if (light_entry) {
        const size_t groups = ((size_t)P + RGK_ENTRY_PIX - 1) / RGK_ENTRY_PIX + 1;
        if ((rc = s->lentry.alloc(groups * RGK_ENTRY_K)) || (rc = s->trange.alloc(groups * 2)) || (rc = s->lbox.alloc(groups * 2))) return rc;
    }
    if ((rc = s->htab.alloc((size_t)192 * prm->multisample))) return rc;
    TIMED(RGK_K_OTHER, rgk_launch_build_halton_table(st, s->dev, prm->multisample, s->htab.p));
    pp.htab = s->htab.p;
    // Deep path loops (depth > 12): the length of the next queue is read back every other bounce from the fourth on; it
    // bounds the grids of the following launches (queues only shrink) and ends the loop once no path is left.
    auto queue_len = [&](const uint32_t* dptr, uint32_t& out) -> int {
        // the copy lands in pinned memory; polling it costs microseconds where hipStreamSynchronize was measured at
        // 2-3 ms per call (blocking wait), more than the launches it saves
        volatile uint32_t* h = s->h_counters;
        h[0] = 0xffffffffu; // never a queue length (queues hold < 2^30 entries)
        HIPCHK(hipMemcpyAsync(s->h_counters, dptr, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        for (uint64_t spins = 0; h[0] == 0xffffffffu; spins++) {
            if ((spins & 0xfffff) != 0xfffff) continue;
            // every ~1 M polls ask the stream: not-ready means keep polling, success means the copy has landed (re-read), anything
            // else is a sticky launch / device error that would otherwise spin here for ever
            const hipError_t q = hipStreamQuery(st);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) return fail(RGK_ERR_DEVICE, "queue-length read-back: %s", hipGetErrorString(q));
            if (h[0] == 0xffffffffu) HIPCHK(hipStreamSynchronize(st));
            if (h[0] == 0xffffffffu) return fail(RGK_ERR_DEVICE, "queue-length read-back never landed");
            break;
        }
        out = h[0];
        return 0;
    };
    // a lane's finished pass: wait for its stream, add its queue counters to the round's totals
    bool pending[2] = {false, false};
    uint32_t pend_n0[2] = {0, 0};
    uint32_t pass_index = 0;
    const bool light_entry_units = light_entry;
    auto harvest = [&](int l) -> int {
        HIPCHK(hipStreamSynchronize(l ? s->stream2 : s->stream));
        pending[l] = false;
        const uint32_t* hc = s->h_counters + (size_t)l * 2 * RGK_CNT_TOTAL;
        const uint32_t n0 = pend_n0[l];
        for (uint32_t b = 0; b < prm->depth; b++) { path_rays += hc[RGK_CNT_QUEUE + b]; shadow_rays += hc[RGK_CNT_SHADOW + b] + hc[RGK_CNT_SRAYS + b]; }
        // what each kernel processed in this pass: rays / vertices, from the queue counters
        units[RGK_K_TRACE_CAMERA] += hc[RGK_CNT_QUEUE]; units[RGK_K_SHADE_FIRST] += hc[RGK_CNT_QUEUE];
        for (uint32_t b = 1; b < prm->depth; b++) { units[RGK_K_TRACE_CLOSEST] += hc[RGK_CNT_QUEUE + b]; units[RGK_K_SHADE] += hc[RGK_CNT_QUEUE + b]; }
        for (uint32_t b = 0; b < prm->depth; b++) {
            units[(b == 0 && light_entry_units) ? RGK_K_SHADOW_FIRST : RGK_K_SHADOW] += hc[RGK_CNT_SHADOW + b];
            units[RGK_K_SHADOW_JOBS] += hc[RGK_CNT_CONN + b]; units[RGK_K_CONNECT] += hc[RGK_CNT_CONN + b];
        }
        for (uint32_t k = 0; k < R; k++) {
            units[RGK_K_LIGHT_TRACE] += hc[RGK_CNT_TOTAL + RGK_CNT_QUEUE + k]; units[RGK_K_LIGHT_SHADE] += hc[RGK_CNT_TOTAL + RGK_CNT_HITS + k];
            units[RGK_K_LIGHT_SPLAT] += hc[RGK_CNT_TOTAL + RGK_CNT_SHADOW + k];
        }
        units[RGK_K_RESOLVE] += n0;
        // (light rays: the reference traces and counts one per path, path_tracer.cpp:126,349 -- the ones culled before the queue included)
        for (uint32_t k = 0; k < R; k++) { path_rays += k == 0 ? n0 : hc[RGK_CNT_TOTAL + RGK_CNT_QUEUE + k]; shadow_rays += hc[RGK_CNT_TOTAL + RGK_CNT_SHADOW + k]; }
        return 0;
    };
    // everything queued so far (pixel and seed lists, entry nodes, the Halton table) is on the first stream: the second waits for it
    HIPCHK(hipEventRecord(s->ev_prelude, s->stream));
    if (two) HIPCHK(hipStreamWaitEvent(s->stream2, s->ev_prelude, 0));
    for (size_t j0 = 0; j0 < P; j0 += npix_pass) {
        pp.j0 = (uint32_t)j0;
        pp.npix = (uint32_t)std::min(npix_pass, P - j0);
        for (uint32_t s0 = 0; s0 < prm->multisample; s0 += ns_pass) {
            pp.s0 = s0;
            pp.ns = std::min(ns_pass, prm->multisample - s0);
            {   // 2^gshift samples of a pixel side by side in the slot order (rgk_kernels.h PassParams); RGK_SAMPLE_GROUP = log2
                uint32_t g = s->tune.sample_group >= 0 ? (uint32_t)s->tune.sample_group : (uint32_t)RGK_SAMPLE_GROUP_DEFAULT;
                while (g && (pp.ns & ((1u << g) - 1u))) g--;
                pp.gshift = g;
                // the bundle walk (k_trace_camera_beam) for passes whose entry lists are not capped yet -- a frame's first round:
                // measured on the headline workload, camera launch 20.6 (per ray, uncapped) -> 17.2 ms (bundles), a one-round
                // frame 135.2 -> 130.8 ms; against CAPPED lists the per-ray walk is the faster one (16.1 vs 17.2: a bundle tests
                // every triangle it meets against all 8 rays, 2.86 tests per ray instead of 2.57, at half the occupancy)
                const bool lists_capped = cap_entries && (size_t)pp.j0 + pp.npix <= s->entry_capped;
                pp.beam = (s->tune.beam == 2 || (s->tune.beam == 1 && !lists_capped)) ? 1u : 0u;
            }
            const uint32_t n0 = pp.npix * pp.ns;
            // which lane: its stream, its half of every workspace array, its counter blocks; a lane's previous pass is harvested
            // (waited for, its counters added up) before the next one is queued on it
            lane = two ? (int)(pass_index & 1u) : 0;
            pass_index++;
            st = lane ? s->stream2 : s->stream;
            if (pending[lane] && (rc = harvest(lane))) return rc;
            const size_t off = (size_t)lane * lane_cap;
            float4* const w_rayA[2] = {s->rayA[0].p + off, s->rayA[1].p + off};
            float4* const w_rayB[2] = {s->rayB[0].p + off, s->rayB[1].p + off};
            float4 *const w_hit = s->hit.p + off, *const w_thr = s->thr.p + off, *const w_tot = s->tot.p + off;
            float4 *const w_shA = s->shA.p + off, *const w_shB = s->shB.p + off, *const w_shC = s->shC.p + off;
            pp.light = s->light.p + off; pp.generic = s->generic.p + off;
            uint32_t* cn = s->counters.p + (size_t)lane * 2 * RGK_CNT_TOTAL; // camera-phase counters
            uint32_t* cl = cn + RGK_CNT_TOTAL;                               // light-phase counters
            uint32_t& stage_target = stage_targets[lane];
            pend_n0[lane] = n0;
            RgkTraceCfg tcl = s->tcfg; // (the lane's own overflow area of the traversal stack)
            if (tcl.ovf) tcl.ovf += (size_t)lane * s->ovf_lane;
            if (R > 0) {
                // light sub-path first (its sampler dimensions are fixed), splats straight into the accumulator
                rgk_launch_set_bound(n0, n0);
                TIMED(RGK_K_OTHER, rgk_launch_init_counters(st, cl, 0u)); // (k_raygen_light queues the light rays that can touch the scene's box)
                TIMED(RGK_K_LIGHT_SHADE, rgk_launch_raygen_light(st, s->dev, cam, pp, w_rayA[0], w_rayB[0], w_thr, cl));
                for (uint32_t k = 0; k < R; k++) {
                    int q = k & 1;
                    TIMED(RGK_K_LIGHT_TRACE, rgk_launch_trace_closest(st, s->dev, tcl, count_stats, w_rayA[q], w_rayB[q], nullptr, w_hit,
                                                      cl + RGK_CNT_QUEUE + k, cl + RGK_CNT_FETCH_T + k, s->stats.p));
                    TIMED(RGK_K_LIGHT_SHADE, rgk_launch_list_hits(st, w_hit, cl + RGK_CNT_QUEUE + k, s->hitlist.p, cl + RGK_CNT_HITS + k));
                    TIMED(RGK_K_LIGHT_SHADE, rgk_launch_shade_light(st, s->dev, cam, pp, k, w_rayA[q], w_rayB[q], w_hit, w_thr,
                                                    w_rayA[q ^ 1], w_rayB[q ^ 1], w_shA, w_shB, w_shC, cl));
                    TIMED(RGK_K_LIGHT_SPLAT, rgk_launch_trace_shadow(st, s->dev, tcl, count_stats, w_shA, w_shB, w_shC, nullptr, nullptr,
                                                     RGK_SHADOW_SPLAT, d_accum_rgb, cl + RGK_CNT_SHADOW + k, cl + RGK_CNT_FETCH_S + k, s->stats.p));
                }
            }
            {   // the camera path: the same pipeline for uni- and bidirectional rounds (R > 0: vertices with connections take
                // the record route -- k_shade<BDPT> -> k_connect -> k_trace_shadow_jobs -- beside the plain NEE rays)
                TIMED(RGK_K_OTHER, rgk_launch_init_counters(st, cn, n0));
                uint32_t ub = n0; // upper bound on bounce b's queue
                for (uint32_t b = 0; b < prm->depth && ub > 0; b++) {
                    int q = b & 1;
                    rgk_launch_set_bound(ub, ub);
                    if (b == 0) // no ray queue at bounce 0: the camera ray of slot i is made where it is traced and shaded
                        TIMED(RGK_K_TRACE_CAMERA, rgk_launch_trace_camera(st, s->dev, cam, pp, tcl, count_stats, w_hit, cn + RGK_CNT_QUEUE, cn + RGK_CNT_FETCH_T, s->stats.p));
                    else
                        TIMED(RGK_K_TRACE_CLOSEST, rgk_launch_trace_closest(st, s->dev, tcl, count_stats, w_rayA[q], w_rayB[q], nullptr, w_hit,
                                                          cn + RGK_CNT_QUEUE + b, cn + RGK_CNT_FETCH_T + b, s->stats.p));
                    if (b == 0 && (cap_entries || light_entry)) {
                        // once per frame and pixel range (later rounds and sample ranges reuse it): how far the first hits of each
                        // pixel group lie -> camera-ray entry lists capped behind them, and where the group's shadow rays can go
                        const bool need_cap = cap_entries && (size_t)pp.j0 + pp.npix > s->entry_capped;
                        const bool need_light = light_entry && (size_t)pp.j0 + pp.npix > s->lentry_done;
                        if (need_cap || need_light) TIMED(RGK_K_OTHER, rgk_launch_group_trange(st, pp, w_hit, s->trange.p));
                        if (need_cap) {
                            const uint32_t g_first = pp.j0 >> RGK_ENTRY_SHIFT, g_last = (uint32_t)(((size_t)pp.j0 + pp.npix + RGK_ENTRY_PIX - 1) >> RGK_ENTRY_SHIFT);
                            TIMED(RGK_K_OTHER, rgk_launch_entry_points(st, s->dev, cam, prm->xres, prm->yres, s->pix_xy.p, (uint32_t)P, g_first, g_last - g_first, s->trange.p, s->entry.p, s->entry_cap.p));
                            s->entry_capped = (size_t)pp.j0 + pp.npix;
                        }
                    }
                    if (b == 0 && light_entry) {
                        pp.lentry = s->lentry.p; pp.lbox = s->lbox.p;
                        if ((size_t)pp.j0 + pp.npix > s->lentry_done) {
                            TIMED(RGK_K_OTHER, rgk_launch_light_entry_points(st, s->dev, cam, pp, (uint32_t)P, s->trange.p, s->lentry.p, s->lbox.p));
                            s->lentry_done = (size_t)pp.j0 + pp.npix;
                            if (s->tune.debug_bvh) { // how many pixel groups got light-side entry nodes below the root
                                const size_t g0 = pp.j0 >> RGK_ENTRY_SHIFT, g1 = ((size_t)pp.j0 + pp.npix + RGK_ENTRY_PIX - 1) >> RGK_ENTRY_SHIFT;
                                std::vector<int> he((g1 - g0) * RGK_ENTRY_K);
                                HIPCHK(hipStreamSynchronize(st));
                                HIPCHK(hipMemcpy(he.data(), s->lentry.p + g0 * RGK_ENTRY_K, he.size() * sizeof(int), hipMemcpyDeviceToHost));
                                size_t below = 0, total_e = 0;
                                for (size_t g = 0; g < g1 - g0; g++) { if (he[g * RGK_ENTRY_K] != 0) below++; for (int k = 0; k < RGK_ENTRY_K; k++) total_e += he[g * RGK_ENTRY_K + k] != 0x7fffffff; }
                                std::fprintf(stderr, "[rgk] light-side entry nodes: %zu of %zu pixel groups start below the root, %.2f entries per group\n", below, g1 - g0, (double)total_e / (double)(g1 - g0));
                            }
                        }
                    }
                    TIMED(b == 0 ? RGK_K_SHADE_FIRST : RGK_K_SHADE, rgk_launch_shade(st, s->dev, cam, pp, b, w_rayA[q], w_rayB[q], w_hit, w_thr, w_tot,
                                              w_rayA[q ^ 1], w_rayB[q ^ 1], w_shA, w_shB, w_shC, cn, R > 0));
                    if (R > 0) TIMED(RGK_K_CONNECT, rgk_launch_connect(st, s->dev, pp, b, s->jobs.p, s->rads.p, cn));
                    if (b == 0 && light_entry)
                        TIMED(RGK_K_SHADOW_FIRST, rgk_launch_trace_shadow_first(st, s->dev, pp, tcl, count_stats, w_shA, w_shB, w_shC, w_tot,
                                                               cn + RGK_CNT_SHADOW + b, cn + RGK_CNT_FETCH_S + b, s->stats.p));
                    else
                        TIMED(RGK_K_SHADOW, rgk_launch_trace_shadow(st, s->dev, tcl, count_stats, w_shA, w_shB, w_shC, w_tot, nullptr,
                                                         RGK_SHADOW_ADD, nullptr, cn + RGK_CNT_SHADOW + b, cn + RGK_CNT_FETCH_S + b, s->stats.p));
                    if (R > 0) // (after the plain rays: both add into the slot sums, a slot has a vertex in ONE of the two queues)
                        TIMED(RGK_K_SHADOW_JOBS, rgk_launch_trace_shadow_jobs(st, s->dev, pp, tcl, count_stats, s->jobs.p, s->rads.p, w_tot,
                                                              cn + RGK_CNT_CONN + b, cn + RGK_CNT_FETCH_J + b, s->stats.p));
                    if ((rc = stage_mark(stage_target + b + 1))) return rc;
                    if (track && b >= 3 && (b & 1) && b + 1 < prm->depth && (rc = queue_len(cn + RGK_CNT_QUEUE + b + 1, ub))) return rc;
                }
            }
            TIMED(RGK_K_RESOLVE, rgk_launch_resolve(st, pp, w_tot, s->pixsum.p, d_accum_rgb, d_accum_count));
            stage_target += std::max(1u, prm->depth);
            if ((rc = stage_mark(stage_target))) return rc;
            HIPCHK(hipMemcpyAsync(s->h_counters + (size_t)lane * 2 * RGK_CNT_TOTAL, cn, 2 * RGK_CNT_TOTAL * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            pending[lane] = true;
            if (!two && (rc = harvest(lane))) return rc;
        }
    }
    for (int l = 0; l < 2; l++) if (pending[l] && (rc = harvest(l))) return rc;
    HIPCHK(hipGetLastError());
    s->prog_rounds++;
    if (counters) {
        counters->paths = (uint64_t)P * prm->multisample;
        counters->path_rays = path_rays;
        counters->shadow_rays = shadow_rays;
        if (count_stats) {
            unsigned long long h[8];
            HIPCHK(hipMemcpy(h, s->stats.p, sizeof(h), hipMemcpyDeviceToHost));
            counters->node_visits = h[0]; counters->tri_tests = h[1]; counters->shadow_node_visits = h[2]; counters->shadow_tri_tests = h[3];
        }
        for (auto& e : evs) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
            rgk_kernel_stat& ks = counters->kernel[e.cls];
            ks.ms += ms; ks.launches++;
            const int k = e.cls;
            if (k == RGK_K_TRACE_CAMERA || k == RGK_K_TRACE_CLOSEST || k == RGK_K_LIGHT_TRACE) { counters->ms_trace += ms; counters->n_trace_launches++; }
            else if (k == RGK_K_SHADOW_FIRST || k == RGK_K_SHADOW || k == RGK_K_SHADOW_JOBS || k == RGK_K_LIGHT_SPLAT) { counters->ms_shadow += ms; counters->n_shadow_launches++; }
            else if (k == RGK_K_SHADE_FIRST || k == RGK_K_SHADE || k == RGK_K_CONNECT || k == RGK_K_LIGHT_SHADE) { counters->ms_shade += ms; counters->n_shade_launches++; }
            else counters->ms_other += ms;
        }
        for (int k = 0; k < RGK_K_COUNT; k++) counters->kernel[k].units = units[k];
        {   // per-kernel traversal counters: differences between consecutive read-backs
            std::array<unsigned long long, 4> prev{};
            for (auto& sn : snaps) {
                rgk_kernel_stat& ks = counters->kernel[sn.first];
                const bool shadow = !(sn.first == RGK_K_TRACE_CAMERA || sn.first == RGK_K_TRACE_CLOSEST || sn.first == RGK_K_LIGHT_TRACE);
                ks.node_visits += sn.second[shadow ? 2 : 0] - prev[shadow ? 2 : 0];
                ks.tri_tests += sn.second[shadow ? 3 : 1] - prev[shadow ? 3 : 1];
                prev = sn.second;
            }
        }
    }
    return RGK_OK;
#undef TIMED
}

int rgk_render_round(rgk_scene* s, const rgk_camera* camera, const rgk_params* prm, const rgk_tile* tiles, uint32_t n_tiles,
                     float* accum_rgb, uint32_t* accum_count, rgk_counters* counters) {
    if (!s || !prm || !accum_rgb || !accum_count) return fail(RGK_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(s->device));
    const size_t P = (size_t)prm->xres * prm->yres;
    float* d_rgb = nullptr;
    uint32_t* d_cnt = nullptr;
    HIPCHK(hipMalloc((void**)&d_rgb, P * 3 * sizeof(float)));
    hipError_t e = hipMalloc((void**)&d_cnt, P * sizeof(uint32_t));
    if (e != hipSuccess) { (void)hipFree(d_rgb); return fail(RGK_ERR_OOM, "hipMalloc: %s", hipGetErrorString(e)); }
    int rc = RGK_OK;
    auto chk = [&](hipError_t x, const char* what) { if (x != hipSuccess && rc == RGK_OK) rc = fail(RGK_ERR_DEVICE, "%s: %s", what, hipGetErrorString(x)); };
    chk(hipMemcpy(d_rgb, accum_rgb, P * 3 * sizeof(float), hipMemcpyHostToDevice), "upload accumulator");
    chk(hipMemcpy(d_cnt, accum_count, P * sizeof(uint32_t), hipMemcpyHostToDevice), "upload counts");
    if (rc == RGK_OK) rc = rgk_render_round_device(s, camera, prm, tiles, n_tiles, d_rgb, d_cnt, counters);
    if (rc == RGK_OK) {
        chk(hipMemcpy(accum_rgb, d_rgb, P * 3 * sizeof(float), hipMemcpyDeviceToHost), "download accumulator");
        chk(hipMemcpy(accum_count, d_cnt, P * sizeof(uint32_t), hipMemcpyDeviceToHost), "download counts");
    }
    (void)hipFree(d_rgb);
    (void)hipFree(d_cnt);
    return rc;
}

int rgk_trace_closest(rgk_scene* s, uint32_t n, const float* rays, const int32_t* ignore, rgk_hit* hits, rgk_counters* counters) {
    if (!s || (!rays && n) || (!hits && n)) return fail(RGK_ERR_INVALID, "null argument");
    if (counters) std::memset(counters, 0, sizeof(*counters));
    if (n == 0) return RGK_OK;
    HIPCHK(hipSetDevice(s->device));
    int rc;
    if ((rc = ensure_workspace(s, n)) || (rc = s->nearfar.alloc(n)) || (rc = s->scratch_f.alloc((size_t)8 * n)) || (rc = s->scratch_u.alloc((size_t)5 * n)))
        return rc;
    hipStream_t st = s->stream;
    HIPCHK(hipMemcpyAsync(s->scratch_f.p, rays, (size_t)8 * n * sizeof(float), hipMemcpyHostToDevice, st));
    int32_t* d_ign = nullptr;
    if (ignore) {
        d_ign = (int32_t*)s->scratch_u.p;
        HIPCHK(hipMemcpyAsync(d_ign, ignore, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipMemsetAsync(s->stats.p, 0, 8 * sizeof(unsigned long long), st));
    rgk_launch_init_counters(st, s->counters.p, n);
    rgk_launch_pack_rays(st, n, s->scratch_f.p, d_ign, s->rayA[0].p, s->rayB[0].p, s->nearfar.p);
    rgk_launch_set_bound(n, n);
    hipEvent_t e0 = nullptr, e1 = nullptr; // kernel time of the traversal alone, for counters->ms_trace
    if (counters) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, st)); }
    rgk_launch_trace_closest(st, s->dev, s->tcfg, false, s->rayA[0].p, s->rayB[0].p, s->nearfar.p, s->hit.p,
                             s->counters.p + RGK_CNT_QUEUE, s->counters.p + RGK_CNT_FETCH_T, s->stats.p);
    if (counters) {
        HIPCHK(hipEventRecord(e1, st));
        // the counting variant runs separately so that the timed launch is the kernel a round runs
        rgk_launch_init_counters(st, s->counters.p, n);
        rgk_launch_trace_closest(st, s->dev, s->tcfg, true, s->rayA[0].p, s->rayB[0].p, s->nearfar.p, s->hit.p,
                                 s->counters.p + RGK_CNT_QUEUE, s->counters.p + RGK_CNT_FETCH_T, s->stats.p);
    }
    rgk_hit* d_hits = (rgk_hit*)s->scratch_u.p; // 5 dwords per hit; reuses the ignore buffer after the trace
    rgk_launch_unpack_hits(st, n, s->hit.p, d_hits);
    HIPCHK(hipMemcpyAsync(hits, d_hits, (size_t)n * sizeof(rgk_hit), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    if (counters) {
        unsigned long long h[8];
        HIPCHK(hipMemcpy(h, s->stats.p, sizeof(h), hipMemcpyDeviceToHost));
        counters->node_visits = h[0]; counters->tri_tests = h[1]; counters->path_rays = n;
        if (s->tune.debug_util) // lane occupancy per phase of the walker: lane-visits / (64 x wave iterations)
            std::fprintf(stderr, "[rgk util] rays %u  node visits %llu in %llu wave iterations (%.3f of lanes)  triangle tests %llu in %llu (%.3f)  outer iterations %llu  refills %llu\n",
                         n, h[0], h[4], h[4] ? (double)h[0] / (64.0 * (double)h[4]) : 0.0, h[1], h[5], h[5] ? (double)h[1] / (64.0 * (double)h[5]) : 0.0, h[6], h[7]);
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        counters->ms_trace = ms; counters->n_trace_launches = 1;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    return RGK_OK;
}

int rgk_trace_visibility(rgk_scene* s, uint32_t n, const float* a, const float* b, uint8_t* visible, rgk_counters* counters) {
    if (!s || (n && (!a || !b || !visible))) return fail(RGK_ERR_INVALID, "null argument");
    if (counters) std::memset(counters, 0, sizeof(*counters));
    if (n == 0) return RGK_OK;
    HIPCHK(hipSetDevice(s->device));
    int rc;
    if ((rc = ensure_workspace(s, n)) || (rc = s->scratch_f.alloc((size_t)6 * n)) || (rc = s->scratch_u.alloc(n))) return rc;
    hipStream_t st = s->stream;
    HIPCHK(hipMemcpyAsync(s->scratch_f.p, a, (size_t)3 * n * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->scratch_f.p + (size_t)3 * n, b, (size_t)3 * n * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(s->stats.p, 0, 8 * sizeof(unsigned long long), st));
    rgk_launch_init_counters(st, s->counters.p, n);
    rgk_launch_pack_visibility(st, s->dev, n, s->scratch_f.p, s->scratch_f.p + (size_t)3 * n, s->shA.p, s->shB.p, s->shC.p);
    rgk_launch_set_bound(n, n);
    rgk_launch_trace_shadow(st, s->dev, s->tcfg, counters != nullptr, s->shA.p, s->shB.p, s->shC.p, s->tot.p, (uint8_t*)s->scratch_u.p,
                            RGK_SHADOW_ADD, nullptr, s->counters.p + RGK_CNT_QUEUE, s->counters.p + RGK_CNT_FETCH_S, s->stats.p);
    HIPCHK(hipMemcpyAsync(visible, s->scratch_u.p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    if (counters) {
        unsigned long long h[8];
        HIPCHK(hipMemcpy(h, s->stats.p, sizeof(h), hipMemcpyDeviceToHost));
        counters->shadow_node_visits = h[2]; counters->shadow_tri_tests = h[3]; counters->shadow_rays = n;
    }
    return RGK_OK;
}

int rgk_bxdf_value(rgk_scene* s, uint32_t n, uint32_t route, const uint32_t* mat, const float* Vi, const float* Vr, const float* uv, float* out_rgb) {
    if (!s || (n && (!mat || !Vi || !Vr || !uv || !out_rgb))) return fail(RGK_ERR_INVALID, "null argument");
    if (n == 0) return RGK_OK;
    for (uint32_t i = 0; i < n; i++) if (mat[i] >= s->n_materials) return fail(RGK_ERR_INVALID, "material index out of range");
    HIPCHK(hipSetDevice(s->device));
    DevBuf<uint32_t> dm; DevBuf<float> dvi, dvr, duv, dout;
    int rc;
    if (!(rc = up(dm, mat, n)) && !(rc = up(dvi, Vi, (size_t)3 * n)) && !(rc = up(dvr, Vr, (size_t)3 * n)) && !(rc = up(duv, uv, (size_t)2 * n)) && !(rc = dout.alloc((size_t)3 * n))) {
        rgk_launch_bxdf_value(s->stream, s->dev, n, route, dm.p, dvi.p, dvr.p, duv.p, dout.p);
        if (hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(RGK_ERR_DEVICE, "bxdf value kernel failed");
        else rc = down(out_rgb, dout, (size_t)3 * n);
    }
    dm.release(); dvi.release(); dvr.release(); duv.release(); dout.release();
    return rc;
}

int rgk_bxdf_sample(rgk_scene* s, uint32_t n, uint32_t route, const uint32_t* mat, const float* Vi, const float* uv, const float* u, float* out_dir,
                    float* out_weight, uint8_t* may_leak) {
    if (!s || (n && (!mat || !Vi || !uv || !u || !out_dir || !out_weight || !may_leak))) return fail(RGK_ERR_INVALID, "null argument");
    if (n == 0) return RGK_OK;
    for (uint32_t i = 0; i < n; i++) if (mat[i] >= s->n_materials) return fail(RGK_ERR_INVALID, "material index out of range");
    HIPCHK(hipSetDevice(s->device));
    DevBuf<uint32_t> dm; DevBuf<float> dvi, duv, du, dd, dw; DevBuf<uint8_t> dl;
    int rc;
    if (!(rc = up(dm, mat, n)) && !(rc = up(dvi, Vi, (size_t)3 * n)) && !(rc = up(duv, uv, (size_t)2 * n)) && !(rc = up(du, u, (size_t)2 * n)) &&
        !(rc = dd.alloc((size_t)3 * n)) && !(rc = dw.alloc((size_t)3 * n)) && !(rc = dl.alloc(n))) {
        rgk_launch_bxdf_sample(s->stream, s->dev, n, route, dm.p, dvi.p, duv.p, du.p, dd.p, dw.p, dl.p);
        if (hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(RGK_ERR_DEVICE, "bxdf sample kernel failed");
        else if (!(rc = down(out_dir, dd, (size_t)3 * n)) && !(rc = down(out_weight, dw, (size_t)3 * n))) rc = down(may_leak, dl, n);
    }
    dm.release(); dvi.release(); duv.release(); du.release(); dd.release(); dw.release(); dl.release();
    return rc;
}

int rgk_texture_sample(rgk_scene* s, uint32_t n, const int32_t* tex, const float* uv, float* rgb, float* slope_right, float* slope_bottom) {
    if (!s || (n && (!tex || !uv || !rgb || !slope_right || !slope_bottom))) return fail(RGK_ERR_INVALID, "null argument");
    if (n == 0) return RGK_OK;
    for (uint32_t i = 0; i < n; i++) if (tex[i] >= (int32_t)s->n_textures) return fail(RGK_ERR_INVALID, "texture index out of range");
    HIPCHK(hipSetDevice(s->device));
    DevBuf<int32_t> dt; DevBuf<float> duv, drgb, dr, db;
    int rc;
    if (!(rc = up(dt, tex, n)) && !(rc = up(duv, uv, (size_t)2 * n)) && !(rc = drgb.alloc((size_t)3 * n)) && !(rc = dr.alloc(n)) && !(rc = db.alloc(n))) {
        rgk_launch_texture_sample(s->stream, s->dev, n, s->texrefs.p, dt.p, duv.p, drgb.p, dr.p, db.p);
        if (hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(RGK_ERR_DEVICE, "texture sample kernel failed");
        else if (!(rc = down(rgb, drgb, (size_t)3 * n)) && !(rc = down(slope_right, dr, n))) rc = down(slope_bottom, db, n);
    }
    dt.release(); duv.release(); drgb.release(); dr.release(); db.release();
    return rc;
}

int rgk_libm_eval(int fn, uint32_t n, const float* a, const float* b, float* out) {
    if (n && (!a || !out || (fn == 4 && !b))) return fail(RGK_ERR_INVALID, "null argument");
    if (fn < 0 || fn > 4) return fail(RGK_ERR_INVALID, "unknown function");
    if (n == 0) return RGK_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RGK_ERR_NO_DEVICE, "no HIP device visible");
    DevBuf<float> da, db, dout;
    int rc;
    if (!(rc = up(da, a, n)) && !(rc = up(db, b ? b : a, n)) && !(rc = dout.alloc(n))) {
        rgk_launch_libm_eval(nullptr, fn, n, da.p, db.p, dout.p);
        rc = down(out, dout, n);
    }
    da.release(); db.release(); dout.release();
    return rc;
}

int rgk_sampler_eval(uint32_t n, const uint32_t* seed, const uint32_t* index, const uint32_t* dim, int is2d, float* out) {
    if (n && (!seed || !index || !dim || !out)) return fail(RGK_ERR_INVALID, "null argument");
    if (n == 0) return RGK_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RGK_ERR_NO_DEVICE, "no HIP device visible");
    std::vector<DevHaltonDim> hd;
    std::vector<uint16_t> hp;
    build_halton(hd, hp);
    DevBuf<DevHaltonDim> dh;
    DevBuf<uint16_t> dp;
    DevBuf<uint32_t> ds_, di, dd;
    DevBuf<float> dout;
    int rc = RGK_OK;
    std::vector<uint32_t> vs(seed, seed + n), vi(index, index + n), vd(dim, dim + n);
    if (!(rc = dh.upload(hd)) && !(rc = dp.upload(hp)) && !(rc = ds_.upload(vs)) && !(rc = di.upload(vi)) && !(rc = dd.upload(vd)) &&
        !(rc = dout.alloc((size_t)2 * n))) {
        DevScene sc{};
        sc.hdims = dh.p; sc.hperm = dp.p;
        rgk_launch_sampler_eval(nullptr, sc, n, ds_.p, di.p, dd.p, is2d, dout.p);
        hipError_t e = hipMemcpy(out, dout.p, (size_t)2 * n * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RGK_ERR_DEVICE, "sampler eval: %s", hipGetErrorString(e));
    }
    dh.release(); dp.release(); ds_.release(); di.release(); dd.release(); dout.release();
    return rc;
}

} // extern "C"
