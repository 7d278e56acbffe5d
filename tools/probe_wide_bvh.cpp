// Probe: node visits per ray of a 4-wide BVH with distance-sorted children (what the product walks) against an 8-wide BVH
// walked in octant order (slot ^ ~octant, one stack entry per node) and in distance order, on the Sponza proxy.
// g++ -O2 -std=c++17 probe.cpp -o probe && ./probe tris.bin rays.bin
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>
struct V3 { float x, y, z; };
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline float comp(V3 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; }
struct Box { V3 lo{1e30f, 1e30f, 1e30f}, hi{-1e30f, -1e30f, -1e30f};
    void add(V3 p) { lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)}; hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)}; }
    void add(const Box& b) { add(b.lo); add(b.hi); }
    float area() const { V3 d = hi - lo; return 2 * (d.x * d.y + d.y * d.z + d.z * d.x); }
    V3 c() const { return (lo + hi) * 0.5f; } };
struct Tri { V3 a, b, c; };
struct N2 { Box box; int l = -1, r = -1, first = 0, cnt = 0; };
std::vector<Tri> T; std::vector<int> order; std::vector<N2> n2;
std::vector<Box> tb;
int build(int b, int e) {
    N2 n; for (int i = b; i < e; i++) n.box.add(tb[order[i]]);
    int id = (int)n2.size(); n2.push_back(n);
    int cnt = e - b;
    if (cnt <= 4) {
        // SAH leaf test
    }
    float best = 1e30f; int bax = -1, bsplit = 0; const int NB = 16;
    Box cb; for (int i = b; i < e; i++) cb.add(tb[order[i]].c());
    for (int ax = 0; ax < 3; ax++) {
        float lo = comp(cb.lo, ax), hi = comp(cb.hi, ax); if (hi - lo < 1e-9f) continue;
        Box bb[NB]; int bc[NB] = {0};
        for (int i = b; i < e; i++) { int k = std::min(NB - 1, (int)((comp(tb[order[i]].c(), ax) - lo) / (hi - lo) * NB)); bb[k].add(tb[order[i]]); bc[k]++; }
        float la[NB], ra[NB]; int lc[NB], rc[NB]; Box acc; int c = 0;
        for (int k = 0; k < NB; k++) { if (bc[k]) acc.add(bb[k]); c += bc[k]; la[k] = c ? acc.area() : 0; lc[k] = c; }
        acc = Box(); c = 0;
        for (int k = NB - 1; k >= 0; k--) { if (bc[k]) acc.add(bb[k]); c += bc[k]; ra[k] = c ? acc.area() : 0; rc[k] = c; }
        for (int k = 0; k < NB - 1; k++) { if (!lc[k] || !rc[k + 1]) continue; float cost = la[k] * lc[k] + ra[k + 1] * rc[k + 1]; if (cost < best) { best = cost; bax = ax; bsplit = k; } }
    }
    if (bax < 0 || (cnt <= 4 && best / n2[id].box.area() + 1.0f >= (float)cnt)) { n2[id].first = b; n2[id].cnt = cnt; if (cnt > 4 && bax < 0) { /* fall through to median */ } else return id; }
    int mid;
    if (bax >= 0) {
        float lo = comp(cb.lo, bax), hi = comp(cb.hi, bax);
        mid = (int)(std::partition(order.begin() + b, order.begin() + e, [&](int t) { return std::min(NB - 1, (int)((comp(tb[t].c(), bax) - lo) / (hi - lo) * NB)) <= bsplit; }) - order.begin());
    } else mid = (b + e) / 2;
    if (mid == b || mid == e) mid = (b + e) / 2;
    n2[id].cnt = 0;
    int l = build(b, mid); int r = build(mid, e);
    n2[id].l = l; n2[id].r = r;
    return id;
}

// ---- insertion-based optimisation of the BVH2 (Bittner et al. 2013, simplified): take a node out, put it back where the
// surface-area cost of the tree grows least.
std::vector<int> par;
static void refit_up(int n) { while (n >= 0) { Box b = n2[n2[n].l].box; b.add(n2[n2[n].r].box); n2[n].box = b; n = par[n]; } }
static double sah_cost() { double c = 0; double ra = n2[0].box.area(); for (auto& n : n2) c += (n.l >= 0 ? 1.0 : (double)n.cnt) * n.box.area() / ra; return c; }
static void optimise(int rounds, float frac) {
    par.assign(n2.size(), -1);
    for (int i = 0; i < (int)n2.size(); i++) if (n2[i].l >= 0) { par[n2[i].l] = i; par[n2[i].r] = i; }
    std::mt19937 rng(7);
    for (int it = 0; it < rounds; it++) {
        std::vector<int> cand;
        for (int i = 1; i < (int)n2.size(); i++) if (par[i] > 0) cand.push_back(i);
        // the nodes with the largest area first (they cost the most), a fraction of them per round
        std::sort(cand.begin(), cand.end(), [&](int a, int b) { return n2[a].box.area() > n2[b].box.area(); });
        size_t take = (size_t)(cand.size() * frac);
        if (it & 1) { std::shuffle(cand.begin(), cand.end(), rng); }
        cand.resize(take);
        for (int n : cand) {
            int p = par[n]; if (p <= 0) continue; int g = par[p]; if (g < 0) continue;
            int s = n2[p].l == n ? n2[p].r : n2[p].l;
            // take n (and p) out: s moves up
            if (n2[g].l == p) n2[g].l = s; else n2[g].r = s;
            par[s] = g; refit_up(g);
            // best place: branch and bound over (induced cost so far) + area(union)
            const Box nb = n2[n].box; const float na = nb.area();
            float best = 1e30f; int bx = -1;
            std::vector<std::pair<float, int>> pq{{0.f, 0}};
            while (!pq.empty()) {
                std::pop_heap(pq.begin(), pq.end(), [](auto& a, auto& b) { return a.first > b.first; });
                auto [ind, x] = pq.back(); pq.pop_back();
                if (ind + na >= best) break;
                Box u = n2[x].box; u.add(nb);
                const float direct = u.area(), total = ind + direct;
                if (total < best) { best = total; bx = x; }
                const float child_ind = total - n2[x].box.area();
                if (n2[x].l >= 0 && child_ind + na < best) {
                    pq.push_back({child_ind, n2[x].l}); std::push_heap(pq.begin(), pq.end(), [](auto& a, auto& b) { return a.first > b.first; });
                    pq.push_back({child_ind, n2[x].r}); std::push_heap(pq.begin(), pq.end(), [](auto& a, auto& b) { return a.first > b.first; });
                }
            }
            // put it back: p becomes the parent of (bx, n) where bx was
            int xp = par[bx];
            if (xp < 0) { // bx is the root: keep node 0 the root by swapping contents
                bx = s; xp = par[bx]; // fall back: restore
            }
            if (n2[xp].l == bx) n2[xp].l = p; else n2[xp].r = p;
            par[p] = xp; n2[p].l = bx; n2[p].r = n; n2[p].cnt = 0; par[bx] = p; par[n] = p;
            refit_up(p);
        }
        fprintf(stderr, "  optimise round %d: SAH cost %.3f\n", it, sah_cost());
    }
}

// wide node
struct NW { Box box[8]; int child[8]; int n = 0; };  // child >= 0: wide node index; < 0: ~leaf n2 index
std::vector<NW> wide;
int collapse(int root2, int width, bool slots) {
    std::vector<int> open{root2};
    // open the largest-area inner child until `width` children
    for (;;) {
        if ((int)open.size() >= width) break;
        int bi = -1; float ba = -1;
        for (int i = 0; i < (int)open.size(); i++) if (n2[open[i]].l >= 0 && n2[open[i]].box.area() > ba) { ba = n2[open[i]].box.area(); bi = i; }
        if (bi < 0) break;
        int n = open[bi]; open[bi] = n2[n].l; open.push_back(n2[n].r);
    }
    int id = (int)wide.size(); wide.emplace_back();
    // slot assignment (8-wide): greedy by |dot(centroid offset, octant direction)|
    std::vector<int> slot(open.size(), -1);
    if (slots && width == 8) {
        Box nb; for (int c : open) nb.add(n2[c].box);
        V3 nc = nb.c();
        std::vector<std::array<float, 3>> cand; // (score, child, slot)
        std::vector<bool> usedc(open.size(), false), useds(8, false);
        for (int k = 0; k < (int)open.size(); k++) {
            float bs = -1e30f; int bc = -1, bsl = -1;
            for (int c = 0; c < (int)open.size(); c++) if (!usedc[c]) for (int s = 0; s < 8; s++) if (!useds[s]) {
                V3 d = n2[open[c]].box.c() - nc; V3 dir{(s & 1) ? 1.f : -1.f, (s & 2) ? 1.f : -1.f, (s & 4) ? 1.f : -1.f};
                float sc = dot(d, dir); if (sc > bs) { bs = sc; bc = c; bsl = s; }
            }
            usedc[bc] = true; useds[bsl] = true; slot[bc] = bsl;
        }
    } else for (int c = 0; c < (int)open.size(); c++) slot[c] = c;
    NW w; for (int s = 0; s < 8; s++) w.child[s] = INT32_MIN;
    w.n = (int)open.size();
    wide[id] = w;
    for (int c = 0; c < (int)open.size(); c++) {
        int n = open[c];
        wide[id].box[slot[c]] = n2[n].box;
        int ch = n2[n].l < 0 ? ~n : collapse(n, width, slots);
        wide[id].child[slot[c]] = ch;
    }
    return id;
}
static bool tri_hit(const Tri& t, V3 o, V3 d, float& tt) {
    V3 e1 = t.b - t.a, e2 = t.c - t.a, p = cross(d, e2); float det = dot(e1, p); if (std::fabs(det) < 1e-12f) return false;
    float inv = 1 / det; V3 s = o - t.a; float u = dot(s, p) * inv; if (u < 0 || u > 1) return false;
    V3 q = cross(s, e1); float v = dot(d, q) * inv; if (v < 0 || u + v > 1) return false;
    tt = dot(e2, q) * inv; return tt > 1e-4f;
}
static bool slab(const Box& b, V3 o, V3 inv, float tmax, float& tn) {
    float t0 = 0, t1 = tmax;
    for (int a = 0; a < 3; a++) { float x0 = (comp(b.lo, a) - comp(o, a)) * comp(inv, a), x1 = (comp(b.hi, a) - comp(o, a)) * comp(inv, a); if (x0 > x1) std::swap(x0, x1); t0 = std::max(t0, x0); t1 = std::min(t1, x1); }
    tn = t0; return t0 <= t1;
}
struct Stat { double nodes = 0, tris = 0, rays = 0; };
// mode 0: distance order; 1: octant order (slot ^ ~oct descending)
float trace(int root, V3 o, V3 d, int mode, Stat& st, int* hit_tri, const std::vector<int>* entries = nullptr) {
    V3 inv{1 / d.x, 1 / d.y, 1 / d.z}; int oct = (d.x < 0) | ((d.y < 0) << 1) | ((d.z < 0) << 2);
    float best = 1e30f; int bt = -1; std::vector<int> stack{root};
    if (entries) stack = *entries;
    while (!stack.empty()) {
        int n = stack.back(); stack.pop_back();
        if (n < 0) { const N2& lf = n2[~n]; for (int i = 0; i < lf.cnt; i++) { st.tris++; float t; if (tri_hit(T[order[lf.first + i]], o, d, t) && t < best) { best = t; bt = order[lf.first + i]; } } continue; }
        st.nodes++;
        const NW& w = wide[n];
        struct E { float t; int c; int key; } e[8]; int ne = 0;
        for (int s = 0; s < 8; s++) { if (w.child[s] == INT32_MIN) continue; float tn; if (slab(w.box[s], o, inv, best, tn)) e[ne++] = {tn, w.child[s], s ^ (7 - oct)}; }
        if (mode == 0) std::sort(e, e + ne, [](const E& a, const E& b) { return a.t > b.t; });           // far first (pushed), near popped first
        else std::sort(e, e + ne, [](const E& a, const E& b) { return a.key < b.key; });                 // low priority first
        for (int i = 0; i < ne; i++) stack.push_back(e[i].c);
    }
    st.rays++; if (hit_tri) *hit_tri = bt; return best;
}

// any-hit walk of the segment a -> b (shadow ray), optionally from an entry list; counts node visits
static bool occluded(int root, V3 a, V3 b, Stat& st, const std::vector<int>* entries = nullptr) {
    V3 d = b - a; float len = std::sqrt(dot(d, d)); d = d * (1 / len); V3 inv{1 / d.x, 1 / d.y, 1 / d.z};
    const float tmax = len * 0.999f;
    std::vector<int> stack{root}; if (entries) stack = *entries;
    st.rays++;
    while (!stack.empty()) {
        int n = stack.back(); stack.pop_back();
        if (n < 0) { const N2& lf = n2[~n]; for (int i = 0; i < lf.cnt; i++) { st.tris++; float t; if (tri_hit(T[order[lf.first + i]], a, d, t) && t < tmax) return true; } continue; }
        st.nodes++;
        const NW& w = wide[n]; float bt = 1e30f; int bi = -1; int hits[8]; int nh = 0;
        for (int s = 0; s < 8; s++) { if (w.child[s] == INT32_MIN) continue; float tn; if (slab(w.box[s], a, inv, tmax, tn)) { if (tn < bt) { bt = tn; bi = nh; } hits[nh++] = w.child[s]; } }
        for (int i = 0; i < nh; i++) if (i != bi) stack.push_back(hits[i]);
        if (bi >= 0) stack.push_back(hits[bi]);
    }
    return false;
}
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET); T.resize(sz / 36); if (fread(T.data(), 36, T.size(), f) != T.size()) return 1; fclose(f);
    f = fopen(argv[2], "rb"); fseek(f, 0, SEEK_END); sz = ftell(f); fseek(f, 0, SEEK_SET); std::vector<std::array<float, 6>> R(sz / 24); if (fread(R.data(), 24, R.size(), f) != R.size()) return 1; fclose(f);
    tb.resize(T.size()); order.resize(T.size());
    for (size_t i = 0; i < T.size(); i++) { tb[i].add(T[i].a); tb[i].add(T[i].b); tb[i].add(T[i].c); order[i] = (int)i; }
    int r2 = build(0, (int)T.size());
    printf("bvh2 nodes %zu, SAH cost %.3f\n", n2.size(), sah_cost());
    const int opt_rounds = argc > 3 ? atoi(argv[3]) : 0;
    if (opt_rounds) { optimise(opt_rounds, argc > 4 ? (float)atof(argv[4]) : 0.25f); printf("after %d reinsertion rounds: SAH cost %.3f\n", opt_rounds, sah_cost()); }
    std::mt19937 rng(1); std::uniform_real_distribution<float> U(0, 1);
    for (int cfg = 0; cfg < (opt_rounds ? 1 : 4); cfg++) {
        int width = cfg == 0 ? 4 : 8; int mode = cfg <= 1 ? 0 : 1; bool slots = cfg == 3 || cfg == 2;
        if (cfg == 2) slots = false; // octant order WITHOUT octant-aware slots: what not to do
        wide.clear(); int root = collapse(r2, width, slots);
        Stat cam, bnc; std::vector<std::array<float, 6>> B;
        rng.seed(1);
        for (auto& r : R) {
            V3 o{r[0], r[1], r[2]}, d{r[3], r[4], r[5]}; int ht; float t = trace(root, o, d, mode, cam, &ht);
            if (ht >= 0) { const Tri& tr = T[ht]; V3 n = cross(tr.b - tr.a, tr.c - tr.a); float l = std::sqrt(dot(n, n)); n = n * (1 / l); if (dot(n, d) > 0) n = n * -1.f;
                V3 p = o + d * t + n * 1e-3f; float u1 = U(rng), u2 = U(rng); float rr = std::sqrt(u1), ph = 6.2831853f * u2;
                V3 a = std::fabs(n.x) > 0.9f ? V3{0, 1, 0} : V3{1, 0, 0}; V3 t1 = cross(n, a); t1 = t1 * (1 / std::sqrt(dot(t1, t1))); V3 t2 = cross(n, t1);
                V3 bd = t1 * (rr * std::cos(ph)) + t2 * (rr * std::sin(ph)) + n * std::sqrt(std::max(0.f, 1 - u1)); B.push_back({p.x, p.y, p.z, bd.x, bd.y, bd.z}); }
        }
        for (auto& r : B) trace(root, {r[0], r[1], r[2]}, {r[3], r[4], r[5]}, mode, bnc, nullptr);
        if (cfg == 0 && getenv("PROBE_SHADOW")) {
            // shadow rays of a point light: all start at the light; grouped by the 2x2 block of camera rays whose hits they go to
            V3 L; sscanf(getenv("PROBE_SHADOW"), "%f,%f,%f", &L.x, &L.y, &L.z);
            const int W = 480, H = 270; const int K = 6; Stat s0, s1; double entries = 0, blocks = 0;
            std::vector<V3> hp(R.size()); std::vector<char> ok(R.size(), 0); Stat dummy;
            for (size_t i = 0; i < R.size(); i++) { auto& r = R[i]; V3 o{r[0], r[1], r[2]}, d{r[3], r[4], r[5]}; int ht; float t = trace(root, o, d, 0, dummy, &ht); if (ht >= 0) { hp[i] = o + d * (t * 0.9999f); ok[i] = 1; } }
            for (int by = 0; by + 1 < H; by += 2) for (int bx = 0; bx + 1 < W; bx += 2) {
                int idx[4] = {by * W + bx, by * W + bx + 1, (by + 1) * W + bx, (by + 1) * W + bx + 1};
                std::vector<std::pair<int, int>> list{{root, 0}};
                for (;;) { bool done = true;
                    for (size_t li = 0; li < list.size(); li++) { int n = list[li].first; if (n < 0) continue; const NW& w = wide[n]; std::vector<int> hitc; bool leafchild = false;
                        for (int sl = 0; sl < 8; sl++) { if (w.child[sl] == INT32_MIN) continue; bool any = false;
                            for (int q = 0; q < 4 && !any; q++) if (ok[idx[q]]) { V3 d = hp[idx[q]] - L; float len = std::sqrt(dot(d, d)); d = d * (1 / len); V3 inv{1 / d.x, 1 / d.y, 1 / d.z}; float tn; any = slab(w.box[sl], L, inv, len, tn); }
                            if (any) { hitc.push_back(w.child[sl]); if (w.child[sl] < 0) leafchild = true; } }
                        if (leafchild) continue;
                        if ((int)(list.size() - 1 + hitc.size()) <= K) { int dpt = list[li].second; list.erase(list.begin() + li); for (int c : hitc) list.push_back({c, dpt + 1}); done = false; break; } }
                    if (done) break; }
                std::vector<int> ent; for (auto& e : list) ent.push_back(e.first);
                for (int q = 0; q < 4; q++) if (ok[idx[q]]) { occluded(root, L, hp[idx[q]], s0); occluded(root, L, hp[idx[q]], s1, &ent); }
                blocks++; entries += list.size();
            }
            printf("shadow rays from the light to the camera hits: %.2f node visits, %.2f triangle tests per ray from the root; %.2f and %.2f from the block's entry list (%.2f entries)\n", s0.nodes / s0.rays, s0.tris / s0.rays, s1.nodes / s1.rays, s1.tris / s1.rays, entries / blocks);
        }
        if (cfg == 0 && getenv("PROBE_BOUNCE")) {
            // bounce rays: they start inside the box of their 2x2 block's first hits and go anywhere; per direction octant the
            // reachable region is that box extended to infinity on three sides -- descend while at most K children overlap it
            const int W = 480, H = 270; const int K = atoi(getenv("PROBE_BOUNCE")); Stat s0, s1; double entries = 0, lists = 0;
            std::vector<V3> hp(R.size()); std::vector<V3> hn(R.size()); std::vector<char> ok(R.size(), 0); Stat dummy;
            for (size_t i = 0; i < R.size(); i++) { auto& r = R[i]; V3 o{r[0], r[1], r[2]}, d{r[3], r[4], r[5]}; int ht; float t = trace(root, o, d, 0, dummy, &ht);
                if (ht >= 0) { const Tri& tr = T[ht]; V3 n = cross(tr.b - tr.a, tr.c - tr.a); n = n * (1 / std::sqrt(dot(n, n))); if (dot(n, d) > 0) n = n * -1.f; hp[i] = o + d * t + n * 1e-3f; hn[i] = n; ok[i] = 1; } }
            std::mt19937 rg(3); std::uniform_real_distribution<float> U(0, 1);
            for (int by = 0; by + 1 < H; by += 2) for (int bx = 0; bx + 1 < W; bx += 2) {
                int idx[4] = {by * W + bx, by * W + bx + 1, (by + 1) * W + bx, (by + 1) * W + bx + 1};
                Box bg; int nok = 0; for (int q = 0; q < 4; q++) if (ok[idx[q]]) { bg.add(hp[idx[q]]); nok++; }
                if (!nok) continue;
                std::vector<int> ent[8];
                for (int oc = 0; oc < 8; oc++) {
                    Box reg; reg.lo = {(oc & 1) ? -1e30f : bg.lo.x, (oc & 2) ? -1e30f : bg.lo.y, (oc & 4) ? -1e30f : bg.lo.z};
                    reg.hi = {(oc & 1) ? bg.hi.x : 1e30f, (oc & 2) ? bg.hi.y : 1e30f, (oc & 4) ? bg.hi.z : 1e30f};
                    std::vector<int> list{root};
                    for (;;) { bool done = true;
                        for (size_t li = 0; li < list.size(); li++) { int n = list[li]; if (n < 0) continue; const NW& w = wide[n]; std::vector<int> hitc; bool leafchild = false;
                            for (int sl = 0; sl < 8; sl++) { if (w.child[sl] == INT32_MIN) continue; const Box& b = w.box[sl];
                                bool ov = b.hi.x >= reg.lo.x && b.lo.x <= reg.hi.x && b.hi.y >= reg.lo.y && b.lo.y <= reg.hi.y && b.hi.z >= reg.lo.z && b.lo.z <= reg.hi.z;
                                if (ov) { hitc.push_back(w.child[sl]); if (w.child[sl] < 0) leafchild = true; } }
                            if (leafchild) continue;
                            if ((int)(list.size() - 1 + hitc.size()) <= K) { list.erase(list.begin() + li); for (int c : hitc) list.push_back(c); done = false; break; } }
                        if (done) break; }
                    ent[oc] = list; entries += list.size(); lists++;
                }
                for (int q = 0; q < 4; q++) if (ok[idx[q]]) for (int rep = 0; rep < 2; rep++) {
                    V3 n = hn[idx[q]]; float u1 = U(rg), u2 = U(rg); float rr = std::sqrt(u1), ph = 6.2831853f * u2;
                    V3 a = std::fabs(n.x) > 0.9f ? V3{0, 1, 0} : V3{1, 0, 0}; V3 t1 = cross(n, a); t1 = t1 * (1 / std::sqrt(dot(t1, t1))); V3 t2 = cross(n, t1);
                    V3 bd = t1 * (rr * std::cos(ph)) + t2 * (rr * std::sin(ph)) + n * std::sqrt(std::max(0.f, 1 - u1));
                    int oc = (bd.x < 0) | ((bd.y < 0) << 1) | ((bd.z < 0) << 2);
                    int h0, h1; trace(root, hp[idx[q]], bd, 0, s0, &h0); trace(root, hp[idx[q]], bd, 0, s1, &h1, &ent[oc]);
                    if (h0 != h1) { printf("MISMATCH\n"); }
                }
            }
            printf("bounce rays, octant entry lists (K = %d): %.2f node visits, %.2f triangle tests per ray from the root; %.2f and %.2f from the lists (%.2f entries per list)\n", K, s0.nodes / s0.rays, s0.tris / s0.rays, s1.nodes / s1.rays, s1.tris / s1.rays, entries / lists);
        }
        if (cfg == 0 && getenv("PROBE_ENTRY")) {
            // entry points: for each 2x2 block of the 480x270 ray grid (~ an 8x8 pixel block at 1080p), descend from the root while
            // at most K children are touched by ANY ray of the block; depth reached = node visits every ray of the block saves
            const int W = 480, H = 270; const int K = atoi(getenv("PROBE_ENTRY"));
            double blocks = 0, entries = 0; Stat est;
            for (int by = 0; by + 1 < H; by += 2) for (int bx = 0; bx + 1 < W; bx += 2) {
                int idx[4] = {by * W + bx, by * W + bx + 1, (by + 1) * W + bx, (by + 1) * W + bx + 1};
                float capt = 1e30f;
                if (getenv("PROBE_CAP")) { capt = 0; Stat dm; for (int q = 0; q < 4; q++) { auto& r = R[idx[q]]; int ht; float t = trace(root, {r[0], r[1], r[2]}, {r[3], r[4], r[5]}, 0, dm, &ht); capt = std::max(capt, ht >= 0 ? t * 1.02f : 1e30f); } }
                std::vector<std::pair<int, int>> list{{root, 0}}; // (node, depth)
                for (;;) {
                    // expand the first inner node whose expansion keeps the list within K
                    bool done = true;
                    for (size_t li = 0; li < list.size(); li++) {
                        int n = list[li].first; if (n < 0) continue;
                        const NW& w = wide[n]; std::vector<int> hitc;
                        for (int sl = 0; sl < 8; sl++) { if (w.child[sl] == INT32_MIN) continue; bool any = false;
                            for (int q = 0; q < 4 && !any; q++) { auto& r = R[idx[q]]; V3 o{r[0], r[1], r[2]}, d{r[3], r[4], r[5]}, inv{1 / d.x, 1 / d.y, 1 / d.z}; float tn; any = slab(w.box[sl], o, inv, capt, tn); }
                            if (any) hitc.push_back(w.child[sl]); }
                        bool leafchild = false; for (int c : hitc) if (c < 0) leafchild = true;
                        if (getenv("PROBE_NOLEAF") && leafchild) continue; // keep a node whose touched children include a leaf
                        if ((int)(list.size() - 1 + hitc.size()) <= K) { int dpt = list[li].second; list.erase(list.begin() + li); for (int c : hitc) list.push_back({c, dpt + 1}); done = false; break; }
                    }
                    if (done) break;
                }
                std::vector<int> ent; for (auto& e : list) ent.push_back(e.first);
                std::reverse(ent.begin(), ent.end());
                for (int q = 0; q < 4; q++) { auto& r = R[idx[q]]; trace(root, {r[0], r[1], r[2]}, {r[3], r[4], r[5]}, 0, est, nullptr, &ent); }
                blocks++; entries += list.size();
            }
            printf("entry points (K = %d): %.2f node visits per camera ray from the block's entry list (%.2f entries per block) against %.2f from the root; triangle tests %.2f against %.2f\n", K, est.nodes / est.rays, entries / blocks, cam.nodes / cam.rays, est.tris / est.rays, cam.tris / cam.rays);
        }
        printf("width %d %s%s: wide nodes %zu | camera rays: %.2f nodes %.2f tris per ray | bounce rays: %.2f nodes %.2f tris per ray\n", width, mode ? "octant order" : "distance order", slots ? " (octant slots)" : "",
               wide.size(), cam.nodes / cam.rays, cam.tris / cam.rays, bnc.nodes / bnc.rays, bnc.tris / bnc.rays);
    }
    return 0;
}
