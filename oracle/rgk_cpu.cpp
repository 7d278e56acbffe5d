// ORACLE -- test infrastructure only.  Nothing under oracle/ is linked, imported or
// executed by the product (rgk_amd/); only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use it, and only as the checker / timed CPU baseline.
//
// rgk_cpu: a dependency-free C++17 CPU restatement of the reference's hot path
// (Enhex/RGK "RGKrt"), function by function, with the reference file:line each part
// follows.  The reference itself cannot be built in this image (GLM, assimp, png++,
// OpenEXR are absent: SURVEY F4) and holds no tests or golden vectors, so:
//   * PINNED:   the Halton radical inverse, against vectors produced by compiling the
//               reference's own external/halton_sampler.h (oracle/_ref, tests/golden).
//   * UNPINNED: everything that depends on GLM / libstdc++ <random> semantics
//               (rgk_math.hpp) and the integrator as a whole -- "parity unpinned";
//               closed-form known-answer tests in tests/ stand in for fixtures.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off, no -ffast-math).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <mutex>
#include <random>
#include <thread>
#include <tuple>
#include <vector>

#include "../include/rgk.h"
#include "rgk_math.hpp"

using namespace orc;

namespace {

// ---------------------------------------------------------------- radiance.hpp:6-88
struct Color {
    float r = 0, g = 0, b = 0;
    Color() {}
    Color(float r_, float g_, float b_) : r(r_), g(g_), b(b_) {}
};
inline Color operator*(float q, const Color& c) { return Color(q * c.r, q * c.g, q * c.b); }
inline Color operator+(const Color& a, const Color& o) { return Color(a.r + o.r, a.g + o.g, a.b + o.b); }

struct Spectrum {
    float r = 1, g = 1, b = 1;
    Spectrum() {}
    explicit Spectrum(float f) : r(f), g(f), b(f) {}
    Spectrum(float r_, float g_, float b_) : r(r_), g(g_), b(b_) {}
    explicit Spectrum(const Color& c) : r(c.r), g(c.g), b(c.b) {}
    Spectrum operator*(float q) const { return Spectrum(q * r, q * g, q * b); }
    Spectrum operator/(float q) const { return Spectrum(r / q, g / q, b / q); }
    Spectrum operator*(const Spectrum& o) const { return Spectrum(o.r * r, o.g * g, o.b * b); }
    Spectrum operator+(const Spectrum& o) const { return Spectrum(r + o.r, g + o.g, b + o.b); }
    float max() const { return std::max(std::max(r, g), b); }
};
struct Radiance {
    float r = 0, g = 0, b = 0;
    Radiance() {}
    Radiance(float r_, float g_, float b_) : r(r_), g(g_), b(b_) {}
    explicit Radiance(const Color& c) : r(c.r), g(c.g), b(c.b) {} // pow(c, 1.0) == c
    Radiance operator+(const Radiance& o) const { return Radiance(r + o.r, g + o.g, b + o.b); }
    Radiance& operator+=(const Radiance& o) { *this = *this + o; return *this; }
    void clamp(float v) { if (r > v) r = v; if (g > v) g = v; if (b > v) b = v; }
};
inline Radiance operator*(const Radiance& r, const Spectrum& s) { return Radiance(r.r * s.r, r.g * s.g, r.b * s.b); }
inline Radiance operator*(const Spectrum& s, const Radiance& r) { return Radiance(r.r * s.r, r.g * s.g, r.b * s.b); }

// ---------------------------------------------------------------- ray.hpp:6-29
struct Ray {
    vec3 origin, direction;
    float near = 0.0f, far = 10000.0f;
    Ray() {}
    Ray(vec3 from, vec3 dir) : origin(from) { direction = normalize(dir); }
    Ray(vec3 from, vec3 to, float eps) {
        origin = from;
        vec3 diff = to - from;
        direction = normalize(diff);
        float len = length(diff);
        near = 0.0f + eps;
        far = len - eps;
    }
    vec3 at(float t) const { return origin + t * direction; }
};

// ---------------------------------------------------------------- primitives.hpp:26-43
struct Light {
    enum Type { FULL_SPHERE, HEMISPHERE };
    Type type = FULL_SPHERE;
    vec3 pos;
    Radiance color;
    float intensity = 0.0f;
    float size = 0.0f;
    vec3 normal; // Q15: defined as 0 for point lights (reference leaves it uninitialised)
    bool valid = true; // Q15: "no light" => zero contribution
    float GetDirectionalFactor(vec3 v) const {
        if (type == FULL_SPHERE) return 1.0f;
        return std::max(0.0f, dot(v, normal));
    }
};

struct Texture {
    uint32_t kind = RGK_TEX_SOLID;
    uint32_t xsize = 0, ysize = 0;
    Color color;
    const float* data = nullptr; // 3 floats per texel, owned by Scene::texel_store
    const uint8_t* data8 = nullptr; // RGK_TEX_RGB8: 3 bytes per texel + byte -> float table
    const float* lut = nullptr;
    Color texel(int idx) const {
        if (kind == RGK_TEX_RGB8) return Color(lut[data8[3 * idx]], lut[data8[3 * idx + 1]], lut[data8[3 * idx + 2]]);
        return Color(data[3 * idx], data[3 * idx + 1], data[3 * idx + 2]);
    }
    // texture.cpp:35-77 (FileTexture) / texture.hpp:64-80 (SolidTexture)
    Color GetPixelInterpolated(vec2 pos) const {
        if (kind == RGK_TEX_SOLID) return color;
        float x = repeat(pos.x) * xsize - 0.5f;
        float y = repeat(pos.y) * ysize - 0.5f;
        float ix0f, iy0f;
        float fx = std::modf(x, &ix0f);
        float fy = std::modf(y, &iy0f);
        int ix0 = (int)ix0f;
        int iy0 = (int)iy0f;
        int ix1 = (ix0 != int(xsize) - 1) ? ix0 + 1 : ix0;
        int iy1 = (iy0 != int(ysize) - 1) ? iy0 + 1 : iy0;
        if (ix0 == -1) ix0 = 0;
        if (iy0 == -1) iy0 = 0;
        Color c00 = texel(iy0 * xsize + ix0);
        Color c01 = texel(iy0 * xsize + ix1);
        Color c10 = texel(iy1 * xsize + ix0);
        Color c11 = texel(iy1 * xsize + ix1);
        fy = 1.0f - fy;
        fx = 1.0f - fx;
        Color c0s = fx * c00 + (1.0f - fx) * c01;
        Color c1s = fx * c10 + (1.0f - fx) * c11;
        return fy * c0s + (1.0f - fy) * c1s;
    }
    // texture.cpp:79-90
    float GetSlopeRight(vec2 pos) const {
        if (kind == RGK_TEX_SOLID) return 0;
        int x = (int)(repeat(pos.x) * xsize - 0.5f);
        int y = (int)(repeat(pos.y) * ysize - 0.5f);
        int x2 = (x != int(xsize) - 1) ? x + 1 : x;
        if (x == -1) x = 0;
        if (y == -1) y = 0;
        Color here = texel(y * xsize + x), there = texel(y * xsize + x2);
        float a = (here.r + here.g + here.b) / 3;
        float b = (there.r + there.g + there.b) / 3;
        return a - b;
    }
    // texture.cpp:91-102
    float GetSlopeBottom(vec2 pos) const {
        if (kind == RGK_TEX_SOLID) return 0;
        int x = (int)(repeat(pos.x) * xsize - 0.5f);
        int y = (int)(repeat(pos.y) * ysize - 0.5f);
        int y2 = (y != int(ysize) - 1) ? y + 1 : y;
        if (x == -1) x = 0;
        if (y == -1) y = 0;
        Color here = texel(y * xsize + x), there = texel(y2 * xsize + x);
        float a = (here.r + here.g + here.b) / 3;
        float b = (there.r + there.g + there.b) / 3;
        return a - b;
    }
};

struct Material {
    uint32_t kind = RGK_BXDF_DIFFUSE;
    bool no_russian = false;
    Radiance emission;
    float roughness = 0, ior = 1, amt1 = 0;
    int diffuse = -1, color = -1, bump = -1, m1 = -1, m2 = -1;
};

struct Triangle {
    unsigned va, vb, vc;
    unsigned mat;
    float p[4]; // plane
};

// scene.hpp:212-253
struct CompressedKdNode {
    union { float split_plane; uint32_t triangles_start; };
    union { uint32_t other_child; uint32_t triangles_num; uint32_t kind; };
    bool IsLeaf() const { return (kind & 0x03) == 0x03; }
    int GetSplitAxis() const { return kind & 0x03; }
    float GetSplitPlane() const { return split_plane; }
    uint32_t GetTrianglesN() const { return triangles_num >> 2; }
    uint32_t GetFirstTrianglePos() const { return triangles_start; }
    uint32_t GetOtherChildIndex() const { return other_child >> 2; }
};

struct Intersection {
    int triangle = -1;
    float t = 0, a = 0, b = 0, c = 0;
    template <typename T> T Interpolate(const T& x, const T& y, const T& z) const { return a * x + b * y + c * z; }
};

struct Scene;

// scene.hpp:185-210, scene.cpp:431-574
struct UncompressedKdNode {
    const Scene* parent_scene = nullptr;
    enum { LEAF, INTERNAL } type = LEAF;
    unsigned depth = 0;
    std::pair<float, float> xBB, yBB, zBB;
    std::vector<unsigned> triangle_indices;
    UncompressedKdNode *ch0 = nullptr, *ch1 = nullptr;
    float prob0 = 0, prob1 = 0;
    int split_axis = 0;
    float split_pos = 0;
    void Subdivide(unsigned max_depth);
    void Free() {
        if (type == INTERNAL) { ch0->Free(); delete ch0; ch1->Free(); delete ch1; }
    }
    void Totals(unsigned& tris, unsigned& nodes, unsigned& maxdepth) const {
        if (type == LEAF) { tris += triangle_indices.size(); nodes += 1; maxdepth = std::max(maxdepth, depth); }
        else { nodes += 1; ch0->Totals(tris, nodes, maxdepth); ch1->Totals(tris, nodes, maxdepth); }
    }
};

constexpr float EMPTY_BONUS = 0.5f, ISECT_COST = 80.0f, TRAV_COST = 2.0f;

struct ArealLight {
    std::vector<std::pair<float, unsigned>> triangles_with_areas;
    float total_area = 0.0f;
    Radiance emission;
    float power = 0.0f;
};

struct TravStats { uint64_t nodes = 0, tris = 0; };

struct Scene {
    std::vector<vec3> vertices, normals, tangents;
    std::vector<vec2> texcoords;
    unsigned n_texcoords = 0;
    std::vector<Triangle> triangles;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    std::vector<std::vector<float>> texel_store;
    std::vector<std::vector<uint8_t>> texel8_store;
    std::vector<Light> pointlights;
    std::vector<std::pair<float, ArealLight>> areal_lights;
    float total_areal_power = 0, total_point_power = 0;
    std::vector<float> xevents, yevents, zevents;
    std::pair<float, float> xBB, yBB, zBB;
    float epsilon = 0.0001f;
    int skybox_mode = RGK_SKY_COLOR;
    Color skybox_color;
    float skybox_intensity = 1.0f, skybox_rotate = 0.0f;
    int skybox_texture = -1;
    std::vector<CompressedKdNode> compressed_array;
    std::vector<unsigned> compressed_triangles;
    unsigned kd_max_depth = 0;
    std::vector<float> ltc[2]; // [0]=Beckmann, [1]=GGX, 5 floats per entry

    void Commit();
    void CompressRec(const UncompressedKdNode* node, unsigned& array_pos, unsigned& triangle_pos);

    bool TestIntersection(const Triangle& tri, const Ray& r, float& t, float& a, float& b) const;
    Intersection FindIntersectKd(const Ray& r, TravStats* st = nullptr) const { return Find(r, -1, st); }
    Intersection FindIntersectKdOtherThan(const Ray& r, int ignore, TravStats* st = nullptr) const { return Find(r, ignore, st); }
    Intersection Find(const Ray& r, int ignore, TravStats* st) const;
    bool Visibility(vec3 a, vec3 b, TravStats* st = nullptr) const {
        Ray r(a, b, epsilon * 20.0f); // scene.cpp:670-673
        return FindIntersectKd(r, st).triangle < 0;
    }
    Light GetRandomLight(vec2 choice_sample, float light_sample, vec2 triangle_sample) const;
    Radiance GetSkyboxRay(vec3 direction) const;

    vec3 TriRandomPoint(const Triangle& t, vec2 sample) const;
    float TriArea(const Triangle& t) const;

    Color TexGet(int id, vec2 uv) const { return id < 0 ? Color(0, 0, 0) : textures[id].GetPixelInterpolated(uv); }
    Spectrum TexSpectrum(int id, vec2 uv) const { return Spectrum(TexGet(id, uv)); }
};

// primitives.cpp:24-36
static void CalculatePlane(const Scene& s, Triangle& t) {
    vec3 v0 = s.vertices[t.va], v1 = s.vertices[t.vb], v2 = s.vertices[t.vc];
    vec3 d0 = v1 - v0, d1 = v2 - v0;
    vec3 n = normalize(cross(d1, d0));
    float d = -dot(n, v0);
    t.p[0] = n.x; t.p[1] = n.y; t.p[2] = n.z; t.p[3] = d;
}
// primitives.cpp:38-45
float Scene::TriArea(const Triangle& t) const {
    vec3 a = vertices[t.va], b = vertices[t.vb], c = vertices[t.vc];
    vec3 q = a - b, r = c - b;
    return 0.5f * length(cross(q, r));
}
// primitives.cpp:61-73
vec3 Scene::TriRandomPoint(const Triangle& t, vec2 sample) const {
    vec2 r = sample;
    vec3 a = vertices[t.va], c = vertices[t.vb], b = vertices[t.vc];
    vec3 Va = a - c, Vb = b - c;
    if (r.x + r.y > 1.0f) { r.x = 1.0f - r.x; r.y = 1.0f - r.y; }
    return c + r.x * Va + r.y * Vb;
}

// primitives.cpp:75-166 (Badouel).  The two plane dot products are double.
bool Scene::TestIntersection(const Triangle& tri, const Ray& r, float& t, float& a, float& b) const {
    const float eps = epsilon;
    vec3 planeN(tri.p[0], tri.p[1], tri.p[2]);
    double dotv = dot(r.direction, planeN); // float dot, widened
    if (std::isnan(dotv)) return false;
    if (dotv < eps && dotv > -eps) return false;
    double dot2 = dot(r.origin, planeN);
    t = (float)(-((double)tri.p[3] + dot2) / dotv);
    int i1, i2;
    vec3 pq = vabs(planeN);
    if (pq.x > pq.y && pq.x > pq.z) { i1 = 1; i2 = 2; }
    else if (pq.y > pq.z) { i1 = 0; i2 = 2; }
    else { i1 = 0; i2 = 1; }
    vec3 vert0 = vertices[tri.va], vert1 = vertices[tri.vb], vert2 = vertices[tri.vc];
    vec2 point(r.origin[i1] + r.direction[i1] * t, r.origin[i2] + r.direction[i2] * t);
    vec2 q0(point.x - vert0[i1], point.y - vert0[i2]);
    vec2 q1(vert1[i1] - vert0[i1], vert1[i2] - vert0[i2]);
    vec2 q2(vert2[i1] - vert0[i1], vert2[i2] - vert0[i2]);
    float alpha, beta;
    if (q1.x > -eps && q1.x < eps) { // uncommon case
        beta = q0.x / q2.x;
        if (beta < 0 || beta > 1) return false;
        alpha = (q0.y - beta * q2.y) / q1.y;
    } else {
        beta = (q0.y * q1.x - q0.x * q1.y) / (q2.y * q1.x - q2.x * q1.y);
        if (beta < 0 || beta > 1) return false;
        alpha = (q0.x - beta * q2.x) / q1.x;
    }
    if (alpha < 0 || (alpha + beta) > 1.0) return false;
    a = alpha;
    b = beta;
    return true;
}

// scene_intersect.cpp:4-116 (ignore<0) and :211-327 (ignore>=0): identical but for the skip.
Intersection Scene::Find(const Ray& r, int ignore, TravStats* st) const {
    Intersection res;
    res.triangle = -1;
    res.t = std::numeric_limits<float>::infinity();
    const std::pair<float, float>* bb[3] = {&xBB, &yBB, &zBB};
    float t0 = r.near, t1 = r.far;
    for (int i = 0; i < 3; ++i) {
        float invRayDir = 1.f / r.direction[i];
        float tNear = (bb[i]->first - r.origin[i]) * invRayDir;
        float tFar = (bb[i]->second - r.origin[i]) * invRayDir;
        if (tNear > tFar) std::swap(tNear, tFar);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return res;
    }
    struct NodeToDo { const CompressedKdNode* node; float tmin, tmax; };
    vec3 invDir(1.f / r.direction.x, 1.f / r.direction.y, 1.f / r.direction.z);
    NodeToDo todo[200];
    int todo_size = 1;
    const CompressedKdNode* base = compressed_array.data();
    todo[0] = NodeToDo{base, t0, t1};
    while (todo_size > 0) {
        todo_size--;
        const CompressedKdNode* node = todo[todo_size].node;
        float tmin = todo[todo_size].tmin, tmax = todo[todo_size].tmax;
        if (r.far < tmin) break;
        if (st) st->nodes++;
        if (node->IsLeaf()) {
            bool hit = false;
            unsigned n = node->GetTrianglesN();
            uint32_t tri_start = node->GetFirstTrianglePos();
            for (unsigned p = 0; p < n; p++) {
                unsigned i = compressed_triangles[tri_start + p];
                const Triangle& tri = triangles[i];
                float t, a, b;
                if ((int)i == ignore) continue;
                if (st) st->tris++;
                if (TestIntersection(tri, r, t, a, b)) {
                    if (t < tmin - epsilon || t > tmax + epsilon) continue;
                    if (t < res.t) {
                        res.triangle = (int)i;
                        res.t = t;
                        float c = 1.0f - a - b;
                        res.a = c; res.b = a; res.c = b;
                        hit = true;
                    }
                }
            }
            if (hit) return res;
        } else {
            int axis = node->GetSplitAxis();
            float tplane = (node->GetSplitPlane() - r.origin[axis]) * invDir[axis];
            const CompressedKdNode *firstChild, *secondChild;
            int belowFirst = (r.origin[axis] < node->GetSplitPlane()) ||
                             (r.origin[axis] == node->GetSplitPlane() && r.direction[axis] <= 0);
            if (belowFirst) { firstChild = node + 1; secondChild = base + node->GetOtherChildIndex(); }
            else { firstChild = base + node->GetOtherChildIndex(); secondChild = node + 1; }
            if (tplane > tmax || tplane <= 0) todo[todo_size++] = NodeToDo{firstChild, tmin, tmax};
            else if (tplane < tmin) todo[todo_size++] = NodeToDo{secondChild, tmin, tmax};
            else {
                todo[todo_size++] = NodeToDo{secondChild, tplane, tmax};
                todo[todo_size++] = NodeToDo{firstChild, tmin, tplane};
            }
        }
    }
    return res;
}

// scene.cpp:431-574
void UncompressedKdNode::Subdivide(unsigned max_depth) {
    if (depth >= max_depth) return;
    unsigned n = triangle_indices.size();
    if (n < 2) return;
    float sizes[3] = {xBB.second - xBB.first, yBB.second - yBB.first, zBB.second - zBB.first};
    unsigned axis = std::max_element(sizes, sizes + 3) - sizes;
    const std::vector<float>* evch[3] = {&parent_scene->xevents, &parent_scene->yevents, &parent_scene->zevents};
    unsigned retries = 0;
    struct BBEvent { float pos; int triangleID; int type; }; // type: 0 BEGIN, 1 END
    std::vector<BBEvent> events;
    int best_offset;
    float best_pos;
    for (;;) { // "retry:" loop
        const std::vector<float>& all_events = *evch[axis];
        events.assign(2 * n, BBEvent{0, 0, 0});
        for (unsigned i = 0; i < n; i++) {
            int t = triangle_indices[i];
            events[2 * i + 0] = BBEvent{all_events[2 * t + 0], t, 0};
            events[2 * i + 1] = BBEvent{all_events[2 * t + 1], t, 1};
        }
        std::sort(events.begin(), events.end(), [](const BBEvent& a, const BBEvent& b) {
            if (a.pos == b.pos) return a.type < b.type;
            return a.pos < b.pos;
        });
        const std::pair<float, float>* axbds[3] = {&xBB, &yBB, &zBB};
        const std::pair<float, float>& axis_bounds = *axbds[axis];
        const float BBsize[3] = {xBB.second - xBB.first, yBB.second - yBB.first, zBB.second - zBB.first};
        best_offset = -1;
        float best_cost = std::numeric_limits<float>::infinity();
        best_pos = std::numeric_limits<float>::infinity();
        float nosplit_cost = ISECT_COST * n;
        unsigned axis2 = (axis + 1) % 3, axis3 = (axis + 2) % 3;
        float invTotalSA = 1.f / (2.f * (BBsize[0] * BBsize[1] + BBsize[0] * BBsize[2] + BBsize[1] * BBsize[2]));
        int n_before = 0, n_after = n;
        for (unsigned i = 0; i < 2 * n; i++) {
            if (events[i].type == 1) n_after--;
            float pos = events[i].pos;
            if (pos > axis_bounds.first && pos < axis_bounds.second) {
                float below_surface_area = 2 * (BBsize[axis2] * BBsize[axis3] +
                                                (pos - axis_bounds.first) * BBsize[axis2] +
                                                (pos - axis_bounds.first) * BBsize[axis3]);
                float above_surface_area = 2 * (BBsize[axis2] * BBsize[axis3] +
                                                (axis_bounds.second - pos) * BBsize[axis2] +
                                                (axis_bounds.second - pos) * BBsize[axis3]);
                float p_before = below_surface_area * invTotalSA;
                float p_after = above_surface_area * invTotalSA;
                float bonus = (n_before == 0 || n_after == 0) ? EMPTY_BONUS : 0.f;
                float cost = TRAV_COST + ISECT_COST * (1.f - bonus) * (p_before * n_before + p_after * n_after);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_offset = i;
                    best_pos = pos;
                    prob0 = p_before;
                    prob1 = p_after;
                }
            }
            if (events[i].type == 0) n_before++;
        }
        if (best_offset == -1 || best_cost > nosplit_cost) {
            if (retries < 2) { retries++; axis = (axis + 1) % 3; continue; }
            return;
        }
        break;
    }
    type = INTERNAL;
    ch0 = new UncompressedKdNode();
    ch1 = new UncompressedKdNode();
    ch0->parent_scene = parent_scene; ch1->parent_scene = parent_scene;
    ch0->depth = depth + 1; ch1->depth = depth + 1;
    split_axis = axis;
    split_pos = best_pos;
    for (unsigned i = 0; i < (unsigned)best_offset; ++i)
        if (events[i].type == 0) ch0->triangle_indices.push_back(events[i].triangleID);
    for (unsigned i = best_offset + 1; i < 2 * n; ++i)
        if (events[i].type == 1) ch1->triangle_indices.push_back(events[i].triangleID);
    std::vector<BBEvent>().swap(events);
    ch0->xBB = (axis == 0) ? std::make_pair(xBB.first, best_pos) : xBB;
    ch0->yBB = (axis == 1) ? std::make_pair(yBB.first, best_pos) : yBB;
    ch0->zBB = (axis == 2) ? std::make_pair(zBB.first, best_pos) : zBB;
    ch1->xBB = (axis == 0) ? std::make_pair(best_pos, xBB.second) : xBB;
    ch1->yBB = (axis == 1) ? std::make_pair(best_pos, yBB.second) : yBB;
    ch1->zBB = (axis == 2) ? std::make_pair(best_pos, zBB.second) : zBB;
    ch0->Subdivide(max_depth);
    ch1->Subdivide(max_depth);
}

// scene.cpp:637-657
void Scene::CompressRec(const UncompressedKdNode* node, unsigned& array_pos, unsigned& triangle_pos) {
    if (node->type == UncompressedKdNode::LEAF) {
        CompressedKdNode c;
        c.triangles_num = ((uint32_t)node->triangle_indices.size() << 2) | 0x03;
        c.triangles_start = triangle_pos;
        compressed_array[array_pos++] = c;
        for (unsigned t : node->triangle_indices) compressed_triangles[triangle_pos++] = t;
    } else {
        unsigned my_pos = array_pos;
        CompressedKdNode c;
        c.kind = node->split_axis;
        c.split_plane = node->split_pos;
        compressed_array[array_pos++] = c;
        CompressRec(node->ch0, array_pos, triangle_pos);
        compressed_array[my_pos].other_child = (compressed_array[my_pos].other_child & 0x03) | (array_pos << 2);
        CompressRec(node->ch1, array_pos, triangle_pos);
    }
}

// scene.cpp:294-429
void Scene::Commit() {
    for (auto& t : triangles) CalculatePlane(*this, t);
    total_areal_power = 0.0f;
    for (auto& q : areal_lights) {
        ArealLight& al = q.second;
        for (auto& p : al.triangles_with_areas) {
            float area = TriArea(triangles[p.second]);
            p.first = area;
            al.total_area += area;
        }
        al.emission = materials[triangles[al.triangles_with_areas[0].second].mat].emission;
        std::sort(al.triangles_with_areas.rbegin(), al.triangles_with_areas.rend());
        float p = al.total_area * (al.emission.r + al.emission.g + al.emission.b);
        al.power = p;
        q.first = p;
        total_areal_power += p;
    }
    total_point_power = 0.0f;
    for (auto& l : pointlights) total_point_power += l.intensity * 4.0f * PI_F;

    unsigned n_triangles = triangles.size();
    xevents.resize(2 * n_triangles); yevents.resize(2 * n_triangles); zevents.resize(2 * n_triangles);
    auto fill = [&](int axis, std::vector<float>& buf) {
        for (unsigned i = 0; i < n_triangles; i++) {
            const Triangle& t = triangles[i];
            auto p = std::minmax({vertices[t.va][axis], vertices[t.vb][axis], vertices[t.vc][axis]});
            buf[2 * i + 0] = p.first;
            buf[2 * i + 1] = p.second;
        }
    };
    fill(0, xevents); fill(1, yevents); fill(2, zevents);
    auto p = std::minmax_element(xevents.begin(), xevents.end());
    auto q = std::minmax_element(yevents.begin(), yevents.end());
    auto r = std::minmax_element(zevents.begin(), zevents.end());
    float xsize = *p.second - *p.first, ysize = *q.second - *q.first, zsize = *r.second - *r.first;
    float diameter = std::sqrt(xsize * xsize + ysize * ysize + zsize * zsize);
    epsilon = 0.00001f * diameter;
    xBB = std::make_pair(*p.first - epsilon, *p.second + epsilon);
    yBB = std::make_pair(*q.first - epsilon, *q.second + epsilon);
    zBB = std::make_pair(*r.first - epsilon, *r.second + epsilon);

    UncompressedKdNode* root = new UncompressedKdNode;
    root->parent_scene = this;
    for (unsigned i = 0; i < n_triangles; i++) root->triangle_indices.push_back(i);
    root->xBB = xBB; root->yBB = yBB; root->zBB = zBB;
    int l = std::log2(n_triangles) + 8;
    root->Subdivide(l);
    unsigned tris = 0, nodes = 0, maxd = 0;
    root->Totals(tris, nodes, maxd);
    kd_max_depth = maxd;
    compressed_array.resize(nodes);
    compressed_triangles.resize(tris);
    unsigned ap = 0, tp = 0;
    CompressRec(root, ap, tp);
    root->Free();
    delete root;
    std::vector<float>().swap(xevents); std::vector<float>().swap(yevents); std::vector<float>().swap(zevents);
}

// scene.cpp:686-745
Light Scene::GetRandomLight(vec2 choice_sample, float light_sample, vec2 triangle_sample) const {
    Light none; none.valid = false; none.type = Light::FULL_SPHERE; // Q15
    float total_power = total_point_power + total_areal_power;
    if (total_power <= 0.0f) return none;
    float q = choice_sample.x * total_power;
    if (q < total_point_power) {
        for (unsigned i = 0; i < pointlights.size(); i++) {
            q -= pointlights[i].intensity * 4.0f * PI_F;
            if (q <= 0.0f) return pointlights[i];
        }
        return none;
    } else {
        q = choice_sample.y * total_areal_power;
        for (unsigned i = 0; i < areal_lights.size(); i++) {
            q -= areal_lights[i].first;
            if (q <= 0.0f) {
                const ArealLight& al = areal_lights[i].second;
                float p = light_sample * al.total_area;
                for (unsigned j = 0; j < al.triangles_with_areas.size(); j++) {
                    p -= al.triangles_with_areas[j].first;
                    if (p <= 0.0f) {
                        const Triangle& t = triangles[al.triangles_with_areas[j].second];
                        Light res;
                        res.type = Light::HEMISPHERE;
                        res.pos = TriRandomPoint(t, triangle_sample);
                        res.color = al.emission;
                        res.intensity = 1.0f;
                        res.normal = normals[t.va];
                        return res;
                    }
                }
                return none;
            }
        }
        return none;
    }
}

// scene.cpp:748-763
Radiance Scene::GetSkyboxRay(vec3 direction) const {
    if (skybox_mode == RGK_SKY_COLOR) return Radiance(skybox_color) * Spectrum(skybox_intensity);
    float alpha = rgk_asinf(direction.y);             // std::asin / std::atan2 in the reference: pinned, include/rgk_libm.h
    float beta = -rgk_atan2f(direction.x, direction.z);
    beta += skybox_rotate * 0.0174533f;
    float x = beta / (2.0f * PI_F) + 0.5f;
    float y = alpha / PI_F + 0.5f;
    Color c = TexGet(skybox_texture, vec2(x, y));
    return Radiance(c) * Spectrum(skybox_intensity);
}

// ---------------------------------------------------------------- glm.cpp:3-59, glm.hpp:18-35
static quat RotationBetweenVectors(vec3 start, vec3 dest) {
    start = normalize(start);
    dest = normalize(dest);
    float cosTheta = dot(start, dest);
    vec3 rotationAxis;
    if (cosTheta < -1 + 0.001f) {
        rotationAxis = cross(vec3(0.0f, 1.0f, 0.0f), start);
        if (length(rotationAxis) < 0.01) rotationAxis = cross(vec3(1.0f, 0.0f, 0.0f), start);
        rotationAxis = normalize(rotationAxis);
        return angleAxis(PI_F, rotationAxis);
    }
    rotationAxis = cross(start, dest);
    float s = std::sqrt((1 + cosTheta) * 2);
    float invs = 1 / s;
    return quat(s * 0.5f, rotationAxis.x * invs, rotationAxis.y * invs, rotationAxis.z * invs);
}
static quat RotationFromY(vec3 dest) {
    dest = normalize(dest);
    float cosTheta = dest.y;
    vec3 rotationAxis;
    if (cosTheta < -1 + 0.00001f) {
        rotationAxis = vec3(1.0, 0.0, 0.0);
        return angleAxis(PI_F, rotationAxis);
    }
    rotationAxis = cross(vec3(0.0, 1.0, 0.0), dest);
    float s = std::sqrt((1 + cosTheta) * 2);
    float invs = 1 / s;
    return quat(s * 0.5f, rotationAxis.x * invs, rotationAxis.y * invs, rotationAxis.z * invs);
}
struct SystemTransform {
    quat global_to_local, local_to_global;
    SystemTransform() {}
    SystemTransform(vec3 global, vec3 local)
        : global_to_local(RotationBetweenVectors(global, local)), local_to_global(inverse(global_to_local)) {}
    vec3 toGlobal(vec3 local) const { return local_to_global * local; }
    vec3 toLocal(vec3 global) const { return global_to_local * global; }
};

// ---------------------------------------------------------------- random_utils.hpp:12-73
namespace RandomUtils {
static vec2 Sample2DToDiscUniform(vec2 sample) {
    float r = std::sqrt(sample.x);
    float a = (float)(sample.y * 2.0f * M_PI);
    return vec2(r * rgk_sinf(a), r * rgk_cosf(a)); // std::sin / std::cos in the reference: pinned, include/rgk_libm.h
}
static vec3 Sample2DToHemisphereCosine(vec2 sample) {
    vec2 p = Sample2DToDiscUniform(sample);
    float y = std::sqrt(std::max(0.00001f, 1 - p.x * p.x - p.y * p.y));
    return vec3(p.x, y, p.y);
}
static vec3 Sample2DToHemisphereCosineZ(vec2 sample) {
    vec2 p = Sample2DToDiscUniform(sample);
    float z = std::sqrt(std::max(0.00001f, 1 - p.x * p.x - p.y * p.y));
    return vec3(p.x, p.y, z);
}
static vec3 Sample2DToHemisphereCosineDirected(vec2 sample, vec3 direction) {
    return RotationFromY(direction) * Sample2DToHemisphereCosine(sample);
}
static vec3 Sample2DToSphereUniform(vec2 sample) {
    float z = sample.x * 2.0f - 1.0f;
    float a = (float)(sample.y * 6.283185);
    float r = std::sqrt(1 - z * z);
    float x = r * rgk_cosf(a);
    float y = r * rgk_sinf(a);
    return vec3(x, y, z);
}
static bool DecideAndRescale(float& sample, float probability) {
    if (probability == 0.0f) return false;
    if (probability == 1.0f) return true;
    if (sample < probability) { sample /= probability; return true; }
    sample = (sample - probability) / (1.0f - probability);
    return false;
}
} // namespace RandomUtils

// ---------------------------------------------------------------- LTC/ltc.cpp:20-143
struct LTCdef { const float* tab; int size; }; // 5 floats per entry: m0,m2,m4,m6,amp
static mat3 ltc_get_n(LTCdef ltc, int theta, int alpha) {
    const float* e = ltc.tab + 5 * (alpha + theta * ltc.size);
    // glm::mat3(m0..m8) column-major: col0=(m0,m1,m2) col1=(m3,m4,m5) col2=(m6,m7,m8)
    return mat3(vec3(e[0], 0.0f, e[1]), vec3(0.0f, e[2], 0.0f), vec3(e[3], 0.0f, 1.0f));
}
static float ltc_amp_n(LTCdef ltc, int theta, int alpha) { return ltc.tab[5 * (alpha + theta * ltc.size) + 4]; }

static std::pair<mat3, float> ltc_get_bilinear(LTCdef ltc, const float theta, const float alpha) {
    float t = std::max(0.0f, std::min(1.0f, theta / (0.5f * 3.14159f)));
    float a = std::max(0.0f, std::min(1.0f, sqrtf(alpha)));
    if (t >= 1.0f) t = 0.999f;
    if (a >= 1.0f) a = 0.999f;
    int s = ltc.size - 1;
    int t1 = floorf(t * s);
    int t2 = t1 + 1;
    int a1 = floorf(a * s);
    int a2 = a1 + 1;
    mat3 Mt1a1 = ltc_get_n(ltc, t1, a1), Mt1a2 = ltc_get_n(ltc, t1, a2);
    mat3 Mt2a1 = ltc_get_n(ltc, t2, a1), Mt2a2 = ltc_get_n(ltc, t2, a2);
    float At1a1 = ltc_amp_n(ltc, t1, a1), At1a2 = ltc_amp_n(ltc, t1, a2);
    float At2a1 = ltc_amp_n(ltc, t2, a1), At2a2 = ltc_amp_n(ltc, t2, a2);
    float dt1 = t * s - t1, dt2 = t2 - t * s, da1 = a * s - a1, da2 = a2 - a * s;
    mat3 resM = Mt1a1 * dt2 * da2 + Mt1a2 * dt2 * da1 + Mt2a1 * dt1 * da2 + Mt2a2 * dt1 * da1;
    float resAMP = At1a1 * dt2 * da2 + At1a2 * dt2 * da1 + At2a1 * dt1 * da2 + At2a2 * dt1 * da1;
    return {resM, resAMP};
}
// LTC::GetPDF(ltc, N, Vr, Vi, alpha) ltc.cpp:59-87
static float ltc_GetPDF(LTCdef ltc, vec3 N, vec3 Vr, vec3 Vi, float alpha) {
    vec3 tangent = cross(N, Vi);
    vec3 Vi_cast = cross(tangent, N);
    mat3 rotate(Vi_cast, tangent, N);
    mat3 unrotate = inverse(rotate);
    vec3 Vr3 = unrotate * Vr;
    float theta = angle(Vi, N);
    auto q = ltc_get_bilinear(ltc, theta, alpha);
    mat3 M = q.first;
    float amplitude = q.second;
    mat3 invM = inverse(M);
    vec3 p = normalize(invM * Vr3);
    vec3 Loriginal = p;
    vec3 L_ = M * Loriginal;
    float l = length(L_);
    float detM = determinant(M);
    float Jacobian = detM / (l * l * l);
    float D = 1.0f / 3.14159f * std::max(0.0f, Loriginal.z);
    return amplitude * D / Jacobian;
}
// LTC::GetRandom ltc.cpp:113-143
static vec3 ltc_GetRandom(LTCdef ltc, vec3 N, vec3 Vi, float roughness, vec3 rand_hscos) {
    vec3 tangent = cross(N, Vi);
    vec3 Vi_cast = cross(tangent, N);
    mat3 rotate(Vi_cast, tangent, N);
    float theta = angle(Vi, N);
    auto q = ltc_get_bilinear(ltc, std::max(theta, PI_F / 4.0f), roughness);
    mat3 M = q.first;
    vec3 s = M * rand_hscos;
    if (s.z < 0.0001f) s.z = 0.0001f;
    s = rotate * s;
    return normalize(s);
}

// ---------------------------------------------------------------- bxdf/bxdf.cpp:192-423, bxdf.hpp:107-159
const vec3 BxDFUpVector(0.0f, 0.0f, 1.0f);

static std::pair<float, float> FresnellDielectric(float eta, float cosTheta) {
    if (cosTheta < 0.0f) { eta = 1.0f / eta; cosTheta = -cosTheta; }
    float sinThetaTSq = eta * eta * (1.0f - cosTheta * cosTheta);
    if (sinThetaTSq > 1.0f) return {1.0f, 0.0f};
    float cosThetaTrans = std::sqrt(std::max(1.0f - sinThetaTSq, 0.0f));
    float Rs = (eta * cosTheta - cosThetaTrans) / (eta * cosTheta + cosThetaTrans);
    float Rp = (eta * cosThetaTrans - cosTheta) / (eta * cosThetaTrans + cosTheta);
    float R = 0.5f * (Rs * Rs + Rp * Rp);
    return {R, cosThetaTrans};
}

static Spectrum bxdf_value(const Scene& sc, const Material& m, vec3 Vi, vec3 Vr, vec2 uv) {
    switch (m.kind) {
    case RGK_BXDF_DIFFUSE:
        if (Vi.z <= 0 || Vr.z <= 0) return Spectrum(0);
        return sc.TexSpectrum(m.diffuse, uv) / PI_F;
    case RGK_BXDF_MIRROR: {
        vec3 reflected(-Vi.x, -Vi.y, Vi.z);
        if (std::fabs(dot(reflected, Vr) - 1) < 0.0001f) return sc.TexSpectrum(m.color, uv);
        return Spectrum(0.0f);
    }
    case RGK_BXDF_DIELECTRIC: {
        float eta = (Vi.z < 0) ? m.ior : (float)(1.0 / m.ior);
        auto fr = FresnellDielectric(eta, Vi.z); // Q7: signed Vi.z here
        float reflectionP = fr.first, cosTheta = fr.second;
        Spectrum c = sc.TexSpectrum(m.color, uv);
        if (Vi.z * Vr.z > 0) {
            vec3 reflected(-Vi.x, -Vi.y, Vi.z);
            if (std::fabs(dot(Vr, reflected) - 1) < 0.001f) return Spectrum(reflectionP) * c;
            return Spectrum(0.0f);
        } else {
            vec3 refracted(-Vi.x * eta, -Vi.y * eta, (Vi.z > 0) ? -cosTheta : cosTheta);
            if (std::fabs(dot(Vr, refracted) - 1) < 0.001f) return Spectrum(1.0f - reflectionP) * c;
            return Spectrum(0.0f);
        }
    }
    case RGK_BXDF_TRANSPARENT: {
        vec3 inv(-Vi.x, -Vi.y, -Vi.z);
        if (std::fabs(dot(inv, Vr) - 1) < 0.0001f) return Spectrum(1.0f);
        return Spectrum(0.0f);
    }
    case RGK_BXDF_MIX: {
        Spectrum s1 = bxdf_value(sc, sc.materials[m.m1], Vi, Vr, uv);
        Spectrum s2 = bxdf_value(sc, sc.materials[m.m2], Vi, Vr, uv);
        return s1 * m.amt1 + s2 * (1.0f - m.amt1);
    }
    case RGK_BXDF_LTC_BECKMANN:
    case RGK_BXDF_LTC_GGX: {
        if (Vi.z <= 0 || Vr.z <= 0) return Spectrum(0);
        LTCdef ltc{sc.ltc[m.kind == RGK_BXDF_LTC_GGX].data(), 64};
        Spectrum spec = sc.TexSpectrum(m.color, uv);
        // Q6 (defined): a black lobe is not evaluated.  The reference forms 0 * pdf, which is 0 -- and the same bits as
        // here -- unless the pdf is NaN (view exactly along N: singular frame, ltc.cpp:62-69), where it poisons the path.
        if (spec.r == 0.0f && spec.g == 0.0f && spec.b == 0.0f) return Spectrum(0);
        return spec * ltc_GetPDF(ltc, BxDFUpVector, Vi, Vr, m.roughness);
    }
    case RGK_BXDF_LTC_BECKMANN_DIFFUSE:
    case RGK_BXDF_LTC_GGX_DIFFUSE: {
        if (Vi.z <= 0 || Vr.z <= 0) return Spectrum(0);
        LTCdef ltc{sc.ltc[m.kind == RGK_BXDF_LTC_GGX_DIFFUSE].data(), 64};
        Spectrum diff = sc.TexSpectrum(m.diffuse, uv);
        Spectrum spec = sc.TexSpectrum(m.color, uv);
        if (spec.r == 0.0f && spec.g == 0.0f && spec.b == 0.0f) return diff / PI_F; // Q6, as above (18 of Sponza's 20 materials have Ks = 0)
        return spec * ltc_GetPDF(ltc, BxDFUpVector, Vi, Vr, m.roughness) + diff / PI_F;
    }
    }
    return Spectrum(0);
}

static std::tuple<vec3, Spectrum, bool> bxdf_sample(const Scene& sc, const Material& m, vec3 Vi, vec2 uv, vec2 sample) {
    switch (m.kind) {
    case RGK_BXDF_DIFFUSE: {
        if (Vi.z <= 0) return std::make_tuple(vec3(0, 1, 0), Spectrum(0), false);
        vec3 v = RandomUtils::Sample2DToHemisphereCosineZ(sample);
        return std::make_tuple(v, sc.TexSpectrum(m.diffuse, uv), false);
    }
    case RGK_BXDF_MIRROR: {
        vec3 reflected(-Vi.x, -Vi.y, Vi.z);
        return std::make_tuple(reflected, sc.TexSpectrum(m.color, uv), false);
    }
    case RGK_BXDF_DIELECTRIC: {
        float eta = (Vi.z < 0) ? m.ior : (float)(1.0 / m.ior);
        auto fr = FresnellDielectric(eta, std::fabs(Vi.z));
        float reflectionP = fr.first, cosTheta = fr.second;
        Spectrum c = sc.TexSpectrum(m.color, uv);
        if (RandomUtils::DecideAndRescale(sample.x, reflectionP)) {
            vec3 reflected(-Vi.x, -Vi.y, Vi.z);
            return std::make_tuple(reflected, c, false);
        } else {
            cosTheta = std::fabs(cosTheta);
            vec3 refracted(-Vi.x * eta, -Vi.y * eta, (Vi.z > 0) ? -cosTheta : cosTheta);
            return std::make_tuple(refracted, c, true);
        }
    }
    case RGK_BXDF_TRANSPARENT:
        return std::make_tuple(vec3(-Vi.x, -Vi.y, -Vi.z), Spectrum(1.0f), true);
    case RGK_BXDF_MIX:
        if (RandomUtils::DecideAndRescale(sample.x, m.amt1)) return bxdf_sample(sc, sc.materials[m.m1], Vi, uv, sample);
        return bxdf_sample(sc, sc.materials[m.m2], Vi, uv, sample);
    case RGK_BXDF_LTC_BECKMANN:
    case RGK_BXDF_LTC_GGX: {
        LTCdef ltc{sc.ltc[m.kind == RGK_BXDF_LTC_GGX].data(), 64};
        vec3 v = RandomUtils::Sample2DToHemisphereCosineZ(sample);
        v = ltc_GetRandom(ltc, BxDFUpVector, Vi, m.roughness, v);
        if (v.z <= 0) return std::make_tuple(v, Spectrum(0), false);
        return std::make_tuple(v, sc.TexSpectrum(m.color, uv), false);
    }
    case RGK_BXDF_LTC_BECKMANN_DIFFUSE:
    case RGK_BXDF_LTC_GGX_DIFFUSE: {
        LTCdef ltc{sc.ltc[m.kind == RGK_BXDF_LTC_GGX_DIFFUSE].data(), 64};
        Color diff = sc.TexGet(m.diffuse, uv);
        Color spec = sc.TexGet(m.color, uv);
        float diffuse_power = diff.r + diff.g + diff.b;
        float specular_power = spec.r + spec.g + spec.b;
        float diffuse_probability = diffuse_power / (diffuse_power + specular_power + 0.0001f);
        if (RandomUtils::DecideAndRescale(sample.x, diffuse_probability)) {
            if (Vi.z <= 0) return std::make_tuple(vec3(0, 1, 0), Spectrum(0), false);
            vec3 v = RandomUtils::Sample2DToHemisphereCosineZ(sample);
            return std::make_tuple(v, sc.TexSpectrum(m.diffuse, uv), false);
        } else {
            vec3 v = RandomUtils::Sample2DToHemisphereCosineZ(sample);
            v = ltc_GetRandom(ltc, BxDFUpVector, Vi, m.roughness, v);
            if (v.z <= 0) return std::make_tuple(v, Spectrum(0), false);
            return std::make_tuple(v, sc.TexSpectrum(m.color, uv), false);
        }
    }
    }
    return std::make_tuple(vec3(0, 1, 0), Spectrum(0), false);
}

// ---------------------------------------------------------------- camera.cpp:7-83
struct Camera {
    vec3 origin, lookat, direction, cameraup, cameraleft, viewscreen, viewscreen_x, viewscreen_y;
    float lens_size;
    int xsize, ysize;
    Camera() {}
    // Camera::Camera, src/camera.cpp:7-24
    Camera(vec3 pos, vec3 la, vec3 up, float yview, float xview, int xres, int yres, float focus_plane, float ls) {
        origin = pos;
        lookat = la;
        cameraup = up;
        xsize = xres; ysize = yres;
        lens_size = ls;
        direction = normalize(lookat - origin);
        cameraleft = normalize(cross(cameraup, direction));
        cameraup = normalize(cross(cameraleft, direction));
        viewscreen_x = -xview * cameraleft * focus_plane;
        viewscreen_y = yview * cameraup * focus_plane;
        viewscreen = origin + direction * focus_plane - 0.5f * viewscreen_y - 0.5f * viewscreen_x;
    }
    // the members as they cross the seam (include/rgk.h rgk_camera = src/camera.hpp:27-41)
    Camera(const rgk_camera& c) {
        origin = vec3(c.origin[0], c.origin[1], c.origin[2]);
        lookat = origin;
        direction = vec3(c.direction[0], c.direction[1], c.direction[2]);
        cameraup = vec3(c.cameraup[0], c.cameraup[1], c.cameraup[2]);
        cameraleft = vec3(c.cameraleft[0], c.cameraleft[1], c.cameraleft[2]);
        viewscreen = vec3(c.viewscreen[0], c.viewscreen[1], c.viewscreen[2]);
        viewscreen_x = vec3(c.viewscreen_x[0], c.viewscreen_x[1], c.viewscreen_x[2]);
        viewscreen_y = vec3(c.viewscreen_y[0], c.viewscreen_y[1], c.viewscreen_y[2]);
        xsize = c.xsize; ysize = c.ysize;
        lens_size = c.lens_size;
    }
    void store(rgk_camera& c) const {
        const vec3* src[7] = {&origin, &direction, &cameraup, &cameraleft, &viewscreen, &viewscreen_x, &viewscreen_y};
        float* dst[7] = {c.origin, c.direction, c.cameraup, c.cameraleft, c.viewscreen, c.viewscreen_x, c.viewscreen_y};
        for (int i = 0; i < 7; i++) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
        c.lens_size = lens_size; c.xsize = xsize; c.ysize = ysize;
    }
    bool IsSimple() const { return lens_size == 0.0f; }
    vec3 GetViewScreenPoint(float x, float y) const {
        vec3 xo = x * viewscreen_x;
        vec3 yo = y * viewscreen_y;
        return viewscreen + xo + yo;
    }
    Ray GetPixelRay(int x, int y, int xres, int yres, vec2 off) const {
        vec3 p = GetViewScreenPoint((x + off.x) / (float)(xres), (y + off.y) / (float)(yres));
        return Ray(origin, p - origin);
    }
    Ray GetPixelRayLens(int x, int y, int xres, int yres, vec2 off, vec2 lenssample) const {
        vec3 p = GetViewScreenPoint((x + off.x) / (float)(xres), (y + off.y) / (float)(yres));
        vec2 lenso = RandomUtils::Sample2DToDiscUniform(lenssample) * lens_size;
        vec3 o = origin + lenso.x * cameraleft + lenso.y * cameraup;
        return Ray(o, p - o);
    }
    bool GetCoordsFromDirection(vec3 dir, int& x, int& y) const {
        vec3 N = direction;
        float q = dot(dir, N);
        if (q < 0.0001) return false;
        float t = dot(viewscreen - origin, N) / q;
        if (t <= 0) return false;
        vec3 p = origin + dir * t;
        vec3 vp = p - viewscreen;
        float plen = length(vp);
        float v1_cast_len = plen * (dot(normalize(vp), normalize(viewscreen_x)));
        float v2_cast_len = plen * (dot(normalize(vp), normalize(viewscreen_y)));
        float x_ratio = v1_cast_len / length(viewscreen_x);
        float y_ratio = v2_cast_len / length(viewscreen_y);
        if (x_ratio < 0.0f || x_ratio > 1.0f || y_ratio < 0.0f || y_ratio > 1.0f) return false;
        x = (int)(xsize * x_ratio);
        y = (int)(ysize * y_ratio);
        if (x > xsize - 1) x = xsize - 1; // Q16
        if (y > ysize - 1) y = ysize - 1;
        return true;
    }
};

// ---------------------------------------------------------------- samplers
// (a) The build's sampler contract (DESIGN.md "Sampler"): Faure-permuted Halton radical
//     inverse exactly as HS::Halton_sampler::sample after init_faure()
//     (external/halton_sampler.h:574-604 init_faure, :627-889 sample, :891-901 invert,
//     :903-.. init_tables, :1418-.. halton2..halton1619), plus a per-(pixel seed, dimension)
//     Cranley-Patterson rotation.  Consumption order = src/sampler.cpp:26-36 (separate 1-D
//     and 2-D counters, 64 table dimensions each, RNG fallback beyond).
struct HaltonTables {
    struct Dim { uint32_t base, digits, perm_off; float scale; };
    std::vector<Dim> dims;
    std::vector<uint16_t> perm; // concatenated Faure permutations, one per base
    HaltonTables() {
        const unsigned max_base = 1619u;
        std::vector<std::vector<uint16_t>> perms(max_base + 1);
        for (unsigned k = 1; k <= 3; ++k) { perms[k].resize(k); for (unsigned i = 0; i < k; ++i) perms[k][i] = i; }
        for (unsigned base = 4; base <= max_base; ++base) {
            perms[base].resize(base);
            const unsigned b = base / 2;
            if (base & 1) {
                for (unsigned i = 0; i < base - 1; ++i)
                    perms[base][i + (i >= b)] = perms[base - 1][i] + (perms[base - 1][i] >= b);
                perms[base][b] = b;
            } else {
                for (unsigned i = 0; i < b; ++i) { perms[base][i] = 2 * perms[b][i]; perms[base][b + i] = 2 * perms[b][i] + 1; }
            }
        }
        unsigned p = 2;
        while (dims.size() < 256) {
            bool prime = true;
            for (unsigned d = 2; d * d <= p; d++) if (p % d == 0) { prime = false; break; }
            if (prime) {
                // digits per table lookup k: largest k with base^k <= 500; lookups G: largest with (base^k)^G < 2^32
                uint64_t bk = p; unsigned k = 1;
                while (bk * p <= 500) { bk *= p; k++; }
                uint64_t tot = bk; unsigned G = 1;
                while (tot * bk < (1ull << 32)) { tot *= bk; G++; }
                Dim d;
                d.base = p; d.digits = k * G; d.perm_off = perm.size();
                d.scale = float(0x1.fffffcp-1 / (double)tot);
                perm.insert(perm.end(), perms[p].begin(), perms[p].end());
                dims.push_back(d);
            }
            p++;
        }
    }
    float sample(unsigned dimension, unsigned index) const {
        if (dimension == 0) { // halton2: bit reversal into the mantissa
            index = (index << 16) | (index >> 16);
            index = ((index & 0x00ff00ff) << 8) | ((index & 0xff00ff00) >> 8);
            index = ((index & 0x0f0f0f0f) << 4) | ((index & 0xf0f0f0f0) >> 4);
            index = ((index & 0x33333333) << 2) | ((index & 0xcccccccc) >> 2);
            index = ((index & 0x55555555) << 1) | ((index & 0xaaaaaaaa) >> 1);
            uint32_t u = 0x3f800000u | (index >> 9);
            float f;
            std::memcpy(&f, &u, 4);
            return f - 1.f;
        }
        const Dim& d = dims[dimension];
        uint32_t acc = 0;
        for (unsigned j = 0; j < d.digits; j++) {
            acc = acc * d.base + perm[d.perm_off + index % d.base];
            index /= d.base;
        }
        return (float)acc * d.scale;
    }
};
static const HaltonTables& halton_tables() { static HaltonTables t; return t; }

static inline uint32_t mix32(uint32_t x) { // "lowbias32" integer finaliser
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static inline float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }
static inline float cp_rotation(uint32_t seed, uint32_t hdim) { return u01(mix32(seed ^ mix32(hdim * 0x9e3779b9u + 0x85ebca6bu))); }
static inline float fallback_uniform(uint32_t seed, uint32_t index, uint32_t hdim) {
    return u01(mix32(mix32(seed ^ mix32(hdim * 0x9e3779b9u + 0x85ebca6bu)) + index * 0xc2b2ae35u));
}
static inline float halton_cp(uint32_t seed, uint32_t index, uint32_t hdim) {
    if (hdim >= 192) return fallback_uniform(seed, index, hdim);
    float u = halton_tables().sample(hdim, index) + cp_rotation(seed, hdim);
    if (u >= 1.0f) u -= 1.0f;
    return u;
}

struct Sampler {
    virtual ~Sampler() {}
    virtual void Advance() = 0;
    virtual float Get1D() = 0;
    virtual vec2 Get2D() = 0;
    // position of the 2-D dimension counter (current_sample2D, src/sampler.hpp:62-66)
    virtual unsigned Pos2D() const = 0;
    virtual void Seek2D(unsigned k) = 0;
};
struct HaltonCPSampler : Sampler {
    uint32_t seed, set = (uint32_t)-1, c1 = 0, c2 = 0;
    HaltonCPSampler(uint32_t s) : seed(s) {}
    void Advance() override { c1 = 0; c2 = 0; set++; }
    unsigned Pos2D() const override { return c2; }
    void Seek2D(unsigned k) override { c2 = k; }
    // logical 2-D dimension k -> Halton dimensions (3k, 3k+1); logical 1-D dimension k -> 3k+2
    float Get1D() override { uint32_t k = c1++; return halton_cp(seed, set, k < 64 ? 3 * k + 2 : 192 + 3 * (k - 64) + 2); }
    vec2 Get2D() override {
        uint32_t k = c2++;
        uint32_t d = k < 64 ? 3 * k : 192 + 3 * (k - 64);
        float x = halton_cp(seed, set, d);
        float y = halton_cp(seed, set, d + 1);
        return vec2(x, y);
    }
};
// (b) The reference's ACTIVE sampler, src/sampler.cpp:5-36,77-116 (libstdc++ <random>),
//     used for the "faithful" CPU baseline and the statistical cross-check.
static unsigned round_up_to_square(unsigned x) {
    float s = std::sqrt((float)x);
    float i;
    float frac = std::modf(s, &i);
    if (frac < 0.0001f) return (unsigned)(i * i);
    return (unsigned)((i + 1) * (i + 1));
}
struct StratifiedSampler : Sampler {
    std::vector<std::vector<float>> samples1D;
    std::vector<std::vector<vec2>> samples2D;
    unsigned dim_count, set_size, cur1 = 0, cur2 = 0, current_set = (unsigned)-1;
    std::mt19937 gen;
    StratifiedSampler(unsigned seed, unsigned dim, unsigned ssize)
        : samples1D(dim, std::vector<float>(round_up_to_square(ssize))),
          samples2D(dim, std::vector<vec2>(round_up_to_square(ssize))),
          dim_count(dim), set_size(round_up_to_square(ssize)), gen(seed) {}
    void PrepareSamples() {
        for (unsigned dim = 0; dim < dim_count; dim++) {
            for (unsigned sample = 0; sample < set_size; sample++) {
                float begin = sample / (float)set_size;
                float len = 1.0f / (float)set_size;
                samples1D[dim][sample] = begin + std::uniform_real_distribution<float>(0.0f, len)(gen);
            }
            std::shuffle(samples1D[dim].begin(), samples1D[dim].end(), gen);
            unsigned sq = std::sqrt(set_size) + 0.5f;
            for (unsigned sy = 0; sy < sq; sy++)
                for (unsigned sx = 0; sx < sq; sx++) {
                    float len = 1.0f / (float)sq;
                    float beginx = sx / (float)sq, beginy = sy / (float)sq;
                    float x = beginx + std::uniform_real_distribution<float>(0.0f, len)(gen);
                    float y = beginy + std::uniform_real_distribution<float>(0.0f, len)(gen);
                    samples2D[dim][sy * sq + sx] = vec2(x, y);
                }
            std::shuffle(samples2D[dim].begin(), samples2D[dim].end(), gen);
        }
    }
    void Advance() override {
        if (current_set == (unsigned)-1) PrepareSamples();
        cur1 = 0; cur2 = 0; current_set++;
    }
    unsigned Pos2D() const override { return cur2; }
    void Seek2D(unsigned k) override { cur2 = k; }
    float Get1D() override {
        return (cur1 < dim_count) ? samples1D[cur1++][current_set] : std::uniform_real_distribution<float>(0.0f, 1.0f)(gen);
    }
    vec2 Get2D() override {
        if (cur2 < dim_count) return samples2D[cur2++][current_set];
        float x = std::uniform_real_distribution<float>(0.0f, 1.0f)(gen);
        float y = std::uniform_real_distribution<float>(0.0f, 1.0f)(gen);
        return vec2(x, y);
    }
};

// ---------------------------------------------------------------- path_tracer.cpp
struct Splat { int x, y; Radiance r; };
struct PixelRenderResult { Radiance main_pixel; std::vector<Splat> side_effects; };

struct PathPoint {
    bool infinity = false;
    vec3 pos, lightN, faceN;
    SystemTransform transform;
    vec3 Vr, Vi;
    const Material* mat = nullptr;
    vec2 texUV;
    Radiance emission;
    float russian_coefficient = 1.0f;
    Spectrum transfer_coefficients;
    Spectrum contribution = Spectrum(1.0f, 1.0f, 1.0f);
    Radiance light_from_source;
};

struct Counters { uint64_t path_rays = 0, shadow_rays = 0; TravStats closest, shadow; bool count_trav = false; };

struct PathTracer {
    const Scene& scene;
    const Camera& camera;
    unsigned xres, yres, multisample;
    float bumpmap_scale, clamp, russian;
    unsigned depth, reverse;
    unsigned samplerSeed;
    unsigned sampler_kind;
    Counters& cnt;

    bool Vis(vec3 a, vec3 b) const {
        cnt.shadow_rays++;
        return scene.Visibility(a, b, cnt.count_trav ? &cnt.shadow : nullptr);
    }

    // path_tracer.cpp:110-306
    std::vector<PathPoint> GeneratePath(Ray r, unsigned depth__, float russian__, Sampler& sampler) const {
        std::vector<PathPoint> path;
        Spectrum cumulative_transfer_coefficients = Spectrum(1.0f, 1.0f, 1.0f);
        Ray current_ray = r;
        unsigned n = 0;
        int last_triangle = -1;
        while (n < depth__) {
            n++;
            cnt.path_rays++;
            Intersection i = scene.FindIntersectKdOtherThan(current_ray, last_triangle, cnt.count_trav ? &cnt.closest : nullptr);
            PathPoint p;
            p.contribution = cumulative_transfer_coefficients;
            if (i.triangle < 0) {
                p.infinity = true;
                p.Vr = -current_ray.direction;
                path.push_back(p);
                break;
            }
            const Triangle& tri = scene.triangles[i.triangle];
            p.pos = current_ray.at(i.t);
            p.faceN = i.Interpolate(scene.normals[tri.va], scene.normals[tri.vb], scene.normals[tri.vc]);
            if (std::isnan(p.faceN.x)) {
                p.faceN = scene.normals[tri.va];
                if (std::isnan(p.faceN.x)) {
                    p.faceN = scene.normals[tri.vb];
                    if (std::isnan(p.faceN.x)) {
                        p.faceN = scene.normals[tri.vc];
                        if (std::isnan(p.faceN.x)) return path;
                    }
                }
            }
            if (length(p.faceN) <= 0.0f) return path;
            p.faceN = normalize(p.faceN);
            p.Vr = -current_ray.direction;
            const Material& mat = scene.materials[tri.mat];
            p.mat = &mat;
            bool has_uv = !(scene.n_texcoords <= tri.va); // Q14
            vec2 a = has_uv ? scene.texcoords[tri.va] : vec2(0, 0);
            vec2 b = has_uv ? scene.texcoords[tri.vb] : vec2(0, 0);
            vec2 c = has_uv ? scene.texcoords[tri.vc] : vec2(0, 0);
            p.texUV = i.Interpolate(a, b, c);
            p.emission = mat.emission;
            if (mat.bump >= 0) {
                const Texture& bm = scene.textures[mat.bump];
                float right = bm.GetSlopeRight(p.texUV);
                float bottom = bm.GetSlopeBottom(p.texUV);
                vec3 tangent = i.Interpolate(scene.tangents[tri.va], scene.tangents[tri.vb], scene.tangents[tri.vc]);
                if (tangent.x * tangent.x + tangent.y * tangent.y + tangent.z * tangent.z < 0.001f) {
                    p.lightN = p.faceN;
                } else {
                    tangent = normalize(tangent);
                    vec3 bitangent = normalize(cross(p.faceN, tangent));
                    vec3 tangent2 = cross(bitangent, p.faceN);
                    p.lightN = normalize(p.faceN + (tangent2 * right + bitangent * bottom) * bumpmap_scale);
                    if (std::isnan(p.lightN.x)) p.lightN = p.faceN;
                }
            } else {
                p.lightN = p.faceN;
            }
            p.transform = SystemTransform(p.lightN, BxDFUpVector);
            vec3 dir;
            bool may_leak;
            vec2 sample = sampler.Get2D();
            std::tie(dir, p.transfer_coefficients, may_leak) = bxdf_sample(scene, mat, p.transform.toLocal(p.Vr), p.texUV, sample);
            bool inside = dir.z < 0;
            dir = p.transform.toGlobal(dir);
            if (!(dot(dir, p.faceN) * dot(p.Vr, p.faceN) > 0) && !may_leak) n += 10000;
            p.Vi = dir;
            if (!mat.no_russian && russian__ > 0.0f && n > 1) p.russian_coefficient = 1.0f / russian__;
            else p.russian_coefficient = 1.0f;
            cumulative_transfer_coefficients = cumulative_transfer_coefficients * p.russian_coefficient;
            cumulative_transfer_coefficients = cumulative_transfer_coefficients * p.transfer_coefficients;
            path.push_back(p);
            if (cumulative_transfer_coefficients.max() < 0.001f) break;
            if (!mat.no_russian && russian__ >= 0.0f && sampler.Get1D() > russian__) break;
            if (n > depth__) break;
            current_ray = Ray(p.pos + p.faceN * scene.epsilon * 10.0f * (inside ? -1.0f : 1.0f), normalize(dir));
            last_triangle = i.triangle;
        }
        return path;
    }

    // path_tracer.cpp:308-512
    PixelRenderResult TracePath(const Ray& r, Sampler& sampler) {
        PixelRenderResult result;
        vec3 camerapos = r.origin;
        vec2 areal_sample = sampler.Get2D();
        vec2 lightdir_sample = sampler.Get2D();
        vec2 choice = sampler.Get2D();
        float tri_pick = sampler.Get1D();
        Light main_light = scene.GetRandomLight(choice, tri_pick, areal_sample);
        const unsigned bounce_dim0 = sampler.Pos2D();

        std::vector<PathPoint> path = GeneratePath(r, depth, russian, sampler);

        vec3 main_light_dir;
        if (main_light.type == Light::FULL_SPHERE) {
            vec3 dir = RandomUtils::Sample2DToSphereUniform(areal_sample);
            main_light.pos += main_light.size * dir;
            if (main_light.size > 0.0f) main_light.normal = dir; // Q15 (defined)
            main_light_dir = RandomUtils::Sample2DToHemisphereCosineDirected(lightdir_sample, normalize(dir));
        } else {
            main_light_dir = RandomUtils::Sample2DToHemisphereCosineDirected(lightdir_sample, main_light.normal);
        }
        std::vector<PathPoint> light_path;
        if (reverse > 0 && main_light.valid) {
            Ray light_ray(main_light.pos + scene.epsilon * main_light.normal * 100.0f, main_light_dir);
            // Sampler contract: the light sub-path draws its BxDF samples from the 2-D
            // dimensions after the `depth` reserved for the forward path, not from wherever the forward
            // path happened to stop (reference: path_tracer.cpp:349 simply continues the counter).  Every
            // table dimension is an independent sample set, so the estimator is unchanged; the fixed
            // position lets the GPU generate the light sub-path before the forward path.
            sampler.Seek2D(bounce_dim0 + depth);
            light_path = GeneratePath(light_ray, reverse, -1.0f, sampler);
        }
        Radiance light_at_path_start =
            Radiance(main_light.color.r, main_light.color.g, main_light.color.b) *
            Spectrum(main_light.intensity * main_light.GetDirectionalFactor(main_light_dir));

        for (unsigned n = 0; n < light_path.size(); n++) {
            PathPoint& p = light_path[n];
            Radiance light_here = p.contribution * light_at_path_start;
            p.light_from_source = light_here;
            if (!p.infinity && Vis(p.pos, camerapos)) {
                vec3 direction = normalize(p.pos - camerapos);
                Radiance q = light_here * bxdf_value(scene, *p.mat, p.transform.toLocal(p.Vr), p.transform.toLocal(-direction), p.texUV);
                float G = std::max(0.0f, dot(p.lightN, -direction)) / distance2(camerapos, p.pos);
                if (G >= 0.00001f && !std::isnan(q.r)) {
                    q = q * Spectrum(G);
                    int x2, y2;
                    if (camera.GetCoordsFromDirection(direction, x2, y2)) result.side_effects.push_back(Splat{x2, y2, q});
                }
            }
        }

        Radiance path_total(0.0f, 0.0f, 0.0f);
        for (unsigned n = 0; n < path.size(); n++) {
            const PathPoint& p = path[n];
            if (p.infinity) {
                Radiance sky_radiance = scene.GetSkyboxRay(p.Vr);
                path_total += p.contribution * sky_radiance;
                continue;
            }
            const Material& mat = *p.mat;
            Radiance total_here(0.0, 0.0, 0.0);
            const Light& light = main_light;
            if (light.valid && Vis(light.pos, p.pos)) {
                vec3 Vi = normalize(light.pos - p.pos);
                Spectrum f = bxdf_value(scene, mat, p.transform.toLocal(Vi), p.transform.toLocal(p.Vr), p.texUV);
                float G = std::fabs(dot(p.lightN, Vi)) / distance2(light.pos, p.pos);
                Radiance inc_l = light.color * Spectrum(light.intensity * light.GetDirectionalFactor(-Vi));
                Radiance out = inc_l * (f * G);
                total_here += out;
            }
            for (unsigned q = 0; q < light_path.size(); q++) {
                const PathPoint& l = light_path[q];
                if (!l.infinity && Vis(l.pos, p.pos)) {
                    vec3 light_to_p = normalize(p.pos - l.pos);
                    vec3 p_to_light = -light_to_p;
                    Spectrum f_light = bxdf_value(scene, *l.mat, l.transform.toLocal(light_to_p), l.transform.toLocal(l.Vr), l.texUV);
                    Spectrum f_point = bxdf_value(scene, *p.mat, p.transform.toLocal(p.Vr), p.transform.toLocal(p_to_light), p.texUV);
                    float G = std::fabs(dot(p.lightN, p_to_light)) / distance2(l.pos, p.pos);
                    total_here += l.light_from_source * (f_light * f_point * G);
                }
            }
            if (dot(p.faceN, p.Vr) > 0) total_here += p.emission;
            total_here.clamp(clamp);
            path_total += total_here * p.contribution;
        }
        path_total.clamp(clamp);
        if (std::isnan(path_total.r) || path_total.r < 0.0f) path_total.r = 0.0f;
        if (std::isnan(path_total.g) || path_total.g < 0.0f) path_total.g = 0.0f;
        if (std::isnan(path_total.b) || path_total.b < 0.0f) path_total.b = 0.0f;
        result.main_pixel = path_total;
        return result;
    }

    // path_tracer.cpp:42-78
    PixelRenderResult RenderPixel(int x, int y) {
        PixelRenderResult total;
        samplerSeed += 0x42424242;
        Sampler* sampler;
        HaltonCPSampler hs(samplerSeed);
        StratifiedSampler* ss = nullptr;
        if (sampler_kind == RGK_SAMPLER_STRATIFIED) { ss = new StratifiedSampler(samplerSeed, 64, multisample); sampler = ss; }
        else sampler = &hs;
        for (unsigned i = 0; i < multisample; i++) {
            sampler->Advance();
            vec2 coords = sampler->Get2D();
            Ray r = camera.IsSimple() ? camera.GetPixelRay(x, y, xres, yres, coords)
                                      : camera.GetPixelRayLens(x, y, xres, yres, coords, sampler->Get2D());
            PixelRenderResult q = TracePath(r, *sampler);
            total.main_pixel += q.main_pixel;
            for (const auto& p : q.side_effects) total.side_effects.push_back(p);
        }
        delete ss;
        return total;
    }
};

struct TileResult {
    std::vector<Radiance> px; // row-major inside the tile
    std::vector<Splat> splats;
};

} // namespace

// ======================================================================= C interface
extern "C" {

void* orc_scene_create(const rgk_scene_desc* d) {
    Scene* s = new Scene;
    s->vertices.resize(d->n_vertices); s->normals.resize(d->n_vertices); s->tangents.resize(d->n_vertices);
    s->texcoords.resize(d->n_vertices);
    s->n_texcoords = d->texcoords ? d->n_vertices : 0;
    for (uint32_t i = 0; i < d->n_vertices; i++) {
        s->vertices[i] = vec3(d->vertices[3 * i], d->vertices[3 * i + 1], d->vertices[3 * i + 2]);
        s->normals[i] = vec3(d->normals[3 * i], d->normals[3 * i + 1], d->normals[3 * i + 2]);
        s->tangents[i] = vec3(d->tangents[3 * i], d->tangents[3 * i + 1], d->tangents[3 * i + 2]);
        if (d->texcoords) s->texcoords[i] = vec2(d->texcoords[2 * i], d->texcoords[2 * i + 1]);
    }
    s->triangles.resize(d->n_triangles);
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        Triangle& t = s->triangles[i];
        t.va = d->tri_indices[3 * i]; t.vb = d->tri_indices[3 * i + 1]; t.vc = d->tri_indices[3 * i + 2];
        t.mat = d->tri_material[i];
    }
    s->textures.resize(d->n_textures);
    s->texel_store.resize(d->n_textures);
    s->texel8_store.resize(d->n_textures);
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rgk_texture& t = d->textures[i];
        Texture& o = s->textures[i];
        o.kind = t.kind; o.xsize = t.width; o.ysize = t.height;
        o.color = Color(t.color[0], t.color[1], t.color[2]);
        if (t.kind == RGK_TEX_RGB32F) {
            s->texel_store[i].assign(t.texels, t.texels + (size_t)3 * t.width * t.height);
            o.data = s->texel_store[i].data();
        } else if (t.kind == RGK_TEX_RGB8) {
            s->texel8_store[i].assign(t.texels8, t.texels8 + (size_t)3 * t.width * t.height);
            s->texel_store[i].assign(t.lut, t.lut + 256);
            o.data8 = s->texel8_store[i].data();
            o.lut = s->texel_store[i].data();
        }
    }
    s->materials.resize(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rgk_material& m = d->materials[i];
        Material& o = s->materials[i];
        o.kind = m.kind; o.no_russian = (m.flags & RGK_MAT_NO_RUSSIAN) != 0;
        o.emission = Radiance(m.emission[0], m.emission[1], m.emission[2]);
        o.roughness = m.roughness; o.ior = m.ior; o.amt1 = m.amount;
        o.diffuse = m.tex_diffuse; o.color = m.tex_color; o.bump = m.tex_bump; o.m1 = m.mix_m1; o.m2 = m.mix_m2;
    }
    for (uint32_t i = 0; i < d->n_pointlights; i++) {
        const rgk_pointlight& l = d->pointlights[i];
        Light o;
        o.type = Light::FULL_SPHERE;
        o.pos = vec3(l.pos[0], l.pos[1], l.pos[2]);
        o.color = Radiance(l.color[0], l.color[1], l.color[2]);
        o.intensity = l.intensity; o.size = l.size;
        s->pointlights.push_back(o);
    }
    for (uint32_t i = 0; i < d->n_areal_lights; i++) {
        ArealLight al;
        for (uint32_t j = d->areal_offsets[i]; j < d->areal_offsets[i + 1]; j++)
            al.triangles_with_areas.push_back(std::make_pair(0.0f, d->areal_tris[j]));
        if (!al.triangles_with_areas.empty()) s->areal_lights.push_back(std::make_pair(0.0f, al));
    }
    s->skybox_mode = d->sky_mode;
    s->skybox_color = Color(d->sky_color[0], d->sky_color[1], d->sky_color[2]);
    s->skybox_intensity = d->sky_intensity; s->skybox_rotate = d->sky_rotate; s->skybox_texture = d->sky_texture;
    if (d->ltc_beckmann) s->ltc[0].assign(d->ltc_beckmann, d->ltc_beckmann + 5 * 64 * 64);
    if (d->ltc_ggx) s->ltc[1].assign(d->ltc_ggx, d->ltc_ggx + 5 * 64 * 64);
    s->Commit();
    return s;
}

void orc_scene_destroy(void* h) { delete (Scene*)h; }

int orc_scene_get_info(void* h, rgk_scene_info* o) {
    Scene* s = (Scene*)h;
    std::memset(o, 0, sizeof(*o));
    o->epsilon = s->epsilon;
    o->bbox_min[0] = s->xBB.first; o->bbox_min[1] = s->yBB.first; o->bbox_min[2] = s->zBB.first;
    o->bbox_max[0] = s->xBB.second; o->bbox_max[1] = s->yBB.second; o->bbox_max[2] = s->zBB.second;
    o->total_areal_power = s->total_areal_power; o->total_point_power = s->total_point_power;
    o->n_nodes = s->compressed_array.size(); o->node_bytes = 8; o->tri_bytes = 4 + 48 + 36;
    o->max_depth = s->kd_max_depth; o->n_leaf_refs = s->compressed_triangles.size();
    return 0;
}

// render_driver.cpp:30-46; ties by (y0,x0) via stable sort of the row-major list (Q18)
int orc_generate_task_list(uint32_t tile_size, uint32_t xres, uint32_t yres, float mid_x, float mid_y,
                           uint32_t seedstart, uint32_t seedcount_base, rgk_tile* tiles, uint32_t* n_tiles) {
    struct T { rgk_tile t; float d; };
    std::vector<T> v;
    for (uint32_t yp = 0; yp < yres; yp += tile_size)
        for (uint32_t xp = 0; xp < xres; xp += tile_size) {
            T t;
            t.t.x0 = xp; t.t.x1 = std::min(xres, xp + tile_size);
            t.t.y0 = yp; t.t.y1 = std::min(yres, yp + tile_size);
            vec2 mp((t.t.x0 + t.t.x1) / 2.0f, (t.t.y0 + t.t.y1) / 2.0f);
            t.d = length(vec2(mid_x - mp.x, mid_y - mp.y));
            v.push_back(t);
        }
    std::stable_sort(v.begin(), v.end(), [](const T& a, const T& b) { return a.d < b.d; });
    if (tiles) for (size_t i = 0; i < v.size(); i++) { tiles[i] = v[i].t; tiles[i].seed = seedstart + seedcount_base + (uint32_t)i; }
    *n_tiles = v.size();
    return 0;
}

// render_driver.cpp:144-190 + tracer.cpp:6-37 + texture.cpp:342-347,403-412
int orc_render_round(void* h, const rgk_camera* cam, const rgk_params* prm, const rgk_tile* tiles, uint32_t n_tiles,
                     float* accum_rgb, uint32_t* accum_count, rgk_counters* out_cnt, int n_threads) {
    Scene* scene = (Scene*)h;
    Camera camera(*cam);
    if (n_threads <= 0) n_threads = std::max(1u, std::thread::hardware_concurrency() - 1); // render_driver.cpp:205-206
    std::vector<TileResult> results(n_tiles);
    std::vector<Counters> cnts(n_threads);
    std::atomic<uint32_t> next(0);
    auto worker = [&](int tid) {
        Counters& cnt = cnts[tid];
        cnt.count_trav = (prm->flags & RGK_FLAG_COUNT_TRAVERSAL) != 0;
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= n_tiles) break;
            const rgk_tile& task = tiles[i];
            PathTracer rt{*scene, camera, prm->xres, prm->yres, prm->multisample, prm->bumpmap_scale, prm->clamp,
                          prm->russian, prm->depth, prm->reverse, task.seed, prm->sampler, cnt};
            TileResult& tr = results[i];
            tr.px.reserve((task.x1 - task.x0) * (task.y1 - task.y0));
            for (unsigned y = task.y0; y < task.y1; y++)
                for (unsigned x = task.x0; x < task.x1; x++) {
                    PixelRenderResult px = rt.RenderPixel(x, y);
                    tr.px.push_back(px.main_pixel);
                    for (auto& s : px.side_effects) tr.splats.push_back(s);
                }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; t++) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    // merge in task order (deterministic; the reference merges in completion order)
    uint64_t paths = 0;
    for (uint32_t i = 0; i < n_tiles; i++) {
        const rgk_tile& task = tiles[i];
        size_t k = 0;
        for (unsigned y = task.y0; y < task.y1; y++)
            for (unsigned x = task.x0; x < task.x1; x++, k++) {
                size_t p = (size_t)y * prm->xres + x;
                accum_rgb[3 * p + 0] += results[i].px[k].r;
                accum_rgb[3 * p + 1] += results[i].px[k].g;
                accum_rgb[3 * p + 2] += results[i].px[k].b;
                accum_count[p] += prm->multisample;
                paths += prm->multisample;
            }
        for (auto& s : results[i].splats) {
            size_t p = (size_t)s.y * prm->xres + s.x;
            accum_rgb[3 * p + 0] += s.r.r; accum_rgb[3 * p + 1] += s.r.g; accum_rgb[3 * p + 2] += s.r.b;
        }
    }
    if (out_cnt) {
        std::memset(out_cnt, 0, sizeof(*out_cnt));
        out_cnt->paths = paths;
        for (auto& c : cnts) {
            out_cnt->path_rays += c.path_rays; out_cnt->shadow_rays += c.shadow_rays;
            out_cnt->node_visits += c.closest.nodes; out_cnt->tri_tests += c.closest.tris;
            out_cnt->shadow_node_visits += c.shadow.nodes; out_cnt->shadow_tri_tests += c.shadow.tris;
        }
    }
    return 0;
}

int orc_trace_closest(void* h, uint32_t n, const float* rays, const int32_t* ignore, rgk_hit* hits, rgk_counters* c) {
    Scene* s = (Scene*)h;
    TravStats st;
    for (uint32_t i = 0; i < n; i++) {
        Ray r;
        r.origin = vec3(rays[8 * i], rays[8 * i + 1], rays[8 * i + 2]);
        r.direction = vec3(rays[8 * i + 3], rays[8 * i + 4], rays[8 * i + 5]);
        r.near = rays[8 * i + 6]; r.far = rays[8 * i + 7];
        Intersection x = s->FindIntersectKdOtherThan(r, ignore ? ignore[i] : -1, &st);
        hits[i].tri = x.triangle; hits[i].t = x.t; hits[i].a = x.a; hits[i].b = x.b; hits[i].c = x.c;
    }
    if (c) { std::memset(c, 0, sizeof(*c)); c->node_visits = st.nodes; c->tri_tests = st.tris; c->path_rays = n; }
    return 0;
}

int orc_trace_visibility(void* h, uint32_t n, const float* a, const float* b, uint8_t* vis, rgk_counters* c) {
    Scene* s = (Scene*)h;
    TravStats st;
    for (uint32_t i = 0; i < n; i++)
        vis[i] = s->Visibility(vec3(a[3 * i], a[3 * i + 1], a[3 * i + 2]), vec3(b[3 * i], b[3 * i + 1], b[3 * i + 2]), &st);
    if (c) { std::memset(c, 0, sizeof(*c)); c->shadow_node_visits = st.nodes; c->shadow_tri_tests = st.tris; c->shadow_rays = n; }
    return 0;
}

float orc_halton_raw(uint32_t dim, uint32_t index) { return halton_tables().sample(dim, index); }

int orc_sampler_eval(uint32_t n, const uint32_t* seed, const uint32_t* index, const uint32_t* dim, int is2d, float* out) {
    for (uint32_t i = 0; i < n; i++) {
        uint32_t k = dim[i];
        if (is2d) {
            uint32_t d = k < 64 ? 3 * k : 192 + 3 * (k - 64);
            out[2 * i] = halton_cp(seed[i], index[i], d);
            out[2 * i + 1] = halton_cp(seed[i], index[i], d + 1);
        } else {
            out[2 * i] = halton_cp(seed[i], index[i], k < 64 ? 3 * k + 2 : 192 + 3 * (k - 64) + 2);
            out[2 * i + 1] = 0.0f;
        }
    }
    return 0;
}

// BxDF::value / BxDF::sample for unit tests (local frame, +Z = shading normal)
int orc_bxdf_value(void* h, uint32_t mat, const float* Vi, const float* Vr, const float* uv, float* out) {
    Scene* s = (Scene*)h;
    Spectrum v = bxdf_value(*s, s->materials[mat], vec3(Vi[0], Vi[1], Vi[2]), vec3(Vr[0], Vr[1], Vr[2]), vec2(uv[0], uv[1]));
    out[0] = v.r; out[1] = v.g; out[2] = v.b;
    return 0;
}
int orc_bxdf_sample(void* h, uint32_t mat, const float* Vi, const float* uv, const float* u, float* out_dir, float* out_w, int* may_leak) {
    Scene* s = (Scene*)h;
    auto r = bxdf_sample(*s, s->materials[mat], vec3(Vi[0], Vi[1], Vi[2]), vec2(uv[0], uv[1]), vec2(u[0], u[1]));
    vec3 d = std::get<0>(r); Spectrum w = std::get<1>(r);
    out_dir[0] = d.x; out_dir[1] = d.y; out_dir[2] = d.z;
    out_w[0] = w.r; out_w[1] = w.g; out_w[2] = w.b;
    *may_leak = std::get<2>(r);
    return 0;
}
int orc_texture_sample(void* h, int tex, const float* uv, float* rgb, float* slope_right, float* slope_bottom) {
    Scene* s = (Scene*)h;
    Color c = s->TexGet(tex, vec2(uv[0], uv[1]));
    rgb[0] = c.r; rgb[1] = c.g; rgb[2] = c.b;
    *slope_right = tex < 0 ? 0 : s->textures[tex].GetSlopeRight(vec2(uv[0], uv[1]));
    *slope_bottom = tex < 0 ? 0 : s->textures[tex].GetSlopeBottom(vec2(uv[0], uv[1]));
    return 0;
}
// the pinned transcendental functions (include/rgk_libm.h) for unit tests: fn 0 sin, 1 cos, 2 acos, 3 asin, 4 atan2(a, b)
int orc_libm(int fn, uint32_t n, const float* a, const float* b, float* out) {
    for (uint32_t i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = rgk_sinf(a[i]); break;
        case 1: out[i] = rgk_cosf(a[i]); break;
        case 2: out[i] = rgk_acosf(a[i]); break;
        case 3: out[i] = rgk_asinf(a[i]); break;
        case 4: out[i] = rgk_atan2f(a[i], b[i]); break;
        default: return -1;
        }
    }
    return 0;
}
// Camera::Camera for unit tests (the oracle's own restatement of src/camera.cpp:7-24)
int orc_camera_init(rgk_camera* out, const float* pos, const float* la, const float* up, float yview, float xview, int xsize, int ysize,
                    float focus_plane, float lens_size) {
    Camera c(vec3(pos[0], pos[1], pos[2]), vec3(la[0], la[1], la[2]), vec3(up[0], up[1], up[2]), yview, xview, xsize, ysize, focus_plane, lens_size);
    c.store(*out);
    return 0;
}
// Camera::GetPixelRay for unit tests
int orc_camera_ray(const rgk_camera* cam, int x, int y, int xres, int yres, const float* sub, const float* lens, float* out6) {
    Camera c(*cam);
    Ray r = c.IsSimple() ? c.GetPixelRay(x, y, xres, yres, vec2(sub[0], sub[1]))
                         : c.GetPixelRayLens(x, y, xres, yres, vec2(sub[0], sub[1]), vec2(lens[0], lens[1]));
    out6[0] = r.origin.x; out6[1] = r.origin.y; out6[2] = r.origin.z;
    out6[3] = r.direction.x; out6[4] = r.direction.y; out6[5] = r.direction.z;
    return 0;
}
int orc_test_intersection(void* h, uint32_t tri, const float* ray8, float* tab) {
    Scene* s = (Scene*)h;
    Ray r;
    r.origin = vec3(ray8[0], ray8[1], ray8[2]); r.direction = vec3(ray8[3], ray8[4], ray8[5]);
    r.near = ray8[6]; r.far = ray8[7];
    float t = 0, a = 0, b = 0;
    bool hit = s->TestIntersection(s->triangles[tri], r, t, a, b);
    tab[0] = t; tab[1] = a; tab[2] = b;
    return hit ? 1 : 0;
}
// the reference's active StratifiedSampler table, for the libstdc++ cross-check
int orc_stratified_sample(uint32_t seed, uint32_t set_size, uint32_t n_sets, uint32_t n_dims, float* out1d, float* out2d) {
    StratifiedSampler s(seed, 64, set_size);
    for (uint32_t i = 0; i < n_sets; i++) {
        s.Advance();
        for (uint32_t d = 0; d < n_dims; d++) {
            out1d[i * n_dims + d] = s.Get1D();
            vec2 v = s.Get2D();
            out2d[2 * (i * n_dims + d)] = v.x; out2d[2 * (i * n_dims + d) + 1] = v.y;
        }
    }
    return 0;
}

} // extern "C"
