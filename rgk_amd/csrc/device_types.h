// Records shared by the host (scene commit / BVH build / upload) and the HIP kernels.
// Everything here is laid out for 16-byte loads: one lane fetches a whole record with
// 1-4 dwordx4 instructions.  Sizes are reported through rgk_scene_info (s_node, s_tri
// of SURVEY 8(d)).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "../../include/rgk.h"

#define RGK_NODE_BYTES 64 // QNode
#define RGK_TRI_BYTES 48  // TriIsect
#define RGK_LTC_TABLE_BYTES (4096u * 32u)

// Host-side intermediate of the build: binary node with both child boxes.  Children < 0 are
// leaves: ~child = (first << 4) | (count - 1), `first` indexing TriIsect records in leaf order.
struct BvhNode {
    float lmin[3], lmax[3]; // left child box
    float rmin[3], rmax[3]; // right child box
    int32_t left, right;
    int32_t pad[2];
};

// The node the kernels traverse: a 4-wide BVH node in ONE 64-byte line (4 dwordx4 per lane).
// Child boxes are 8-bit offsets from the node's own box minimum `p`, in units of a per-axis
// power-of-two step: lo = p + qlo * s, hi = p + qhi * s, rounded outward at build time, so a decoded
// box always contains the (epsilon-padded) child box it stands for.  Box tests only steer the walk;
// results come from the triangle records, so the quantisation cannot change a hit.
// child[i] >= 0: inner node index; < 0: leaf, ~child = (first << 4) | (count - 1);
// RGK_QNODE_EMPTY: unused slot (its box is inverted: qlo = 255, qhi = 0).
#define RGK_QNODE_EMPTY 0x7fffffff
struct QNode {
    float p[3];
    float sx;          // quantisation step of axis x (a power of two); sy, sz at the end of the line
    int32_t child[4];
    uint8_t qlo[3][4]; // [axis][child]
    uint8_t qhi[3][4];
    float sy, sz;
};

// Everything Triangle::TestIntersection (reference src/primitives.cpp:75-166) reads for
// one triangle, with the vertex differences it recomputes per call hoisted to commit time
// (same float subtractions on the same operands, so the values are bit-identical).
struct TriIsect {
    float n[3], d;       // Triangle::p  (plane, CalculatePlane primitives.cpp:24-36)
    float v0a, v0b;      // vert0[i1], vert0[i2]
    float q1x, q1y;      // vert1[i] - vert0[i]
    float q2x, q2y;      // vert2[i] - vert0[i]
    uint32_t axes;       // i1 | (i2 << 2)
    uint32_t tri;        // original triangle index
};

// Everything shading needs about one triangle, gathered behind ONE index (the hit's triangle id):
// 7 x 16 B (+ 16 B pad = one 128-byte line), contiguous -- instead of an index record followed by six scattered vertex fetches.
//   {nA.xyz, uvA.x} {nB.xyz, uvA.y} {nC.xyz, uvB.x} {tA.xyz, uvB.y} {tB.xyz, uvC.x} {tC.xyz, uvC.y} {mat,-,-,-}
struct TriShade {
    float q[6][4];
    uint32_t mat, pad[7]; // padded to 128 B: one record = one cache line
};

// A texture as a material sees it, 16 B so it rides inside the material record (no second,
// dependent descriptor fetch).  kind RGK_TEXREF_NONE = EmptyTexture (id -1: black, Empty()).
// SOLID: a,b,c = colour (float bits).  RGB32F: a = width | height << 16, b = index of texel (0,0)
// in the float4 texel pool.  RGB8: same, b indexes the RGBA8 pool (one dword per texel) and c the
// texture's 256-entry byte -> float table in the LUT pool.
#define RGK_TEXREF_NONE 0xffffffffu
#ifndef RGK_TEX_TILED
#define RGK_TEX_TILED 1 // byte texels in tiles of 8 x 4 = one 128-byte line (rgk_device.h tex_row / tex_col, rgk_host.cpp texel pool); 0: row by row
#endif
struct TexRef {
    uint32_t kind, a, b, c;
};

struct DevMaterial { // 96 B = 6 x 16 B
    uint32_t kind, flags;
    float roughness, ior;
    float emission[3];
    float amount;
    int32_t mix_m1, mix_m2;
    int32_t pad[2];
    TexRef t_diffuse, t_color, t_bump;
};

struct DevPointLight {
    float pos[3];
    float intensity;
    float color[3];
    float size;
};

struct DevArealLight {
    float power;      // areal_lights[i].first  (scene.cpp:338-340)
    float total_area; // ArealLight::total_area
    float emission[3];
    uint32_t first, count; // into areal_tris (sorted by area, descending)
    uint32_t pad;
};

struct DevArealTri {
    float area;
    uint32_t tri;
    float a[3], b[3], c[3]; // vertices A, B, C of the triangle (GetRandomPoint primitives.cpp:61-73)
    float normal_a[3];      // GetNormalA()
    uint32_t light;         // index of the DevArealLight this triangle belongs to
    uint32_t pad;
};

struct DevHaltonDim { // one Halton dimension of the Faure-permuted sampler
    uint32_t base;
    uint32_t digits;    // total digits D the reference's table walk covers
    uint32_t perm_off;  // into the u16 permutation pool
    float scale;        // float(0x1.fffffcp-1 / base^D)
    uint32_t magic;     // division by `base`: q = (mulhi(n, magic) + ((n - mulhi) >> 1)) >> shift
    uint32_t shift;
    uint32_t pad[2];
};

struct DevCamera {
    float origin[3], direction[3], up[3], left[3];
    float viewscreen[3], viewscreen_x[3], viewscreen_y[3];
    float lens_size;
    int32_t xsize, ysize;
};

struct DevScene {
    const DevScene* self; // this record in device memory, for the out-of-line generic BxDF route (rgk_device.h)
    const QNode* nodes;
    const TriIsect* tris;
    const TriShade* tri_shade;
    const DevMaterial* materials;
    const float4* texels; // RGBA float, A unused: one 16-byte load per texel
    const uint32_t* texels8; // RGBA8, A unused: one dword per texel, decoded through `luts`
    const float* luts;
    uint32_t n_lut_floats, n_materials;
    const DevPointLight* pointlights;
    const DevArealLight* areal;
    const DevArealTri* areal_tris;
    const float4* ltc; // GGX table, then Beckmann: each 4096 x {m0,m2,m4,m6}{amp,0,0,0} = RGK_LTC_TABLE_BYTES
    const DevHaltonDim* hdims;
    const uint16_t* hperm;
    uint32_t n_pointlights, n_areal;
    float total_point_power, total_areal_power;
    float epsilon;
    float bb_min[3], bb_max[3];
    uint32_t has_texcoords;
    uint32_t walk_q; // traversal scheduling knob (rgk_trace.h), 0 = walk until every lane is at a leaf
    uint32_t sky_mode;
    float sky_color[3];
    float sky_intensity, sky_rotate;
    TexRef sky_tex;
};
