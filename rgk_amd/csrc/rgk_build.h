// Accelerator build on the device (rgk_build.hip), called by rgk_scene_create when rgk_scene_desc.build_flags asks for it.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.h"

struct RgkBuildPrim { // one reference: the (possibly clipped) box of a triangle and the triangle it stands for
    float bmin[3], bmax[3];
    uint32_t tri;
    float pb[4]; // the piece in the triangle's own coordinates (rgk_host.cpp Prim::pb), carried to the leaf order for refits
};

// h_prims: n references on the host.  d_recs: TriIsect per ORIGINAL triangle id, on the device.  d_nodes (capacity n QNodes) and
// d_leaf_recs (n records) are filled.  rotate: passes of the refit's quality step (tree rotations; 0 = plain LBVH).  Returns 0, or a negative
// rgk_status with *err set.
int rgk_build_bvh4_device(hipStream_t st, const RgkBuildPrim* h_prims, uint32_t n, const float smin[3], const float smax[3], float pad,
                          uint32_t max_leaf, int rotate, const TriIsect* d_recs, QNode* d_nodes, TriIsect* d_leaf_recs, float4* d_leaf_pb, uint32_t* n_nodes,
                          uint32_t* n_levels, const char** err);

// Refit for moved vertices (rgk_scene_refit): the references' intersection records recomputed from d_vertices (TriIsect.tri names the
// triangle of each), their boxes, then every node of the 4-wide tree bottom-up -- child boxes, node box, 8-bit codes; the
// topology stays.  d_normals / d_tangents (either may be null): the shading records' normals / tangents as well.
int rgk_refit_bvh4_device(hipStream_t st, uint32_t n_refs, uint32_t n_nodes, uint32_t n_tris, const float* d_vertices, const float* d_normals, const float* d_tangents,
                          const uint32_t* d_idx, const float4* d_leaf_pb, float pad, TriIsect* d_leaf_recs, QNode* d_nodes, TriShade* d_shade, const char** err);
