// Accelerator build on the device (rgk_build.hip), called by rgk_scene_create when rgk_scene_desc.build_flags asks for it.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.h"

struct RgkBuildPrim { // one reference: the (possibly clipped) box of a triangle and the triangle it stands for
    float bmin[3], bmax[3];
    uint32_t tri;
};

// h_prims: n references on the host.  d_recs: TriIsect per ORIGINAL triangle id, on the device.  d_nodes (capacity n QNodes) and
// d_leaf_recs (n records) are filled.  Returns 0, or a negative rgk_status with *err set.
int rgk_build_bvh4_device(hipStream_t st, const RgkBuildPrim* h_prims, uint32_t n, const float smin[3], const float smax[3], float pad,
                          uint32_t max_leaf, const TriIsect* d_recs, QNode* d_nodes, TriIsect* d_leaf_recs, uint32_t* n_nodes,
                          uint32_t* n_levels, const char** err);
