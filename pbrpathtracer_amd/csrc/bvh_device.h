// Device BVH builder: binned SAH built level by level ON THE GPU, collapsed to the 4-wide quantised
// node layout of ptk_device.h, leaf order included.  Replaces BVHNode::Construct (reference PathTracing/src/mesh.cpp:169-211,
// called from pathtracer.cpp:260-274) for scenes large enough to matter; bvh_build.cpp (host) stays as the small-scene
// builder, the fallback and the comparator of the tests.  Closest-hit results do not depend on the tree.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace ptk {

struct DeviceBvh {
    float4* d_nodes = nullptr;       // num_nodes x 4 float4 (ptk_device.h BVH4 record); owned by the caller after a successful build
    int32_t* d_order = nullptr;      // leaf order -> scene triangle index, n entries; owned by the caller
    int32_t num_nodes = 0, depth = 0, stack_need = 0;
    float pad = 0.0f;
    double ms_levels = 0.0, ms_collapse = 0.0;      // host wall time of the two phases (diagnostics)
};

// d_verts: [n][9] world-space v1 v2 v3 on the device.  max_stack: entries of the kernel's traversal stack (also bounds the
// binary depth); leaf_max: 1..8.  Runs on `stream` (with small device->host reads between levels).  Returns false - with
// nothing allocated in `out` - when the tree cannot meet the bound (the caller then uses the host builder) or on a HIP error.
bool build_bvh_device(const float* d_verts, int32_t n, int max_stack, int leaf_max, hipStream_t stream, DeviceBvh& out, std::string* err);

// record packing on the device (ptk_device.h layouts) from the boundary's flat arrays already in device memory
void launch_pack_tris(const float* d_verts, const int32_t* d_order, const int32_t* d_material, const int32_t* d_mat_opacity_tex, float4* d_tris, int n, hipStream_t stream);
void launch_pack_shade(const float* d_normals, const float* d_uvs, const float* d_tbn, const uint8_t* d_smoothing, const int32_t* d_material, float4* d_shade, int n,
                       hipStream_t stream);

}  // namespace ptk
