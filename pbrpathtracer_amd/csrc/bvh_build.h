// Host BVH builder for the device layout of ptk_device.h.
//
// Replaces BVHNode::Construct (reference PathTracing/src/mesh.cpp:169-211: random-axis median split,
// one triangle per leaf, O(n log^2 n) with comparator-rebuilt boxes) by a binned-SAH tree with up to
// LEAF_MAX triangles per leaf, collapsed to 4-wide nodes whose child boxes are quantised (outward) to 8 bits on a
// per-node grid - one 64-byte record per node, half the dependent fetches of the binary tree - under a hard bound on the
// traversal stack so the kernel's LDS stack can never overflow.
// Closest-hit results do not depend on the tree, only traversal cost does.
#pragma once

#include <cstdint>
#include <vector>

namespace ptk {

struct BuiltBvh {
    std::vector<float> nodes;        // num_nodes * 16 floats (4 x float4, see ptk_device.h)
    std::vector<int32_t> order;      // leaf-order -> scene triangle index
    int32_t num_nodes = 0;
    int32_t depth = 0;               // wide nodes on the longest root-to-leaf chain
    int32_t stack_need = 0;          // most entries an ordered depth-first traversal can have deferred at once
    float pad = 0.0f;                // box padding that makes culling conservative
};

// verts: [n][9] world-space v1 v2 v3.  max_depth: stack entries available per lane (also bounds the binary tree's depth).
// leaf_max: 1..8 triangles per leaf.  Returns false if the stack bound cannot be met.
bool build_bvh(const float* verts, int32_t n, int max_depth, int leaf_max, BuiltBvh& out);

// Builder tuning of both builders (ptk_set_option "bvh_leaf_max", "bvh_trav_cost", "bvh_verbose"): 0 = the builders' own
// choices.  They shape the tree, never a result (closest hits do not depend on the tree).
struct BvhTuning { int leaf_max = 0; float trav_cost = 0.0f; int verbose = 0; };
extern BvhTuning g_bvh_tuning;

}  // namespace ptk
