// CPU test of the host BVH builder (pbrpathtracer_amd/csrc/bvh_build.cpp): structural validity of the
// device layout (indices in range, every triangle in exactly one leaf, child boxes enclose their
// triangles with the conservative padding, depth bound honoured).  Built and run by tests/test_host_cpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <vector>

#include "bvh_build.h"

using namespace ptk;

static int check(const std::vector<float>& verts, int n, int max_depth, int leaf_max, const char* name)
{
    BuiltBvh b;
    if (!build_bvh(verts.data(), n, max_depth, leaf_max, b)) { std::printf("FAIL %s: build_bvh returned false\n", name); return 1; }
    if (n == 0) { if (b.num_nodes != 0) { std::printf("FAIL %s: nodes for empty scene\n", name); return 1; } return 0; }
    if ((int)b.order.size() != n) { std::printf("FAIL %s: order size\n", name); return 1; }
    std::vector<int> seen(n, 0);
    int maxdepth = 0; long errors = 0;
    std::function<void(int, int)> walk = [&](int node, int depth) {
        if (node < 0 || node >= b.num_nodes) { errors++; return; }
        if (depth > maxdepth) maxdepth = depth;
        if (depth > max_depth + 1) { errors++; return; }
        const float* q = b.nodes.data() + (size_t)node * 16;
        int32_t child[2]; std::memcpy(child, q + 12, 8);
        // planes are stored as (left, right) pairs: min x, y, z then max x, y, z
        const float box[2][6] = { { q[0], q[2], q[4], q[6], q[8], q[10] }, { q[1], q[3], q[5], q[7], q[9], q[11] } };
        for (int c = 0; c < 2; c++)
        {
            if (std::isnan(box[c][0])) continue;      // empty child
            if (child[c] >= 0) { if (child[c] <= node) errors++; walk(child[c], depth + 1); }
            else
            {
                int code = ~child[c]; int first = code >> 3, count = (code & 7) + 1;
                if (first < 0 || first + count > n || count > leaf_max) { errors++; continue; }
                for (int k = first; k < first + count; k++)
                {
                    int t = b.order[k];
                    if (t < 0 || t >= n) { errors++; continue; }
                    seen[t]++;
                    for (int v = 0; v < 3; v++)
                        for (int a = 0; a < 3; a++)
                        {
                            float x = verts[(size_t)t * 9 + v * 3 + a];
                            if (!(x > box[c][a] && x < box[c][3 + a])) errors++;
                        }
                }
            }
        }
    };
    walk(0, 1);
    for (int i = 0; i < n; i++) if (seen[i] != 1) errors++;
    if (maxdepth != b.depth || b.depth > max_depth) errors++;
    if (errors) { std::printf("FAIL %s: %ld errors (depth %d reported %d)\n", name, errors, maxdepth, b.depth); return 1; }
    std::printf("ok %s: n=%d nodes=%d depth=%d pad=%g\n", name, n, b.num_nodes, b.depth, b.pad);
    return 0;
}

int main()
{
    int bad = 0;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    auto random_soup = [&](int n, float size) {
        std::vector<float> v((size_t)n * 9);
        for (int i = 0; i < n; i++)
        {
            float c[3] = { U(rng), U(rng), U(rng) };
            for (int k = 0; k < 9; k++) v[(size_t)i * 9 + k] = c[k % 3] + size * U(rng);
        }
        return v;
    };
    bad += check({}, 0, 32, 4, "empty");
    for (int n : { 1, 2, 3, 4, 5, 12, 100, 5000 })
    {
        char name[64]; std::snprintf(name, sizeof name, "soup%d", n);
        bad += check(random_soup(n, 0.2f), n, 32, 4, name);
    }
    bad += check(random_soup(200000, 0.01f), 200000, 32, 4, "soup200k");
    // all triangles identical: centroid bounds degenerate -> median fallback
    {
        std::vector<float> v; auto one = random_soup(1, 0.3f);
        for (int i = 0; i < 1000; i++) v.insert(v.end(), one.begin(), one.end());
        bad += check(v, 1000, 32, 4, "identical1000");
    }
    // geometric progression along x makes SAH want a degenerate (linked-list) tree: depth bound must hold
    {
        int n = 4000; std::vector<float> v((size_t)n * 9);
        for (int i = 0; i < n; i++)
        {
            float x = std::pow(1.002f, (float)i);
            float tri[9] = { x, 0, 0, x * 1.001f, 1, 0, x * 1.001f, 0, 1 };
            std::memcpy(&v[(size_t)i * 9], tri, sizeof tri);
        }
        bad += check(v, n, 14, 4, "geometric_depth14");
        bad += check(v, n, 32, 1, "geometric_leaf1");
    }
    return bad ? 1 : 0;
}
