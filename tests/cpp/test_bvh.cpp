// CPU test of the host BVH builder (pbrpathtracer_amd/csrc/bvh_build.cpp): structural validity of the
// device layout (4-wide nodes with 8-bit quantised child boxes: indices in range, every triangle in exactly one leaf,
// the quantised child boxes enclose their triangles strictly, nested boxes nest, stack bound honoured and reported).  Built and run by tests/test_host_cpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <vector>

#include "bvh_build.h"

using namespace ptk;

// device node (ptk_device.h): 16 dwords = origin.xyz, scale.xyz, 4 links, lo.x lo.y lo.z hi.x hi.y hi.z (one byte per child)
static int check(const std::vector<float>& verts, int n, int max_stack, int leaf_max, const char* name)
{
    BuiltBvh b;
    if (!build_bvh(verts.data(), n, max_stack, leaf_max, b)) { std::printf("FAIL %s: build_bvh returned false\n", name); return 1; }
    if (n == 0) { if (b.num_nodes != 0) { std::printf("FAIL %s: nodes for empty scene\n", name); return 1; } return 0; }
    if ((int)b.order.size() != n) { std::printf("FAIL %s: order size\n", name); return 1; }
    std::vector<int> seen(n, 0), visited(b.num_nodes, 0);
    int maxdepth = 0, maxneed = 0; long errors = 0; long children = 0;
    // (node, wide depth, entries deferred above it)
    std::function<void(int, int, int)> walk = [&](int node, int depth, int used) {
        if (node < 0 || node >= b.num_nodes) { errors++; return; }
        if (visited[node]++) { errors++; return; }
        if (depth > maxdepth) maxdepth = depth;
        const float* q = b.nodes.data() + (size_t)node * 16;
        int32_t link[4]; uint32_t lo[3], hi[3];
        std::memcpy(link, q + 6, 16); std::memcpy(lo, q + 10, 12); std::memcpy(hi, q + 13, 12);
        int nc = 0;
        for (int c = 0; c < 4; c++) if (((lo[0] >> (8 * c)) & 255u) <= ((hi[0] >> (8 * c)) & 255u)) nc++;
        if (nc == 0) { errors++; return; }
        children += nc;
        const int used_here = used + nc - 1;
        if (used_here > maxneed) maxneed = used_here;
        for (int c = 0; c < 4; c++)
        {
            double box[6];
            bool empty = false;
            for (int a = 0; a < 3; a++)
            {
                const unsigned ql = (lo[a] >> (8 * c)) & 255u, qh = (hi[a] >> (8 * c)) & 255u;
                if (ql > qh) empty = true;
                box[a] = (double)q[a] + ql * (double)q[3 + a]; box[3 + a] = (double)q[a] + qh * (double)q[3 + a];
            }
            if (empty)
            {
                // an empty slot is empty on every axis and never followed
                for (int a = 0; a < 3; a++) if (((lo[a] >> (8 * c)) & 255u) <= ((hi[a] >> (8 * c)) & 255u)) errors++;
                continue;
            }
            // every triangle below must lie strictly inside the quantised box (it encloses the padded boxes)
            std::function<void(int32_t)> inside = [&](int32_t l) {
                if (l >= 0)
                {
                    const float* cq = b.nodes.data() + (size_t)l * 16;
                    int32_t cl[4]; uint32_t clo[3], chi[3]; std::memcpy(cl, cq + 6, 16); std::memcpy(clo, cq + 10, 12); std::memcpy(chi, cq + 13, 12);
                    // the child's own union box must be inside this child box (up to the grid steps of both levels)
                    for (int a = 0; a < 3; a++)
                    {
                        double cmin = 1e300, cmax = -1e300;
                        for (int k = 0; k < 4; k++)
                        {
                            const unsigned ql = (clo[a] >> (8 * k)) & 255u, qh = (chi[a] >> (8 * k)) & 255u;
                            if (ql > qh) continue;
                            cmin = std::min(cmin, (double)cq[a] + ql * (double)cq[3 + a]); cmax = std::max(cmax, (double)cq[a] + qh * (double)cq[3 + a]);
                        }
                        const double slack = (double)cq[3 + a] + (double)q[3 + a];
                        if (cmin < box[a] - slack || cmax > box[3 + a] + slack) errors++;
                    }
                    return;
                }
                int code = ~l; int first = code >> 3, count = (code & 7) + 1;
                if (first < 0 || first + count > n || count > leaf_max) { errors++; return; }
                for (int k = first; k < first + count; k++)
                {
                    int t = b.order[k];
                    if (t < 0 || t >= n) { errors++; continue; }
                    seen[t]++;
                    for (int v = 0; v < 3; v++)
                        for (int a = 0; a < 3; a++)
                        {
                            double x = verts[(size_t)t * 9 + v * 3 + a];
                            if (!(x > box[a] && x < box[3 + a])) errors++;
                        }
                }
            };
            inside(link[c]);
            if (link[c] >= 0) { if (link[c] <= node) errors++; walk(link[c], depth + 1, used_here); }
        }
    };
    walk(0, 1, 0);
    for (int i = 0; i < n; i++) if (seen[i] != 1) errors++;
    for (int k = 0; k < b.num_nodes; k++) if (visited[k] != 1) errors++;
    if (maxdepth != b.depth) errors++;
    if (maxneed != b.stack_need || b.stack_need > max_stack) errors++;
    if (errors) { std::printf("FAIL %s: %ld errors (depth %d reported %d, stack %d reported %d)\n", name, errors, maxdepth, b.depth, maxneed, b.stack_need); return 1; }
    std::printf("ok %s: n=%d nodes=%d depth=%d stack=%d children/node=%.2f pad=%g\n", name, n, b.num_nodes, b.depth, b.stack_need,
                (double)children / b.num_nodes, b.pad);
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
