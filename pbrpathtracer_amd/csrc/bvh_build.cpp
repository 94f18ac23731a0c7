// Binned-SAH BVH2 builder (host).  See bvh_build.h.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>

namespace ptk {
namespace {

constexpr int kBins = 16;
float kTravCost = 1.0f;             // one node record = two slab tests (PTK_TRAV_COST overrides, experiments)
constexpr float kTriCost = 1.0f;

struct Box {
    float mn[3], mx[3];
    void reset()
    {
        for (int a = 0; a < 3; a++) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -mn[a]; }
    }
    void grow(const Box& o)
    {
        for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], o.mn[a]); mx[a] = std::max(mx[a], o.mx[a]); }
    }
    void grow(const float* p)
    {
        for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); }
    }
    float half_area() const
    {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct TmpNode {
    Box box;
    int32_t left = -1, right = -1;   // interior: tmp indices
    int32_t first = 0, count = 0;    // leaf: range in `order`
};

struct Builder {
    const float* verts;
    int32_t n;
    int max_depth, leaf_max;
    std::vector<Box> tbox;
    std::vector<float> cent;         // [n][3]
    std::vector<int32_t> order;
    std::vector<TmpNode> nodes;
    std::atomic<int32_t> next{0};
    std::atomic<bool> failed{false};

    // interior levels a balanced median-split subtree over c triangles needs
    int need(int64_t c) const
    {
        int lv = 0;
        while (c > leaf_max) { c = (c + 1) / 2; lv++; }
        return lv;
    }

    int32_t alloc() { return next.fetch_add(1); }

    int32_t make_leaf(int32_t id, int32_t first, int32_t count)
    {
        nodes[id].first = first; nodes[id].count = count; nodes[id].left = nodes[id].right = -1;
        return id;
    }

    // level: 1-based count of interior nodes from the root down to (and including) this node, were it interior
    int32_t build(int32_t first, int32_t count, int level, int par_depth)
    {
        int32_t id = alloc();
        Box b; b.reset();
        Box cb; cb.reset();
        for (int32_t i = first; i < first + count; i++)
        {
            int32_t t = order[i];
            b.grow(tbox[t]);
            cb.grow(&cent[(size_t)t * 3]);
        }
        nodes[id].box = b;
        if (count <= 1) return make_leaf(id, first, count);

        int32_t mid = -1;
        bool force_median = (level - 1 + need(count)) >= max_depth;   // no slack left: stay balanced
        float leaf_cost = kTriCost * (float)count;
        if (!force_median)
        {
            float best_cost = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_bin = -1;
            float parent_area = std::max(b.half_area(), 1e-30f);
            for (int axis = 0; axis < 3; axis++)
            {
                float lo = cb.mn[axis], hi = cb.mx[axis];
                if (!(hi > lo)) continue;
                float scale = (float)kBins / (hi - lo);
                Box bb[kBins]; int32_t bc[kBins];
                for (int k = 0; k < kBins; k++) { bb[k].reset(); bc[k] = 0; }
                for (int32_t i = first; i < first + count; i++)
                {
                    int32_t t = order[i];
                    int k = (int)((cent[(size_t)t * 3 + axis] - lo) * scale);
                    k = std::min(std::max(k, 0), kBins - 1);
                    bb[k].grow(tbox[t]); bc[k]++;
                }
                float right_area[kBins]; int32_t right_cnt[kBins];
                Box acc; acc.reset(); int32_t c = 0;
                for (int k = kBins - 1; k > 0; k--)
                {
                    acc.grow(bb[k]); c += bc[k];
                    right_area[k] = acc.half_area(); right_cnt[k] = c;
                }
                acc.reset(); c = 0;
                for (int k = 0; k < kBins - 1; k++)
                {
                    acc.grow(bb[k]); c += bc[k];
                    if (c == 0 || right_cnt[k + 1] == 0) continue;
                    float cost = kTravCost + kTriCost * (acc.half_area() * (float)c + right_area[k + 1] * (float)right_cnt[k + 1]) / parent_area;
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = k; }
                }
            }
            if (count <= leaf_max && !(best_cost < leaf_cost)) return make_leaf(id, first, count);
            if (best_axis >= 0)
            {
                float lo = cb.mn[best_axis], hi = cb.mx[best_axis];
                float scale = (float)kBins / (hi - lo);
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int32_t t) {
                    int k = (int)((cent[(size_t)t * 3 + best_axis] - lo) * scale);
                    k = std::min(std::max(k, 0), kBins - 1);
                    return k <= best_bin;
                });
                mid = (int32_t)(it - order.begin());
                int32_t nl = mid - first, nr = count - nl;
                if (nl == 0 || nr == 0 || level + need(std::max(nl, nr)) > max_depth) mid = -1;   // would break the bound
            }
        }
        if (mid < 0)
        {
            if (count <= leaf_max && force_median == false) return make_leaf(id, first, count);
            if (count <= leaf_max) return make_leaf(id, first, count);
            // object median along the longest centroid axis
            int axis = 0;
            float e0 = cb.mx[0] - cb.mn[0], e1 = cb.mx[1] - cb.mn[1], e2 = cb.mx[2] - cb.mn[2];
            if (e1 > e0 && e1 >= e2) axis = 1; else if (e2 > e0 && e2 > e1) axis = 2;
            mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](int32_t a, int32_t c) {
                                 float ca = cent[(size_t)a * 3 + axis], cc = cent[(size_t)c * 3 + axis];
                                 return ca < cc || (ca == cc && a < c);
                             });
        }
        if (level > max_depth) { failed = true; return make_leaf(id, first, count); }
        int32_t nl = mid - first, nr = count - nl;
        int32_t l, r;
        if (par_depth < 3 && count > 32768)
        {
            auto fut = std::async(std::launch::async, [&, this] { return build(first, nl, level + 1, par_depth + 1); });
            r = build(mid, nr, level + 1, par_depth + 1);
            l = fut.get();
        }
        else
        {
            l = build(first, nl, level + 1, par_depth + 1);
            r = build(mid, nr, level + 1, par_depth + 1);
        }
        nodes[id].left = l; nodes[id].right = r;
        return id;
    }
};

inline int32_t leaf_code(int32_t first, int32_t count) { return ~((first << 3) | (count - 1)); }

}  // namespace

bool build_bvh(const float* verts, int32_t n, int max_depth, int leaf_max, BuiltBvh& out)
{
    out = BuiltBvh();
    if (n <= 0) return true;
    kTravCost = 1.0f;
    // A handful of triangles (a bare Cornell box) gains nothing from a hierarchy: the surface-area
    // heuristic overrates splits whose children still span the whole room, and every extra level costs
    // a dependent 64-byte fetch.  Measured on MI355X (tools/perf_probe.py): two 6-triangle leaves under
    // one node beat the 5-node tree by 16 % on the 12-triangle box; larger scenes prefer <= 4 per leaf.
    if (n <= 16) { leaf_max = 8; kTravCost = 2.0f; }
    if (const char* e = std::getenv("PTK_LEAF_MAX")) leaf_max = std::atoi(e);          // experiments only
    if (const char* e = std::getenv("PTK_TRAV_COST")) kTravCost = (float)std::atof(e);
    if (leaf_max < 1) leaf_max = 1;
    if (leaf_max > 8) leaf_max = 8;
    if ((int64_t)n >= (1ll << 27)) return false;     // leaf code packs first << 3 into 31 bits

    Builder B;
    B.verts = verts; B.n = n; B.max_depth = max_depth; B.leaf_max = leaf_max;
    B.tbox.resize(n); B.cent.resize((size_t)n * 3); B.order.resize(n);
    float ext = 1.0f;
    for (int32_t i = 0; i < n; i++)
    {
        const float* p = verts + (size_t)i * 9;
        Box b; b.reset();
        b.grow(p); b.grow(p + 3); b.grow(p + 6);
        for (int a = 0; a < 3; a++)
        {
            B.cent[(size_t)i * 3 + a] = 0.5f * (b.mn[a] + b.mx[a]);
            ext = std::max(ext, std::max(std::fabs(b.mn[a]), std::fabs(b.mx[a])));
        }
        B.tbox[i] = b; B.order[i] = i;
    }
    if (!std::isfinite(ext)) ext = 1.0f;
    // Padding makes box culling conservative with respect to the float rounding of the slab test and
    // of Moeller-Trumbore (a ray that the triangle test accepts enters the padded box strictly
    // earlier); the reference only inflates zero-thickness boxes by EPS (mesh.cpp:33-46).
    const float pad = 1e-5f * ext;
    for (int32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { B.tbox[i].mn[a] -= pad; B.tbox[i].mx[a] += pad; }
    if (B.need(n) > max_depth) return false;
    B.nodes.resize((size_t)2 * n + 2);
    int32_t root = B.build(0, n, 1, 0);
    if (B.failed) return false;

    // flatten: interior nodes only, DFS pre-order; child boxes are stored in the parent
    std::vector<int32_t> index(B.nodes.size(), -1);
    std::vector<int32_t> stack;
    int32_t num = 0;
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    auto is_leaf = [&](int32_t id) { return B.nodes[id].left < 0; };
    std::vector<int32_t> interior;
    if (is_leaf(root))
    {
        // a scene of <= leaf_max triangles: synthesise a root whose right child is an empty (NaN) box
        out.nodes.assign(16, 0.0f);
        const TmpNode& L = B.nodes[root];
        float* q = out.nodes.data();
        for (int a = 0; a < 3; a++)
        {
            q[2 * a] = L.box.mn[a]; q[2 * a + 1] = qnan;            // (left, right) plane pairs
            q[6 + 2 * a] = L.box.mx[a]; q[6 + 2 * a + 1] = qnan;
        }
        int32_t lc = leaf_code(L.first, L.count), rc = leaf_code(L.first, 1);
        std::memcpy(&q[12], &lc, 4); std::memcpy(&q[13], &rc, 4);
        out.num_nodes = 1; out.depth = 1;
    }
    else
    {
        stack.push_back(root);
        while (!stack.empty())
        {
            int32_t id = stack.back(); stack.pop_back();
            index[id] = num++;
            interior.push_back(id);
            int32_t l = B.nodes[id].left, r = B.nodes[id].right;
            if (!is_leaf(r)) stack.push_back(r);
            if (!is_leaf(l)) stack.push_back(l);
        }
        out.nodes.assign((size_t)num * 16, 0.0f);
        out.num_nodes = num;
        for (int32_t id : interior)
        {
            const TmpNode& N = B.nodes[id];
            const TmpNode& L = B.nodes[N.left];
            const TmpNode& R = B.nodes[N.right];
            float* q = out.nodes.data() + (size_t)index[id] * 16;
            for (int a = 0; a < 3; a++)
            {
                q[2 * a] = L.box.mn[a]; q[2 * a + 1] = R.box.mn[a];   // (left, right) plane pairs
                q[6 + 2 * a] = L.box.mx[a]; q[6 + 2 * a + 1] = R.box.mx[a];
            }
            int32_t lc = is_leaf(N.left) ? leaf_code(L.first, L.count) : index[N.left];
            int32_t rc = is_leaf(N.right) ? leaf_code(R.first, R.count) : index[N.right];
            std::memcpy(&q[12], &lc, 4); std::memcpy(&q[13], &rc, 4);
        }
        // depth = max number of interior nodes on a root-to-leaf chain
        std::vector<std::pair<int32_t, int>> st; st.push_back({ root, 1 });
        int depth = 0;
        while (!st.empty())
        {
            auto [id, d] = st.back(); st.pop_back();
            depth = std::max(depth, d);
            int32_t l = B.nodes[id].left, r = B.nodes[id].right;
            if (!is_leaf(l)) st.push_back({ l, d + 1 });
            if (!is_leaf(r)) st.push_back({ r, d + 1 });
        }
        out.depth = depth;
    }
    out.order = std::move(B.order);
    out.pad = pad;
    return out.depth <= max_depth;
}

}  // namespace ptk
