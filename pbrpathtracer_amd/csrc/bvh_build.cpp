// Binned-SAH builder (host): BVH2 by binned SAH, collapsed to a 4-wide tree with 8-bit quantised child boxes.  See bvh_build.h.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>

namespace ptk {

BvhTuning g_bvh_tuning;
namespace {

constexpr int32_t NODE_EXIT_CODE = INT32_MIN;     // = NODE_EXIT of ptk_device.h: a link the walk never follows

#ifndef PTK_BVH_BINS
#define PTK_BVH_BINS 16
#endif
constexpr int kBins = PTK_BVH_BINS;
float kTravCost = 1.0f;             // one node record = two slab tests (ptk_set_option "bvh_trav_cost" overrides, experiments)
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

// ---- BVH2 -> BVH4 collapse + 8-bit quantisation (device node layout: ptk_device.h) --------------------------------
struct WideNode { int32_t child[4]; int n = 0; };        // tmp-node ids of the children (interior or leaf)

struct Collapser {
    const std::vector<TmpNode>& nodes;
    std::vector<int> hb;                                  // binary height in interior nodes below and including a tmp node (leaf = 0)
    std::vector<WideNode> wide;                           // emitted in DFS pre-order: wide[k] is device node k
    std::vector<int32_t> wide_of;                         // tmp interior id -> device node index (only for roots of wide nodes)
    int stack_need = 0, depth = 0;

    explicit Collapser(const std::vector<TmpNode>& n) : nodes(n), hb(n.size(), 0), wide_of(n.size(), -1) {}
    bool is_leaf(int32_t id) const { return nodes[id].left < 0; }

    int height(int32_t id)
    {
        // iterative post-order (trees of a million triangles are deep enough to matter for the C stack only when degenerate)
        std::vector<std::pair<int32_t, int>> st; st.push_back({ id, 0 });
        while (!st.empty())
        {
            auto [x, phase] = st.back(); st.pop_back();
            if (is_leaf(x)) { hb[x] = 0; continue; }
            if (phase == 0) { st.push_back({ x, 1 }); st.push_back({ nodes[x].left, 0 }); st.push_back({ nodes[x].right, 0 }); }
            else hb[x] = 1 + std::max(hb[nodes[x].left], hb[nodes[x].right]);
        }
        return hb[id];
    }

    // The traversal keeps at most (children - 1) deferred entries per wide node on the current root-to-leaf chain, so a
    // subtree needs (n - 1) + max over its interior children of their need.  `budget` = stack entries left for the subtree
    // of tmp node `id`; invariant budget >= hb[id] (a purely binary subtree needs exactly hb).  Children are absorbed
    // largest-surface-first (the usual SAH collapse) while every member still fits the budget left after this node.
    void collapse(int32_t root, int budget0)
    {
        struct Item { int32_t id; int budget; int used; int level; };
        std::vector<Item> st; st.push_back({ root, budget0, 0, 1 });
        while (!st.empty())
        {
            Item it = st.back(); st.pop_back();
            WideNode w;
            w.child[0] = nodes[it.id].left; w.child[1] = nodes[it.id].right; w.n = 2;
            for (;;)
            {
                if (w.n == 4) break;
                int best = -1; float best_area = -1.0f;
                const int rem = it.budget - w.n;           // budget of the children once this node holds w.n + 1 of them
                for (int k = 0; k < w.n; k++)
                {
                    const int32_t c = w.child[k];
                    if (is_leaf(c)) continue;
                    bool ok = hb[nodes[c].left] <= rem && hb[nodes[c].right] <= rem;
                    for (int j = 0; j < w.n && ok; j++) if (j != k && hb[w.child[j]] > rem) ok = false;
                    if (!ok) continue;
                    const float a = nodes[c].box.half_area();
                    if (a > best_area) { best_area = a; best = k; }
                }
                if (best < 0) break;
                const int32_t c = w.child[best];
                w.child[best] = nodes[c].left;
                w.child[w.n++] = nodes[c].right;
            }
            wide_of[it.id] = (int32_t)wide.size();
            wide.push_back(w);
            const int used = it.used + (w.n - 1);
            stack_need = std::max(stack_need, used);
            depth = std::max(depth, it.level);
            // DFS pre-order: push in reverse so that child 0's subtree follows its parent in memory
            for (int k = w.n - 1; k >= 0; k--)
                if (!is_leaf(w.child[k])) st.push_back({ w.child[k], it.budget - (w.n - 1), used, it.level + 1 });
        }
    }
};

// One device node: see ptk_device.h.  Boxes are quantised OUTWARD on a per-node 8-bit grid (origin + q * scale).
void emit_node(const std::vector<TmpNode>& nodes, const Collapser& C, const WideNode& w, float* q)
{
    Box u; u.reset();
    for (int k = 0; k < w.n; k++) u.grow(nodes[w.child[k]].box);
    float origin[3], scale[3];
    for (int a = 0; a < 3; a++)
    {
        origin[a] = u.mn[a];
        double ext = (double)u.mx[a] - (double)u.mn[a];
        float s = (float)(ext / 255.0 * (1.0 + 1e-6));
        if (!(s > 1e-30f)) s = 1e-30f;
        while ((double)origin[a] + 255.0 * (double)s < (double)u.mx[a]) s = std::nextafter(s, std::numeric_limits<float>::infinity());
        scale[a] = s;
    }
    uint32_t lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
    int32_t link[4];
    for (int k = 0; k < 4; k++)
    {
        if (k >= w.n)
        {
            // empty slot: an inverted box no ray can enter
            for (int a = 0; a < 3; a++) { lo[a] |= 255u << (8 * k); hi[a] |= 0u << (8 * k); }
            link[k] = NODE_EXIT_CODE;
            continue;
        }
        const TmpNode& c = nodes[w.child[k]];
        for (int a = 0; a < 3; a++)
        {
            const double o = origin[a], s = scale[a];
            int ql = (int)std::floor(((double)c.box.mn[a] - o) / s);
            int qh = (int)std::ceil(((double)c.box.mx[a] - o) / s);
            ql = std::min(std::max(ql, 0), 255); qh = std::min(std::max(qh, 0), 255);
            while (ql > 0 && o + ql * s > (double)c.box.mn[a]) ql--;
            while (qh < 255 && o + qh * s < (double)c.box.mx[a]) qh++;
            lo[a] |= (uint32_t)ql << (8 * k); hi[a] |= (uint32_t)qh << (8 * k);
        }
        link[k] = c.left < 0 ? leaf_code(c.first, c.count) : C.wide_of[w.child[k]];
    }
    q[0] = origin[0]; q[1] = origin[1]; q[2] = origin[2]; q[3] = scale[0];
    q[4] = scale[1]; q[5] = scale[2];
    std::memcpy(&q[6], &link[0], 4); std::memcpy(&q[7], &link[1], 4); std::memcpy(&q[8], &link[2], 4); std::memcpy(&q[9], &link[3], 4);
    std::memcpy(&q[10], &lo[0], 4); std::memcpy(&q[11], &lo[1], 4); std::memcpy(&q[12], &lo[2], 4);
    std::memcpy(&q[13], &hi[0], 4); std::memcpy(&q[14], &hi[1], 4); std::memcpy(&q[15], &hi[2], 4);
}

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
    if (g_bvh_tuning.leaf_max > 0) leaf_max = g_bvh_tuning.leaf_max;                   // ptk_set_option "bvh_leaf_max" / "bvh_trav_cost"
    if (g_bvh_tuning.trav_cost > 0.0f) kTravCost = g_bvh_tuning.trav_cost;
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

    // BVH2 -> BVH4: collapse under the stack budget, then emit quantised 64-byte nodes in DFS pre-order
    auto is_leaf = [&](int32_t id) { return B.nodes[id].left < 0; };
    if (is_leaf(root))
    {
        // a scene of <= leaf_max triangles: one node whose only child is the leaf
        Collapser C(B.nodes);
        WideNode w; w.child[0] = root; w.n = 1;
        out.nodes.assign(16, 0.0f);
        emit_node(B.nodes, C, w, out.nodes.data());
        out.num_nodes = 1; out.depth = 1; out.stack_need = 0;
    }
    else
    {
        Collapser C(B.nodes);
        const int hroot = C.height(root);
        if (hroot > max_depth) return false;
        C.collapse(root, max_depth);
        // second pass for the links: wide_of[] of every child is known only after the whole collapse
        out.num_nodes = (int32_t)C.wide.size();
        out.nodes.assign((size_t)out.num_nodes * 16, 0.0f);
        for (int32_t k = 0; k < out.num_nodes; k++) emit_node(B.nodes, C, C.wide[k], out.nodes.data() + (size_t)k * 16);
        out.depth = C.depth; out.stack_need = C.stack_need;
    }
    out.order = std::move(B.order);
    out.pad = pad;
    return out.stack_need <= max_depth;
}

}  // namespace ptk
