// Device BVH builder for gfx950: see bvh_device.h.
//
// Level-synchronous binned SAH (the same cost model, 16 bins and leaf rule as the host builder, bvh_build.cpp):
//   * triangles are NOT moved while the tree grows: every triangle carries the id of the node it currently belongs to;
//   * per level one pass over the triangles moves each one to its child (by the bin of its centroid on the parent's split
//     axis) and bins it into the child's histogram - 3 axes x 16 bins x (count + box), integer atomics on order-preserving
//     encodings of the floats - and one pass over the level's nodes sweeps the histograms, takes the cheapest split (or
//     closes the node as a leaf) and allocates the children, so that the nodes of a level are contiguous;
//   * leaves then get their triangle ranges by a prefix sum, triangles are scattered into leaf order (sorted inside a leaf,
//     so the layout is deterministic), and the binary tree is collapsed top-down, level by level, into the 4-wide nodes
//     with 8-bit quantised child boxes of ptk_device.h under the same traversal-stack budget rule as the host's collapse.
// The host reads one counter per level (next level's size); everything else stays on the device.
// Child centroid bounds are not tracked per bin: a child inherits the parent's centroid bounds cut at the split plane and
// clipped to its own box (a superset), so trees differ slightly from the host's; closest hits do not depend on the tree.
#include "bvh_device.h"
#include "bvh_build.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ptk_device.h"

namespace ptk {
namespace {

#ifndef PTK_BVH_BINS
#define PTK_BVH_BINS 16
#endif
constexpr int kBins = PTK_BVH_BINS;
constexpr int kHistWords = 3 * kBins * 7;            // per node: [axis][bin] { count, enc(mn.xyz), enc(mx.xyz) }
constexpr int kThreads = 256;

enum : int32_t { BN_LEAF = -1, BN_OPEN = -2 };
enum { CNT_NEXT_FREE = 0, CNT_FAIL = 1, CNT_STACK = 2, CNT_QUEUE = 3, CNT_EXT = 4, CNT_ROOT = 8 /* 12 words: enc mn, mx, cmn, cmx */, CNT_WORDS = 32 };

struct BNode {                       // 64 bytes
    float mn[3]; int32_t left;       // >= 0: interior, children left and left + 1;  BN_LEAF;  BN_OPEN: not decided yet
    float mx[3]; int32_t count;      // triangles below (-1: unknown until binned - children of a parity split)
    float cmn[3]; int32_t axis_bin;  // interior: split axis | bin << 2 (left child: bins <= bin); axis 3 = parity split
    float cmx[3]; int32_t first;     // leaf: first position in the leaf order
};

__device__ __forceinline__ uint32_t enc(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float dec(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__device__ __forceinline__ float half_area(const float* mn, const float* mx)
{
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (!(dx >= 0.0f)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}
// interior levels a balanced median-split subtree over c triangles needs (bvh_build.cpp Builder::need)
__device__ __forceinline__ int need_levels(long long c, int leaf_max) { int lv = 0; while (c > leaf_max) { c = (c + 1) / 2; lv++; } return lv; }

// ---- triangle boxes, scene extent, root bounds -------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void tri_bounds_kernel(const float* __restrict__ verts, int n, float4* __restrict__ boxes, uint32_t* counters)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY }, ext = 0.0f;
    if (i < n)
    {
        const float* p = verts + (size_t)i * 9;
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) { const float v = p[k * 3 + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
        boxes[(size_t)i * 2] = make_float4(mn[0], mn[1], mn[2], 0.0f);
        boxes[(size_t)i * 2 + 1] = make_float4(mx[0], mx[1], mx[2], 0.0f);
        for (int a = 0; a < 3; a++) ext = fmaxf(ext, fmaxf(fabsf(mn[a]), fabsf(mx[a])));
    }
    float cmn[3], cmx[3];
    for (int a = 0; a < 3; a++) { const float c = 0.5f * (mn[a] + mx[a]); cmn[a] = i < n ? c : INFINITY; cmx[a] = i < n ? c : -INFINITY; }
    // wave reduction, then one set of atomics per wave
    for (int off = 32; off > 0; off >>= 1)
    {
        for (int a = 0; a < 3; a++)
        {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
            cmn[a] = fminf(cmn[a], __shfl_xor(cmn[a], off)); cmx[a] = fmaxf(cmx[a], __shfl_xor(cmx[a], off));
        }
        ext = fmaxf(ext, __shfl_xor(ext, off));
    }
    if ((threadIdx.x & 63) == 0 && mn[0] <= mx[0])
    {
        for (int a = 0; a < 3; a++)
        {
            atomicMin(&counters[CNT_ROOT + a], enc(mn[a])); atomicMax(&counters[CNT_ROOT + 3 + a], enc(mx[a]));
            atomicMin(&counters[CNT_ROOT + 6 + a], enc(cmn[a])); atomicMax(&counters[CNT_ROOT + 9 + a], enc(cmx[a]));
        }
        atomicMax(&counters[CNT_EXT], __float_as_uint(ext));         // non-negative floats order as unsigned integers
    }
}

__global__ void init_kernel(uint32_t* counters, int32_t* node_of, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) node_of[i] = 0;
    if (i < CNT_WORDS)
    {
        uint32_t v = 0u;
        if (i >= CNT_ROOT && i < CNT_ROOT + 12) v = ((i - CNT_ROOT) / 3) % 2 == 0 ? 0xffffffffu : 0u;    // mins start high, maxes low
        counters[i] = v;
    }
}

__global__ void root_kernel(BNode* nodes, uint32_t* counters, int n)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    BNode r;
    for (int a = 0; a < 3; a++)
    {
        r.mn[a] = dec(counters[CNT_ROOT + a]); r.mx[a] = dec(counters[CNT_ROOT + 3 + a]);
        r.cmn[a] = dec(counters[CNT_ROOT + 6 + a]); r.cmx[a] = dec(counters[CNT_ROOT + 9 + a]);
    }
    r.left = BN_OPEN; r.count = n; r.axis_bin = 0; r.first = 0;
    nodes[0] = r;
    counters[CNT_NEXT_FREE] = 1u;
}

__global__ __launch_bounds__(kThreads) void hist_init_kernel(uint32_t* hist, size_t words)
{
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= words) return;
    const int w = (int)(i % 7);
    hist[i] = w == 0 ? 0u : (w <= 3 ? 0xffffffffu : 0u);
}

__device__ __forceinline__ int bin_of(float c, float lo, float hi)
{
    if (!(hi > lo)) return 0;
    const float scale = (float)kBins / (hi - lo);
    int k = (int)((c - lo) * scale);
    return min(max(k, 0), kBins - 1);
}

// One pass over the triangles per level: move each triangle of a node that was split at the previous level to its child,
// then bin it into that child's histogram if the child is still open.  node_of[i] < 0: the triangle sits in leaf ~node_of[i].
__global__ __launch_bounds__(kThreads) void level_bin_kernel(const float4* __restrict__ boxes, int32_t* __restrict__ node_of, const BNode* __restrict__ nodes,
                                                             uint32_t* __restrict__ hist, int n, int lvl_start, int level)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    int nd = node_of[i];
    if (nd < 0) return;
    const float4 b0 = boxes[(size_t)i * 2], b1 = boxes[(size_t)i * 2 + 1];
    const float c[3] = { 0.5f * (b0.x + b1.x), 0.5f * (b0.y + b1.y), 0.5f * (b0.z + b1.z) };
    const int pl = nodes[nd].left;
    if (pl >= 0)
    {
        const int ab = nodes[nd].axis_bin, axis = ab & 3;
        int side;
        if (axis == 3) side = (int)((mix32((uint32_t)i) >> (level & 31)) & 1u);
        else side = bin_of(c[axis], nodes[nd].cmn[axis], nodes[nd].cmx[axis]) > (ab >> 2) ? 1 : 0;
        nd = pl + side;
        node_of[i] = nd;
    }
    const int st = nodes[nd].left;
    if (st == BN_LEAF) { node_of[i] = ~nd; return; }
    if (st != BN_OPEN) return;                       // (cannot happen: an interior node's triangles were moved above)
    uint32_t* h = hist + (size_t)(nd - lvl_start) * kHistWords;
    const float mn[3] = { b0.x, b0.y, b0.z }, mx[3] = { b1.x, b1.y, b1.z };
    for (int a = 0; a < 3; a++)
    {
        const int k = bin_of(c[a], nodes[nd].cmn[a], nodes[nd].cmx[a]);
        uint32_t* w = h + (a * kBins + k) * 7;
        atomicAdd(w, 1u);
        atomicMin(w + 1, enc(mn[0])); atomicMin(w + 2, enc(mn[1])); atomicMin(w + 3, enc(mn[2]));
        atomicMax(w + 4, enc(mx[0])); atomicMax(w + 5, enc(mx[1])); atomicMax(w + 6, enc(mx[2]));
    }
}

// The same pass for the top of the tree, where a level has only a handful of nodes and a million triangles would hammer a
// few hundred histogram words with global atomics (measured on the 1 M-triangle scene: 46 ms for level 1, 29 ms for level 2,
// ... 118 of the build's 134 ms in the first eight levels): every workgroup bins its share of the triangles into a private
// copy of the level's histograms in LDS and then merges the bins it filled into the global ones.
constexpr int kLdsLevelNodes = 96 * 16 / kBins;        // 96 x 1344 B = 126 KiB of the CU's 160 KiB (16 bins)
__global__ __launch_bounds__(1024) void level_bin_lds_kernel(const float4* __restrict__ boxes, int32_t* __restrict__ node_of, const BNode* __restrict__ nodes,
                                                             uint32_t* __restrict__ hist, int n, int lvl_start, int lvl_count, int level)
{
    extern __shared__ uint32_t lh[];
    const int words = lvl_count * kHistWords;
    for (int w = threadIdx.x; w < words; w += 1024) { const int k = w % 7; lh[w] = k == 0 ? 0u : (k <= 3 ? 0xffffffffu : 0u); }
    __syncthreads();
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024)
    {
        int nd = node_of[i];
        if (nd < 0) continue;
        const float4 b0 = boxes[(size_t)i * 2], b1 = boxes[(size_t)i * 2 + 1];
        const float c[3] = { 0.5f * (b0.x + b1.x), 0.5f * (b0.y + b1.y), 0.5f * (b0.z + b1.z) };
        const int pl = nodes[nd].left;
        if (pl >= 0)
        {
            const int ab = nodes[nd].axis_bin, axis = ab & 3;
            int side;
            if (axis == 3) side = (int)((mix32((uint32_t)i) >> (level & 31)) & 1u);
            else side = bin_of(c[axis], nodes[nd].cmn[axis], nodes[nd].cmx[axis]) > (ab >> 2) ? 1 : 0;
            nd = pl + side;
            node_of[i] = nd;
        }
        const int st = nodes[nd].left;
        if (st == BN_LEAF) { node_of[i] = ~nd; continue; }
        if (st != BN_OPEN) continue;
        uint32_t* h = lh + (nd - lvl_start) * kHistWords;
        const float mn[3] = { b0.x, b0.y, b0.z }, mx[3] = { b1.x, b1.y, b1.z };
        for (int a = 0; a < 3; a++)
        {
            const int k = bin_of(c[a], nodes[nd].cmn[a], nodes[nd].cmx[a]);
            uint32_t* w = h + (a * kBins + k) * 7;
            atomicAdd(w, 1u);
            atomicMin(w + 1, enc(mn[0])); atomicMin(w + 2, enc(mn[1])); atomicMin(w + 3, enc(mn[2]));
            atomicMax(w + 4, enc(mx[0])); atomicMax(w + 5, enc(mx[1])); atomicMax(w + 6, enc(mx[2]));
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < words; w += 1024)
    {
        const int k = w % 7;
        if (lh[w - k] == 0u) continue;                  // this workgroup put nothing into the bin
        if (k == 0) atomicAdd(hist + w, lh[w]);
        else if (k <= 3) atomicMin(hist + w, lh[w]);
        else atomicMax(hist + w, lh[w]);
    }
}

struct BinBox { float mn[3], mx[3]; };
__device__ __forceinline__ void box_reset(BinBox& b) { for (int a = 0; a < 3; a++) { b.mn[a] = INFINITY; b.mx[a] = -INFINITY; } }
__device__ __forceinline__ void box_grow(BinBox& b, const uint32_t* w)
{
    for (int a = 0; a < 3; a++) { b.mn[a] = fminf(b.mn[a], dec(w[1 + a])); b.mx[a] = fmaxf(b.mx[a], dec(w[4 + a])); }
}

// One thread per node of the level: sweep its histograms, close it as a leaf or split it (bvh_build.cpp Builder::build).
__global__ __launch_bounds__(kThreads) void level_split_kernel(BNode* __restrict__ nodes, const uint32_t* __restrict__ hist, uint32_t* counters, int lvl_start, int lvl_end,
                                                               int level /* 1-based */, int max_depth, int leaf_max, float trav_cost)
{
    const int id = lvl_start + blockIdx.x * kThreads + threadIdx.x;
    if (id >= lvl_end) return;
    BNode N = nodes[id];
    if (N.left != BN_OPEN) return;
    const uint32_t* h = hist + (size_t)(id - lvl_start) * kHistWords;
    int count = 0;
    for (int k = 0; k < kBins; k++) count += (int)h[k * 7];
    N.count = count;
    auto close_leaf = [&]() { N.left = BN_LEAF; nodes[id] = N; };
    if (count <= 1) { close_leaf(); return; }
    if (level > max_depth) { atomicExch(&counters[CNT_FAIL], 1u); close_leaf(); return; }

    const bool force_median = (level - 1 + need_levels(count, leaf_max)) >= max_depth;       // no slack left: stay balanced
    const float leaf_cost = (float)count;
    float best_cost = INFINITY;
    int best_axis = -1, best_bin = -1;
    const float parent_area = fmaxf(half_area(N.mn, N.mx), 1e-30f);
    if (!force_median)
        for (int axis = 0; axis < 3; axis++)
        {
            if (!(N.cmx[axis] > N.cmn[axis])) continue;
            const uint32_t* ha = h + axis * kBins * 7;
            float right_area[kBins]; int right_cnt[kBins];
            BinBox acc; box_reset(acc); int c = 0;
            for (int k = kBins - 1; k > 0; k--)
            {
                if (ha[k * 7]) box_grow(acc, ha + k * 7);
                c += (int)ha[k * 7];
                right_area[k] = half_area(acc.mn, acc.mx); right_cnt[k] = c;
            }
            box_reset(acc); c = 0;
            for (int k = 0; k < kBins - 1; k++)
            {
                if (ha[k * 7]) box_grow(acc, ha + k * 7);
                c += (int)ha[k * 7];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                const float cost = trav_cost + (half_area(acc.mn, acc.mx) * (float)c + right_area[k + 1] * (float)right_cnt[k + 1]) / parent_area;
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = k; }
            }
        }
    if (!force_median && count <= leaf_max && !(best_cost < leaf_cost)) { close_leaf(); return; }
    int nl = 0;
    if (best_axis >= 0)
    {
        for (int k = 0; k <= best_bin; k++) nl += (int)h[(best_axis * kBins + k) * 7];
        if (level + need_levels(max(nl, count - nl), leaf_max) > max_depth) best_axis = -1;       // would break the bound
    }
    if (best_axis < 0)
    {
        if (count <= leaf_max) { close_leaf(); return; }
        // object median by bins along the longest centroid axis; a node whose centroids share one bin is halved by parity
        int axis = 0;
        const float e0 = N.cmx[0] - N.cmn[0], e1 = N.cmx[1] - N.cmn[1], e2 = N.cmx[2] - N.cmn[2];
        if (e1 > e0 && e1 >= e2) axis = 1; else if (e2 > e0 && e2 > e1) axis = 2;
        int c = 0, best_d = count + 1;
        for (int k = 0; k < kBins - 1; k++)
        {
            c += (int)h[(axis * kBins + k) * 7];
            const int d = abs(2 * c - count);
            if (c > 0 && c < count && d < best_d) { best_d = d; best_axis = axis; best_bin = k; nl = c; }
        }
        if (best_axis >= 0 && level + need_levels(max(nl, count - nl), leaf_max) > max_depth) { atomicExch(&counters[CNT_FAIL], 1u); close_leaf(); return; }
    }
    const int c0 = (int)atomicAdd(&counters[CNT_NEXT_FREE], 2u);
    BNode L, R;
    if (best_axis < 0)
    {
        // parity split: both halves keep the parent's bounds, their counts are known once they have been binned
        L = N; R = N;
        L.count = R.count = -1;
        N.axis_bin = 3;
        if (level + 1 + need_levels((count + 1) / 2 + 64, leaf_max) > max_depth) atomicExch(&counters[CNT_FAIL], 1u);
    }
    else
    {
        const uint32_t* ha = h + best_axis * kBins * 7;
        BinBox bl, br; box_reset(bl); box_reset(br);
        for (int k = 0; k < kBins; k++)
            if (ha[k * 7]) { if (k <= best_bin) box_grow(bl, ha + k * 7); else box_grow(br, ha + k * 7); }
        const float lo = N.cmn[best_axis], hi = N.cmx[best_axis];
        const float plane = lo + (float)(best_bin + 1) * ((hi - lo) / (float)kBins);
        for (int a = 0; a < 3; a++)
        {
            L.mn[a] = bl.mn[a]; L.mx[a] = bl.mx[a]; R.mn[a] = br.mn[a]; R.mx[a] = br.mx[a];
            // a centroid lies inside its triangle's box, so inside the child's box
            L.cmn[a] = fmaxf(N.cmn[a], bl.mn[a]); L.cmx[a] = fminf(N.cmx[a], bl.mx[a]);
            R.cmn[a] = fmaxf(N.cmn[a], br.mn[a]); R.cmx[a] = fminf(N.cmx[a], br.mx[a]);
        }
        // ... and on its side of the split plane, up to the rounding of the bin index (bins are clamped, so a centroid a
        // hair outside these bounds still lands in an end bin: the bounds only steer the binning)
        const float slack = (hi - lo) * 1e-5f;
        L.cmx[best_axis] = fminf(L.cmx[best_axis], plane + slack);
        R.cmn[best_axis] = fmaxf(R.cmn[best_axis], plane - slack);
        L.count = nl; R.count = count - nl;
        N.axis_bin = best_axis | (best_bin << 2);
    }
    L.left = L.count == 1 ? BN_LEAF : BN_OPEN; R.left = R.count == 1 ? BN_LEAF : BN_OPEN;
    L.axis_bin = R.axis_bin = 0; L.first = R.first = 0;
    nodes[c0] = L; nodes[c0 + 1] = R;
    N.left = c0;
    nodes[id] = N;
}

// triangles whose node was closed at the last level are marked too
__global__ __launch_bounds__(kThreads) void finalise_kernel(int32_t* node_of, const BNode* nodes, int n, uint32_t* counters)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int nd = node_of[i];
    if (nd < 0) return;
    if (nodes[nd].left == BN_LEAF) node_of[i] = ~nd; else atomicExch(&counters[CNT_FAIL], 1u);
}

// ---- exclusive prefix sum of the leaf sizes (three small kernels) ------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_blocks_kernel(const BNode* nodes, int num, int32_t* first, int32_t* block_sum)
{
    __shared__ int32_t s[1024];
    const int t = threadIdx.x, i = blockIdx.x * 1024 + t;
    const int32_t v = (i < num && nodes[i].left == BN_LEAF) ? nodes[i].count : 0;
    s[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1)
    {
        const int32_t add = t >= off ? s[t - off] : 0;
        __syncthreads();
        s[t] += add;
        __syncthreads();
    }
    if (i < num) first[i] = s[t] - v;
    if (t == 1023) block_sum[blockIdx.x] = s[t];
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(int32_t* block_sum, int blocks)
{
    __shared__ int32_t s[1024];
    __shared__ int32_t carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < blocks; base += 1024)
    {
        const int i = base + t;
        const int32_t v = i < blocks ? block_sum[i] : 0;
        s[t] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1)
        {
            const int32_t add = t >= off ? s[t - off] : 0;
            __syncthreads();
            s[t] += add;
            __syncthreads();
        }
        if (i < blocks) block_sum[i] = carry + s[t] - v;
        __syncthreads();
        if (t == 1023) carry += s[t];
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void scan_add_kernel(BNode* nodes, int num, const int32_t* first, const int32_t* block_sum, int32_t* fill)
{
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i >= num) return;
    nodes[i].first = first[i] + block_sum[blockIdx.x];
    fill[i] = 0;
}

__global__ __launch_bounds__(kThreads) void scatter_kernel(const int32_t* node_of, const BNode* nodes, int32_t* fill, int32_t* order, int n)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int leaf = ~node_of[i];
    order[nodes[leaf].first + atomicAdd(&fill[leaf], 1)] = i;
}
// ascending triangle index inside a leaf: the layout does not depend on the order the atomics landed in
__global__ __launch_bounds__(kThreads) void leaf_sort_kernel(const BNode* nodes, int num, int32_t* order)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= num || nodes[i].left != BN_LEAF) return;
    int32_t* o = order + nodes[i].first;
    const int c = nodes[i].count;
    for (int a = 1; a < c; a++)
    {
        const int32_t v = o[a];
        int b = a - 1;
        while (b >= 0 && o[b] > v) { o[b + 1] = o[b]; b--; }
        o[b + 1] = v;
    }
}

// binary heights, deepest level first
__global__ __launch_bounds__(kThreads) void height_kernel(const BNode* nodes, int32_t* hb, int lvl_start, int lvl_end)
{
    const int id = lvl_start + blockIdx.x * kThreads + threadIdx.x;
    if (id >= lvl_end) return;
    const int l = nodes[id].left;
    hb[id] = l < 0 ? 0 : 1 + max(hb[l], hb[l + 1]);
}

// ---- BVH2 -> BVH4 collapse, one wide level per launch (bvh_build.cpp Collapser::collapse + emit_node) -----------------
struct WideItem { int32_t bnode, budget, used, pad; };

__global__ __launch_bounds__(kThreads) void collapse_kernel(const BNode* __restrict__ nodes, const int32_t* __restrict__ hb, const WideItem* __restrict__ queue, int count,
                                                            WideItem* __restrict__ next_queue, uint32_t* counters, float4* __restrict__ out_nodes, int wide_start,
                                                            int next_start, float pad, int leaf_max)
{
    const int q = blockIdx.x * kThreads + threadIdx.x;
    if (q >= count) return;
    const WideItem it = queue[q];
    int child[4]; int nc = 2;
    child[0] = nodes[it.bnode].left; child[1] = child[0] + 1;
    for (;;)
    {
        if (nc == 4) break;
        int best = -1; float best_area = -1.0f;
        const int rem = it.budget - nc;                 // budget of the children once this node holds nc + 1 of them
        for (int k = 0; k < nc; k++)
        {
            const int c = child[k];
            const int cl = nodes[c].left;
            if (cl < 0) continue;
            bool ok = hb[cl] <= rem && hb[cl + 1] <= rem;
            for (int j = 0; j < nc && ok; j++) if (j != k && hb[child[j]] > rem) ok = false;
            if (!ok) continue;
            const float a = half_area(nodes[c].mn, nodes[c].mx);
            if (a > best_area) { best_area = a; best = k; }
        }
        if (best < 0) break;
        const int c = child[best];
        child[best] = nodes[c].left;
        child[nc++] = nodes[c].left + 1;
    }
    // a parity split (identical centroids) may leave one side without triangles: such a leaf is no child at all
    {
        int m = 0;
        for (int k = 0; k < nc; k++)
            if (!(nodes[child[k]].left == BN_LEAF && nodes[child[k]].count <= 0)) child[m++] = child[k];
        nc = m;
    }
    atomicMax(&counters[CNT_STACK], (uint32_t)(it.used + max(nc, 1) - 1));
    // padded child boxes (the host pads every triangle box by `pad` before building; min / max commute with that)
    float cmn[4][3], cmx[4][3], umn[3] = { INFINITY, INFINITY, INFINITY }, umx[3] = { -INFINITY, -INFINITY, -INFINITY };
    int32_t link[4];
    // the interior children of one node get CONSECUTIVE numbers (one counter bump per node, not per child): a ray that enters
    // two of them finds the second record in the 128-byte line the first brought in (or in its neighbour)
    int n_interior = 0;
    for (int k = 0; k < nc; k++) if (nodes[child[k]].left >= 0) n_interior++;
    int pos = n_interior > 0 ? (int)atomicAdd(&counters[CNT_QUEUE], (uint32_t)n_interior) : 0;
    for (int k = 0; k < 4; k++)
    {
        link[k] = NODE_EXIT;
        if (k >= nc) continue;
        const BNode& c = nodes[child[k]];
        for (int a = 0; a < 3; a++)
        {
            cmn[k][a] = c.mn[a] - pad; cmx[k][a] = c.mx[a] + pad;
            umn[a] = fminf(umn[a], cmn[k][a]); umx[a] = fmaxf(umx[a], cmx[k][a]);
        }
        if (c.left < 0) link[k] = ~((c.first << 3) | (c.count - 1));
        else
        {
            WideItem w; w.bnode = child[k]; w.budget = it.budget - (nc - 1); w.used = it.used + (nc - 1); w.pad = 0;
            next_queue[pos] = w;
            link[k] = next_start + pos;
            pos++;
        }
    }
    float scale[3];
    uint32_t lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
    for (int a = 0; a < 3; a++)
    {
        const double ext = (double)umx[a] - (double)umn[a];
        float s = (float)(ext / 255.0 * (1.0 + 1e-6));
        if (!(s > 1e-30f)) s = 1e-30f;
        while ((double)umn[a] + 255.0 * (double)s < (double)umx[a]) s = nextafterf(s, INFINITY);
        scale[a] = s;
        for (int k = 0; k < 4; k++)
        {
            if (k >= nc) { lo[a] |= 255u << (8 * k); continue; }          // empty slot: inverted box
            const double o = umn[a], sd = s;
            int ql = (int)floor(((double)cmn[k][a] - o) / sd), qh = (int)ceil(((double)cmx[k][a] - o) / sd);
            ql = min(max(ql, 0), 255); qh = min(max(qh, 0), 255);
            while (ql > 0 && o + ql * sd > (double)cmn[k][a]) ql--;
            while (qh < 255 && o + qh * sd < (double)cmx[k][a]) qh++;
            lo[a] |= (uint32_t)ql << (8 * k); hi[a] |= (uint32_t)qh << (8 * k);
        }
    }
    float4* o4 = out_nodes + (size_t)(wide_start + q) * NODE_F4;
    o4[0] = make_float4(umn[0], umn[1], umn[2], scale[0]);
    o4[1] = make_float4(scale[1], scale[2], __int_as_float(link[0]), __int_as_float(link[1]));
    o4[2] = make_float4(__int_as_float(link[2]), __int_as_float(link[3]), __uint_as_float(lo[0]), __uint_as_float(lo[1]));
    o4[3] = make_float4(__uint_as_float(lo[2]), __uint_as_float(hi[0]), __uint_as_float(hi[1]), __uint_as_float(hi[2]));
}

inline unsigned blocks_for(long long n, int per) { return (unsigned)std::max<long long>(1, (n + per - 1) / per); }

}  // namespace

#define DCHK(expr)                                                                                       \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) { if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_); goto fail; } \
    } while (0)

bool build_bvh_device(const float* d_verts, int32_t n, int max_stack, int leaf_max, hipStream_t stream, DeviceBvh& out, std::string* err)
{
    out = DeviceBvh();
    if (n < 2 || (int64_t)n >= (1ll << 27)) { if (err) *err = "triangle count outside the device builder's range"; return false; }
    leaf_max = std::min(std::max(leaf_max, 1), 8);
    float trav_cost = 1.0f;             // SAH: one node visit in units of one triangle test
    if (g_bvh_tuning.trav_cost > 0.0f) trav_cost = g_bvh_tuning.trav_cost;                          // ptk_set_option "bvh_trav_cost"
    const auto t0 = std::chrono::steady_clock::now();
    const bool verbose = g_bvh_tuning.verbose != 0;                     // ptk_set_option "bvh_verbose": phase times on stderr
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };

    const size_t node_cap = (size_t)2 * n + 2;
    const size_t hist_nodes = (size_t)n + 2;                     // histograms are indexed by a node's position in its level: a level holds <= n nodes
    float4* d_boxes = nullptr; int32_t* d_node_of = nullptr; BNode* d_bnodes = nullptr; uint32_t* d_hist = nullptr; uint32_t* d_cnt = nullptr;
    int32_t *d_hb = nullptr, *d_first = nullptr, *d_fill = nullptr, *d_blocksum = nullptr, *d_order = nullptr;
    WideItem* d_queue[2] = { nullptr, nullptr };
    float4* d_wide = nullptr; float4* d_final = nullptr;
    std::vector<int> lvl_start;
    uint32_t h_cnt[CNT_WORDS];
    int num_bnodes = 0, wide_total = 0, wide_levels = 0;
    unsigned lds_blocks = 256;
    float pad = 0.0f;
    auto t1 = t0;

    DCHK(hipMalloc(&d_boxes, (size_t)n * 2 * sizeof(float4)));
    DCHK(hipMalloc(&d_node_of, (size_t)n * sizeof(int32_t)));
    DCHK(hipMalloc(&d_bnodes, node_cap * sizeof(BNode)));
    DCHK(hipMalloc(&d_hist, hist_nodes * kHistWords * sizeof(uint32_t)));
    DCHK(hipMalloc(&d_cnt, CNT_WORDS * sizeof(uint32_t)));
    DCHK(hipMalloc(&d_order, (size_t)n * sizeof(int32_t)));

    if (verbose) std::fprintf(stderr, "[bvh_device] n=%d alloc %.2f ms\n", n, ms_since(t0));
    hipLaunchKernelGGL(init_kernel, dim3(blocks_for(std::max<int>(n, CNT_WORDS), kThreads)), dim3(kThreads), 0, stream, d_cnt, d_node_of, n);
    hipLaunchKernelGGL(tri_bounds_kernel, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, stream, d_verts, n, d_boxes, d_cnt);
    hipLaunchKernelGGL(root_kernel, dim3(1), dim3(64), 0, stream, d_bnodes, d_cnt, n);
    DCHK(hipGetLastError());

    // ---- levels --------------------------------------------------------------------------------------------------------
    // (the LDS variant needs up to 126 KiB of dynamic LDS: one 1024-thread workgroup per CU)
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) lds_blocks = (unsigned)cus;
        DCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(level_bin_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLevelNodes * kHistWords * (int)sizeof(uint32_t)));
    }
    lvl_start.push_back(0);
    {
        int lvl_end = 1;
        for (int level = 1; level <= max_stack + 1; level++)
        {
            const int ls = lvl_start.back(), count = lvl_end - ls;
            if ((size_t)count > hist_nodes) { if (err) *err = "level wider than the histogram store"; goto fail; }
            hipLaunchKernelGGL(hist_init_kernel, dim3(blocks_for((long long)count * kHistWords, kThreads)), dim3(kThreads), 0, stream, d_hist, (size_t)count * kHistWords);
            if (count <= kLdsLevelNodes)
                hipLaunchKernelGGL(level_bin_lds_kernel, dim3(std::min<unsigned>(lds_blocks, blocks_for(n, 1024))), dim3(1024), (size_t)count * kHistWords * sizeof(uint32_t), stream,
                                   d_boxes, d_node_of, d_bnodes, d_hist, n, ls, count, level);
            else
                hipLaunchKernelGGL(level_bin_kernel, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, stream, d_boxes, d_node_of, d_bnodes, d_hist, n, ls, level);
            hipLaunchKernelGGL(level_split_kernel, dim3(blocks_for(count, kThreads)), dim3(kThreads), 0, stream, d_bnodes, d_hist, d_cnt, ls, lvl_end, level, max_stack, leaf_max, trav_cost);
            DCHK(hipGetLastError());
            DCHK(hipMemcpyAsync(h_cnt, d_cnt, CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            DCHK(hipStreamSynchronize(stream));
            if (h_cnt[CNT_FAIL]) { if (err) *err = "device build cannot meet the stack bound"; goto fail; }
            const int next_end = (int)h_cnt[CNT_NEXT_FREE];
            if (verbose) std::fprintf(stderr, "[bvh_device] level %d: %d nodes -> %d children, t=%.2f ms\n", level, count, next_end - lvl_end, ms_since(t0));
            if ((size_t)next_end > node_cap) { if (err) *err = "node store overflow"; goto fail; }
            lvl_start.push_back(lvl_end);
            if (next_end == lvl_end) break;             // nothing was split: the level just closed is the last
            lvl_end = next_end;
            if (level == max_stack + 1) { if (err) *err = "device build exceeded the depth bound"; goto fail; }
        }
        num_bnodes = (int)h_cnt[CNT_NEXT_FREE];
    }
    {
        uint32_t ext_bits = h_cnt[CNT_EXT];
        float ext; std::memcpy(&ext, &ext_bits, 4);
        if (!(ext >= 1.0f)) ext = 1.0f;
        if (!std::isfinite(ext)) ext = 1.0f;
        pad = 1e-5f * ext;                               // as bvh_build.cpp: conservative w.r.t. the float rounding of the tests
    }
    // (one more pass moves the triangles of the nodes split at the last level that had splits, and marks every leaf)
    hipLaunchKernelGGL(level_bin_kernel, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, stream, d_boxes, d_node_of, d_bnodes, d_hist, n, 0x3fffffff, 0);
    hipLaunchKernelGGL(finalise_kernel, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, stream, d_node_of, d_bnodes, n, d_cnt);
    DCHK(hipGetLastError());

    // ---- leaf ranges, leaf order, heights -------------------------------------------------------------------------------
    {
        const int sblocks = (int)blocks_for(num_bnodes, 1024);
        DCHK(hipMalloc(&d_hb, (size_t)num_bnodes * sizeof(int32_t)));
        DCHK(hipMalloc(&d_first, (size_t)num_bnodes * sizeof(int32_t)));
        DCHK(hipMalloc(&d_fill, (size_t)num_bnodes * sizeof(int32_t)));
        DCHK(hipMalloc(&d_blocksum, (size_t)sblocks * sizeof(int32_t)));
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(sblocks), dim3(1024), 0, stream, d_bnodes, num_bnodes, d_first, d_blocksum);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, stream, d_blocksum, sblocks);
        hipLaunchKernelGGL(scan_add_kernel, dim3(sblocks), dim3(1024), 0, stream, d_bnodes, num_bnodes, d_first, d_blocksum, d_fill);
        hipLaunchKernelGGL(scatter_kernel, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, stream, d_node_of, d_bnodes, d_fill, d_order, n);
        hipLaunchKernelGGL(leaf_sort_kernel, dim3(blocks_for(num_bnodes, kThreads)), dim3(kThreads), 0, stream, d_bnodes, num_bnodes, d_order);
        for (int l = (int)lvl_start.size() - 2; l >= 0; l--)
        {
            const int ls = lvl_start[l], le = l + 1 < (int)lvl_start.size() - 1 ? lvl_start[l + 1] : num_bnodes;
            if (le > ls) hipLaunchKernelGGL(height_kernel, dim3(blocks_for(le - ls, kThreads)), dim3(kThreads), 0, stream, d_bnodes, d_hb, ls, le);
        }
        DCHK(hipGetLastError());
    }
    if (verbose) { (void)hipStreamSynchronize(stream); std::fprintf(stderr, "[bvh_device] leaves + heights done t=%.2f ms (%d binary nodes)\n", ms_since(t0), num_bnodes); }
    out.ms_levels = ms_since(t0);
    t1 = std::chrono::steady_clock::now();

    // ---- collapse to 4-wide quantised nodes, one wide level per launch ----------------------------------------------------
    {
        int32_t hroot = 0;
        DCHK(hipMemcpyAsync(&hroot, d_hb, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        DCHK(hipMemcpyAsync(h_cnt, d_cnt, CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        DCHK(hipStreamSynchronize(stream));
        if (h_cnt[CNT_FAIL]) { if (err) *err = "device build left triangles outside the leaves"; goto fail; }
        if (hroot > max_stack || hroot < 1) { if (err) *err = "binary tree deeper than the stack bound"; goto fail; }
        const size_t wide_cap = (size_t)n + 2;
        DCHK(hipMalloc(&d_queue[0], wide_cap * sizeof(WideItem)));
        DCHK(hipMalloc(&d_queue[1], wide_cap * sizeof(WideItem)));
        DCHK(hipMalloc(&d_wide, wide_cap * NODE_F4 * sizeof(float4)));
        WideItem root; root.bnode = 0; root.budget = max_stack; root.used = 0; root.pad = 0;
        DCHK(hipMemcpyAsync(d_queue[0], &root, sizeof(root), hipMemcpyHostToDevice, stream));
        int count = 1, start = 0, cur = 0;
        while (count > 0)
        {
            if ((size_t)(start + count) > wide_cap) { if (err) *err = "wide node store overflow"; goto fail; }
            DCHK(hipMemsetAsync(d_cnt + CNT_QUEUE, 0, sizeof(uint32_t), stream));
            hipLaunchKernelGGL(collapse_kernel, dim3(blocks_for(count, kThreads)), dim3(kThreads), 0, stream, d_bnodes, d_hb, d_queue[cur], count, d_queue[cur ^ 1], d_cnt,
                               d_wide, start, start + count, pad, leaf_max);
            DCHK(hipGetLastError());
            DCHK(hipMemcpyAsync(h_cnt, d_cnt, CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            DCHK(hipStreamSynchronize(stream));
            start += count; count = (int)h_cnt[CNT_QUEUE]; cur ^= 1; wide_levels++;
            if (wide_levels > 64) { if (err) *err = "collapse did not terminate"; goto fail; }
        }
        wide_total = start;
        if ((int)h_cnt[CNT_STACK] > max_stack) { if (err) *err = "collapsed tree exceeds the stack bound"; goto fail; }
        DCHK(hipMalloc(&d_final, (size_t)wide_total * NODE_F4 * sizeof(float4)));
        DCHK(hipMemcpyAsync(d_final, d_wide, (size_t)wide_total * NODE_F4 * sizeof(float4), hipMemcpyDeviceToDevice, stream));
        DCHK(hipStreamSynchronize(stream));
    }
    out.d_nodes = d_final; out.d_order = d_order;
    out.num_nodes = wide_total; out.depth = wide_levels; out.stack_need = (int)h_cnt[CNT_STACK]; out.pad = pad;
    out.ms_collapse = ms_since(t1);
    if (verbose) std::fprintf(stderr, "[bvh_device] collapse %.2f ms: %d wide nodes, %d levels, stack %d; total %.2f ms\n", out.ms_collapse, wide_total, wide_levels, out.stack_need, ms_since(t0));
    (void)hipFree(d_boxes); (void)hipFree(d_node_of); (void)hipFree(d_bnodes); (void)hipFree(d_hist); (void)hipFree(d_cnt);
    (void)hipFree(d_hb); (void)hipFree(d_first); (void)hipFree(d_fill); (void)hipFree(d_blocksum);
    (void)hipFree(d_queue[0]); (void)hipFree(d_queue[1]); (void)hipFree(d_wide);
    return true;

fail:
    (void)hipStreamSynchronize(stream);
    (void)hipFree(d_boxes); (void)hipFree(d_node_of); (void)hipFree(d_bnodes); (void)hipFree(d_hist); (void)hipFree(d_cnt);
    (void)hipFree(d_hb); (void)hipFree(d_first); (void)hipFree(d_fill); (void)hipFree(d_blocksum); (void)hipFree(d_order);
    (void)hipFree(d_queue[0]); (void)hipFree(d_queue[1]); (void)hipFree(d_wide); (void)hipFree(d_final);
    out = DeviceBvh();
    return false;
}


// ---- device-side packing of the triangle and shading records (ptk_device.h) from the boundary's flat arrays -----------
namespace {
__global__ __launch_bounds__(kThreads) void pack_tris_kernel(const float* __restrict__ verts, const int32_t* __restrict__ order, const int32_t* __restrict__ material,
                                                             const int32_t* __restrict__ mat_opacity_tex, float4* __restrict__ tris, int n)
{
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= n) return;
    const int i = order[k];
    const float* v = verts + (size_t)i * 9;
    // edge1 = v2 - v1, edge2 = v3 - v1: the very floats IntersectTriangle recomputes per call (pathtracer.cpp:382-383)
    float4* q = tris + (size_t)k * TRI_F4;
    q[0] = make_float4(v[0], v[1], v[2], v[3] - v[0]);
    q[1] = make_float4(v[4] - v[1], v[5] - v[2], v[6] - v[0], v[7] - v[1]);
    q[2] = make_float4(v[8] - v[2], __int_as_float(i), __int_as_float(mat_opacity_tex[material[i]]), 0.0f);
}
__global__ __launch_bounds__(kThreads) void pack_shade_kernel(const float* __restrict__ normals, const float* __restrict__ uvs, const float* __restrict__ tbn,
                                                              const uint8_t* __restrict__ smoothing, const int32_t* __restrict__ material, float4* __restrict__ shade, int n)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float* nn = normals + (size_t)i * 9; const float* uv = uvs + (size_t)i * 6; const float* tb = tbn + (size_t)i * 9;
    float4* q = shade + (size_t)i * SHADE_F4;
    q[0] = make_float4(tb[0], tb[1], tb[2], __int_as_float((int32_t)((uint32_t)material[i] | (smoothing[i] ? 0x80000000u : 0u))));
    q[1] = make_float4(uv[0], uv[1], uv[2], uv[3]);
    q[2] = make_float4(uv[4], uv[5], nn[0], nn[1]);
    q[3] = make_float4(nn[2], nn[3], nn[4], nn[5]);
    q[4] = make_float4(nn[6], nn[7], nn[8], tb[3]);
    q[5] = make_float4(tb[4], tb[5], tb[6], tb[7]);
    q[6] = make_float4(tb[8], 0.0f, 0.0f, 0.0f);
}
}  // namespace

void launch_pack_tris(const float* d_verts, const int32_t* d_order, const int32_t* d_material, const int32_t* d_mat_opacity_tex, float4* d_tris, int n, hipStream_t stream)
{
    if (n > 0) hipLaunchKernelGGL(pack_tris_kernel, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, stream, d_verts, d_order, d_material, d_mat_opacity_tex, d_tris, n);
}
void launch_pack_shade(const float* d_normals, const float* d_uvs, const float* d_tbn, const uint8_t* d_smoothing, const int32_t* d_material, float4* d_shade, int n,
                       hipStream_t stream)
{
    if (n > 0) hipLaunchKernelGGL(pack_shade_kernel, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, stream, d_normals, d_uvs, d_tbn, d_smoothing, d_material, d_shade, n);
}

}  // namespace ptk
