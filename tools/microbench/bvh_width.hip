// Traversal microbenchmark for DESIGN.md 5(g): the product's 4-wide quantised BVH (64-B nodes, nearest child first, the other
// hit children pushed one by one) against an 8-wide compressed tree (80-B nodes: bf16 scales, consecutive children, per-node
// triangle base, a hit MASK per node, the rest of a node's hit children pushed as ONE 64-bit group entry, children visited in
// slot order or reverse by the ray's sign on the node's sort axis) on the SAME incoherent closest-hit rays, outside the
// megakernel: one ray per lane, while-while (a leaf's triangles are tested as soon as the leaf is reached), one-wave
// workgroups with an 8 KiB LDS stack each - 16 waves per CU, as trace_kernel.  Both trees are written by
// tools/bvh_width_probe.py from the tree the product built (ptk_download_bvh); Moeller-Trumbore and the closest-hit rule are
// the product's, so both kernels must return the identical hit for every ray (checked).
//
//   hipcc -O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -o bvh_width bvh_width.hip && ./bvh_width scene.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define EPS 0.00001f
#define NOHIT 0x7fffffff
#define NODE_EXIT INT32_MIN

struct Hit { int tri; float t, u, v; };
struct Ray { float ox, oy, oz, dx, dy, dz; };

__device__ __forceinline__ float rcp_ieee(float a)
{
    const float y = __builtin_amdgcn_rcpf(a);
    const float e = __builtin_fmaf(-a, y, 1.0f);
    return __builtin_fmaf(y, e, y);
}

// the product's tri_test (ptk_kernels.hip) without opacity textures
__device__ __forceinline__ void tri_test(const float4* __restrict__ tris, int i, const Ray& r, Hit& best, unsigned& ntri)
{
    const float4 t0 = tris[i * 3], t1 = tris[i * 3 + 1], t2 = tris[i * 3 + 2];
    ntri++;
    const float v0x = t0.x, v0y = t0.y, v0z = t0.z, e1x = t0.w, e1y = t1.x, e1z = t1.y, e2x = t1.z, e2y = t1.w, e2z = t2.x;
    const float hx = r.dy * e2z - e2y * r.dz, hy = r.dz * e2x - e2z * r.dx, hz = r.dx * e2y - e2x * r.dy;
    const float a = e1x * hx + e1y * hy + e1z * hz;
    const float f = rcp_ieee(a);
    const float sx = r.ox - v0x, sy = r.oy - v0y, sz = r.oz - v0z;
    const float u = f * (sx * hx + sy * hy + sz * hz);
    const float qx = sy * e1z - e1y * sz, qy = sz * e1x - e1z * sx, qz = sx * e1y - e1x * sy;
    const float v = f * (r.dx * qx + r.dy * qy + r.dz * qz);
    const float t = f * (e2x * qx + e2y * qy + e2z * qz);
    const int tri = __float_as_int(t2.y);
    bool ok = !(fabsf(a) < EPS) & !(u < 0.0f) & !(v < 0.0f) & !(u + v > 1.0f) & (t > EPS);
    ok = ok & (t < __builtin_inff()) & ((t < best.t) | ((t == best.t) & (tri < best.tri)));
    best.tri = ok ? tri : best.tri; best.t = ok ? t : best.t; best.u = ok ? u : best.u; best.v = ok ? v : best.v;
}

struct RayConst { float ix, iy, iz, cnx, cny, cnz, cfx, cfy, cfz; unsigned sx, sy, sz; };
__device__ __forceinline__ RayConst ray_const(const Ray& r, float scene_bound)
{
    RayConst c;
    c.ix = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(r.dx), -1e18f, 1e18f);
    c.iy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(r.dy), -1e18f, 1e18f);
    c.iz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(r.dz), -1e18f, 1e18f);
    const float r21 = (fmaxf(fmaxf(fabsf(r.ox), fabsf(r.oy)), fabsf(r.oz)) + scene_bound) * 0x1p-21f;
    const float slx = r21 * fabsf(c.ix), sly = r21 * fabsf(c.iy), slz = r21 * fabsf(c.iz);
    c.cnx = __builtin_fmaf(-r.ox, c.ix, -slx); c.cny = __builtin_fmaf(-r.oy, c.iy, -sly); c.cnz = __builtin_fmaf(-r.oz, c.iz, -slz);
    c.cfx = __builtin_fmaf(-r.ox, c.ix, slx); c.cfy = __builtin_fmaf(-r.oy, c.iy, sly); c.cfz = __builtin_fmaf(-r.oz, c.iz, slz);
    c.sx = (unsigned)(__float_as_int(c.ix) >> 31); c.sy = (unsigned)(__float_as_int(c.iy) >> 31); c.sz = (unsigned)(__float_as_int(c.iz) >> 31);
    return c;
}

// ---- 4-wide: the product's node arm (walk_step), leaves tested at once ------------------------------------------------------
__global__ __launch_bounds__(64, 4) void trav4(const float4* __restrict__ nodes, const float4* __restrict__ tris, const Ray* __restrict__ rays, int n,
                                               float scene_bound, Hit* __restrict__ out, unsigned long long* counters)
{
    __shared__ int stack_[33 * 64];
    const int lane = threadIdx.x;
    const int i = blockIdx.x * 64 + lane;
    if (i >= n) return;
    const Ray r = rays[i];
    const RayConst c = ray_const(r, scene_bound);
    Hit best = { NOHIT, __builtin_inff(), 0.0f, 0.0f };
    int* top = stack_ + lane; int* const base = top;
    int node = 0;
    unsigned nnode = 0, ntri = 0;
    while (node != NODE_EXIT)
    {
        if (node < 0)
        {
            const int code = ~node;
            const int first = code >> 3, cnt = (code & 7) + 1;
            for (int k = 0; k < cnt; k++) tri_test(tris, first + k, r, best, ntri);
            if (top == base) node = NODE_EXIT; else { top -= 64; node = *top; }
            continue;
        }
        const float4* np = nodes + (size_t)node * 4;
        const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
        nnode++;
        const float Ax = q0.w * c.ix, Ay = q1.x * c.iy, Az = q1.y * c.iz;
        const float Bnx = __builtin_fmaf(q0.x, c.ix, c.cnx), Bny = __builtin_fmaf(q0.y, c.iy, c.cny), Bnz = __builtin_fmaf(q0.z, c.iz, c.cnz);
        const float Bfx = __builtin_fmaf(q0.x, c.ix, c.cfx), Bfy = __builtin_fmaf(q0.y, c.iy, c.cfy), Bfz = __builtin_fmaf(q0.z, c.iz, c.cfz);
        const unsigned lox = __float_as_uint(q2.z), loy = __float_as_uint(q2.w), loz = __float_as_uint(q3.x);
        const unsigned hix = __float_as_uint(q3.y), hiy = __float_as_uint(q3.z), hiz = __float_as_uint(q3.w);
        const unsigned nx = (hix & c.sx) | (lox & ~c.sx), fx = (lox & c.sx) | (hix & ~c.sx);
        const unsigned ny = (hiy & c.sy) | (loy & ~c.sy), fy = (loy & c.sy) | (hiy & ~c.sy);
        const unsigned nz = (hiz & c.sz) | (loz & ~c.sz), fz = (loz & c.sz) | (hiz & ~c.sz);
        const int link[4] = { __float_as_int(q1.z), __float_as_int(q1.w), __float_as_int(q2.x), __float_as_int(q2.y) };
        const float tmax = best.t * 1.0000153f;
        int key[4]; bool hit[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const float tnx = __builtin_fmaf((float)((nx >> (8 * k)) & 255u), Ax, Bnx), tfx = __builtin_fmaf((float)((fx >> (8 * k)) & 255u), Ax, Bfx);
            const float tny = __builtin_fmaf((float)((ny >> (8 * k)) & 255u), Ay, Bny), tfy = __builtin_fmaf((float)((fy >> (8 * k)) & 255u), Ay, Bfy);
            const float tnz = __builtin_fmaf((float)((nz >> (8 * k)) & 255u), Az, Bnz), tfz = __builtin_fmaf((float)((fz >> (8 * k)) & 255u), Az, Bfz);
            const float tn = fmaxf(fmaxf(tnx, tny), tnz), tf = fminf(fminf(tfx, tfy), tfz);
            hit[k] = fmaxf(tn, 0.0f) <= fminf(tf, tmax);
            key[k] = hit[k] ? ((__float_as_int(tn) & ~3) | k) : 0x7fffffff;
        }
        const int kmin = min(min(key[0], key[1]), min(key[2], key[3]));
        const bool o0 = key[0] != kmin, o1 = key[1] != kmin, o2 = key[2] != kmin;
        int next = !o0 ? link[0] : (!o1 ? link[1] : (!o2 ? link[2] : link[3]));
        *top = link[0]; top += (hit[0] & o0) ? 64 : 0;
        *top = link[1]; top += (hit[1] & o1) ? 64 : 0;
        *top = link[2]; top += (hit[2] & o2) ? 64 : 0;
        *top = link[3]; top += (hit[3] & (key[3] != kmin)) ? 64 : 0;
        if (kmin == 0x7fffffff) { if (top == base) next = NODE_EXIT; else { top -= 64; next = *top; } }
        node = next;
    }
    out[i] = best;
    if (counters) { atomicAdd(&counters[0], (unsigned long long)nnode); atomicAdd(&counters[1], (unsigned long long)ntri); }
}

// ---- 8-wide compressed node, 80 B = 5 x float4 ---------------------------------------------------------------------------
//   q0 = (origin.xyz, bits [scale.x bf16 | scale.y bf16 << 16])
//   q1 = (bits [scale.z bf16 | imask << 16 | axis << 24], bits child_base, bits tri_base, bits cnt16: (count - 1) of slot k in bits 2k..2k+1)
//   q2 = (lo.x[0..3], lo.x[4..7], lo.y[0..3], lo.y[4..7])   q3 = (lo.z[0..3], lo.z[4..7], hi.x[0..3], hi.x[4..7])
//   q4 = (hi.y[0..3], hi.y[4..7], hi.z[0..3], hi.z[4..7])    byte k of a word = the plane of slot k (4 + k in the second word); empty slot: lo 255, hi 0
// interior child of slot s = node child_base + popcount(imask & ((1 << s) - 1)); leaf of slot s = records tri_base + sum of the counts of
// the leaf slots below s, count(s) of them.  Stack entry (64 bit): x = child_base, y = mask | imask << 8 | descending << 16.
__device__ __forceinline__ unsigned hits4(unsigned nx, unsigned fx, unsigned ny, unsigned fy, unsigned nz, unsigned fz, float Ax, float Ay, float Az,
                                          float Bnx, float Bny, float Bnz, float Bfx, float Bfy, float Bfz, float tmax)
{
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const float tnx = __builtin_fmaf((float)((nx >> (8 * k)) & 255u), Ax, Bnx), tfx = __builtin_fmaf((float)((fx >> (8 * k)) & 255u), Ax, Bfx);
        const float tny = __builtin_fmaf((float)((ny >> (8 * k)) & 255u), Ay, Bny), tfy = __builtin_fmaf((float)((fy >> (8 * k)) & 255u), Ay, Bfy);
        const float tnz = __builtin_fmaf((float)((nz >> (8 * k)) & 255u), Az, Bnz), tfz = __builtin_fmaf((float)((fz >> (8 * k)) & 255u), Az, Bfz);
        const float tn = fmaxf(fmaxf(tnx, tny), tnz), tf = fminf(fminf(tfx, tfy), tfz);
        m |= (fmaxf(tn, 0.0f) <= fminf(tf, tmax)) ? (1u << k) : 0u;
    }
    return m;
}

__global__ __launch_bounds__(64, 4) void trav8(const float4* __restrict__ nodes, const float4* __restrict__ tris, const Ray* __restrict__ rays, int n,
                                               float scene_bound, Hit* __restrict__ out, unsigned long long* counters)
{
    __shared__ uint2 stack_[16 * 64];
    const int lane = threadIdx.x;
    const int i = blockIdx.x * 64 + lane;
    if (i >= n) return;
    const Ray r = rays[i];
    const RayConst c = ray_const(r, scene_bound);
    const unsigned sgn3 = (c.sx & 1u) | (c.sy & 2u) | (c.sz & 4u);
    Hit best = { NOHIT, __builtin_inff(), 0.0f, 0.0f };
    uint2* top = stack_ + lane; uint2* const base = top;
    unsigned g_base = 0, g_bits = 1u | (1u << 8);           // the root: a group of one interior child, node 0
    unsigned nnode = 0, ntri = 0;
    for (;;)
    {
        if ((g_bits & 255u) == 0u)
        {
            if (top == base) break;
            top -= 64;
            const uint2 e = *top;
            g_base = e.x; g_bits = e.y;
        }
        // the next child of the group: lowest or highest hit slot
        const unsigned mask = g_bits & 255u, imask = (g_bits >> 8) & 255u;
        const int s = (g_bits >> 16) & 1u ? 31 - __clz((int)mask) : __ffs((int)mask) - 1;
        g_bits &= ~(1u << s);
        const int node = (int)g_base + __popc(imask & ((1u << s) - 1u));
        if (g_bits & 255u) { *top = make_uint2(g_base, g_bits); top += 64; }
        const float4* np = (const float4*)((const char*)nodes + (size_t)node * 80);
        const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3], q4 = np[4];
        nnode++;
        const unsigned w0 = __float_as_uint(q0.w), w1 = __float_as_uint(q1.x);
        const float scx = __uint_as_float(w0 << 16), scy = __uint_as_float(w0 & 0xffff0000u), scz = __uint_as_float(w1 << 16);
        const unsigned im = (w1 >> 16) & 255u, axis = (w1 >> 24) & 3u;
        const unsigned child_base = __float_as_uint(q1.y), tri_base = __float_as_uint(q1.z), cnt16 = __float_as_uint(q1.w);
        const float Ax = scx * c.ix, Ay = scy * c.iy, Az = scz * c.iz;
        const float Bnx = __builtin_fmaf(q0.x, c.ix, c.cnx), Bny = __builtin_fmaf(q0.y, c.iy, c.cny), Bnz = __builtin_fmaf(q0.z, c.iz, c.cnz);
        const float Bfx = __builtin_fmaf(q0.x, c.ix, c.cfx), Bfy = __builtin_fmaf(q0.y, c.iy, c.cfy), Bfz = __builtin_fmaf(q0.z, c.iz, c.cfz);
        const float tmax = best.t * 1.0000153f;
        const unsigned lx0 = __float_as_uint(q2.x), lx1 = __float_as_uint(q2.y), ly0 = __float_as_uint(q2.z), ly1 = __float_as_uint(q2.w);
        const unsigned lz0 = __float_as_uint(q3.x), lz1 = __float_as_uint(q3.y), hx0 = __float_as_uint(q3.z), hx1 = __float_as_uint(q3.w);
        const unsigned hy0 = __float_as_uint(q4.x), hy1 = __float_as_uint(q4.y), hz0 = __float_as_uint(q4.z), hz1 = __float_as_uint(q4.w);
        unsigned hitmask = hits4((hx0 & c.sx) | (lx0 & ~c.sx), (lx0 & c.sx) | (hx0 & ~c.sx), (hy0 & c.sy) | (ly0 & ~c.sy), (ly0 & c.sy) | (hy0 & ~c.sy),
                                 (hz0 & c.sz) | (lz0 & ~c.sz), (lz0 & c.sz) | (hz0 & ~c.sz), Ax, Ay, Az, Bnx, Bny, Bnz, Bfx, Bfy, Bfz, tmax);
        hitmask |= hits4((hx1 & c.sx) | (lx1 & ~c.sx), (lx1 & c.sx) | (hx1 & ~c.sx), (hy1 & c.sy) | (ly1 & ~c.sy), (ly1 & c.sy) | (hy1 & ~c.sy),
                         (hz1 & c.sz) | (lz1 & ~c.sz), (lz1 & c.sz) | (hz1 & ~c.sz), Ax, Ay, Az, Bnx, Bny, Bnz, Bfx, Bfy, Bfz, tmax) << 4;
        // leaves of this node: tested at once, in slot order
        unsigned hl = hitmask & ~im;
        while (hl)
        {
            const int ls = __ffs((int)hl) - 1;
            hl &= hl - 1u;
            const unsigned below = cnt16 & ((1u << (2 * ls)) - 1u);
            const int first = (int)tri_base + __popc(~im & ((1u << ls) - 1u) & 255u) + __popc(below & 0x5555u) + 2 * __popc(below & 0xaaaau);
            const int cnt = (int)((cnt16 >> (2 * ls)) & 3u) + 1;
            for (int k = 0; k < cnt; k++) tri_test(tris, first + k, r, best, ntri);
        }
        // interior children hit: the new current group (what was left of the old one is on the stack)
        const unsigned hi = hitmask & im;
        const unsigned desc = (sgn3 >> axis) & 1u;          // the ray runs towards -axis: far slots first
        g_base = child_base; g_bits = hi | (im << 8) | (desc << 16);
    }
    out[i] = best;
    if (counters) { atomicAdd(&counters[0], (unsigned long long)nnode); atomicAdd(&counters[1], (unsigned long long)ntri); }
}

template <class T>
static T* upload(const std::vector<char>& blob, size_t off, size_t bytes)
{
    T* d = nullptr;
    CHECK(hipMalloc(&d, bytes ? bytes : 16));
    CHECK(hipMemcpy(d, blob.data() + off, bytes, hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: bvh_width scene.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> blob((size_t)sz);
    if (fread(blob.data(), 1, (size_t)sz, f) != (size_t)sz) return 2;
    fclose(f);
    // header: 8 x int64: magic, n4 (nodes), n8 (nodes), ntri, nrays, depth8 (stack entries the 8-wide tree needs), 0, 0; then float scene_bound + pad
    const int64_t* h = (const int64_t*)blob.data();
    if (h[0] != 0x3848564257ll) { fprintf(stderr, "bad magic\n"); return 2; }
    const int64_t n4 = h[1], n8 = h[2], ntri = h[3], nrays = h[4], depth8 = h[5];
    float scene_bound; std::memcpy(&scene_bound, blob.data() + 64, 4);
    if (depth8 > 16) { fprintf(stderr, "the 8-wide tree needs %lld stack entries (> 16)\n", (long long)depth8); return 3; }
    size_t off = 128;
    const float4* d_n4 = upload<float4>(blob, off, (size_t)n4 * 64); off += (size_t)n4 * 64;
    const float4* d_t4 = upload<float4>(blob, off, (size_t)ntri * 48); off += (size_t)ntri * 48;
    const float4* d_n8 = upload<float4>(blob, off, (size_t)n8 * 80); off += (size_t)n8 * 80;
    const float4* d_t8 = upload<float4>(blob, off, (size_t)ntri * 48); off += (size_t)ntri * 48;
    const Ray* d_rays = upload<Ray>(blob, off, (size_t)nrays * sizeof(Ray)); off += (size_t)nrays * sizeof(Ray);
    Hit *d_h4 = nullptr, *d_h8 = nullptr; unsigned long long* d_cnt = nullptr;
    CHECK(hipMalloc(&d_h4, (size_t)nrays * sizeof(Hit))); CHECK(hipMalloc(&d_h8, (size_t)nrays * sizeof(Hit))); CHECK(hipMalloc(&d_cnt, 32));
    const int blocks = (int)((nrays + 63) / 64);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best_ms[2] = { 1e30, 1e30 };
    unsigned long long cnt[2][2] = { { 0, 0 }, { 0, 0 } };
    for (int rep = 0; rep < 6; rep++)
        for (int w = 0; w < 2; w++)
        {
            CHECK(hipMemset(d_cnt, 0, 32));
            unsigned long long* cp = rep == 0 ? d_cnt : nullptr;          // (counters in the first, untimed repetition only)
            CHECK(hipEventRecord(e0));
            if (w == 0) hipLaunchKernelGGL(trav4, dim3(blocks), dim3(64), 0, 0, d_n4, d_t4, d_rays, (int)nrays, scene_bound, d_h4, cp);
            else hipLaunchKernelGGL(trav8, dim3(blocks), dim3(64), 0, 0, d_n8, d_t8, d_rays, (int)nrays, scene_bound, d_h8, cp);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 0) CHECK(hipMemcpy(cnt[w], d_cnt, 16, hipMemcpyDeviceToHost));
            else if (ms < best_ms[w]) best_ms[w] = ms;
        }
    std::vector<Hit> h4((size_t)nrays), h8((size_t)nrays);
    CHECK(hipMemcpy(h4.data(), d_h4, (size_t)nrays * sizeof(Hit), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h8.data(), d_h8, (size_t)nrays * sizeof(Hit), hipMemcpyDeviceToHost));
    long long differ = 0, hits = 0;
    for (int64_t i = 0; i < nrays; i++)
    {
        if (std::memcmp(&h4[(size_t)i], &h8[(size_t)i], sizeof(Hit)) != 0) differ++;
        if (h4[(size_t)i].tri != NOHIT) hits++;
    }
    printf("{\"rays\": %lld, \"hit_fraction\": %.4f, \"hits_differ\": %lld, \"bvh4\": {\"nodes\": %lld, \"ms\": %.4f, \"Mrays_per_s\": %.1f, \"node_visits_per_ray\": %.3f, "
           "\"tri_tests_per_ray\": %.3f}, \"bvh8\": {\"nodes\": %lld, \"ms\": %.4f, \"Mrays_per_s\": %.1f, \"node_visits_per_ray\": %.3f, \"tri_tests_per_ray\": %.3f, "
           "\"stack_need\": %lld}, \"bvh8_over_bvh4\": %.4f}\n",
           (long long)nrays, (double)hits / (double)nrays, differ, (long long)n4, best_ms[0], nrays / best_ms[0] / 1e3, (double)cnt[0][0] / nrays, (double)cnt[0][1] / nrays,
           (long long)n8, best_ms[1], nrays / best_ms[1] / 1e3, (double)cnt[1][0] / nrays, (double)cnt[1][1] / nrays, (long long)depth8, best_ms[0] / best_ms[1]);
    return differ ? 1 : 0;
}
