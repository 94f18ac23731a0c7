// HIP kernels for gfx950 (CDNA4, wave64): the per-pixel render loop of the reference
//   PathTracer::RenderFrame -> Trace -> Hit -> {IntersectTriangle, Image::tex2D, DirectIllumimation}
//   (reference PathTracing/src/pathtracer.cpp:367-822, mesh.cpp:48-59, image.cpp:63-86)
// as a path-tracing kernel of persistent waves (trace_kernel: one path per lane) followed by a streaming
// accumulate_kernel.
//
// Design (not a translation of the reference's recursion):
//   * Trace is iterative: L += T*e; L += T*direct; T *= weight, with the reference's two counters
//     (depth arms Russian roulette, iter is the hard stop) and its quirks kept.
//   * Work item = (8x8 pixel quadrant, chunk of samples); work unit = (live pixel of it, sample): parallel over
//     pixels AND samples, so a frame with few non-trivial pixels still fills 256 CUs.  Units are dealt to
//     whichever lane needs work; the waves are persistent and pull items from per-XCD queues built over a
//     device-side list of the quadrants that have live pixels (live_mask_kernel / live_compact_kernel), so all
//     64 lanes of every wave trace until the launch runs dry.  Each finished path stores its radiance as one
//     float4 into a sample buffer in HBM; accumulate_kernel then folds the samples of a pixel into the float
//     accumulator strictly in sample order (the reference's one-add-per-RenderFrame semantics,
//     pathtracer.cpp:798-800) and writes the RGB8 resolve.
//   * A lane is a state machine {NEED, GEN, TRAV, SHADE, DONE}.  Every wave iteration takes a 64-bit ballot per
//     state and runs one block: the BVH walk keeps stepping (node arm every iteration, triangle arm when enough
//     lanes hold a leaf; bounce and shadow rays share the walk; a finished shadow ray rolls straight into the
//     bounce ray) until the lane-iterations wasted by lanes parked for shading / a camera ray outweigh the
//     lanes that block would leave idle.  Divergent blocks therefore run with full-ish EXEC masks instead of
//     once per ray.  Scenes of <= 16 triangles skip the hierarchy (FLAT: scalar triangle loads, shadow and
//     bounce ray in one pass).
//   * Closest hit: 4-wide BVH, one 64-byte record per node holding the four child boxes quantised OUTWARD to 8 bits on
//     the node's own grid (t = fma(q, A, B) per plane), nearest child first, the others on a per-lane stack in LDS laid
//     out [level][lane] (conflict-free ds_read/write_b32); the node record is requested before the triangle arm runs.
//     Result = min over accepted triangles with an order-independent tie rule, so it does not depend on the tree (the
//     reference's own tree is random, mesh.cpp:171-172).  Box tests are acceleration only, so they use v_rcp and fused
//     multiply-adds - but they are CONSERVATIVE for any ray origin and triangle size (explicit slack for the slab
//     arithmetic's absolute error and for Moeller-Trumbore's own error in t, see walk_step); everything that reaches the
//     image (Moeller-Trumbore, shading, samplers) is IEEE and in the reference's order, with 1/x and sqrt as short
//     sequences proven bit-identical by enumeration (rcp_ieee, sqrt_ieee).
//   * Shadow rays keep the reference's closest-hit + identity test (pathtracer.cpp:522-526): the light triangle is
//     tested first (its record comes with the light sample), then any hit the closest-hit rule accepts is nearer and
//     ends the walk (Walk::occl_tri).
//   * Pinhole cameras without opacity textures: the camera ray's hit is cached per pixel, pixels whose camera ray
//     misses are never traced, and a new path starts directly in the SHADE block (PTK_FUSED_START).
//   * RNG: PCG-RXS-M-XS-32 per path, keyed on (seed, pixel, sample) - never on lane/block/GPU.
//
// Float arithmetic is written operation by operation in the reference's order and this file is
// compiled with -ffp-contract=off: results are reproducible against the CPU oracle bit for bit.

#include "ptk_device.h"

// This file is compiled TWICE into libptk.so.  PTK_CONTRACT 0 (default): -ffp-contract=off, the bit-exact product kernels in
// namespace ptk.  PTK_CONTRACT >= 1: the same trace kernels in namespace ptk::fma, built with -ffp-contract=fast (the
// compiler fuses a * b + c into v_fma_f32 wherever it appears: Moeller-Trumbore, dot products, normalisations, shading) for
// the "contract" option of ptk_set_option: results within the north star's tolerance (RMSE <= 1e-3 per channel on the mean
// image, asserted by tests/test_gpu_contract.py), no longer bit-identical to the oracle.  Level 2 also takes 1 / x, sqrt
// and 1 / sqrt straight from the hardware's 1-ulp instructions.
#ifndef PTK_CONTRACT
#define PTK_CONTRACT 0
#endif

namespace ptk {
#if PTK_CONTRACT >= 2
namespace fast {
#elif PTK_CONTRACT
namespace fma {
#endif

#define PTK_EPS 0.00001f                        // mesh.h:12
#define PTK_FLT_EPSILON 1.1920928955078125e-7f
#define PTK_PI_D 3.14159265358979323846
#define PTK_BLOCK 256
#ifndef PTK_TRACE_BLOCK
#define PTK_TRACE_BLOCK 64          // trace_kernel: one wave per workgroup -> finest-grained dispatch
#endif
#ifndef PTK_TRACE_WAVES
#define PTK_TRACE_WAVES 5           // waves per SIMD the register allocator must allow, FLAT variant: 96 VGPRs (4 spilled) since the parameters
                                    // are read through the constant address space; C2 +3.4-4.5 %, C1 +4 % over four waves (six: 80 VGPRs, 37 spilled, -6 %)
#endif
#ifndef PTK_TRACE_WAVES_BVH
#define PTK_TRACE_WAVES_BVH 4       // ... BVH variant
#endif
#define PTK_NOHIT 0x7fffffff
#ifndef PTK_GEN_CACHED_FAST
#define PTK_GEN_CACHED_FAST 1        // cached camera hits: the camera-ray block only loads the pixel's direction and hit
#endif

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 mulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 muls(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
// glm 0.9.3.1 dot / cross / normalize / reflect (include/glm/core/func_geometric.inl:161-283)
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 x, v3 y)
{
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
// 1.0f / a, BIT FOR BIT the IEEE-754 round-to-nearest quotient the oracle and the reference compute, for every float with
// 2^-126 <= |a| <= 2^126: v_rcp_f32 (1 ulp) and one Newton step with an exact residual.  Proven by enumeration - all 2^32
// bit patterns on this GPU, tools/microbench/exact_math.hip, profiles/r02/exact_math.json: 0 mismatches in that range - and
// 13 issue cycles instead of the 43 of the compiler's v_div_scale / v_div_fmas / v_div_fixup expansion (which exists for the
// denormal ranges).  The range cannot be left by a triangle's determinant or a vector's length while scene coordinates stay
// below 2^61 in magnitude, which ptk_upload_scene enforces.
#ifndef PTK_SHORT_RCP
#define PTK_SHORT_RCP 1
#endif
__device__ __forceinline__ float rcp_ieee(float a)
{
#if PTK_CONTRACT >= 2
    return __builtin_amdgcn_rcpf(a);
#elif PTK_SHORT_RCP
    const float y = __builtin_amdgcn_rcpf(a);
    const float e = __builtin_fmaf(-a, y, 1.0f);
    return __builtin_fmaf(y, e, y);
#else
    return 1.0f / a;
#endif
}
// ... plus IEEE results for zeros, infinities and NaNs (one v_div_fixup_f32): where a zero length can occur
__device__ __forceinline__ float rcp_ieee_any(float a)
{
#if PTK_CONTRACT >= 2
    return __builtin_amdgcn_rcpf(a);
#elif PTK_SHORT_RCP
    return __builtin_amdgcn_div_fixupf(rcp_ieee(a), a, 1.0f);
#else
    return 1.0f / a;
#endif
}
// sqrtf(x), BIT FOR BIT the correctly rounded IEEE-754 root: v_sqrt_f32 (1 ulp) and the exact residuals (fma) of its two
// neighbours - the core of the compiler's own expansion without its range scaling (x < 2^-96 is multiplied by 2^32 first)
// and special-case selects.  Enumerated over all 2^32 bit patterns (tools/microbench/exact_math.hip, profiles/r02/
// exact_math.json): identical to sqrtf for +0, +inf and every x >= 2^-104 (the largest input that differs is 0x0b6e9372,
// where the residuals underflow); anything below - a positive length under 2^-52, which no scene produces, and negative
// or NaN arguments - takes the compiler's expansion behind a branch that is practically never taken.
#ifndef PTK_NODE_PREFETCH
#define PTK_NODE_PREFETCH 1
#endif
#ifndef PTK_FLAT_SHARED_ORIGIN
#define PTK_FLAT_SHARED_ORIGIN 1    // FLAT pass: the origin-only part of Moeller-Trumbore once per lane, not once per ray
#endif
#ifndef PTK_FLAT_EXEC_UPDATE
#define PTK_FLAT_EXEC_UPDATE 1     // FLAT pass: the closest hit is updated with exec-masked moves behind a branch, not four selects per ray (C2 +1 %; the same in the BVH walk's tri_test: +-0)
#endif
#ifndef PTK_HIT_CLAMP
#define PTK_HIT_CLAMP 1            // node arm: max(entry, 0) <= min(exit, closest hit) - one compare per child (round 4: C4 +1.5 %)
#endif
#ifndef PTK_PUSH_BRANCHLESS
#define PTK_PUSH_BRANCHLESS 1      // node arm: unconditional stack writes, conditional pointer bumps (round 4: C4 +2 %; with the above: 28 -> 11 scalar instructions per node)
#endif
#ifndef PTK_TRI_PER_EXEC
#define PTK_TRI_PER_EXEC 2          // triangles one execution of walk_step's (voted) triangle arm tests per lane
#endif
#ifndef PTK_FUSED_START
#define PTK_FUSED_START 1
#endif
#ifndef PTK_ROBUST_BOXES
#define PTK_ROBUST_BOXES 1
#endif
#ifndef PTK_SHORT_SQRT
#define PTK_SHORT_SQRT 1
#endif
__device__ __forceinline__ float sqrt_ieee(float x)
{
#if PTK_CONTRACT >= 2
    return __builtin_amdgcn_sqrtf(x);
#elif PTK_SHORT_SQRT
    if (__builtin_expect(!(x >= 0x1p-104f), 0)) return sqrtf(x);          // (zero too: correct either way, and as rare)
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = rm <= 0.0f ? sm : s;
    return rp > 0.0f ? sp : s;
#else
    return sqrtf(x);
#endif
}
__device__ __forceinline__ v3 normalize(v3 a)
{
    float sqr = a.x * a.x + a.y * a.y + a.z * a.z;
#if PTK_CONTRACT >= 2
    float inv = __builtin_amdgcn_rsqf(sqr);
#else
    float inv = rcp_ieee_any(sqrt_ieee(sqr));
#endif
    return muls(a, inv);
}
__device__ __forceinline__ v3 reflect(v3 I, v3 N)
{
    float d = dot(N, I);
    return sub(I, muls(muls(N, d), 2.0f));
}

// sin/cos on [0, 2*pi]: fixed polynomial shared (by construction, not by source) with the oracle
__device__ __forceinline__ void sincos_2pi(float a, float& s, float& c)
{
    int k = (int)(a * 0.636619772367581343f + 0.5f);
    float r = (float)((double)a - (double)k * 1.57079632679489661923);
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    int q = k & 3;
    float ss = (q & 1) ? cp : sp;
    float cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = (q == 1 || q == 2) ? -cc : cc;
}

// ---- RNG (replaces PathTracer::Rand, pathtracer.cpp:367-371) -----------------------------------
__device__ __forceinline__ uint32_t pcg_out(uint32_t st)
{
    uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
    return (w >> 22u) ^ w;
}
__device__ __forceinline__ uint32_t hash32(uint32_t x) { return pcg_out(x * 747796405u + 2891336453u); }
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }

struct Rng {
    uint32_t state, inc, key;
    __device__ __forceinline__ float next()
    {
        uint32_t old = state;
        state = old * 747796405u + inc;
        return u01(pcg_out(old));
    }
    __device__ __forceinline__ float opacity(uint32_t ray, uint32_t tri) const
    {
        return u01(hash32(tri + hash32(ray + key)));
    }
};

__device__ __forceinline__ float4 ldg4(const float4* p) { return *p; }
typedef float f2 __attribute__((ext_vector_type(2)));
// (float)byte / 255.0f, exactly: the double product rounds to the same float for all 256 bytes
// (checked exhaustively in tests/test_host_cpu.py); saves the IEEE division sequence
__device__ __forceinline__ float unorm8(uint32_t b) { return (float)((double)b * (1.0 / 255.0)); }

// ---- Image::tex2D (image.cpp:63-86), nearest + repeat, RGBA8 atlas -----------------------------------
template <class PT>
__device__ __forceinline__ float4 tex2d(const PT& P, int tex, float uvx, float uvy)
{
    int4 ti = P.texinfo[tex];
    float u = uvx - truncf(uvx);              // == fmodf(uvx, 1.0f), exact
    float v = uvy - truncf(uvy);
    if (u < 0.0f) u += 1.0f;
    if (v < 0.0f) v += 1.0f;
    int cx = (int)((float)ti.x * u);
    int cy = (int)((float)ti.y * v);
    cx = min(cx, ti.x - 1); cy = min(cy, ti.y - 1);
    cx = max(cx, 0); cy = max(cy, 0);
    uint32_t w = P.texels[(size_t)ti.z + (size_t)cy * (size_t)ti.x + (size_t)cx];
    float4 r;
    r.x = unorm8(w & 255u);
    r.y = unorm8((w >> 8) & 255u);
    r.z = unorm8((w >> 16) & 255u);
    r.w = unorm8(w >> 24);
    return r;
}
template <class PT>
__device__ __forceinline__ float tex2d_r(const PT& P, int tex, float uvx, float uvy)
{
    int4 ti = P.texinfo[tex];
    float u = uvx - truncf(uvx);
    float v = uvy - truncf(uvy);
    if (u < 0.0f) u += 1.0f;
    if (v < 0.0f) v += 1.0f;
    int cx = (int)((float)ti.x * u);
    int cy = (int)((float)ti.y * v);
    cx = min(cx, ti.x - 1); cy = min(cy, ti.y - 1);
    cx = max(cx, 0); cy = max(cy, 0);
    uint32_t w = P.texels[(size_t)ti.z + (size_t)cy * (size_t)ti.x + (size_t)cx];
    return unorm8(w & 255u);
}

struct Hit { int tri; float t, u, v; };

struct Counters { uint32_t rays, shadow, nodes, tris, shaded, tex, walk_iters, walk_lanes, shade_execs, shade_lanes, gen_execs, gen_lanes, tri_execs, tri_lanes, cur_nodes, max_nodes, started; };

// ---- closest hit (replaces the recursive PathTracer::Hit, pathtracer.cpp:411-492) ---------------------
// The walk is re-entrant: all of its state lives in this struct so a wave can interleave BVH steps
// with shading of other lanes.  stack: this thread's column of the block's LDS stack, element k at
// stack[k * PTK_BLOCK].
struct Walk {
    v3 ro, rd, inv;
    v3 cn, cf;                   // per axis: -(ro * inv + slack) and slack - ro * inv, the constant terms of a node's near / far slab
                                 // distances for this ray; slack = what the slab arithmetic can be off by for any node (walk_step)
    uint32_t sgnx, sgny, sgnz;   // per axis: all ones when the ray travels towards -axis (selects the near / far plane bytes with one v_bfi each)
    int node;
    int* top;                    // this lane's stack top in LDS (== its column's base when empty); unused by the FLAT kernel
    int tri_next, tri_left;      // pending leaf: records [tri_next, tri_next + tri_left) still to test
    Hit best;
    // occl_tri >= 0 marks a shadow ray towards light triangle occl_tri.  DirectIllumimation's test (pathtracer.cpp:522-526) is
    // "the closest hit along the ray is the light triangle (or nothing)".  The light triangle is tested FIRST, before the walk
    // (its record comes with the light sample), so `best` already holds its hit - if the ray hits it at all - and any other
    // triangle the walk then accepts is, by the closest-hit rule, nearer: it decides the test and ends the walk.  Order
    // independent by construction.  (Round 1 ended the walk on any hit nearer than 0.9999 x the distance to the light SAMPLE:
    // wrong when Moeller-Trumbore places a grazing hit on the light triangle itself nearer than that - found by
    // tools/soak_random_scenes.py, one pixel-sample in 19 of 3000 random scenes.)
    int occl_tri;

    // node: >= 0 interior node to test next; NODE_EXIT nothing left on the node side; any other negative
    // value = a leaf waiting for the triangle queue (tri_next, tri_left) to drain
    __device__ __forceinline__ void begin(v3 o, v3 d, int num_nodes, int* stack, float scene_bound)
    {
        ro = o; rd = d;
        // acceleration only: 1-ulp reciprocals are fine for conservative slab tests.  Clamped to +-1e18 so that a ray
        // parallel to an axis (a zero component: a hemisphere sample with w == 0 about an axis-aligned normal, one path in
        // 2^24) keeps FINITE slab distances of the right sign - with +-inf the quantised form q * (scale * inv) + (origin -
        // ro) * inv turns into NaNs on that axis, the axis stops culling and such a ray walks the whole tree (measured:
        // 228 153 node visits for one ray of the 1 M-triangle scene, a 0.5 s tail per launch)
        inv = V(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -1e18f, 1e18f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -1e18f, 1e18f),
                __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -1e18f, 1e18f));
#if PTK_ROBUST_BOXES
        // The slab arithmetic of walk_step, t = fma(q, A, B) with A = scale * inv and B = (origin - ro) * inv, is off by at most
        // 2^-21 (|B| + 256 |A|) (see there).  Every node origin lies inside the scene's padded bounds and a node's 255 grid
        // steps span at most the scene, so per axis that is at most 2^-21 (max |ro| + 3.1 scene_bound) |inv| - a property of the RAY,
        // computed here once instead of twelve instructions per node visited.  (In position units 5e-7 x the scene's size:
        // nothing next to a node's own extent until rays come from ~10^5 scene sizes away, where it is exactly what is needed.)
        const float r21 = (fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + scene_bound) * 0x1p-21f;      // (one bound for the three axes)
        const v3 slack = V(r21 * fabsf(inv.x), r21 * fabsf(inv.y), r21 * fabsf(inv.z));
        // ... and folded, with the ray's own share of B, into the constant of ONE fma per plane family and axis:
        //   B -+ slack = origin * inv - ro * inv -+ slack = fma(origin, inv, cn | cf),   cn = fma(-ro, inv, -slack), cf = fma(-ro, inv, slack)
        // (origin * inv - ro * inv instead of (origin - ro) * inv: the cancellation costs 2^-24 (|origin| + |ro|) |inv| at most,
        // which the bound above was derived with - |origin - ro| <= |origin| + |ro| - so it is covered)
        cn = V(__builtin_fmaf(-o.x, inv.x, -slack.x), __builtin_fmaf(-o.y, inv.y, -slack.y), __builtin_fmaf(-o.z, inv.z, -slack.z));
        cf = V(__builtin_fmaf(-o.x, inv.x, slack.x), __builtin_fmaf(-o.y, inv.y, slack.y), __builtin_fmaf(-o.z, inv.z, slack.z));
#else
        cn = V(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z)); cf = cn;
#endif
        sgnx = (uint32_t)(__float_as_int(inv.x) >> 31); sgny = (uint32_t)(__float_as_int(inv.y) >> 31); sgnz = (uint32_t)(__float_as_int(inv.z) >> 31);
        node = num_nodes > 0 ? 0 : NODE_EXIT;
        top = stack;
        tri_next = 0; tri_left = 0;
        best.tri = PTK_NOHIT; best.t = __builtin_inff(); best.u = 0.0f; best.v = 0.0f;
    }
    __device__ __forceinline__ bool done() const { return node == NODE_EXIT && tri_left == 0; }
    template <int STRIDE>
    __device__ __forceinline__ int pop(const int* stack)
    {
        if (top == stack) return NODE_EXIT;
        top -= STRIDE;
        return *top;
    }
};

// Candidate test of one triangle record: Hit's leaf branch (pathtracer.cpp:463-489) = Moeller-Trumbore
// + order-independent closest rule + stochastic opacity.  Returns true when the walk can stop (an
// occluder decided a shadow ray).
template <bool STATS, class PT>
__device__ __forceinline__ bool tri_test(const PT& P, Walk& W, float4 t0, float4 t1, float4 t2, const Rng& rng,
                                         uint32_t ray, Counters& cnt)
{
    const v3 ro = W.ro, rd = W.rd;
    if (STATS) cnt.tris++;
    // Moeller-Trumbore, PathTracer::IntersectTriangle pathtracer.cpp:373-409.  The reference returns
    // early after each rejection test; here every quantity is computed and the SAME tests (in their
    // negated form, so NaNs fall through exactly as they do there) are AND-ed: identical results for
    // every accepted hit, no divergent branches in the hot loop.
    v3 v0 = V(t0.x, t0.y, t0.z);
    v3 edge1 = V(t0.w, t1.x, t1.y);
    v3 edge2 = V(t1.z, t1.w, t2.x);
    v3 h = cross(rd, edge2);
    float a = dot(edge1, h);
    float f = rcp_ieee(a);                      // (|a| < EPS, a NaN or infinite: rejected below whatever f is)
    v3 s = sub(ro, v0);
    float u = f * dot(s, h);
    v3 q = cross(s, edge1);
    float v = f * dot(rd, q);
    float t = f * dot(edge2, q);
    int tri = __float_as_int(t2.y);
    // (the reference also returns on u > 1, pathtracer.cpp:393: implied here - v >= 0 makes fl(u + v) >= u, rounding being
    // monotone, so u > 1 fails the u + v test, and a NaN u passes both forms alike)
    bool ok = !(fabsf(a) < PTK_EPS) & !(u < 0.0f) & !(v < 0.0f) & !(u + v > 1.0f) & (t > PTK_EPS);
    // (t < inf: with a ray origin ~1e30 away q overflows, v is NaN, t +inf - the reference rejects that on u > 1 or on a NaN
    // of its own; without the test the tie rule below would take t == best.t == inf for a hit.  ptk_set_camera bounds the
    // camera position, so only a path that has already left every float range could get here)
    ok = ok & (t < __builtin_inff()) & ((t < W.best.t) | ((t == W.best.t) & (tri < W.best.tri)));
    int otex = __float_as_int(t2.z);
    if (ok && otex >= 0)
    {
        // stochastic opacity, pathtracer.cpp:469-476 (GetUV :533-536); rare: skipped with s_cbranch_execz
        const float4* sp4 = P.shade + (size_t)tri * SHADE_F4;
        float4 s1 = ldg4(sp4 + 1), s2 = ldg4(sp4 + 2);
        float w = 1.0f - u - v;
        float ux = w * s1.x + u * s1.z + v * s2.x;
        float uy = w * s1.y + u * s1.w + v * s2.y;
        float op = tex2d_r(P, otex, ux, uy);
        if (STATS) cnt.tex++;
        ok = rng.opacity(ray, (uint32_t)tri) < op;
    }
    W.best.tri = ok ? tri : W.best.tri;
    W.best.t = ok ? t : W.best.t;
    W.best.u = ok ? u : W.best.u;
    W.best.v = ok ? v : W.best.v;
    return ok & (W.occl_tri >= 0) & (tri != W.occl_tri);
}

// FLAT pass, both rays of a lane against one triangle in PACKED f32: x = the bounce ray (W), y = the shadow ray (WS).
// Every Moeller-Trumbore operation is the same IEEE mul / add / sub as in tri_test, on two values at once
// (v_pk_mul_f32 / v_pk_add_f32 run at the rate of their scalar forms; the triangle's words come from SGPRs), which
// nearly halves the instructions of the hottest loop of the Cornell configs.  No early returns: all quantities are
// computed and the reference's tests AND-ed in their negated form, exactly as tri_test does.  Returns true when the
// shadow ray has been decided by an occluder.
struct RayPair { f2 ox, oy, oz, dx, dy, dz; };

template <bool STATS, class PT>
__device__ __forceinline__ bool tri_test_pair(const PT& P, Walk& W, Walk& WS, const RayPair& R, const bool shadow_live, float4 t0, float4 t1,
                                              float4 t2, const Rng& rng, uint32_t ray_bounce, uint32_t ray_shadow, Counters& cnt)
{
    if (STATS) cnt.tris += shadow_live ? 2u : 1u;
    const float v0x = t0.x, v0y = t0.y, v0z = t0.z, e1x = t0.w, e1y = t1.x, e1z = t1.y, e2x = t1.z, e2y = t1.w, e2z = t2.x;
    // h = cross(rd, edge2)
    const f2 hx = R.dy * e2z - R.dz * e2y, hy = R.dz * e2x - R.dx * e2z, hz = R.dx * e2y - R.dy * e2x;
    const f2 a = hx * e1x + hy * e1y + hz * e1z;                 // dot(edge1, h)
    const f2 f = { rcp_ieee(a.x), rcp_ieee(a.y) };
#if PTK_FLAT_SHARED_ORIGIN
    // The two rays of a lane leave the SAME point (shade_interaction starts both at p), so everything of Moeller-Trumbore that
    // depends on the origin alone - s = ro - v0, q = cross(s, edge1), dot(edge2, q) - is the same number for both: computed once
    // in scalar f32 (full rate) instead of twice in packed f32 (half rate), 17 of the ~60 operations per ray and triangle.  The
    // same IEEE operations on the same inputs: bit-identical.  (No shadow ray: its half of the packed values is ignored anyway.)
    const float sx = W.ro.x - v0x, sy = W.ro.y - v0y, sz = W.ro.z - v0z;      // s = ro - v0
    const f2 u = f * (sx * hx + sy * hy + sz * hz);
    const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;      // q = cross(s, edge1)
    const f2 v = f * (R.dx * qx + R.dy * qy + R.dz * qz);
    const f2 t = f * (qx * e2x + qy * e2y + qz * e2z);
#else
    const f2 sx = R.ox - v0x, sy = R.oy - v0y, sz = R.oz - v0z;  // s = ro - v0
    const f2 u = f * (sx * hx + sy * hy + sz * hz);
    // q = cross(s, edge1)
    const f2 qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
    const f2 v = f * (R.dx * qx + R.dy * qy + R.dz * qz);
    const f2 t = f * (qx * e2x + qy * e2y + qz * e2z);
#endif
    const f2 uv = u + v;
    const int tri = __float_as_int(t2.y);
    const int otex = __float_as_int(t2.z);
    bool okb = !(fabsf(a.x) < PTK_EPS) & !(u.x < 0.0f) & !(v.x < 0.0f) & !(uv.x > 1.0f) & (t.x > PTK_EPS);     // (u > 1: implied, see tri_test)
    bool oks = !(fabsf(a.y) < PTK_EPS) & !(u.y < 0.0f) & !(v.y < 0.0f) & !(uv.y > 1.0f) & (t.y > PTK_EPS);
    // the flat list is in ascending triangle index, so the tie rule's "equal t and smaller index" can never hold for a
    // later record: nearer-than-best is the whole rule
    okb = okb & (t.x < W.best.t);
    oks = oks & shadow_live & (t.y < WS.best.t);
    if ((okb | oks) && otex >= 0)
    {
        // stochastic opacity, pathtracer.cpp:469-476 (GetUV :533-536); rare: skipped with s_cbranch_execz
        const float4* sp4 = P.shade + (size_t)tri * SHADE_F4;
        const float4 s1 = ldg4(sp4 + 1), s2 = ldg4(sp4 + 2);
        if (okb)
        {
            const float w = 1.0f - u.x - v.x;
            const float op = tex2d_r(P, otex, w * s1.x + u.x * s1.z + v.x * s2.x, w * s1.y + u.x * s1.w + v.x * s2.y);
            if (STATS) cnt.tex++;
            okb = rng.opacity(ray_bounce, (uint32_t)tri) < op;
        }
        if (oks)
        {
            const float w = 1.0f - u.y - v.y;
            const float op = tex2d_r(P, otex, w * s1.x + u.y * s1.z + v.y * s2.x, w * s1.y + u.y * s1.w + v.y * s2.y);
            if (STATS) cnt.tex++;
            oks = rng.opacity(ray_shadow, (uint32_t)tri) < op;
        }
    }
#if PTK_FLAT_EXEC_UPDATE
    // (exec-masked moves - full rate - instead of four half-rate selects per ray; skipped outright when no lane accepts)
    if (okb) { W.best.tri = tri; W.best.t = t.x; W.best.u = u.x; W.best.v = v.x; asm volatile("" : "+v"(W.best.t), "+v"(W.best.u), "+v"(W.best.v)); }
    if (oks) { WS.best.tri = tri; WS.best.t = t.y; WS.best.u = u.y; WS.best.v = v.y; asm volatile("" : "+v"(WS.best.t), "+v"(WS.best.u), "+v"(WS.best.v)); }
#else
    W.best.tri = okb ? tri : W.best.tri; W.best.t = okb ? t.x : W.best.t; W.best.u = okb ? u.x : W.best.u; W.best.v = okb ? v.x : W.best.v;
    WS.best.tri = oks ? tri : WS.best.tri; WS.best.t = oks ? t.y : WS.best.t; WS.best.u = oks ? u.y : WS.best.u; WS.best.v = oks ? v.y : WS.best.v;
#endif
    return oks & (tri != WS.occl_tri);
}

// What the walk loop reads of the launch parameters, held in SGPRs for the length of the loop.  The parameters themselves
// live in the constant address space (trace_kernel), where a field is an s_load at its point of use - right for the hundreds of
// fields-times-places outside the hot loop, wrong inside it: the compiler re-issued the loads of the node and triangle pointers
// in EVERY walk iteration and waited for them before the node record could even be requested.  readfirstlane makes the
// values opaque (not re-materialisable as loads).
// (The pointers keep the GLOBAL address space through the integer round trip: a generic pointer would turn every record fetch
// into a flat_load, which is slower and counts against the LDS counter as well.)
#define PTK_GLOBAL __attribute__((address_space(1)))
struct WalkParams {
    const float4* nodes; const float4* tris; const float4* shade;
    const int4* texinfo; const uint32_t* texels;
    int tri_thr, shade_thr, gen_thr;
};
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p)
{
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (T*)(PTK_GLOBAL T*)(uintptr_t)(((uint64_t)hi << 32) | lo);        // integer -> GLOBAL pointer -> generic: the loads stay global_load
}
template <class PT>
__device__ __forceinline__ WalkParams walk_params(const PT& P)
{
    WalkParams w;
    w.nodes = uniform_ptr(P.nodes); w.tris = uniform_ptr(P.tris); w.shade = uniform_ptr(P.shade);
    w.texinfo = uniform_ptr(P.texinfo); w.texels = uniform_ptr(P.texels);
    w.tri_thr = __builtin_amdgcn_readfirstlane(P.tri_thr); w.shade_thr = __builtin_amdgcn_readfirstlane(P.shade_thr);
    w.gen_thr = __builtin_amdgcn_readfirstlane(P.gen_thr);
    return w;
}

// One BVH step of a lane: up to TWO units of work - one triangle of the pending leaf (arm A) AND one
// interior node (arm B).  A leaf reached by arm B is parked in the lane's one-entry triangle queue and the
// descent continues with the next node from the stack, so the two arms overlap instead of alternating
// (the wave executes both arms every iteration anyway).  The price is slightly later t-max tightening;
// the result is unaffected (closest hit is order-independent).
struct NodeRec { float4 q0, q1, q2, q3; };      // one 64-byte node record in flight / in registers
// the record of the node a lane will test next (a lane without a node reads the root - every such lane the same 64 bytes - which
// costs less than a branch around the loads and zeroing sixteen registers for the lanes that skip them)
template <class PT>
__device__ __forceinline__ void request_node(const PT& P, const Walk& W, NodeRec& r)
{
    // (a 32-bit byte offset from the wave-uniform base: the load takes its base from an SGPR pair, no 64-bit address arithmetic)
    const float4* np = (const float4*)((const char*)P.nodes + (uint32_t)max(W.node, 0) * (uint32_t)(NODE_F4 * 16));
    r.q0 = ldg4(np); r.q1 = ldg4(np + 1); r.q2 = ldg4(np + 2); r.q3 = ldg4(np + 3);
}
// PIPELINED: the caller's loop keeps a node record in flight ACROSS iterations - `rec` was requested (request_node) before the
// loop or at the end of the lane's previous step, and the record of the node this step ends on is requested before the step
// returns, so its round trip also covers the loop's wave-uniform bookkeeping (ballots, debts, ~30 dependent scalar instructions)
// instead of starting behind it.
template <bool STATS, int STRIDE, bool PIPELINED = false, int TRI_PER_EXEC = PTK_TRI_PER_EXEC, class PT>
__device__ __forceinline__ void walk_step(const PT& P, Walk& W, const Rng& rng, uint32_t ray, int* stack, Counters& cnt,
                                          const bool run_tri_arm = true, NodeRec* rec = nullptr)
{
#if PTK_NODE_PREFETCH
    // the node record of arm B is requested BEFORE arm A runs, so that its round trip overlaps arm A's loads and arithmetic
    // (one memory latency per iteration instead of two; the compiler would otherwise issue it after arm A's join)
    NodeRec here;
    if (PIPELINED) here = *rec; else request_node(P, W, here);
    const float4 q0 = here.q0, q1 = here.q1, q2 = here.q2, q3 = here.q3;
    asm volatile("" ::: "memory");
#endif
    const bool node_was = W.node >= 0;
    if (TRI_PER_EXEC == 2 && run_tri_arm && W.tri_left > 0)       // ---- arm A: up to TWO triangles
    {
        // The second triangle: the pending leaf's next one, or - the pending leaf has only this one left and the lane is BLOCKED
        // on a second leaf (W.node holds it: the one-leaf queue was busy) - the first triangle of that leaf, whose remainder then
        // becomes the pending leaf while the lane pops its next node.  Both records are requested together and tested one after
        // the other: the same tri_test calls in the same order as one per execution, so results cannot differ.  Leaves hold
        // 1.1-1.5 triangles on average, so what this buys is mostly the blocked leaf - its lane walks on an iteration earlier -
        // and a triangle arm that is voted 44 % less often (round 4: C4 +2 %, C5 +4 %, C3 +4 %; three or four per execution,
        // and the pair in packed f32, measured slower: DESIGN 12).
        const bool two = W.tri_left >= 2;
        const bool blocked = !two & (W.node < 0) & (W.node != NODE_EXIT);
        const int code = ~W.node;
        const int iA = W.tri_next, iB = two ? iA + 1 : (blocked ? (code >> 3) : iA);
        const float4* tpa = (const float4*)((const char*)P.tris + (uint32_t)iA * (uint32_t)(TRI_F4 * 16));
        const float4* tpb = (const float4*)((const char*)P.tris + (uint32_t)iB * (uint32_t)(TRI_F4 * 16));
        float4 a0 = ldg4(tpa), a1 = ldg4(tpa + 1), a2 = ldg4(tpa + 2);
        float4 b0 = ldg4(tpb), b1 = ldg4(tpb + 1), b2 = ldg4(tpb + 2);
        bool stop = tri_test<STATS>(P, W, a0, a1, a2, rng, ray, cnt);
        if ((two | blocked) && !stop) stop = tri_test<STATS>(P, W, b0, b1, b2, rng, ray, cnt);
        if (blocked)
        {
            W.tri_next = (code >> 3) + 1; W.tri_left = code & 7;
            W.node = W.template pop<STRIDE>(stack);
        }
        else { W.tri_next = iA + (two ? 2 : 1); W.tri_left -= two ? 2 : 1; }
        W.top = stop ? stack : W.top;                     // an occluder decides a shadow ray: drop everything
        W.tri_left = stop ? 0 : W.tri_left;
        W.node = stop ? NODE_EXIT : W.node;
    }
    if (TRI_PER_EXEC != 2 && run_tri_arm && W.tri_left > 0)       // ---- arm A: one triangle
    {
        const float4* tp = (const float4*)((const char*)P.tris + (uint32_t)W.tri_next * (uint32_t)(TRI_F4 * 16));
        float4 t0 = ldg4(tp), t1 = ldg4(tp + 1), t2 = ldg4(tp + 2);
        W.tri_next++; W.tri_left--;
        const bool stop = tri_test<STATS>(P, W, t0, t1, t2, rng, ray, cnt);
        W.top = stop ? stack : W.top;                     // an occluder decides a shadow ray: drop everything
        W.tri_left = stop ? 0 : W.tri_left;
        W.node = stop ? NODE_EXIT : W.node;
    }
    if (node_was && W.node >= 0)                          // ---- arm B: one 4-wide interior node (its record is `here`; a node popped by arm A waits a step)
    {
#if !PTK_NODE_PREFETCH
        const float4* np = P.nodes + (size_t)W.node * NODE_F4;
        const float4 q0 = ldg4(np), q1 = ldg4(np + 1), q2 = ldg4(np + 2), q3 = ldg4(np + 3);
#endif
        if (STATS) { cnt.nodes++; cnt.cur_nodes++; }
        // child planes live on the node's 8-bit grid: plane = origin + q * scale, so along the ray
        //   t = (plane - ro) * inv = q * (scale * inv) + (origin - ro) * inv = fma(q, A, B)
        // (box tests are acceleration only - any conservative test gives the same closest hit - so fused
        // multiply-adds and approximate reciprocals are fine here; the grid boxes enclose the padded boxes)
        const float Ax = q0.w * W.inv.x, Ay = q1.x * W.inv.y, Az = q1.y * W.inv.z;
        // CONSERVATIVE for every ray, however far its origin: t = fma(q, A, B) is the sum of two possibly large terms, so its
        // error is absolute - at most 2^-22 (|B| + 255 |A|) from the roundings of the products, the 1-ulp reciprocal and the
        // fmas - i.e. a position error of ~6e-8 x the distance between the ray's origin and the node, which exceeds an 8-bit
        // grid step once that distance is > 65 000 node extents (and Moeller-Trumbore's own decisions carry the same
        // uncertainty, so no padding of the tree can stand in for it).  Near planes are taken that much (x 2) too early and far
        // planes too late; found by tools/soak_bvh.py: two clusters of 1e-3 at +-1e3 gave tree-dependent hits.  The bound is
        // taken per RAY (Walk::begin: |B| <= (|ro| + scene bound) |inv|, 256 |A| <= 2.01 scene bound |inv|) and folded into the
        // ray's constants: six fused multiply-adds per node here (round 2: fifteen instructions).
        const float Bnx = __builtin_fmaf(q0.x, W.inv.x, W.cn.x), Bny = __builtin_fmaf(q0.y, W.inv.y, W.cn.y), Bnz = __builtin_fmaf(q0.z, W.inv.z, W.cn.z);
        const float Bfx = __builtin_fmaf(q0.x, W.inv.x, W.cf.x), Bfy = __builtin_fmaf(q0.y, W.inv.y, W.cf.y), Bfz = __builtin_fmaf(q0.z, W.inv.z, W.cf.z);
        // the ray enters a slab through the low plane when it travels in +axis, through the high plane otherwise:
        // pick the near / far plane bytes of all four children at once by the sign of the direction
        const uint32_t mx = W.sgnx, my = W.sgny, mz = W.sgnz;
        const uint32_t lox = __float_as_uint(q2.z), loy = __float_as_uint(q2.w), loz = __float_as_uint(q3.x);
        const uint32_t hix = __float_as_uint(q3.y), hiy = __float_as_uint(q3.z), hiz = __float_as_uint(q3.w);
        const uint32_t nx = (hix & mx) | (lox & ~mx), fx = (lox & mx) | (hix & ~mx);
        const uint32_t ny = (hiy & my) | (loy & ~my), fy = (loy & my) | (hiy & ~my);
        const uint32_t nz = (hiz & mz) | (loz & ~mz), fz = (loz & mz) | (hiz & ~mz);
        const int link0 = __float_as_int(q1.z), link1 = __float_as_int(q1.w), link2 = __float_as_int(q2.x), link3 = __float_as_int(q2.y);
#if PTK_ROBUST_BOXES
        // ... and a node is only culled against the closest hit so far when it lies beyond it by more than Moeller-Trumbore's
        // own error in t (relative ~1e-7 / cos of the incidence angle: which of two triangles 1e-6 apart is "closest" is
        // decided by that arithmetic, not by geometry - the second half of the same soak finding)
        const float tmax = W.best.t * 1.0000153f;
#else
        const float tmax = W.best.t;
#endif
        int key[4];
        bool hit[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const float tnx = __builtin_fmaf((float)((nx >> (8 * k)) & 255u), Ax, Bnx), tfx = __builtin_fmaf((float)((fx >> (8 * k)) & 255u), Ax, Bfx);
            const float tny = __builtin_fmaf((float)((ny >> (8 * k)) & 255u), Ay, Bny), tfy = __builtin_fmaf((float)((fy >> (8 * k)) & 255u), Ay, Bfy);
            const float tnz = __builtin_fmaf((float)((nz >> (8 * k)) & 255u), Az, Bnz), tfz = __builtin_fmaf((float)((fz >> (8 * k)) & 255u), Az, Bfz);
            // NaNs (0 * inf for axis-parallel rays) drop out of min3 / max3: that axis then does not constrain - conservative
            const float tn = fmaxf(fmaxf(tnx, tny), tnz), tf = fminf(fminf(tfx, tfy), tfz);
#if PTK_HIT_CLAMP
            // entry no earlier than the ray's start, exit no later than the closest hit: ONE compare instead of three (and no
            // scalar ands of three lane masks per child)
            hit[k] = fmaxf(tn, 0.0f) <= fminf(tf, tmax);
#elif PTK_ROBUST_BOXES
            hit[k] = (tn <= tf) & (tf >= 0.0f) & (tn <= tmax);
#else
            hit[k] = (tn <= tf * 1.000001f) & (tf >= 0.0f) & (tn <= tmax);
#endif
            // order key: the entry distance with the slot in its low bits (negative distances - origin inside - sort first)
            key[k] = hit[k] ? ((__float_as_int(tn) & ~3) | k) : 0x7fffffff;
        }
        const int kmin = min(min(key[0], key[1]), min(key[2], key[3]));
        // the nearest child is the one whose key is the minimum (keys of hit children differ in their slot bits); the same
        // four compares decide which of the others wait on the stack (two hits: exactly far-after-near; more: slot order)
        const bool o0 = key[0] != kmin, o1 = key[1] != kmin, o2 = key[2] != kmin;
        int next = !o0 ? link0 : (!o1 ? link1 : (!o2 ? link2 : link3));
#if PTK_PUSH_BRANCHLESS
        // every link is written at the running top and the top moves on only behind a link that stays: no exec-mask juggling around
        // four conditional stores (the stack has one row of slack above the tree's own need: lds_stack is PTK_MAX_BVH_DEPTH + 1 rows)
        *W.top = link0; W.top += (hit[0] & o0) ? STRIDE : 0;
        *W.top = link1; W.top += (hit[1] & o1) ? STRIDE : 0;
        *W.top = link2; W.top += (hit[2] & o2) ? STRIDE : 0;
        *W.top = link3; W.top += (hit[3] & (key[3] != kmin)) ? STRIDE : 0;
#else
        if (hit[0] & o0) { *W.top = link0; W.top += STRIDE; }
        if (hit[1] & o1) { *W.top = link1; W.top += STRIDE; }
        if (hit[2] & o2) { *W.top = link2; W.top += STRIDE; }
        if (hit[3] & (key[3] != kmin)) { *W.top = link3; W.top += STRIDE; }
#endif
        if (kmin == 0x7fffffff) next = W.template pop<STRIDE>(stack);
        W.node = next;
    }
    if (W.node < 0 && W.node != NODE_EXIT && W.tri_left == 0)   // a leaf and the triangle queue is free
    {
        const int code = ~W.node;
        W.tri_next = code >> 3;
        W.tri_left = (code & 7) + 1;
        W.node = W.template pop<STRIDE>(stack);
    }
    if (PIPELINED) request_node(P, W, *rec);                    // for this lane's next step (a finished walk asks for the root: where its next ray starts)
}

// hemisphere / lobe sampler, pathtracer.cpp:606-611 (:618-623 lobe form): see oracle sample_about()
__device__ __forceinline__ v3 sample_about(v3 n_for_test, float thr, v3 basis_from, v3 pole, float w, float theta)
{
    v3 u = fabsf(n_for_test.x) < thr ? cross(V(1.0f, 0.0f, 0.0f), basis_from) : cross(V(1.0f, 1.0f, 1.0f), basis_from);
    u = normalize(u);
    v3 v = normalize(cross(u, basis_from));
    float ang = (float)(2.0f * PTK_PI_D * theta);
    float sn, cs;
    sincos_2pi(ang, sn, cs);
    v3 d = add(add(muls(u, w * cs), muls(v, w * sn)), muls(pole, sqrt_ieee(1.0f - w * w)));
    return normalize(d);
}

__device__ __forceinline__ uint32_t pixel_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t pixel)
{
    uint32_t a = hash32(seed_hi);
    uint32_t b = hash32(seed_lo + a);
    return hash32(pixel + b);
}

// DirectIllumimation's sampling half (pathtracer.cpp:494-521, 527-530; SampleTriangle :494-503): picks a light triangle and a
// point on it from three draws, in the reference's order, and returns false when the surface faces away (:518-520).  Its
// visibility half (:522-526, closest hit along l is the light) is the shadow walk the caller starts: towards `l`, with
// occl_tri = light_tri, after testing the light triangle itself (lt0..lt2, its record) first.  di is the value DirectIllumimation returns when that walk finds the
// light (:530).
template <class PT>
__device__ __forceinline__ bool sample_direct_light(const PT& P, v3 p, v3 n, v3 diffuse, float u_light, float u_su, float u_sv,
                                                    v3& l, v3& di, int& light_tri, float4& lt0, float4& lt1, float4& lt2)
{
    int lightId = (int)floorf(u_light * (float)P.num_lights);
    if (lightId == P.num_lights && lightId > 0) lightId--;
    const float4* lp = P.lights + (size_t)lightId * LIGHT_F4;
    float4 l0 = ldg4(lp), l1 = ldg4(lp + 1), l2 = ldg4(lp + 2), l3 = ldg4(lp + 3);
    float su = sqrt_ieee(u_su);
    float sv = u_sv;
    float w0 = 1.0f - su, w1 = su * (1.0f - sv), w2 = su * sv;
    v3 vLight = add(add(muls(V(l0.x, l0.y, l0.z), w0), muls(V(l1.x, l1.y, l1.z), w1)),
                    muls(V(l2.x, l2.y, l2.z), w2));
    const v3 dl = sub(vLight, p);
    l = normalize(dl);
    float ndl = dot(neg(n), neg(l));
    light_tri = __float_as_int(l0.w);
    // the light triangle's own record, as the walk would fetch it: v0, e1 = v2 - v1, e2 = v3 - v1 (the same subtractions the
    // record packers perform), its index and opacity texture
    lt0 = make_float4(l0.x, l0.y, l0.z, l1.x - l0.x);
    lt1 = make_float4(l1.y - l0.y, l1.z - l0.z, l2.x - l0.x, l2.y - l0.y);
    lt2 = make_float4(l2.z - l0.z, l0.w, l3.y, 0.0f);
    if (ndl <= 0.0f) return false;              // (:519 in its own form: a NaN normal - a normal map on a mesh without uvs - goes ON, as there)
    v3 lColor = V(l1.w, l2.w, l3.x);
    di = muls(mulv(lColor, diffuse), ndl);      // :530
    return true;
}

// One surface interaction of PathTracer::Trace (pathtracer.cpp:551-727) for the hit W.best of the ray (W.ro, W.rd): emission,
// Russian roulette, material branch, direction sampling, the light sample of DirectIllumimation.  Returns true when the path
// ends here; otherwise W holds the next ray to walk (BVH kernels: the shadow ray towards the sampled light - W.occl_tri >= 0,
// W.best = its light triangle's own hit, Tdi its contribution, nextDir the bounce direction that follows - or the bounce
// ray itself; FLAT kernel: W the bounce ray and WS the shadow ray, tested in one pass).  Shared by every trace kernel.
template <bool STATS, bool FLAT, class PT>
__device__ __forceinline__ bool shade_interaction(const PT& P, Walk& W, Walk& WS, int* stack, Rng& rng, v3& L, v3& T, v3& Tdi, v3& nextDir,
                                                  int& depth, int& iter, bool& inside, const uint32_t ray, Counters& cnt)
{
    const Hit h = W.best;
    const v3 ro = W.ro, rd = W.rd;
    if (STATS) cnt.shaded++;
    const float4* sp4 = P.shade + (size_t)h.tri * SHADE_F4;
    float4 s0 = ldg4(sp4);
    int mbits = __float_as_int(s0.w);
    int matid = mbits & 0x7fffffff;
    bool smoothing = mbits < 0;
    const float4* mp = P.mats + (size_t)matid * MAT_F4;
    // the whole 96-byte material in one batch (two of its words are only needed further down: asked for there,
    // they cost the block another memory round trip), and the vertex normals of a smoothed triangle with it
    float4 m0 = ldg4(mp), m1 = ldg4(mp + 1), m2 = ldg4(mp + 2), m3 = ldg4(mp + 3);
    float4 m4f = ldg4(mp + 4), m5f = ldg4(mp + 5);
    float4 sn2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), sn3 = sn2, sn4 = sn2;
    if (smoothing) { sn2 = ldg4(sp4 + 2); sn3 = ldg4(sp4 + 3); sn4 = ldg4(sp4 + 4); }
    asm volatile("" ::: "memory");
    int tex_diffuse = __float_as_int(m4f.x), tex_normal = __float_as_int(m4f.y);
    int tex_emiss = __float_as_int(m4f.z), tex_rough = __float_as_int(m4f.w);
    int tex_metal = __float_as_int(m5f.x);
    bool any_tex = __float_as_int(m5f.z) != 0;

    v3 p = add(ro, muls(rd, h.t));                  // :553
    float uvx = 0.0f, uvy = 0.0f;
    if (any_tex)
    {
        float4 s1 = ldg4(sp4 + 1), s2 = ldg4(sp4 + 2);
        float w = 1.0f - h.u - h.v;                 // GetUV :533-536
        uvx = w * s1.x + h.u * s1.z + h.v * s2.x;
        uvy = w * s1.y + h.u * s1.w + h.v * s2.y;
    }
    v3 n = V(s0.x, s0.y, s0.z);
    if (smoothing)                                  // :556, GetSmoothNormal :538-543
    {
        const float4 s2 = sn2, s3 = sn3, s4 = sn4;
        float w = 1.0f - h.u - h.v;
        v3 n1 = V(s2.z, s2.w, s3.x), n2 = V(s3.y, s3.z, s3.w), n3 = V(s4.x, s4.y, s4.z);
        v3 sn = add(add(muls(n1, w), muls(n2, h.u)), muls(n3, h.v));
        n = normalize(sn);
    }
    if (tex_normal >= 0)                            // :558-566
    {
        float4 s4 = ldg4(sp4 + 4), s5 = ldg4(sp4 + 5), s6 = ldg4(sp4 + 6);
        float4 c = tex2d(P, tex_normal, uvx, uvy);
        if (STATS) cnt.tex++;
        v3 nt = V(c.x * 2.0f - 1.0f, c.y * 2.0f - 1.0f, c.z * 2.0f - 1.0f);
        if (nt.z <= 0.0f) nt = V(nt.x, nt.y, PTK_EPS);
        nt = normalize(nt);
        v3 tg = V(s4.w, s5.x, s5.y), bt = V(s5.z, s5.w, s6.x);
        v3 m = V(tg.x * nt.x + bt.x * nt.y + n.x * nt.z,
                 tg.y * nt.x + bt.y * nt.y + n.y * nt.z,
                 tg.z * nt.x + bt.z * nt.y + n.z * nt.z);
        n = normalize(m);
    }
    if (dot(n, rd) > 0.0f) n = neg(n);              // :567-568
    p = add(p, muls(n, PTK_EPS));                   // :569

    bool ended = false;
    if (!(iter < P.max_depth)) ended = true;        // :571 terminal bounce: no emission
    else
    {
        v3 diffuse = V(m0.x, m0.y, m0.z);
        if (tex_diffuse >= 0) { float4 c = tex2d(P, tex_diffuse, uvx, uvy); diffuse = V(c.x, c.y, c.z); if (STATS) cnt.tex++; }
        v3 emiss = V(m2.x, m2.y, m2.z);
        if (tex_emiss >= 0) { float4 c = tex2d(P, tex_emiss, uvx, uvy); emiss = V(c.x, c.y, c.z); if (STATS) cnt.tex++; }
        float roughness = m2.w;
        if (tex_rough >= 0) { roughness = tex2d_r(P, tex_rough, uvx, uvy); if (STATS) cnt.tex++; }
        float reflectiveness = m3.x;
        if (tex_metal >= 0) { reflectiveness = tex2d_r(P, tex_metal, uvx, uvy); if (STATS) cnt.tex++; }
        const int mtype = __float_as_int(m0.w);
        const v3 specular = V(m1.x, m1.y, m1.z);
        const float emissI = m1.w;

        depth++; iter++;                            // :586-587
        const float prob = m3.w;                    // min(0.95, max(diffuse)) of the constant colour
        if (depth >= P.max_depth)
        {
            if (fabsf(rng.next()) > prob) ended = true;     // :590-594, no 1/prob compensation
        }
        if (!ended)
        {
            v3 r = reflect(rd, n);                  // :596
            v3 dir;
            v3 weight;
            bool diffuse_bounce = false;
            // the reference spells the same three-way roughness sampler out three times
            // (:603-624, :679-700) and the hemisphere sampler twice more (:631-636, :717-722);
            // here the branch only picks the sampler's arguments and ONE call does the work
            int sampler = 0;                        // 0: mirror direction r, 1: hemisphere about n, 2: lobe about r
            if (mtype == 0)
            {
                if (rng.next() < reflectiveness)    // :601
                {
                    sampler = roughness == 1.0f ? 1 : (roughness == 0.0f ? 0 : 2);
                    iter--;
                    weight = specular;              // :626
                }
                else
                {
                    sampler = 1;                    // :631-636
                    diffuse_bounce = true;
                    weight = diffuse;               // :638
                }
            }
            else
            {
                bool refract = false;
                v3 refractN = n;
                if (roughness != 0.0f)              // :645-654
                {
                    float w = rng.next() * roughness, th = rng.next();
                    refractN = sample_about(n, 1.0f - PTK_FLT_EPSILON, r, n, w, th);
                }
                float nc = 1.0f, ng = m3.z;
                float eta = inside ? ng / nc : nc / ng;     // :658
                float r0 = (nc - ng) / (nc + ng);
                r0 = r0 * r0;
                float c = fabsf(dot(rd, refractN));
                float k = 1.0f - eta * eta * (1.0f - c * c);
                if (k < 0.0f) refract = false;
                else
                {
                    float re = r0 + (1.0f - r0) * (1.0f - c) * (1.0f - c);    // :668
                    if (fabsf(rng.next()) < re) refract = false;
                    else if (rng.next() < reflectiveness) refract = false;
                    else refract = true;
                }
                if (!refract)
                {
                    sampler = roughness == 1.0f ? 1 : (roughness == 0.0f ? 0 : 2);
                    iter--;
                    weight = specular;              // :702
                }
                else if (rng.next() < m3.y)         // :706 translucency
                {
                    float a = eta * dot(n, rd) + sqrt_ieee(k);
                    dir = normalize(sub(muls(rd, eta), muls(refractN, a)));   // :708
                    p = sub(p, muls(muls(n, PTK_EPS), 2.0f));                  // :709
                    inside = !inside;
                    iter--;
                    weight = diffuse;               // :712
                    sampler = 3;                    // direction already set
                }
                else
                {
                    sampler = 1;                    // :717-722
                    diffuse_bounce = true;
                    weight = diffuse;               // :724
                }
            }
            if (sampler == 0) dir = r;
            else if (sampler != 3)
            {
                const bool lobe = sampler == 2;
                float w = rng.next();
                if (lobe) w = w * roughness;
                float th = rng.next();
                dir = sample_about(n, lobe ? 1.0f - PTK_FLT_EPSILON : 1.0f - PTK_EPS, lobe ? r : n, lobe ? r : n, w, th);
            }

            L = add(L, mulv(T, muls(emiss, emissI)));      // emiss * emissiveIntensity term
            v3 next_ro = p, next_rd = dir;
            float4 lt0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), lt1 = lt0, lt2 = lt0;
            if (diffuse_bounce && P.num_lights > 0)
            {
                // DirectIllumimation + SampleTriangle, pathtracer.cpp:494-531
                const float u_light = rng.next(), u_su = rng.next(), u_sv = rng.next();
                v3 l, di;
                int light_tri;
                if (sample_direct_light(P, p, n, diffuse, u_light, u_su, u_sv, l, di, light_tri, lt0, lt1, lt2))
                {
                    Tdi = mulv(T, di);
                    if (FLAT)
                    {
                        // the shadow ray rides along with the bounce ray in the next flat pass, which finds its
                        // closest hit over ALL triangles (no early end: the pass runs for the bounce ray anyway)
                        WS.begin(p, l, P.num_nodes, stack, P.scene_bound);
                        WS.occl_tri = light_tri;
                    }
                    else
                    {
                        W.occl_tri = light_tri;
                        nextDir = dir;
                        next_rd = l;
                    }
                }
            }
            T = mulv(T, weight);
            W.begin(next_ro, next_rd, P.num_nodes, stack, P.scene_bound);
            // a shadow ray meets its light triangle before anything else (see Walk::occl_tri)
            if (!FLAT && W.occl_tri >= 0) (void)tri_test<STATS>(P, W, lt0, lt1, lt2, rng, ray, cnt);
        }
    }
    return ended;
}

// ---- work distribution shared by the persistent trace kernels -------------------------------------------------------
// the current item: read only when units are dealt, so it lives in LDS, not in registers the hot loops want
enum { IT_NLIVE = 0, IT_X0, IT_Y0, IT_SBEGIN, IT_OUTBASE,
       IT_STEAL,                // queues (group + steal) & 7 ... are the ones not yet seen empty; 8 = none left
       IT_LO, IT_HI, IT_G,      // slots [lo, hi) of queue g this wave has popped and not yet used
       IT_REMAIN,               // slots that queue had left after that pop (sizes the next batch)
       IT_TAKEN,                // items this wave has traced so far (against the launch's per-wave quota, if any)
       IT_NLIVE_MAGIC,          // ceil(2^32 / live pixels): unit / live pixels = mulhi(unit, magic), exact while unit * live pixels < 2^32
       IT_WORDS };

// Takes the next non-empty work item from the per-XCD queues (-> lds_item, lds_pixel_of_rank); returns its unit count, 0
// when every queue is empty (or this wave's quota of a multi-generation launch is used up).  Wave-uniform; one wave per
// workgroup.
template <class PT>
__device__ __forceinline__ uint32_t acquire_work_item(const PT& P, uint32_t* lds_item, unsigned char* lds_pixel_of_rank, const int lane)
{
    // the launch geometry is re-read from the queue block (one coalesced load, fields broadcast with
    // v_readlane) instead of living in SGPRs across the hot loops, where it forced spills
    const int geo = ((const int*)(P.queues + 8 * PTK_QUEUE_STRIDE))[lane & 15];
    const int num_chunks = __builtin_amdgcn_readlane(geo, QG_NUM_CHUNKS), world = __builtin_amdgcn_readlane(geo, QG_WORLD);
    const int rank = __builtin_amdgcn_readlane(geo, QG_RANK);
    const int tiles_x = __builtin_amdgcn_readlane(geo, QG_TILES_X), chunk = __builtin_amdgcn_readlane(geo, QG_CHUNK);
    const uint32_t spp = (uint32_t)__builtin_amdgcn_readlane(geo, QG_SPP);
    const int slots_per_queue = __builtin_amdgcn_readlane(geo, QG_SLOTS), live_count = __builtin_amdgcn_readlane(geo, QG_LIVE_COUNT);
    // multi-GPU runs launch several generations of waves, each retiring after its quota of items, so that the
    // kernels of the exchange step (RCCL, on another stream) find free wave slots while this kernel is running
    const uint32_t quota = (uint32_t)__builtin_amdgcn_readlane(geo, QG_QUOTA);
    const int my_group = (int)blockIdx.x & 7;
    int steal = (int)lds_item[IT_STEAL];
    int lo = (int)lds_item[IT_LO], hi = (int)lds_item[IT_HI], cur_g = (int)lds_item[IT_G];
    uint32_t remain = lds_item[IT_REMAIN];
    const int pullers = max(1, (int)gridDim.x >> 3);           // waves that share one queue
    if (quota != 0u && lds_item[IT_TAKEN] >= quota && lo >= hi) steal = 8;
    for (;;)
    {
        while (lo >= hi && steal < 8)
        {
            // pop the next slot(s).  P.max_batch > 1 pops guided batches (a share of what is left per puller,
            // shrinking towards the end); measured slower on every config - consecutive slots are the chunks of
            // ONE quadrant, which are better traced by several waves at the same time - so the default is 1
            const int g = (my_group + steal) & 7;
            // (what the queue had left after this wave's previous pop has roughly halved since: the other
            // pullers popped meanwhile)
            const uint32_t left = g == cur_g && remain != 0xffffffffu ? remain / 2u : (uint32_t)slots_per_queue;
            const int want = max(1, min(P.max_batch, (int)(left / (4u * (uint32_t)pullers))));
            int old = 0;
            if (lane == 0) old = (int)atomicAdd(&P.queues[g * PTK_QUEUE_STRIDE], (unsigned)want);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old < slots_per_queue)
            {
                lo = old; hi = min(slots_per_queue, old + want); cur_g = g;
                remain = (uint32_t)(slots_per_queue - hi);
                break;
            }
            // this queue is empty: look at all eight counters at once (one load) and move on to the next
            // one that still has items, instead of finding each of them empty with an atomic of its own
            unsigned seen = ~0u;
            if (lane < 8) seen = __hip_atomic_load(&P.queues[((my_group + lane) & 7) * PTK_QUEUE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned avail = (unsigned)__ballot(seen < (unsigned)slots_per_queue) & 0xffu & (0xffu << (steal + 1));
            steal = avail ? __builtin_ctz(avail) : 8;
            if (steal < 8)
            {
                // size the first pop from the victim by what it was just seen to have left
                cur_g = (my_group + steal) & 7;
                remain = 2u * ((uint32_t)slots_per_queue - (uint32_t)__builtin_amdgcn_readlane((int)seen, steal));
            }
        }
        if (lo >= hi) break;
        const int slot = lo++;
        if (slot >= slots_per_queue) continue;                             // (one-item-per-wave mode: surplus block)
        // slot -> (entry of the live list, chunk): a queue holds runs of four consecutive entries - the
        // quadrants of one tile, as a rule - so that tile is traced within one XCD
        const int e = slot / num_chunks, chunk_id = slot - e * num_chunks;
        const int v = ((e >> 2) * 8 + cur_g) * 4 + (e & 3);
        if (v >= live_count) continue;                                     // padding of the last runs
        const int subtile = (int)P.live_list[v];                           // (owned tile) * 4 + quadrant
        const unsigned long long live_mask = P.live_mask[subtile];         // its pixels that need tracing
        const int item = subtile * num_chunks + chunk_id;
        const int owned = subtile >> 2, quad = subtile & 3;
        const int tile = owned * world + rank;
        // rows are rotated by 3 tiles each so that a rank's tiles form diagonals, not columns (load balance)
        const int ty = tile / tiles_x, tx = (tile % tiles_x + tiles_x - (3 * ty) % tiles_x) % tiles_x;
        const int x0 = tx * PTK_TILE + (quad & 1) * 8, y0 = ty * PTK_TILE + (quad >> 1) * 8;
        const uint32_t s_begin = (uint32_t)chunk_id * (uint32_t)chunk;
        const uint32_t s_count = min((uint32_t)chunk, spp - s_begin);      // host guarantees s_begin < spp
        const uint32_t n_live = (uint32_t)__popcll(live_mask);
        if (n_live == 0) continue;
        __syncthreads();                    // every lane is done with the previous item's table
        if ((live_mask >> lane) & 1ull) lds_pixel_of_rank[__popcll(live_mask & ((1ull << lane) - 1ull))] = (unsigned char)lane;
        if (lane == 0)
        {
            lds_item[IT_NLIVE] = n_live; lds_item[IT_NLIVE_MAGIC] = (uint32_t)((0x100000000ull + n_live - 1u) / n_live); lds_item[IT_X0] = (uint32_t)x0; lds_item[IT_Y0] = (uint32_t)y0;
            lds_item[IT_SBEGIN] = s_begin; lds_item[IT_STEAL] = (uint32_t)steal; lds_item[IT_TAKEN] += 1u;
            lds_item[IT_LO] = (uint32_t)lo; lds_item[IT_HI] = (uint32_t)hi; lds_item[IT_G] = (uint32_t)cur_g; lds_item[IT_REMAIN] = remain;
            lds_item[IT_OUTBASE] = (uint32_t)item * (uint32_t)chunk * 64u;   // sample s of quadrant pixel q: P.samples[base + s * 64 + q]
        }
        __syncthreads();
        return n_live * s_count;
    }
    if (lane == 0) { lds_item[IT_STEAL] = 8u; lds_item[IT_LO] = lds_item[IT_HI] = 0u; }
    __syncthreads();
    return 0u;
}

// (the two states that wait for a camera ray - no unit yet, unit dealt - are the two smallest: one compare counts both)
enum : int { ST_NEED = 0, ST_GEN = 1, ST_TRAV = 2, ST_SHADE = 3, ST_DONE = 4 };

// FLAT = the scene has so few triangles (P.flat_count <= 16) that no hierarchy is walked: a traversing
// lane tests every triangle, the records are fetched with SCALAR loads (one s_load per triangle per
// wave, operands broadcast from SGPRs, no vector memory traffic and no LDS stack), and the whole walk
// is one block of the state machine, so lanes re-synchronise by themselves.
template <bool STATS, bool FLAT>
__global__ __launch_bounds__(PTK_TRACE_BLOCK, (FLAT ? PTK_TRACE_WAVES : PTK_TRACE_WAVES_BVH)) void trace_kernel(const RenderParams* __restrict__ Pp)
{
    // the parameters live in the launch's queue block, in the constant address space: fields are s_load-ed where they are
    // used instead of being preloaded whole into SGPRs (ptk_device.h: 30 / 42 SGPR spills -> 0)
    typedef const __attribute__((address_space(4))) RenderParams ConstParams;
    ConstParams& P = *(ConstParams*)(uintptr_t)Pp;
    __shared__ int lds_stack[FLAT ? 1 : (PTK_MAX_BVH_DEPTH + PTK_PUSH_BRANCHLESS) * PTK_TRACE_BLOCK];
    if (P.exit_flag && __hip_atomic_load(P.exit_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= P.exit_gen) return;     // an Exit() named this render or a later one

    const int tid = threadIdx.x;
    int* stack = lds_stack + tid;
    const int lane = tid & 63;
    // ---- work distribution -------------------------------------------------------------------------
    // Work item = (owned 16x16 tile, 8x8 quadrant, chunk of samples); its work UNITS = (live pixel of the
    // quadrant, sample of the chunk), sample-major.  Units are dealt to whichever lane needs work next, NOT
    // pinned pixel-to-lane, and the waves are PERSISTENT: a wave that has dealt the last unit of its item
    // takes the next item from a queue while its other lanes are still finishing paths of the previous one,
    // so all 64 lanes keep tracing until the whole launch runs dry (a lane's state carries its own pixel,
    // sample and output slot).  A pixel-per-lane mapping idled the wave while its unluckiest pixel finished
    // its samples, and a wave per item left a tail at the end of every item.
    //
    // XCD-aware queues: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), each
    // with its own L2.  There is one queue per XCD group, holding whole 16x16 tiles (runs of 4 x num_chunks
    // items) tile by tile round-robin, so the waves that trace the same pixels - the same BVH nodes and
    // triangles - share an L2; a group whose queue is empty steals from the others.  (One contiguous eighth
    // of the frame per XCD was 2x slower when the dispatcher dealt the items: cheap and dear regions.)
    static_assert(PTK_TRACE_BLOCK == 64, "one wave per workgroup");
    __shared__ unsigned char lds_pixel_of_rank[64];
    __shared__ uint32_t lds_item[IT_WORDS];
    if (lane == 0)
    {
        // small launches (P.persistent == 0: few items per wave slot) run one item per wave, named by blockIdx,
        // with no queue traffic at all - the hardware dispatcher balances those better than 4096 waves
        // contending for eight counters could
        lds_item[IT_STEAL] = P.persistent ? 0u : 8u;
        lds_item[IT_LO] = P.persistent ? 0u : (uint32_t)blockIdx.x >> 3;
        lds_item[IT_HI] = P.persistent ? 0u : ((uint32_t)blockIdx.x >> 3) + 1u;
        lds_item[IT_G] = (uint32_t)blockIdx.x & 7u;
        lds_item[IT_REMAIN] = 0xffffffffu;      // "unknown": the first pop is sized from the whole queue
        lds_item[IT_TAKEN] = 0u;
    }
    __syncthreads();
    uint32_t total_units = 0, next_unit = 0;    // wave-uniform
    // per-lane: the unit this lane is tracing
    uint32_t pix = 0;               // row-major pixel index from the top
    uint32_t sample_abs = 0;        // sample index inside this launch (chunk start + index in the chunk)
    uint32_t out_idx = 0;           // its slot in the sample buffer

    // takes the next non-empty item (-> lds_item, lds_pixel_of_rank); returns its unit count, 0 when every queue is empty
    auto acquire_item = [&]() -> uint32_t { return acquire_work_item(P, lds_item, lds_pixel_of_rank, lane); };

    const v3 camPos0 = V(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    const v3 camRight = V(P.cam_right[0], P.cam_right[1], P.cam_right[2]);
    const v3 camUp = V(P.cam_up[0], P.cam_up[1], P.cam_up[2]);

    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    Rng rng;
    rng.inc = 1u; rng.state = 0; rng.key = 0;

    // per-lane path state
    Walk W;
    W.begin(camPos0, V(0.0f, 0.0f, 1.0f), 0, stack, 0.0f);
    W.occl_tri = -1;
    Walk WS;                        // FLAT only: the shadow ray, tested in the same pass as the bounce ray
    WS.begin(camPos0, V(0.0f, 0.0f, 1.0f), 0, stack, 0.0f);
    WS.occl_tri = -1;
    v3 L = V(0.0f, 0.0f, 0.0f), T = V(1.0f, 1.0f, 1.0f);
    v3 Tdi = V(0.0f, 0.0f, 0.0f), nextDir = V(0.0f, 0.0f, 1.0f);
    int depth = 0, iter = 0;
    bool inside = false;
    uint32_t ray = 0;
    int st = ST_NEED;               // every lane works, whatever its own pixel is

    // (what the walk loop's exits read of the parameters, in SGPRs for the kernel's life: see WalkParams)
    float4* const samples_u = uniform_ptr(P.samples);
    const int num_nodes_u = __builtin_amdgcn_readfirstlane(P.num_nodes);
    const float scene_bound_u = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(P.scene_bound)));
    // a finished path: its radiance goes to the sample buffer, the lane moves to its next sample
#define PTK_FINISH_PATH()                                                                         \
    do {                                                                                          \
        samples_u[out_idx] = make_float4(L.x, L.y, L.z, 0.0f);                                    \
        st = ST_NEED;                                                                             \
    } while (0)

    // a finished walk: shadow rays resolve DirectIllumimation's visibility and roll into the sampled
    // bounce; bounce rays end the path on a miss or queue for shading
#define PTK_WALK_DONE()                                                                           \
    do {                                                                                          \
        if (STATS) { cnt.rays++; cnt.max_nodes = max(cnt.max_nodes, cnt.cur_nodes); cnt.cur_nodes = 0; }     \
        ray++;                                                                                    \
        const bool hit_ = W.best.tri != PTK_NOHIT;                                                \
        if (W.occl_tri >= 0)                                                                      \
        {                                                                                         \
            /* pathtracer.cpp:522-526: lit unless something else is closest */                    \
            if (STATS) cnt.shadow++;                                                              \
            if (!(hit_ && W.best.tri != W.occl_tri)) L = add(L, Tdi);                             \
            W.occl_tri = -1;                                                                      \
            W.begin(W.ro, nextDir, num_nodes_u, stack, scene_bound_u);                                   \
        }                                                                                         \
        else if (!hit_) PTK_FINISH_PATH();              /* :550 miss -> black */                  \
        else st = ST_SHADE;                                                                       \
    } while (0)

    int debt_shade = 0, debt_gen = 0;      // wave-uniform: lane-iterations wasted by parked lanes
    for (;;)
    {
        // deal the next work units to the lanes that need one (wave-uniform code)
        {
            unsigned long long m_need = __ballot(st == ST_NEED);
            while (m_need)
            {
                if (next_unit >= total_units)
                {
                    next_unit = 0;
                    total_units = (lds_item[IT_STEAL] < 8u || lds_item[IT_LO] < lds_item[IT_HI]) ? acquire_item() : 0u;
                    if (total_units == 0) { if (st == ST_NEED) st = ST_DONE; break; }
                }
                const uint32_t n_live = lds_item[IT_NLIVE];
                const uint32_t u = next_unit + (uint32_t)__popcll(m_need & ((1ull << lane) - 1ull));
                if (st == ST_NEED && u < total_units)
                {
                    const uint32_t s_in_chunk = n_live == 1u ? u : __umulhi(u, lds_item[IT_NLIVE_MAGIC]);   // = u / n_live without the integer-division expansion
                    const uint32_t q = lds_pixel_of_rank[u - s_in_chunk * n_live];
                    pix = (lds_item[IT_Y0] + (q >> 3)) * (uint32_t)P.width + lds_item[IT_X0] + (q & 7u);
                    sample_abs = lds_item[IT_SBEGIN] + s_in_chunk;
                    out_idx = lds_item[IT_OUTBASE] + s_in_chunk * 64u + q;
#if PTK_FUSED_START
                    // cached camera rays (pinhole, no stochastic opacity): the path starts at its first surface
                    // interaction, so the lane queues for the SHADE block directly and sets its path up there (depth < 0
                    // marks it) - one voted block less to wait for per path, and bigger shading batches
                    st = P.primary_hit ? ST_SHADE : ST_GEN;
                    depth = -1;
#else
                    st = ST_GEN;
#endif
                }
                next_unit = min(total_units, next_unit + (uint32_t)__popcll(m_need));
                m_need = __ballot(st == ST_NEED);
            }
        }
        const unsigned long long m_trav = __ballot(st == ST_TRAV);
        const unsigned long long m_shade = __ballot(st == ST_SHADE);
        const unsigned long long m_gen = __ballot(st == ST_GEN);
        const int n_trav = __popcll(m_trav), n_shade = __popcll(m_shade), n_gen = __popcll(m_gen);
        const int n_live = n_trav + n_shade + n_gen;
        if (n_live == 0) break;

        // Block choice ("ski rental"): a lane parked in SHADE / GEN wastes one lane-iteration for
        // every walk iteration it sits out, while running its block now wastes the lanes that are not
        // parked there.  The walk keeps stepping until the lane-iterations wasted by the parked lanes
        // (debt, accumulated per walk iteration) exceed lambda x the lanes the block would leave idle,
        // with lambda ~ cost(block) / cost(walk step) (P.shade_thr, P.gen_thr in eighths).  Short walks
        // (a bare Cornell box) thus batch ~3/4 of a wave per shade call, deep trees with straggling
        // rays shade small batches early instead of idling - measured optimum in both regimes.
        bool run_shade = n_trav == 0 && n_shade >= n_gen && n_shade > 0;
        bool run_gen = n_trav == 0 && !run_shade;
        if (FLAT)
        {
            // every block is one complete unit of work for a lane: run the one most lanes wait for
            // (weights in eighths: a cheap block may run with fewer lanes than an expensive one)
            const int sc_trav = n_trav * 8, sc_shade = n_shade * P.flat_shade_w, sc_gen = n_gen * P.flat_gen_w;
            if (n_trav > 0 && sc_trav >= sc_shade && sc_trav >= sc_gen)
            {
                if (STATS && lane == 0) { cnt.walk_iters++; cnt.walk_lanes += (uint32_t)n_trav; }
                if (st == ST_TRAV)
                {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    typedef const __attribute__((address_space(4))) f4v* cf4;        // constant address space -> s_load
                    const cf4 ct = (cf4)(uintptr_t)P.flat_tris;            // the scene's triangles in ascending index order
                    // both rays of a diffuse bounce in one pass over the triangles: the shadow ray
                    // (ray number `ray`) and the sampled bounce (`ray + 1`), pathtracer.cpp:638 / :724
                    const bool shadow = WS.occl_tri >= 0;
                    const uint32_t bounce_ray = shadow ? ray + 1u : ray;
                    RayPair R;
                    R.ox = f2{ W.ro.x, WS.ro.x }; R.oy = f2{ W.ro.y, WS.ro.y }; R.oz = f2{ W.ro.z, WS.ro.z };
                    R.dx = f2{ W.rd.x, WS.rd.x }; R.dy = f2{ W.rd.y, WS.rd.y }; R.dz = f2{ W.rd.z, WS.rd.z };
                    for (int k = 0; k < P.flat_count; k++)
                    {
                        const f4v a0 = ct[k * TRI_F4], a1 = ct[k * TRI_F4 + 1], a2 = ct[k * TRI_F4 + 2];
                        const float4 t0 = make_float4(a0.x, a0.y, a0.z, a0.w), t1 = make_float4(a1.x, a1.y, a1.z, a1.w),
                                     t2 = make_float4(a2.x, a2.y, a2.z, a2.w);
                        (void)tri_test_pair<STATS>(P, W, WS, R, shadow, t0, t1, t2, rng, bounce_ray, ray, cnt);
                    }
                    if (shadow)
                    {
                        // DirectIllumimation visibility (pathtracer.cpp:522-526): lit unless something else is closest
                        if (STATS) { cnt.rays++; cnt.shadow++; }
                        if (!(WS.best.tri != PTK_NOHIT && WS.best.tri != WS.occl_tri)) L = add(L, Tdi);
                        WS.occl_tri = -1;
                        ray++;
                    }
                    W.node = NODE_EXIT;
                    PTK_WALK_DONE();
                }
                continue;
            }
            run_shade = n_shade > 0 && (sc_shade >= sc_gen || n_gen == 0);
            run_gen = !run_shade;
        }
        else if (n_trav > 0)
        {
            // ---- BVH walk ----
            // The loop's bookkeeping is wave-uniform and kept to four ballots per iteration: a lane is walking exactly when
            // its walk has a leaf triangle pending or a node to visit (W is `done` in every other state, from W.begin's
            // initial call on), so the two masks that vote the triangle arm also count the walking lanes.
            int ds = __builtin_amdgcn_readfirstlane(debt_shade), dg = __builtin_amdgcn_readfirstlane(debt_gen);
            unsigned long long m_tq = __ballot(W.tri_left > 0), m_nr = __ballot(W.node >= 0);
            bool want_shade = false, want_gen = false;
            const WalkParams WP = walk_params(P);          // the loop's share of the parameters, in SGPRs
            NodeRec nrec;
            request_node(WP, W, nrec);                     // the walk keeps one node record in flight across iterations (walk_step)
            do
            {
                if (STATS) { const uint32_t nt = (uint32_t)__popcll(__ballot(st == ST_TRAV)); if (lane == 0) { cnt.walk_iters++; cnt.walk_lanes += nt; } }
                // arm A (triangles) is voted: about one triangle is tested per four nodes visited, so run
                // every iteration it would execute with ~1/7 of the lanes.  A lane parks its leaf and walks
                // on until it reaches the next leaf; the arm runs once enough lanes wait for it (P.tri_thr
                // eighths of the lanes that still have a node to visit) or nobody can walk on.  Deferring
                // only delays t-max tightening: the closest hit is order-independent.  (Measured: waiting
                // for half of the walking lanes, thr 4, is the optimum - beyond that the rays whose walk
                // is finished but for the parked leaf idle too long; a 4-deep leaf ring per lane made
                // that worse, not better.)
                const int n_tq = __popcll(m_tq), n_nr = __popcll(m_nr);
                const bool run_tri_arm = (n_tq > 0) & ((n_nr == 0) | (n_tq * 8 >= WP.tri_thr * n_nr));      // (bitwise: no scalar branches)
                if (STATS && lane == 0 && run_tri_arm) { cnt.tri_execs++; cnt.tri_lanes += (uint32_t)n_tq; }
                if (st == ST_TRAV)
                {
                    walk_step<STATS, PTK_TRACE_BLOCK, true>(WP, W, rng, ray, stack, cnt, run_tri_arm, &nrec);
                    if (W.done()) PTK_WALK_DONE();
                }
                m_tq = __ballot(W.tri_left > 0); m_nr = __ballot(W.node >= 0);
                const int nt = __popcll(m_tq | m_nr);
                const int ns = __popcll(__ballot(st == ST_SHADE));
                // a lane whose path ended in the walk (its ray left the scene) waits in NEED for a new unit: it runs
                // up the same debt as a lane waiting for the camera-ray block, and is dealt its unit first
                const int ng = __popcll(__ballot(st < ST_TRAV));
                const int nl = nt + ns + ng;
                ds += ns; dg += ng;
                // one exit test per iteration, which exit it was is sorted out after the loop
                want_shade = (ns > 0) & (ds * 8 >= WP.shade_thr * (nl - ns));
                want_gen = (ng > 0) & (dg * 8 >= WP.gen_thr * (nl - ng));
                if (want_shade | want_gen | (nt == 0)) break;
            } while (true);
            if (want_shade) run_shade = true;
            else if (want_gen) run_gen = __ballot(st == ST_NEED) == 0ull;      // NEED lanes: re-vote via the top of the loop
            debt_shade = ds; debt_gen = dg;
            if (!run_shade && !run_gen) continue;        // the walk ran dry: re-vote
        }
        if (run_shade) debt_shade = 0; else debt_gen = 0;
        if (run_shade)
        {
            if (STATS) { const uint32_t nsx = (uint32_t)__popcll(__ballot(st == ST_SHADE)); if (lane == 0) { cnt.shade_execs++; cnt.shade_lanes += nsx; } }
            if (st == ST_SHADE)
            {
#if PTK_FUSED_START
                if (depth < 0)
                {
                    // a path dealt since the last shade block (see the camera-ray block below for the reasoning)
                    const uint2 pr = P.pixel_rng[pix];
                    const float4 c = P.primary_hit[pix], r = P.primary_rd[pix];
                    rng.inc = pr.y;
                    rng.state = hash32(P.first_sample + sample_abs + pr.x);
                    rng.key = rng.state;
                    rng.state = rng.state * (747796405u * 747796405u) + rng.inc * (747796405u + 1u);
                    L = V(0.0f, 0.0f, 0.0f); T = V(1.0f, 1.0f, 1.0f);
                    depth = 0; iter = 0; inside = false; ray = 1;
                    W.occl_tri = -1;
                    if (STATS) cnt.started++;
                    W.ro = camPos0; W.rd = V(r.x, r.y, r.z);
                    W.best.tri = __float_as_int(c.x); W.best.t = c.y; W.best.u = c.z; W.best.v = c.w;
                    W.node = NODE_EXIT; W.top = stack; W.tri_left = 0;
                }
#endif
                // ---- one surface interaction of PathTracer::Trace, pathtracer.cpp:551-727 ----
                const bool ended = shade_interaction<STATS, FLAT>(P, W, WS, stack, rng, L, T, Tdi, nextDir, depth, iter, inside, ray, cnt);
                if (ended) PTK_FINISH_PATH();
                else st = ST_TRAV;
            }
        }
        else
        {
            if (STATS) { const uint32_t ngx = (uint32_t)__popcll(__ballot(st == ST_GEN)); if (lane == 0) { cnt.gen_execs++; cnt.gen_lanes += ngx; } }
            if (st == ST_GEN)
            {
                // ---- camera ray with thin-lens DOF, pathtracer.cpp:785-791 + SampleCircle :734-739 ----
                // the pixel's RNG stream constants (pixel_key, increment) come from a per-frame table: they depend on
                // (seed, pixel) only, and four of the five hashes of a path's start went into them
                const uint2 pr = P.pixel_rng[pix];
                const uint32_t pkey = pr.x;
                rng.inc = pr.y;
                rng.state = hash32(P.first_sample + sample_abs + pkey);
                rng.key = rng.state;
                L = V(0.0f, 0.0f, 0.0f); T = V(1.0f, 1.0f, 1.0f);
                depth = 0; iter = 0; inside = false; ray = 0;
                W.occl_tri = -1;
                if (STATS) cnt.started++;
                if (!PTK_FUSED_START && PTK_GEN_CACHED_FAST && P.primary_hit)      // (fused start: such lanes never come here)
                {
                    // Pinhole camera, no stochastic opacity: every sample of this pixel shoots the same camera ray, so its
                    // direction and closest hit were computed once (primary_hits_kernel) and the path starts at its first
                    // surface interaction.  The two SampleCircle draws a pinhole frame still consumes (pathtracer.cpp:787,
                    // always two) only advance the stream - two LCG steps in one:
                    //   s2 = (s * a + inc) * a + inc = s * a^2 + inc * (a + 1)      (mod 2^32: the identical state)
                    rng.state = rng.state * (747796405u * 747796405u) + rng.inc * (747796405u + 1u);
                    const float4 c = P.primary_hit[pix], r = P.primary_rd[pix];
                    W.ro = camPos0; W.rd = V(r.x, r.y, r.z);
                    W.best.tri = __float_as_int(c.x); W.best.t = c.y; W.best.u = c.z; W.best.v = c.w;
                    W.node = NODE_EXIT; W.top = stack; W.tri_left = 0;
                    ray = 1;
                    st = ST_SHADE;              // (pixels whose camera ray misses never get here)
                }
                else
                {
                    // per-pixel constants are re-read here (L1/L2 hits) instead of living in registers
                    const float4 d0 = P.primary[pix];
                    const v3 dir0 = V(d0.x, d0.y, d0.z);
                    v3 focalPoint = add(camPos0, muls(dir0, P.focal_dist));
                    float r1 = rng.next(), r2 = rng.next();          // always two draws, even with a pinhole
                    v3 ro = camPos0;
                    if (P.aperture != 0.0f)
                    {
                        float angle = (float)((double)r1 * 2. * PTK_PI_D);
                        float radius = sqrt_ieee(r2);
                        float sn, cs;
                        sincos_2pi(angle, sn, cs);
                        float offx = (cs * radius) * P.aperture, offy = (sn * radius) * P.aperture;
                        ro = add(camPos0, add(muls(camRight, offx), muls(camUp, offy)));
                    }
                    v3 rd = normalize(sub(focalPoint, ro));
                    W.begin(ro, rd, P.num_nodes, stack, P.scene_bound);
                    st = ST_TRAV;
                    if (!PTK_FUSED_START && P.primary_hit)
                    {
                        const float4 c = P.primary_hit[pix];
                        W.best.tri = __float_as_int(c.x); W.best.t = c.y; W.best.u = c.z; W.best.v = c.w;
                        W.node = NODE_EXIT;
                        ray = 1;
                        st = ST_SHADE;
                    }
                }
            }
        }
    }
#undef PTK_WALK_DONE
#undef PTK_FINISH_PATH
    if (STATS)
    {
        atomicAdd(&P.stats[0], (unsigned long long)cnt.started);
        atomicAdd(&P.stats[1], (unsigned long long)cnt.rays);
        atomicAdd(&P.stats[2], (unsigned long long)cnt.shadow);
        atomicAdd(&P.stats[3], (unsigned long long)cnt.nodes);
        atomicAdd(&P.stats[4], (unsigned long long)cnt.tris);
        atomicAdd(&P.stats[5], (unsigned long long)cnt.shaded);
        atomicAdd(&P.stats[6], (unsigned long long)cnt.tex);
        atomicMax(&P.stats[15], (unsigned long long)cnt.max_nodes);
        if (lane == 0)
        {
            atomicAdd(&P.stats[7], (unsigned long long)cnt.walk_iters);
            atomicAdd(&P.stats[8], (unsigned long long)cnt.walk_lanes);
            atomicAdd(&P.stats[9], (unsigned long long)cnt.shade_execs);
            atomicAdd(&P.stats[10], (unsigned long long)cnt.shade_lanes);
            atomicAdd(&P.stats[11], (unsigned long long)cnt.gen_execs);
            atomicAdd(&P.stats[12], (unsigned long long)cnt.gen_lanes);
            atomicAdd(&P.stats[13], (unsigned long long)cnt.tri_execs);
            atomicAdd(&P.stats[14], (unsigned long long)cnt.tri_lanes);
        }
    }
}

#if !PTK_CONTRACT      // ---- everything but the trace kernels exists once, in the exact build
// Streaming fold of the sample buffer into the float accumulator, strictly in sample order
// (`mTotalImg[px] += color` once per RenderFrame(), pathtracer.cpp:798-800), plus the 8-bit resolve
// (pathtracer.cpp:802-812).  One thread per pixel; each sample read is a coalesced 1 KiB per wave.
__global__ __launch_bounds__(PTK_BLOCK) void accumulate_kernel(const RenderParams P)
{
    // an aborted pass adds nothing: trace waves that saw the exit flag returned without writing their samples, so the
    // sample buffer may hold another pass's values (the reference adds nothing for the rows it skips, pathtracer.cpp:779-780)
    if (P.exit_flag && __hip_atomic_load(P.exit_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= P.exit_gen) return;     // an Exit() named this render or a later one
    const int tid = threadIdx.x;
    const int lane = tid & 63, quad = tid >> 6;
    const int owned = blockIdx.x;
    const int tile = owned * P.world + P.rank;
    if (tile >= P.num_tiles) return;
    // rows are rotated by 3 tiles each so that a rank's tiles form diagonals, not columns (load balance)
    const int ty = tile / P.tiles_x, tx = (tile % P.tiles_x + P.tiles_x - (3 * ty) % P.tiles_x) % P.tiles_x;
    const int px = tx * PTK_TILE + (quad & 1) * 8 + (lane & 7);
    const int py = ty * PTK_TILE + (quad >> 1) * 8 + (lane >> 3);
    if (px >= P.width || py >= P.height) return;
    const size_t accidx = ((size_t)(P.height - 1 - py) * P.width + px) * 3;   // bottom-up (pathtracer.cpp:796)
    v3 acc = V(P.accum[accidx], P.accum[accidx + 1], P.accum[accidx + 2]);
    const size_t subtile = (size_t)owned * 4 + quad;
    // (a pixel that is not in its quadrant's live mask - cached camera ray misses, or no lens ray reaches the scene - was not
    // traced: nothing was stored for it and it receives nothing)
    const bool black = ((P.live_mask[subtile] >> lane) & 1ull) == 0ull;
    if (!black)
    {
        // the samples of one pixel are a strided array (chunk after chunk of its quadrant's items): sample s sits at
        // in[s * 64].  Eight loads in flight per lane, added strictly in sample order.
        const float4* in = P.samples + (subtile * P.num_chunks * P.chunk) * 64 + lane;
        uint32_t s = 0;
        for (; s + 8 <= P.spp; s += 8)
        {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = in[(size_t)(s + k) * 64];
#pragma unroll
            for (int k = 0; k < 8; k++) acc = add(acc, V(v[k].x, v[k].y, v[k].z));
        }
        for (; s < P.spp; s++)
        {
            const float4 col = in[(size_t)s * 64];
            acc = add(acc, V(col.x, col.y, col.z));
        }
    }
    P.accum[accidx] = acc.x; P.accum[accidx + 1] = acc.y; P.accum[accidx + 2] = acc.z;
    float c3[3] = { acc.x / P.resolve_samples, acc.y / P.resolve_samples, acc.z / P.resolve_samples };
    uint8_t b3[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
    {
        float x = c3[k];
        x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
        if (!(x == x)) x = 0.0f;
        b3[k] = (uint8_t)(x * 255);
        P.rgb8[accidx + k] = b3[k];
    }
    if (P.rgb8_host)
    {
        // The hand-off: straight into the caller's page-locked buffer, over PCIe.  A pixel that receives nothing AND holds
        // nothing resolves to 0 whatever the sample count - it was written when the buffer was bound / reset and is
        // skipped (four fifths of the C2 frame); one that holds light from before a camera move keeps dimming and is
        // written.  Byte stores of single pixels crawl over the link (measured: 0.5 MB in 70 us), so a row of the quadrant
        // - 8 pixels, 24 contiguous bytes - is gathered with lane shuffles and leaves as six dwords.
        const bool skip_host = black && acc.x == 0.0f && acc.y == 0.0f && acc.z == 0.0f && !P.rgb8_host_full;
        const uint32_t mine = (uint32_t)b3[0] | ((uint32_t)b3[1] << 8) | ((uint32_t)b3[2] << 16);
        const unsigned long long row_live = (__ballot(!skip_host) >> (lane & ~7)) & 0xffull;     // this row's pixels that must be written
        // dword d (0..5) of the row holds bytes 4d..4d+3 = pixels (4d)/3 .. (4d+3)/3; lanes 0..5 of each row write one each
        const int d = lane & 7;
        const int p0 = (4 * d) / 3, p1 = min(7, (4 * d + 3) / 3), sh = (4 * d) % 3;          // first pixel, last pixel, byte offset in the first
        const uint32_t w0 = (uint32_t)__shfl((int)mine, (lane & ~7) + min(p0, 7)), w1 = (uint32_t)__shfl((int)mine, (lane & ~7) + p1);
        // bytes of pixel p0 from offset sh, then pixel p0 + 1 (= p1 unless the dword lies within one pixel... it never does: 4 > 3)
        const uint32_t word = (w0 >> (8 * sh)) | (w1 << (8 * (3 - sh)));
        const bool aligned = (((size_t)P.width * 3) & 3) == 0 && (((uintptr_t)P.rgb8_host) & 3) == 0;
        const int row_px = min(8, P.width - (px - (lane & 7)));                                // pixels of this row on the image (>= 1 here)
        if (aligned && row_px == 8)
        {
            if (d < 6 && row_live != 0ull) *(uint32_t*)(P.rgb8_host + accidx - (size_t)(lane & 7) * 3 + d * 4) = word;
        }
        else if (!skip_host)
        {
            P.rgb8_host[accidx] = b3[0]; P.rgb8_host[accidx + 1] = b3[1]; P.rgb8_host[accidx + 2] = b3[2];
        }
    }
}

// Primary ray directions before DOF: one thread per image row walks the row with the reference's
// incremental `pixel += camRight * deltaX` (pathtracer.cpp:782-785, :814), so every direction is
// the value the reference computes.  Runs once per camera / resolution change.
__global__ void primary_dirs_kernel(const PrimaryParams P)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.height) return;
    v3 up = V(P.cam_up[0], P.cam_up[1], P.cam_up[2]);
    v3 right = V(P.cam_right[0], P.cam_right[1], P.cam_right[2]);
    v3 pos = V(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    v3 pixel = sub(V(P.top_left[0], P.top_left[1], P.top_left[2]), muls(up, (float)i * P.delta_y));
    v3 step = muls(right, P.delta_x);
    float4* row = P.primary + (size_t)i * P.width;
    for (int j = 0; j < P.width; j++)
    {
        v3 d = normalize(sub(pixel, pos));
        row[j] = make_float4(d.x, d.y, d.z, 0.0f);
        pixel = add(pixel, step);
    }
}

// Primary-visibility cache for pinhole cameras (aperture == 0) in scenes without opacity textures: the
// camera ray of a pixel is the same for every sample (pathtracer.cpp:785-791 with a zero lens offset),
// so its closest hit is found once per camera / scene change instead of once per sample.
__global__ __launch_bounds__(PTK_BLOCK) void primary_hits_kernel(const RenderParams P, float4* out, float4* out_rd)
{
    __shared__ int lds_stack[(PTK_MAX_BVH_DEPTH + PTK_PUSH_BRANCHLESS) * PTK_BLOCK];
    const int i = blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (i >= P.width * P.height) return;
    Rng rng; rng.inc = 1u; rng.state = 0u; rng.key = 0u;            // no opacity draws can occur here
    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    float4 d = P.primary[i];
    const v3 camPos0 = V(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    v3 focalPoint = add(camPos0, muls(V(d.x, d.y, d.z), P.focal_dist));
    v3 rd = normalize(sub(focalPoint, camPos0));
    Walk W;
    W.occl_tri = -1;
    W.begin(camPos0, rd, P.num_nodes, lds_stack + threadIdx.x, P.scene_bound);
    while (!W.done()) walk_step<false, PTK_BLOCK>(P, W, rng, 0u, lds_stack + threadIdx.x, cnt);
    out[i] = make_float4(__int_as_float(W.best.tri), W.best.t, W.best.u, W.best.v);
    out_rd[i] = make_float4(rd.x, rd.y, rd.z, 0.0f);           // the very floats the camera-ray block computes for a zero lens offset
}

// Parity probe: closest hit for a list of rays (no opacity draws differ: key 0, ray 0).
__global__ __launch_bounds__(PTK_BLOCK) void probe_hits_kernel(const ProbeParams P)
{
    __shared__ int lds_stack[(PTK_MAX_BVH_DEPTH + PTK_PUSH_BRANCHLESS) * PTK_BLOCK];
    int i = blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (i >= P.n) return;
    Rng rng; rng.inc = (hash32(0u ^ 0x9E3779B9u) << 1) | 1u; rng.state = hash32(0u); rng.key = rng.state;
    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    v3 ro = V(P.ro[i * 3], P.ro[i * 3 + 1], P.ro[i * 3 + 2]);
    v3 rd = V(P.rd[i * 3], P.rd[i * 3 + 1], P.rd[i * 3 + 2]);
    Walk W;
    W.occl_tri = -1;
    W.begin(ro, rd, P.num_nodes, lds_stack + threadIdx.x, P.scene_bound);
    while (!W.done()) walk_step<false, PTK_BLOCK>(P, W, rng, 0u, lds_stack + threadIdx.x, cnt);
    bool hit = W.best.tri != PTK_NOHIT;
    P.tri[i] = hit ? W.best.tri : -1;
    P.tuv[i * 3] = hit ? W.best.t : 0.0f; P.tuv[i * 3 + 1] = hit ? W.best.u : 0.0f; P.tuv[i * 3 + 2] = hit ? W.best.v : 0.0f;
}

// Parity probe of DirectIllumimation (pathtracer.cpp:505-531) with its three draws on tape: the sampling half above, then the
// shadow walk and the visibility rule exactly as trace_kernel applies them (PTK_WALK_DONE).
__global__ __launch_bounds__(PTK_BLOCK) void probe_direct_kernel(const ProbeParams P, const float* __restrict__ pts, const float* __restrict__ nrm,
                                                                 const float* __restrict__ dif, const float* __restrict__ tape, float* __restrict__ out)
{
    __shared__ int lds_stack[(PTK_MAX_BVH_DEPTH + PTK_PUSH_BRANCHLESS) * PTK_BLOCK];
    const int i = blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (i >= P.n) return;
    Rng rng; rng.inc = (hash32(0u ^ 0x9E3779B9u) << 1) | 1u; rng.state = hash32(0u); rng.key = rng.state;
    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const v3 p = V(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]), n = V(nrm[i * 3], nrm[i * 3 + 1], nrm[i * 3 + 2]);
    const v3 diffuse = V(dif[i * 3], dif[i * 3 + 1], dif[i * 3 + 2]);
    v3 l, di, res = V(0.0f, 0.0f, 0.0f);
    int light_tri;
    float4 lt0, lt1, lt2;
    if (P.num_lights > 0 && sample_direct_light(P, p, n, diffuse, tape[i * 3], tape[i * 3 + 1], tape[i * 3 + 2], l, di, light_tri, lt0, lt1, lt2))
    {
        Walk W;
        W.begin(p, l, P.num_nodes, lds_stack + threadIdx.x, P.scene_bound);
        W.occl_tri = light_tri;
        (void)tri_test<false>(P, W, lt0, lt1, lt2, rng, 0u, cnt);
        while (!W.done()) walk_step<false, PTK_BLOCK>(P, W, rng, 0u, lds_stack + threadIdx.x, cnt);
        if (!(W.best.tri != PTK_NOHIT && W.best.tri != W.occl_tri)) res = di;        // :522-526: lit unless something else is closest
    }
    out[i * 3] = res.x; out[i * 3 + 1] = res.y; out[i * 3 + 2] = res.z;
}
void launch_probe_direct(const ProbeParams& p, const float* pts, const float* nrm, const float* dif, const float* tape, float* out, hipStream_t stream)
{
    if (p.n > 0) hipLaunchKernelGGL(probe_direct_kernel, dim3((p.n + PTK_BLOCK - 1) / PTK_BLOCK), dim3(PTK_BLOCK), 0, stream, p, pts, nrm, dif, tape, out);
}

// Parity probe of the exact-arithmetic helpers the kernels use in place of `1.0f / a` and `sqrtf(x)` (op 0: rcp_ieee,
// 1: rcp_ieee_any, 2: sqrt_ieee, 3: the normalisation's 1 / sqrt(x)): the tests hold them against the host's IEEE results.
__global__ __launch_bounds__(PTK_BLOCK) void probe_math_kernel(int op, const float* __restrict__ in, float* __restrict__ out, int n)
{
    const int i = blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    out[i] = op == 0 ? rcp_ieee(x) : (op == 1 ? rcp_ieee_any(x) : (op == 2 ? sqrt_ieee(x) : rcp_ieee_any(sqrt_ieee(x))));
}
void launch_probe_math(int op, const float* d_in, float* d_out, int n, hipStream_t stream)
{
    if (n > 0) hipLaunchKernelGGL(probe_math_kernel, dim3((n + PTK_BLOCK - 1) / PTK_BLOCK), dim3(PTK_BLOCK), 0, stream, op, d_in, d_out, n);
}

#endif  // !PTK_CONTRACT

struct QueueGeometry { int w[QG_WORDS]; };
static_assert(sizeof(RenderParams) % 4 == 0, "copied word by word");
__device__ __forceinline__ RenderParams* queue_block_params_dev(unsigned* block)
{
    return (RenderParams*)((char*)block + ((8 * PTK_QUEUE_STRIDE + QG_WORDS) * sizeof(unsigned) + 63) / 64 * 64);
}
__global__ void queue_init_kernel(unsigned* block, const QueueGeometry geo, const unsigned* live_count, const RenderParams params)
{
    const int t = threadIdx.x;
    // the launch's parameter block, which the trace kernel reads through the constant address space
    {
        const uint32_t* src = (const uint32_t*)&params;
        uint32_t* dst = (uint32_t*)queue_block_params_dev(block);
        for (int i = t; i < (int)(sizeof(RenderParams) / 4); i += 64) dst[i] = src[i];
    }
    if (t < 8) block[t * PTK_QUEUE_STRIDE] = 0u;
    int w = t < QG_WORDS ? geo.w[t] : 0;
    const int n = (int)*live_count;
    if (t == QG_LIVE_COUNT) w = n;
    if (t == QG_SLOTS) w = (n + 31) / 32 * 4 * geo.w[QG_NUM_CHUNKS];      // per queue: runs of 4 entries, 8 queues
    if (t == QG_QUOTA && w > 0) w = (n * geo.w[QG_NUM_CHUNKS] + w - 1) / w;  // in: blocks of a multi-generation launch; out: items per block
    if (t < QG_WORDS) ((int*)(block + 8 * PTK_QUEUE_STRIDE))[t] = w;
}

#if !PTK_CONTRACT
// Uncached cameras (thin lens; pinhole with opacity textures): can ANY camera ray of this pixel reach the scene?  Every lens ray
// of a pixel starts inside the aperture square around the camera position and passes through the pixel's focal point
// (pathtracer.cpp:785-791; the camera-ray block of trace_kernel): origin o = cam + x right + y up with |x|, |y| <= aperture,
// direction parallel to F - o.  Per axis that is o_k in [cam_k - h_k, cam_k + h_k], d_k in [F_k - cam_k - h_k, F_k - cam_k + h_k]
// with h_k = aperture (|right_k| + |up_k|); taking the two intervals as independent (a superset of the bundle), the ray
// parameters t >= 0 at which SOME such ray is inside the scene's bounding box on axis k form an interval given by two linear
// inequalities; the pixel is dead - black for every sample, never traced, nothing stored - when the three intervals have no
// common point.  Conservative by construction and by margin: the box is padded by 1e-4 of the scene's size and of the camera's
// distance (Moeller-Trumbore accepts nothing measurably outside a triangle, and every triangle lies in the box), the intervals by
// the float rounding of o, F and the normalised direction; evaluated in double, once per camera / frame / scene change.
// Exact: bit-identical images with the cull on and off (tests/test_gpu_host_api.py::test_lens_cull_is_exact).
__device__ bool lens_rays_may_reach_scene(const RenderParams& P, const float4 d0)
{
    double ext = 0.0, far_ = 0.0;
    for (int k = 0; k < 3; k++)
    {
        ext = fmax(ext, (double)P.scene_hi[k] - (double)P.scene_lo[k]);
        far_ = fmax(far_, fmax(fabs((double)P.scene_lo[k] - (double)P.cam_pos[k]), fabs((double)P.scene_hi[k] - (double)P.cam_pos[k])));
    }
    const double pad = 1e-4 * (ext + far_) + 1e-5;
    const double ap = fabs((double)P.aperture) * 1.0001;
    const float dir0[3] = { d0.x, d0.y, d0.z };
    double tlo = 0.0, thi = 1e300;
    bool feasible = true;
    for (int k = 0; k < 3; k++)
    {
        const float Ff = P.cam_pos[k] + dir0[k] * P.focal_dist;            // the focal point as the camera-ray block computes it
        const double oc = (double)P.cam_pos[k], F = (double)Ff;
        const double h = ap * (fabs((double)P.cam_right[k]) + fabs((double)P.cam_up[k])) + 1e-6 * fabs(oc);
        const double dc = F - oc, hd = h + 1e-6 * (fabs(F) + fabs(oc) + fabs(dc));
        const double lo = (double)P.scene_lo[k] - pad, hi = (double)P.scene_hi[k] + pad;
        // the smallest coordinate any ray of the bundle has at parameter t must not exceed hi, the largest not fall short of lo
        const double a1 = (oc - h) - hi, b1 = dc - hd;                     // a1 + t b1 <= 0
        const double a2 = lo - (oc + h), b2 = -(dc + hd);                  // a2 + t b2 <= 0
        if (b1 > 0.0) thi = fmin(thi, -a1 / b1); else if (b1 < 0.0) tlo = fmax(tlo, -a1 / b1); else if (a1 > 0.0) feasible = false;
        if (b2 > 0.0) thi = fmin(thi, -a2 / b2); else if (b2 < 0.0) tlo = fmax(tlo, -a2 / b2); else if (a2 > 0.0) feasible = false;
    }
    return feasible && tlo <= thi * (1.0 + 1e-9) + 1e-12;
}

// Which pixels of every owned 8x8 quadrant need tracing: on the image, and - when the camera ray's closest hit is
// cached - not a miss (pathtracer.cpp:550: such a pixel is black for every sample); uncached cameras: not a pixel whose
// lens rays all miss the scene's bounds (above).  One wave per quadrant.
__global__ __launch_bounds__(PTK_BLOCK) void live_mask_kernel(const RenderParams P, unsigned long long* mask, int num_subtiles)
{
    const int subtile = blockIdx.x * (PTK_BLOCK / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (subtile >= num_subtiles) return;
    const int owned = subtile >> 2, quad = subtile & 3;
    const int tile = owned * P.world + P.rank;
    bool live = false;
    if (tile < P.num_tiles)
    {
        const int ty = tile / P.tiles_x, tx = (tile % P.tiles_x + P.tiles_x - (3 * ty) % P.tiles_x) % P.tiles_x;
        const int px = tx * PTK_TILE + (quad & 1) * 8 + (lane & 7), py = ty * PTK_TILE + (quad >> 1) * 8 + (lane >> 3);
        live = px < P.width && py < P.height;
        if (live && P.primary_hit) live = __float_as_int(P.primary_hit[(size_t)py * P.width + px].x) != PTK_NOHIT;
        else if (live && P.lens_cull) live = lens_rays_may_reach_scene(P, P.primary[(size_t)py * P.width + px]);
    }
    const unsigned long long m = __ballot(live);
    if (lane == 0) mask[subtile] = m;
}

// Ordered list of the quadrants that have live pixels (single workgroup: a few hundred thousand quadrants at most).
__global__ __launch_bounds__(1024) void live_compact_kernel(const unsigned long long* mask, int num_subtiles, unsigned* list, unsigned* count)
{
    __shared__ unsigned wave_total[16];
    __shared__ unsigned base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int s0 = 0; s0 < num_subtiles; s0 += 1024)
    {
        const int sidx = s0 + t;
        const bool live = sidx < num_subtiles && mask[sidx] != 0ull;
        const unsigned long long b = __ballot(live);
        if (lane == 0) wave_total[wave] = (unsigned)__popcll(b);
        __syncthreads();
        unsigned before = base;
        for (int w = 0; w < wave; w++) before += wave_total[w];
        if (live) list[before + (unsigned)__popcll(b & ((1ull << lane) - 1ull))] = (unsigned)sidx;
        __syncthreads();
        if (t == 0) { unsigned sum = 0; for (int w = 0; w < 16; w++) sum += wave_total[w]; base += sum; }
        __syncthreads();
    }
    if (t == 0) *count = base;
}

__global__ __launch_bounds__(PTK_BLOCK) void pixel_rng_kernel(uint32_t seed_lo, uint32_t seed_hi, int n, uint2* out)
{
    const int i = blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t pkey = pixel_key(seed_lo, seed_hi, (uint32_t)i);
    out[i] = make_uint2(pkey, (hash32(pkey ^ 0x9E3779B9u) << 1) | 1u);
}
void launch_pixel_rng(uint32_t seed_lo, uint32_t seed_hi, int n, uint2* out, hipStream_t stream)
{
    if (n > 0) hipLaunchKernelGGL(pixel_rng_kernel, dim3((n + PTK_BLOCK - 1) / PTK_BLOCK), dim3(PTK_BLOCK), 0, stream, seed_lo, seed_hi, n, out);
}

void launch_live_list(const RenderParams& p, int num_subtiles, unsigned long long* mask, unsigned* list, unsigned* count, hipStream_t stream)
{
    if (num_subtiles <= 0) return;
    const int per_block = PTK_BLOCK / 64;
    hipLaunchKernelGGL(live_mask_kernel, dim3((num_subtiles + per_block - 1) / per_block), dim3(PTK_BLOCK), 0, stream, p, mask, num_subtiles);
    hipLaunchKernelGGL(live_compact_kernel, dim3(1), dim3(1024), 0, stream, (const unsigned long long*)mask, num_subtiles, list, count);
}

#endif  // !PTK_CONTRACT

void launch_trace(const RenderParams& p0, int num_subtiles, int resident_waves, hipStream_t stream, bool stats)
{
    if (num_subtiles <= 0) return;
#if PTK_CONTRACT
    stats = false;                  // the counters belong to the exact build
#endif
    RenderParams p = p0;
    // the items = (entry of the live list, chunk) go into 8 queues (one per XCD group) in runs of four entries; the
    // host only knows an upper bound of the list's length (every quadrant live), the device the real one
    const int padded = (num_subtiles + 31) / 32 * 32 * p.num_chunks;
    // queue block = 8 zeroed slot counters (one per 128-B line) followed by the launch geometry the item set-up reads;
    // written by a one-wave kernel from its own arguments (stream-ordered, nothing for the host to wait on)
    QueueGeometry geo = {};
    geo.w[QG_NUM_CHUNKS] = p.num_chunks; geo.w[QG_WORLD] = p.world; geo.w[QG_RANK] = p.rank;
    geo.w[QG_TILES_X] = p.tiles_x; geo.w[QG_CHUNK] = p.chunk; geo.w[QG_SPP] = (int)p.spp;
    // big launches: persistent waves, as many one-wave workgroups as the chip holds at once, each pulling items until
    // none is left; small ones: a wave per (possible) item, the dispatcher balances those better
    if (!(p.flat_count > 0) && PTK_TRACE_WAVES_BVH != 4) resident_waves = resident_waves / 16 * 4 * PTK_TRACE_WAVES_BVH;
    if (p.flat_count > 0 && PTK_TRACE_WAVES != 4) resident_waves = resident_waves / 16 * 4 * PTK_TRACE_WAVES;
    if (p.persistent < 0) p.persistent = padded > 4 * resident_waves ? 1 : 0;
    const int generations = p.persistent ? std::max(1, std::min(p.generations, padded / resident_waves)) : 1;
    const int blocks = p.persistent ? resident_waves * generations : padded;
    geo.w[QG_QUOTA] = generations > 1 ? blocks : 0;
    hipLaunchKernelGGL(queue_init_kernel, dim3(1), dim3(64), 0, stream, p.queues, geo, p.live_count, p);
    const RenderParams* dp = queue_block_params(p.queues);
    const bool flat = p.flat_count > 0;
#if !PTK_CONTRACT
    if (stats && flat) hipLaunchKernelGGL((trace_kernel<true, true>), dim3(blocks), dim3(PTK_TRACE_BLOCK), 0, stream, dp);
    else if (stats) hipLaunchKernelGGL((trace_kernel<true, false>), dim3(blocks), dim3(PTK_TRACE_BLOCK), 0, stream, dp);
    else
#endif
    if (flat) hipLaunchKernelGGL((trace_kernel<false, true>), dim3(blocks), dim3(PTK_TRACE_BLOCK), 0, stream, dp);
    else hipLaunchKernelGGL((trace_kernel<false, false>), dim3(blocks), dim3(PTK_TRACE_BLOCK), 0, stream, dp);
}

#if !PTK_CONTRACT
// ---- multi-GPU exchange step: packed form of the float accumulator ---------------------------------
// Packed layout of rank r of `world` (include/ptk.h ptk_packed_layout): its owned tiles in ascending tile order, 768
// floats each = the tile's 16 x 16 pixels row-major from the tile's top-left, RGB; pixels off the image hold 0.
// pack: accumulator -> packed (one workgroup per owned tile, 768 B contiguous per wave-store);
// unpack: the packed buffers of ALL ranks (rank r's starts at float offset base[r]) -> full image, one workgroup per tile.
struct ExchangeBases { long long base[PTK_MAX_RANKS]; };

__global__ __launch_bounds__(PTK_BLOCK) void pack_owned_kernel(const float* __restrict__ accum, float* __restrict__ packed, int width, int height,
                                                               int tiles_x, int num_tiles, int rank, int world)
{
    const int owned = blockIdx.x, tile = owned * world + rank;
    if (tile >= num_tiles) return;
    const int ty = tile / tiles_x, tx = (tile % tiles_x + tiles_x - (3 * ty) % tiles_x) % tiles_x;
    const int p = threadIdx.x, px = tx * PTK_TILE + (p & 15), py = ty * PTK_TILE + (p >> 4);
    float r = 0.0f, g = 0.0f, b = 0.0f;
    if (px < width && py < height)
    {
        const size_t a = ((size_t)(height - 1 - py) * width + px) * 3;
        r = accum[a]; g = accum[a + 1]; b = accum[a + 2];
    }
    float* o = packed + (size_t)owned * (PTK_BLOCK * 3) + p * 3;
    o[0] = r; o[1] = g; o[2] = b;
}

__global__ __launch_bounds__(PTK_BLOCK) void unpack_all_kernel(const float* __restrict__ packed, const ExchangeBases bases, float* __restrict__ image,
                                                               int width, int height, int tiles_x, int num_tiles, int world)
{
    const int tile = blockIdx.x;
    if (tile >= num_tiles) return;
    const int rank = tile % world, owned = tile / world;
    const int ty = tile / tiles_x, tx = (tile % tiles_x + tiles_x - (3 * ty) % tiles_x) % tiles_x;
    const int p = threadIdx.x, px = tx * PTK_TILE + (p & 15), py = ty * PTK_TILE + (p >> 4);
    if (px >= width || py >= height) return;
    const float* in = packed + bases.base[rank] + (size_t)owned * (PTK_BLOCK * 3) + p * 3;
    const size_t a = ((size_t)(height - 1 - py) * width + px) * 3;
    image[a] = in[0]; image[a + 1] = in[1]; image[a + 2] = in[2];
}

void launch_pack_owned(const float* accum, float* packed, int width, int height, int rank, int world, hipStream_t stream)
{
    const int tiles_x = (width + PTK_TILE - 1) / PTK_TILE, num_tiles = tiles_x * ((height + PTK_TILE - 1) / PTK_TILE);
    const int owned = num_tiles <= rank ? 0 : (num_tiles - rank + world - 1) / world;
    if (owned > 0) hipLaunchKernelGGL(pack_owned_kernel, dim3(owned), dim3(PTK_BLOCK), 0, stream, accum, packed, width, height, tiles_x, num_tiles, rank, world);
}
void launch_unpack_all(const float* packed, const long long* bases, float* image, int width, int height, int world, hipStream_t stream)
{
    const int tiles_x = (width + PTK_TILE - 1) / PTK_TILE, num_tiles = tiles_x * ((height + PTK_TILE - 1) / PTK_TILE);
    ExchangeBases b = {};
    for (int r = 0; r < world && r < PTK_MAX_RANKS; r++) b.base[r] = bases[r];
    hipLaunchKernelGGL(unpack_all_kernel, dim3(num_tiles), dim3(PTK_BLOCK), 0, stream, packed, b, image, width, height, tiles_x, num_tiles, world);
}

void launch_accumulate(const RenderParams& p, int owned_tiles, hipStream_t stream)
{
    if (owned_tiles <= 0) return;
    hipLaunchKernelGGL(accumulate_kernel, dim3(owned_tiles), dim3(PTK_BLOCK), 0, stream, p);
}
void launch_primary_hits(const RenderParams& p, float4* out, float4* out_rd, hipStream_t stream)
{
    int n = p.width * p.height;
    hipLaunchKernelGGL(primary_hits_kernel, dim3((n + PTK_BLOCK - 1) / PTK_BLOCK), dim3(PTK_BLOCK), 0, stream, p, out, out_rd);
}
void launch_primary(const PrimaryParams& p, hipStream_t stream)
{
    int threads = 64;
    hipLaunchKernelGGL(primary_dirs_kernel, dim3((p.height + threads - 1) / threads), dim3(threads), 0, stream, p);
}
void launch_probe(const ProbeParams& p, hipStream_t stream)
{
    if (p.n <= 0) return;
    hipLaunchKernelGGL(probe_hits_kernel, dim3((p.n + PTK_BLOCK - 1) / PTK_BLOCK), dim3(PTK_BLOCK), 0, stream, p);
}

#endif  // !PTK_CONTRACT

#if PTK_CONTRACT
}  // namespace fma / fast
#endif
}  // namespace ptk
