// Shared between the host API (ptk_api.hip) and the kernels (ptk_kernels.hip): device record layouts
// and the by-value kernel parameter block.  All records are float4-aligned so one lane fetches a
// record with 16-byte loads (global_load_dwordx4).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptk.h"

namespace ptk {

// BVH4 node, 64 B = one dependent fetch per FOUR child boxes (round 1's binary walk ran at the rate at which the vector
// memory pipeline gathers records, profiles/r02/gather_ceiling.json: records per ray are what counted).
// Child boxes are quantised outward to 8 bits on a per-node grid: plane = origin + q * scale.
//   q0 = (origin.xyz, scale.x)   q1 = (scale.yz, bits link0, bits link1)
//   q2 = (bits link2, bits link3, bits lo.x, bits lo.y)   q3 = (bits lo.z, bits hi.x, bits hi.y, bits hi.z)
// lo.a / hi.a: byte k = child k's low / high plane on axis a; an empty slot has lo = 255, hi = 0 (never entered).
// link >= 0: interior node index; link < 0: leaf, ~link = (first_record << 3) | (count - 1)
constexpr int NODE_F4 = 4;
constexpr int LEAF_MAX = 8;
constexpr int32_t NODE_EXIT = INT32_MIN;

// Triangle intersection record, 48 B, stored in BVH leaf order:
//   t0 = (v0.xyz, e1.x)  t1 = (e1.yz, e2.xy)  t2 = (e2.z, bits tri_index, bits opacity_tex, 0)
constexpr int TRI_F4 = 3;

// Shading record, 112 B, indexed by the scene's triangle index (fetched only for the accepted hit):
//   s0 = (normal.xyz, bits (material | smoothing << 31))
//   s1 = (uv1.xy, uv2.xy)  s2 = (uv3.xy, n1.xy)  s3 = (n1.z, n2.xyz)  s4 = (n3.xyz, tangent.x)
//   s5 = (tangent.yz, bitangent.xy)  s6 = (bitangent.z, 0, 0, 0)
constexpr int SHADE_F4 = 7;

// Material record, 96 B:
//   m0 = (diffuse.rgb, bits type)  m1 = (specular.rgb, emissiveIntensity)  m2 = (emissive.rgb, roughness)
//   m3 = (reflectiveness, translucency, ior, rr_prob)  m4 = bits tex[0..3]  m5 = bits (tex[4], tex[5], any_tex, 0)
constexpr int MAT_F4 = 6;

// Light record, 64 B:  l0 = (v1.xyz, bits tri)  l1 = (v2.xyz, c.r)  l2 = (v3.xyz, c.g)  l3 = (c.b, bits opacity_tex, 0, 0)
// with c = emissive * emissiveIntensity of the light's material (pathtracer.cpp:528); tri and opacity_tex let the shadow walk
// rebuild the light triangle's own intersection record (v1, v2 - v1, v3 - v1: the record packers' subtractions) and test it first
constexpr int LIGHT_F4 = 4;

// Radiance samples travel from the trace kernel to the accumulate kernel through HBM:
//   samples[(item * chunk + s) * 64 + lane] = (r, g, b, 0) of sample s of the item's chunk for the pixel
//   of `lane`; item = (owned 16x16 tile * 4 + 8x8 quadrant) * num_chunks + chunk index.
// The accumulate kernel folds them into the float accumulator strictly in sample order, which keeps
// the reference's `mTotalImg[px] += color` once per RenderFrame() semantics (pathtracer.cpp:798-800)
// while the tracing itself is parallel over pixels AND samples.
#define PTK_MAX_RANKS 64         // ranks of one exchange group (one node: 8)
#define PTK_QUEUE_STRIDE 32     // one queue counter per 128-B line
// words of the launch geometry that follows the 8 counters in the queue block
enum { QG_NUM_CHUNKS = 0, QG_WORLD, QG_RANK, QG_TILES_X, QG_CHUNK, QG_SPP, QG_SLOTS, QG_LIVE_COUNT, QG_QUOTA, QG_WORDS = 16 };

struct RenderParams {
    float4* samples;
    int chunk, num_chunks;      // samples per work item, work items per 8x8 tile
    int num_items;              // owned tiles * 4 * num_chunks
    int max_batch;              // most slots a wave pops from a queue at once
    int generations;            // persistent launches: waves retire after 1/generations of their share (1 = never)
    int persistent;             // 1: resident waves pull items from the queues; 0: one item per wave, named by blockIdx
    const unsigned long long* live_mask;   // per owned quadrant: its pixels that need tracing
    const unsigned* live_list;  // quadrants with a non-zero mask, ascending; *live_count entries
    const unsigned* live_count;
    unsigned* queues;           // queue block: 8 slot counters PTK_QUEUE_STRIDE words apart + QG_WORDS of launch geometry
    int shade_thr, gen_thr;     // scheduling lambdas (eighths): cost of the shade / gen block in walk steps
    int tri_thr;                // triangle arm runs when lanes with a queued triangle >= tri_thr/8 x lanes with a node
    const float4* nodes;
    const float4* tris;
    const float4* flat_tris;    // scenes of <= 16 triangles: the same records in ascending triangle index (FLAT kernel)
    const float4* shade;
    const float4* mats;
    const float4* lights;
    const int4* texinfo;        // (width, height, first texel index, 0)
    const uint32_t* texels;     // RGBA8 atlas as packed words (r = low byte)
    const float4* primary;      // [H][W] unit primary directions before DOF (top-down rows)
    const uint2* pixel_rng;     // [H][W] (pixel key, PCG increment) of the pixel's RNG streams for this launch's seed
    const float4* primary_hit;  // [H][W] (bits tri | PTK_NOHIT, t, u, v) of the camera ray, or null when not cacheable
    const float4* primary_rd;   // [H][W] its unit direction (valid with primary_hit)
    float* accum;               // [H][W][3] float RGB, rows bottom-up (mTotalImg)
    uint8_t* rgb8;              // [H][W][3] RGB8, rows bottom-up (mOutImg)
    uint8_t* rgb8_host;         // the caller's page-locked hand-off buffer, device-visible (ptk_bind_out_image), or null
    int rgb8_host_full;         // 1: every pixel is written there; 0: pixels whose cached camera ray misses are skipped (they hold 0)
    const uint32_t* exit_flag;  // holds the generation of the newest render an Exit() was aimed at (0: none)
    uint32_t exit_gen;          // this render's generation: its kernels stand down when *exit_flag >= exit_gen (Exit() cuts everything in flight)
    unsigned long long* stats;  // 7 counters (STATS variant only)
    int num_nodes, num_lights;
    float scene_bound;          // 3.1 x the largest |coordinate| of the scene's padded bounds: the per-ray slack of the slab tests (Walk::begin)
    float scene_lo[3], scene_hi[3];   // the vertices' bounding box (live_mask_kernel: pixels none of whose camera rays can reach it are never traced)
    int lens_cull;              // 1: uncached cameras (thin lens, opacity textures) cull such pixels from the live mask
    int flat_shade_w, flat_gen_w; // FLAT block-choice weights (eighths) of the shade / camera-ray blocks vs the triangle pass
    int flat_count;             // > 0: tiny scene, test all flat_count triangle records per ray without a BVH walk
    int width, height, max_depth;
    int tiles_x, num_tiles;
    int rank, world;
    uint32_t first_sample, spp;
    uint32_t seed_lo, seed_hi;
    float cam_pos[3], cam_right[3], cam_up[3];
    float focal_dist, aperture;
    float resolve_samples;      // (float)(first_sample + spp) = (float)mSamples
};

struct PrimaryParams {
    float4* primary;
    int width, height;
    float cam_pos[3], cam_right[3], cam_up[3], top_left[3];
    float delta_x, delta_y;
};

struct ProbeParams {
    const float4* nodes;
    const float4* tris;
    const float4* flat_tris;    // scenes of <= 16 triangles: the same records in ascending triangle index (FLAT kernel)
    const float4* shade;
    const float4* mats;
    const int4* texinfo;
    const uint32_t* texels;
    const float* ro;
    const float* rd;
    int32_t* tri;
    float* tuv;
    int n, num_nodes;
    float scene_bound;
    const float4* lights;       // probe_direct only
    int num_lights;
};

// Queue block of one trace launch: 8 slot counters (one per 128-B line) + QG_WORDS of launch geometry + the launch's
// RenderParams.  The trace kernels take the parameters BY POINTER into this block (constant address space: every field is
// an s_load where it is used) instead of by value: a by-value RenderParams is preloaded whole into SGPRs, which cost the
// product kernels 30 / 42 spilled SGPRs (v_writelane / v_readlane at the head of the shade block) and three to five VGPRs.
constexpr size_t PTK_QUEUE_BLOCK_BYTES = ((8 * PTK_QUEUE_STRIDE + QG_WORDS) * sizeof(unsigned) + 63) / 64 * 64 + (sizeof(RenderParams) + 63) / 64 * 64;
inline const RenderParams* queue_block_params(const unsigned* block) { return (const RenderParams*)((const char*)block + ((8 * PTK_QUEUE_STRIDE + QG_WORDS) * sizeof(unsigned) + 63) / 64 * 64); }
void launch_trace(const RenderParams& p, int num_subtiles, int resident_waves, hipStream_t stream, bool stats);
// the same trace kernels from the second and third builds of ptk_kernels.hip (-ffp-contract=fast; `fast` also with the
// hardware's 1-ulp reciprocal / square root): the "contract" option, results within tolerance instead of bit-exact
namespace fma { void launch_trace(const RenderParams& p, int num_subtiles, int resident_waves, hipStream_t stream, bool stats); }
namespace fast { void launch_trace(const RenderParams& p, int num_subtiles, int resident_waves, hipStream_t stream, bool stats); }
void launch_pixel_rng(uint32_t seed_lo, uint32_t seed_hi, int n, uint2* out, hipStream_t stream);
void launch_live_list(const RenderParams& p, int num_subtiles, unsigned long long* mask, unsigned* list, unsigned* count, hipStream_t stream);
void launch_accumulate(const RenderParams& p, int owned_tiles, hipStream_t stream);
void launch_pack_owned(const float* accum, float* packed, int width, int height, int rank, int world, hipStream_t stream);
void launch_unpack_all(const float* packed, const long long* bases, float* image, int width, int height, int world, hipStream_t stream);
void launch_primary(const PrimaryParams& p, hipStream_t stream);
void launch_primary_hits(const RenderParams& p, float4* out, float4* out_rd, hipStream_t stream);
void launch_probe(const ProbeParams& p, hipStream_t stream);
void launch_probe_direct(const ProbeParams& p, const float* pts, const float* nrm, const float* dif, const float* tape, float* out, hipStream_t stream);
void launch_probe_math(int op, const float* d_in, float* d_out, int n, hipStream_t stream);

}  // namespace ptk
