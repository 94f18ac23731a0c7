/* TEST INFRASTRUCTURE ONLY — never imported, linked or executed by the product path.
 *
 * CPU restatement (plain C99) of the reference's per-pixel render loop
 *   PathTracer::RenderFrame -> Trace -> Hit -> {IntersectTriangle, Image::tex2D, DirectIllumimation}
 *   (reference PathTracing/src/pathtracer.cpp:367-822, mesh.cpp:48-83, image.cpp:63-86)
 * used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker for the HIP
 * kernels.  Parity pinning: tests/golden/ holds vectors generated from the REAL reference
 * (oracle/_ref, built from /root/reference by oracle/Makefile.ref) with oracle/gen_golden.py; the
 * "not gpu" tests replay them against this file (tiers K / T / S of SURVEY.md §8c4).
 *
 * Deliberate, documented differences from the reference (none changes the estimator):
 *   - Trace is iterative (L += T*(emission + direct); T *= weight) — exact for a recursion that is
 *     linear in the recursive term; differs from the recursion by float rounding only (<=1e-5 rel).
 *   - The RNG is counter-based PCG (key = seed, pixel, sample) instead of one raced std::mt19937;
 *     a "tape" mode replays the reference's own draw stream for the tier-T vectors.
 *   - Closest hit is the minimum over all accepted triangles with the order-independent tie rule
 *     (smaller t, then smaller triangle index); the reference's tie depends on its random tree.
 *   - Stochastic opacity is evaluated only for candidates nearer than the current best hit, each
 *     with an independent hashed draw (same distribution of the closest accepted hit).
 *   - tex2D clamps the texel coordinate to the image (the reference reads one texel past a row when
 *     u rounds to 1.0 after the wrap, image.cpp:71-77).
 *   - sin/cos use the polynomial below instead of libm (<=1 ulp from cosf/sinf) so that this file
 *     and the HIP kernel can be bit-identical.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same memory layout as include/ptk.h (the drop-in boundary's data format), restated here so the
 * oracle does not include product headers. */
typedef struct {
    int32_t type;              /* 0 OPAQUE, 1 TRANSLUCENT (mesh.h:15-19) */
    float diffuse[3];
    float specular[3];
    float emissive[3];
    float emissive_intensity;
    float roughness;
    float reflectiveness;
    float translucency;
    float ior;
    int32_t tex[6];            /* diffuse, normal, emissive, roughness, metallic, opacity; -1 none */
} orc_material;                /* 84 bytes */

typedef struct {
    int32_t width, height;
    int64_t offset;            /* byte offset of the RGBA8 texels in the atlas */
} orc_texture;

typedef struct {
    int32_t num_triangles;
    const float* verts;        /* [N][9]  v1 v2 v3 (world space) */
    const float* normals;      /* [N][9]  n1 n2 n3 */
    const float* uvs;          /* [N][6]  uv1 uv2 uv3 */
    const float* tbn;          /* [N][9]  normal, tangent, bitangent (Triangle::Init) */
    const uint8_t* smoothing;  /* [N] */
    const int32_t* material;   /* [N] index into materials */
    int32_t num_materials;
    const orc_material* materials;
    int32_t num_textures;
    const orc_texture* textures;
    const uint8_t* texels;
    int64_t texel_bytes;
    int32_t num_lights;
    const int32_t* lights;     /* triangle indices with |emissive| >= EPS (pathtracer.cpp:267-273) */
} orc_scene_desc;

typedef struct {
    float pos[3], dir[3], up[3];   /* dir/up already normalised (SetCamera, pathtracer.cpp:333-338) */
    float focal, fovy;             /* SetProjection (already clamped) */
    float focal_dist, aperture;
} orc_camera;

typedef struct orc_scene orc_scene;

orc_scene* orc_create(const orc_scene_desc* desc);   /* copies everything, builds its own BVH */
void orc_destroy(orc_scene* s);

/* Render samples [first_sample, first_sample+spp) of every pixel, OpenMP over rows.
 * total: float RGB W*H*3, bottom-up rows (the reference's mTotalImg layout, pathtracer.cpp:796);
 * it is READ and accumulated into, sample by sample, like successive RenderFrame() calls.
 * rgb8 (may be NULL): clamp(total/(first_sample+spp))*255 truncated (pathtracer.cpp:802-812).
 * rank/world: only pixels of tiles owned by `rank` are rendered (tile = 16x16 pixels, owner of tile
 * (tx, ty) = (ty*tiles_x + (tx + 3*ty) % tiles_x) % world, as include/ptk.h ptk_set_tile).  threads<=0: OpenMP default. */
void orc_render(const orc_scene* s, const orc_camera* cam, int width, int height, int max_depth,
                uint32_t first_sample, uint32_t spp, uint64_t seed, int rank, int world,
                float* total, uint8_t* rgb8, int threads);

/* Primary ray directions before DOF, row-major top-down [H][W][3]; follows the incremental
 * `pixel += camRight*deltaX` arithmetic of pathtracer.cpp:755-766,782-785,814. */
void orc_primary_dirs(const orc_camera* cam, int width, int height, float* out);
int orc_render_tape(const orc_scene* s, const orc_camera* cam, int width, int height, int max_depth, const float* tape, int tape_len, float* total);

/* Radiance of one path with the draws taken from `tape` (tier T).  Returns #draws consumed. */
int orc_trace_tape(const orc_scene* s, const float* ro, const float* rd, int max_depth,
                   const float* tape, int tape_len, int mode, float* out3);
/* mode 0 = iterative form (what orc_render and the HIP kernel compute), 1 = recursive form with
 * DirectIllumimation evaluated before the recursive Trace, 2 = recursive, Trace first (the order
 * g++ picked for the reference build the fixtures were recorded from). */
void orc_trace_counter(const orc_scene* s, const float* ro, const float* rd, int max_depth,
                       uint64_t seed, uint32_t pixel, uint32_t sample, int mode, float* out3);

/* Per-function probes (tier K) */
void orc_intersect_triangle(const float* ro, const float* rd, const float* v0, const float* v1,
                            const float* v2, float* out3);
int  orc_hit(const orc_scene* s, const float* ro, const float* rd, float* tuv, int32_t* tri);
int  orc_hit_brute(const orc_scene* s, const float* ro, const float* rd, float* tuv, int32_t* tri);
void orc_tex2d(const orc_scene* s, int tex, float u, float v, float* out4);
void orc_triangle_init(const float* in15, float* out9);
void orc_sincos(float a, float* s, float* c);
float orc_rand_u01(uint64_t seed, uint32_t pixel, uint32_t sample, int n);  /* n-th draw of a path */
void orc_sample_circle(float r1, float r2, float* out2);
int  orc_aabb_intersect(const float* bmin, const float* bmax, const float* ro, const float* rd);   /* AABB::Intersect */
void orc_aabb_build(const float* pts, int n, float* out6);                                         /* AABB::Build + Check */
void orc_direct_illumination_tape(const orc_scene* s, const float* p, const float* n, const float* diffuse, const float* tape3, float* out3);

/* traversal statistics of the oracle's own BVH (diagnostics only) */
void orc_bvh_info(const orc_scene* s, int32_t* nodes, int32_t* depth);

#ifdef __cplusplus
}
#endif
#endif
