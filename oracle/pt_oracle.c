/* TEST INFRASTRUCTURE ONLY — see pt_oracle.h.  Plain C99 restatement of the reference's render loop.
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/PathTracing/src/).  Compile with -ffp-contract=off and WITHOUT -ffast-math: the
 * float expression order below is the parity contract with the HIP kernel. */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_EPS 0.00001f           /* mesh.h:12 */
#define ORC_PI_D 3.14159265358979323846
#define ORC_FLT_EPSILON 1.1920928955078125e-7f
#define ORC_TILE 16

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* glm 0.9.3.1 core/func_geometric.inl:161-171 */
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* glm core/func_geometric.inl:200-211 */
static inline v3 cross(v3 x, v3 y)
{
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* glm normalize = x * inversesqrt(dot(x,x)), inversesqrt = 1/sqrt (func_geometric.inl:239-248,
 * func_exponential.inl:145-153) */
static inline v3 normalize(v3 a)
{
    float sqr = a.x * a.x + a.y * a.y + a.z * a.z;
    float inv = 1.0f / sqrtf(sqr);
    return muls(a, inv);
}
/* glm reflect = I - N * dot(N, I) * 2 (func_geometric.inl:276-283) */
static inline v3 reflect(v3 I, v3 N)
{
    float d = dot(N, I);
    return sub(I, muls(muls(N, d), 2.0f));
}

/* ---- sin/cos on [0, 2*pi] -------------------------------------------------------------------
 * Replaces libm cosf/sinf (pathtracer.cpp:610, :738) by a fixed polynomial so that CPU and GPU
 * agree bit for bit.  Reduction in double, cephes single-precision kernels on [-pi/4, pi/4]. */
#ifdef ORC_LIBM_SINCOS
/* (tools/fuzz_trace_vs_reference.py --libm: libm's own sinf / cosf, as the reference calls them - with these the recursion below
 * reproduces the reference's radiance BIT FOR BIT; the polynomial is what the product computes with, DESIGN.md section 2, difference 6) */
void orc_sincos(float a, float* s, float* c) { *s = sinf(a); *c = cosf(a); }
#else
void orc_sincos(float a, float* s, float* c)
{
    int k = (int)(a * 0.636619772367581343f + 0.5f);
    float r = (float)((double)a - (double)k * 1.57079632679489661923);
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    switch (k & 3)
    {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
    }
}
#endif

/* ---- RNG -------------------------------------------------------------------------------------
 * Replaces PathTracer::Rand (pathtracer.cpp:367-371; one std::mt19937 raced by all workers) by a
 * counter-based PCG-RXS-M-XS-32 stream per (seed, pixel, sample).  Tape mode replays recorded
 * reference draws. */
static inline uint32_t pcg_out(uint32_t st)
{
    uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
    return (w >> 22u) ^ w;
}
static inline uint32_t hash32(uint32_t x) { return pcg_out(x * 747796405u + 2891336453u); }

typedef struct {
    uint32_t state, inc, key;      /* counter mode */
    const float* tape; int tape_len, tape_pos;   /* tape mode when tape != NULL */
} rng_t;

static inline uint32_t pixel_key(uint64_t seed, uint32_t pixel)
{
    uint32_t a = hash32((uint32_t)(seed >> 32));
    uint32_t b = hash32((uint32_t)seed + a);
    return hash32(pixel + b);
}
static inline void rng_init(rng_t* r, uint32_t pkey, uint32_t sample)
{
    r->state = hash32(sample + pkey);
    r->inc = (hash32(pkey ^ 0x9E3779B9u) << 1) | 1u;
    r->key = r->state;
    r->tape = 0; r->tape_len = 0; r->tape_pos = 0;
}
static inline float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }
static inline float rnd(rng_t* r)
{
    if (r->tape)
    {
        float v = r->tape_pos < r->tape_len ? r->tape[r->tape_pos] : 0.5f;
        r->tape_pos++;
        return v;
    }
    uint32_t old = r->state;
    r->state = old * 747796405u + r->inc;
    return u01(pcg_out(old));
}
/* independent draw for the stochastic-opacity test of triangle `tri` on ray number `ray` of this
 * path: order-independent replacement for the Rand() at pathtracer.cpp:475 */
static inline float rnd_opacity(rng_t* r, uint32_t ray, uint32_t tri)
{
    if (r->tape) return rnd(r);
    return u01(hash32(tri + hash32(ray + r->key)));
}
float orc_rand_u01(uint64_t seed, uint32_t pixel, uint32_t sample, int n)
{
    rng_t r; float v = 0.0f;
    rng_init(&r, pixel_key(seed, pixel), sample);
    for (int i = 0; i <= n; i++) v = rnd(&r);
    return v;
}

/* ---- scene ------------------------------------------------------------------------------------*/
typedef struct { float bmin[3], bmax[3]; int32_t left, right, first, count; } node_t;

struct orc_scene {
    int32_t nt;
    float *verts, *normals, *uvs, *tbn;
    uint8_t* smoothing;
    int32_t* material;
    int32_t nm; orc_material* mats;
    int32_t ntex; orc_texture* tex; uint8_t* texels; int64_t texel_bytes;
    int32_t nl; int32_t* lights;
    /* own BVH */
    node_t* nodes; int32_t nnodes, cap; int32_t* order; int32_t depth;
};

static void* dup(const void* p, size_t n)
{
    void* q = malloc(n ? n : 1);
    if (n && p) memcpy(q, p, n);
    return q;
}

/* ---- Moeller-Trumbore, PathTracer::IntersectTriangle pathtracer.cpp:373-409 -----------------*/
static inline int intersect_triangle(v3 ro, v3 rd, v3 v0, v3 v1, v3 v2, float* t, float* u, float* v)
{
    v3 edge1 = sub(v1, v0);
    v3 edge2 = sub(v2, v0);
    v3 h = cross(rd, edge2);
    float a = dot(edge1, h);
    if (fabsf(a) < ORC_EPS) return 0;
    float f = 1.0f / a;
    v3 s = sub(ro, v0);
    float uu = f * dot(s, h);
    if (uu < 0.0f || uu > 1.0f) return 0;
    v3 q = cross(s, edge1);
    float vv = f * dot(rd, q);
    if (vv < 0.0f || uu + vv > 1.0f) return 0;
    float tt = f * dot(edge2, q);
    if (tt > ORC_EPS) { *t = tt; *u = uu; *v = vv; return 1; }
    return 0;
}
void orc_intersect_triangle(const float* ro, const float* rd, const float* v0, const float* v1,
                            const float* v2, float* out3)
{
    float t, u, v;
    if (intersect_triangle(ld3(ro), ld3(rd), ld3(v0), ld3(v1), ld3(v2), &t, &u, &v)) { out3[0] = t; out3[1] = u; out3[2] = v; }
    else out3[0] = out3[1] = out3[2] = 0.0f;
}

/* ---- Image::tex2D image.cpp:63-86 -----------------------------------------------------------*/
static inline void tex2d(const orc_scene* s, int tex, float uvx, float uvy, float* out4)
{
    if (tex < 0 || tex >= s->ntex || s->tex[tex].width <= 0 || s->tex[tex].height <= 0)
    { out4[0] = out4[1] = out4[2] = out4[3] = 0.0f; return; }
    const orc_texture* T = &s->tex[tex];
    float u = fmodf(uvx, 1.0f);
    float v = fmodf(uvy, 1.0f);
    if (u < 0.0f) u += 1.0f;
    if (v < 0.0f) v += 1.0f;
    int cx = (int)((float)T->width * u);
    int cy = (int)((float)T->height * v);
    if (cx > T->width - 1) cx = T->width - 1;     /* clamp: see header (reference over-reads) */
    if (cy > T->height - 1) cy = T->height - 1;
    if (cx < 0) cx = 0;                            /* NaN uv */
    if (cy < 0) cy = 0;
    const uint8_t* p = s->texels + T->offset + 4 * ((int64_t)cy * T->width + cx);
    out4[0] = (float)p[0] / 255.0f;
    out4[1] = (float)p[1] / 255.0f;
    out4[2] = (float)p[2] / 255.0f;
    out4[3] = (float)p[3] / 255.0f;
}
void orc_tex2d(const orc_scene* s, int tex, float u, float v, float* out4) { tex2d(s, tex, u, v, out4); }

/* PathTracer::GetUV pathtracer.cpp:533-536 */
static inline void get_uv(const orc_scene* s, int tri, float cx, float cy, float* ux, float* uy)
{
    const float* uv = s->uvs + (size_t)tri * 6;
    float w = 1.0f - cx - cy;
    *ux = w * uv[0] + cx * uv[2] + cy * uv[4];
    *uy = w * uv[1] + cx * uv[3] + cy * uv[5];
}

/* ---- Triangle::Init mesh.cpp:61-83 ------------------------------------------------------------*/
void orc_triangle_init(const float* in, float* out9)
{
    v3 v1 = ld3(in), v2 = ld3(in + 3), v3_ = ld3(in + 6);
    v3 e1 = sub(v2, v1), e2 = sub(v3_, v1);
    float d1x = in[11] - in[9], d1y = in[12] - in[10];
    float d2x = in[13] - in[9], d2y = in[14] - in[10];
    float f = 1.0f / (d1x * d2y - d2x * d1y);
    v3 tangent = V(f * (d2y * e1.x - d1y * e2.x), f * (d2y * e1.y - d1y * e2.y), f * (d2y * e1.z - d1y * e2.z));
    v3 bitangent = V(f * (-d2x * e1.x + d1x * e2.x), f * (-d2x * e1.y + d1x * e2.y), f * (-d2x * e1.z + d1x * e2.z));
    v3 normal = cross(e1, e2);
    tangent = normalize(tangent);
    bitangent = normalize(bitangent);
    normal = normalize(normal);
    out9[0] = normal.x; out9[1] = normal.y; out9[2] = normal.z;
    out9[3] = tangent.x; out9[4] = tangent.y; out9[5] = tangent.z;
    out9[6] = bitangent.x; out9[7] = bitangent.y; out9[8] = bitangent.z;
}

/* ---- own BVH (acceleration only; closest hit is tree-independent) ------------
 * The reference builds a random-axis median-split pointer tree (mesh.cpp:169-211) and walks both
 * children unconditionally (pathtracer.cpp:411-492).  Here: longest-axis object-median split,
 * leaves <= 2 triangles, boxes padded so culling is conservative, ordered traversal with t-max. */
static void tri_bounds(const orc_scene* s, int tri, float* mn, float* mx)
{
    const float* p = s->verts + (size_t)tri * 9;
    for (int a = 0; a < 3; a++)
    {
        float lo = p[a], hi = p[a];
        for (int k = 1; k < 3; k++) { float x = p[k * 3 + a]; if (x < lo) lo = x; if (x > hi) hi = x; }
        mn[a] = lo; mx[a] = hi;
    }
}
typedef struct { const orc_scene* s; int axis; } cmp_ctx;
static cmp_ctx g_cmp;   /* build is single-threaded */
static int cmp_centroid(const void* a, const void* b)
{
    int ia = *(const int32_t*)a, ib = *(const int32_t*)b;
    const float* pa = g_cmp.s->verts + (size_t)ia * 9 + g_cmp.axis;
    const float* pb = g_cmp.s->verts + (size_t)ib * 9 + g_cmp.axis;
    float ca = pa[0] + pa[3] + pa[6], cb = pb[0] + pb[3] + pb[6];
    if (ca < cb) return -1;
    if (ca > cb) return 1;
    return ia < ib ? -1 : (ia > ib ? 1 : 0);
}
static int32_t build(orc_scene* s, int first, int count, int depth, float pad)
{
    int32_t id = s->nnodes++;
    node_t* n = &s->nodes[id];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int i = 0; i < count; i++)
    {
        float a[3], b[3];
        tri_bounds(s, s->order[first + i], a, b);
        for (int k = 0; k < 3; k++) { if (a[k] < mn[k]) mn[k] = a[k]; if (b[k] > mx[k]) mx[k] = b[k]; }
    }
    for (int k = 0; k < 3; k++) { n->bmin[k] = mn[k] - pad; n->bmax[k] = mx[k] + pad; }
    if (depth > s->depth) s->depth = depth;
    if (count <= 2) { n->left = n->right = -1; n->first = first; n->count = count; return id; }
    int axis = 0;
    float e0 = mx[0] - mn[0], e1 = mx[1] - mn[1], e2 = mx[2] - mn[2];
    if (e1 > e0 && e1 >= e2) axis = 1; else if (e2 > e0 && e2 > e1) axis = 2;
    g_cmp.s = s; g_cmp.axis = axis;
    qsort(s->order + first, (size_t)count, sizeof(int32_t), cmp_centroid);
    int half = count / 2;
    n->first = 0; n->count = 0;
    int32_t l = build(s, first, half, depth + 1, pad);
    int32_t r = build(s, first + half, count - half, depth + 1, pad);
    n = &s->nodes[id];
    n->left = l; n->right = r;
    return id;
}

orc_scene* orc_create(const orc_scene_desc* d)
{
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    s->nt = d->num_triangles;
    size_t n = (size_t)s->nt;
    s->verts = (float*)dup(d->verts, n * 9 * sizeof(float));
    s->normals = (float*)dup(d->normals, n * 9 * sizeof(float));
    s->uvs = (float*)dup(d->uvs, n * 6 * sizeof(float));
    s->tbn = (float*)dup(d->tbn, n * 9 * sizeof(float));
    s->smoothing = (uint8_t*)dup(d->smoothing, n);
    s->material = (int32_t*)dup(d->material, n * sizeof(int32_t));
    s->nm = d->num_materials;
    s->mats = (orc_material*)dup(d->materials, (size_t)s->nm * sizeof(orc_material));
    s->ntex = d->num_textures;
    s->tex = (orc_texture*)dup(d->textures, (size_t)s->ntex * sizeof(orc_texture));
    s->texel_bytes = d->texel_bytes;
    s->texels = (uint8_t*)dup(d->texels, (size_t)d->texel_bytes);
    s->nl = d->num_lights;
    s->lights = (int32_t*)dup(d->lights, (size_t)s->nl * sizeof(int32_t));
    s->cap = (int32_t)(2 * n + 2);
    s->nodes = (node_t*)calloc((size_t)s->cap, sizeof(node_t));
    s->order = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t));
    for (size_t i = 0; i < n; i++) s->order[i] = (int32_t)i;
    float ext = 1.0f;
    for (size_t i = 0; i < n * 9; i++) { float a = fabsf(s->verts[i]); if (a > ext) ext = a; }
    if (n > 0) build(s, 0, (int)n, 1, 1e-5f * ext);
    return s;
}
void orc_destroy(orc_scene* s)
{
    if (!s) return;
    free(s->verts); free(s->normals); free(s->uvs); free(s->tbn); free(s->smoothing); free(s->material);
    free(s->mats); free(s->tex); free(s->texels); free(s->lights); free(s->nodes); free(s->order);
    free(s);
}
void orc_bvh_info(const orc_scene* s, int32_t* nodes, int32_t* depth) { *nodes = s->nnodes; *depth = s->depth; }

typedef struct { int32_t tri; float t, u, v; } hit_t;

/* candidate test of one triangle: Hit leaf branch, pathtracer.cpp:463-489 */
static inline void test_triangle(const orc_scene* s, int tri, v3 ro, v3 rd, rng_t* rng, uint32_t ray, hit_t* best)
{
    const float* p = s->verts + (size_t)tri * 9;
    float t, u, v;
    if (!intersect_triangle(ro, rd, ld3(p), ld3(p + 3), ld3(p + 6), &t, &u, &v)) return;
    /* (an infinite t - an overflowed determinant of a ray far outside every scene - is no hit: with best->t starting at
       infinity the tie rule would otherwise accept it; the kernels test the same) */
    if (!(t < INFINITY) || !(t < best->t || (t == best->t && tri < best->tri))) return;
    int otex = s->mats[s->material[tri]].tex[5];
    if (otex >= 0)
    {
        float ux, uy, c[4];
        get_uv(s, tri, u, v, &ux, &uy);
        tex2d(s, otex, ux, uy, c);
        if (!(rnd_opacity(rng, ray, (uint32_t)tri) < c[0])) return;
    }
    best->tri = tri; best->t = t; best->u = u; best->v = v;
}

static int closest_hit(const orc_scene* s, v3 ro, v3 rd, rng_t* rng, uint32_t ray, hit_t* out)
{
    hit_t best; best.tri = 0x7fffffff; best.t = INFINITY; best.u = best.v = 0.0f;
    if (s->nt == 0) return 0;
    v3 inv = V(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    int32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    while (sp)
    {
        const node_t* n = &s->nodes[stack[--sp]];
        float t0x = (n->bmin[0] - ro.x) * inv.x, t1x = (n->bmax[0] - ro.x) * inv.x;
        float t0y = (n->bmin[1] - ro.y) * inv.y, t1y = (n->bmax[1] - ro.y) * inv.y;
        float t0z = (n->bmin[2] - ro.z) * inv.z, t1z = (n->bmax[2] - ro.z) * inv.z;
        float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
        float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
        if (!(tn <= tf * 1.0000004f) || tf < 0.0f || tn > best.t) continue;
        if (n->left < 0)
        {
            for (int i = 0; i < n->count; i++) test_triangle(s, s->order[n->first + i], ro, rd, rng, ray, &best);
        }
        else if (sp + 2 <= 128)
        {
            stack[sp++] = n->right;
            stack[sp++] = n->left;
        }
    }
    if (best.tri == 0x7fffffff) return 0;
    *out = best;
    return 1;
}
static int closest_hit_brute(const orc_scene* s, v3 ro, v3 rd, rng_t* rng, uint32_t ray, hit_t* out)
{
    hit_t best; best.tri = 0x7fffffff; best.t = INFINITY; best.u = best.v = 0.0f;
    for (int i = 0; i < s->nt; i++) test_triangle(s, i, ro, rd, rng, ray, &best);
    if (best.tri == 0x7fffffff) return 0;
    *out = best;
    return 1;
}
int orc_hit(const orc_scene* s, const float* ro, const float* rd, float* tuv, int32_t* tri)
{
    rng_t r; rng_init(&r, 0, 0); hit_t h;
    if (!closest_hit(s, ld3(ro), ld3(rd), &r, 0, &h)) { *tri = -1; tuv[0] = tuv[1] = tuv[2] = 0.0f; return 0; }
    *tri = h.tri; tuv[0] = h.t; tuv[1] = h.u; tuv[2] = h.v; return 1;
}
int orc_hit_brute(const orc_scene* s, const float* ro, const float* rd, float* tuv, int32_t* tri)
{
    rng_t r; rng_init(&r, 0, 0); hit_t h;
    if (!closest_hit_brute(s, ld3(ro), ld3(rd), &r, 0, &h)) { *tri = -1; tuv[0] = tuv[1] = tuv[2] = 0.0f; return 0; }
    *tri = h.tri; tuv[0] = h.t; tuv[1] = h.u; tuv[2] = h.v; return 1;
}

/* ---- hemisphere / lobe samplers, pathtracer.cpp:606-611 and :618-623 ------------------------------
 * pole: the axis the sample is built around (n or r); basis_from: vector crossed to make u,v
 * (n for the hemisphere form, r for the lobe form); nx_test: |n.x| against thr chooses the helper. */
static inline v3 sample_about(v3 n_for_test, float thr, v3 basis_from, v3 pole, float w, float theta)
{
    v3 u = fabsf(n_for_test.x) < thr ? cross(V(1.0f, 0.0f, 0.0f), basis_from) : cross(V(1.0f, 1.0f, 1.0f), basis_from);
    u = normalize(u);
    v3 v = normalize(cross(u, basis_from));
    float ang = (float)(2.0f * ORC_PI_D * theta);       /* 2.0f*M_PI*theta evaluated in double */
    float sn, cs;
    orc_sincos(ang, &sn, &cs);
    v3 d = add(add(muls(u, w * cs), muls(v, w * sn)), muls(pole, sqrtf(1.0f - w * w)));
    return normalize(d);
}

/* PathTracer::SampleTriangle + DirectIllumimation, pathtracer.cpp:494-531 */
static v3 direct_illumination(const orc_scene* s, v3 p, v3 n, v3 diffuse, rng_t* rng, uint32_t* ray)
{
    if (s->nl == 0) return V(0.0f, 0.0f, 0.0f);
    int lightId = (int)floorf(rnd(rng) * (float)s->nl);
    if (lightId == s->nl && lightId > 0) lightId--;
    int ltri = s->lights[lightId];
    const float* lp = s->verts + (size_t)ltri * 9;
    float u = sqrtf(rnd(rng));
    float v = rnd(rng);
    float w0 = 1.0f - u, w1 = u * (1.0f - v), w2 = u * v;
    v3 vLight = add(add(muls(ld3(lp), w0), muls(ld3(lp + 3), w1)), muls(ld3(lp + 6), w2));
    v3 l = normalize(sub(vLight, p));
    float ndl = dot(neg(n), neg(l));
    if (ndl <= 0.0f) return V(0.0f, 0.0f, 0.0f);
    hit_t h;
    uint32_t r = (*ray)++;
    if (closest_hit(s, p, l, rng, r, &h))
    {
        if (h.tri != ltri) return V(0.0f, 0.0f, 0.0f);
    }
    const orc_material* lm = &s->mats[s->material[ltri]];
    v3 lColor = muls(ld3(lm->emissive), lm->emissive_intensity);
    return muls(mulv(lColor, diffuse), ndl);
}

/* One surface interaction of PathTracer::Trace (pathtracer.cpp:551-727): everything between the
 * closest hit and the recursive call.  Returns 0 when the path ends here with no contribution
 * (terminal bounce :571, Russian roulette :590-594). */
typedef struct {
    v3 p, n, dir;        /* offset hit point, shading normal, sampled continuation direction */
    v3 e;                /* emiss * emissiveIntensity */
    v3 weight;           /* factor on the recursive term (specular or diffuse) */
    v3 diffuse;          /* textured diffuse colour (argument of DirectIllumimation) */
    int diffuse_bounce;  /* 1: DirectIllumimation is added (:638, :724) */
} bounce_t;

static int shade(const orc_scene* s, v3 ro, v3 rd, const hit_t* h, int D, int* depth_io, int* iter_io,
                 int* inside_io, rng_t* rng, bounce_t* b)
{
    int depth = *depth_io, iter = *iter_io, inside = *inside_io;
    const orc_material* mat = &s->mats[s->material[h->tri]];
    v3 p = add(ro, muls(rd, h->t));                                       /* :553 */
    float uvx, uvy;
    get_uv(s, h->tri, h->u, h->v, &uvx, &uvy);
    const float* tb = s->tbn + (size_t)h->tri * 9;
    v3 n = ld3(tb);
    if (s->smoothing[h->tri])                                             /* :556, :538-543 */
    {
        const float* nn = s->normals + (size_t)h->tri * 9;
        float w = 1.0f - h->u - h->v;
        v3 sn = add(add(muls(ld3(nn), w), muls(ld3(nn + 3), h->u)), muls(ld3(nn + 6), h->v));
        n = normalize(sn);
    }
    if (mat->tex[1] >= 0)                                                 /* :558-566 */
    {
        float c[4];
        tex2d(s, mat->tex[1], uvx, uvy, c);
        v3 nt = V(c[0] * 2.0f - 1.0f, c[1] * 2.0f - 1.0f, c[2] * 2.0f - 1.0f);
        if (nt.z <= 0.0f) nt = V(nt.x, nt.y, ORC_EPS);
        nt = normalize(nt);
        v3 tg = ld3(tb + 3), bt = ld3(tb + 6);
        v3 m = V(tg.x * nt.x + bt.x * nt.y + n.x * nt.z,
                 tg.y * nt.x + bt.y * nt.y + n.y * nt.z,
                 tg.z * nt.x + bt.z * nt.y + n.z * nt.z);                  /* glm mat3*vec3 */
        n = normalize(m);
    }
    if (dot(n, rd) > 0.0f) n = neg(n);                                    /* :567-568 */
    p = add(p, muls(n, ORC_EPS));                                          /* :569 */

    if (!(iter < D)) return 0;                                            /* :571 */

    v3 diffuse = ld3(mat->diffuse);
    float c4[4];
    if (mat->tex[0] >= 0) { tex2d(s, mat->tex[0], uvx, uvy, c4); diffuse = V(c4[0], c4[1], c4[2]); }
    v3 emiss = ld3(mat->emissive);
    if (mat->tex[2] >= 0) { tex2d(s, mat->tex[2], uvx, uvy, c4); emiss = V(c4[0], c4[1], c4[2]); }
    float roughness = mat->roughness;
    if (mat->tex[3] >= 0) { tex2d(s, mat->tex[3], uvx, uvy, c4); roughness = c4[0]; }
    float reflectiveness = mat->reflectiveness;
    if (mat->tex[4] >= 0) { tex2d(s, mat->tex[4], uvx, uvy, c4); reflectiveness = c4[0]; }

    depth++; iter++;                                                      /* :586-587 */
    float mx = mat->diffuse[0] < mat->diffuse[1] ? mat->diffuse[1] : mat->diffuse[0];   /* glm::max */
    mx = mx < mat->diffuse[2] ? mat->diffuse[2] : mx;
    float prob = 0.95f < mx ? 0.95f : mx;                                 /* glm::min(0.95f, mx) */
    if (depth >= D)
    {
        if (fabsf(rnd(rng)) > prob) return 0;                             /* :590-594, no 1/prob */
    }

    v3 r = reflect(rd, n);                                                /* :596 */
    v3 dir;
    int diffuse_bounce = 0;
    v3 weight;

    if (mat->type == 0)
    {
        if (rnd(rng) < reflectiveness)                                    /* :601 */
        {
            if (roughness == 1.0f) { float w = rnd(rng), th = rnd(rng); dir = sample_about(n, 1.0f - ORC_EPS, n, n, w, th); }
            else if (roughness == 0.0f) dir = r;
            else { float w = rnd(rng) * roughness, th = rnd(rng); dir = sample_about(n, 1.0f - ORC_FLT_EPSILON, r, r, w, th); }
            iter--;
            weight = ld3(mat->specular);                                  /* :626 */
        }
        else
        {
            float w = rnd(rng), th = rnd(rng);
            dir = sample_about(n, 1.0f - ORC_EPS, n, n, w, th);           /* :631-636 */
            diffuse_bounce = 1;
            weight = diffuse;                                             /* :638 */
        }
    }
    else
    {
        int refract = 0;
        v3 refractN = n;
        if (roughness != 0.0f)                                            /* :645-654 */
        {
            float w = rnd(rng) * roughness, th = rnd(rng);
            refractN = sample_about(n, 1.0f - ORC_FLT_EPSILON, r, n, w, th);
        }
        float nc = 1.0f, ng = mat->ior;
        float eta = inside ? ng / nc : nc / ng;                           /* :658 */
        float r0 = (nc - ng) / (nc + ng);
        r0 = r0 * r0;
        float c = fabsf(dot(rd, refractN));
        float k = 1.0f - eta * eta * (1.0f - c * c);
        if (k < 0.0f) refract = 0;
        else
        {
            float re = r0 + (1.0f - r0) * (1.0f - c) * (1.0f - c);        /* :668 (squared) */
            if (fabsf(rnd(rng)) < re) refract = 0;
            else if (rnd(rng) < reflectiveness) refract = 0;
            else refract = 1;
        }
        if (!refract)
        {
            if (roughness == 1.0f) { float w = rnd(rng), th = rnd(rng); dir = sample_about(n, 1.0f - ORC_EPS, n, n, w, th); }
            else if (roughness == 0.0f) dir = r;
            else { float w = rnd(rng) * roughness, th = rnd(rng); dir = sample_about(n, 1.0f - ORC_FLT_EPSILON, r, r, w, th); }
            iter--;
            weight = ld3(mat->specular);                                  /* :702 */
        }
        else if (rnd(rng) < mat->translucency)                            /* :706 */
        {
            float a = eta * dot(n, rd) + sqrtf(k);
            dir = normalize(sub(muls(rd, eta), muls(refractN, a)));       /* :708 */
            p = sub(p, muls(muls(n, ORC_EPS), 2.0f));                      /* :709 */
            inside = !inside;
            iter--;
            weight = diffuse;                                             /* :712 */
        }
        else
        {
            float w = rnd(rng), th = rnd(rng);
            dir = sample_about(n, 1.0f - ORC_EPS, n, n, w, th);           /* :717-722 */
            diffuse_bounce = 1;
            weight = diffuse;                                             /* :724 */
        }
    }
    b->p = p; b->n = n; b->dir = dir;
    b->e = muls(emiss, mat->emissive_intensity);
    b->weight = weight; b->diffuse = diffuse; b->diffuse_bounce = diffuse_bounce;
    *depth_io = depth; *iter_io = iter; *inside_io = inside;
    return 1;
}

/* PathTracer::Trace pathtracer.cpp:545-732, ITERATIVE form: L += T*emission; L += T*direct; T *= weight.
 * This is the estimator the HIP kernel implements and the one orc_render uses.  Draw order per
 * bounce: [RR] branch, direction (w, theta), then the three DirectIllumimation draws. */
static v3 trace(const orc_scene* s, v3 ro, v3 rd, int D, rng_t* rng)
{
    v3 L = V(0.0f, 0.0f, 0.0f), T = V(1.0f, 1.0f, 1.0f);
    int depth = 0, iter = 0, inside = 0;
    uint32_t ray = 0;
    for (;;)
    {
        hit_t h; bounce_t b;
        if (!closest_hit(s, ro, rd, rng, ray++, &h)) break;                  /* :550 */
        if (!shade(s, ro, rd, &h, D, &depth, &iter, &inside, rng, &b)) break;
        L = add(L, mulv(T, b.e));                                            /* emiss * I */
        if (b.diffuse_bounce)
        {
            v3 di = direct_illumination(s, b.p, b.n, b.diffuse, rng, &ray);   /* :638 / :724 */
            L = add(L, mulv(T, di));
        }
        T = mulv(T, b.weight);
        ro = b.p; rd = b.dir;
    }
    return L;
}

/* The same estimator in the reference's RECURSIVE form, with the operand evaluation order g++
 * chose for `emiss*I + DirectIllumimation(..) + Trace(..)*diffuse` (pathtracer.cpp:638, :724):
 * nee_last=1 evaluates the recursive Trace before DirectIllumimation.  Used only to replay the
 * reference's draw tapes (tier T) and to tie the iterative form to them. */
static v3 trace_rec(const orc_scene* s, v3 ro, v3 rd, int D, int depth, int iter, int inside,
                    rng_t* rng, uint32_t* ray, int nee_last)
{
    hit_t h; bounce_t b;
    if (!closest_hit(s, ro, rd, rng, (*ray)++, &h)) return V(0.0f, 0.0f, 0.0f);
    if (!shade(s, ro, rd, &h, D, &depth, &iter, &inside, rng, &b)) return V(0.0f, 0.0f, 0.0f);
    if (!b.diffuse_bounce)
        return add(b.e, mulv(trace_rec(s, b.p, b.dir, D, depth, iter, inside, rng, ray, nee_last), b.weight));
    v3 di, sub_;
    if (nee_last)
    {
        sub_ = trace_rec(s, b.p, b.dir, D, depth, iter, inside, rng, ray, nee_last);
        di = direct_illumination(s, b.p, b.n, b.diffuse, rng, ray);
    }
    else
    {
        di = direct_illumination(s, b.p, b.n, b.diffuse, rng, ray);
        sub_ = trace_rec(s, b.p, b.dir, D, depth, iter, inside, rng, ray, nee_last);
    }
    return add(add(b.e, di), mulv(sub_, b.weight));
}

/* mode 0: iterative (NEE draws first); 1: recursive, NEE first; 2: recursive, NEE last */
int orc_trace_tape(const orc_scene* s, const float* ro, const float* rd, int max_depth,
                   const float* tape, int tape_len, int mode, float* out3)
{
    rng_t r; rng_init(&r, 0, 0);
    r.tape = tape; r.tape_len = tape_len; r.tape_pos = 0;
    v3 c;
    uint32_t ray = 0;
    if (mode == 0) c = trace(s, ld3(ro), ld3(rd), max_depth, &r);
    else c = trace_rec(s, ld3(ro), ld3(rd), max_depth, 0, 0, 0, &r, &ray, mode == 2);
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
    return r.tape_pos;
}

/* Counter-RNG radiance of sample `sample` of `pixel` from an explicit ray, iterative (mode 0) or
 * recursive NEE-first (mode 1): ties the two forms together on identical draws. */
void orc_trace_counter(const orc_scene* s, const float* ro, const float* rd, int max_depth,
                       uint64_t seed, uint32_t pixel, uint32_t sample, int mode, float* out3)
{
    rng_t r; rng_init(&r, pixel_key(seed, pixel), sample);
    v3 c;
    uint32_t ray = 0;
    if (mode == 0) c = trace(s, ld3(ro), ld3(rd), max_depth, &r);
    else c = trace_rec(s, ld3(ro), ld3(rd), max_depth, 0, 0, 0, &r, &ray, 0);
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}

/* ---- camera, PathTracer::RenderFrame pathtracer.cpp:755-766 -----------------------------------*/
typedef struct { v3 pos, dir, up, right, topLeft; float deltaX, deltaY; } frame_t;

static void frame_setup(const orc_camera* cam, int W, int H, frame_t* f)
{
    f->pos = ld3(cam->pos); f->dir = ld3(cam->dir); f->up = ld3(cam->up);
    v3 imgCenter = add(f->pos, muls(f->dir, cam->focal));
    float imgHeight = (float)((double)(2.0f * cam->focal) * tan((double)(cam->fovy / 2.0f) * ORC_PI_D / (double)180.0f));
    float aspect = (float)W / (float)H;
    float imgWidth = imgHeight * aspect;
    f->deltaX = imgWidth / (float)W;
    f->deltaY = imgHeight / (float)H;
    f->right = normalize(cross(f->up, f->dir));
    v3 topLeft = sub(imgCenter, muls(f->right, imgWidth * 0.5f));
    f->topLeft = add(topLeft, muls(f->up, imgHeight * 0.5f));
}

static void primary_row(const frame_t* f, int W, int i, float* out /* [W][3] */)
{
    v3 pixel = sub(f->topLeft, muls(f->up, (float)i * f->deltaY));             /* :782 */
    v3 step = muls(f->right, f->deltaX);
    for (int j = 0; j < W; j++)
    {
        v3 rayDir = normalize(sub(pixel, f->pos));                             /* :785 */
        out[j * 3] = rayDir.x; out[j * 3 + 1] = rayDir.y; out[j * 3 + 2] = rayDir.z;
        pixel = add(pixel, step);                                              /* :814 */
    }
}

void orc_primary_dirs(const orc_camera* cam, int W, int H, float* out)
{
    frame_t f; frame_setup(cam, W, H, &f);
    for (int i = 0; i < H; i++) primary_row(&f, W, i, out + (size_t)i * W * 3);
}

/* PathTracer::SampleCircle pathtracer.cpp:734-739 with the two draws passed in */
void orc_sample_circle(float r1, float r2, float* out2)
{
    float angle = (float)((double)r1 * 2. * ORC_PI_D);
    float radius = sqrtf(r2);
    float sn, cs;
    orc_sincos(angle, &sn, &cs);
    out2[0] = cs * radius; out2[1] = sn * radius;
}

/* AABB::Intersect, mesh.cpp:48-59, verbatim semantics: (min - ro) / rd with IEEE division (rd components may be 0),
 * glm min/max = (y < x ? y : x) / (x < y ? y : x), no t-max, boxes behind the ray are NOT rejected.  The oracle's own
 * traversal does not use it (its boxes only accelerate; closest hit is tree-independent) - this probe pins the row a5. */
int orc_aabb_intersect(const float* bmin, const float* bmax, const float* ro, const float* rd)
{
    float t1[3], t2[3];
    for (int a = 0; a < 3; a++)
    {
        float tmin = (bmin[a] - ro[a]) / rd[a], tmax = (bmax[a] - ro[a]) / rd[a];
        t1[a] = tmax < tmin ? tmax : tmin;          /* glm::min(tMin, tMax) */
        t2[a] = tmin < tmax ? tmax : tmin;          /* glm::max(tMin, tMax) */
    }
    float m01 = t1[0] < t1[1] ? t1[1] : t1[0];
    float tNear = m01 < t1[2] ? t1[2] : m01;
    float n01 = t2[1] < t2[0] ? t2[1] : t2[0];
    float tFar = t2[2] < n01 ? t2[2] : n01;
    return tNear >= tFar ? 0 : 1;
}

/* AABB::Build over n points then AABB::Check, mesh.cpp:6-46 (min/max start at +-(float)0xFFFF, mesh.h:13,63-64;
 * Check inflates a zero-thickness axis by EPS) */
void orc_aabb_build(const float* pts, int n, float* out6)
{
    const float INF_ = (float)0xFFFF;
    float mn[3] = { INF_, INF_, INF_ }, mx[3] = { -INF_, -INF_, -INF_ };
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++)
        {
            float v = pts[i * 3 + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    for (int a = 0; a < 3; a++) if (mn[a] == mx[a]) mx[a] += ORC_EPS;
    for (int a = 0; a < 3; a++) { out6[a] = mn[a]; out6[3 + a] = mx[a]; }
}

/* DirectIllumimation (pathtracer.cpp:504-531) with its three draws taken from `tape3` */
void orc_direct_illumination_tape(const orc_scene* s, const float* p, const float* n, const float* diffuse, const float* tape3, float* out3)
{
    rng_t rng; memset(&rng, 0, sizeof rng);
    rng.tape = tape3; rng.tape_len = 3; rng.tape_pos = 0;
    uint32_t ray = 0;
    v3 c = direct_illumination(s, ld3(p), ld3(n), ld3(diffuse), &rng, &ray);
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}

void orc_render(const orc_scene* s, const orc_camera* cam, int W, int H, int D,
                uint32_t first_sample, uint32_t spp, uint64_t seed, int rank, int world,
                float* total, uint8_t* rgb8, int threads)
{
    frame_t f; frame_setup(cam, W, H, &f);
    int tiles_x = (W + ORC_TILE - 1) / ORC_TILE, tiles_y = (H + ORC_TILE - 1) / ORC_TILE;
    if (world < 1) world = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#endif
    /* the rows' primary directions first (each row is an incremental walk from its first column, :782-785, :814) ... */
    float* dirs_all = (float*)malloc((size_t)W * H * 3 * sizeof(float));
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (int i = 0; i < H; i++) primary_row(&f, W, i, dirs_all + (size_t)i * W * 3);
    /* ... then the pixels, dealt to the threads as 16 x 16 tiles (the reference deals whole rows to its workers, :777: with
     * hundreds of threads and a frame whose lit pixels sit in a band of rows that leaves most of them idle; the result
     * of a pixel does not depend on who computes it) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int t = 0; t < tiles_x * tiles_y; t++)
    {
        const int ty = t / tiles_x, tx = t % tiles_x;
        const int tile = ty * tiles_x + (tx + 3 * ty) % tiles_x;    /* rows rotated by 3 tiles: diagonal ownership */
        if (tile % world != rank) continue;
        for (int i = ty * ORC_TILE; i < (ty + 1) * ORC_TILE && i < H; i++)     /* :777 */
        {
            const float* dirs = dirs_all + (size_t)i * W * 3;
            for (int j = tx * ORC_TILE; j < (tx + 1) * ORC_TILE && j < W; j++) /* :783 */
            {
                uint32_t pkey = pixel_key(seed, (uint32_t)(i * W + j));
                size_t px = ((size_t)(H - 1 - i) * W + j) * 3;                 /* :796 bottom-up */
                v3 acc = ld3(total + px);
                v3 rayDir0 = ld3(dirs + j * 3);
                for (uint32_t k = 0; k < spp; k++)
                {
                    rng_t rng; rng_init(&rng, pkey, first_sample + k);
                    v3 camPos = f.pos;                                         /* :787 */
                    v3 focalPoint = add(camPos, muls(rayDir0, cam->focal_dist));   /* :788 */
                    float r1 = rnd(&rng), r2 = rnd(&rng), off[2];
                    orc_sample_circle(r1, r2, off);
                    off[0] = off[0] * cam->aperture; off[1] = off[1] * cam->aperture;   /* :789 */
                    camPos = add(camPos, add(muls(f.right, off[0]), muls(f.up, off[1])));   /* :790 */
                    v3 rayDir = normalize(sub(focalPoint, camPos));            /* :791 */
                    v3 color = trace(s, camPos, rayDir, D, &rng);              /* :793 */
                    acc = add(acc, color);                                     /* :798-800 */
                }
                total[px] = acc.x; total[px + 1] = acc.y; total[px + 2] = acc.z;
                if (rgb8)
                {
                    float ns = (float)(first_sample + spp);                    /* (float)mSamples */
                    float c[3] = { acc.x / ns, acc.y / ns, acc.z / ns };
                    for (int k = 0; k < 3; k++)
                    {
                        float x = c[k];
                        x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);           /* glm::clamp; NaN -> NaN */
                        if (!(x == x)) x = 0.0f;
                        rgb8[px + k] = (uint8_t)(x * 255);                     /* :810-812 truncation */
                    }
                }
            }
        }
    }
    free(dirs_all);
}

/* RenderFrame (pathtracer.cpp:741-817) for ONE frame with the draws of the reference's single engine on tape, in the order its
 * loop consumes them when it runs on one thread: rows top to bottom, columns left to right, per pixel the two draws of
 * SampleCircle (:736-737) and then the path's (recursive form, g++'s operand order).  Adds into `total` (rows bottom-up, :796)
 * and returns the number of draws consumed.  Test infrastructure for tools/fuzz_frame_vs_reference.py. */
int orc_render_tape(const orc_scene* s, const orc_camera* cam, int W, int H, int D, const float* tape, int tape_len, float* total)
{
    frame_t f; frame_setup(cam, W, H, &f);
    float* dirs = (float*)malloc((size_t)W * 3 * sizeof(float));
    rng_t r; rng_init(&r, 0, 0);
    r.tape = tape; r.tape_len = tape_len; r.tape_pos = 0;
    for (int i = 0; i < H; i++)
    {
        primary_row(&f, W, i, dirs);
        for (int j = 0; j < W; j++)
        {
            size_t px = ((size_t)(H - 1 - i) * W + j) * 3;
            v3 rayDir0 = ld3(dirs + j * 3);
            v3 camPos = f.pos;
            v3 focalPoint = add(camPos, muls(rayDir0, cam->focal_dist));
            float r1 = rnd(&r), r2 = rnd(&r), off[2];
            orc_sample_circle(r1, r2, off);
            off[0] = off[0] * cam->aperture; off[1] = off[1] * cam->aperture;
            camPos = add(camPos, add(muls(f.right, off[0]), muls(f.up, off[1])));
            v3 rayDir = normalize(sub(focalPoint, camPos));
            uint32_t ray = 0;
            v3 color = trace_rec(s, camPos, rayDir, D, 0, 0, 0, &r, &ray, 1);
            total[px] += color.x; total[px + 1] += color.y; total[px + 2] += color.z;
        }
    }
    free(dirs);
    return r.tape_pos;
}

