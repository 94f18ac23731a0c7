// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// C-ABI harness around the *real* reference renderer.  It is compiled by
// oracle/Makefile.ref against the reference's own sources where they lie under
// /root/reference (mesh.cpp, image.cpp, pathtracer.cpp + vendored glm / tinyobj /
// stb headers); five one-line MSVC->g++ portability edits are applied by sed into a
// scratch directory under /tmp and only the resulting shared object
// lands in oracle/_ref/.  No reference source text is stored in this repository.
//
// `#define private public` opens PathTracer's private section so the harness can seed
// mRng and call Trace / Hit / IntersectTriangle single-threaded, which makes the
// reference deterministic inside one process.
//
// What the harness exposes (all extern "C"):
//   scene set-up through the reference's own public API (LoadObject, SetMaterial,
//   Set*TextureForElement, BuildBVH, SetCamera, ...), RenderFrame, and per-function probes used
//   by oracle/gen_golden.py to write the committed fixtures under tests/golden/.

#include <cstdio>
#include <algorithm>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <random>

#define STB_IMAGE_IMPLEMENTATION
#include <stb_image.h>
#define TINYOBJLOADER_IMPLEMENTATION
#include <tiny_obj_loader.h>

#define private public
#include "pathtracer.h"
#undef private

#include <omp.h>
#include <glm/gtc/matrix_transform.hpp>

// static storage: the reference never initialises mBvh (pathtracer.cpp:11-27) and relies on
// its single instance being a zero-initialised global (main.cpp:77).
static PathTracer g_pt;
static std::vector<unsigned char> g_out;

extern "C" {

void ref_clear() { g_pt.ClearScene(); }

// model: 16 floats, column-major (glm layout)
void ref_load_obj(const char* path, const float* model)
{
    glm::mat4 M;
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            M[c][r] = model[c * 4 + r];
    g_pt.LoadObject(path, M);
}

int ref_num_objects() { return (int)g_pt.mLoadedObjects.size(); }
int ref_num_elements(int obj) { return (int)g_pt.mLoadedObjects[obj].elements.size(); }
// names of a loaded object (elem < 0) and of its elements as LoadObject (pathtracer.cpp:49-62) took them from tinyobj's shapes
int ref_name(int obj, int elem, char* out, int cap)
{
    const std::string& n = elem < 0 ? g_pt.mLoadedObjects[obj].name : g_pt.mLoadedObjects[obj].elements[elem].name;
    const int len = (int)std::min<size_t>(n.size(), (size_t)std::max(cap - 1, 0));
    std::memcpy(out, n.data(), (size_t)len);
    if (cap > 0) out[len] = 0;
    return (int)n.size();
}

// m: type, diffuse rgb, specular rgb, emissive rgb, emissiveIntensity, roughness,
//    reflectiveness, translucency, ior   (14 floats; type as 0/1)
void ref_set_material(int obj, int elem, const float* m)
{
    Material mat;
    mat.type = m[0] != 0.0f ? MaterialType::TRANSLUCENT : MaterialType::OPAQUE;
    mat.diffuse = glm::vec3(m[1], m[2], m[3]);
    mat.specular = glm::vec3(m[4], m[5], m[6]);
    mat.emissive = glm::vec3(m[7], m[8], m[9]);
    mat.emissiveIntensity = m[10];
    mat.roughness = m[11];
    mat.reflectiveness = m[12];
    mat.translucency = m[13];
    mat.ior = m[14];
    g_pt.SetMaterial(obj, elem, mat);
}

// slot: 0 diffuse 1 normal 2 emissive 3 roughness 4 metallic 5 opacity
void ref_set_texture(int obj, int elem, int slot, const char* path)
{
    switch (slot)
    {
    case 0: g_pt.SetDiffuseTextureForElement(obj, elem, path); break;
    case 1: g_pt.SetNormalTextureForElement(obj, elem, path); break;
    case 2: g_pt.SetEmissTextureForElement(obj, elem, path); break;
    case 3: g_pt.SetRoughnessTextureForElement(obj, elem, path); break;
    case 4: g_pt.SetMetallicTextureForElement(obj, elem, path); break;
    case 5: g_pt.SetOpacityTextureForElement(obj, elem, path); break;
    }
}

void ref_build() { g_pt.BuildBVH(); }
int ref_num_triangles() { return g_pt.GetTriangleCount(); }
int ref_num_lights() { return (int)g_pt.mLights.size(); }

// Export the (post-BuildBVH, i.e. sorted) triangle array as the reference holds it.
// per triangle 38 floats: v1 v2 v3 n1 n2 n3 (18) uv1 uv2 uv3 (6) normal tangent bitangent (9)
// smoothing, objectId, elementId (3) + 2 pad
void ref_get_triangles(float* out)
{
    for (size_t i = 0; i < g_pt.mTriangles.size(); i++)
    {
        const Triangle& t = g_pt.mTriangles[i];
        float* o = out + i * 38;
        const glm::vec3* v3s[] = { &t.v1, &t.v2, &t.v3, &t.n1, &t.n2, &t.n3 };
        for (int k = 0; k < 6; k++) { o[k * 3] = v3s[k]->x; o[k * 3 + 1] = v3s[k]->y; o[k * 3 + 2] = v3s[k]->z; }
        o[18] = t.uv1.x; o[19] = t.uv1.y; o[20] = t.uv2.x; o[21] = t.uv2.y; o[22] = t.uv3.x; o[23] = t.uv3.y;
        const glm::vec3* w3s[] = { &t.normal, &t.tangent, &t.bitangent };
        for (int k = 0; k < 3; k++) { o[24 + k * 3] = w3s[k]->x; o[25 + k * 3] = w3s[k]->y; o[26 + k * 3] = w3s[k]->z; }
        o[33] = t.smoothing ? 1.0f : 0.0f;
        o[34] = (float)t.objectId;
        o[35] = (float)t.elementId;
        o[36] = o[37] = 0.0f;
    }
}

void ref_set_camera(const float* pos, const float* dir, const float* up)
{
    g_pt.SetCamera(glm::vec3(pos[0], pos[1], pos[2]), glm::vec3(dir[0], dir[1], dir[2]),
                   glm::vec3(up[0], up[1], up[2]));
}
void ref_get_camera(float* out9)
{
    out9[0] = g_pt.mCamPos.x; out9[1] = g_pt.mCamPos.y; out9[2] = g_pt.mCamPos.z;
    out9[3] = g_pt.mCamDir.x; out9[4] = g_pt.mCamDir.y; out9[5] = g_pt.mCamDir.z;
    out9[6] = g_pt.mCamUp.x; out9[7] = g_pt.mCamUp.y; out9[8] = g_pt.mCamUp.z;
}
void ref_set_projection(float f, float fovy) { g_pt.SetProjection(f, fovy); }
void ref_get_projection(float* out2) { out2[0] = g_pt.mCamFocal; out2[1] = g_pt.mCamFovy; }
void ref_set_focal_dist(float d) { g_pt.SetCameraFocalDist(d); }
void ref_set_aperture(float a) { g_pt.SetCameraAperture(a); }
void ref_set_depth(int d) { g_pt.SetTraceDepth(d); }

void ref_set_resolution(int w, int h)
{
    if (g_pt.mTotalImg) { delete[] g_pt.mTotalImg; g_pt.mTotalImg = 0; }
    g_pt.SetResolution(glm::ivec2(w, h));
    g_out.assign((size_t)w * h * 3, 0);
    g_pt.SetOutImage(g_out.data());
    g_pt.ResetImage();
}

void ref_seed(unsigned int s) { g_pt.mRng.seed(s); }

// Run n RenderFrame() calls as the reference does (OpenMP, shared engine).  threads<=0: default.
// Returns wall seconds.
double ref_render_frames(int n, int threads)
{
    int saved = omp_get_max_threads();
    if (threads > 0) omp_set_num_threads(threads);
    double t0 = omp_get_wtime();
    for (int i = 0; i < n; i++) g_pt.RenderFrame();
    double t1 = omp_get_wtime();
    omp_set_num_threads(saved);
    return t1 - t0;
}
int ref_samples() { return g_pt.GetSamples(); }
void ref_read_total(float* out) { std::memcpy(out, g_pt.mTotalImg, sizeof(float) * g_pt.mResolution.x * g_pt.mResolution.y * 3); }
void ref_read_rgb8(unsigned char* out) { std::memcpy(out, g_out.data(), g_out.size()); }

// ---- per-function probes -------------------------------------------------------------------

void ref_intersect_triangle(const float* ro, const float* rd, const float* v0, const float* v1,
                            const float* v2, float* out3)
{
    glm::vec3 r = g_pt.IntersectTriangle(glm::vec3(ro[0], ro[1], ro[2]), glm::vec3(rd[0], rd[1], rd[2]),
                                         glm::vec3(v0[0], v0[1], v0[2]), glm::vec3(v1[0], v1[1], v1[2]),
                                         glm::vec3(v2[0], v2[1], v2[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

int ref_aabb_intersect(const float* bmin, const float* bmax, const float* ro, const float* rd)
{
    AABB b;
    b.min = glm::vec3(bmin[0], bmin[1], bmin[2]);
    b.max = glm::vec3(bmax[0], bmax[1], bmax[2]);
    return b.Intersect(glm::vec3(ro[0], ro[1], ro[2]), glm::vec3(rd[0], rd[1], rd[2])) ? 1 : 0;
}

// AABB::Build over n points then Check(); out6 = min, max
void ref_aabb_build(const float* pts, int n, float* out6)
{
    AABB b;
    for (int i = 0; i < n; i++) b.Build(glm::vec3(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]));
    b.Check();
    out6[0] = b.min.x; out6[1] = b.min.y; out6[2] = b.min.z;
    out6[3] = b.max.x; out6[4] = b.max.y; out6[5] = b.max.z;
}

// Triangle::Init on (v1 v2 v3 uv1 uv2 uv3) = 15 floats -> normal, tangent, bitangent (9 floats)
void ref_triangle_init(const float* in15, float* out9)
{
    Triangle t;
    t.v1 = glm::vec3(in15[0], in15[1], in15[2]);
    t.v2 = glm::vec3(in15[3], in15[4], in15[5]);
    t.v3 = glm::vec3(in15[6], in15[7], in15[8]);
    t.uv1 = glm::vec2(in15[9], in15[10]);
    t.uv2 = glm::vec2(in15[11], in15[12]);
    t.uv3 = glm::vec2(in15[13], in15[14]);
    t.Init();
    out9[0] = t.normal.x; out9[1] = t.normal.y; out9[2] = t.normal.z;
    out9[3] = t.tangent.x; out9[4] = t.tangent.y; out9[5] = t.tangent.z;
    out9[6] = t.bitangent.x; out9[7] = t.bitangent.y; out9[8] = t.bitangent.z;
}

static Image* g_img = 0;
int ref_image_load(const char* path, int* w, int* h)
{
    if (g_img) delete g_img;
    g_img = new Image(path);
    *w = g_img->width(); *h = g_img->height();
    return g_img->data() ? 1 : 0;
}
void ref_image_data(unsigned char* out) { std::memcpy(out, g_img->data(), (size_t)g_img->width() * g_img->height() * 4); }
void ref_tex2d(float u, float v, float* out4)
{
    glm::vec4 r = g_img->tex2D(glm::vec2(u, v));
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}

// Closest hit through the reference's recursive Hit.  Returns 1 on hit; out: t,u,v ; tri = index
// into the exported triangle array.
int ref_hit(const float* ro, const float* rd, float* out3, int* tri)
{
    Triangle* t = 0; float d = 0.0f; glm::vec2 c;
    bool h = g_pt.Hit(g_pt.mBvh, glm::vec3(ro[0], ro[1], ro[2]), glm::vec3(rd[0], rd[1], rd[2]), t, d, c);
    if (!h) { *tri = -1; out3[0] = out3[1] = out3[2] = 0.0f; return 0; }
    *tri = (int)(t - g_pt.mTriangles.data());
    out3[0] = d; out3[1] = c.x; out3[2] = c.y;
    return 1;
}

// Draw tape: the floats PathTracer::Rand() WILL return from the current engine state, without
// advancing it (one Rand() == one mt19937 draw through uniform_real_distribution<float>).
void ref_peek_tape(float* tape, int n)
{
    std::mt19937 clone = g_pt.mRng;
    for (int i = 0; i < n; i++)
    {
        std::uniform_real_distribution<float> dis(0.0f, 1.0f);
        tape[i] = dis(clone);
    }
}

// Number of draws the engine advanced by since `before` was captured is recovered by replaying a
// clone until its state matches (bounded by cap).
static std::mt19937 g_mark;
void ref_mark() { g_mark = g_pt.mRng; }
int ref_draws_since_mark(int cap)
{
    std::mt19937 c = g_mark;
    for (int i = 0; i <= cap; i++)
    {
        if (c == g_pt.mRng) return i;
        c();
    }
    return -1;
}

// Single-threaded radiance estimate for one ray through the reference's recursive Trace.
void ref_trace(const float* ro, const float* rd, float* out3)
{
    glm::vec3 c = g_pt.Trace(glm::vec3(ro[0], ro[1], ro[2]), glm::vec3(rd[0], rd[1], rd[2]));
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}

// ... entered at any point of a path: Trace's own depth / iter / inside arguments, the engine advanced by `skip` draws first
// (debugging aid of tools/fuzz_trace_vs_reference.py: where along a path does a replay part ways with the reference)
void ref_trace_state(const float* ro, const float* rd, int depth, int iter, int inside, int skip, float* out3)
{
    for (int i = 0; i < skip; i++) g_pt.mRng();
    glm::vec3 c = g_pt.Trace(glm::vec3(ro[0], ro[1], ro[2]), glm::vec3(rd[0], rd[1], rd[2]), depth, iter, inside != 0);
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}

void ref_sample_circle(float* out2)
{
    glm::vec2 c = g_pt.SampleCircle();
    out2[0] = c.x; out2[1] = c.y;
}

void ref_direct_illumination(const float* rd, const float* p, const float* n, const float* diffuse, float* out3)
{
    glm::vec3 c = g_pt.DirectIllumimation(glm::vec3(rd[0], rd[1], rd[2]), glm::vec3(p[0], p[1], p[2]),
                                          glm::vec3(n[0], n[1], n[2]), glm::vec3(diffuse[0], diffuse[1], diffuse[2]));
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}

// glm 0.9.3.1 helpers used by the scene layer (degrees!) so the scene-ingest restatement can be
// pinned too: Previewer-style TRS (previewer.h:104-112) and Euler camera (previewer.cpp:883-902).
void ref_trs_matrix(const float* loc, const float* rot, const float* scl, float* out16)
{
    glm::mat4 T = glm::translate(glm::mat4(1.0f), glm::vec3(loc[0], loc[1], loc[2]));
    glm::mat4 R = glm::rotate(T, rot[0], glm::vec3(1.0f, 0.0f, 0.0f));
    R = glm::rotate(R, rot[1], glm::vec3(0.0f, 1.0f, 0.0f));
    R = glm::rotate(R, rot[2], glm::vec3(0.0f, 0.0f, 1.0f));
    glm::mat4 S = glm::scale(R, glm::vec3(scl[0], scl[1], scl[2]));
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            out16[c * 4 + r] = S[c][r];
}
void ref_euler_camera(const float* rot, float* out6)
{
    glm::mat4 Rx = glm::rotate(glm::mat4(1.0f), rot[0], glm::vec3(1.0f, 0.0f, 0.0f));
    glm::mat4 Ry = glm::rotate(glm::mat4(1.0f), rot[1], glm::vec3(0.0f, 1.0f, 0.0f));
    glm::mat4 Rz = glm::rotate(glm::mat4(1.0f), rot[2], glm::vec3(0.0f, 0.0f, 1.0f));
    glm::vec3 d = glm::normalize(glm::vec3(Rz * Ry * Rx * glm::vec4(0.0f, 0.0f, 1.0f, 1.0f)));
    glm::vec3 u = glm::normalize(glm::vec3(Rz * Ry * Rx * glm::vec4(0.0f, 1.0f, 0.0f, 1.0f)));
    out6[0] = d.x; out6[1] = d.y; out6[2] = d.z; out6[3] = u.x; out6[4] = u.y; out6[5] = u.z;
}

} // extern "C"
