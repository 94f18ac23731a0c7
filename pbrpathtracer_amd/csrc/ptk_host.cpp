// C wrapper over the C++ host layer — see include/ptk_host.h.
#include "ptk_host.h"

#include <algorithm>
#include <cstring>
#include <string>

#include "host_scene.h"
#include "pathtracer.h"

using namespace ptkhost;

bool ptk_write_png_rgb8_bottom_up(const char* path, const unsigned char* rgb, int w, int h);   // image.cpp

struct pth_tracer {
    PathTracer pt;
    std::string error;
};

extern "C" {

pth_tracer* pth_create(int device_ordinal)
{
    pth_tracer* t = new (std::nothrow) pth_tracer();
    if (t) t->pt.SetDevice(device_ordinal);
    return t;
}
void pth_destroy(pth_tracer* t) { delete t; }

void pth_load_object(pth_tracer* t, const char* file, const float* mm)
{
    glm::mat4 M(1.0f);
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) M[c][r] = mm[c * 4 + r];
    t->pt.LoadObject(file, M);
}
void pth_set_material(pth_tracer* t, int obj, int elem, const float* m)
{
    Material mat;
    mat.type = m[0] != 0.0f ? MaterialType::TRANSLUCENT : MaterialType::OPAQUE;
    mat.diffuse = glm::vec3(m[1], m[2], m[3]);
    mat.specular = glm::vec3(m[4], m[5], m[6]);
    mat.emissive = glm::vec3(m[7], m[8], m[9]);
    mat.emissiveIntensity = m[10]; mat.roughness = m[11]; mat.reflectiveness = m[12];
    mat.translucency = m[13]; mat.ior = m[14];
    t->pt.SetMaterial(obj, elem, mat);
}
void pth_set_texture(pth_tracer* t, int obj, int elem, int slot, const char* file)
{
    switch (slot)
    {
    case 0: t->pt.SetDiffuseTextureForElement(obj, elem, file); break;
    case 1: t->pt.SetNormalTextureForElement(obj, elem, file); break;
    case 2: t->pt.SetEmissTextureForElement(obj, elem, file); break;
    case 3: t->pt.SetRoughnessTextureForElement(obj, elem, file); break;
    case 4: t->pt.SetMetallicTextureForElement(obj, elem, file); break;
    case 5: t->pt.SetOpacityTextureForElement(obj, elem, file); break;
    default: break;
    }
}
void pth_build_bvh(pth_tracer* t) { t->pt.BuildBVH(); }
void pth_reset_image(pth_tracer* t) { t->pt.ResetImage(); }
void pth_clear_scene(pth_tracer* t) { t->pt.ClearScene(); }
int pth_get_samples(pth_tracer* t) { return t->pt.GetSamples(); }
int pth_get_triangle_count(pth_tracer* t) { return t->pt.GetTriangleCount(); }
int pth_get_trace_depth(pth_tracer* t) { return t->pt.GetTraceDepth(); }
void pth_set_trace_depth(pth_tracer* t, int d) { t->pt.SetTraceDepth(d); }
void pth_set_out_image(pth_tracer* t, uint8_t* out) { t->pt.SetOutImage(out); }
void pth_set_out_gl_buffer(pth_tracer* t, unsigned int gl_buffer) { t->pt.SetOutGLBuffer(gl_buffer); }
void pth_set_out_device_image(pth_tracer* t, void* device_rgb8) { t->pt.SetOutDeviceImage(device_rgb8); }
void pth_set_resolution(pth_tracer* t, int w, int h) { t->pt.SetResolution(glm::ivec2(w, h)); }
void pth_get_resolution(pth_tracer* t, int* w, int* h) { glm::ivec2 r = t->pt.GetResolution(); *w = r.x; *h = r.y; }
int pth_num_objects(pth_tracer* t) { return (int)t->pt.GetLoadedObjects().size(); }
int pth_num_elements(pth_tracer* t, int obj)
{
    auto o = t->pt.GetLoadedObjects();
    return obj >= 0 && obj < (int)o.size() ? (int)o[obj].elements.size() : 0;
}
int pth_name(pth_tracer* t, int obj, int elem, char* out, int cap)
{
    const auto objs = t->pt.GetLoadedObjects();
    if (obj < 0 || obj >= (int)objs.size() || elem >= (int)objs[obj].elements.size()) return -1;
    const std::string& n = elem < 0 ? objs[obj].name : objs[obj].elements[elem].name;
    const int len = (int)std::min<size_t>(n.size(), (size_t)std::max(cap - 1, 0));
    std::memcpy(out, n.data(), (size_t)len);
    if (cap > 0) out[len] = 0;
    return (int)n.size();
}
void pth_set_camera(pth_tracer* t, const float* p, const float* d, const float* u)
{
    t->pt.SetCamera(glm::vec3(p[0], p[1], p[2]), glm::vec3(d[0], d[1], d[2]), glm::vec3(u[0], u[1], u[2]));
}
void pth_set_projection(pth_tracer* t, float f, float fovy) { t->pt.SetProjection(f, fovy); }
void pth_set_focal_dist(pth_tracer* t, float d) { t->pt.SetCameraFocalDist(d); }
void pth_set_aperture(pth_tracer* t, float a) { t->pt.SetCameraAperture(a); }
void pth_render_frame(pth_tracer* t) { t->pt.RenderFrame(); }
void pth_exit(pth_tracer* t) { t->pt.Exit(); }

void pth_set_seed(pth_tracer* t, uint64_t seed) { t->pt.SetSeed(seed); }
void pth_set_tile(pth_tracer* t, int rank, int world) { t->pt.SetTile(rank, world); }
void pth_render_frames(pth_tracer* t, int count) { t->pt.RenderFrames(count); }
int pth_read_accum(pth_tracer* t, float* out) { return t->pt.ReadAccumulation(out) ? 1 : 0; }
const char* pth_last_error(pth_tracer* t)
{
    std::string e = t->pt.LastError();
    if (!e.empty()) t->error = e;
    return t->error.c_str();
}
ptk_ctx* pth_context(pth_tracer* t) { return t->pt.Context(); }
const ptk_scene_desc* pth_staged_scene(pth_tracer* t) { return t->pt.StagedScene(); }

int pth_load_scene_file(pth_tracer* t, const char* path)
{
    SceneFile s; std::string err;
    if (!read_pts(path, s, &err)) { t->error = std::string(path) + ": " + err; return -1; }
    send_scene(s, t->pt);
    return 0;
}
int pth_pts_roundtrip(const char* in_path, const char* out_path)
{
    SceneFile s;
    if (!read_pts(in_path, s, 0)) return -1;
    return write_pts(out_path, s) ? 0 : -2;
}

void pth_trs_matrix(const float* loc, const float* rot, const float* scl, float* out16)
{
    glm::mat4 M = trs_matrix(loc, rot, scl);
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) out16[c * 4 + r] = M[c][r];
}
void pth_euler_camera(const float* rot, float* out6) { euler_camera(rot, out6, out6 + 3); }
void pth_triangle_init(const float* in, float* out9)
{
    StagedTriangle t;
    std::memset(&t, 0, sizeof(t));
    for (int k = 0; k < 3; k++) { for (int a = 0; a < 3; a++) t.v[k][a] = in[k * 3 + a]; t.uv[k][0] = in[9 + k * 2]; t.uv[k][1] = in[10 + k * 2]; }
    triangle_init(t);
    for (int a = 0; a < 3; a++) { out9[a] = t.normal[a]; out9[3 + a] = t.tangent[a]; out9[6 + a] = t.bitangent[a]; }
}
int pth_export_png(const char* path, const uint8_t* rgb8_bottom_up, int w, int h)
{
    return ptk_write_png_rgb8_bottom_up(path, rgb8_bottom_up, w, h) ? 1 : 0;
}
static Image g_img;
int pth_image_load(const char* file, int* w, int* h)
{
    g_img.Load(file);
    *w = g_img.width(); *h = g_img.height();
    return g_img.data() ? 1 : 0;
}
void pth_image_data(uint8_t* out) { if (g_img.data()) std::memcpy(out, g_img.data(), (size_t)g_img.width() * g_img.height() * 4); }
void pth_image_tex2d(float u, float v, float* out4)
{
    glm::vec4 r = g_img.tex2D(glm::vec2(u, v));
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}

}  // extern "C"
