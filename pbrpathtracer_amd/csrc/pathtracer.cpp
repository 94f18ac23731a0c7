// Host side of the drop-in `PathTracer` class (include/pathtracer.h): same public behaviour as the
// reference's PathTracing/src/pathtracer.cpp:11-365 (scene staging, setters, silent error handling), with
// the render loop (pathtracer.cpp:367-822) delegated to the HIP kernels through the C-ABI (ptk.h).
#include "pathtracer.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "host_scene.h"
#include "ptk.h"

namespace ptkhost {

// ---- glm 0.9.3.1 arithmetic used at staging time, operation order preserved -----------------------
static inline void normalize3(float* v)
{
    float sqr = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float inv = 1.0f / std::sqrt(sqr);
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

// glm mat4 * vec4 (core/type_mat4x4.inl:689-700), xyz of the result
static inline void xform(const glm::mat4& m, const float* v, float w, float* out)
{
    out[0] = m[0][0] * v[0] + m[1][0] * v[1] + m[2][0] * v[2] + m[3][0] * w;
    out[1] = m[0][1] * v[0] + m[1][1] * v[1] + m[2][1] * v[2] + m[3][1] * w;
    out[2] = m[0][2] * v[0] + m[1][2] * v[1] + m[2][2] * v[2] + m[3][2] * w;
}

// Triangle::Init, mesh.cpp:61-83
void triangle_init(StagedTriangle& t)
{
    float e1[3], e2[3];
    for (int a = 0; a < 3; a++) { e1[a] = t.v[1][a] - t.v[0][a]; e2[a] = t.v[2][a] - t.v[0][a]; }
    float d1x = t.uv[1][0] - t.uv[0][0], d1y = t.uv[1][1] - t.uv[0][1];
    float d2x = t.uv[2][0] - t.uv[0][0], d2y = t.uv[2][1] - t.uv[0][1];
    float f = 1.0f / (d1x * d2y - d2x * d1y);
    for (int a = 0; a < 3; a++)
    {
        t.tangent[a] = f * (d2y * e1[a] - d1y * e2[a]);
        t.bitangent[a] = f * (-d2x * e1[a] + d1x * e2[a]);
    }
    t.normal[0] = e1[1] * e2[2] - e2[1] * e1[2];
    t.normal[1] = e1[2] * e2[0] - e2[2] * e1[0];
    t.normal[2] = e1[0] * e2[1] - e2[0] * e1[1];
    normalize3(t.tangent);
    normalize3(t.bitangent);
    normalize3(t.normal);
}

// ---- Wavefront OBJ reader ------------------------------------------------------------------------------
// Replaces tinyobj::LoadObj as used by PathTracer::LoadObject (pathtracer.cpp:43-47): v / vt / vn / f
// (negative indices, v, v/vt, v//vn, v/vt/vn; polygons fan-triangulated), one shape per `o` or `g`
// statement that is followed by faces (tiny_obj_loader.h:2820-2900), per-face smoothing group from `s`.
bool load_obj(const std::string& file, ObjData& out)
{
    std::ifstream in(file);
    if (!in) return false;
    out = ObjData();
    ObjShape shape;
    unsigned smoothing = 0;
    std::string line;
    auto flush = [&]() {
        if (!shape.indices.empty()) out.shapes.push_back(shape);
        shape = ObjShape();
    };
    auto fix = [](int idx, size_t n) -> int {
        if (idx > 0) return idx - 1;
        if (idx < 0) return (int)n + idx;
        return -1;
    };
    while (std::getline(in, line))
    {
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') p++;
        if (*p == '\0' || *p == '#' || *p == '\r') continue;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t'))
        {
            float x = 0, y = 0, z = 0;
            std::sscanf(p + 2, "%f %f %f", &x, &y, &z);
            out.positions.push_back(x); out.positions.push_back(y); out.positions.push_back(z);
        }
        else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t'))
        {
            float x = 0, y = 0, z = 0;
            std::sscanf(p + 3, "%f %f %f", &x, &y, &z);
            out.normals.push_back(x); out.normals.push_back(y); out.normals.push_back(z);
        }
        else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t'))
        {
            float x = 0, y = 0;
            std::sscanf(p + 3, "%f %f", &x, &y);
            out.texcoords.push_back(x); out.texcoords.push_back(y);
        }
        else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t'))
        {
            std::vector<ObjIndex> face;
            const char* q = p + 2;
            while (*q)
            {
                while (*q == ' ' || *q == '\t' || *q == '\r') q++;
                if (!*q) break;
                ObjIndex ix = { -1, -1, -1 };
                char* end = 0;
                long v = std::strtol(q, &end, 10);
                if (end == q) break;
                ix.v = fix((int)v, out.positions.size() / 3);
                q = end;
                if (*q == '/')
                {
                    q++;
                    if (*q != '/')
                    {
                        long t = std::strtol(q, &end, 10);
                        if (end != q) { ix.t = fix((int)t, out.texcoords.size() / 2); q = end; }
                    }
                    if (*q == '/')
                    {
                        q++;
                        long n = std::strtol(q, &end, 10);
                        if (end != q) { ix.n = fix((int)n, out.normals.size() / 3); q = end; }
                    }
                }
                face.push_back(ix);
            }
            for (size_t k = 2; k < face.size(); k++)
            {
                shape.indices.push_back(face[0]); shape.indices.push_back(face[k - 1]); shape.indices.push_back(face[k]);
                shape.smoothing.push_back(smoothing);
            }
        }
        else if ((p[0] == 'o' || p[0] == 'g') && (p[1] == ' ' || p[1] == '\t' || p[1] == '\0' || p[1] == '\r'))
        {
            flush();
            const char* q = p + 1;
            while (*q == ' ' || *q == '\t') q++;
            std::string name(q);
            while (!name.empty() && (name.back() == '\r' || name.back() == ' ' || name.back() == '\t')) name.pop_back();
            if (p[0] == 'g') { size_t sp = name.find_first_of(" \t"); (void)sp; }
            shape.name = name;
        }
        else if (p[0] == 's' && (p[1] == ' ' || p[1] == '\t'))
        {
            const char* q = p + 2;
            while (*q == ' ' || *q == '\t') q++;
            if (!std::strncmp(q, "off", 3)) smoothing = 0;
            else { int id = std::atoi(q); smoothing = id < 0 ? 0u : (unsigned)id; }
        }
    }
    flush();
    return true;
}

}  // namespace ptkhost

using namespace ptkhost;

struct PathTracer::Impl {
    std::vector<StagedTriangle> triangles;
    std::vector<PathTracerLoader::Object> objects;
    std::vector<Image*> textures;

    glm::ivec2 resolution = glm::ivec2(0, 0);
    GLubyte* out_img = 0;
    int max_depth = 3;                                     // pathtracer.cpp:15

    float cam_pos[3] = { 0, 0, 0 }, cam_dir[3] = { 0, 0, 1 }, cam_up[3] = { 0, 1, 0 };   // :17-18
    float focal = 0.1f, fovy = 90.0f, focal_dist = 5.0f, aperture = 0.0f;               // :19-22

    int samples = 0;
    bool need_reset = false;
    bool have_resolution = false;
    bool scene_uploaded = false;
    // edits after BuildBVH: the reference's triangles point into the loaded materials, so they take effect at the next
    // RenderFrame() without a rebuild (pathtracer.cpp:250-258); the light list stays as BuildBVH collected it (:267-273)
    bool materials_dirty = false, textures_dirty = false, built_once = false;
    std::vector<int32_t> built_lights;
    bool camera_dirty = true, frame_dirty = true;

    uint64_t seed = 0;
    int device = 0, rank = 0, world = 1;
    ptk_ctx* ctx = 0;
    bool ctx_failed = false;
    std::string error;
    FlatScene flat;
    ptk_scene_desc flat_desc;

    bool ensure_ctx()
    {
        if (ctx) return true;
        if (ctx_failed) return false;
        int rc = ptk_create(&ctx, device);
        if (rc != PTK_OK) { ctx = 0; ctx_failed = true; error = "ptk_create failed: no usable HIP device"; return false; }
        ptk_set_tile(ctx, rank, world);
        return true;
    }
    void note(int rc) { if (rc != PTK_OK && ctx) error = ptk_last_error(ctx); }

    void set_texture(int objId, int elementId, int slot, const std::string& file)
    {
        // the reference does not bounds-check here (pathtracer.cpp:149); stay silent and safe
        if (objId < 0 || objId >= (int)objects.size()) return;
        if (elementId < 0 || elementId >= (int)objects[objId].elements.size()) return;
        Material& mat = objects[objId].elements[elementId].material;
        Image** slots[6] = { &mat.diffuseTex, &mat.normalTex, &mat.emissTex, &mat.roughnessTex, &mat.metallicTex, &mat.opacityTex };
        Image*& tex = *slots[slot];
        if (tex) tex->Load(file);                          // pathtracer.cpp:150-154: reload in place
        else
        {
            tex = new Image(file);                         // :155-160
            textures.push_back(tex);
        }
        textures_dirty = true;
    }
};

PathTracer::PathTracer() : m(new Impl()) {}

PathTracer::~PathTracer()
{
    if (m->ctx) ptk_destroy(m->ctx);
    for (auto t : m->textures) delete t;
    delete m;
}

// pathtracer.cpp:41-145
void PathTracer::LoadObject(const std::string& file, const glm::mat4& model)
{
    ObjData obj;
    if (!load_obj(file, obj)) return;                      // parse failure is silently ignored (:47)

    int nameStartIndex = (int)file.find_last_of('/') + 1;
    if (nameStartIndex > (int)file.size() - 1) nameStartIndex = 0;
    size_t nameEnd = file.find_last_of(".");
    int nameEndIndex = nameEnd == std::string::npos ? (int)file.size() - 1 : (int)nameEnd;
    std::string objName = nameEndIndex >= nameStartIndex ? file.substr(nameStartIndex, nameEndIndex - nameStartIndex) : std::string();
    PathTracerLoader::Object object(objName);

    const bool has_normals = !obj.normals.empty();
    const bool has_uvs = !obj.texcoords.empty();
    const int nv = (int)obj.positions.size() / 3, nn = (int)obj.normals.size() / 3, nt = (int)obj.texcoords.size() / 2;
    for (size_t i = 0; i < obj.shapes.size(); i++)
    {
        object.elements.push_back(PathTracerLoader::Element(obj.shapes[i].name));
        const ObjShape& sh = obj.shapes[i];
        for (size_t j = 0; j + 2 < sh.indices.size() + 0 && j / 3 < sh.smoothing.size(); j += 3)
        {
            StagedTriangle t;
            std::memset(&t, 0, sizeof(t));
            bool ok = true;
            for (int k = 0; k < 3; k++)
            {
                const ObjIndex& ix = sh.indices[j + k];
                if (ix.v < 0 || ix.v >= nv) { ok = false; break; }
                float p[3] = { -obj.positions[3 * ix.v], obj.positions[3 * ix.v + 1], obj.positions[3 * ix.v + 2] };   // x-negate (:74)
                xform(model, p, 1.0f, t.v[k]);
                if (has_normals && ix.n >= 0 && ix.n < nn)
                {
                    float n[3] = { -obj.normals[3 * ix.n], obj.normals[3 * ix.n + 1], obj.normals[3 * ix.n + 2] };
                    xform(model, n, 0.0f, t.n[k]);                                                                      // (:80-83)
                }
                if (has_uvs && ix.t >= 0 && ix.t < nt)
                {
                    t.uv[k][0] = obj.texcoords[2 * ix.t];
                    t.uv[k][1] = 1.0f - obj.texcoords[2 * ix.t + 1];                                                    // v-flip (:87-88)
                }
            }
            if (!ok) continue;
            triangle_init(t);
            t.smoothing = sh.smoothing[j / 3] != 0;            // :131-135
            t.objectId = (int)m->objects.size();
            t.elementId = (int)i;
            m->triangles.push_back(t);
        }
    }
    m->objects.push_back(object);
    m->scene_uploaded = false;
}

void PathTracer::SetDiffuseTextureForElement(int o, int e, const std::string& f) { m->set_texture(o, e, 0, f); }
void PathTracer::SetNormalTextureForElement(int o, int e, const std::string& f) { m->set_texture(o, e, 1, f); }
void PathTracer::SetEmissTextureForElement(int o, int e, const std::string& f) { m->set_texture(o, e, 2, f); }
void PathTracer::SetRoughnessTextureForElement(int o, int e, const std::string& f) { m->set_texture(o, e, 3, f); }
void PathTracer::SetMetallicTextureForElement(int o, int e, const std::string& f) { m->set_texture(o, e, 4, f); }
void PathTracer::SetOpacityTextureForElement(int o, int e, const std::string& f) { m->set_texture(o, e, 5, f); }

// pathtracer.cpp:243-258: bad ids are ignored; textures already bound to the element are kept
void PathTracer::SetMaterial(int objId, int elementId, Material& material)
{
    if (objId < 0 || objId >= (int)m->objects.size()) return;
    if (elementId < 0 || elementId >= (int)m->objects[objId].elements.size()) return;
    const Material& cur = m->objects[objId].elements[elementId].material;
    material.diffuseTex = cur.diffuseTex;
    material.normalTex = cur.normalTex;
    material.emissTex = cur.emissTex;
    material.roughnessTex = cur.roughnessTex;
    material.metallicTex = cur.metallicTex;
    material.opacityTex = cur.opacityTex;
    m->objects[objId].elements[elementId].material = material;
    m->materials_dirty = true;
}

// pathtracer.cpp:260-274: (re)build the acceleration structure, bind materials, collect lights —
// here: flatten to the boundary's arrays and hand them to ptk_upload_scene (BVH built there)
void PathTracer::BuildBVH()
{
    FlatScene fs;
    flatten_scene(m->triangles, m->objects, fs);
    if (!m->ensure_ctx()) return;
    ptk_scene_desc d = fs.desc();
    int rc = ptk_upload_scene(m->ctx, &d);
    m->note(rc);
    m->scene_uploaded = rc == PTK_OK;
    m->built_lights = fs.lights;
    m->built_once = m->built_once || rc == PTK_OK;
    m->materials_dirty = m->textures_dirty = false;
}

void PathTracer::ResetImage() { m->need_reset = true; }                // :276-279

void PathTracer::ClearScene()                                           // :281-295
{
    m->triangles.clear();
    m->objects.clear();
    for (auto t : m->textures) delete t;
    m->textures.clear();
    m->scene_uploaded = false;
    m->have_resolution = false;                                        // mTotalImg is freed (:292-294)
}

void PathTracer::SetOutImage(GLubyte* out) { m->out_img = out; }       // :297-300

void PathTracer::SetResolution(const glm::ivec2& res)                  // :302-306
{
    m->resolution = res;
    m->have_resolution = true;
    m->frame_dirty = true;
}

std::vector<PathTracerLoader::Object> PathTracer::GetLoadedObjects() const { return m->objects; }
const glm::ivec2 PathTracer::GetResolution() const { return m->resolution; }
const int PathTracer::GetTriangleCount() const { return (int)m->triangles.size(); }
const int PathTracer::GetTraceDepth() const { return m->max_depth; }
void PathTracer::SetTraceDepth(int depth) { m->max_depth = depth; m->frame_dirty = true; }   // :328-331

void PathTracer::SetCamera(const glm::vec3& pos, const glm::vec3& dir, const glm::vec3& up)  // :333-338
{
    m->cam_pos[0] = pos.x; m->cam_pos[1] = pos.y; m->cam_pos[2] = pos.z;
    m->cam_dir[0] = dir.x; m->cam_dir[1] = dir.y; m->cam_dir[2] = dir.z;
    m->cam_up[0] = up.x; m->cam_up[1] = up.y; m->cam_up[2] = up.z;
    m->camera_dirty = true;                                            // normalisation happens in ptk_set_camera
}

void PathTracer::SetProjection(float f, float fovy)                    // :340-350 (clamps applied in ptk_set_camera)
{
    m->focal = f;
    if (m->focal <= 0.0f) m->focal = 0.1f;
    m->fovy = fovy;
    if (m->fovy <= 0.0f) m->fovy = 0.1f;
    else if (m->fovy >= 180.0f) m->fovy = 179.5;
    m->camera_dirty = true;
}

void PathTracer::SetCameraFocalDist(float dist) { m->focal_dist = dist; m->camera_dirty = true; }      // :352-355
void PathTracer::SetCameraAperture(float aperture) { m->aperture = aperture; m->camera_dirty = true; } // :357-360
const int PathTracer::GetSamples() const { return m->ctx ? ptk_samples(m->ctx) : m->samples; }         // :362-365

void PathTracer::RenderFrame() { RenderFrames(1); }                    // :741-817

void PathTracer::RenderFrames(int count)
{
    if (count <= 0) return;
    if (!m->scene_uploaded && m->built_once && !m->triangles.empty())
        m->error = "geometry changed after BuildBVH(): call BuildBVH() again before RenderFrame()";   // (the reference would chase dangling pointers)
    if (!m->scene_uploaded || !m->have_resolution || !m->ensure_ctx()) return;
    int rc;
    if (m->materials_dirty || m->textures_dirty)
    {
        // SetMaterial / Set...TextureForElement after BuildBVH: seen by this frame, as in the reference
        FlatScene fs;
        flatten_scene(m->triangles, m->objects, fs);
        fs.lights = m->built_lights;                                    // mLights is BuildBVH's (pathtracer.cpp:267-273)
        if (m->textures_dirty)
        {
            ptk_scene_desc d = fs.desc();
            rc = ptk_upload_scene(m->ctx, &d);                          // new texels: stage everything again
        }
        else rc = ptk_update_materials(m->ctx, (int32_t)fs.materials.size(), fs.materials.data());
        m->note(rc);
        if (rc != PTK_OK) return;
        m->materials_dirty = m->textures_dirty = false;
    }
    if (m->camera_dirty)
    {
        rc = ptk_set_camera(m->ctx, m->cam_pos, m->cam_dir, m->cam_up, m->focal, m->fovy, m->focal_dist, m->aperture);
        m->note(rc);
        m->camera_dirty = false;
    }
    if (m->frame_dirty)
    {
        rc = ptk_set_frame(m->ctx, m->resolution.x, m->resolution.y, m->max_depth);
        m->note(rc);
        if (rc != PTK_OK) return;
        m->frame_dirty = false;
    }
    if (m->need_reset)                                                 // :745-751
    {
        m->note(ptk_reset(m->ctx));
        m->need_reset = false;
        m->samples = 0;
    }
    uint32_t first = (uint32_t)ptk_samples(m->ctx);
    rc = ptk_render(m->ctx, first, (uint32_t)count, m->seed);         // mSamples += count (:753)
    m->note(rc);
    m->samples = ptk_samples(m->ctx);
    if (rc == PTK_OK && m->out_img) m->note(ptk_resolve_rgb8(m->ctx, m->out_img));   // :802-812 into the caller's buffer
}

void PathTracer::Exit() { if (m->ctx) ptk_request_exit(m->ctx); }      // :819-822

// ---- extensions -----------------------------------------------------------------------------------------
void PathTracer::SetSeed(uint64_t seed) { m->seed = seed; }
void PathTracer::SetDevice(int ordinal) { if (!m->ctx) m->device = ordinal; }
void PathTracer::SetTile(int rank, int world)
{
    m->rank = rank; m->world = world;
    if (m->ctx) m->note(ptk_set_tile(m->ctx, rank, world));
}
bool PathTracer::ReadAccumulation(float* out)
{
    if (!m->ctx || !out) return false;
    int rc = ptk_read_accum(m->ctx, out);
    m->note(rc);
    return rc == PTK_OK;
}
std::string PathTracer::LastError() const { return m->error; }
ptk_ctx* PathTracer::Context() { m->ensure_ctx(); return m->ctx; }
const ptk_scene_desc* PathTracer::StagedScene()
{
    flatten_scene(m->triangles, m->objects, m->flat);
    m->flat_desc = m->flat.desc();
    return &m->flat_desc;
}
