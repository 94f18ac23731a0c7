// Headless scene layer: `.pts` reader / writer (reader grammar of the reference's LoadScene,
// PathTracing/src/main.cpp:261-438), Previewer-style TRS matrix in DEGREES (previewer.h:104-112),
// Euler camera (previewer.cpp:883-902), and the push into PathTracer that
// Previewer::SendObjectsToPathTracer / SetPathTracerCamera / PathTracerLoop perform
// (previewer.cpp:770-817, :924-930, main.cpp:3570-3581).  Also flatten_scene(): BuildBVH's material
// binding and light list (pathtracer.cpp:267-273) expressed as flat arrays.
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "host_scene.h"

namespace ptkhost {

ptk_scene_desc FlatScene::desc() const
{
    ptk_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.num_triangles = (int32_t)material.size();
    d.verts = verts.data(); d.normals = normals.data(); d.uvs = uvs.data(); d.tbn = tbn.data();
    d.smoothing = smoothing.data(); d.material = material.data();
    d.num_materials = (int32_t)materials.size(); d.materials = materials.data();
    d.num_textures = (int32_t)textures.size(); d.textures = textures.data();
    d.texels = texels.data(); d.texel_bytes = (int64_t)texels.size();
    d.num_lights = (int32_t)lights.size(); d.lights = lights.data();
    return d;
}

void flatten_scene(const std::vector<StagedTriangle>& tris, const std::vector<PathTracerLoader::Object>& objects, FlatScene& out)
{
    out = FlatScene();
    // one material per (object, element); one texture per distinct Image*
    std::vector<int> base(objects.size() + 1, 0);
    std::map<const Image*, int> tex_index;
    for (size_t i = 0; i < objects.size(); i++)
    {
        base[i] = (int)out.materials.size();
        for (const auto& e : objects[i].elements)
        {
            const Material& m = e.material;
            ptk_material pm;
            pm.type = m.type == MaterialType::TRANSLUCENT ? 1 : 0;
            pm.diffuse[0] = m.diffuse.x; pm.diffuse[1] = m.diffuse.y; pm.diffuse[2] = m.diffuse.z;
            pm.specular[0] = m.specular.x; pm.specular[1] = m.specular.y; pm.specular[2] = m.specular.z;
            pm.emissive[0] = m.emissive.x; pm.emissive[1] = m.emissive.y; pm.emissive[2] = m.emissive.z;
            pm.emissive_intensity = m.emissiveIntensity; pm.roughness = m.roughness;
            pm.reflectiveness = m.reflectiveness; pm.translucency = m.translucency; pm.ior = m.ior;
            Image* slots[6] = { m.diffuseTex, m.normalTex, m.emissTex, m.roughnessTex, m.metallicTex, m.opacityTex };
            for (int k = 0; k < 6; k++)
            {
                pm.tex[k] = -1;
                Image* img = slots[k];
                if (!img) continue;
                auto it = tex_index.find(img);
                if (it == tex_index.end())
                {
                    ptk_texture t;
                    // an Image whose load failed has mData == 0 and samples as 0 (image.cpp:65-66): zero extent
                    t.width = img->data() ? img->width() : 0;
                    t.height = img->data() ? img->height() : 0;
                    t.offset = (int64_t)out.texels.size();
                    size_t bytes = (size_t)t.width * t.height * 4;
                    if (bytes) out.texels.insert(out.texels.end(), img->data(), img->data() + bytes);
                    it = tex_index.emplace(img, (int)out.textures.size()).first;
                    out.textures.push_back(t);
                }
                pm.tex[k] = it->second;
            }
            out.materials.push_back(pm);
        }
    }
    const size_t n = tris.size();
    out.verts.resize(n * 9); out.normals.resize(n * 9); out.uvs.resize(n * 6); out.tbn.resize(n * 9);
    out.smoothing.resize(n); out.material.resize(n);
    // which materials emit: glm::length(emissive) >= EPS, constant colour only (pathtracer.cpp:271-272)
    std::vector<uint8_t> emits(out.materials.size(), 0);
    for (size_t k = 0; k < out.materials.size(); k++)
    {
        const ptk_material& pm = out.materials[k];
        const float sqr = pm.emissive[0] * pm.emissive[0] + pm.emissive[1] * pm.emissive[1] + pm.emissive[2] * pm.emissive[2];
        emits[k] = std::sqrt(sqr) >= EPS ? 1 : 0;
    }
    parallel_for(n, [&](size_t i) {
        const StagedTriangle& t = tris[i];
        float* v = &out.verts[i * 9]; float* nn = &out.normals[i * 9]; float* uv = &out.uvs[i * 6]; float* tb = &out.tbn[i * 9];
        for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) { v[k * 3 + a] = t.v[k][a]; nn[k * 3 + a] = t.n[k][a]; }
        for (int k = 0; k < 3; k++) for (int a = 0; a < 2; a++) uv[k * 2 + a] = t.uv[k][a];
        for (int a = 0; a < 3; a++) { tb[a] = t.normal[a]; tb[3 + a] = t.tangent[a]; tb[6 + a] = t.bitangent[a]; }
        out.smoothing[i] = t.smoothing ? 1 : 0;
        out.material[i] = base[t.objectId] + t.elementId;
    }, 16384);
    // the light list, in triangle order (mLights, pathtracer.cpp:267-273)
    for (size_t i = 0; i < n; i++)
        if (emits[out.material[i]]) out.lights.push_back((int32_t)i);
}

// ---- glm 0.9.3.1 matrix_transform (gtc/matrix_transform.inl:31-95), degrees -------------------------------
namespace {
struct V4 { float x, y, z, w; };
inline V4 mul(const glm::vec4& a, float s) { return V4{ a.x * s, a.y * s, a.z * s, a.w * s }; }
inline V4 add(V4 a, V4 b) { return V4{ a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
inline glm::vec4 tov(V4 a) { return glm::vec4(a.x, a.y, a.z, a.w); }

glm::mat4 glm_translate(const glm::mat4& m, const float v[3])
{
    glm::mat4 r = m;
    V4 t = add(add(add(mul(m[0], v[0]), mul(m[1], v[1])), mul(m[2], v[2])), V4{ m[3].x, m[3].y, m[3].z, m[3].w });
    r[3] = tov(t);
    return r;
}
glm::mat4 glm_rotate(const glm::mat4& m, float angle, float ax, float ay, float az)
{
    const float pi = float(3.1415926535897932384626433832795);
    float a = angle * (pi / float(180));
    float c = std::cos(a), s = std::sin(a);
    float sq = ax * ax + ay * ay + az * az;
    float inv = 1.0f / std::sqrt(sq);
    float axis[3] = { ax * inv, ay * inv, az * inv };
    float temp[3] = { (1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2] };
    float R[3][3];
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = 0 + temp[0] * axis[1] + s * axis[2];
    R[0][2] = 0 + temp[0] * axis[2] - s * axis[1];
    R[1][0] = 0 + temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = 0 + temp[1] * axis[2] + s * axis[0];
    R[2][0] = 0 + temp[2] * axis[0] + s * axis[1];
    R[2][1] = 0 + temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];
    glm::mat4 r(0.0f);
    for (int k = 0; k < 3; k++) r[k] = tov(add(add(mul(m[0], R[k][0]), mul(m[1], R[k][1])), mul(m[2], R[k][2])));
    r[3] = m[3];
    return r;
}
glm::mat4 glm_scale(const glm::mat4& m, const float v[3])
{
    glm::mat4 r(0.0f);
    r[0] = tov(mul(m[0], v[0])); r[1] = tov(mul(m[1], v[1])); r[2] = tov(mul(m[2], v[2])); r[3] = m[3];
    return r;
}
glm::mat4 mat_mul(const glm::mat4& a, const glm::mat4& b)    // core/type_mat4x4.inl:757-779
{
    glm::mat4 r(0.0f);
    for (int k = 0; k < 4; k++)
        r[k] = tov(add(add(add(mul(a[0], b[k].x), mul(a[1], b[k].y)), mul(a[2], b[k].z)), mul(a[3], b[k].w)));
    return r;
}
inline float glm_mod(float x, float y) { return x - y * std::floor(x / y); }
}  // namespace

glm::mat4 trs_matrix(const float loc[3], const float rot[3], const float scl[3])
{
    glm::mat4 T = glm_translate(glm::mat4(1.0f), loc);
    glm::mat4 R = glm_rotate(T, rot[0], 1.0f, 0.0f, 0.0f);
    R = glm_rotate(R, rot[1], 0.0f, 1.0f, 0.0f);
    R = glm_rotate(R, rot[2], 0.0f, 0.0f, 1.0f);
    return glm_scale(R, scl);
}

void euler_camera(const float rotation[3], float dir[3], float up[3])
{
    float r[3];
    for (int k = 0; k < 3; k++) { r[k] = glm_mod(rotation[k], 360.0f); if (r[k] < 0.0f) r[k] += 360.0f; }
    glm::mat4 Rx = glm_rotate(glm::mat4(1.0f), r[0], 1.0f, 0.0f, 0.0f);
    glm::mat4 Ry = glm_rotate(glm::mat4(1.0f), r[1], 0.0f, 1.0f, 0.0f);
    glm::mat4 Rz = glm_rotate(glm::mat4(1.0f), r[2], 0.0f, 0.0f, 1.0f);
    glm::mat4 M = mat_mul(mat_mul(Rz, Ry), Rx);
    auto apply = [&](float x, float y, float z, float w, float* out) {
        float o[3] = { M[0][0] * x + M[1][0] * y + M[2][0] * z + M[3][0] * w,
                       M[0][1] * x + M[1][1] * y + M[2][1] * z + M[3][1] * w,
                       M[0][2] * x + M[1][2] * y + M[2][2] * z + M[3][2] * w };
        float sq = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
        float inv = 1.0f / std::sqrt(sq);
        out[0] = o[0] * inv; out[1] = o[1] * inv; out[2] = o[2] * inv;
    };
    apply(0.0f, 0.0f, 1.0f, 1.0f, dir);
    apply(0.0f, 1.0f, 0.0f, 1.0f, up);
}

// ---- .pts ---------------------------------------------------------------------------------------------
namespace {
int compare_versions(const std::string& a, const std::string& b)   // main.cpp:205-228
{
    std::istringstream sa(a), sb(b);
    for (int i = 0; i < 3; i++)
    {
        int x = 0, y = 0; char dot;
        sa >> x; sb >> y; sa >> dot; sb >> dot;
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}
void chomp(std::string& s) { while (!s.empty() && (s.back() == '\r' || s.back() == '\n')) s.pop_back(); }
}  // namespace

bool read_pts(const std::string& path, SceneFile& out, std::string* err)
{
    auto fail = [&](const char* m) { if (err) *err = m; return false; };
    std::ifstream fr(path);
    if (!fr) return fail("cannot open file");
    std::string line;
    if (!std::getline(fr, line)) return fail("empty file");
    chomp(line);
    if (line != "Path Tracer Scene File") return fail("not a Path Tracer Scene File");     // main.cpp:267
    if (!std::getline(fr, line)) return fail("missing version");
    chomp(line);
    std::string ver = line.substr(line.find_first_of('=') + 1);
    if (compare_versions(ver, "2.0.0") < 0) return fail("file version < 2.0.0");           // main.cpp:269-270, :74-75
    out = SceneFile();
    if (!(fr >> out.trace_depth)) return fail("trace depth");
    if (!(fr >> out.width >> out.height)) return fail("resolution");
    if (!(fr >> out.auto_res)) return fail("autoRes");
    if (!(fr >> out.cam_pos[0] >> out.cam_pos[1] >> out.cam_pos[2])) return fail("camera position");
    if (!(fr >> out.cam_rot[0] >> out.cam_rot[1] >> out.cam_rot[2])) return fail("camera rotation");
    if (!(fr >> out.focal_dist)) return fail("focal distance");
    if (!(fr >> out.camera_f)) return fail("camera F");
    int nobj = 0;
    if (!(fr >> nobj)) return fail("object count");
    std::getline(fr, line);
    for (int i = 0; i < nobj; i++)
    {
        SceneObject o;
        if (!std::getline(fr, o.file)) return fail("object path");
        chomp(o.file);
        if (!std::getline(fr, o.name)) return fail("object name");
        chomp(o.name);
        if (!(fr >> o.location[0] >> o.location[1] >> o.location[2])) return fail("location");
        if (!(fr >> o.rotation[0] >> o.rotation[1] >> o.rotation[2])) return fail("rotation");
        if (!(fr >> o.scale[0] >> o.scale[1] >> o.scale[2])) return fail("scale");
        int nel = 0;
        if (!(fr >> nel)) return fail("element count");
        std::getline(fr, line);
        for (int j = 0; j < nel; j++)
        {
            SceneElement e;
            if (!std::getline(fr, e.name)) return fail("element name");
            chomp(e.name);
            Material& m = e.material;
            int type = 0;
            if (!(fr >> m.diffuse.x >> m.diffuse.y >> m.diffuse.z)) return fail("diffuse");
            if (!(fr >> m.specular.x >> m.specular.y >> m.specular.z)) return fail("specular");
            if (!(fr >> m.emissive.x >> m.emissive.y >> m.emissive.z)) return fail("emissive");
            if (!(fr >> m.emissiveIntensity)) return fail("emissive intensity");
            if (!(fr >> type >> m.roughness >> m.reflectiveness >> m.translucency >> m.ior)) return fail("material scalars");
            m.type = (MaterialType)type;
            std::getline(fr, line);                                     // rest of the scalar line (main.cpp:413)
            for (int k = 0; k < 6; k++)
            {
                if (!std::getline(fr, e.tex_files[k])) return fail("texture path");
                chomp(e.tex_files[k]);
            }
            o.elements.push_back(e);
        }
        out.objects.push_back(o);
    }
    return true;
}

bool write_pts(const std::string& path, const SceneFile& s)
{
    std::ofstream fw(path);
    if (!fw) return false;
    fw.precision(9);
    fw << "Path Tracer Scene File\nVersion=2.1.0\n";
    fw << s.trace_depth << "\n" << s.width << " " << s.height << "\n" << s.auto_res << "\n";
    fw << s.cam_pos[0] << " " << s.cam_pos[1] << " " << s.cam_pos[2] << "\n";
    fw << s.cam_rot[0] << " " << s.cam_rot[1] << " " << s.cam_rot[2] << "\n";
    fw << s.focal_dist << "\n" << s.camera_f << "\n" << s.objects.size() << "\n";
    for (const auto& o : s.objects)
    {
        fw << o.file << "\n" << o.name << "\n";
        fw << o.location[0] << " " << o.location[1] << " " << o.location[2] << "\n";
        fw << o.rotation[0] << " " << o.rotation[1] << " " << o.rotation[2] << "\n";
        fw << o.scale[0] << " " << o.scale[1] << " " << o.scale[2] << "\n";
        fw << o.elements.size() << "\n";
        for (const auto& e : o.elements)
        {
            const Material& m = e.material;
            fw << e.name << "\n";
            fw << m.diffuse.x << " " << m.diffuse.y << " " << m.diffuse.z << "\n";
            fw << m.specular.x << " " << m.specular.y << " " << m.specular.z << "\n";
            fw << m.emissive.x << " " << m.emissive.y << " " << m.emissive.z << "\n";
            fw << m.emissiveIntensity << "\n";
            fw << (int)m.type << " " << m.roughness << " " << m.reflectiveness << " " << m.translucency << " " << m.ior << "\n";
            for (int k = 0; k < 6; k++) fw << e.tex_files[k] << "\n";
        }
    }
    return (bool)fw;
}

bool send_scene(const SceneFile& s, PathTracer& pt)
{
    pt.ClearScene();                                                    // main.cpp:3572
    for (size_t i = 0; i < s.objects.size(); i++)                       // previewer.cpp:770-817
    {
        const SceneObject& o = s.objects[i];
        pt.LoadObject(o.file, trs_matrix(o.location, o.rotation, o.scale));
        for (size_t j = 0; j < o.elements.size(); j++)
        {
            Material mat = o.elements[j].material;
            pt.SetMaterial((int)i, (int)j, mat);
            const std::string* f = o.elements[j].tex_files;
            if (!f[0].empty()) pt.SetDiffuseTextureForElement((int)i, (int)j, f[0]);
            if (!f[1].empty()) pt.SetNormalTextureForElement((int)i, (int)j, f[1]);
            if (!f[2].empty()) pt.SetEmissTextureForElement((int)i, (int)j, f[2]);
            if (!f[3].empty()) pt.SetRoughnessTextureForElement((int)i, (int)j, f[3]);
            if (!f[4].empty()) pt.SetMetallicTextureForElement((int)i, (int)j, f[4]);
            if (!f[5].empty()) pt.SetOpacityTextureForElement((int)i, (int)j, f[5]);
        }
    }
    pt.BuildBVH();
    float dir[3], up[3];
    euler_camera(s.cam_rot, dir, up);                                   // previewer.cpp:883-902
    pt.SetCamera(glm::vec3(s.cam_pos[0], s.cam_pos[1], s.cam_pos[2]), glm::vec3(dir[0], dir[1], dir[2]),
                 glm::vec3(up[0], up[1], up[2]));                       // previewer.cpp:924-930
    pt.SetProjection(kPtsFocal, kPtsFovy);
    pt.SetCameraFocalDist(s.focal_dist);
    pt.SetCameraAperture(kPtsFocal / s.camera_f);
    pt.SetResolution(glm::ivec2(s.width, s.height));                    // main.cpp:3578-3581
    pt.SetTraceDepth(s.trace_depth);
    pt.ResetImage();
    return true;
}

}  // namespace ptkhost
