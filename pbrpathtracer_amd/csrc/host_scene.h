// Internal host-side scene staging shared by pathtracer.cpp / scene_io.cpp / ptk_host.cpp.
#pragma once

#include <algorithm>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "pathtracer.h"
#include "ptk.h"

namespace ptkhost {

// the reference's Triangle (mesh.h:71-96) without the Material pointer
struct StagedTriangle {
    float v[3][3];
    float n[3][3];
    float uv[3][2];
    float normal[3], tangent[3], bitangent[3];
    bool smoothing;
    int objectId, elementId;
};

struct ObjIndex { int v, t, n; };
struct ObjShape {
    std::string name;
    std::vector<ObjIndex> indices;        // 3 per triangle
    std::vector<unsigned> smoothing;      // per triangle
};
struct ObjData {
    std::vector<float> positions, normals, texcoords;
    std::vector<ObjShape> shapes;
};

// host-side data parallelism for scene ingest (a million triangles: parsing, staging, flattening)
inline size_t host_threads() { const unsigned h = std::thread::hardware_concurrency(); return std::max<size_t>(1, std::min<size_t>(h ? h : 1, 32)); }
// f(k) for k in [0, n), spread over the host's cores (one contiguous block of indices per thread)
template <class F>
void parallel_for(size_t n, F f, size_t min_per_thread = 1)
{
    const size_t t = std::max<size_t>(1, std::min<size_t>(host_threads(), n / std::max<size_t>(1, min_per_thread)));
    if (t <= 1) { for (size_t k = 0; k < n; k++) f(k); return; }
    std::vector<std::thread> pool;
    for (size_t w = 0; w < t; w++)
        pool.emplace_back([=, &f] { const size_t a = n * w / t, b = n * (w + 1) / t; for (size_t k = a; k < b; k++) f(k); });
    for (auto& th : pool) th.join();
}

bool load_obj(const std::string& file, ObjData& out);
void triangle_init(StagedTriangle& t);

// flat arrays in the layout of ptk_scene_desc
struct FlatScene {
    std::vector<float> verts, normals, uvs, tbn;
    std::vector<uint8_t> smoothing;
    std::vector<int32_t> material;
    std::vector<ptk_material> materials;
    std::vector<ptk_texture> textures;
    std::vector<uint8_t> texels;
    std::vector<int32_t> lights;
    ptk_scene_desc desc() const;
};

void flatten_scene(const std::vector<StagedTriangle>& tris, const std::vector<PathTracerLoader::Object>& objects, FlatScene& out);

// ---- scene layer (reference previewer.{h,cpp} + main.cpp .pts reader), headless ---------------------------
struct SceneElement {
    std::string name;
    Material material;                    // texture pointers unused here
    std::string tex_files[6];             // diffuse, normal, emissive, roughness, metallic, opacity ("" = none)
};
struct SceneObject {
    std::string file, name;
    float location[3] = { 0, 0, 0 }, rotation[3] = { 0, 0, 0 }, scale[3] = { 1, 1, 1 };
    std::vector<SceneElement> elements;
};
struct SceneFile {
    int trace_depth = 3;
    int width = 0, height = 0;
    int auto_res = 0;
    float cam_pos[3] = { 0, 0, -10 };     // previewer.cpp:11
    float cam_rot[3] = { 0, 0, 0 };
    float focal_dist = 5.0f;              // previewer.cpp:15
    float camera_f = 32.0f;               // previewer.cpp:16
    std::vector<SceneObject> objects;
};
constexpr float kPtsFocal = 0.05f;        // previewer.cpp:13
constexpr float kPtsFovy = 70.0f;         // previewer.cpp:14

bool read_pts(const std::string& path, SceneFile& out, std::string* err);
bool write_pts(const std::string& path, const SceneFile& s);
glm::mat4 trs_matrix(const float loc[3], const float rot_deg[3], const float scl[3]);   // previewer.h:104-112
void euler_camera(const float rot_deg[3], float dir[3], float up[3]);                  // previewer.cpp:883-902
// Previewer::SendObjectsToPathTracer + SetPathTracerCamera + the PathTracerLoop set-up (main.cpp:3570-3581)
bool send_scene(const SceneFile& s, PathTracer& pt);

}  // namespace ptkhost
