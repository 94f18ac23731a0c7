// Drop-in C++ host API for the MI355X render path: the public surface of the reference's
// `class PathTracer` (reference PathTracing/src/pathtracer.h:48-131), `Material` / `MaterialType`
// (mesh.h:15-59), `Image` (image.h:7-28) and `PathTracerLoader::{Object,Element}` (pathtracer.h:13-46),
// with the same names, argument meaning, call-order contract and silent error behaviour.
// The bodies stage the scene into flat arrays and call the C-ABI of include/ptk.h; the per-pixel
// render loop runs in HIP kernels on the GPU (there is no CPU render path in this library).
//
//   Load*/Set*  ->  BuildBVH()  ->  SetResolution  ->  SetOutImage  ->  ResetImage  ->  RenderFrame() x N
//
// Extensions that the reference does not have are grouped at the end of the class and marked.
#ifndef PTK_PATHTRACER_H
#define PTK_PATHTRACER_H

#include <cstdint>
#include <string>
#include <vector>

#include "ptk_glm.h"

#ifndef __glew_h__
typedef unsigned char GLubyte;   // the only thing the reference takes from <GL/glew.h> here (pathtracer.h:8,59)
#endif

struct ptk_ctx;
struct ptk_scene_desc;

const float EPS = 0.00001f;      // mesh.h:12
const float INF = (float)0xFFFF; // mesh.h:13

// image.h:7-28 — RGBA8 image, longest side <= 1024 after Load (image.cpp:38-61)
class Image
{
private:
    std::string mFilename;
    int mWidth;
    int mHeight;
    unsigned char* mData;

public:
    Image();
    Image(const std::string& filename);
    ~Image();
    Image(const Image&) = delete;
    Image& operator=(const Image&) = delete;

    const int width() const;
    const int height() const;
    void Load(const std::string& filename);
    glm::vec4 tex2D(const glm::vec2& uv);   // host-side probe of the sampler semantics (image.cpp:63-86)
    unsigned char* data();
};

enum class MaterialType
{
    OPAQUE,
    TRANSLUCENT
};

// mesh.h:21-59, same field names and defaults
struct Material
{
    MaterialType type;
    glm::vec3 diffuse;
    glm::vec3 specular;
    glm::vec3 emissive;

    float emissiveIntensity;
    float roughness;
    float reflectiveness;
    float translucency;
    float ior;

    Image* diffuseTex;
    Image* normalTex;
    Image* emissTex;
    Image* roughnessTex;
    Image* metallicTex;
    Image* opacityTex;

    Material() :
        type(MaterialType::OPAQUE),
        emissiveIntensity(1.0f),
        roughness(1.0f),
        reflectiveness(0.0f),
        translucency(1.0f),
        ior(1.5f),
        diffuseTex(0),
        normalTex(0),
        emissTex(0),
        roughnessTex(0),
        metallicTex(0),
        opacityTex(0)
    {
        diffuse = glm::vec3(1.0f);
        specular = glm::vec3(1.0f);
        emissive = glm::vec3(0.0f);
    }
};

namespace PathTracerLoader
{
    struct Element
    {
        std::string name;
        Material material;
        Element() { name = ""; }
        Element(const std::string& name) { this->name = name; }
    };

    struct Object
    {
        std::string name;
        std::vector<Element> elements;
        Object() { name = ""; }
        Object(const std::string& name) { this->name = name; }
    };
}

class PathTracer
{
public:
    PathTracer();
    ~PathTracer();
    PathTracer(const PathTracer&) = delete;
    PathTracer& operator=(const PathTracer&) = delete;

    // ---- the reference's public API (pathtracer.h:100-130), unchanged ----
    void LoadObject(const std::string& file, const glm::mat4& model);

    void SetDiffuseTextureForElement(int objId, int elementId, const std::string& file);
    void SetNormalTextureForElement(int objId, int elementId, const std::string& file);
    void SetEmissTextureForElement(int objId, int elementId, const std::string& file);
    void SetRoughnessTextureForElement(int objId, int elementId, const std::string& file);
    void SetMetallicTextureForElement(int objId, int elementId, const std::string& file);
    void SetOpacityTextureForElement(int objId, int elementId, const std::string& file);

    void SetMaterial(int objId, int elementId, Material& material);

    void BuildBVH();
    void ResetImage();
    void ClearScene();

    const int GetSamples() const;
    const int GetTriangleCount() const;
    const int GetTraceDepth() const;
    void SetTraceDepth(int depth);
    void SetOutImage(GLubyte* out);
    void SetResolution(const glm::ivec2& res);
    const glm::ivec2 GetResolution() const;
    std::vector<PathTracerLoader::Object> GetLoadedObjects() const;

    void SetCamera(const glm::vec3& pos, const glm::vec3& dir, const glm::vec3& up);
    void SetProjection(float f, float fovy);
    void SetCameraFocalDist(float dist);
    void SetCameraAperture(float aperture);
    void RenderFrame();
    void Exit();

    // ---- extensions (not in the reference) ----
    // The reference seeds one std::mt19937 from std::random_device (pathtracer.cpp:11); here the RNG
    // is counter-based and keyed on (seed, pixel, sample index), default seed 0.
    void SetSeed(uint64_t seed);
    // EXPERIMENTAL (never executed: headless build boxes).  SetOutImage for a display path that stays on the GPU: the 8-bit image
    // (the layout of texData) is written into this OpenGL buffer object - the viewer's GL_PIXEL_UNPACK_BUFFER - instead of a host
    // buffer.  Call it on the thread whose OpenGL context is current (the viewer's GUI thread): the buffer is registered HERE
    // (ptk_bind_gl_buffer, include/ptk.h), not inside RenderFrame(), which the viewer runs on a thread without a context; the call
    // waits for a RenderFrame() in flight.  Read the buffer only between two RenderFrame() calls.  0 lets it go (same thread).
    void SetOutGLBuffer(unsigned int gl_buffer);
    // ... or into W*H*3 bytes of this GPU's memory (ptk_bind_out_device); NULL or SetOutImage(ptr) switches back.
    void SetOutDeviceImage(void* device_rgb8);
    // One process per GPU: which device this instance drives, and which 16x16 pixel tiles it owns.
    void SetDevice(int ordinal);
    void SetTile(int rank, int world);
    // `count` RenderFrame() calls in one kernel launch (identical image; the accumulator stays in
    // registers between samples).  The RGB8 host copy happens once at the end.
    void RenderFrames(int count);
    // mTotalImg (float RGB, rows bottom-up), W*H*3 floats
    bool ReadAccumulation(float* out);
    // last error text of the device layer ("" when none); the reference API itself stays silent
    std::string LastError() const;
    ptk_ctx* Context();
    // the staged scene as the flat arrays BuildBVH() uploads (include/ptk.h); valid until the scene changes
    const ptk_scene_desc* StagedScene();

    struct Impl;
private:
    Impl* m;
};

#endif
