// The reference's render loop (PathTracerLoop, main.cpp:3563-3618, with SendObjectsToPathTracer previewer.cpp:770-817)
// written against THIS repository's drop-in header and libptk.so - the native C++ use of the boundary, no Python:
//   dropin_render_loop <cornell.obj> <width> <height> <depth> <frames> <out.rgb>
// loads the OBJ, makes the element called "light" emissive and the walls coloured as the Cornell configs do, renders
// `frames` RenderFrame() calls into the caller-owned RGB8 buffer (rows bottom-up) and writes that buffer out.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "pathtracer.h"

int main(int argc, char** argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage: %s obj width height depth frames out.rgb\n", argv[0]); return 2; }
    const std::string obj = argv[1];
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), depth = std::atoi(argv[4]), frames = std::atoi(argv[5]);

    PathTracer pathTracer;
    pathTracer.ClearScene();
    pathTracer.LoadObject(obj, glm::mat4(1.0f));                       // model matrix = identity
    std::vector<PathTracerLoader::Object> objects = pathTracer.GetLoadedObjects();
    if (objects.size() != 1) { std::fprintf(stderr, "LoadObject failed\n"); return 3; }
    for (size_t e = 0; e < objects[0].elements.size(); e++)
    {
        Material m;                                                    // reference defaults (mesh.h:21-59)
        const std::string& name = objects[0].elements[e].name;
        m.diffuse = glm::vec3(0.75f, 0.75f, 0.75f);
        if (name == "left") m.diffuse = glm::vec3(0.75f, 0.25f, 0.25f);
        if (name == "right") m.diffuse = glm::vec3(0.25f, 0.75f, 0.25f);
        if (name == "light") { m.emissive = glm::vec3(1.0f, 1.0f, 1.0f); m.emissiveIntensity = 1.0f; }
        pathTracer.SetMaterial(0, (int)e, m);
    }
    pathTracer.BuildBVH();
    if (pathTracer.GetTriangleCount() != 12) { std::fprintf(stderr, "expected the 12-triangle Cornell box\n"); return 4; }

    pathTracer.SetResolution(glm::ivec2(w, h));
    pathTracer.SetTraceDepth(depth);
    pathTracer.SetCamera(glm::vec3(0.0f, 0.0f, -3.5f), glm::vec3(0.0f, 0.0f, 1.0f), glm::vec3(0.0f, 1.0f, 0.0f));
    pathTracer.SetProjection(0.05f, 70.0f);
    pathTracer.SetCameraFocalDist(3.5f);
    pathTracer.SetCameraAperture(0.0f);
    std::vector<GLubyte> texData((size_t)w * h * 3);                    // `new GLubyte[w*h*3]`, main.cpp:3435
    pathTracer.SetOutImage(texData.data());
    pathTracer.ResetImage();
    for (int f = 0; f < frames; f++) pathTracer.RenderFrame();          // one sample per pixel each
    if (pathTracer.GetSamples() != frames) { std::fprintf(stderr, "GetSamples() = %d\n", pathTracer.GetSamples()); return 5; }

    FILE* out = std::fopen(argv[6], "wb");
    if (!out) return 6;
    std::fwrite(texData.data(), 1, texData.size(), out);
    std::fclose(out);
    unsigned long long sum = 0;
    for (GLubyte b : texData) sum += b;
    std::printf("samples %d triangles %d checksum %llu\n", pathTracer.GetSamples(), pathTracer.GetTriangleCount(), sum);
    pathTracer.Exit();
    return 0;
}
