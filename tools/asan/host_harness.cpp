#include <cstdio>
#include <cstring>
#include <string>
#include "pathtracer.h"
#include "host_scene.h"
#include "bvh_build.h"
using namespace ptkhost;
int main(int argc, char** argv)
{
    long tris = 0, nodes = 0;
    for (int i = 1; i < argc; i++)
    {
        const std::string f = argv[i];
        PathTracer pt;
        if (f.size() > 4 && f.substr(f.size() - 4) == ".pts")
        {
            SceneFile s; std::string err;
            if (read_pts(f, s, &err)) send_scene(s, pt);
        }
        else pt.LoadObject(f, glm::mat4(1.0f));
        tris += pt.GetTriangleCount();
        const ptk_scene_desc* d = pt.StagedScene();
        if (d && d->num_triangles > 0)
        {
            ptk::BuiltBvh b;
            if (ptk::build_bvh(d->verts, d->num_triangles, 32, 4, b)) nodes += b.num_nodes;
        }
    }
    std::printf("triangles %ld nodes %ld\n", tris, nodes);
    return 0;
}
