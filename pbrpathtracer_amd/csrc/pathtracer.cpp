// Host side of the drop-in `PathTracer` class (include/pathtracer.h): same public behaviour as the
// reference's PathTracing/src/pathtracer.cpp:11-365 (scene staging, setters, silent error handling), with
// the render loop (pathtracer.cpp:367-822) delegated to the HIP kernels through the C-ABI (ptk.h).
#include "pathtracer.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <mutex>
#include <sstream>
#include <thread>

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
// (negative indices, v, v/vt, v//vn, v/vt/vn; quads split along their shorter diagonal and larger polygons ear-clipped exactly as
// tinyobj 2.0.0 triangulates them, tiny_obj_loader.h:1449-1856 - the reference's triangles ARE that triangulation), one shape
// per `o` or `g` statement that is followed by faces (:2820-2900; a bare `g` / `o` without a blank behind it is no statement
// there), per-face smoothing group from `s`.
// Large files (a million triangles is 60 MB of text) are parsed by all cores: the file is cut at line ends into one chunk
// per thread; a first pass counts the v / vn / vt statements of every chunk (relative face indices and the output offsets
// need the counts before a line), a second parses numbers and faces in place; shapes and smoothing groups, which are
// sequential state, are stitched from per-chunk fragments afterwards.  Statement semantics are those of the line-by-line
// reader this replaces (and of the fixtures recorded from the reference's tinyobjloader).
namespace {

struct ObjFragment {
    bool new_shape = false;               // begins with an `o` / `g` statement
    std::string name;
    std::vector<ObjIndex> indices;        // the faces' corners, concatenated
    std::vector<unsigned> smoothing;      // per face; kUnknownSmoothing until the chunk's first `s`: the value flows in from before
    std::vector<uint32_t> sizes;          // corners per face - only kept once a face is not a triangle (has_poly)
    bool has_poly = false;
    char kind = 0;                        // 'g' / 'o': which statement began it (they keep an empty shape by different rules, see load_obj)
    bool any_f = false, any_l = false, any_p = false;     // `f` (of any corner count), `l`, `p` statements in it
};
constexpr unsigned kUnknownSmoothing = 0xffffffffu;

struct ObjChunk {
    const char* begin = nullptr; const char* end = nullptr;
    size_t nv = 0, nvn = 0, nvt = 0;      // statements in this chunk
    size_t v0 = 0, vn0 = 0, vt0 = 0;      // ... before it
    std::vector<ObjFragment> frags;
    bool has_s = false; unsigned last_s = 0;
};

inline const char* skip_blank(const char* p, const char* e) { while (p < e && (*p == ' ' || *p == '\t')) p++; return p; }
inline const char* line_end(const char* p, const char* e) { while (p < e && *p != '\n') p++; return p; }

// up to `count` floats of one line (sscanf("%f %f %f") semantics: missing ones stay 0)
inline void parse_floats(const char* p, const char* e, float* out, int count)
{
    for (int k = 0; k < count; k++)
    {
        p = skip_blank(p, e);
        if (p >= e || *p == '\r') return;
        char* q = nullptr;
        const float v = std::strtof(p, &q);
        if (q == p) return;
        out[k] = v; p = q;
    }
}

void count_chunk(ObjChunk& c)
{
    for (const char* p = c.begin; p < c.end;)
    {
        const char* e = line_end(p, c.end);
        const char* q = skip_blank(p, e);
        if (q + 1 < e && q[0] == 'v')
        {
            if (q[1] == ' ' || q[1] == '\t') c.nv++;
            else if (q[1] == 'n' && q + 2 < e && (q[2] == ' ' || q[2] == '\t')) c.nvn++;
            else if (q[1] == 't' && q + 2 < e && (q[2] == ' ' || q[2] == '\t')) c.nvt++;
        }
        p = e + 1;
    }
}

void parse_chunk(ObjChunk& c, float* positions, float* normals, float* texcoords)
{
    size_t nv = c.v0, nvn = c.vn0, nvt = c.vt0;
    c.frags.emplace_back();
    bool s_known = false; unsigned smoothing = kUnknownSmoothing;
    auto fix = [](long idx, size_t n) -> int {
        if (idx > 0) return (int)idx - 1;
        if (idx < 0) return (int)n + (int)idx;
        return -1;
    };
    std::vector<ObjIndex> face;
    for (const char* p0 = c.begin; p0 < c.end;)
    {
        const char* e = line_end(p0, c.end);
        const char* p = skip_blank(p0, e);
        p0 = e + 1;
        if (p >= e || *p == '#' || *p == '\r') continue;
        const char c1 = p + 1 < e ? p[1] : '\0', c2 = p + 2 < e ? p[2] : '\0';
        if (p[0] == 'v' && (c1 == ' ' || c1 == '\t'))
        {
            float v[3] = { 0, 0, 0 };
            parse_floats(p + 2, e, v, 3);
            positions[nv * 3] = v[0]; positions[nv * 3 + 1] = v[1]; positions[nv * 3 + 2] = v[2]; nv++;
        }
        else if (p[0] == 'v' && c1 == 'n' && (c2 == ' ' || c2 == '\t'))
        {
            float v[3] = { 0, 0, 0 };
            parse_floats(p + 3, e, v, 3);
            normals[nvn * 3] = v[0]; normals[nvn * 3 + 1] = v[1]; normals[nvn * 3 + 2] = v[2]; nvn++;
        }
        else if (p[0] == 'v' && c1 == 't' && (c2 == ' ' || c2 == '\t'))
        {
            float v[2] = { 0, 0 };
            parse_floats(p + 3, e, v, 2);
            texcoords[nvt * 2] = v[0]; texcoords[nvt * 2 + 1] = v[1]; nvt++;
        }
        else if (p[0] == 'f' && (c1 == ' ' || c1 == '\t'))
        {
            face.clear();
            const char* q = p + 2;
            while (q < e)
            {
                while (q < e && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
                if (q >= e) break;
                ObjIndex ix = { -1, -1, -1 };
                char* end = nullptr;
                long v = std::strtol(q, &end, 10);
                if (end == q) break;
                ix.v = fix(v, nv);
                q = end;
                if (q < e && *q == '/')
                {
                    q++;
                    if (q < e && *q != '/')
                    {
                        long t = std::strtol(q, &end, 10);
                        if (end != q) { ix.t = fix(t, nvt); q = end; }
                    }
                    if (q < e && *q == '/')
                    {
                        q++;
                        long n = std::strtol(q, &end, 10);
                        if (end != q) { ix.n = fix(n, nvn); q = end; }
                    }
                }
                face.push_back(ix);
            }
            ObjFragment& f = c.frags.back();
            f.any_f = true;
            if (face.size() < 3) continue;                 // "degenerated face" (tiny_obj_loader.h:1449-1455)
            if (face.size() != 3 && !f.has_poly) { f.has_poly = true; f.sizes.assign(f.smoothing.size(), 3u); }     // the first polygon of this fragment
            if (f.has_poly) f.sizes.push_back((uint32_t)face.size());
            f.indices.insert(f.indices.end(), face.begin(), face.end());
            f.smoothing.push_back(s_known ? smoothing : kUnknownSmoothing);
        }
        else if ((p[0] == 'o' || p[0] == 'g') && (c1 == ' ' || c1 == '\t'))
        {
            // (`g` alone on its line is not a group statement for tinyobj - it wants a blank behind the letter - and neither is `o`)
            std::string name;
            if (p[0] == 'o') { name.assign(p + 2, e); while (!name.empty() && name.back() == '\r') name.pop_back(); }     // the rest of the line as it stands (:2887-2890)
            else
            {
                // the group's names joined by single blanks (:2841-2868)
                for (const char* q = p + 1; q < e;)
                {
                    while (q < e && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
                    const char* w = q;
                    while (q < e && *q != ' ' && *q != '\t' && *q != '\r') q++;
                    if (q > w) { if (!name.empty()) name += ' '; name.append(w, q); }
                }
            }
            c.frags.emplace_back();
            c.frags.back().new_shape = true;
            c.frags.back().kind = p[0];
            c.frags.back().name = name;
        }
        else if ((p[0] == 'l' || p[0] == 'p') && (c1 == ' ' || c1 == '\t'))
        {
            (p[0] == 'l' ? c.frags.back().any_l : c.frags.back().any_p) = true;       // lines / points: nothing to stage, but they keep a shape alive
        }
        else if (p[0] == 's' && (c1 == ' ' || c1 == '\t'))
        {
            const char* q = skip_blank(p + 2, e);
            if (q >= e || *q == '\r') continue;            // `s` without a value changes nothing (:2957-2963)
            if (e - q >= 3 && !std::strncmp(q, "off", 3)) smoothing = 0;
            else { int id = (int)std::strtol(q, nullptr, 10); smoothing = id < 0 ? 0u : (unsigned)id; }
            s_known = true; c.has_s = true; c.last_s = smoothing;
        }
    }
}

// tinyobj 2.0.0's triangulation of one face of four or more corners (tiny_obj_loader.h:1457-1856), in its float arithmetic
// and with its quirks - the reference's triangle list is this function's output, so a fan would stage other triangles.
void triangulate_face(const ObjIndex* face, size_t n, const float* v, size_t nv3, std::vector<ObjIndex>& out)
{
    auto valid = [&](int vi) { return 3 * (size_t)vi + 2 < nv3; };         // (a negative index wraps to a huge size_t, as there)
    if (n == 4)
    {
        // the quad is cut along its shorter diagonal
        if (!valid(face[0].v) || !valid(face[1].v) || !valid(face[2].v) || !valid(face[3].v)) return;
        const float* a = v + 3 * (size_t)face[0].v; const float* b = v + 3 * (size_t)face[1].v;
        const float* c = v + 3 * (size_t)face[2].v; const float* d = v + 3 * (size_t)face[3].v;
        const float e02x = c[0] - a[0], e02y = c[1] - a[1], e02z = c[2] - a[2];
        const float e13x = d[0] - b[0], e13y = d[1] - b[1], e13z = d[2] - b[2];
        const float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
        const float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
        if (sqr02 < sqr13) { out.insert(out.end(), { face[0], face[1], face[2], face[0], face[2], face[3] }); }
        else { out.insert(out.end(), { face[0], face[1], face[3], face[1], face[2], face[3] }); }
        return;
    }
    // the two axes of the plane to work in: from the first corner that is not degenerate
    size_t axes[2] = { 1, 2 };
    for (size_t k = 0; k < n; k++)
    {
        const int i0 = face[k % n].v, i1 = face[(k + 1) % n].v, i2 = face[(k + 2) % n].v;
        if (!valid(i0) || !valid(i1) || !valid(i2)) continue;
        const float* p0 = v + 3 * (size_t)i0; const float* p1 = v + 3 * (size_t)i1; const float* p2 = v + 3 * (size_t)i2;
        const float e0x = p1[0] - p0[0], e0y = p1[1] - p0[1], e0z = p1[2] - p0[2];
        const float e1x = p2[0] - p1[0], e1y = p2[1] - p1[1], e1z = p2[2] - p1[2];
        const float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
        const float eps = std::numeric_limits<float>::epsilon();
        if (cx > eps || cy > eps || cz > eps)
        {
            if (!(cx > cy && cx > cz))
            {
                axes[0] = 0;
                if (cz > cx && cz > cy) axes[1] = 1;
            }
            break;
        }
    }
    // ear clipping with a bounded number of fruitless rounds
    std::vector<ObjIndex> rest(face, face + n);
    size_t guess = 0, rounds_left = n, previous = n;
    while (rest.size() > 3 && rounds_left > 0)
    {
        const size_t m = rest.size();
        if (guess >= m) guess -= m;
        if (previous != m) { previous = m; rounds_left = m; }
        else rounds_left--;
        ObjIndex ind[3]; float vx[3], vy[3];
        for (size_t k = 0; k < 3; k++)
        {
            ind[k] = rest[(guess + k) % m];
            const size_t vi = (size_t)ind[k].v;
            if (vi * 3 + axes[0] >= nv3 || vi * 3 + axes[1] >= nv3) { vx[k] = 0.0f; vy[k] = 0.0f; }
            else { vx[k] = v[vi * 3 + axes[0]]; vy[k] = v[vi * 3 + axes[1]]; }
        }
        const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
        const float cross = e0x * e1y - e0y * e1x;
        const float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
        if (cross * area < 0.0f) { guess += 1; continue; }                 // an internal angle
        bool overlap = false;
        for (size_t other = 3; other < m; other++)
        {
            const size_t idx = (guess + other) % m;
            const size_t ovi = (size_t)rest[idx].v;
            if (ovi * 3 + axes[0] >= nv3 || ovi * 3 + axes[1] >= nv3) continue;
            const float tx = v[ovi * 3 + axes[0]], ty = v[ovi * 3 + axes[1]];
            // point in triangle by crossing number (pnpoly, :1415-1427)
            int inside = 0;
            for (int i = 0, j = 2; i < 3; j = i++)
                if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) inside = !inside;
            if (inside) { overlap = true; break; }
        }
        if (overlap) { guess += 1; continue; }
        out.insert(out.end(), { ind[0], ind[1], ind[2] });                // an ear
        rest.erase(rest.begin() + (ptrdiff_t)((guess + 1) % m));
    }
    if (rest.size() == 3) out.insert(out.end(), { rest[0], rest[1], rest[2] });
}

}  // namespace

bool load_obj(const std::string& file, ObjData& out)
{
    out = ObjData();
    std::string buf;
    {
        FILE* f = std::fopen(file.c_str(), "rb");
        if (!f) return false;
        std::fseek(f, 0, SEEK_END);
        const long size = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        if (size < 0) { std::fclose(f); return false; }
        buf.resize((size_t)size);
        const size_t got = size ? std::fread(&buf[0], 1, (size_t)size, f) : 0;
        std::fclose(f);
        buf.resize(got);
    }
    const char* base = buf.c_str();                        // NUL-terminated: strtof / strtol never run past the end
    const size_t size = buf.size();
    const int nchunks = (int)std::max<size_t>(1, std::min<size_t>(host_threads(), size / (256 * 1024)));
    std::vector<ObjChunk> chunks(nchunks);
    {
        const char* p = base;
        for (int k = 0; k < nchunks; k++)
        {
            chunks[k].begin = p;
            const char* e = k + 1 == nchunks ? base + size : line_end(base + (size_t)(k + 1) * size / nchunks, base + size);
            if (e < base + size) e++;                      // past the newline
            if (e < p) e = p;
            chunks[k].end = e; p = e;
        }
    }
    parallel_for(nchunks, [&](size_t k) { count_chunk(chunks[k]); });
    size_t nv = 0, nvn = 0, nvt = 0;
    for (ObjChunk& c : chunks) { c.v0 = nv; c.vn0 = nvn; c.vt0 = nvt; nv += c.nv; nvn += c.nvn; nvt += c.nvt; }
    out.positions.resize(nv * 3); out.normals.resize(nvn * 3); out.texcoords.resize(nvt * 2);
    parallel_for(nchunks, [&](size_t k) { parse_chunk(chunks[k], out.positions.data(), out.normals.data(), out.texcoords.data()); });
    // stitch: shapes begin at `o` / `g` statements; the smoothing group carries across chunks.  Which shapes survive follows
    // tinyobj to the letter, because a shape is an ELEMENT and element numbers are what materials are set by: a `g` statement
    // keeps the shape before it only if it has triangles (tiny_obj_loader.h:2826-2828), an `o` statement also if it has lines
    // or points (:2880-2883), and the end of the file keeps the last shape if it saw ANY `f` / `l` / `p` statement, even when
    // no triangle came of it - corners < 3, or a polygon the ear clipper gave up on (:3015-3022).
    ObjShape shape;
    unsigned smoothing = 0;
    bool had_f = false, had_l = false, had_p = false;
    for (ObjChunk& c : chunks)
    {
        for (ObjFragment& f : c.frags)
        {
            if (f.new_shape)
            {
                if (!shape.indices.empty() || (f.kind == 'o' && (had_l || had_p))) out.shapes.push_back(std::move(shape));
                shape = ObjShape();
                shape.name = f.name;
                had_f = had_l = had_p = false;
            }
            had_f |= f.any_f; had_l |= f.any_l; had_p |= f.any_p;
            for (unsigned& sm : f.smoothing) { if (sm != kUnknownSmoothing) break; sm = smoothing; }
            if (!f.has_poly)
            {
                shape.indices.insert(shape.indices.end(), f.indices.begin(), f.indices.end());
                shape.smoothing.insert(shape.smoothing.end(), f.smoothing.begin(), f.smoothing.end());
            }
            else
            {
                // polygons among the faces: triangulated now that every position is known
                size_t at = 0;
                for (size_t k = 0; k < f.sizes.size(); k++)
                {
                    const size_t before = shape.indices.size();
                    if (f.sizes[k] == 3) shape.indices.insert(shape.indices.end(), f.indices.begin() + (ptrdiff_t)at, f.indices.begin() + (ptrdiff_t)at + 3);
                    else triangulate_face(f.indices.data() + at, f.sizes[k], out.positions.data(), out.positions.size(), shape.indices);
                    shape.smoothing.insert(shape.smoothing.end(), (shape.indices.size() - before) / 3, f.smoothing[k]);
                    at += f.sizes[k];
                }
            }
        }
        if (c.has_s) smoothing = c.last_s;
    }
    if (!shape.indices.empty() || had_f || had_l || had_p) out.shapes.push_back(std::move(shape));
    return true;
}

}  // namespace ptkhost

using namespace ptkhost;

struct PathTracer::Impl {
    std::vector<StagedTriangle> triangles;
    std::vector<PathTracerLoader::Object> objects;
    std::vector<Image*> textures;

    glm::ivec2 resolution = glm::ivec2(0, 0);
    // The hand-off target.  The viewer calls SetOutImage from its GUI thread (InitializeFrame, main.cpp:3425-3446) while
    // PathTracerLoop may be inside RenderFrame() on another (main.cpp:3665-3678): the setters only STORE, under bind_mu, and
    // RenderFrames() takes one snapshot per call and does all the device-layer work on its own thread.
    std::mutex bind_mu;
    GLubyte* out_img = 0;
    GLubyte* bound_img = 0;             // what ptk_bind_out_image was last told (render thread only)
    unsigned out_gl = 0;                // SetOutGLBuffer: an OpenGL buffer object instead of a host buffer (registered by the setter itself)
    void* out_dev = 0; void* bound_dev = 0;   // SetOutDeviceImage: a device buffer instead of a host buffer
    bool bind_dirty = true;             // the library may have dropped the binding: say again what is bound, whatever the pointer
    std::mutex render_mu;               // held by RenderFrames() for its duration and by SetOutGLBuffer while it talks to the device layer
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
    // (a dot that lies BEFORE the file name - "./mesh", "dir.v2/mesh" - makes the count negative: as a size_t it takes the rest, :56)
    std::string objName = file.substr((size_t)nameStartIndex, (size_t)(nameEndIndex - nameStartIndex));
    PathTracerLoader::Object object(objName);

    const bool has_normals = !obj.normals.empty();
    const bool has_uvs = !obj.texcoords.empty();
    const int nv = (int)obj.positions.size() / 3, nn = (int)obj.normals.size() / 3, nt = (int)obj.texcoords.size() / 2;
    for (size_t i = 0; i < obj.shapes.size(); i++)
    {
        object.elements.push_back(PathTracerLoader::Element(obj.shapes[i].name));
        const ObjShape& sh = obj.shapes[i];
        const size_t ntri = std::min(sh.indices.size() / 3, sh.smoothing.size());
        const size_t first = m->triangles.size();
        m->triangles.resize(first + ntri);
        std::vector<uint8_t> okv(ntri, 1);
        const int objectId = (int)m->objects.size();
        StagedTriangle* dst = m->triangles.data() + first;
        // every triangle is staged independently (x-negation, model transform, v-flip, Triangle::Init): all cores
        parallel_for(ntri, [&](size_t tj) {
            const size_t j = tj * 3;
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
            if (!ok) { okv[tj] = 0; return; }
            triangle_init(t);
            t.smoothing = sh.smoothing[tj] != 0;               // :131-135
            t.objectId = objectId;
            t.elementId = (int)i;
            dst[tj] = t;
        }, 4096);
        // faces that name a vertex the file does not have are dropped, the rest keep their order
        size_t kept = 0;
        for (size_t tj = 0; tj < ntri; tj++)
            if (okv[tj]) { if (kept != tj) dst[kept] = dst[tj]; kept++; }
        m->triangles.resize(first + kept);
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

// :297-300: a pointer store, as in the reference - safe from the viewer's GUI thread while RenderFrame() runs on another.  The
// buffer stays the caller's in every respect: ordinary memory (`new GLubyte[w * h * 3]`, main.cpp:3435) is never page-locked or
// otherwise registered with the runtime (the viewer deletes texData BEFORE it hands over the next buffer, main.cpp:3433-3445, and
// at exit without telling the tracer, :3622); RenderFrame() copies the resolved frame into it.  Memory from ptk_host_alloc is
// written by the accumulate kernel directly (no copy); free such a buffer only after the RenderFrame() in flight has returned.
void PathTracer::SetOutImage(GLubyte* out)
{
    std::lock_guard<std::mutex> g(m->bind_mu);
    m->bind_dirty = true;               // (even for the same pointer: a block freed and allocated again at the same address is another buffer)
    m->out_img = out;
    if (out) { m->out_gl = 0; m->out_dev = 0; }
}

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
    std::lock_guard<std::mutex> render_guard(m->render_mu);
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
        std::lock_guard<std::mutex> g(m->bind_mu);
        m->bind_dirty = true;                                           // (a new resolution unbinds the hand-off buffer: say again what is bound)
    }
    GLubyte* out_img; unsigned out_gl; void* out_dev; bool rebind;
    {
        // one snapshot of the hand-off target per call (the setters may run on another thread)
        std::lock_guard<std::mutex> g(m->bind_mu);
        out_img = m->out_img; out_gl = m->out_gl; out_dev = m->out_dev;
        rebind = m->bind_dirty || out_img != m->bound_img || out_dev != m->bound_dev;
        m->bind_dirty = false;
    }
    if (rebind && !out_gl)
    {
        // SetOutImage (:297-300): a buffer the GPU can write (ptk_host_alloc) receives the 8-bit resolve straight from the
        // accumulate kernel from now on; ordinary memory is not bound, the frame is copied into it below
        if (out_dev) m->note(ptk_bind_out_device(m->ctx, out_dev));
        else m->note(ptk_bind_out_image(m->ctx, out_img));
        m->bound_img = out_img; m->bound_dev = out_dev;
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
    if (rc == PTK_OK && out_img) m->note(ptk_resolve_rgb8(m->ctx, out_img));   // :802-812 into the caller's buffer
    else if (rc == PTK_OK && (out_gl || out_dev)) m->note(ptk_synchronize(m->ctx));   // RenderFrame() returns with the frame in the OpenGL / device buffer
}

void PathTracer::Exit() { if (m->ctx) ptk_request_exit(m->ctx); }      // :819-822

// ---- extensions -----------------------------------------------------------------------------------------
void PathTracer::SetSeed(uint64_t seed) { m->seed = seed; }
// EXPERIMENTAL (built, its refusal tested, never executed: the GPU boxes are headless).  Registration talks to the OpenGL driver
// through the CALLING thread's current context, so it happens here, in the setter the viewer's GUI thread calls - not inside
// RenderFrame(), which the viewer runs on a thread without a context (PathTracerLoop, main.cpp:3665-3678).  The setter waits for a
// RenderFrame() in flight.  RenderFrame() maps the buffer on its stream before the first pass and unmaps it behind the
// accumulate kernel; the GUI thread may read the buffer (glTexSubImage2D from GL_PIXEL_UNPACK_BUFFER) only BETWEEN frames, i.e.
// it sequences itself with the render thread exactly as it must for texData.
void PathTracer::SetOutGLBuffer(unsigned int gl_buffer)
{
    std::lock_guard<std::mutex> render_guard(m->render_mu);
    {
        std::lock_guard<std::mutex> g(m->bind_mu);
        m->out_gl = gl_buffer;
        if (gl_buffer) { m->out_img = 0; m->out_dev = 0; }
        m->bind_dirty = true;
    }
    if (!m->ensure_ctx()) return;
    m->note(ptk_bind_gl_buffer(m->ctx, gl_buffer));                   // (0: lets a registered buffer go, on the thread that owns the context)
    m->bound_img = 0; m->bound_dev = 0;
}
void PathTracer::SetOutDeviceImage(void* device_rgb8)
{
    std::lock_guard<std::mutex> g(m->bind_mu);
    m->out_dev = device_rgb8;
    m->bind_dirty = true;
    if (device_rgb8) { m->out_img = 0; m->out_gl = 0; }
}
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
    if (m->scene_uploaded) m->flat.lights = m->built_lights;         // mLights is BuildBVH's, whatever was edited since (pathtracer.cpp:267-273)
    m->flat_desc = m->flat.desc();
    return &m->flat_desc;
}
