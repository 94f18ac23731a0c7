// Host side of the ptk C-ABI (include/ptk.h): context, scene staging into the device record layouts
// of ptk_device.h, BVH build, kernel launches, hand-off copies, RCCL exchange step.
//
// Reference behaviour restated here (PathTracing/src/pathtracer.cpp): BuildBVH + light list :260-274,
// setters :297-360, frame set-up arithmetic of RenderFrame :755-766, reset :745-751.
#include <hip/hip_runtime.h>
#include <hip/hip_gl_interop.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "bvh_build.h"
#include "bvh_device.h"
#include "ptk_device.h"

using namespace ptk;

struct ptk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string error;
    std::mutex err_mu;

    // scene (device)
    float4* d_flat_tris = nullptr;      // <= 16 triangles: records in ascending triangle index for the FLAT kernel
    float4 *d_nodes = nullptr, *d_tris = nullptr, *d_shade = nullptr, *d_mats = nullptr, *d_lights = nullptr;
    int4* d_texinfo = nullptr;
    uint32_t* d_texels = nullptr;
    int num_nodes = 0, num_tris = 0, num_lights = 0, bvh_depth = 0, bvh_stack = 0, num_leaf_tris = 0;
    float scene_bound = 0.0f;           // 3.1 x (1.01 x the largest |vertex coordinate| + 1e-3): RenderParams::scene_bound
    float scene_lo[3] = { 0, 0, 0 }, scene_hi[3] = { 0, 0, 0 };     // the vertices' bounding box
    int opt_lens_cull = 1;              // uncached cameras: pixels whose whole bundle of lens rays misses that box are never traced
    unsigned long long view_generation = 0;      // bumped whenever camera, frame or scene change what an uncached pixel can see
    bool have_scene = false;
    double upload_ms[4] = { 0, 0, 0, 0 };       // last ptk_upload_scene: BVH build, record packing, device copies, total
    // host copies kept for ptk_update_materials: what was uploaded, the texture index map, the light records
    std::vector<ptk_material> h_materials;
    std::vector<int32_t> h_texmap, h_light_material;
    std::vector<float> h_lights;

    // camera (host copies, already normalised / clamped like the reference setters)
    float cam_pos[3] = { 0, 0, 0 }, cam_dir[3] = { 0, 0, 1 }, cam_up[3] = { 0, 1, 0 };
    float focal = 0.1f, fovy = 90.0f, focal_dist = 5.0f, aperture = 0.0f;   // pathtracer.cpp:17-22
    float cam_right[3] = { 1, 0, 0 };
    bool primary_dirty = true;

    // frame
    int width = 0, height = 0, max_depth = 3;                                 // pathtracer.cpp:15
    float4* d_primary = nullptr;
    float4* d_primary_hit = nullptr;  // primary-visibility cache (pinhole, no opacity textures)
    float4* d_primary_rd = nullptr;   // ... and the camera ray's direction per pixel
    uint2* d_pixel_rng = nullptr;     // per pixel: (pixel key, PCG increment) for pixel_rng_seed
    uint64_t pixel_rng_seed = 0; bool pixel_rng_valid = false;
    bool primary_hit_dirty = true, scene_has_opacity = false;
    int opt_primary_cache = 1;
    int opt_flat_shade_w = 8, opt_flat_gen_w = 64;
    int opt_flat = 1;                    // scenes of <= PTK_FLAT_MAX triangles: no BVH walk (trace_kernel<.., FLAT>)
    float* d_accum = nullptr;        // owned accumulator
    float* d_accum_bound = nullptr;  // caller-owned accumulator (ptk_bind_accum) or null
    uint8_t* d_rgb8 = nullptr;
    // hand-off buffer bound by ptk_bind_out_image: the caller's pointer, its device-visible alias, whether this context
    // page-locked it, and whether the next accumulate kernel must write every pixel (after binding, reset, a camera or
    // scene change: the set of always-black pixels may have changed)
    uint8_t* out_host = nullptr; uint8_t* out_host_dev = nullptr;
    bool out_registered = false;
    std::atomic<bool> out_full_next{ true };
    // ... or a DEVICE buffer bound by ptk_bind_out_device, or an OpenGL buffer object registered by ptk_bind_gl_buffer (mapped
    // for the length of each render: run_passes)
    uint8_t* out_device = nullptr;
    hipGraphicsResource_t gl_res = nullptr;
    unsigned gl_buffer = 0;
    int rank = 0, world = 1;

    std::atomic<int> samples{ 0 };
    std::atomic<uint32_t> render_gen{ 1 };       // generation of the newest render: Exit() names it and thereby every render still in flight (kernels stand down
                                                 // when the named generation >= their own); renders issued afterwards are not affected
    uint32_t* d_exit = nullptr;
    unsigned long long* d_stats = nullptr;
    unsigned* d_queues = nullptr;                // item queues of trace_kernel's persistent waves
    // live-quadrant list (which 8x8 quadrants of the owned tiles have pixels to trace) and what it was built for
    unsigned long long* d_live_mask = nullptr;
    unsigned* d_live_list = nullptr;             // [capacity] entries + 1 word: the count
    int live_capacity = 0;
    unsigned long long hit_generation = 0;       // bumped whenever the primary-hit cache is recomputed
    struct LiveKey { int width, height, rank, world, cached; unsigned long long generation; } live_key = { 0, 0, -1, 0, -1, 0 };
    int resident_waves = 4096;                   // one-wave workgroups the device holds at once (CUs x 16)

    // sample buffer between trace_kernel and accumulate_kernel (grown on demand, never shrunk)
    // Two sample buffers / queue blocks / trace streams: the trace kernel of pass k+1 runs on the other stream and
    // overlaps the tail of pass k's (a launch ends with the few waves that hold its longest paths - Russian roulette
    // lets one path in tens of millions live for 60+ bounces, ~0.4 ms with the rest of the chip idle) and pass k's
    // accumulate kernel.  The accumulate kernels stay on the context's stream, in order, each behind its trace kernel,
    // so everything the caller orders after ptk_render on that stream still sees the finished batch.
    float4* d_samples = nullptr;                 // (non-overlapped path)
    size_t samples_bytes = 0;
    float4* d_samples2[2] = { nullptr, nullptr };
    size_t samples_bytes2[2] = { 0, 0 };
    unsigned* d_queues2[2] = { nullptr, nullptr };
    hipStream_t trace_stream[2] = { nullptr, nullptr };
    hipEvent_t ev_trace_done[2] = { nullptr, nullptr }, ev_acc_done[2] = { nullptr, nullptr }, ev_inputs = nullptr;
    bool acc_pending[2] = { false, false }, inputs_recorded = false;
    bool inputs_dirty = true;                    // scene / camera tables were (re)written on the context's stream since ev_inputs
    unsigned pass_counter = 0;
    int opt_overlap = 1;
    int opt_register_out = 0;                    // 1: ptk_bind_out_image page-locks a pageable caller buffer in place (round 3's default; the caller must unbind before freeing)
    int opt_contract = 0;                        // 0: bit-exact kernels; 1: -ffp-contract=fast build; 2: ... with 1-ulp hardware rcp / sqrt
    int opt_chunk = 0;                           // samples per work item; 0 = automatic (8, or 4 for small shares)
    int num_cus = 256;
    int opt_generations = 0;                     // 0 automatic: 1 on a single GPU, 2 when the frame is split over ranks
    int opt_persistent = -1;                     // -1 automatic (by launch size), 0 one item per wave, 1 persistent waves
    int opt_max_batch = 1;                       // slots a persistent wave pops from its queue at once; > 1 measured slower everywhere
    int opt_device_build = 1;                    // scenes of >= 4096 triangles: BVH built and records packed on the GPU (bvh_device.hip)
    bool built_on_device = false;
    int opt_tri_thr = 6;                         // triangle arm of the walk runs when queued lanes >= tri_thr/8 x walking lanes
    int opt_shade_thr = 0, opt_gen_thr = 16;     // scheduling lambdas in eighths, see trace_kernel; 0 = by tree depth
    size_t opt_pass_bytes = (size_t)16 << 30;    // sample-buffer budget per pass (two such buffers at most, of the 288 GB: C4 traces its 256 spp in one launch, C5 its 1024 in two: +0.4 / +1.2 % over 4 GiB)

    // multi-GPU exchange step (ptk_gather_accum): packed gather of every rank's owned tiles to the root
    ncclComm_t comm = nullptr;                   // the context's own communicator (ptk_comm_init), or null
    int comm_rank = 0, comm_world = 1;
    hipStream_t xstream = nullptr;               // high priority: its kernels and the collective take wave slots ahead of queued trace waves
    hipEvent_t ev_rendered = nullptr, ev_packed = nullptr, ev_gathered = nullptr;
    float* d_packed = nullptr; size_t packed_floats = 0;      // non-root: this rank's packed tiles; root: every rank's, back to back
    float* d_gathered = nullptr; size_t gathered_floats = 0;  // root: the combined image (W*H*3, rows bottom-up)
    int gathered_w = 0, gathered_h = 0;          // the frame the last ptk_gather_accum combined (ptk_read_gathered refuses any other)
    bool gather_pending = false;
    // every wait of the exchange step is bounded (ptk_set_option "comm_timeout_s"): a rank that never arrives must end the run with
    // an error that names it, not hang the node (the first real 8-GPU run is the driver's, not ours)
    double opt_comm_timeout_s = 120.0;
    unsigned long long gather_step = 0;          // exchanges queued so far (named in the timeout message)
    size_t gather_bytes = 0;                     // bytes the last exchange moves on this rank (root: what it receives)
    int gather_root = 0;

    // log of every trace launch's duration (ptk_kernel_log): event pairs on the launch's own stream
    std::vector<hipEvent_t> klog_ev;             // 2 x capacity
    int klog_n = 0;
    static constexpr int kMaxTimedPasses = 64;
    hipEvent_t ev[kMaxTimedPasses][3] = {};      // per pass: before trace, after trace, after accumulate
    int last_passes = 0;
    int last_launches = 0;
    bool timed = false;
};

namespace {

int fail(ptk_ctx* c, int code, const std::string& msg)
{
    if (c) { std::lock_guard<std::mutex> g(c->err_mu); c->error = msg; }
    return code;
}
#define HIPCHK(c, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(c, PTK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

template <class T>
void dfree(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

float* accum_ptr(ptk_ctx* c) { return c->d_accum_bound ? c->d_accum_bound : c->d_accum; }

inline float as_float(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

void normalize3(const float* in, float* out)
{
    // glm::normalize = x * (1 / sqrt(dot(x, x)))   (pathtracer.cpp:336-337)
    float sqr = in[0] * in[0] + in[1] * in[1] + in[2] * in[2];
    float inv = 1.0f / std::sqrt(sqr);
    out[0] = in[0] * inv; out[1] = in[1] * inv; out[2] = in[2] * inv;
}

// device material record (ptk_device.h) from the boundary's ptk_material
void pack_material(const ptk_material& m, const std::vector<int32_t>& texmap, float* q)
{
    q[0] = m.diffuse[0]; q[1] = m.diffuse[1]; q[2] = m.diffuse[2]; q[3] = as_float(m.type != 0 ? 1 : 0);
    q[4] = m.specular[0]; q[5] = m.specular[1]; q[6] = m.specular[2]; q[7] = m.emissive_intensity;
    q[8] = m.emissive[0]; q[9] = m.emissive[1]; q[10] = m.emissive[2]; q[11] = m.roughness;
    // Russian-roulette probability, pathtracer.cpp:589: glm::min(0.95f, glm::max(glm::max(d.x, d.y), d.z))
    float mx = m.diffuse[0] < m.diffuse[1] ? m.diffuse[1] : m.diffuse[0];
    mx = mx < m.diffuse[2] ? m.diffuse[2] : mx;
    float prob = 0.95f < mx ? 0.95f : mx;
    q[12] = m.reflectiveness; q[13] = m.translucency; q[14] = m.ior; q[15] = prob;
    int any = 0;
    for (int k = 0; k < 6; k++)
    {
        int32_t t = m.tex[k] >= 0 ? texmap[m.tex[k]] : -1;
        q[16 + k] = as_float(t);
        if (t >= 0) any = 1;
    }
    q[22] = as_float(any); q[23] = 0.0f;
}

// image-plane set-up of RenderFrame, pathtracer.cpp:755-766 (host, once per camera/resolution change)
void frame_setup(ptk_ctx* c, PrimaryParams& pp)
{
    const float* pos = c->cam_pos; const float* dir = c->cam_dir; const float* up = c->cam_up;
    float center[3] = { pos[0] + dir[0] * c->focal, pos[1] + dir[1] * c->focal, pos[2] + dir[2] * c->focal };
    float img_h = (float)((double)(2.0f * c->focal) * std::tan((double)(c->fovy / 2.0f) * 3.14159265358979323846 / (double)180.0f));
    float aspect = (float)c->width / (float)c->height;
    float img_w = img_h * aspect;
    pp.delta_x = img_w / (float)c->width;
    pp.delta_y = img_h / (float)c->height;
    // camRight = normalize(cross(up, dir))
    float cr[3] = { up[1] * dir[2] - dir[1] * up[2], up[2] * dir[0] - dir[2] * up[0], up[0] * dir[1] - dir[0] * up[1] };
    normalize3(cr, c->cam_right);
    float hw = img_w * 0.5f, hh = img_h * 0.5f;
    for (int a = 0; a < 3; a++)
    {
        float tl = center[a] - c->cam_right[a] * hw;
        tl = tl + up[a] * hh;
        pp.top_left[a] = tl;
        pp.cam_pos[a] = pos[a]; pp.cam_right[a] = c->cam_right[a]; pp.cam_up[a] = up[a];
    }
    pp.primary = c->d_primary; pp.width = c->width; pp.height = c->height;
}

// on-image pixels of the tiles this rank owns (same tile -> rank map as the kernels)
unsigned long long owned_pixels(const ptk_ctx* c)
{
    const int tiles_x = (c->width + PTK_TILE - 1) / PTK_TILE, tiles_y = (c->height + PTK_TILE - 1) / PTK_TILE;
    unsigned long long n = 0;
    for (int tile = c->rank; tile < tiles_x * tiles_y; tile += c->world)
    {
        const int ty = tile / tiles_x, tx = (tile % tiles_x + tiles_x - (3 * ty) % tiles_x) % tiles_x;
        const int w = std::min(PTK_TILE, c->width - tx * PTK_TILE), h = std::min(PTK_TILE, c->height - ty * PTK_TILE);
        n += (unsigned long long)w * h;
    }
    return n;
}

int ensure_primary(ptk_ctx* c)
{
    if (!c->primary_dirty) return PTK_OK;
    if (!c->d_primary) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    PrimaryParams pp;
    frame_setup(c, pp);
    launch_primary(pp, c->stream);
    HIPCHK(c, hipGetLastError());
    c->primary_dirty = false;
    c->view_generation++;
    c->primary_hit_dirty = true;
    c->inputs_dirty = true;
    return PTK_OK;
}

bool primary_cacheable(const ptk_ctx* c)
{
    return c->opt_primary_cache && c->aperture == 0.0f && !c->scene_has_opacity && c->have_scene && c->d_primary_hit;
}

void fill_params(ptk_ctx* c, RenderParams& p, uint32_t first, uint32_t spp, uint64_t seed)
{
    std::memset(&p, 0, sizeof(p));
    p.nodes = c->d_nodes; p.tris = c->d_tris; p.flat_tris = c->d_flat_tris; p.shade = c->d_shade; p.mats = c->d_mats; p.lights = c->d_lights;
    p.texinfo = c->d_texinfo; p.texels = c->d_texels; p.primary = c->d_primary;
    p.primary_hit = nullptr;
    p.samples = c->d_samples;
    // shallow trees give short, uniform walks: waiting for stragglers is cheap and re-synchronises the
    // wave (lambda 25); deep trees have heavy-tailed walks: shade small batches early (lambda 5)
    // lambda of the ski-rental rule = cost(shade block) / cost(one walk iteration), in eighths: a walk iteration of a tree far
    // larger than the L2 waits for memory and costs more, so the shade block is run for smaller batches there (measured
    // optimum: ~72 for 10-20 k nodes, ~46 for 370 k nodes; flat within 1 % around either)
    p.shade_thr = c->opt_shade_thr > 0 ? c->opt_shade_thr : (c->bvh_depth <= 4 ? 200 : (c->num_nodes >= 131072 ? 46 : 68));
    p.gen_thr = c->opt_gen_thr;
    p.tri_thr = c->opt_tri_thr;
    p.max_batch = c->opt_max_batch;
    p.persistent = c->opt_persistent;
    p.generations = c->opt_generations > 0 ? c->opt_generations : (c->world > 1 ? 2 : 1);
    p.rgb8_host = nullptr; p.rgb8_host_full = 1;
    p.accum = accum_ptr(c); p.rgb8 = c->d_rgb8; p.exit_flag = c->d_exit; p.exit_gen = c->render_gen.load(); p.stats = c->d_stats;
    p.num_nodes = c->num_nodes; p.num_lights = c->num_lights; p.scene_bound = c->scene_bound;
    for (int a = 0; a < 3; a++) { p.scene_lo[a] = c->scene_lo[a]; p.scene_hi[a] = c->scene_hi[a]; }
    p.lens_cull = c->opt_lens_cull;
    p.flat_count = (c->opt_flat && c->num_tris <= 16 && c->d_flat_tris) ? c->num_tris : 0;
    p.flat_shade_w = c->opt_flat_shade_w; p.flat_gen_w = c->opt_flat_gen_w;
    p.width = c->width; p.height = c->height; p.max_depth = c->max_depth;
    p.tiles_x = (c->width + PTK_TILE - 1) / PTK_TILE;
    p.num_tiles = p.tiles_x * ((c->height + PTK_TILE - 1) / PTK_TILE);
    p.rank = c->rank; p.world = c->world;
    p.first_sample = first; p.spp = spp;
    p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
    for (int a = 0; a < 3; a++) { p.cam_pos[a] = c->cam_pos[a]; p.cam_right[a] = c->cam_right[a]; p.cam_up[a] = c->cam_up[a]; }
    p.focal_dist = c->focal_dist; p.aperture = c->aperture;
    p.resolve_samples = (float)(first + spp);
}

// keep_gl: a new resolution lets host and device buffers go (the caller reallocates them) but keeps an OpenGL buffer object
// registered - unregistering talks to the OpenGL driver and belongs to the thread that owns the context (ptk_bind_gl_buffer);
// the buffer's size is checked each time it is mapped
void unbind_out_image(ptk_ctx* c, bool keep_gl = false)
{
    if (c->out_registered && c->out_host) { (void)hipHostUnregister(c->out_host); (void)hipGetLastError(); }
    c->out_host = nullptr; c->out_host_dev = nullptr; c->out_registered = false; c->out_full_next = true;
    c->out_device = nullptr;
    if (keep_gl) return;
    if (c->gl_res) { (void)hipGraphicsUnregisterResource(c->gl_res); (void)hipGetLastError(); }
    c->gl_res = nullptr; c->gl_buffer = 0;
}

// An OpenGL buffer object is HIP's for the length of a render only: mapped before the first pass, unmapped (on the render
// stream, so behind the accumulate kernel that writes it) on every way out of run_passes.
struct GlMapping
{
    ptk_ctx* c; bool mapped = false;
    ~GlMapping() { if (mapped) { (void)hipGraphicsUnmapResources(1, &c->gl_res, c->stream); (void)hipGetLastError(); } }
};

int owned_tiles(const RenderParams& p)
{
    // tiles t with t % world == rank
    if (p.num_tiles <= p.rank) return 0;
    return (p.num_tiles - p.rank + p.world - 1) / p.world;
}

// One pass = trace_kernel over (owned tiles x 4 quadrants x chunks) + accumulate_kernel.  The pass
// size is bounded by the sample-buffer budget; chunk boundaries never change results (the RNG is
// keyed on the absolute sample index and the accumulate kernel adds in sample order).
int run_passes(ptk_ctx* c, uint32_t first, uint32_t spp, uint64_t seed, bool stats, float* accum, uint8_t* rgb8,
               const uint32_t* exit_flag, bool timed)
{
    RenderParams p;
    fill_params(c, p, first, spp, seed);
    p.accum = accum; p.rgb8 = rgb8; p.exit_flag = exit_flag;
    if (primary_cacheable(c))
    {
        if (c->primary_hit_dirty)
        {
            launch_primary_hits(p, c->d_primary_hit, c->d_primary_rd, c->stream);
            HIPCHK(c, hipGetLastError());
            c->primary_hit_dirty = false;
            c->hit_generation++;
            c->out_full_next = true;
            c->inputs_dirty = true;
        }
        p.primary_hit = c->d_primary_hit; p.primary_rd = c->d_primary_rd;
    }
    if (!c->pixel_rng_valid || c->pixel_rng_seed != seed)
    {
        launch_pixel_rng((uint32_t)seed, (uint32_t)(seed >> 32), c->width * c->height, c->d_pixel_rng, c->stream);
        HIPCHK(c, hipGetLastError());
        c->pixel_rng_seed = seed; c->pixel_rng_valid = true;
        c->inputs_dirty = true;
    }
    p.pixel_rng = c->d_pixel_rng;
    GlMapping glmap{ c };
    uint8_t* handoff = nullptr;
    if (rgb8 == c->d_rgb8)
    {
        handoff = c->out_host_dev ? c->out_host_dev : c->out_device;
        if (!handoff && c->gl_res)
        {
            HIPCHK(c, hipGraphicsMapResources(1, &c->gl_res, c->stream));
            glmap.mapped = true;
            void* ptr = nullptr; size_t bytes = 0;
            HIPCHK(c, hipGraphicsResourceGetMappedPointer(&ptr, &bytes, c->gl_res));
            if (!ptr || bytes < (size_t)c->width * c->height * 3)
                return fail(c, PTK_ERR_BAD_ARG, "the bound OpenGL buffer holds fewer than width * height * 3 bytes");
            handoff = (uint8_t*)ptr;
        }
    }
    if (handoff)
    {
        // (after the primary-hit cache has been brought up to date: a recomputed cache changes which pixels are always black)
        p.rgb8_host = handoff; p.rgb8_host_full = c->out_full_next.exchange(false) ? 1 : 0;
    }
    const int tiles = owned_tiles(p);
    c->last_passes = 0; c->last_launches = 0;
    if (tiles == 0) return PTK_OK;
    {
        // the list of quadrants with live pixels: rebuilt when the frame, the tile ownership or the cached camera hits change
        const int subtiles = tiles * 4;
        if (subtiles > c->live_capacity)
        {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            dfree(c->d_live_mask); dfree(c->d_live_list); c->live_capacity = 0;
            HIPCHK(c, hipMalloc(&c->d_live_mask, (size_t)subtiles * sizeof(unsigned long long)));
            HIPCHK(c, hipMalloc(&c->d_live_list, ((size_t)subtiles + 1) * sizeof(unsigned)));
            c->live_capacity = subtiles;
            c->live_key.rank = -1;
        }
        // (uncached cameras: the mask holds the pixels whose lens rays can reach the scene at all - it follows camera, frame and scene)
        const ptk_ctx::LiveKey key = { c->width, c->height, c->rank, c->world, p.primary_hit ? 1 : (p.lens_cull ? 2 : 0),
                                       p.primary_hit ? c->hit_generation : (p.lens_cull ? c->view_generation : 0ull) };
        const ptk_ctx::LiveKey& o = c->live_key;
        if (key.width != o.width || key.height != o.height || key.rank != o.rank || key.world != o.world || key.cached != o.cached ||
            key.generation != o.generation)
        {
            launch_live_list(p, subtiles, c->d_live_mask, c->d_live_list, c->d_live_list + c->live_capacity, c->stream);
            HIPCHK(c, hipGetLastError());
            c->live_key = key;
            c->inputs_dirty = true;
        }
        p.live_mask = c->d_live_mask; p.live_list = c->d_live_list; p.live_count = c->d_live_list + c->live_capacity;
    }
    const size_t per_sample = (size_t)tiles * 4 * 64 * sizeof(float4);
    // Samples per work item.  The waves are persistent and lanes take units from item after item, so short items
    // cost no SIMD utilisation any more; what they buy is balance at the end of the launch (the last items are
    // the tail) and concurrency on the same pixels.  8 is the measured optimum for whole frames (C2, C4); a rank
    // that owns 1/4 or 1/8 of the tiles does better with 4.  Each item costs one queue pop (~2 us of latency).
    int chunk_opt = c->opt_chunk;
    if (chunk_opt <= 0)
        chunk_opt = (double)spp * (double)tiles * 4.0 / 8.0 >= 49152.0 ? 8 : 4;
    // (a bound hand-off buffer means the caller waits for every frame: nothing to overlap, and one stream is two event
    // hops per frame less)
    const bool overlap = c->opt_overlap != 0 && c->trace_stream[0] != nullptr && !handoff;
    // The pass size: what the sample-buffer budget allows - and what the device can actually give.  When an allocation of that
    // size fails (memory shared with the caller's framework, a huge scene, a 144-megapixel frame) the pass is halved and tried
    // again instead of failing the render: more passes, the same image (chunk boundaries never change a bit).
    size_t budget = c->opt_pass_bytes;
    {
        const size_t have = overlap ? std::min(c->samples_bytes2[0], c->samples_bytes2[1]) : c->samples_bytes;
        const size_t want = std::min(budget, per_sample * ((size_t)spp + (size_t)chunk_opt));      // (what this render asks for at most)
        size_t free_b = 0, total_b = 0;
        if (want > have)             // (only a render that has to allocate asks the driver: the interactive loop never does)
        {
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            {
                // the buffers still to be allocated may take at most half of what is free now (there are two of them when overlapping)
                const size_t share = free_b / (overlap ? 4 : 2);
                if (budget > std::max(share, have)) budget = std::max(share, have);
            }
            else (void)hipGetLastError();
        }
    }
    uint32_t max_pass = (uint32_t)std::max<size_t>(1, budget / per_sample);
    if (max_pass > (uint32_t)chunk_opt) max_pass -= max_pass % (uint32_t)chunk_opt;
    // grows *buf to `need` bytes; false (nothing allocated, *bytes = 0) when the device refuses
    auto grow = [&](float4*& buf, size_t& bytes, size_t need) -> bool {
        dfree(buf); bytes = 0;
        if (hipMalloc(&buf, need) != hipSuccess) { (void)hipGetLastError(); buf = nullptr; return false; }
        bytes = need;
        return true;
    };
    uint32_t done = 0;
    while (done < spp)
    {
        uint32_t n = std::min(spp - done, max_pass);
        int chunk = (int)std::min<uint32_t>(n, (uint32_t)chunk_opt);
        int num_chunks = (int)((n + chunk - 1) / chunk);
        size_t need = per_sample * (size_t)chunk * num_chunks;
        const int b = (int)(c->pass_counter & 1u);
        hipStream_t tstream = overlap ? c->trace_stream[b] : c->stream;
        for (;;)
        {
            bool ok = true;
            if (overlap)
            {
                // BOTH buffers are brought to size together: the second one would otherwise be allocated - a blocking call of tens of
                // milliseconds for gigabytes - in the middle of the second render (BENCH_r03: C3's first timed steps)
                if (need > c->samples_bytes2[0] || need > c->samples_bytes2[1])
                {
                    HIPCHK(c, hipStreamSynchronize(c->trace_stream[0]));
                    HIPCHK(c, hipStreamSynchronize(c->trace_stream[1]));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    for (int k = 0; k < 2 && ok; k++)
                        if (need > c->samples_bytes2[k]) ok = grow(c->d_samples2[k], c->samples_bytes2[k], need);
                }
            }
            else if (need > c->samples_bytes)
            {
                HIPCHK(c, hipStreamSynchronize(c->stream));
                ok = grow(c->d_samples, c->samples_bytes, need);
            }
            if (ok) break;
            if (n <= (uint32_t)chunk && chunk <= 1) return fail(c, PTK_ERR_HIP, "hipMalloc: no memory for the sample buffer of even one sample per pixel");
            // halve the pass (whole chunks while there are several, then the chunk itself) and try again
            if (n > (uint32_t)chunk) { n = std::max<uint32_t>((uint32_t)chunk, (n / 2) / (uint32_t)chunk * (uint32_t)chunk); }
            else { chunk = std::max(1, chunk / 2); n = (uint32_t)chunk; }
            max_pass = n;
            num_chunks = (int)((n + chunk - 1) / chunk);
            need = per_sample * (size_t)chunk * num_chunks;
        }
        if (overlap)
        {
            p.samples = c->d_samples2[b];
            if (c->inputs_dirty || !c->inputs_recorded)
            {
                HIPCHK(c, hipEventRecord(c->ev_inputs, c->stream));
                c->inputs_dirty = false; c->inputs_recorded = true;
            }
            HIPCHK(c, hipStreamWaitEvent(tstream, c->ev_inputs, 0));
            if (c->acc_pending[b]) HIPCHK(c, hipStreamWaitEvent(tstream, c->ev_acc_done[b], 0));   // buffer b was last read by pass k-2's accumulate
        }
        else p.samples = c->d_samples;
        p.first_sample = first + done; p.spp = n;
        p.chunk = chunk; p.num_chunks = num_chunks;
        p.resolve_samples = (float)(first + done + n);
        const long long items = (long long)tiles * 4 * num_chunks;
        if (items > (1ll << 30)) return fail(c, PTK_ERR_LIMIT, "too many work items in one pass");
        p.num_items = (int)items;
        const int pi = c->last_passes < ptk_ctx::kMaxTimedPasses ? c->last_passes : -1;
        if (timed && pi >= 0) HIPCHK(c, hipEventRecord(c->ev[pi][0], tstream));
        const int kl = (!stats && (size_t)(2 * c->klog_n + 1) < c->klog_ev.size()) ? c->klog_n : -1;
        if (kl >= 0) HIPCHK(c, hipEventRecord(c->klog_ev[2 * kl], tstream));
        p.queues = overlap ? c->d_queues2[b] : c->d_queues;
        if (c->opt_contract == 1 && !stats) fma::launch_trace(p, tiles * 4, c->resident_waves, tstream, false);
        else if (c->opt_contract == 2 && !stats) fast::launch_trace(p, tiles * 4, c->resident_waves, tstream, false);
        else launch_trace(p, tiles * 4, c->resident_waves, tstream, stats);
        HIPCHK(c, hipGetLastError());
        if (timed && pi >= 0) HIPCHK(c, hipEventRecord(c->ev[pi][1], tstream));
        if (kl >= 0) { HIPCHK(c, hipEventRecord(c->klog_ev[2 * kl + 1], tstream)); c->klog_n = kl + 1; }
        if (overlap)
        {
            HIPCHK(c, hipEventRecord(c->ev_trace_done[b], tstream));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_trace_done[b], 0));
        }
        launch_accumulate(p, tiles, c->stream);
        HIPCHK(c, hipGetLastError());
        if (timed && pi >= 0) HIPCHK(c, hipEventRecord(c->ev[pi][2], c->stream));
        if (overlap)
        {
            HIPCHK(c, hipEventRecord(c->ev_acc_done[b], c->stream));
            c->acc_pending[b] = true;
            c->pass_counter++;
        }
        c->last_passes++; c->last_launches += 3;            // queue_init_kernel (queue + parameter block), trace, accumulate
        done += n;
    }
    return PTK_OK;
}

}  // namespace

extern "C" {

int ptk_create(ptk_ctx** out, int device_ordinal)
{
    if (!out) return PTK_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return PTK_ERR_HIP;
    if (device_ordinal < 0 || device_ordinal >= ndev) return PTK_ERR_BAD_ARG;
    if (hipSetDevice(device_ordinal) != hipSuccess) return PTK_ERR_HIP;
    ptk_ctx* c = new (std::nothrow) ptk_ctx();
    if (!c) return PTK_ERR_HIP;
    c->device = device_ordinal;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return PTK_ERR_HIP; }
    c->own_stream = true;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) == hipSuccess && cus > 0)
            c->resident_waves = cus * 16;        // 4 SIMDs x 4 waves of the BVH trace kernel per CU (launch_trace scales it to x 5 for the FLAT kernel)
        if (cus > 0) c->num_cus = cus;
    }
    if (hipMalloc(&c->d_exit, sizeof(uint32_t)) != hipSuccess || hipMemset(c->d_exit, 0, sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(&c->d_stats, 16 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&c->d_queues, PTK_QUEUE_BLOCK_BYTES) != hipSuccess)
    {
        ptk_destroy(c);
        return PTK_ERR_HIP;
    }
    for (int i = 0; i < ptk_ctx::kMaxTimedPasses; i++)
        for (int k = 0; k < 3; k++)
            if (hipEventCreate(&c->ev[i][k]) != hipSuccess) { ptk_destroy(c); return PTK_ERR_HIP; }
    for (int b = 0; b < 2; b++)
        if (hipStreamCreateWithFlags(&c->trace_stream[b], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_trace_done[b], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_acc_done[b], hipEventDisableTiming) != hipSuccess ||
            hipMalloc(&c->d_queues2[b], PTK_QUEUE_BLOCK_BYTES) != hipSuccess)
        { ptk_destroy(c); return PTK_ERR_HIP; }
    if (hipEventCreateWithFlags(&c->ev_inputs, hipEventDisableTiming) != hipSuccess) { ptk_destroy(c); return PTK_ERR_HIP; }
    {
        int lo = 0, hi = 0;                      // (numerically lowest = greatest priority)
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&c->xstream, hipStreamNonBlocking, hi) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_rendered, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_gathered, hipEventDisableTiming) != hipSuccess)
        { ptk_destroy(c); return PTK_ERR_HIP; }
    }
    *out = c;
    return PTK_OK;
}

void ptk_destroy(ptk_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    unbind_out_image(c);
    dfree(c->d_nodes); dfree(c->d_tris); dfree(c->d_shade); dfree(c->d_mats); dfree(c->d_lights);
    dfree(c->d_flat_tris);
    dfree(c->d_texinfo); dfree(c->d_texels); dfree(c->d_primary); dfree(c->d_primary_hit); dfree(c->d_primary_rd); dfree(c->d_accum); dfree(c->d_rgb8);
    dfree(c->d_pixel_rng);
    for (int b = 0; b < 2; b++)
    {
        if (c->trace_stream[b]) { (void)hipStreamSynchronize(c->trace_stream[b]); (void)hipStreamDestroy(c->trace_stream[b]); }
        if (c->ev_trace_done[b]) (void)hipEventDestroy(c->ev_trace_done[b]);
        if (c->ev_acc_done[b]) (void)hipEventDestroy(c->ev_acc_done[b]);
        dfree(c->d_samples2[b]); dfree(c->d_queues2[b]);
    }
    if (c->ev_inputs) (void)hipEventDestroy(c->ev_inputs);
    if (c->xstream) { (void)hipStreamSynchronize(c->xstream); }
    if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    if (c->ev_rendered) (void)hipEventDestroy(c->ev_rendered);
    if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
    if (c->ev_gathered) (void)hipEventDestroy(c->ev_gathered);
    dfree(c->d_packed); dfree(c->d_gathered);
    dfree(c->d_exit); dfree(c->d_stats); dfree(c->d_queues); dfree(c->d_samples);
    dfree(c->d_live_mask); dfree(c->d_live_list);
    for (int i = 0; i < ptk_ctx::kMaxTimedPasses; i++)
        for (int k = 0; k < 3; k++)
            if (c->ev[i][k]) (void)hipEventDestroy(c->ev[i][k]);
    for (hipEvent_t e : c->klog_ev) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* ptk_last_error(ptk_ctx* c)
{
    if (!c) return "null context";
    std::lock_guard<std::mutex> g(c->err_mu);
    return c->error.c_str();
}

int ptk_set_stream(ptk_ctx* c, void* s)
{
    if (!c) return PTK_ERR_BAD_ARG;
    if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    c->inputs_dirty = true;                      // re-anchor the trace streams behind the new stream
    return PTK_OK;
}

int ptk_upload_scene(ptk_ctx* c, const ptk_scene_desc* s)
{
    if (!c || !s) return PTK_ERR_BAD_ARG;
    const int32_t n = s->num_triangles;
    if (n < 0 || s->num_materials < 0 || s->num_textures < 0 || s->num_lights < 0)
        return fail(c, PTK_ERR_BAD_ARG, "negative count in scene description");
    if (n > 0 && (!s->verts || !s->normals || !s->uvs || !s->tbn || !s->smoothing || !s->material || !s->materials))
        return fail(c, PTK_ERR_BAD_ARG, "null triangle/material array");
    // (the walk addresses triangle and node records with 32-bit BYTE offsets from wave-uniform bases - ptk_kernels.hip
    // request_node / walk_step - so a record array ends below 4 GiB: 89 478 485 triangles of 48 bytes; checked before anything
    // reads the arrays.  The node array of such a scene, ~0.4 nodes of 64 bytes per triangle, stays below that by itself and is
    // checked once it exists)
    if ((uint64_t)n * (uint64_t)(TRI_F4 * sizeof(float4)) > 0xffffffffull)
        return fail(c, PTK_ERR_LIMIT, "more triangles than the kernels' 32-bit record offsets address (89 478 485)");
    if (s->num_lights > 0 && !s->lights) return fail(c, PTK_ERR_BAD_ARG, "null light array");
    // (a scene whose textures are ALL missing files - a .pts moved to another machine - has texture entries of zero extent and not
    // one texel: no atlas to point to)
    if (s->num_textures > 0 && (!s->textures || (!s->texels && s->texel_bytes > 0))) return fail(c, PTK_ERR_BAD_ARG, "null texture array");
    for (int32_t i = 0; i < n; i++)
        if (s->material[i] < 0 || s->material[i] >= s->num_materials)
            return fail(c, PTK_ERR_BAD_ARG, "triangle material index out of range");
    for (int32_t i = 0; i < s->num_lights; i++)
        if (s->lights[i] < 0 || s->lights[i] >= n) return fail(c, PTK_ERR_BAD_ARG, "light triangle index out of range");
    // the kernels' exact short reciprocal (ptk_kernels.hip rcp_ieee) covers determinants and lengths up to 2^126: coordinates
    // must stay below 2^61 in magnitude (the reference's own float arithmetic is long meaningless out there)
    float vmax = 0.0f;
    float vlo[3] = { INFINITY, INFINITY, INFINITY }, vhi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = 0; i < (size_t)n * 9; i++)
    {
        const float v = s->verts[i];
        if (!(std::fabs(v) < 2.305843e18f)) return fail(c, PTK_ERR_LIMIT, "vertex coordinate is not finite or exceeds 2^61");
        vmax = std::max(vmax, std::fabs(v));
        const int a = (int)(i % 3);
        vlo[a] = std::min(vlo[a], v); vhi[a] = std::max(vhi[a], v);
    }
    // (node origins lie within the triangle boxes padded by 1e-5 x their extent plus the degenerate-box epsilon)
    const float scene_bound = 3.1f * (1.01f * vmax + 1e-3f);
    for (int32_t i = 0; i < s->num_textures; i++)
    {
        const ptk_texture& t = s->textures[i];
        if (t.width < 0 || t.height < 0 || t.offset < 0 || (t.offset & 3) ||
            t.offset + (int64_t)t.width * t.height * 4 > s->texel_bytes)
            return fail(c, PTK_ERR_BAD_ARG, "texture outside the texel atlas");
    }
    for (int32_t i = 0; i < s->num_materials; i++)
        for (int k = 0; k < 6; k++)
            if (s->materials[i].tex[k] >= s->num_textures) return fail(c, PTK_ERR_BAD_ARG, "material texture index out of range");

    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));

    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    // BVH: on the device for scenes large enough to matter (replaces BVHNode::Construct, mesh.cpp:169-211), else - or when
    // the device build cannot meet the traversal-stack bound - by the host builder, which always can
    BuiltBvh bvh;
    DeviceBvh dbvh;
    bool on_device = false;
    float* d_verts = nullptr;
    struct TmpFree { std::vector<void*> p; ~TmpFree() { for (void* q : p) if (q) (void)hipFree(q); } } tmp;
    if (c->opt_device_build && n >= 4096)
    {
        HIPCHK(c, hipMalloc(&d_verts, (size_t)n * 9 * sizeof(float)));
        tmp.p.push_back(d_verts);
        HIPCHK(c, hipMemcpyAsync(d_verts, s->verts, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        std::string derr;
        int leaf_max = 4;
        if (g_bvh_tuning.leaf_max > 0) leaf_max = g_bvh_tuning.leaf_max;                              // ptk_set_option "bvh_leaf_max"
        on_device = build_bvh_device(d_verts, n, PTK_MAX_BVH_DEPTH, leaf_max, c->stream, dbvh, &derr);
        if (on_device && dbvh.stack_need > PTK_MAX_BVH_DEPTH) { (void)hipFree(dbvh.d_nodes); (void)hipFree(dbvh.d_order); on_device = false; }
    }
    if (!on_device)
    {
        if (!build_bvh(s->verts, n, PTK_MAX_BVH_DEPTH, 4, bvh))
            return fail(c, PTK_ERR_LIMIT, "BVH exceeds the kernel's depth / index limits");
        if (bvh.stack_need > PTK_MAX_BVH_DEPTH) return fail(c, PTK_ERR_LIMIT, "BVH needs more entries than the LDS traversal stack holds");
    }
    struct DevBvhGuard { DeviceBvh& d; bool armed; ~DevBvhGuard() { if (armed) { (void)hipFree(d.d_nodes); (void)hipFree(d.d_order); } } } dguard{ dbvh, on_device };
    if ((uint64_t)(on_device ? dbvh.num_nodes : bvh.num_nodes) * (uint64_t)(NODE_F4 * sizeof(float4)) > 0xffffffffull)
        return fail(c, PTK_ERR_LIMIT, "more BVH nodes than the kernels' 32-bit record offsets address");

    c->upload_ms[0] = ms_since(t_begin);
    const auto t_pack = std::chrono::steady_clock::now();
    // a texture with zero extent behaves like a missing image: tex2D returns 0 (image.cpp:65-66);
    // staged as a 1x1 black texel so the kernel needs no special case
    std::vector<int32_t> texmap(s->num_textures, -1);
    std::vector<int4> texinfo;
    std::vector<uint32_t> texels;
    texels.push_back(0u);                                   // texel 0: shared black texel
    for (int32_t i = 0; i < s->num_textures; i++)
    {
        const ptk_texture& t = s->textures[i];
        int4 ti;
        if (t.width == 0 || t.height == 0) { ti = make_int4(1, 1, 0, 0); }
        else
        {
            ti = make_int4(t.width, t.height, (int)texels.size(), 0);
            size_t cnt = (size_t)t.width * t.height;
            size_t base = texels.size();
            texels.resize(base + cnt);
            std::memcpy(texels.data() + base, s->texels + t.offset, cnt * 4);
        }
        texmap[i] = (int32_t)texinfo.size();
        texinfo.push_back(ti);
    }
    if (texels.size() >= (1ull << 31)) return fail(c, PTK_ERR_LIMIT, "texel atlas too large");

    std::vector<float> mats((size_t)s->num_materials * MAT_F4 * 4, 0.0f);
    for (int32_t i = 0; i < s->num_materials; i++) pack_material(s->materials[i], texmap, mats.data() + (size_t)i * MAT_F4 * 4);

    std::vector<float> tris, shade;
    c->num_leaf_tris = n;
    if (!on_device) tris.assign((size_t)n * TRI_F4 * 4, 0.0f);
    for (int32_t k = 0; k < n && !on_device; k++)
    {
        int32_t i = bvh.order[k];
        const float* v = s->verts + (size_t)i * 9;
        float* q = tris.data() + (size_t)k * TRI_F4 * 4;
        q[0] = v[0]; q[1] = v[1]; q[2] = v[2];
        q[3] = v[3] - v[0]; q[4] = v[4] - v[1]; q[5] = v[5] - v[2];      // edge1 = v2 - v1 (pathtracer.cpp:382)
        q[6] = v[6] - v[0]; q[7] = v[7] - v[1]; q[8] = v[8] - v[2];      // edge2 = v3 - v1 (pathtracer.cpp:383)
        q[9] = as_float(i);
        int32_t ot = s->materials[s->material[i]].tex[5];
        q[10] = as_float(ot >= 0 ? texmap[ot] : -1);
        q[11] = 0.0f;
    }
    if (!on_device) shade.assign((size_t)n * SHADE_F4 * 4, 0.0f);
    for (int32_t i = 0; i < n && !on_device; i++)
    {
        const float* nn = s->normals + (size_t)i * 9;
        const float* uv = s->uvs + (size_t)i * 6;
        const float* tb = s->tbn + (size_t)i * 9;
        float* q = shade.data() + (size_t)i * SHADE_F4 * 4;
        q[0] = tb[0]; q[1] = tb[1]; q[2] = tb[2];
        q[3] = as_float((int32_t)((uint32_t)s->material[i] | (s->smoothing[i] ? 0x80000000u : 0u)));
        q[4] = uv[0]; q[5] = uv[1]; q[6] = uv[2]; q[7] = uv[3];
        q[8] = uv[4]; q[9] = uv[5]; q[10] = nn[0]; q[11] = nn[1];
        q[12] = nn[2]; q[13] = nn[3]; q[14] = nn[4]; q[15] = nn[5];
        q[16] = nn[6]; q[17] = nn[7]; q[18] = nn[8]; q[19] = tb[3];
        q[20] = tb[4]; q[21] = tb[5]; q[22] = tb[6]; q[23] = tb[7];
        q[24] = tb[8];
    }
    std::vector<float> lights((size_t)s->num_lights * LIGHT_F4 * 4, 0.0f);
    for (int32_t k = 0; k < s->num_lights; k++)
    {
        int32_t i = s->lights[k];
        const float* v = s->verts + (size_t)i * 9;
        const ptk_material& m = s->materials[s->material[i]];
        float* q = lights.data() + (size_t)k * LIGHT_F4 * 4;
        q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = as_float(i);
        q[4] = v[3]; q[5] = v[4]; q[6] = v[5]; q[7] = m.emissive[0] * m.emissive_intensity;    // pathtracer.cpp:528
        q[8] = v[6]; q[9] = v[7]; q[10] = v[8]; q[11] = m.emissive[1] * m.emissive_intensity;
        q[12] = m.emissive[2] * m.emissive_intensity;
        q[13] = as_float(m.tex[5] >= 0 ? texmap[m.tex[5]] : -1);          // its opacity texture (the shadow walk tests the light triangle first)
    }

    c->upload_ms[1] = ms_since(t_pack);
    const auto t_copy = std::chrono::steady_clock::now();
    dfree(c->d_nodes); dfree(c->d_tris); dfree(c->d_shade); dfree(c->d_mats); dfree(c->d_lights);
    dfree(c->d_texinfo); dfree(c->d_texels);
    c->have_scene = false;
    auto up = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
        size_t alloc = bytes ? bytes : 64;      // (never null: the walk reads node 0 unconditionally)
        hipError_t e = hipMalloc(dst, alloc);
        if (e != hipSuccess) return e;
        if (bytes) return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return hipSuccess;
    };
    if (on_device)
    {
        dfree(c->d_flat_tris);
        // the records are packed on the device from the boundary's flat arrays (one pass each; no host-side copies)
        float *d_normals = nullptr, *d_uvs = nullptr, *d_tbn = nullptr; uint8_t* d_smooth = nullptr; int32_t *d_material = nullptr, *d_optex = nullptr;
        std::vector<int32_t> optex(std::max<int32_t>(s->num_materials, 1), -1);
        for (int32_t i = 0; i < s->num_materials; i++) { const int32_t ot = s->materials[i].tex[5]; optex[i] = ot >= 0 ? texmap[ot] : -1; }
        auto tup = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
            hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
            if (e != hipSuccess) return e;
            tmp.p.push_back(*dst);
            return bytes ? hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, c->stream) : hipSuccess;
        };
        HIPCHK(c, tup((void**)&d_normals, s->normals, (size_t)n * 9 * sizeof(float)));
        HIPCHK(c, tup((void**)&d_uvs, s->uvs, (size_t)n * 6 * sizeof(float)));
        HIPCHK(c, tup((void**)&d_tbn, s->tbn, (size_t)n * 9 * sizeof(float)));
        HIPCHK(c, tup((void**)&d_smooth, s->smoothing, (size_t)n));
        HIPCHK(c, tup((void**)&d_material, s->material, (size_t)n * sizeof(int32_t)));
        HIPCHK(c, tup((void**)&d_optex, optex.data(), optex.size() * sizeof(int32_t)));
        HIPCHK(c, hipMalloc(&c->d_tris, (size_t)n * TRI_F4 * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_shade, (size_t)n * SHADE_F4 * sizeof(float4)));
        launch_pack_tris(d_verts, dbvh.d_order, d_material, d_optex, c->d_tris, n, c->stream);
        launch_pack_shade(d_normals, d_uvs, d_tbn, d_smooth, d_material, c->d_shade, n, c->stream);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->d_nodes = dbvh.d_nodes; dguard.armed = false;
        (void)hipFree(dbvh.d_order); dbvh.d_order = nullptr;
        bvh.num_nodes = dbvh.num_nodes; bvh.depth = dbvh.depth; bvh.stack_need = dbvh.stack_need;
    }
    else
    {
        HIPCHK(c, up((void**)&c->d_nodes, bvh.nodes.data(), bvh.nodes.size() * 4));
        HIPCHK(c, up((void**)&c->d_tris, tris.data(), tris.size() * 4));
        dfree(c->d_flat_tris);
        if (n > 0 && n <= 16)
        {
            std::vector<float> flat((size_t)n * TRI_F4 * 4);
            for (int32_t k = 0; k < n; k++)             // record of leaf position k holds triangle bvh.order[k]
                std::memcpy(flat.data() + (size_t)bvh.order[k] * TRI_F4 * 4, tris.data() + (size_t)k * TRI_F4 * 4, TRI_F4 * 16);
            HIPCHK(c, up((void**)&c->d_flat_tris, flat.data(), flat.size() * 4));
        }
        HIPCHK(c, up((void**)&c->d_shade, shade.data(), shade.size() * 4));
    }
    c->built_on_device = on_device;
    HIPCHK(c, up((void**)&c->d_mats, mats.data(), mats.size() * 4));
    HIPCHK(c, up((void**)&c->d_lights, lights.data(), lights.size() * 4));
    HIPCHK(c, up((void**)&c->d_texinfo, texinfo.data(), texinfo.size() * sizeof(int4)));
    HIPCHK(c, up((void**)&c->d_texels, texels.data(), texels.size() * 4));
    c->num_nodes = bvh.num_nodes; c->num_tris = n; c->num_lights = s->num_lights; c->bvh_depth = bvh.depth; c->bvh_stack = bvh.stack_need;
    c->scene_bound = scene_bound;
    for (int a = 0; a < 3; a++) { c->scene_lo[a] = n > 0 ? vlo[a] : 0.0f; c->scene_hi[a] = n > 0 ? vhi[a] : 0.0f; }
    c->view_generation++;
    c->scene_has_opacity = false;
    for (int32_t i = 0; i < n; i++)
        if (s->materials[s->material[i]].tex[5] >= 0) { c->scene_has_opacity = true; break; }
    c->upload_ms[2] = ms_since(t_copy); c->upload_ms[3] = ms_since(t_begin);
    c->h_materials.assign(s->materials, s->materials + s->num_materials);
    c->h_texmap = texmap;
    c->h_lights = lights;
    c->h_light_material.resize(s->num_lights);
    for (int32_t k = 0; k < s->num_lights; k++) c->h_light_material[k] = s->material[s->lights[k]];
    c->primary_hit_dirty = true;
    c->have_scene = true;
    c->inputs_dirty = true;
    return PTK_OK;
}

// Material edits after BuildBVH: in the reference `Triangle::mat` points into mLoadedObjects, so a SetMaterial is seen by
// the very next RenderFrame() without a rebuild, while the light LIST stays the one BuildBVH collected
// (pathtracer.cpp:250-258, :267-273, :528).  Same here: the material table and the lights' colours are rewritten in
// place; geometry, BVH, textures and the light list are untouched.  Texture bindings must be the uploaded ones.
int ptk_update_materials(ptk_ctx* c, int32_t num_materials, const ptk_material* materials)
{
    if (!c || !materials) return PTK_ERR_BAD_ARG;
    if (!c->have_scene) return fail(c, PTK_ERR_BAD_ARG, "ptk_upload_scene has not been called");
    if (num_materials != (int32_t)c->h_materials.size()) return fail(c, PTK_ERR_BAD_ARG, "material count differs from the uploaded scene");
    for (int32_t i = 0; i < num_materials; i++)
        for (int k = 0; k < 6; k++)
            if (materials[i].tex[k] != c->h_materials[i].tex[k])
                return fail(c, PTK_ERR_BAD_ARG, "texture bindings changed: upload the scene again");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<float> mats((size_t)num_materials * MAT_F4 * 4, 0.0f);
    for (int32_t i = 0; i < num_materials; i++) pack_material(materials[i], c->h_texmap, mats.data() + (size_t)i * MAT_F4 * 4);
    for (size_t k = 0; k < c->h_light_material.size(); k++)
    {
        const ptk_material& m = materials[c->h_light_material[k]];
        float* q = c->h_lights.data() + k * LIGHT_F4 * 4;
        q[7] = m.emissive[0] * m.emissive_intensity; q[11] = m.emissive[1] * m.emissive_intensity; q[12] = m.emissive[2] * m.emissive_intensity;
    }
    // behind everything already queued (stream-ordered): renders in flight keep the old table
    if (!mats.empty()) HIPCHK(c, hipMemcpyAsync(c->d_mats, mats.data(), mats.size() * 4, hipMemcpyHostToDevice, c->stream));
    if (!c->h_lights.empty()) HIPCHK(c, hipMemcpyAsync(c->d_lights, c->h_lights.data(), c->h_lights.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // the host vectors above go out of scope
    c->h_materials.assign(materials, materials + num_materials);
    c->inputs_dirty = true;
    return PTK_OK;
}

int ptk_set_camera(ptk_ctx* c, const float pos[3], const float dir[3], const float up[3],
                   float focal, float fovy_deg, float focal_dist, float aperture)
{
    if (!c || !pos || !dir || !up) return PTK_ERR_BAD_ARG;
    // the same bound as for scene coordinates (ptk_upload_scene): the kernels' short reciprocal and the implied u > 1 test
    // of the triangle intersection hold while ray origins stay below 2^61
    for (int a = 0; a < 3; a++)
        if (!(std::fabs(pos[a]) < 2.305843e18f)) return fail(c, PTK_ERR_LIMIT, "camera position is not finite or exceeds 2^61");
    for (int a = 0; a < 3; a++) c->cam_pos[a] = pos[a];
    normalize3(dir, c->cam_dir);                           // pathtracer.cpp:336
    normalize3(up, c->cam_up);                             // pathtracer.cpp:337
    c->focal = focal;                                      // SetProjection, pathtracer.cpp:340-350
    if (c->focal <= 0.0f) c->focal = 0.1f;
    c->fovy = fovy_deg;
    if (c->fovy <= 0.0f) c->fovy = 0.1f;
    else if (c->fovy >= 180.0f) c->fovy = 179.5;
    c->focal_dist = focal_dist;
    c->aperture = aperture;
    c->primary_dirty = true;
    return PTK_OK;
}

int ptk_set_frame(ptk_ctx* c, int width, int height, int max_depth)
{
    if (!c) return PTK_ERR_BAD_ARG;
    if (width <= 0 || height <= 0 || (int64_t)width * height > (1ll << 28)) return fail(c, PTK_ERR_BAD_ARG, "bad resolution");
    HIPCHK(c, hipSetDevice(c->device));
    c->max_depth = max_depth;
    if (width != c->width || height != c->height || !c->d_accum)
    {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dfree(c->d_primary); dfree(c->d_primary_hit); dfree(c->d_primary_rd); dfree(c->d_accum); dfree(c->d_rgb8); dfree(c->d_pixel_rng);
        c->pixel_rng_valid = false;
        size_t px = (size_t)width * height;
        HIPCHK(c, hipMalloc(&c->d_primary, px * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_primary_hit, px * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_primary_rd, px * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_pixel_rng, px * sizeof(uint2)));
        HIPCHK(c, hipMalloc(&c->d_accum, px * 3 * sizeof(float)));
        HIPCHK(c, hipMalloc(&c->d_rgb8, px * 3));
        c->width = width; c->height = height;
        c->d_accum_bound = nullptr;
        unbind_out_image(c, true);               // another resolution: the caller's buffer has another size (main.cpp:3425-3446)
        HIPCHK(c, hipMemsetAsync(c->d_accum, 0, px * 3 * sizeof(float), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_rgb8, 0, px * 3, c->stream));
        c->samples = 0;
        c->primary_dirty = true;
    }
    return PTK_OK;
}

int ptk_set_tile(ptk_ctx* c, int rank, int world)
{
    if (!c || world < 1 || rank < 0 || rank >= world) return PTK_ERR_BAD_ARG;
    if (rank != c->rank || world != c->world) c->out_full_next = true;      // other tiles: a bound hand-off buffer is rewritten whole
    c->rank = rank; c->world = world;
    return PTK_OK;
}

int ptk_reset(ptk_ctx* c)
{
    if (!c) return PTK_ERR_BAD_ARG;
    if (!accum_ptr(c)) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    size_t px = (size_t)c->width * c->height;
    HIPCHK(c, hipMemsetAsync(accum_ptr(c), 0, px * 3 * sizeof(float), c->stream));     // pathtracer.cpp:745-751
    HIPCHK(c, hipMemsetAsync(c->d_rgb8, 0, px * 3, c->stream));
    c->samples = 0;
    if (c->out_host_dev)
    {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::memset(c->out_host, 0, px * 3);
    }
    if (c->out_device) HIPCHK(c, hipMemsetAsync(c->out_device, 0, px * 3, c->stream));
    c->out_full_next = true;                     // (a bound OpenGL buffer is rewritten whole by the next render)
    return PTK_OK;
}

int ptk_render(ptk_ctx* c, uint32_t first_sample, uint32_t spp_count, uint64_t seed)
{
    if (!c) return PTK_ERR_BAD_ARG;
    if (!c->have_scene) return fail(c, PTK_ERR_BAD_ARG, "ptk_upload_scene has not been called");
    if (!accum_ptr(c) || !c->d_primary) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    if (c->bvh_stack > PTK_MAX_BVH_DEPTH) return fail(c, PTK_ERR_LIMIT, "BVH needs more entries than the LDS traversal stack holds");
    HIPCHK(c, hipSetDevice(c->device));
    c->last_launches = 0;
    c->timed = false;
    if (spp_count == 0) return PTK_OK;
    // RenderFrame() begins with mExit = false (pathtracer.cpp:742): an Exit() only cuts the frame(s) in flight, the next
    // call renders again.
    // (an Exit() names the generation of the render it interrupts; this render gets a new one, so nothing needs clearing
    // and consecutive renders stay free to overlap)
    c->render_gen.fetch_add(1);
    int rc = ensure_primary(c);
    if (rc != PTK_OK) return rc;
    rc = run_passes(c, first_sample, spp_count, seed, false, accum_ptr(c), c->d_rgb8, c->d_exit, true);
    if (rc != PTK_OK) return rc;
    c->timed = true;
    c->samples = (int)(first_sample + spp_count);
    return PTK_OK;
}

int ptk_collect_stats(ptk_ctx* c, uint32_t first_sample, uint32_t spp_count, uint64_t seed, ptk_stats* out)
{
    if (!c || !out) return PTK_ERR_BAD_ARG;
    if (!c->have_scene || !c->d_primary) return fail(c, PTK_ERR_BAD_ARG, "scene / frame not set");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_primary(c);
    if (rc != PTK_OK) return rc;
    // untimed counters-enabled variant; renders into a scratch accumulator so the image is untouched
    size_t px = (size_t)c->width * c->height;
    float* scratch = nullptr; uint8_t* scratch8 = nullptr;
    HIPCHK(c, hipMalloc(&scratch, px * 3 * sizeof(float)));
    if (hipMalloc(&scratch8, px * 3) != hipSuccess) { (void)hipFree(scratch); return fail(c, PTK_ERR_HIP, "hipMalloc"); }
    (void)hipMemsetAsync(scratch, 0, px * 3 * sizeof(float), c->stream);
    (void)hipMemsetAsync(c->d_stats, 0, 16 * sizeof(unsigned long long), c->stream);
    c->inputs_dirty = true;                      // ... and the zeroed counters
    rc = run_passes(c, first_sample, spp_count, seed, true, scratch, scratch8, nullptr, false);
    c->timed = false;
    unsigned long long h[16] = { 0 };
    hipError_t e = hipMemcpyAsync(h, c->d_stats, sizeof(h), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(scratch); (void)hipFree(scratch8);
    if (rc != PTK_OK) return rc;
    if (e != hipSuccess) return fail(c, PTK_ERR_HIP, hipGetErrorString(e));
    out->samples = owned_pixels(c) * (unsigned long long)spp_count; out->rays = h[1]; out->shadow_rays = h[2]; out->node_visits = h[3];
    out->tri_tests = h[4]; out->hits_shaded = h[5]; out->tex_fetches = h[6];
    out->walk_wave_iters = h[7]; out->walk_lane_iters = h[8]; out->shade_wave_execs = h[9]; out->shade_lanes = h[10];
    out->gen_wave_execs = h[11]; out->gen_lanes = h[12];
    out->tri_wave_execs = h[13]; out->tri_lanes = h[14];
    out->max_walk_nodes = h[15]; out->paths_started = h[0];
    return PTK_OK;
}

int ptk_bind_out_image(ptk_ctx* c, uint8_t* host_out)
{
    if (!c) return PTK_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    // (the same buffer again: nothing to do - unless this context page-locked it, see below: a block freed and allocated again at
    // the same address has other pages behind it than the ones that were locked)
    if (host_out == c->out_host && !c->out_registered && !c->out_device && !c->gl_res) return PTK_OK;
    if (c->stream) HIPCHK(c, hipStreamSynchronize(c->stream));      // nothing may still be writing into the old buffer
    unbind_out_image(c);
    if (!host_out) return PTK_OK;
    if (c->width <= 0 || !c->d_rgb8) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    // Zero-copy needs memory the GPU can write.  A buffer that already IS such memory - ptk_host_alloc, or anything the caller
    // page-locked and mapped itself - is used as it stands.  FOREIGN pageable memory (the viewer's `new GLubyte[w * h * 3]`,
    // main.cpp:3435) is left alone by default: the reference's caller deletes texData BEFORE it hands over the next buffer
    // (InitializeFrame, main.cpp:3433-3445) and at exit without a word to the tracer (OnExit, :3622), and memory freed while
    // still registered with the runtime poisons what the allocator puts there next (tools/soak_api.py, round 3).
    // Such a buffer is simply not bound: ptk_resolve_rgb8 copies the frame into it, which is always correct.
    // ptk_set_option("register_out_image", 1) opts in to page-locking it in place (hipHostRegister) for callers that promise
    // to unbind - ptk_bind_out_image(ctx, NULL) - BEFORE they free it.
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, host_out, 0) == hipSuccess && dev) c->out_host_dev = (uint8_t*)dev;       // ptk_host_alloc / already page-locked
    else
    {
        (void)hipGetLastError();
        if (c->opt_register_out)
        {
            const size_t bytes = (size_t)c->width * c->height * 3;
            if (hipHostRegister(host_out, bytes, hipHostRegisterMapped) == hipSuccess)
            {
                if (hipHostGetDevicePointer(&dev, host_out, 0) == hipSuccess && dev) { c->out_host_dev = (uint8_t*)dev; c->out_registered = true; }
                else (void)hipHostUnregister(host_out);
            }
            (void)hipGetLastError();             // not lockable: ptk_resolve_rgb8 keeps copying
        }
    }
    if (!c->out_host_dev) return PTK_OK;         // not bound (see above)
    c->out_host = host_out;
    std::memset(host_out, 0, (size_t)c->width * c->height * 3);     // pixels that are black for every sample are written once at most
    c->out_full_next = true;
    return PTK_OK;
}

// The interactive loop waits for frames of a few hundred microseconds: poll the stream for a while before handing the wait to
// the runtime, whose blocking wait costs tens of microseconds to wake up.
static int wait_polled(ptk_ctx* c)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;)
    {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return PTK_OK;
        if (q != hipErrorNotReady) return fail(c, PTK_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PTK_OK;
}

int ptk_bind_out_device(ptk_ctx* c, void* device_rgb8)
{
    if (!c) return PTK_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (device_rgb8 && device_rgb8 == c->out_device) return PTK_OK;
    if (c->stream) HIPCHK(c, hipStreamSynchronize(c->stream));
    unbind_out_image(c);
    if (!device_rgb8) return PTK_OK;
    if (c->width <= 0 || !c->d_rgb8) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, device_rgb8) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != c->device)
    {
        (void)hipGetLastError();
        return fail(c, PTK_ERR_BAD_ARG, "ptk_bind_out_device: not a device allocation of this context's GPU");
    }
    c->out_device = (uint8_t*)device_rgb8;
    HIPCHK(c, hipMemsetAsync(c->out_device, 0, (size_t)c->width * c->height * 3, c->stream));
    c->out_full_next = true;
    return PTK_OK;
}

int ptk_bind_gl_buffer(ptk_ctx* c, unsigned int gl_buffer)
{
    if (!c) return PTK_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (gl_buffer != 0 && gl_buffer == c->gl_buffer && c->gl_res) return PTK_OK;
    if (c->stream) HIPCHK(c, hipStreamSynchronize(c->stream));
    unbind_out_image(c);
    if (gl_buffer == 0) return PTK_OK;
    // (no frame needed yet: the buffer's size is checked against the frame each time it is mapped, run_passes)
    // Registration talks to the OpenGL driver through the calling thread's CURRENT context (the viewer's: main.cpp creates it
    // before the path tracer renders anything).  The library does not link OpenGL: the process that calls this has it loaded,
    // and a process without a current context gets an error here instead of a crash inside the runtime.
    typedef void* (*current_ctx_fn)(void);
    bool have_ctx = false;
    for (const char* name : { "glXGetCurrentContext", "eglGetCurrentContext", "wglGetCurrentContext" })
    {
        current_ctx_fn fn = (current_ctx_fn)dlsym(RTLD_DEFAULT, name);
        if (fn && fn() != nullptr) { have_ctx = true; break; }
    }
    if (!have_ctx) return fail(c, PTK_ERR_BAD_ARG, "ptk_bind_gl_buffer: no OpenGL context is current on the calling thread");
    hipGraphicsResource_t res = nullptr;
    const hipError_t e = hipGraphicsGLRegisterBuffer(&res, gl_buffer, hipGraphicsRegisterFlagsNone);
    if (e != hipSuccess || !res)
    {
        (void)hipGetLastError();
        return fail(c, PTK_ERR_HIP, std::string("hipGraphicsGLRegisterBuffer: ") + hipGetErrorString(e));
    }
    c->gl_res = res; c->gl_buffer = gl_buffer;
    c->out_full_next = true;
    return PTK_OK;
}

int ptk_resolve_rgb8(ptk_ctx* c, uint8_t* host_out)
{
    if (!c || !host_out) return PTK_ERR_BAD_ARG;
    if (!c->d_rgb8) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    if (host_out == c->out_host && c->out_host_dev && !c->out_full_next)
    {
        // bound: the accumulate kernel of the last render has written the frame there (out_full_next is only set while no
        // render has filled the buffer since it was bound / reset).  The interactive loop waits for a frame of a few hundred
        // microseconds: poll the stream for a while before handing the wait to the runtime, whose blocking wait costs
        // tens of microseconds to wake up
        return wait_polled(c);
    }
    HIPCHK(c, hipMemcpyAsync(host_out, c->d_rgb8, (size_t)c->width * c->height * 3, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PTK_OK;
}

void* ptk_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void ptk_host_free(void* p) { if (p) (void)hipHostFree(p); }

int ptk_read_accum(ptk_ctx* c, float* host_out)
{
    if (!c || !host_out) return PTK_ERR_BAD_ARG;
    if (!accum_ptr(c)) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(host_out, accum_ptr(c), (size_t)c->width * c->height * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PTK_OK;
}

int ptk_write_accum(ptk_ctx* c, const float* host_in, int samples)
{
    if (!c || !host_in || samples < 0) return PTK_ERR_BAD_ARG;
    if (!accum_ptr(c)) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(accum_ptr(c), host_in, (size_t)c->width * c->height * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->samples = samples;
    c->out_full_next = true;
    return PTK_OK;
}

int ptk_samples(ptk_ctx* c) { return c ? c->samples.load() : 0; }

int ptk_request_exit(ptk_ctx* c)
{
    if (!c) return PTK_ERR_BAD_ARG;
    // blocks that have not started yet read the flag and return (kernel prologue); written with a
    // blocking copy outside the render stream so it lands while a render is in flight
    uint32_t gen = c->render_gen.load();
    c->out_full_next = true;                     // a cut render may have skipped a full write of the bound hand-off buffer
    (void)hipSetDevice(c->device);
    (void)hipMemcpy(c->d_exit, &gen, sizeof(gen), hipMemcpyHostToDevice);
    return PTK_OK;
}

int ptk_synchronize(ptk_ctx* c)
{
    if (!c) return PTK_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->out_device || c->gl_res) return wait_polled(c);     // a frame of the interactive loop (device-resident hand-off)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PTK_OK;
}

int ptk_accum_device_ptr(ptk_ctx* c, void** dev_ptr, size_t* bytes)
{
    if (!c || !dev_ptr) return PTK_ERR_BAD_ARG;
    *dev_ptr = accum_ptr(c);
    if (bytes) *bytes = (size_t)c->width * c->height * 3 * sizeof(float);
    return *dev_ptr ? PTK_OK : fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
}

int ptk_rgb8_device_ptr(ptk_ctx* c, void** dev_ptr, size_t* bytes)
{
    if (!c || !dev_ptr) return PTK_ERR_BAD_ARG;
    *dev_ptr = c->d_rgb8;
    if (bytes) *bytes = (size_t)c->width * c->height * 3;
    return *dev_ptr ? PTK_OK : fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
}

int ptk_bind_accum(ptk_ctx* c, void* dev_ptr)
{
    if (!c) return PTK_ERR_BAD_ARG;
    if (!c->d_accum) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    if ((float*)dev_ptr != c->d_accum_bound) c->out_full_next = true;       // another accumulator: pixels that were skipped as "always black" may hold light there
    c->d_accum_bound = (float*)dev_ptr;
    return PTK_OK;
}

// ---- multi-GPU exchange step -------------------------------------------------------------------------------------
static int64_t packed_floats_of(int width, int height, int rank, int world)
{
    const int64_t tiles = (int64_t)((width + PTK_TILE - 1) / PTK_TILE) * ((height + PTK_TILE - 1) / PTK_TILE);
    const int64_t owned = tiles <= rank ? 0 : (tiles - rank + world - 1) / world;
    return owned * PTK_TILE * PTK_TILE * 3;
}

int64_t ptk_packed_floats(int width, int height, int rank, int world)
{
    if (width <= 0 || height <= 0 || world < 1 || rank < 0 || rank >= world) return -1;
    return packed_floats_of(width, height, rank, world);
}

int ptk_packed_layout(int width, int height, int rank, int world, int64_t* src_index)
{
    if (width <= 0 || height <= 0 || world < 1 || rank < 0 || rank >= world || !src_index) return PTK_ERR_BAD_ARG;
    const int tiles_x = (width + PTK_TILE - 1) / PTK_TILE, num_tiles = tiles_x * ((height + PTK_TILE - 1) / PTK_TILE);
    int64_t k = 0;
    for (int tile = rank; tile < num_tiles; tile += world)
    {
        const int ty = tile / tiles_x, tx = (tile % tiles_x + tiles_x - (3 * ty) % tiles_x) % tiles_x;
        for (int p = 0; p < PTK_TILE * PTK_TILE; p++)
        {
            const int px = tx * PTK_TILE + (p & 15), py = ty * PTK_TILE + (p >> 4);
            const bool on = px < width && py < height;
            const int64_t a = ((int64_t)(height - 1 - py) * width + px) * 3;
            for (int ch = 0; ch < 3; ch++) src_index[k++] = on ? a + ch : -1;
        }
    }
    return PTK_OK;
}

int ptk_comm_unique_id(void* id_out)
{
    if (!id_out) return PTK_ERR_BAD_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "ptk.h promises 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return PTK_ERR_RCCL;
    std::memcpy(id_out, &id, sizeof(id));
    return PTK_OK;
}

int ptk_comm_init(ptk_ctx* c, const void* id_in, int rank, int world)
{
    if (!c || !id_in || world < 1 || world > PTK_MAX_RANKS || rank < 0 || rank >= world) return PTK_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
    // ncclCommInitRank blocks until EVERY rank of the group has called it: a rank that died on the way (or was never started)
    // would hang the others for good.  It runs on a helper thread and is waited for with a bound; on a timeout the caller gets an
    // error that names the rank and is expected to end the process (the helper thread is abandoned with its own state).
    struct InitJob { ncclComm_t comm = nullptr; ncclResult_t r = ncclSuccess; std::atomic<int> done{ 0 }; };
    auto job = std::make_shared<InitJob>();
    ncclUniqueId id;
    std::memcpy(&id, id_in, sizeof(id));
    const int device = c->device;
    std::thread([job, id, rank, world, device] {
        (void)hipSetDevice(device);
        job->r = ncclCommInitRank(&job->comm, world, id, rank);
        job->done.store(1);
    }).detach();
    const auto t0 = std::chrono::steady_clock::now();
    while (!job->done.load())
    {
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > c->opt_comm_timeout_s)
        {
            char msg[256];
            std::snprintf(msg, sizeof(msg), "ncclCommInitRank: rank %d of %d (HIP device %d) waited %.0f s for the other ranks to join the communicator - "
                          "is every rank running, on a device of its own?", rank, world, device, waited);
            return fail(c, PTK_ERR_RCCL, msg);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    if (job->r != ncclSuccess || !job->comm)
    {
        char msg[256];
        std::snprintf(msg, sizeof(msg), "ncclCommInitRank: rank %d of %d (HIP device %d): %s", rank, world, device, ncclGetErrorString(job->r));
        return fail(c, PTK_ERR_RCCL, msg);
    }
    c->comm = job->comm;
    c->comm_rank = rank; c->comm_world = world;
    c->rank = rank; c->world = world;            // the frame is split over the group (ptk_set_tile)
    return PTK_OK;
}

int ptk_comm_info(ptk_ctx* c, int* rank, int* world, int* comm_device, int* ctx_device)
{
    if (!c) return PTK_ERR_BAD_ARG;
    if (ctx_device) *ctx_device = c->device;
    if (!c->comm) return fail(c, PTK_ERR_BAD_ARG, "no communicator: call ptk_comm_init");
    int r = -1, w = 0, d = -1;
    if (ncclCommCount(c->comm, &w) != ncclSuccess || ncclCommUserRank(c->comm, &r) != ncclSuccess || ncclCommCuDevice(c->comm, &d) != ncclSuccess)
        return fail(c, PTK_ERR_RCCL, "ncclCommCount / ncclCommUserRank / ncclCommCuDevice failed");
    if (rank) *rank = r;
    if (world) *world = w;
    if (comm_device) *comm_device = d;
    return PTK_OK;
}

int ptk_comm_destroy(ptk_ctx* c)
{
    if (!c) return PTK_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
    return PTK_OK;
}

// Packed gather.  Everything is queued on the context's exchange stream behind what the render stream holds now:
//   pack kernel (snapshot of the owned tiles; the render stream waits only for this) -> grouped ncclSend / ncclRecv
//   (each rank's 1/world of the image goes straight to the root over its own xGMI link) -> root: unpack kernel.
// The next ptk_render may be issued at once: its trace kernel does not touch the accumulator and overlaps the exchange.
int ptk_gather_accum(ptk_ctx* c, void* rccl_comm, int root)
{
    if (!c) return PTK_ERR_BAD_ARG;
    ncclComm_t comm = rccl_comm ? (ncclComm_t)rccl_comm : c->comm;
    if (!comm) return fail(c, PTK_ERR_BAD_ARG, "no communicator: pass one or call ptk_comm_init");
    if (!accum_ptr(c)) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    int world = 0, rank = 0;
    if (ncclCommCount(comm, &world) != ncclSuccess || ncclCommUserRank(comm, &rank) != ncclSuccess)
        return fail(c, PTK_ERR_RCCL, "ncclCommCount / ncclCommUserRank failed");
    if (world != c->world || rank != c->rank) return fail(c, PTK_ERR_BAD_ARG, "communicator rank / size differ from ptk_set_tile");
    if (root < 0 || root >= world || world > PTK_MAX_RANKS) return fail(c, PTK_ERR_BAD_ARG, "bad root");
    const int W = c->width, H = c->height;
    long long bases[PTK_MAX_RANKS] = { 0 };
    size_t total = 0;
    for (int r = 0; r < world; r++) { bases[r] = (long long)total; total += (size_t)packed_floats_of(W, H, r, world); }
    const size_t mine = (size_t)packed_floats_of(W, H, rank, world);
    const size_t need = rank == root ? total : mine;
    if (need > c->packed_floats)
    {
        HIPCHK(c, hipStreamSynchronize(c->xstream));
        dfree(c->d_packed); c->packed_floats = 0;
        HIPCHK(c, hipMalloc(&c->d_packed, std::max<size_t>(need, 4) * sizeof(float)));
        c->packed_floats = need;
    }
    const size_t img = (size_t)W * H * 3;
    if (rank == root && img > c->gathered_floats)
    {
        HIPCHK(c, hipStreamSynchronize(c->xstream));
        dfree(c->d_gathered); c->gathered_floats = 0;
        HIPCHK(c, hipMalloc(&c->d_gathered, img * sizeof(float)));
        c->gathered_floats = img;
    }
    HIPCHK(c, hipEventRecord(c->ev_rendered, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->xstream, c->ev_rendered, 0));
    float* my_slot = c->d_packed + (rank == root ? bases[rank] : 0);
    launch_pack_owned(accum_ptr(c), my_slot, W, H, rank, world, c->xstream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_packed, c->xstream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_packed, 0));      // the next accumulate_kernel may overwrite the accumulator from here on
    if (world > 1)
    {
        ncclResult_t r = ncclGroupStart();
        if (r == ncclSuccess)
        {
            if (rank == root)
            {
                for (int src = 0; src < world && r == ncclSuccess; src++)
                {
                    const size_t n = (size_t)packed_floats_of(W, H, src, world);
                    if (src != root && n) r = ncclRecv(c->d_packed + bases[src], n, ncclFloat, src, comm, c->xstream);
                }
            }
            else if (mine) r = ncclSend(my_slot, mine, ncclFloat, root, comm, c->xstream);
            ncclResult_t e = ncclGroupEnd();
            if (r == ncclSuccess) r = e;
        }
        if (r != ncclSuccess) return fail(c, PTK_ERR_RCCL, std::string("packed gather (ncclSend/ncclRecv): ") + ncclGetErrorString(r));
    }
    if (rank == root)
    {
        launch_unpack_all(c->d_packed, bases, c->d_gathered, W, H, world, c->xstream);
        HIPCHK(c, hipGetLastError());
        c->gathered_w = W; c->gathered_h = H;
    }
    HIPCHK(c, hipEventRecord(c->ev_gathered, c->xstream));
    c->gather_pending = true;
    c->gather_step++; c->gather_root = root;
    c->gather_bytes = (rank == root ? total - mine : mine) * sizeof(float);
    return PTK_OK;
}

// Bounded: polls the exchange's last event; when it has not fired within comm_timeout_s (a rank never entered its
// ptk_gather_accum, or died in it) the communicator is aborted - which releases the transfer kernel stuck on the exchange
// stream - and the caller gets PTK_ERR_RCCL with rank, step and the bytes that were expected.  An asynchronous RCCL error
// (a peer's process gone) ends the wait at once.
int ptk_gather_wait(ptk_ctx* c)
{
    if (!c) return PTK_ERR_BAD_ARG;
    if (!c->gather_pending) return PTK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;)
    {
        const hipError_t q = hipEventQuery(c->ev_gathered);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { c->gather_pending = false; return fail(c, PTK_ERR_HIP, std::string("hipEventQuery (exchange): ") + hipGetErrorString(q)); }
        (void)hipGetLastError();
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ncclResult_t async = ncclSuccess;
        const bool poll_comm = c->comm && (++spins & 1023u) == 0;
        if (poll_comm && ncclCommGetAsyncError(c->comm, &async) != ncclSuccess) async = ncclSystemError;
        if (waited > c->opt_comm_timeout_s || (async != ncclSuccess && async != ncclInProgress))
        {
            char msg[384];
            std::snprintf(msg, sizeof(msg), "exchange step %llu on rank %d of %d (root %d) %s after %.1f s: %s %zu bytes%s",
                          c->gather_step, c->rank, c->world, c->gather_root,
                          async != ncclSuccess && async != ncclInProgress ? "failed" : "timed out", waited,
                          c->rank == c->gather_root ? "still expecting" : "still sending", c->gather_bytes,
                          async != ncclSuccess && async != ncclInProgress ? (std::string(" - RCCL: ") + ncclGetErrorString(async)).c_str() : " - a rank never joined this step");
            if (c->comm)
            {
                // the abort releases RCCL's own kernels, but it also waits for whatever else sits on the stream: it runs on a helper
                // thread and this call gives it one more timeout's worth (at most 5 s) before it returns regardless
                ncclComm_t doomed = c->comm;
                c->comm = nullptr;
                auto done = std::make_shared<std::atomic<bool>>(false);
                std::thread([doomed, done] { (void)ncclCommAbort(doomed); done->store(true); }).detach();
                const auto a0 = std::chrono::steady_clock::now();
                const double grace = std::min(c->opt_comm_timeout_s, 5.0);
                while (!done->load() && std::chrono::duration<double>(std::chrono::steady_clock::now() - a0).count() < grace)
                    std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            c->gather_pending = false;
            return fail(c, PTK_ERR_RCCL, msg);
        }
        if (waited > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    c->gather_pending = false;
    return PTK_OK;
}

// Test hook of the bounded waits: one lane that keeps the exchange stream busy for a fixed time (wall clock, 100 MHz), so that
// a single GPU can show what ptk_gather_wait does when an exchange step does not complete in time.  It always ends by itself.
__global__ void stall_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

int ptk_debug_stall_exchange(ptk_ctx* c, int milliseconds)
{
    if (!c || milliseconds < 0 || milliseconds > 10000) return PTK_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    stall_kernel<<<1, 1, 0, c->xstream>>>((unsigned long long)milliseconds * 100000ull);
    HIPCHK(c, hipGetLastError());
    return PTK_OK;
}

int ptk_gathered_device_ptr(ptk_ctx* c, void** dev_ptr, size_t* bytes)
{
    if (!c || !dev_ptr) return PTK_ERR_BAD_ARG;
    *dev_ptr = nullptr;
    if (bytes) *bytes = 0;
    if (!c->d_gathered || c->gathered_w == 0) return fail(c, PTK_ERR_BAD_ARG, "no gathered image on this rank (not the root, or ptk_gather_accum not called)");
    if (c->gathered_w != c->width || c->gathered_h != c->height)
        return fail(c, PTK_ERR_BAD_ARG, "the gathered image was combined for another resolution: call ptk_gather_accum again after ptk_set_frame");
    *dev_ptr = c->d_gathered;
    if (bytes) *bytes = (size_t)c->gathered_w * c->gathered_h * 3 * sizeof(float);
    return PTK_OK;
}

int ptk_read_gathered(ptk_ctx* c, float* host_out)
{
    if (!c || !host_out) return PTK_ERR_BAD_ARG;
    if (!c->d_gathered || c->gathered_w == 0) return fail(c, PTK_ERR_BAD_ARG, "no gathered image on this rank (not the root, or ptk_gather_accum not called)");
    if (c->gathered_w != c->width || c->gathered_h != c->height)
        return fail(c, PTK_ERR_BAD_ARG, "the gathered image was combined for another resolution: call ptk_gather_accum again after ptk_set_frame");
    int rc = ptk_gather_wait(c);
    if (rc != PTK_OK) return rc;
    HIPCHK(c, hipMemcpy(host_out, c->d_gathered, (size_t)c->width * c->height * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return PTK_OK;
}

// parity probes of the two exchange kernels (one GPU can play every rank of a split)
int ptk_probe_pack(ptk_ctx* c, int rank, int world, float* host_out)
{
    if (!c || !host_out || world < 1 || world > PTK_MAX_RANKS || rank < 0 || rank >= world) return PTK_ERR_BAD_ARG;
    if (!accum_ptr(c)) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)packed_floats_of(c->width, c->height, rank, world);
    if (n == 0) return PTK_OK;
    float* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n * sizeof(float)));
    launch_pack_owned(accum_ptr(c), d, c->width, c->height, rank, world, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host_out, d, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, PTK_ERR_HIP, hipGetErrorString(e));
    return PTK_OK;
}

int ptk_probe_unpack(ptk_ctx* c, int world, const float* host_packed, float* host_image)
{
    if (!c || !host_packed || !host_image || world < 1 || world > PTK_MAX_RANKS) return PTK_ERR_BAD_ARG;
    if (c->width <= 0) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    long long bases[PTK_MAX_RANKS] = { 0 };
    size_t total = 0;
    for (int r = 0; r < world; r++) { bases[r] = (long long)total; total += (size_t)packed_floats_of(c->width, c->height, r, world); }
    const size_t img = (size_t)c->width * c->height * 3;
    float *d = nullptr, *di = nullptr;
    HIPCHK(c, hipMalloc(&d, std::max<size_t>(total, 4) * sizeof(float)));
    if (hipMalloc(&di, img * sizeof(float)) != hipSuccess) { (void)hipFree(d); return fail(c, PTK_ERR_HIP, "hipMalloc"); }
    hipError_t e = hipMemcpyAsync(d, host_packed, total * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(di, 0xff, img * sizeof(float), c->stream);       // NaNs: every pixel must be written
    if (e == hipSuccess) { launch_unpack_all(d, bases, di, c->width, c->height, world, c->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(host_image, di, img * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d); (void)hipFree(di);
    if (e != hipSuccess) return fail(c, PTK_ERR_HIP, hipGetErrorString(e));
    return PTK_OK;
}

int ptk_last_render_ms(ptk_ctx* c, float* ms, int* launches)
{
    if (!c || !ms) return PTK_ERR_BAD_ARG;
    float t = 0.0f, a = 0.0f;
    int rc = ptk_last_kernel_ms(c, &t, &a);
    *ms = t + a;
    if (launches) *launches = c->last_launches;
    return rc;
}

int ptk_kernel_log(ptk_ctx* c, int capacity)
{
    if (!c || capacity < 0 || capacity > (1 << 20)) return PTK_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    for (hipEvent_t e : c->klog_ev) (void)hipEventDestroy(e);
    c->klog_ev.clear(); c->klog_n = 0;
    for (int i = 0; i < 2 * capacity; i++)
    {
        hipEvent_t e = nullptr;
        HIPCHK(c, hipEventCreate(&e));
        c->klog_ev.push_back(e);
    }
    return PTK_OK;
}

int ptk_kernel_log_read(ptk_ctx* c, float* trace_ms, int max_entries, int* num_entries)
{
    if (!c || !num_entries || (max_entries > 0 && !trace_ms)) return PTK_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    for (int k = 0; k < 2; k++) if (c->trace_stream[k]) HIPCHK(c, hipStreamSynchronize(c->trace_stream[k]));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int n = std::min(c->klog_n, max_entries);
    for (int i = 0; i < n; i++) HIPCHK(c, hipEventElapsedTime(&trace_ms[i], c->klog_ev[2 * i], c->klog_ev[2 * i + 1]));
    *num_entries = n;
    c->klog_n = 0;
    return PTK_OK;
}

int ptk_last_kernel_ms(ptk_ctx* c, float* trace_ms, float* accumulate_ms)
{
    if (!c || !trace_ms || !accumulate_ms) return PTK_ERR_BAD_ARG;
    *trace_ms = 0.0f; *accumulate_ms = 0.0f;
    if (!c->timed) return PTK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int n = std::min(c->last_passes, (int)ptk_ctx::kMaxTimedPasses);
    for (int i = 0; i < n; i++)
    {
        float a = 0.0f, b = 0.0f;
        HIPCHK(c, hipEventSynchronize(c->ev[i][2]));
        HIPCHK(c, hipEventElapsedTime(&a, c->ev[i][0], c->ev[i][1]));
        HIPCHK(c, hipEventElapsedTime(&b, c->ev[i][1], c->ev[i][2]));
        *trace_ms += a; *accumulate_ms += b;
    }
    return PTK_OK;
}

int ptk_set_option(ptk_ctx* c, const char* name, double value)
{
    if (!c || !name) return PTK_ERR_BAD_ARG;
    if (!std::strcmp(name, "chunk"))
    {
        if (!(value >= 0 && value <= 4096)) return fail(c, PTK_ERR_BAD_ARG, "chunk must be in [0, 4096] (0 = automatic)");
        c->opt_chunk = (int)value;
        return PTK_OK;
    }
    if (!std::strcmp(name, "shade_threshold") || !std::strcmp(name, "gen_threshold"))
    {
        if (!(value >= 0 && value <= 4096)) return fail(c, PTK_ERR_BAD_ARG, "threshold (lambda in eighths) must be in [0, 4096]");
        (name[0] == 's' ? c->opt_shade_thr : c->opt_gen_thr) = (int)value;
        return PTK_OK;
    }
    if (!std::strcmp(name, "overlap"))
    {
        c->opt_overlap = value != 0.0 ? 1 : 0;
        return PTK_OK;
    }
    if (!std::strcmp(name, "comm_timeout_s"))
    {
        // bound of every wait of the exchange step (ptk_comm_init, ptk_gather_wait, ptk_read_gathered)
        if (!(value >= 0.05 && value <= 86400.0)) return fail(c, PTK_ERR_BAD_ARG, "comm_timeout_s must be in [0.05, 86400]");
        c->opt_comm_timeout_s = value;
        return PTK_OK;
    }
    if (!std::strcmp(name, "lens_cull"))
    {
        // 0: uncached cameras trace every pixel of the frame (the cull is exact: same image either way; tests / A-B only)
        c->opt_lens_cull = value != 0.0 ? 1 : 0;
        return PTK_OK;
    }
    if (!std::strcmp(name, "register_out_image"))
    {
        // 1: ptk_bind_out_image page-locks a pageable caller buffer in place (zero-copy hand-off; the caller unbinds before freeing it)
        c->opt_register_out = value != 0.0 ? 1 : 0;
        return PTK_OK;
    }
    if (!std::strcmp(name, "bvh_leaf_max") || !std::strcmp(name, "bvh_trav_cost") || !std::strcmp(name, "bvh_verbose"))
    {
        // builder tuning (process-wide, takes effect at the next ptk_upload_scene; 0 = the builders' own choices)
        if (!(value >= 0 && value <= 64)) return fail(c, PTK_ERR_BAD_ARG, "builder tuning value must be in [0, 64]");
        if (name[4] == 'l') g_bvh_tuning.leaf_max = (int)value; else if (name[4] == 't') g_bvh_tuning.trav_cost = (float)value; else g_bvh_tuning.verbose = (int)value;
        return PTK_OK;
    }
    if (!std::strcmp(name, "contract"))
    {
        if (!(value == 0.0 || value == 1.0 || value == 2.0)) return fail(c, PTK_ERR_BAD_ARG, "contract must be 0 (bit-exact), 1 (fused multiply-adds) or 2 (... and 1-ulp reciprocal / square root)");
        c->opt_contract = (int)value;
        return PTK_OK;
    }
    if (!std::strcmp(name, "generations"))
    {
        if (!(value >= 0 && value <= 64)) return fail(c, PTK_ERR_BAD_ARG, "generations must be in [0, 64] (0 = automatic)");
        c->opt_generations = (int)value;
        return PTK_OK;
    }
    if (!std::strcmp(name, "persistent"))
    {
        if (!(value >= -1 && value <= 1)) return fail(c, PTK_ERR_BAD_ARG, "persistent must be -1 (automatic), 0 or 1");
        c->opt_persistent = (int)value;
        return PTK_OK;
    }
    if (!std::strcmp(name, "max_batch"))
    {
        if (!(value >= 1 && value <= 64)) return fail(c, PTK_ERR_BAD_ARG, "max_batch must be in [1, 64]");
        c->opt_max_batch = (int)value;
        return PTK_OK;
    }
    if (!std::strcmp(name, "tri_threshold"))
    {
        if (!(value >= 0 && value <= 4096)) return fail(c, PTK_ERR_BAD_ARG, "tri_threshold (eighths) must be in [0, 4096]");
        c->opt_tri_thr = (int)value;
        return PTK_OK;
    }
    if (!std::strcmp(name, "flat_shade_weight") || !std::strcmp(name, "flat_gen_weight"))
    {
        if (!(value >= 1 && value <= 4096)) return fail(c, PTK_ERR_BAD_ARG, "weight (eighths) must be in [1, 4096]");
        (name[5] == 's' ? c->opt_flat_shade_w : c->opt_flat_gen_w) = (int)value;
        return PTK_OK;
    }
    if (!std::strcmp(name, "flat"))
    {
        c->opt_flat = value != 0.0;
        return PTK_OK;
    }
    if (!std::strcmp(name, "device_build"))
    {
        c->opt_device_build = value != 0.0;
        return PTK_OK;
    }
    if (!std::strcmp(name, "primary_cache"))
    {
        c->opt_primary_cache = value != 0.0;
        return PTK_OK;
    }
    if (!std::strcmp(name, "pass_bytes"))
    {
        // upper bound: the kernel indexes the sample buffer (16-byte entries) with 32 bits
        if (!(value >= 1 << 20 && value <= 32.0 * 1073741824.0)) return fail(c, PTK_ERR_BAD_ARG, "pass_bytes must be in [1 MiB, 32 GiB]");
        c->opt_pass_bytes = (size_t)value;
        return PTK_OK;
    }
    return fail(c, PTK_ERR_BAD_ARG, std::string("unknown option ") + name);
}

int ptk_bvh_info(ptk_ctx* c, int32_t* num_nodes, int32_t* depth, int32_t* num_leaf_tris)
{
    if (!c) return PTK_ERR_BAD_ARG;
    if (num_nodes) *num_nodes = c->num_nodes;
    if (depth) *depth = c->bvh_depth;
    if (num_leaf_tris) *num_leaf_tris = c->num_leaf_tris;
    return PTK_OK;
}

int ptk_upload_timing(ptk_ctx* c, double* ms4, int* built_on_device)
{
    if (!c || !ms4) return PTK_ERR_BAD_ARG;
    for (int k = 0; k < 4; k++) ms4[k] = c->upload_ms[k];
    if (built_on_device) *built_on_device = c->built_on_device ? 1 : 0;
    return PTK_OK;
}

int ptk_bvh_layout(ptk_ctx* c, int32_t* node_width, int32_t* node_bytes, int32_t* stack_need)
{
    if (!c) return PTK_ERR_BAD_ARG;
    if (node_width) *node_width = 4;
    if (node_bytes) *node_bytes = NODE_F4 * 16;
    if (stack_need) *stack_need = c->bvh_stack;
    return PTK_OK;
}

int ptk_download_bvh(ptk_ctx* c, float* nodes16, int32_t* leaf_order)
{
    if (!c) return PTK_ERR_BAD_ARG;
    if (!c->have_scene) return fail(c, PTK_ERR_BAD_ARG, "ptk_upload_scene has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (nodes16 && c->num_nodes > 0) HIPCHK(c, hipMemcpy(nodes16, c->d_nodes, (size_t)c->num_nodes * NODE_F4 * sizeof(float4), hipMemcpyDeviceToHost));
    if (leaf_order && c->num_tris > 0)
    {
        std::vector<float> t((size_t)c->num_tris * TRI_F4 * 4);
        HIPCHK(c, hipMemcpy(t.data(), c->d_tris, t.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int32_t k = 0; k < c->num_tris; k++) std::memcpy(&leaf_order[k], &t[(size_t)k * TRI_F4 * 4 + 9], 4);
    }
    return PTK_OK;
}

int ptk_probe_hits(ptk_ctx* c, int n, const float* ro, const float* rd, int32_t* tri, float* tuv)
{
    if (!c || n < 0 || (n > 0 && (!ro || !rd || !tri || !tuv))) return PTK_ERR_BAD_ARG;
    if (!c->have_scene) return fail(c, PTK_ERR_BAD_ARG, "ptk_upload_scene has not been called");
    if (n == 0) return PTK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    float *d_ro = nullptr, *d_rd = nullptr, *d_tuv = nullptr; int32_t* d_tri = nullptr;
    size_t b3 = (size_t)n * 3 * sizeof(float);
    hipError_t e = hipMalloc(&d_ro, b3);
    if (e == hipSuccess) e = hipMalloc(&d_rd, b3);
    if (e == hipSuccess) e = hipMalloc(&d_tuv, b3);
    if (e == hipSuccess) e = hipMalloc(&d_tri, (size_t)n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d_ro, ro, b3, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rd, rd, b3, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
    {
        ProbeParams p = {};
        p.nodes = c->d_nodes; p.tris = c->d_tris; p.shade = c->d_shade; p.mats = c->d_mats;
        p.texinfo = c->d_texinfo; p.texels = c->d_texels; p.ro = d_ro; p.rd = d_rd; p.tri = d_tri; p.tuv = d_tuv;
        p.n = n; p.num_nodes = c->num_nodes; p.scene_bound = c->scene_bound;
        launch_probe(p, c->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(tri, d_tri, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tuv, d_tuv, b3, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_ro); (void)hipFree(d_rd); (void)hipFree(d_tuv); (void)hipFree(d_tri);
    if (e != hipSuccess) return fail(c, PTK_ERR_HIP, hipGetErrorString(e));
    return PTK_OK;
}

int ptk_probe_direct(ptk_ctx* c, int n, const float* pts, const float* normals, const float* diffuse, const float* tape3, float* out3)
{
    if (!c || n < 0 || (n > 0 && (!pts || !normals || !diffuse || !tape3 || !out3))) return PTK_ERR_BAD_ARG;
    if (!c->have_scene) return fail(c, PTK_ERR_BAD_ARG, "ptk_upload_scene has not been called");
    if (n == 0) return PTK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t b3 = (size_t)n * 3 * sizeof(float);
    float* d[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    const float* src[4] = { pts, normals, diffuse, tape3 };
    hipError_t e = hipSuccess;
    for (int k = 0; k < 5 && e == hipSuccess; k++) e = hipMalloc(&d[k], b3);
    for (int k = 0; k < 4 && e == hipSuccess; k++) e = hipMemcpyAsync(d[k], src[k], b3, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
    {
        ProbeParams p = {};
        p.nodes = c->d_nodes; p.tris = c->d_tris; p.shade = c->d_shade; p.mats = c->d_mats;
        p.texinfo = c->d_texinfo; p.texels = c->d_texels; p.lights = c->d_lights; p.num_lights = c->num_lights;
        p.n = n; p.num_nodes = c->num_nodes; p.scene_bound = c->scene_bound;
        launch_probe_direct(p, d[0], d[1], d[2], d[3], d[4], c->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out3, d[4], b3, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    for (int k = 0; k < 5; k++) (void)hipFree(d[k]);
    if (e != hipSuccess) return fail(c, PTK_ERR_HIP, hipGetErrorString(e));
    return PTK_OK;
}

int ptk_probe_math(ptk_ctx* c, int op, int n, const float* in, float* out)
{
    if (!c || op < 0 || op > 3 || n < 0 || (n > 0 && (!in || !out))) return PTK_ERR_BAD_ARG;
    if (n == 0) return PTK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    float *d_in = nullptr, *d_out = nullptr;
    const size_t bytes = (size_t)n * sizeof(float);
    hipError_t e = hipMalloc(&d_in, bytes);
    if (e == hipSuccess) e = hipMalloc(&d_out, bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, in, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { launch_probe_math(op, d_in, d_out, n, c->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_in); (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, PTK_ERR_HIP, hipGetErrorString(e));
    return PTK_OK;
}

int ptk_probe_primary_dirs(ptk_ctx* c, float* host_out)
{
    if (!c || !host_out) return PTK_ERR_BAD_ARG;
    if (!c->d_primary) return fail(c, PTK_ERR_BAD_ARG, "ptk_set_frame has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_primary(c);
    if (rc != PTK_OK) return rc;
    size_t px = (size_t)c->width * c->height;
    std::vector<float4> tmp(px);
    HIPCHK(c, hipMemcpyAsync(tmp.data(), c->d_primary, px * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < px; i++) { host_out[i * 3] = tmp[i].x; host_out[i * 3 + 1] = tmp[i].y; host_out[i * 3 + 2] = tmp[i].z; }
    return PTK_OK;
}

}  // extern "C"
