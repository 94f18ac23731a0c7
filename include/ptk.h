/* ptk — "path-trace kernels": the C-ABI drop-in boundary of the MI355X render path.
 *
 * This is what a binding in the reference application would call instead of the body of
 * PathTracer::RenderFrame() and friends (reference PathTracing/src/pathtracer.h:100-130,
 * pathtracer.cpp:260-365, :741-822).  Plain C: opaque context, plain pointers and sizes, int status
 * codes; no C++/torch types cross it.  One context drives one GPU on one HIP stream.
 *
 * Call order (mirrors the reference's LoadObject / Set... -> BuildBVH -> SetResolution -> SetOutImage ->
 * ResetImage -> RenderFrame()xN contract):
 *   ptk_create -> ptk_upload_scene -> ptk_set_camera -> ptk_set_frame -> ptk_reset ->
 *   ptk_render(first, n, seed) ... -> ptk_resolve_rgb8 / ptk_read_accum -> ptk_destroy
 *
 * Every function returns PTK_OK (0) or a negative error class; ptk_last_error() gives the text.
 * Nothing throws across this boundary.  Host pointers are borrowed for the duration of the call.
 */
#ifndef PTK_H
#define PTK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTK_OK 0
#define PTK_ERR_BAD_ARG (-1)     /* null / out-of-range argument, call-order violation */
#define PTK_ERR_HIP (-2)         /* a HIP runtime call failed (no device, out of memory, ...) */
#define PTK_ERR_RCCL (-3)        /* an RCCL call failed */
#define PTK_ERR_LIMIT (-4)       /* scene exceeds a kernel limit (BVH depth, index width) */

#define PTK_TILE 16              /* pixel tile edge: one 256-thread block = one 16x16 tile */
#ifndef PTK_MAX_BVH_DEPTH
#define PTK_MAX_BVH_DEPTH 32     /* entries of the per-lane LDS traversal stack (bounds what the 4-wide tree may defer at once) */
#endif

/* replaces Material (mesh.h:21-59) with texture *indices* instead of Image pointers */
typedef struct ptk_material {
    int32_t type;                /* 0 OPAQUE, 1 TRANSLUCENT (mesh.h:15-19) */
    float diffuse[3];
    float specular[3];
    float emissive[3];
    float emissive_intensity;
    float roughness;
    float reflectiveness;
    float translucency;
    float ior;
    int32_t tex[6];              /* diffuse, normal, emissive, roughness, metallic, opacity; -1 = none */
} ptk_material;                  /* 84 bytes, packed the same as the natural layout */

/* replaces Image (image.h:7-28): RGBA8 texels live in one atlas */
typedef struct ptk_texture {
    int32_t width, height;
    int64_t offset;              /* byte offset of this image's first texel in `texels` */
} ptk_texture;

/* replaces std::vector<Triangle> mTriangles + mLoadedObjects materials + mLights
 * (pathtracer.h:51-56; Triangle = mesh.h:71-96) as flat arrays */
typedef struct ptk_scene_desc {
    int32_t num_triangles;
    const float* verts;          /* [N][9]  v1 v2 v3, world space */
    const float* normals;        /* [N][9]  n1 n2 n3 */
    const float* uvs;            /* [N][6]  uv1 uv2 uv3 */
    const float* tbn;            /* [N][9]  normal, tangent, bitangent (Triangle::Init, mesh.cpp:61-83) */
    const uint8_t* smoothing;    /* [N] */
    const int32_t* material;     /* [N] index into materials */
    int32_t num_materials;
    const ptk_material* materials;
    int32_t num_textures;
    const ptk_texture* textures;
    const uint8_t* texels;
    int64_t texel_bytes;
    int32_t num_lights;
    const int32_t* lights;       /* triangles whose material has |emissive| >= 1e-5 (pathtracer.cpp:267-273) */
} ptk_scene_desc;

/* traversal statistics from the counters-enabled (untimed) kernel variant; feeds the
 * algorithmic-bytes roofline of SURVEY.md §8(d4) */
typedef struct ptk_stats {
    uint64_t samples;            /* pixel*spp processed */
    uint64_t rays;               /* closest-hit traversals (bounce + shadow rays) */
    uint64_t shadow_rays;
    uint64_t node_visits;        /* 64-byte BVH node records fetched (4 child boxes each) */
    uint64_t tri_tests;          /* 48-byte triangle records fetched */
    uint64_t hits_shaded;        /* surface interactions shaded */
    uint64_t tex_fetches;        /* 4-byte texel fetches */
    /* SIMD utilisation of the wave state machine: lanes that took part / (64 * wave-level executions) */
    uint64_t walk_wave_iters, walk_lane_iters;     /* BVH-walk loop iterations per wave, and lanes active in them */
    uint64_t shade_wave_execs, shade_lanes;        /* shading block executions per wave, lanes shaded */
    uint64_t gen_wave_execs, gen_lanes;            /* camera-ray block executions per wave, lanes generated */
    uint64_t tri_wave_execs, tri_lanes;            /* triangle-arm executions of the BVH walk per wave, lanes testing */
    uint64_t max_walk_nodes;                       /* most node records a single ray fetched (tail diagnostic) */
    uint64_t paths_started;                        /* samples actually traced (the others: camera ray known to miss) */
} ptk_stats;

typedef struct ptk_ctx ptk_ctx;

/* ctor/dtor of the device side of PathTracer (pathtracer.cpp:11-39) */
int  ptk_create(ptk_ctx** out, int device_ordinal);
void ptk_destroy(ptk_ctx* ctx);

/* replaces BuildBVH (pathtracer.cpp:260-274, mesh.cpp:169-211): stages the scene, builds the BVH (own builders; closest
 * hit is tree-independent) and leaves everything resident in HBM.  Scenes of >= 4096 triangles are built ON THE GPU - binned
 * SAH level by level, collapse to the 4-wide quantised nodes, record packing (csrc/bvh_device.hip); smaller ones, and any
 * scene whose device-built tree would not fit the traversal stack, by the host builder (option "device_build" = 0 forces it).
 * Limits (PTK_ERR_LIMIT): at most 89 478 485 triangles (the walk addresses its 48-byte records with 32-bit byte offsets),
 * |coordinate| < 2^61, a tree that defers at most PTK_MAX_BVH_DEPTH entries (the host builder always meets that). */
int ptk_upload_scene(ptk_ctx* ctx, const ptk_scene_desc* scene);

/* SetMaterial after BuildBVH (pathtracer.cpp:243-258): the reference's triangles point into the loaded materials, so an
 * edited material is used by the next RenderFrame() without a rebuild, while the light list stays as BuildBVH collected
 * it.  Rewrites the material table and the lights' colours in place; counts and texture bindings must be the uploaded ones. */
int ptk_update_materials(ptk_ctx* ctx, int32_t num_materials, const ptk_material* materials);

/* SetCamera + SetProjection + SetCameraFocalDist + SetCameraAperture (pathtracer.cpp:333-360).
 * dir/up are normalised and focal/fovy clamped exactly as the reference setters do. */
int ptk_set_camera(ptk_ctx* ctx, const float pos[3], const float dir[3], const float up[3],
                   float focal, float fovy_deg, float focal_dist, float aperture);

/* SetResolution + SetTraceDepth (pathtracer.cpp:302-306, :328-331); (re)allocates the float
 * accumulator (mTotalImg) and the RGB8 image on the device */
int ptk_set_frame(ptk_ctx* ctx, int width, int height, int max_depth);

/* multi-GPU: this context renders only the 16x16 pixel tiles it owns: tile (tx, ty) belongs to rank
 * (ty*tiles_x + (tx + 3*ty) % tiles_x) % world (round-robin, each tile row rotated by 3 so that a
 * rank's tiles form diagonals and every rank gets an equal share of every image region) */
int ptk_set_tile(ptk_ctx* ctx, int rank, int world);

/* ResetImage (pathtracer.cpp:276-279, :745-751): zero the accumulator and the sample count */
int ptk_reset(ptk_ctx* ctx);

/* RenderFrame (pathtracer.cpp:741-817) x spp_count: adds samples [first_sample, first_sample +
 * spp_count) to every owned pixel and refreshes the device RGB8 image.  Asynchronous on the
 * context's stream.  The RNG is keyed on (seed, pixel, sample index): results do not depend on how
 * samples are batched into calls or on the GPU count. */
int ptk_render(ptk_ctx* ctx, uint32_t first_sample, uint32_t spp_count, uint64_t seed);

/* the host-buffer hand-off of mOutImg (pathtracer.cpp:802-812, main.cpp:3026-3029):
 * W*H*3 bytes, RGB, rows bottom-up, tightly packed; waits for the stream */
int ptk_resolve_rgb8(ptk_ctx* ctx, uint8_t* host_out);
/* SetOutImage (pathtracer.cpp:297-300) for the interactive loop - one RenderFrame(), one glTexSubImage2D(texData)
 * (main.cpp:3587, :3026-3029): binds a W*H*3 hand-off buffer THE GPU CAN WRITE - memory from ptk_host_alloc, or memory the caller
 * page-locked and mapped itself - so that the accumulate kernel's 8-bit resolve (pathtracer.cpp:802-812) is written STRAIGHT into
 * it over PCIe, and ptk_resolve_rgb8 into the bound buffer only waits for the stream: no copy command, no second launch.
 * Ordinary pageable memory (`new GLubyte[w*h*3]`, main.cpp:3435) is NOT bound and never registered with the runtime: the call
 * succeeds, and ptk_resolve_rgb8 copies the frame into it as before.  (Round 3 page-locked such a buffer in place; the reference's
 * viewer deletes texData BEFORE handing over its successor - InitializeFrame, main.cpp:3433-3445 - and at exit without telling
 * the tracer - OnExit, :3622 - and memory freed while registered poisons what the allocator puts there next.)
 * ptk_set_option("register_out_image", 1) opts back in to page-locking in place, for callers that promise to unbind - NULL, or
 * the successor - BEFORE they free a bound buffer.  NULL unbinds; ptk_set_frame with another resolution unbinds too (the caller
 * reallocates texData then, main.cpp:3425-3446).  A bound buffer stays caller-owned and must stay allocated while a render is
 * in flight.  While a buffer is bound, renders run on the context's stream alone (each frame is waited for anyway).
 * Threading: like every call that takes the context, from one thread at a time (the thread that renders); the C++ class's
 * SetOutImage is a pointer store for that reason and binds on the render thread. */
int ptk_bind_out_image(ptk_ctx* ctx, uint8_t* host_out);
/* The same hand-off without the PCIe hop, for a display path that lives on the GPU (N3: the viewer's frameTex ← texData upload,
 * main.cpp:3026-3029, :3425-3446).  The accumulate kernel writes the 8-bit image - W*H*3 bytes, RGB, rows bottom-up, the layout
 * of texData - into the bound DEVICE memory; ptk_synchronize waits for it.  One binding at a time: binding a host buffer, a
 * device buffer or an OpenGL buffer replaces whatever was bound; NULL / 0 unbinds; ptk_set_frame with another resolution unbinds.
 *   ptk_bind_out_device: any allocation of this context's GPU (hipMalloc, a torch tensor, imported external memory).
 *   ptk_bind_gl_buffer:  EXPERIMENTAL - an OpenGL buffer object of >= W*H*3 bytes (the viewer's pixel-unpack buffer), registered
 *     with hipGraphicsGLRegisterBuffer HERE, on the calling thread, which must be the one whose OpenGL context is current (the
 *     viewer's GUI thread; without a current context the call fails with PTK_ERR_BAD_ARG - the library does not link OpenGL and
 *     never creates a context); 0 lets the registration go, on the same thread.  ptk_render - on the render thread, which needs no
 *     context - maps the buffer on its stream before the first pass and unmaps it behind the kernel that writes it: HIP owns the
 *     buffer for the length of a ptk_render, so the GUI thread may source a glTexSubImage2D from it only BETWEEN renders (the same
 *     sequencing texData needs).  A new resolution keeps the registration; the buffer's size is checked at every map.  No frame
 *     needs to be set when this is called.
 *     NOT exercised on hardware: the build boxes are headless (no display server, no EGL); only the error paths are tested. */
int ptk_bind_out_device(ptk_ctx* ctx, void* device_rgb8);
int ptk_bind_gl_buffer(ptk_ctx* ctx, unsigned int gl_buffer);
/* Page-locked host memory for the hand-off buffer (what `new GLubyte[w*h*3]` is in main.cpp:3435): a
 * ptk_resolve_rgb8 into it is one DMA transfer instead of a staged copy.  Caller-owned, like texData:
 * release with ptk_host_free before the context that allocated it is destroyed or after - either order. */
void* ptk_host_alloc(size_t bytes);
void  ptk_host_free(void* p);
/* mTotalImg: W*H*3 floats, rows bottom-up; waits for the stream */
int ptk_read_accum(ptk_ctx* ctx, float* host_out);
int ptk_write_accum(ptk_ctx* ctx, const float* host_in, int samples);   /* resume from a saved accumulator */

int ptk_samples(ptk_ctx* ctx);         /* GetSamples (pathtracer.cpp:362-365); thread-safe */
/* Exit (pathtracer.cpp:819-822); thread-safe.  Cuts EVERY render in flight - ptk_render is asynchronous while no output
 * image is bound, so several may be queued: all of them, not only the newest - : passes whose kernels have not started are
 * skipped whole (an aborted pass adds nothing to the accumulator), the sample count still advances as mSamples does;
 * renders issued after the call are not affected, as RenderFrame() resets mExit on entry (pathtracer.cpp:742) */
int ptk_request_exit(ptk_ctx* ctx);
int ptk_synchronize(ptk_ctx* ctx);
const char* ptk_last_error(ptk_ctx* ctx);

/* device-resident hand-off (no host copy): the accumulator's device address, for a caller that
 * gathers it with its own collective (torch.distributed / RCCL) or maps it into a GL texture */
int ptk_accum_device_ptr(ptk_ctx* ctx, void** dev_ptr, size_t* bytes);
int ptk_rgb8_device_ptr(ptk_ctx* ctx, void** dev_ptr, size_t* bytes);
/* render into a caller-owned device accumulator (W*H*3 floats) instead of the internal one */
int ptk_bind_accum(ptk_ctx* ctx, void* dev_ptr);
int ptk_set_stream(ptk_ctx* ctx, void* hip_stream);

/* ---- multi-GPU exchange step (no counterpart in the reference, which renders on one CPU: SURVEY.md 8e) -------------
 * One process per GPU; the frame is tile-split with ptk_set_tile and each rank accumulates only its owned tiles.
 * The exchange is a PACKED GATHER: every rank sends its owned tiles (1/world of the float accumulator) straight to the
 * root with grouped ncclSend / ncclRecv, the root scatters them into a full image of its own (the local accumulators
 * are untouched and keep accumulating).  Pure copies: the gathered image is bit-identical to a single-GPU render.
 *
 * Packed layout of rank r: its owned tiles in ascending tile order, 768 floats per tile = 16 x 16 pixels row-major from
 * the tile's top-left corner x RGB; pixels beyond the image edge hold 0.  ptk_packed_layout writes, for every packed
 * float, its index in the accumulator (W*H*3, rows bottom-up) or -1 for padding - host-only, needs no GPU. */
int64_t ptk_packed_floats(int width, int height, int rank, int world);
int ptk_packed_layout(int width, int height, int rank, int world, int64_t* src_index /* [ptk_packed_floats] */);

/* the context's own RCCL communicator: rank 0 makes the 128-byte id, every rank (after receiving it by whatever
 * rendezvous the host has - bench.py broadcasts it through torch.distributed) calls ptk_comm_init, which also sets
 * the tile split to (rank, world) */
int ptk_comm_unique_id(void* id_out /* 128 bytes */);
int ptk_comm_init(ptk_ctx* ctx, const void* id /* 128 bytes */, int rank, int world);
int ptk_comm_destroy(ptk_ctx* ctx);
/* what the context's communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) and the HIP
 * ordinal of the context: lets a benchmark line prove how many RCCL ranks the exchange really spans */
int ptk_comm_info(ptk_ctx* ctx, int* rank, int* world, int* comm_device, int* ctx_device);

/* Start the exchange of the accumulator as it is after everything queued on the context's stream so far.
 * rccl_comm: an ncclComm_t passed as void*, or NULL for the context's own (ptk_comm_init).  Asynchronous, on the
 * context's high-priority exchange stream: pack kernel -> grouped send/recv over xGMI -> (root) unpack kernel; the
 * render stream only waits for the pack (snapshot), so the next ptk_render overlaps the transfer. */
int ptk_gather_accum(ptk_ctx* ctx, void* rccl_comm, int root);
int ptk_gather_wait(ptk_ctx* ctx);                                    /* host waits for the last exchange */
int ptk_read_gathered(ptk_ctx* ctx, float* host_out);                 /* root: W*H*3 floats, rows bottom-up; waits */
int ptk_gathered_device_ptr(ptk_ctx* ctx, void** dev_ptr, size_t* bytes);
/* test hook: keeps the exchange stream busy for `milliseconds` (0 .. 10 000) ahead of whatever is queued on it next, so that the
 * bound of ptk_gather_wait ("comm_timeout_s") can be exercised on one GPU; the kernel behind it always ends by itself */
int ptk_debug_stall_exchange(ptk_ctx* ctx, int milliseconds);
/* parity probes of the pack / unpack kernels: one GPU can play every rank of a split */
int ptk_probe_pack(ptk_ctx* ctx, int rank, int world, float* host_out /* [ptk_packed_floats] */);
int ptk_probe_unpack(ptk_ctx* ctx, int world, const float* host_packed /* all ranks, back to back */, float* host_image);

/* tuning (none changes any result): "chunk" = samples per work item (default 0 = automatic: 8, or 4 when
 * the rank's share of the frame is small); "max_batch" = queue slots a persistent wave pops at once (default 1); "persistent" = -1/0/1: waves pull
 * work items from queues until none is left (1), or one item per wave (0); default -1 = by launch size; "generations" = persistent waves retire after 1/g of their share so that other
 * streams' kernels (the exchange step) get wave slots mid-launch (default 0 = 1 on one GPU, 2 when tile-split: measured cost 0-2 %); "overlap" = 0/1 (default 1): consecutive batches trace on two alternating internal streams so that a batch's
 * tail (its few longest paths) overlaps the next batch; results and stream ordering are unchanged;
 * "flat" = 0/1, scenes of <= 16 triangles skip the BVH walk and test
 * every triangle with scalar loads (default 1), "flat_shade_weight" / "flat_gen_weight" = its block-choice
 * weights in eighths (defaults 8 / 64); "pass_bytes" = HBM
 * budget of the sample buffer between the trace and accumulate kernels (default 16 GiB; there are two such buffers at most);
 * "shade_threshold" / "gen_threshold" = scheduling lambdas of the wave state machine in eighths
 * (cost of the shading / camera-ray block relative to one BVH walk step; defaults: shading 200 for
 * trees of depth <= 4, else 68, or 46 for trees of 131 072 nodes and more (0 = this automatic choice), camera rays 16);
 * "tri_threshold" = the
 * triangle arm of the BVH walk runs once the lanes holding a leaf reach this many eighths of the lanes
 * that can still walk (default 6; 0 = every iteration); "device_build" = 0/1 (default 1): build the BVH of scenes of >= 4096 triangles on the GPU;
 * "primary_cache" = 0/1, reuse the camera ray's closest
 * hit across samples when the camera is a pinhole and the scene has no opacity texture (default 1);
 * "lens_cull" = 0/1 (default 1): cameras whose rays are NOT cached (thin lens, opacity textures) leave out the pixels none of whose
 * lens rays can reach the scene's bounding box - exact: the image is the same bit for bit either way (tests / A-B only);
 * "comm_timeout_s" = bound in seconds of every wait on another rank (ptk_comm_init, ptk_gather_wait, ptk_read_gathered; default 120);
 * "register_out_image" = 0/1 (default 0): see ptk_bind_out_image;
 * "pass_bytes" is an upper bound: a render never asks for more than half of the device memory that is free (hipMemGetInfo), and
 * a pass whose sample buffer cannot be allocated is halved and tried again - more passes, the same image;
 * "bvh_leaf_max" (1..8), "bvh_trav_cost" (SAH cost of a node visit in triangle tests), "bvh_verbose" = builder tuning, process-wide,
 * effective at the next ptk_upload_scene (0 = the builders' own choices: 4 / 1.0 / quiet): they shape the tree, and closest hits do
 * not depend on the tree.
 * ONE option changes results, within the stated tolerance: "contract" = 0 (default: every kernel bit-identical to the CPU oracle),
 * 1 = the trace kernels built with -ffp-contract=fast (a * b + c fuses), 2 = ... and 1-ulp hardware reciprocal / square root / rsq:
 * per-channel RMSE of the mean image against the exact kernels <= 1e-3 (measured ~1e-5 .. 9e-5, tests/test_gpu_contract.py).
 * No environment variable reaches the library. */
int ptk_set_option(ptk_ctx* ctx, const char* name, double value);

/* measurement: HIP-event times of the last ptk_render's kernels (per pass, summed).  With the "overlap" option on and
 * several passes or back-to-back renders in flight, a trace kernel's time includes waiting for the wave slots its
 * predecessor's tail still holds: set "overlap" 0 for isolated per-launch durations (bench.py does) */
int ptk_last_render_ms(ptk_ctx* ctx, float* ms, int* launches);
int ptk_last_kernel_ms(ptk_ctx* ctx, float* trace_ms, float* accumulate_ms);
/* ... and of EVERY trace-kernel launch of a stretch of renders, overlapped or not, each timed with a pair of HIP events on the
 * stream the kernel is launched on (the events sit behind the launch's stream waits: a duration is the kernel's own, from the
 * moment it may start).  ptk_kernel_log(ctx, capacity) starts a log of up to `capacity` launches (0 stops and frees it);
 * ptk_kernel_log_read waits for the streams and returns the durations in launch order (bench.py: the per-launch times INSIDE
 * its timed region, beside ms_per_step). */
int ptk_kernel_log(ptk_ctx* ctx, int capacity);
int ptk_kernel_log_read(ptk_ctx* ctx, float* trace_ms, int max_entries, int* num_entries);
int ptk_collect_stats(ptk_ctx* ctx, uint32_t first_sample, uint32_t spp_count, uint64_t seed, ptk_stats* out);
int ptk_bvh_info(ptk_ctx* ctx, int32_t* num_nodes, int32_t* depth /* wide nodes on the longest chain */, int32_t* num_leaf_tris);
/* child boxes per node record (4), bytes per record (64), most stack entries a traversal of this tree can need */
/* host milliseconds of the last ptk_upload_scene: [0] BVH build, [1] record packing, [2] device allocation + copies, [3] total */
int ptk_upload_timing(ptk_ctx* ctx, double* ms4, int* built_on_device /* may be NULL */);
int ptk_bvh_layout(ptk_ctx* ctx, int32_t* node_width, int32_t* node_bytes, int32_t* stack_need);

/* the tree as it lies in HBM, for structural tests of the builders: num_nodes x 16 floats (ptk_device.h BVH4 record) and,
 * for every triangle record in leaf order, the scene triangle it holds */
int ptk_download_bvh(ptk_ctx* ctx, float* nodes16, int32_t* leaf_order);

/* probes used by the parity tests (same semantics as the kernels' device functions) */
int ptk_probe_hits(ptk_ctx* ctx, int n, const float* ro, const float* rd, int32_t* tri, float* tuv);
int ptk_probe_primary_dirs(ptk_ctx* ctx, float* host_out /* [H][W][3] top-down */);
/* DirectIllumimation (pathtracer.cpp:505-531) at n surface points with its three draws (light choice, the two of SampleTriangle)
 * on tape: light sample, n.l test, shadow walk and visibility rule exactly as the trace kernel applies them; out3 = the value
 * the reference's function returns (zero when unlit).  Inputs / output [n][3]. */
int ptk_probe_direct(ptk_ctx* ctx, int n, const float* points, const float* normals, const float* diffuse, const float* tape3, float* out3);
/* the kernels' exact-arithmetic helpers on an array: op 0 = the short reciprocal (valid for 2^-126 <= |a| <= 2^126), 1 = the
 * reciprocal with IEEE special cases, 2 = the short square root, 3 = 1 / sqrt(x) as the normalisations compute it.  Each must
 * return the bits of the IEEE-754 operation the reference's CPU code performs (1.0f / a, sqrtf(x)). */
int ptk_probe_math(ptk_ctx* ctx, int op, int n, const float* in, float* out);

#ifdef __cplusplus
}
#endif
#endif
