/* ptk_host — C wrapper over the C++ host layer (include/pathtracer.h + the headless scene layer), so the
 * reference-compatible `PathTracer` class can be driven from C, Python (ctypes) or any other FFI.
 * One pth_tracer = one PathTracer instance = one GPU.  Functions mirror the class's methods one to
 * one (reference PathTracing/src/pathtracer.h:100-130); like them they return nothing and ignore bad
 * ids — pth_last_error() exposes the device layer's last error text.
 */
#ifndef PTK_HOST_H
#define PTK_HOST_H

#include <stdint.h>

#include "ptk.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_tracer pth_tracer;

pth_tracer* pth_create(int device_ordinal);               /* no GPU work until the first BuildBVH / render */
void pth_destroy(pth_tracer* t);

/* PathTracer API */
void pth_load_object(pth_tracer* t, const char* file, const float model_colmajor[16]);
void pth_set_material(pth_tracer* t, int obj, int elem, const float m15[15]);  /* type, diffuse3, specular3, emissive3, I, roughness, reflectiveness, translucency, ior */
void pth_set_texture(pth_tracer* t, int obj, int elem, int slot, const char* file); /* slot 0..5: diffuse normal emissive roughness metallic opacity */
void pth_build_bvh(pth_tracer* t);
void pth_reset_image(pth_tracer* t);
void pth_clear_scene(pth_tracer* t);
int  pth_get_samples(pth_tracer* t);
int  pth_get_triangle_count(pth_tracer* t);
int  pth_get_trace_depth(pth_tracer* t);
void pth_set_trace_depth(pth_tracer* t, int depth);
void pth_set_out_image(pth_tracer* t, uint8_t* out);      /* caller-owned W*H*3 buffer (may be NULL) */
void pth_set_out_gl_buffer(pth_tracer* t, unsigned int gl_buffer);   /* SetOutGLBuffer (extension): 0 switches back */
void pth_set_out_device_image(pth_tracer* t, void* device_rgb8);      /* SetOutDeviceImage (extension): NULL switches back */
void pth_set_resolution(pth_tracer* t, int w, int h);
void pth_get_resolution(pth_tracer* t, int* w, int* h);
int  pth_num_objects(pth_tracer* t);
int  pth_num_elements(pth_tracer* t, int obj);
int  pth_name(pth_tracer* t, int obj, int elem, char* out, int cap);   /* name of object `obj` (elem < 0) or of its element; returns its length, -1 if there is none */
void pth_set_camera(pth_tracer* t, const float pos[3], const float dir[3], const float up[3]);
void pth_set_projection(pth_tracer* t, float f, float fovy);
void pth_set_focal_dist(pth_tracer* t, float d);
void pth_set_aperture(pth_tracer* t, float a);
void pth_render_frame(pth_tracer* t);
void pth_exit(pth_tracer* t);

/* extensions */
void pth_set_seed(pth_tracer* t, uint64_t seed);
void pth_set_tile(pth_tracer* t, int rank, int world);
void pth_render_frames(pth_tracer* t, int count);
int  pth_read_accum(pth_tracer* t, float* out);           /* 1 on success */
const char* pth_last_error(pth_tracer* t);
ptk_ctx* pth_context(pth_tracer* t);
const ptk_scene_desc* pth_staged_scene(pth_tracer* t);    /* flat arrays of the staged scene (host only) */

/* headless scene layer: .pts file -> PathTracer (LoadScene + SendObjectsToPathTracer + SetPathTracerCamera) */
int  pth_load_scene_file(pth_tracer* t, const char* pts_path);   /* 0 ok, <0 parse error (text via pth_last_error) */
int  pth_pts_roundtrip(const char* in_path, const char* out_path); /* read_pts + write_pts, 0 ok */

/* ExportAt (main.cpp:760-771): write the bottom-up RGB8 hand-off buffer as a top-down PNG; 1 on success */
int  pth_export_png(const char* path, const uint8_t* rgb8_bottom_up, int w, int h);

/* host-only probes (no GPU): glm-0.9.3.1-compatible TRS / Euler camera, Triangle::Init, Image */
void pth_trs_matrix(const float loc[3], const float rot_deg[3], const float scl[3], float out16[16]);
void pth_euler_camera(const float rot_deg[3], float out6[6]);
void pth_triangle_init(const float in15[15], float out9[9]);
int  pth_image_load(const char* file, int* w, int* h);    /* loads into a process-wide scratch image; 1 ok */
void pth_image_data(uint8_t* out_rgba);
void pth_image_tex2d(float u, float v, float out4[4]);

#ifdef __cplusplus
}
#endif
#endif
