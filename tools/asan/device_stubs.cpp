// the device layer, absent: every call fails the way a box without a GPU makes it fail
#include "ptk.h"
extern "C" {
int ptk_create(ptk_ctx** out, int) { if (out) *out = 0; return PTK_ERR_HIP; }
void ptk_destroy(ptk_ctx*) {}
const char* ptk_last_error(ptk_ctx*) { return "stub"; }
int ptk_bind_gl_buffer(ptk_ctx*, unsigned int) { return PTK_ERR_HIP; }
int ptk_bind_out_device(ptk_ctx*, void*) { return PTK_ERR_HIP; }
int ptk_bind_out_image(ptk_ctx*, uint8_t*) { return PTK_ERR_HIP; }
int ptk_read_accum(ptk_ctx*, float*) { return PTK_ERR_HIP; }
int ptk_render(ptk_ctx*, uint32_t, uint32_t, uint64_t) { return PTK_ERR_HIP; }
int ptk_request_exit(ptk_ctx*) { return PTK_ERR_HIP; }
int ptk_reset(ptk_ctx*) { return PTK_ERR_HIP; }
int ptk_resolve_rgb8(ptk_ctx*, uint8_t*) { return PTK_ERR_HIP; }
int ptk_samples(ptk_ctx*) { return 0; }
int ptk_set_camera(ptk_ctx*, const float*, const float*, const float*, float, float, float, float) { return PTK_ERR_HIP; }
int ptk_set_frame(ptk_ctx*, int, int, int) { return PTK_ERR_HIP; }
int ptk_set_tile(ptk_ctx*, int, int) { return PTK_ERR_HIP; }
int ptk_synchronize(ptk_ctx*) { return PTK_ERR_HIP; }
int ptk_update_materials(ptk_ctx*, int32_t, const ptk_material*) { return PTK_ERR_HIP; }
int ptk_upload_scene(ptk_ctx*, const ptk_scene_desc*) { return PTK_ERR_HIP; }
}
