// TEMPORARY STUB (replaced by the real TSDF engine in the next commit).
#include "dr_common.h"
struct drf_s { int dummy; };
#define UNSUP(name) return dr::guarded([&] { dr::fail(DR_ERR_UNSUPPORTED, name ": not implemented yet"); })
extern "C" {
int drf_create(const drf_options_t *, int, drf_t **) { UNSUP("drf_create"); }
void drf_destroy(drf_t *) {}
int drf_integrate_scan_async(drf_t *, const uint8_t *, const float *, const float *) { UNSUP("drf_integrate_scan_async"); }
int drf_render_async(drf_t *, const float *const *, int) { UNSUP("drf_render_async"); }
int drf_get_render_result(drf_t *, uint8_t **, float **, int) { UNSUP("drf_get_render_result"); }
int drf_extract_mesh_async(drf_t *, const float *, const float *) { UNSUP("drf_extract_mesh_async"); }
int drf_get_mesh_sync(drf_t *, size_t, size_t *, float *, float *) { UNSUP("drf_get_mesh_sync"); }
int drf_save_mesh(drf_t *, const char *, const float *, const float *) { UNSUP("drf_save_mesh"); }
int drf_synchronize(drf_t *) { UNSUP("drf_synchronize"); }
int drf_stats(drf_t *, uint64_t *) { UNSUP("drf_stats"); }
int drf_export_blocks(drf_t *, int, int32_t *, uint8_t *, int *) { UNSUP("drf_export_blocks"); }
int drf_integrate_device(drf_t *, const void *, const void *, const float *) { UNSUP("drf_integrate_device"); }
int dr_device_alloc(int, size_t, void **) { UNSUP("dr_device_alloc"); }
int dr_device_free(void *) { UNSUP("dr_device_free"); }
int dr_memcpy_h2d(void *, const void *, size_t) { UNSUP("dr_memcpy_h2d"); }
int dr_memcpy_d2h(void *, const void *, size_t) { UNSUP("dr_memcpy_d2h"); }
int drf_bench_integrate(drf_t *, const void *, const void *, const float *, int, float *, float *) { UNSUP("drf_bench_integrate"); }
}
