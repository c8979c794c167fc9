// Host orchestration of the mesher kernels (temporary stub until mesh_kernels.hip lands).
#include "host_ctx.hpp"
int mesh_alloc(immesh_ctx* c) { (void)c; return 0; }
void mesh_free(immesh_ctx* c) { (void)c; }
int mesh_scan_device(immesh_ctx* c, const float*, int, const double*, int) { c->err = "mesher not built yet"; return IMMESH_E_INVAL; }
int mesh_transform_full(immesh_ctx* c, const float*, int, const imh::State&) { c->err = "mesher not built yet"; return IMMESH_E_INVAL; }
void mesh_counters(immesh_ctx*, immesh_counters_t*) {}
extern "C" {
int immesh_mesh_scan(immesh_ctx* c, const float*, int32_t, const double*, int32_t) { if (c) c->err = "mesher not built yet"; return IMMESH_E_INVAL; }
int immesh_mesh_sizes(immesh_ctx* c, immesh_mesh_sizes_t*) { if (c) c->err = "mesher not built yet"; return IMMESH_E_INVAL; }
int immesh_mesh_fetch(immesh_ctx* c, float*, int32_t*, uint8_t*, int32_t*, int32_t*, uint8_t*, int32_t*, double*) { if (c) c->err = "mesher not built yet"; return IMMESH_E_INVAL; }
}
