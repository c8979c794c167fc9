// TEST INFRASTRUCTURE ONLY: lets the drop-in shim be linked against the CPU oracle (oracle/liboracle.so, entry points renamed orc_* by -D macros in
// the Makefile) so that its marshalling / call order / host-mirror logic is exercised on a machine without a GPU (tests/test_dropin_cpu.py).  The
// three entry points below have no oracle counterpart.  Never part of the product.
#include <cstring>
#include "immesh_c_api.h"
extern "C" {
void immesh_default_config(immesh_config* c) {   // avia.yaml + mapping_avia.launch values (mirrors immesh_amd/csrc/c_api.cpp)
    std::memset(c, 0, sizeof(*c));
    c->voxel_size = 0.5; c->max_layer = 2;
    for (int i = 0; i < 5; i++) c->layer_init[i] = 5;
    c->max_points_size = 100; c->planer_threshold = 0.01; c->dept_err = 0.02; c->beam_err = 0.05; c->calib_laser = 0;
    c->sigma_num = 3.0; c->max_iter = 4;
    c->extR[0] = c->extR[4] = c->extR[8] = 1.0;
    c->extT[0] = 0.04165; c->extT[1] = 0.02326; c->extT[2] = -0.0284;
    c->mesh_min_spacing = 0.1; c->mesh_voxel = 0.4; c->mesh_region = 10.0; c->mesh_append_budget = 10000;
}
const char* immesh_create_error(void) { return "oracle"; }
const char* immesh_last_error(immesh_ctx*) { return "oracle"; }
}
