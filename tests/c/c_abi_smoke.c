/* The drop-in boundary from plain C99: include/immesh_c_api.h must compile as C, every entry point must link, and the host-only entry points
 * must work without a GPU (immesh_create must FAIL without one: there is no CPU fallback).  Built and run by tests/test_capi_symbols.py. */
#include "immesh_c_api.h"
#include <stdio.h>
#include <string.h>
#include <math.h>

int main(void) {
    immesh_config cfg;
    immesh_default_config(&cfg);
    if (cfg.voxel_size != 0.5 || cfg.max_layer != 2 || cfg.max_iter != 4 || cfg.mesh_append_budget != 10000) { printf("bad defaults\n"); return 1; }
    /* sharding helper: host mirror of the kernels' ownership function */
    cfg.shard_world = 4; cfg.shard_rank = 1; cfg.shard_brick_log2 = 5;
    int64_t key[3] = {100, -200, 7};
    int o = immesh_shard_owner(&cfg, key);
    int64_t key2[3] = {127, -193, 31};   /* same 32^3 brick */
    if (o < 0 || o >= 4 || immesh_shard_owner(&cfg, key2) != o) { printf("bad owner\n"); return 2; }
    /* constant-velocity prior: identity rotation, velocity 1 m/s along x, dt 0.1 -> x advances 0.1, covariance grows */
    double s[IMMESH_STATE_DOUBLES], out[IMMESH_STATE_DOUBLES];
    memset(s, 0, sizeof(s));
    s[0] = s[4] = s[8] = 1.0; s[12] = 1.0;
    for (int i = 0; i < 18; i++) s[24 + i * 19] = 1e-4;
    if (immesh_forward_without_imu(s, 0.1, 0.3, 0.5, out) != 0) { printf("forward failed\n"); return 3; }
    if (fabs(out[9] - 0.1) > 1e-15 || out[24 + 3 * 19] <= 1e-4 || out[0] != 1.0) { printf("bad prior %g %g\n", out[9], out[24 + 3 * 19]); return 4; }
    /* a context needs a HIP device */
    cfg.shard_world = 0;
    immesh_ctx* c = immesh_create(&cfg);
    if (c) { printf("created (GPU present)\n"); immesh_destroy(c); }
    else printf("create refused: %s\n", immesh_create_error());
    /* every other entry point links */
    void* fns[] = {(void*)immesh_map_build, (void*)immesh_register, (void*)immesh_residuals, (void*)immesh_map_update, (void*)immesh_mesh_scan, (void*)immesh_mesh_wait,
                   (void*)immesh_mesh_sizes, (void*)immesh_mesh_fetch, (void*)immesh_process_scan, (void*)immesh_dump_planes, (void*)immesh_counters,
                   (void*)immesh_last_timing, (void*)immesh_profile_enable, (void*)immesh_profile_read, (void*)immesh_set_allreduce, (void*)immesh_downsample,
                   (void*)immesh_downsample_result, (void*)immesh_last_error};
    for (unsigned i = 0; i < sizeof(fns) / sizeof(fns[0]); i++) if (!fns[i]) return 5;
    printf("c abi ok\n");
    return 0;
}
