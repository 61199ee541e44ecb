/* walk_lab.h -- entry points of the trace-only harness (walk_lab.hip).  Diagnostics: a library of their own (librt1w_lab.so, linked
 * against librt1w.so), not part of include/rt1w.h; tools/walk_lab.py and the GPU tests bind them with ctypes. */
#ifndef RT1W_WALK_LAB_H
#define RT1W_WALK_LAB_H

#include <stdint.h>

#include "rt1w.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

typedef struct rt1w_lab rt1w_lab;

int rt1w_lab_create(rt1w_context* c, const rt1w_scene* s, rt1w_lab** out);
void rt1w_lab_destroy(rt1w_lab* l);
/* out = {W1 available, inner pair records, leaf groups, kernel variant of the product's walk} */
int rt1w_lab_info(const rt1w_lab* l, uint32_t out[4]);
/* rays traced at bounces 0 .. n_bounces-1 by the paths of the tile (spp samples per pixel): out[b][path][8] =
 * {origin, direction, time, valid}; path = (row * tile_w + column) * spp + sample */
int rt1w_lab_dump_rays(rt1w_lab* l, const rt1w_render_params* p, uint32_t n_bounces, double* out_host);
/* L1 gather probe: records (64 B of the node array at pseudo-random indices) per launch and the kernel ms of
 * {own record, quad-shared} x {independent, dependent} index streams */
int rt1w_lab_gather_probe(rt1w_lab* l, uint32_t iters, uint32_t blocks_per_cu, double out_ms[4], uint64_t* records_per_launch);
/* rays[n][8] (slot 7 ignored) */
int rt1w_lab_set_rays(rt1w_lab* l, const double* rays, uint64_t n);
/* closest hit (t_min 0.001, t_max inf: main.rs:62) of every ray with walk `mode`; kernel time = best of `repeats` */
int rt1w_lab_trace(rt1w_lab* l, int mode, const uint32_t params[4], int repeats, double* out_t, uint32_t* out_prim, uint32_t* out_flags,
                   double* ms_best, uint64_t stats_out[8]);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
