/* rt1w_internal.h -- the three hooks librt1w.so exports beside include/rt1w.h, for its own diagnostics library librt1w_lab.so
 * (walk_lab.hip: the trace-only harness).  Not an interface for anybody else. */
#ifndef RT1W_INTERNAL_H
#define RT1W_INTERNAL_H
#include "rt1w.h"
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)
const void* rt1w_internal_view(const rt1w_context* c); /* the context's RtSceneView (device pointers) */
int rt1w_internal_device(const rt1w_context* c);
void rt1w_internal_set_error(const char* msg);         /* what rt1w_last_error() will return on this thread */
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
