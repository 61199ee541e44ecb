/* rt1w_internal.h -- the four hooks librt1w.so exports beside include/rt1w.h, for its own diagnostics library librt1w_lab.so
 * (walk_lab.hip: the trace-only harness; wavefront.hip: the wavefront form behind RT1W_WAVEFRONT).  Not an interface for anybody else. */
#ifndef RT1W_INTERNAL_H
#define RT1W_INTERNAL_H
#include "rt1w.h"
#ifdef __cplusplus
extern "C" {
#endif
/* one wavefront render, as the product hands it to librt1w_lab.so: everything is enqueued on `stream`; the chunk sums go to
 * d_partial[n_chunks][npix][3]; the product resolves, times and synchronises */
typedef struct rt1w_wf_call {
    int device, variant;
    const void* view;   /* RtSceneView */
    const void* frame;  /* RtFrame */
    unsigned long long npix;
    uint32_t n_nodes, scope_depth, stack_need, n_h_nodes;
    const void* h_nodes; /* host copy of the flat node array (RtNode) */
    void* stream;        /* hipStream_t */
    double* d_partial;
    void** state;        /* the form's own per-context state, created on first use */
    rt1w_stats* stats;   /* grid, block, variant, sorted are filled by the form */
    const unsigned long long* h_segments; /* out: pinned word that holds the segment count once the stream has drained */
} rt1w_wf_call;
typedef int (*rt1w_wf_render_fn)(rt1w_wf_call*);
typedef void (*rt1w_wf_destroy_fn)(void* state);
#pragma GCC visibility push(default)
void rt1w_internal_register_wavefront(rt1w_wf_render_fn render, rt1w_wf_destroy_fn destroy); /* called by librt1w_lab.so when it is loaded */
const void* rt1w_internal_view(const rt1w_context* c); /* the context's RtSceneView (device pointers) */
int rt1w_internal_device(const rt1w_context* c);
void rt1w_internal_set_error(const char* msg);         /* what rt1w_last_error() will return on this thread */
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
