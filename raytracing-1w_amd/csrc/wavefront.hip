/* wavefront.hip -- the wavefront form behind RT1W_WAVEFRONT (rt_wavefront.h: per-ray state as SoA queues in HBM, a trace kernel + a
 * shade kernel per bounce, a finish kernel for the tail), as part of librt1w_lab.so.
 *
 * It is the north star's other shape, built and rebuilt in rounds 1-3, bit-identical to the default kernels and measured at 0.4-0.9x of
 * their rate (docs/LAB_NOTES.md).  A measured loser does not belong in the product library: this file registers the form with
 * librt1w.so when the lab library is loaded (rt1w_internal_register_wavefront); without it RT1W_WAVEFRONT answers RT1W_ERR_UNSUPPORTED.
 * The product hands over one render at a time (rt1w_wf_call, csrc/rt1w_internal.h), resolves, times and synchronises. */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "rt1w.h"
#include "rt1w_internal.h"
#include "rt_kernel_plain.h"
#include "rt_wavefront.h"

namespace {

void wf_error(const std::string& m) { rt1w_internal_set_error(m.c_str()); }
bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    wf_error(std::string("wavefront form: ") + what + ": " + hipGetErrorString(e));
    return false;
}

/* per-context state of the form: two SoA path queues, per-sample radiance of one pass, device-side counters, walk records, grids */
struct WfState {
    WfQueue wf_q[2] = {{nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
    double* wf_rad = nullptr;
    WfCounters* wf_counters = nullptr; unsigned long long* wf_hcounters = nullptr;
    size_t wf_cap = 0;
    WfRecs wf_recs = {nullptr, 0.0, 1.0}; void* d_wf_recs = nullptr; /* walk records of the big scenes (null: not eligible) */
    int wf_grid_trace[22] = {0}, wf_grid_shade[3] = {0, 0, 0}, wf_grid_finish[3] = {0, 0, 0};
    bool wf_recs_tried = false; std::string wf_recs_error;
};

/* RT1W_WAVEFRONT: walk records of the vote-scheduled trace kernel, at the first wavefront render; eligible when every
 * MovingSphere has the same (time0, time1) */
int ensure_wf_recs(WfState* c, const rt1w_wf_call* k) {
    if (c->wf_recs.p) return RT1W_OK;
    if (c->wf_recs_tried) { wf_error(c->wf_recs_error); return RT1W_ERR_UNSUPPORTED; }
    c->wf_recs_tried = true;
    const RtNode* h_nodes = static_cast<const RtNode*>(k->h_nodes);
    std::vector<WfRec> recs(k->n_h_nodes);
    bool ok_ms = true, seen = false;
    double t0 = 0.0, t1 = 1.0;
    for (size_t i = 0; i < recs.size(); ++i) {
        const RtNode& n = h_nodes[i];
        WfRec& r = recs[i];
        r.kind = n.kind; r.b = n.b;
        for (int k = 0; k < 6; ++k) r.d[k] = n.d[k];
        r.d[6] = 0.0;
        if ((n.kind & RT_KIND_MASK) == RT_MSPHERE) {
            r.d[6] = n.e[2];
            if (!seen) { t0 = n.e[0]; t1 = n.e[1]; seen = true; }
            else if (memcmp(&t0, &n.e[0], 8) != 0 || memcmp(&t1, &n.e[1], 8) != 0) ok_ms = false;
        }
    }
    if (!ok_ms) { c->wf_recs_error = "wavefront form: the scene's moving spheres do not share one shutter interval"; wf_error(c->wf_recs_error); return RT1W_ERR_UNSUPPORTED; }
    if (!hip_ok(hipMalloc(&c->d_wf_recs, recs.size() * sizeof(WfRec) + 16), "hipMalloc(walk records)") ||
        !hip_ok(hipMemcpy(c->d_wf_recs, recs.data(), recs.size() * sizeof(WfRec), hipMemcpyHostToDevice), "hipMemcpy(walk records)")) { c->wf_recs_error = rt1w_last_error(); return RT1W_ERR_DEVICE; }
    c->wf_recs.p = (const WfRec*)c->d_wf_recs; c->wf_recs.ms_time0 = t0; c->wf_recs.ms_time1 = t1;
    return RT1W_OK;
}


/* ---- wavefront form (rt_wavefront.h) ---- */
typedef void (*wf_trace_t)(RtSceneView, RtFrame, WfQueue, WfCounters*, uint32_t, uint32_t, WfRecs);
typedef void (*wf_shade_t)(RtSceneView, RtFrame, WfQueue, WfQueue, WfCounters*, uint32_t, uint32_t, double*);
/* trace kernels: [variant 2/3][scene has wrappers][stack capacity 16/32] */
static wf_trace_t const g_wf_trace[3][2][2] = {
    {{wf_trace<RtCfgV2, false, 16>, wf_trace<RtCfgV2, false, 32>}, {wf_trace<RtCfgV2, true, 16>, wf_trace<RtCfgV2, true, 32>}},
    {{wf_trace<RtCfgV3, false, 16>, wf_trace<RtCfgV3, false, 32>}, {wf_trace<RtCfgV3, true, 16>, wf_trace<RtCfgV3, true, 32>}},
    {{wf_trace<RtCfgV4, false, 16>, wf_trace<RtCfgV4, false, 32>}, {wf_trace<RtCfgV4, true, 16>, wf_trace<RtCfgV4, true, 32>}}};
static wf_trace_t const g_wf_trace_lds[3][2] = {{wf_trace_lds<RtCfgV2, false>, wf_trace_lds<RtCfgV2, true>}, {wf_trace_lds<RtCfgV3, false>, wf_trace_lds<RtCfgV3, true>},
                                                {wf_trace_lds<RtCfgV4, false>, wf_trace_lds<RtCfgV4, true>}};
/* the plain trace kernels (the product's own walk by itself): V2, V3, V4, and V5 for scenes without wrappers */
static wf_trace_t const g_wf_trace_plain[4] = {wf_trace_plain<RtCfgV2>, wf_trace_plain<RtCfgV3>, wf_trace_plain<RtCfgV4>, wf_trace_plain<RtCfgV5>};
static wf_shade_t const g_wf_shade[3] = {wf_shade<RtCfgV2>, wf_shade<RtCfgV3>, wf_shade<RtCfgV4>};
typedef void (*wf_finish_t)(RtSceneView, RtFrame, WfQueue, WfCounters*, uint32_t, uint32_t, double*);
static wf_finish_t const g_wf_finish[3] = {wf_finish<RtCfgV2>, wf_finish<RtCfgV3>, wf_finish<RtCfgV4>};
#define RT_WF_PASS_PATHS (16ull << 20) /* paths in flight per pass: 2 queues x 16 Mi x 128 B = 4 GiB */

/* the wavefront form of one render: per chunk of samples, passes of <= RT_WF_PASS_PATHS paths; per pass
 * generate -> (trace -> shade) x max_depth with the queue lengths kept on the device; then the pass's samples are added
 * to the chunk sum in sample order */
int wf_render(rt1w_wf_call* k) {
    if (!*k->state) *k->state = new (std::nothrow) WfState();
    WfState* c = static_cast<WfState*>(*k->state);
    if (!c) { wf_error("out of memory"); return RT1W_ERR_NOMEM; }
    if (hipSetDevice(k->device) != hipSuccess) { wf_error("hipSetDevice"); return RT1W_ERR_DEVICE; }
    struct { hipStream_t stream; double* d_partial; } l{(hipStream_t)k->stream, k->d_partial};
    RtFrame f;
    memcpy(&f, k->frame, sizeof f);
    RtSceneView view;
    memcpy(&view, k->view, sizeof view);
    const unsigned long long npix = k->npix;
    struct { int variant; } L{k->variant};
    rt1w_stats* stats = k->stats;
    const int v = L.variant == 5 ? 2 : L.variant; /* no kernels of its own for the wrapper-free variant: V2's cover it */
    if (v < 2) { wf_error("the wavefront form exists for the stack-walk variants only"); return RT1W_ERR_INVALID; }
    if (npix > RT_WF_PASS_PATHS) { wf_error("wavefront form: tile larger than one pass (render it in strips)"); return RT1W_ERR_UNSUPPORTED; }
    if (f.max_depth > WF_MAX_BOUNCES) { wf_error("wavefront form: max_depth above WF_MAX_BOUNCES"); return RT1W_ERR_UNSUPPORTED; }
    const uint32_t s_pass_max = (uint32_t)(RT_WF_PASS_PATHS / npix);
    const uint32_t n_pass = (f.chunk + s_pass_max - 1u) / s_pass_max;      /* passes per chunk, of (nearly) equal size */
    const uint32_t s_pass = (f.chunk + n_pass - 1u) / n_pass;
    const size_t cap = (size_t)npix * s_pass;
    if (cap > c->wf_cap) {
        for (int k = 0; k < 2; ++k) {
            if (c->wf_q[k].f) (void)hipFree(c->wf_q[k].f);
            if (c->wf_q[k].u) (void)hipFree(c->wf_q[k].u);
            c->wf_q[k] = WfQueue{nullptr, nullptr, 0};
        }
        if (c->wf_rad) (void)hipFree(c->wf_rad);
        c->wf_rad = nullptr; c->wf_cap = 0;
        for (int k = 0; k < 2; ++k) {
            if (!hip_ok(hipMalloc((void**)&c->wf_q[k].f, cap * WF_NF * sizeof(double)), "hipMalloc(path queue)") ||
                !hip_ok(hipMalloc((void**)&c->wf_q[k].u, cap * WU_NU * sizeof(uint32_t)), "hipMalloc(path queue)")) return RT1W_ERR_NOMEM;
            c->wf_q[k].cap = cap;
        }
        if (!hip_ok(hipMalloc((void**)&c->wf_rad, cap * 3 * sizeof(double)), "hipMalloc(sample radiance)")) return RT1W_ERR_NOMEM;
        c->wf_cap = cap;
    }
    if (!c->wf_counters) {
        if (!hip_ok(hipMalloc((void**)&c->wf_counters, sizeof(WfCounters)), "hipMalloc(counters)") ||
            !hip_ok(hipHostMalloc((void**)&c->wf_hcounters, 2 * sizeof(unsigned long long), hipHostMallocDefault), "hipHostMalloc(counters)")) return RT1W_ERR_NOMEM;
    }
    /* trace kernel: the plain one (the product's walk by itself; default since round 3), or the vote-scheduled one of round 2
     * (RT1W_WF_TRACE=vote: kept for the A/B, profiles/r03_wavefront_*) */
    const char* wf_trace_env = getenv("RT1W_WF_TRACE");
    const bool plain_trace = !(wf_trace_env && wf_trace_env[0] == 'v');
    if (!plain_trace) { const int rcw = ensure_wf_recs(c, k); if (rcw < 0) return rcw; } /* the plain kernel reads the flat nodes themselves */
    const bool lds_recs = !plain_trace && k->n_nodes <= RT_WF_LDS_NODES && k->stack_need <= 16u && !getenv("RT1W_WF_NO_LDS");
    const int tblock = lds_recs ? RT_WF_LDS_BLOCK : RT_BLOCK;
    const wf_trace_t trace = plain_trace ? g_wf_trace_plain[L.variant == 5 ? 3 : v - 2]
                           : lds_recs ? g_wf_trace_lds[v - 2][k->scope_depth > 0u ? 1 : 0]
                                      : g_wf_trace[v - 2][k->scope_depth > 0u ? 1 : 0][k->stack_need <= 16u ? 0 : 1];
    const wf_shade_t shade = g_wf_shade[v - 2];
    const int gi = plain_trace ? 18 + (L.variant == 5 ? 3 : v - 2)
                 : lds_recs ? 12 + (v - 2) * 2 + (k->scope_depth > 0u ? 1 : 0) : (v - 2) * 4 + (k->scope_depth > 0u ? 2 : 0) + (k->stack_need <= 16u ? 0 : 1);
    if (!c->wf_grid_trace[gi] || !c->wf_grid_shade[v - 2]) {
        int per_cu = 0, per_cu_s = 0;
        hipDeviceProp_t prop;
        if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trace, tblock, 0), "occupancy query") ||
            !hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_s, shade, RT_BLOCK, 0), "occupancy query") ||
            !hip_ok(hipGetDeviceProperties(&prop, k->device), "hipGetDeviceProperties")) return RT1W_ERR_DEVICE;
        c->wf_grid_trace[gi] = prop.multiProcessorCount * (per_cu < 1 ? 1 : per_cu);
        c->wf_grid_shade[v - 2] = prop.multiProcessorCount * (per_cu_s < 1 ? 1 : per_cu_s) * 2; /* grid-stride; short blocks */
    }
    const int grid_t = c->wf_grid_trace[gi], grid_s = c->wf_grid_shade[v - 2];
    const wf_finish_t finish = g_wf_finish[v - 2];
    if (!c->wf_grid_finish[v - 2]) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, finish, RT_BLOCK, 0), "occupancy query") ||
            !hip_ok(hipGetDeviceProperties(&prop, k->device), "hipGetDeviceProperties")) return RT1W_ERR_DEVICE;
        c->wf_grid_finish[v - 2] = prop.multiProcessorCount * (per_cu < 1 ? 1 : per_cu);
    }
    uint32_t wf_bounces_want = plain_trace ? RT_WF_BOUNCES_PLAIN : RT_WF_BOUNCES;
    if (const char* e = getenv("RT1W_WF_BOUNCES")) { const int x = atoi(e); if (x >= 1 && x <= (int)WF_MAX_BOUNCES) wf_bounces_want = (uint32_t)x; }
    const uint32_t wf_bounces = f.max_depth < wf_bounces_want ? f.max_depth : wf_bounces_want;
    if (!hip_ok(hipMemsetAsync(&c->wf_counters->segs, 0, sizeof(unsigned long long), l.stream), "counter reset")) return RT1W_ERR_DEVICE;
    for (uint32_t ch = 0; ch < f.n_chunks; ++ch) {
        const uint32_t s_begin = ch * f.chunk;
        const uint32_t s_cnt = s_begin + f.chunk < f.spp ? f.chunk : f.spp - s_begin;
        for (uint32_t s0 = 0; s0 < s_cnt; s0 += s_pass) {
            const uint32_t s_n = s_cnt - s0 < s_pass ? s_cnt - s0 : s_pass;
            const unsigned long long n0 = npix * s_n;
            hipLaunchKernelGGL(wf_init_counters, dim3(1), dim3(128), 0, l.stream, c->wf_counters, f.max_depth ? n0 : 0ull);
            hipLaunchKernelGGL(wf_generate, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, l.stream, view, f, c->wf_q[0], s_begin + s0, s_n, c->wf_rad);
            for (uint32_t b = 0; b < wf_bounces; ++b) {
                hipLaunchKernelGGL(trace, dim3(grid_t), dim3(tblock), 0, l.stream, view, f, c->wf_q[b & 1u], c->wf_counters, b, s_begin + s0, c->wf_recs);
                hipLaunchKernelGGL(shade, dim3(grid_s), dim3(RT_BLOCK), 0, l.stream, view, f, c->wf_q[b & 1u], c->wf_q[(b + 1u) & 1u], c->wf_counters, b,
                                   s_begin + s0, c->wf_rad);
            }
            /* whatever is still alive after the wavefront bounces runs to its end in one launch */
            if (wf_bounces < f.max_depth)
                hipLaunchKernelGGL(finish, dim3(c->wf_grid_finish[v - 2]), dim3(RT_BLOCK), 0, l.stream, view, f, c->wf_q[wf_bounces & 1u], c->wf_counters,
                                   wf_bounces, s_begin + s0, c->wf_rad);
            hipLaunchKernelGGL(wf_chunk_sum, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, l.stream, (const double*)c->wf_rad,
                               l.d_partial + (size_t)ch * npix * 3, npix, s_n, s0 == 0u ? 1u : 0u);
        }
    }
    if (!hip_ok(hipGetLastError(), "kernel launch") ||
        !hip_ok(hipMemcpyAsync(c->wf_hcounters + 1, &c->wf_counters->segs, sizeof(unsigned long long), hipMemcpyDeviceToHost, l.stream), "counter copy")) return RT1W_ERR_DEVICE;
    k->h_segments = c->wf_hcounters + 1;
    if (stats) {
        stats->grid = (uint32_t)grid_t; stats->block = (uint32_t)tblock;
        stats->variant = (uint32_t)v; stats->sorted = 8u | (lds_recs ? 2u : 0u) | (plain_trace ? 64u : 0u); /* bit 3: wavefront form; bit 1: walk records in LDS; bit 6: plain trace kernel */
    }
    return RT1W_OK;
}

void wf_destroy(void* state) {
    WfState* c = static_cast<WfState*>(state);
    if (!c) return;
    void* bufs[] = {c->wf_q[0].f, c->wf_q[0].u, c->wf_q[1].f, c->wf_q[1].u, c->wf_rad, c->wf_counters, c->d_wf_recs};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (c->wf_hcounters) (void)hipHostFree(c->wf_hcounters);
    delete c;
}

struct Registrar { Registrar() { rt1w_internal_register_wavefront(wf_render, wf_destroy); } };
static Registrar g_registrar; /* runs when librt1w_lab.so is loaded */

} // namespace
