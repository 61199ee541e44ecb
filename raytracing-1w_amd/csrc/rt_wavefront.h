/* rt_wavefront.h -- the same path tracer as a wavefront, for the big scenes (stack-walk variants).
 *
 * The persistent megakernel keeps a path in registers from camera ray to termination; on the big scenes that leaves
 * 8-22 % of the lanes busy (PMC): walks differ in length by an order of magnitude and a wave waits for its longest.
 * Here the path state lives in HBM (this is the record the roofline of SURVEY 8(d) counts) and one bounce is two
 * kernels:
 *   wf_trace  persistent; a lane takes a path from the queue, walks it, stores the hit, and takes the NEXT path the
 *             moment its walk ends -- no lane waits for another lane's walk (wave-aggregated atomic per refill round);
 *   wf_shade  one thread per queued path: rt_path_shade (the megakernel's own function), then either the sample's
 *             radiance to its slot or the path back into the queue for the next bounce.
 * A path is the same pure function of its state as in the megakernel (same RNG stream keyed by pixel and sample, same
 * core), and a pixel's samples are still summed in sample order per chunk (wf_chunk_sum), so the frame is bit-identical.
 */
#ifndef RT_WAVEFRONT_H
#define RT_WAVEFRONT_H

#include "rt_kernel_sorted.h"

struct WfPath {
    RtPath p;
    double t;              /* closest hit of the current ray (wf_trace -> wf_shade) */
    uint32_t prim, scope;
};

/* camera rays of samples [s0, s0+s_cnt) of every pixel of the tile; path id = s_local * npix + pixel */
__global__ void wf_generate(RtSceneView sc, RtFrame f, WfPath* __restrict__ paths, uint32_t* __restrict__ queue,
                            uint32_t s0, uint32_t s_cnt) {
    const unsigned long long npix = (unsigned long long)f.tile_w * f.tile_h;
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= npix * s_cnt) return;
    const uint32_t s_local = (uint32_t)(gid / npix);
    const uint32_t pixel = (uint32_t)(gid % npix);
    const uint32_t px = pixel % f.tile_w, py = pixel / f.tile_w;
    WfPath P;
    rt_path_begin(sc, f, f.x0 + px, rt_frame_row(f, py), f.sample_offset + s0 + s_local, P.p);
    P.t = 0.0; P.prim = RT_NONE; P.scope = RT_NONE;
    paths[gid] = P;
    queue[gid] = (uint32_t)gid;
}

#ifndef RT_WF_REFILL
#define RT_WF_REFILL 24u /* idle lanes of a wave that trigger a refill */
#endif
#define RT_WF_BATCH 256ull /* queue entries a wave takes from the global counter at a time */
/* counters: [0] next queue index (reset per launch), [1] traced segments, [2] length of the next queue */
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK) void wf_trace(RtSceneView sc, WfPath* __restrict__ paths, const uint32_t* __restrict__ queue,
                                                     unsigned long long n, unsigned long long* __restrict__ counters, uint32_t refill) {
    __shared__ uint32_t stack_mem[RT_STACK_CAP * RT_BLOCK];
    LdsStack stk;
    stk.base = stack_mem + threadIdx.x;
    stk.sp = 0;
    RtGlobalNodes ns{sc.nodes};
    bool have = false, exhausted = false;
    uint32_t id = 0;
    RtWalk k;
    RtRng rng = rt_rng_make(0u, 0u, 0u, 0u, 0u);
    unsigned long long segs = 0;
    /* queue indices are taken from the global counter a batch at a time and handed out inside the wave (one atomic per
     * RT_WF_BATCH paths: one per refill round would serialise the whole GPU on one address -- measured 9x slower) */
    unsigned long long w_next = 0, w_end = 0; /* wave-uniform: this wave's current batch [w_next, w_end) */
    for (;;) {
        const bool want = !have && !exhausted;
        /* refill when enough lanes are idle to pay for the fetch latency (the whole wave waits for the new rays), or when
         * nothing else is left to do */
        const uint32_t n_want = (uint32_t)__popcll(__ballot(want));
        if (n_want >= refill || (n_want != 0u && !RT_WAVE_ANY(have))) { /* uniform control flow: every lane keeps w_next / w_end up to date */
            if (w_next >= w_end) {
                unsigned long long base = 0;
                if ((threadIdx.x & 63u) == 0u) base = atomicAdd(&counters[0], (unsigned long long)RT_WF_BATCH);
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
                w_next = ((unsigned long long)hi << 32) | lo;
                w_end = w_next + RT_WF_BATCH < n ? w_next + RT_WF_BATCH : n;
                if (w_next > w_end) w_next = w_end; /* queue exhausted */
            }
            const unsigned long long need = __ballot(want);
            const uint32_t rank = lane_prefix(need);
            const unsigned long long qi = w_next + rank;
            const unsigned long long avail = w_end - w_next;
            const uint32_t cnt = (uint32_t)__popcll(need);
            if (want) {
                if (qi < w_end) {
                    id = queue[qi];
                    const WfPath& P = paths[id];
                    if (P.p.depth_left == 0u) { /* main.rs:59-61: no hit test at depth 0 */
                        paths[id].t = 0.0; paths[id].prim = RT_NONE; paths[id].scope = RT_NONE;
                    } else {
                        RtRay ray = P.p.ray;
                        if (Cfg::media) rng = P.p.rng; /* ConstantMedium draws while being traversed */
                        segs += 1ull;
                        rt_walk_begin(k, sc.root, ray, 0.001, RT_INF, stk);
                        have = true;
                    }
                } else if (avail == 0ull) {
                    exhausted = true; /* the batch just fetched starts at or beyond n */
                }
            }
            w_next += (cnt < avail) ? cnt : avail;
        }
        if (!RT_WAVE_ANY(have)) {
            if (!RT_WAVE_ANY(!exhausted)) break;
            continue;
        }
        if (have) {
            rt_walk_step<Cfg, true>(sc, ns, k, rng, stk);
            if (rt_walk_done(k, stk)) {
                WfPath& P = paths[id];
                P.t = k.best_t; P.prim = k.best_prim; P.scope = k.best_scope;
                if (Cfg::media) P.p.rng = rng;
                have = false;
            }
        }
    }
    if (segs) atomicAdd(&counters[1], segs);
}

/* one bounce of shading; survivors are appended to queue_out (order irrelevant), finished samples store their radiance */
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK) void wf_shade(RtSceneView sc, WfPath* __restrict__ paths, const uint32_t* __restrict__ queue_in,
                                                     unsigned long long n, uint32_t* __restrict__ queue_out,
                                                     unsigned long long* __restrict__ counters, double* __restrict__ sample_rad,
                                                     unsigned long long rad_base) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool alive = false;
    uint32_t id = 0;
    if (gid < n) {
        id = queue_in[gid];
        WfPath P = paths[id];
        RtTrace tr;
        tr.t = P.t; tr.prim = P.prim; tr.scope = P.scope; tr.cls = 0u;
        rt_path_shade<Cfg>(sc, P.p, tr);
        if (P.p.alive) {
            paths[id].p = P.p;
            alive = true;
        } else {
            double* dst = sample_rad + (rad_base + id) * 3ull;
            dst[0] = P.p.radiance.x; dst[1] = P.p.radiance.y; dst[2] = P.p.radiance.z;
        }
    }
    const unsigned long long m = __ballot(alive);
    if (m) {
        const uint32_t cnt = (uint32_t)__popcll(m), rank = lane_prefix(m);
        unsigned long long base = 0;
        const uint32_t first = (uint32_t)(__ffsll((long long)m) - 1);
        if ((threadIdx.x & 63u) == first) base = atomicAdd(&counters[2], (unsigned long long)cnt);
        const uint32_t lo = __shfl((uint32_t)base, (int)first), hi = __shfl((uint32_t)(base >> 32), (int)first);
        if (alive) queue_out[(((unsigned long long)hi << 32) | lo) + rank] = id;
    }
}

/* Σ of a pixel's samples of one chunk, in sample order (the megakernel's `sum = sum + radiance` per sample) */
__global__ void wf_chunk_sum(const double* __restrict__ sample_rad, double* __restrict__ partial_chunk,
                             unsigned long long npix, uint32_t s_cnt) {
    const unsigned long long pixel = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pixel >= npix) return;
    RtV3 sum = rt_v3(0.0, 0.0, 0.0);
    for (uint32_t s = 0; s < s_cnt; ++s) {
        const double* r = sample_rad + ((unsigned long long)s * npix + pixel) * 3ull;
        sum = sum + rt_v3(r[0], r[1], r[2]);
    }
    double* dst = partial_chunk + pixel * 3ull;
    dst[0] = sum.x; dst[1] = sum.y; dst[2] = sum.z;
}

#endif
