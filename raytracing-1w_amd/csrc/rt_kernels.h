/* rt_kernels.h -- the render kernels of librt1w.so as __global__ entry points: launch bounds, template arguments and the two small
 * kernels behind them (resolve, quantise).  Their bodies are rt_kernel_sorted.h / rt_kernel_plain.h / rt_walk_pair.h.  Kept apart from
 * context.hip so that the identity of the kernels' source text (bench.py: kernel_sources_id, what a stored PMC measurement is matched
 * against) does not move with host-side edits.  Included by context.hip only. */
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include "rt_kernel_plain.h"

namespace {

template <class Cfg, bool CACHE = false>
__global__ __launch_bounds__(RT_BLOCK, RT_PLAIN_WAVES(Cfg, CACHE)) void rt_render_kernel(RtSceneView sc, RtFrame f, double* __restrict__ partial,
                                                                                  unsigned long long* __restrict__ counters) {
    rt_render_plain_body<Cfg, CACHE>(sc, f, partial, counters);
}

/* sphere scenes (random_scene): the plain kernel with the pair walk of rt_walk_pair.h */
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK, RT_STACK_WAVES) void rt_render_kernel_pw(RtSceneView sc, RtPwView pw, RtFrame f, double* __restrict__ partial,
                                                                           unsigned long long* __restrict__ counters) {
    rt_render_plain_body<Cfg, false, true>(sc, f, partial, counters, &pw);
}

/* sliced stack walk + reordering of the finished paths at the end of every slice (rt_kernel_plain.h: rt_render_ss_body) */
template <class Cfg, int CAP, int PARTS>
__global__ __launch_bounds__(RT_BLOCK, RT_STACK_WAVES) void rt_render_kernel_ss(RtSceneView sc, RtFrame f, double* __restrict__ partial,
                                                                           unsigned long long* __restrict__ counters) {
    rt_render_ss_body<Cfg, CAP, PARTS>(sc, f, partial, counters);
}

/* the same with the scene's most visited nodes in LDS (rt_walk_table.h): scenes whose walk needs at most RT_SS_HC_CAP stack entries -- the
 * 16 KB that the shorter stack columns leave are the cache */
#define RT_SS_HC_CAP 16
#ifndef RT_SS_HC_RECORDS
#define RT_SS_HC_RECORDS 256
#endif
#ifndef RT_SS_HC_PARTS
#define RT_SS_HC_PARTS 3
#endif
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK, RT_STACK_WAVES) void rt_render_kernel_ss_hc(RtSceneView sc, RtFrame f, double* __restrict__ partial,
                                                                              unsigned long long* __restrict__ counters) {
    rt_render_ss_body<Cfg, RT_SS_HC_CAP, RT_SS_HC_PARTS, false, RT_SS_HC_RECORDS>(sc, f, partial, counters);
}

/* node visits of a small render, counted per node: what the context ranks the walk table by (context.hip) */
struct RtCountingNodes {
    static constexpr bool virt = false;
    const RtNode* p;
    uint32_t* visits;
    __device__ RtNodeHot hot(uint32_t n) const { atomicAdd(&visits[n], 1u); return *reinterpret_cast<const RtNodeHot*>(p + n); }
};
struct RtLocalStack {
    uint32_t e[RT_STACK_CAP + 2];
    int sp;
    __device__ void push(uint32_t v) { e[sp++] = v; }
    __device__ void poke(int above, uint32_t v) { e[sp + above] = v; }
    __device__ uint32_t pop() { return e[--sp]; }
};
__global__ void rt_visit_count_kernel(RtSceneView sc, RtFrame f, uint32_t* __restrict__ visits) {
    const unsigned long long n = (unsigned long long)f.tile_w * f.tile_h * f.spp;
    RtCountingNodes ns; ns.p = sc.nodes; ns.visits = visits;
    RtLocalStack stk; stk.sp = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t s = (uint32_t)(i % f.spp);
        const unsigned long long pix = i / f.spp;
        RtPath path;
        rt_path_begin(sc, f, (uint32_t)(pix % f.tile_w), (uint32_t)(pix / f.tile_w), s, path);
        while (path.alive) {
            const RtTrace tr = rt_path_trace<RtCfgV3>(sc, ns, path, stk);
            rt_path_shade<RtCfgV3>(sc, path, tr);
        }
    }
}

/* the same for sphere scenes: the pair walk in slices + the reordering of the finished paths (RT_PW_SS_STACK, rt_kernel_plain.h) */
#ifndef RT_SS_CAP
#define RT_SS_CAP RT_STACK_CAP /* stack entries per lane of the stack-walk kernels that reorder (experiments: 16 for four workgroups per CU) */
#endif
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK, RT_STACK_WAVES) void rt_render_kernel_pw_ss(RtSceneView sc, RtPwView pw, RtFrame f, double* __restrict__ partial,
                                                                              unsigned long long* __restrict__ counters) {
    rt_render_ss_body<Cfg, RT_PW_SS_STACK, RT_PW_SS_PARTS, true>(sc, f, partial, counters, &pw);
}

/* the reordering kernel proper (rt_kernel_sorted.h) */
template <class Cfg>
__global__ __launch_bounds__(RT_SORT_BLOCK, RT_SORT_WAVES(Cfg)) void rt_render_kernel_sorted(RtSceneView sc, RtFrame f, double* __restrict__ partial,
                                                                                       unsigned long long* __restrict__ counters) {
    rt_render_sorted_body<Cfg>(sc, f, partial, counters);
}

/* Sum the chunk partials of each pixel in chunk order; then Color::into_sampled
 * (color.rs:14-21) unless raw sums were asked for. */
__global__ void rt_resolve_kernel(const double* __restrict__ partial, double* __restrict__ out,
                                  unsigned long long npix, uint32_t n_chunks, uint32_t spp, uint32_t out_sum, uint32_t carry) {
    /* carry bit 0: `out` already holds the raw sums of earlier sample passes, the chunks of this pass are added to them IN ORDER (the same
     * sequence of additions as one pass over all chunks: same bits); bit 1: more passes follow, `out` stays raw */
    unsigned long long p = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    RtV3 total = (carry & 1u) ? rt_v3(out[p * 3 + 0], out[p * 3 + 1], out[p * 3 + 2]) : rt_v3(0.0, 0.0, 0.0);
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const double* src = partial + ((unsigned long long)c * npix + p) * 3ull;
        total = total + rt_v3(src[0], src[1], src[2]);
    }
    if (!out_sum && !(carry & 2u)) total = rt_into_sampled(total, spp);
    out[p * 3 + 0] = total.x; out[p * 3 + 1] = total.y; out[p * 3 + 2] = total.z;
}

/* Display for SampledColor (src/color.rs:56-65) on the device: gamma-2, clamp, *256, truncate; rows flipped into the
 * order the reference prints them (j = height-1 first, src/main.rs:957-960,1003-1007).  in = means [tile_h][tile_w][3] f64
 * (row 0 = j = y0), out = [tile_h][tile_w][3] u8 top-down. */
__global__ void rt_quantize_kernel(const double* __restrict__ means, uint8_t* __restrict__ out, uint32_t w, uint32_t h) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n = (unsigned long long)w * h;
    if (i >= n) return;
    uint32_t row = (uint32_t)(i / w), col = (uint32_t)(i % w);
    const double* src = means + ((unsigned long long)(h - 1u - row) * w + col) * 3ull;
    uint8_t* dst = out + i * 3ull;
    dst[0] = (uint8_t)rt_quantize(src[0]); dst[1] = (uint8_t)rt_quantize(src[1]); dst[2] = (uint8_t)rt_quantize(src[2]);
}

} // namespace

#endif
