/* rt_kernel_plain.h -- body of the plain persistent render kernel (no workgroup-level reordering), shared by context.hip and by
 * the reference-stream build of the same kernel (context_ref.hip, compiled with RT_RNG_REFSTREAM inside its own namespace). */
#ifndef RT_KERNEL_PLAIN_H
#define RT_KERNEL_PLAIN_H

#include "rt_kernel_sorted.h"

/* 16-bit entries (node index < 32768, wrapper-exit flag in bit 15): half the LDS, used together with
 * the LDS node cache */
struct LdsStack16 {
    uint16_t* base;
    int sp;
    __device__ __forceinline__ void push(uint32_t v) { base[sp * RT_BLOCK] = (uint16_t)((v & 0x7FFFu) | ((v & RT_POP_FLAG) ? 0x8000u : 0u)); ++sp; }
    __device__ __forceinline__ uint32_t pop() { --sp; uint32_t x = base[sp * RT_BLOCK]; return (x & 0x7FFFu) | ((x & 0x8000u) ? RT_POP_FLAG : 0u); }
};
/* hot halves of all nodes copied into LDS once per workgroup (scenes of <= RT_LDS_NODE_CAP nodes): the
 * stack walk's dependent node fetches then cost LDS latency instead of L2/HBM latency */
#define RT_LDS_NODE_CAP 1024
struct LdsNodes {
    const RtNodeHot* base;
    __device__ __forceinline__ RtNodeHot hot(uint32_t n) const { return base[n]; }
};

/* counters[0] = next work item, counters[1] = traced segments */
#ifndef RT_SWEEP_WAVES
#define RT_SWEEP_WAVES 3 /* waves per SIMD the register allocator must leave room for in the sweep variants */
#endif
#define RT_PLAIN_WAVES(Cfg, CACHE) (Cfg::sweep && !Cfg::media ? RT_SWEEP_WAVES : (Cfg::sweep || CACHE ? 2 : 3))
template <class Cfg, bool CACHE>
__device__ __forceinline__ void rt_render_plain_body(const RtSceneView& sc, const RtFrame& f, rt_f64* __restrict__ partial,
                                                     unsigned long long* __restrict__ counters) {
    /* the sweep variants need no traversal stack (and no LDS at all) */
    typedef typename std::conditional<CACHE, uint16_t, uint32_t>::type stack_word;
    __shared__ stack_word stack_mem[Cfg::sweep ? 1 : RT_STACK_CAP * RT_BLOCK];
    __shared__ RtNodeHot node_cache[CACHE ? RT_LDS_NODE_CAP : 1];
    typename std::conditional<CACHE, LdsStack16, LdsStack>::type stk;
    stk.base = stack_mem + threadIdx.x;
    stk.sp = 0;
    typename std::conditional<CACHE, LdsNodes, RtGlobalNodes>::type ns;
    if constexpr (CACHE) {
        /* cooperative copy: 16 bytes per lane per step */
        const uint4* src = reinterpret_cast<const uint4*>(sc.nodes);
        uint4* dst = reinterpret_cast<uint4*>(node_cache);
        for (uint32_t i = threadIdx.x; i < sc.n_nodes * 4u; i += RT_BLOCK) dst[i] = src[(i >> 2) * 6u + (i & 3u)]; /* 96-B records, first 64 B */
        __syncthreads();
        ns.base = node_cache;
    } else {
        ns.p = sc.nodes;
    }

#ifdef RT_STAMPS
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 16; ++k) rt_stamp_acc[threadIdx.x >> 6][k] = 0;
        rt_stamp_last[threadIdx.x >> 6] = __builtin_amdgcn_s_memtime();
    }
#endif
    const unsigned long long n_items = rt_item_count(f);
    const unsigned long long npix = (unsigned long long)f.tile_w * f.tile_h;
    unsigned long long item = (unsigned long long)blockIdx.x * RT_BLOCK + threadIdx.x;
    bool fresh = true; /* `item` holds an id that was not decoded yet */
    bool have = false;
    uint32_t px = 0, py = 0, chunk = 0, s = 0, s_end = 0;
    RtV3d sum = rt_v3d(0.0, 0.0, 0.0);
    RtPath path;
    path.alive = false;
    unsigned long long segs = 0;

    for (;;) {
        RT_STAMP(0);
        if (!path.alive) {
            if (have && s == s_end) {
                rt_f64* dst = partial + ((unsigned long long)chunk * npix + (unsigned long long)py * f.tile_w + px) * 3ull;
                dst[0] = sum.x; dst[1] = sum.y; dst[2] = sum.z;
                have = false;
            }
            while (!have) {
                if (!fresh) {
                    /* wave-aggregated fetch: the lanes that need an item share one atomic */
                    unsigned long long need = __ballot(1);
                    uint32_t cnt = (uint32_t)__popcll(need);
                    uint32_t rank = lane_prefix(need);
                    unsigned long long base_item = 0;
                    if (rank == 0u) base_item = atomicAdd(&counters[0], (unsigned long long)cnt);
                    /* rank 0 is the first active lane: broadcast its value */
                    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_item);
                    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_item >> 32));
                    item = (((unsigned long long)hi << 32) | lo) + rank;
                }
                fresh = false;
                if (item >= n_items) break;
                rt_item_decode(f, item, px, py, chunk);
                if (px < f.tile_w && py < f.tile_h) {
                    s = chunk * f.chunk;
                    s_end = s + f.chunk < f.spp ? s + f.chunk : f.spp;
                    sum = rt_v3d(0.0, 0.0, 0.0);
                    have = true;
                }
            }
            if (!have) break; /* no work left: this lane retires */
            rt_path_begin(sc, f, f.x0 + px, rt_frame_row(f, py), f.sample_offset + s, path);
            RT_STAMP(1);
        }
        segs += path.depth_left != 0u ? 1ull : 0ull;
        rt_path_step<Cfg>(sc, ns, path, stk);
        if (!path.alive) {
            sum = rt_v3d_add(sum, path.radiance); /* pixel_color += ray_color(..), main.rs:972-989 */
            ++s;
        }
        RT_STAMP(6);
    }
    if (segs) atomicAdd(&counters[1], segs);
#ifdef RT_STAMPS
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 16; ++k) atomicAdd(&g_stamp_total[k], rt_stamp_acc[threadIdx.x >> 6][k]);
#endif
}


#endif
