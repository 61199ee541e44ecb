/* rt_kernel_sorted.h -- body of the reordering render kernel, shared by the kernels built into librt1w.so
 * (context.hip) and by the topology-specialised kernels compiled at run time with hiprtc (jit.cpp). */
#ifndef RT_KERNEL_SORTED_H
#define RT_KERNEL_SORTED_H

#ifdef RT_STAMPS
/* DIAGNOSTIC BUILDS (librt1w_stamps.so; specialised kernels compiled with RT1W_JIT_STAMPS=1): per-wave cycle accounting by phase.  Buckets:
 * 0 loop/other, 1 regeneration (new sample / new work item), 2 traversal, 3 hit record,
 * 4 lambertian shading, 5 other shading, 6 sample bookkeeping.  Stamp values go only to
 * g_stamp_total, which no render code reads. */
__shared__ unsigned long long rt_stamp_acc[4][16];
__shared__ unsigned long long rt_stamp_last[4];
__device__ unsigned long long g_stamp_total[16];
__device__ __forceinline__ void rt_stamp_fn(int k) {
    unsigned long long t = __builtin_amdgcn_s_memtime();
    unsigned long long m = __ballot(1);
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) {
        int w = threadIdx.x >> 6;
        rt_stamp_acc[w][k] += t - rt_stamp_last[w];
        rt_stamp_last[w] = t;
    }
}
#define RT_STAMP(k) rt_stamp_fn(k)
#endif
#include "rt_core.h"

#define RT_BLOCK 256

struct LdsStack {
    uint32_t* base; /* this lane's entry 0; entry e at base[e * RT_BLOCK] */
    int sp;
    __device__ __forceinline__ void push(uint32_t v) { base[sp * RT_BLOCK] = v; ++sp; }
    __device__ __forceinline__ void poke(int above, uint32_t v) { base[(sp + above) * RT_BLOCK] = v; } /* write without moving the top */
    __device__ __forceinline__ uint32_t pop() { --sp; return base[sp * RT_BLOCK]; }
    __device__ __forceinline__ uint32_t at(int i) const { return base[i * RT_BLOCK]; } /* entry i, whatever the level */
    __device__ __forceinline__ void put(int i, uint32_t v) { base[i * RT_BLOCK] = v; }
};

/* N box-only steps in a row (rt_core.h: rt_walk_box_step) with the TOP OF THE STACK IN A REGISTER: in the one-step form every step
 * writes the children to LDS and the next step reads the top back -- an LDS round trip inside the chain pop -> fetch -> test -> push
 * that paces the walk.  Here the entry below the top is requested together with the node record (it is needed only after a miss), a
 * hit continues with the left child (the next node in pre-order) from the register, and the top goes back to LDS once, at the end.
 * Same entries, same order: a lane stops at its first entry that is not a BVH node, as rt_walk_box_step does.  For walks whose
 * stack starts at level 0 (the render kernels' outer walk).  Written without nested branches: the stores below the top that a lane
 * does not need land in dead slots (its own top's slot, rewritten at the end; slot 0 of a finished lane). */
template <class Cfg, int N, class NS, class Stack>
__device__ __forceinline__ void rt_walk_box_run(const NS& ns, RtWalk& k, Stack& stk) {
    int sp = stk.sp;
    bool go = sp > 0;
    uint32_t top = stk.at(sp > 0 ? sp - 1 : 0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool node = go && !(Cfg::scope_depth > 0 && (top & RT_POP_FLAG));
        const uint32_t under = stk.at(sp > 1 ? sp - 2 : 0);
        const uint32_t e = node ? top : 0u;
        const RtNodeHot nd = ns.hot(e);
        const uint32_t km = nd.kind & RT_KIND_MASK;
        const bool take = node && km <= RT_BVH1;
        const bool two = km == RT_BVH2;
        const bool hit = take && rt_walk_slab<Cfg, false>(k, nd);
        const uint32_t left = rt_ns_child<NS>(e, nd);
        uint32_t first = left, second = nd.b; /* left child (the next node in pre-order), then the right one: bvh.rs:38-47 */
        if constexpr (Cfg::ordered) {
            const uint32_t ord = (nd.kind >> RT_BVH_ORDER_SHIFT) & RT_BVH_ORDER_MASK;
            const double da = ord == 1u ? k.cur.d.x : (ord == 2u ? k.cur.d.y : k.cur.d.z);
            const bool left_lower = (nd.kind & RT_BVH_LEFT_LOWER) != 0u;
            if (two && ord != 0u && ((da < RT_R(0.0) && left_lower) || (da > RT_R(0.0) && !left_lower))) { first = nd.b; second = left; }
        }
        stk.put(sp > 0 ? sp - 1 : 0, second); /* the top's own slot: the second child of a hit BVHChild::Two, dead otherwise */
        sp += hit ? (two ? 1 : 0) : (take ? -1 : 0);
        top = hit ? first : (take ? under : top);
        go = take && sp > 0;
    }
    stk.put(sp > 0 ? sp - 1 : 0, top);
    stk.sp = sp;
}

__device__ __forceinline__ uint32_t lane_prefix(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

/* ---- render kernel with workgroup-level reordering ------------------------------------------
 * Same per-path arithmetic as rt_render_kernel; what changes is WHICH LANE runs which path.
 * Every iteration, between the closest-hit search and the shading, the 256 paths of a workgroup
 * are sorted by what they have to do next (Lambertian / dielectric / metal / other scatter /
 * terminal / idle): per-wave ballots + mbcnt give each lane its rank, a 4x6 count table in LDS
 * gives the class bases, and every lane hands its whole path state (52 dwords) to the lane at its
 * sorted position through an LDS exchange buffer ([qword][slot], so the gather side is
 * conflict-free).  Waves then shade (mostly) one material each instead of every wave running
 * every material's code for a few lanes, and the paths that ended sit together in the last
 * wave(s), so regeneration is coherent too.  Results cannot depend on the lane a path runs on
 * (a path is a pure function of its state; a pixel chunk's samples still run one after the other
 * and are summed in order), so the framebuffer is bit-identical to the unsorted kernel's. */
#ifndef RT_SORTED_CARRY_RADIANCE
#define RT_SORTED_CARRY_RADIANCE 0 /* A/B (RT1W_JIT_EXTRA_OPTS=-DRT_SORTED_CARRY_RADIANCE=1): round 3's form, the zero radiance carried across the loop */
#endif
#define RT_XCH_QW 26 /* qwords of per-path state exchanged */
#ifndef RT_XCH_PARTS
#define RT_XCH_PARTS 1 /* rounds the exchange is done in (LDS per workgroup = ceil(26 / parts) qwords x paths) */
#endif
#define RT_XCH_PER ((RT_XCH_QW + RT_XCH_PARTS - 1) / RT_XCH_PARTS)
#ifndef RT_SORT_BLOCK
#define RT_SORT_BLOCK 256 /* paths sorted together = workgroup size of the reordering kernel */
#endif
#ifdef RT_SORT_WAVES_OVERRIDE /* experiments (RT1W_JIT_EXTRA_OPTS=-DRT_SORT_WAVES_OVERRIDE=4) */
#define RT_SORT_WAVES(Cfg) RT_SORT_WAVES_OVERRIDE
#else
/* waves per SIMD the kernel is built for: 3 for the media-free sweep variants and for every scene-specialised kernel (the
 * specialised cornel_smoke kernel sits right at the 168-register step: 169 registers = 2 waves = -22 %, measured) */
#define RT_SORT_WAVES(Cfg) ((Cfg::sweep && (!Cfg::media || !std::is_void<typename Cfg::Topo>::value)) ? (RT_SORT_BLOCK == 512 ? 2 : 3) : 2)
#endif
template <class Cfg>
__device__ __forceinline__ void rt_render_sorted_body(const RtSceneView& sc, const RtFrame& f, rt_f64* __restrict__ partial,
                                                      unsigned long long* __restrict__ counters) {
    constexpr int NW = RT_SORT_BLOCK / 64;
    static_assert(Cfg::sweep, "the reordering kernel is built for the stackless variants");
    __shared__ unsigned long long xch[RT_XCH_PER * RT_SORT_BLOCK];
    __shared__ uint32_t cnt[NW][RT_N_CLS];
    LdsStack stk;
    stk.base = nullptr;
    stk.sp = 0;
    RtGlobalNodes ns{sc.nodes};

    const unsigned long long n_items = rt_item_count(f);
    const unsigned long long npix = (unsigned long long)f.tile_w * f.tile_h;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long item = (unsigned long long)blockIdx.x * RT_SORT_BLOCK + threadIdx.x;
    bool fresh = true, have = false, retired = false;
    uint32_t px = 0, py = 0, chunk = 0, s = 0;
    RtV3d sum = rt_v3d(RT_R(0.0), RT_R(0.0), RT_R(0.0));
    RtPath path;
    path.alive = false;
    path.depth_left = 0u;
    path.ray.o = path.ray.d = path.beta = path.radiance = rt_v3(RT_R(0.0), RT_R(0.0), RT_R(0.0));
    path.ray.time = RT_R(0.0);
    path.rng = rt_rng_make(0u, 0u, 0u, f.global_seed, RT_DOMAIN_RENDER);
    unsigned long long segs = 0;
#ifdef RT_STAMPS
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 16; ++k) rt_stamp_acc[threadIdx.x >> 6][k] = 0;
        rt_stamp_last[threadIdx.x >> 6] = __builtin_amdgcn_s_memtime();
    }
#endif

    for (;;) {
        RT_STAMP(6);
        /* 1. regeneration: next sample of the lane's item, or a new item */
        if (!path.alive && !retired) {
            uint32_t s_end = chunk * f.chunk + f.chunk < f.spp ? chunk * f.chunk + f.chunk : f.spp;
            if (have && s == s_end) {
                rt_f64* dst = partial + ((unsigned long long)chunk * npix + (unsigned long long)py * f.tile_w + px) * 3ull;
                dst[0] = sum.x; dst[1] = sum.y; dst[2] = sum.z;
                have = false;
            }
            while (!have) {
                if (!fresh) {
                    unsigned long long need = __ballot(1);
                    uint32_t cntn = (uint32_t)__popcll(need);
                    uint32_t rank = lane_prefix(need);
                    unsigned long long base_item = 0;
                    if (rank == 0u) base_item = atomicAdd(&counters[0], (unsigned long long)cntn);
                    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_item);
                    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_item >> 32));
                    item = (((unsigned long long)hi << 32) | lo) + rank;
                }
                fresh = false;
                if (item >= n_items) break;
                rt_item_decode(f, item, px, py, chunk);
                if (px < f.tile_w && py < f.tile_h) {
                    s = chunk * f.chunk;
                    sum = rt_v3d(RT_R(0.0), RT_R(0.0), RT_R(0.0));
                    have = true;
                }
            }
            if (!have) retired = true;
            else rt_path_begin(sc, f, f.x0 + px, rt_frame_row(f, py), f.sample_offset + s, path);
        }
        RT_STAMP(1);
        /* 2. closest hit + class */
        RtTrace tr;
        tr.t = RT_R(0.0); tr.prim = RT_NONE; tr.scope = RT_NONE; tr.cls = RT_CLS_IDLE;
        if (!retired) {
            segs += path.depth_left != 0u ? 1ull : 0ull;
            tr = rt_path_trace<Cfg>(sc, ns, path, stk);
        }
        RT_STAMP(2);
        /* 3. sort the workgroup's paths by class */
        uint32_t my_rank = 0;
#pragma unroll
        for (uint32_t c = 0; c < RT_N_CLS; ++c) {
            unsigned long long m = __ballot(tr.cls == c);
            if (tr.cls == c) my_rank = lane_prefix(m);
            if (lane == 0u) cnt[wave][c] = (uint32_t)__popcll(m);
        }
        RT_STAMP(8);
        __syncthreads();
        RT_STAMP(9);
        uint32_t dest = my_rank, idle_total = 0;
#pragma unroll
        for (uint32_t c = 0; c < RT_N_CLS; ++c) {
#pragma unroll
            for (uint32_t w = 0; w < (uint32_t)NW; ++w) {
                uint32_t v = cnt[w][c];
                if (c < tr.cls || (c == tr.cls && w < wave)) dest += v;
                if (c == RT_CLS_IDLE) idle_total += v;
            }
        }
        if (idle_total == RT_SORT_BLOCK) break; /* every path of the workgroup is done (uniform) */
        {
            /* the path state as 26 qwords (compile-time indices only: these stay registers), handed over in RT_XCH_PARTS rounds
             * through an exchange buffer of ceil(26 / parts) qwords per slot */
            unsigned long long st[RT_XCH_QW];
#define RT_PK2(a, b) (((unsigned long long)(b) << 32) | (unsigned long long)(uint32_t)(a))
            st[0] = rt_d2u(path.ray.o.x); st[1] = rt_d2u(path.ray.o.y); st[2] = rt_d2u(path.ray.o.z);
            st[3] = rt_d2u(path.ray.d.x); st[4] = rt_d2u(path.ray.d.y); st[5] = rt_d2u(path.ray.d.z);
            st[6] = rt_d2u(path.ray.time);
            st[7] = rt_d2u(path.beta.x); st[8] = rt_d2u(path.beta.y); st[9] = rt_d2u(path.beta.z);
            st[10] = rt_d2u(sum.x); st[11] = rt_d2u(sum.y); st[12] = rt_d2u(sum.z);
            st[13] = rt_d2u(tr.t);
            st[14] = RT_PK2(path.rng.k0, path.rng.k1); st[15] = RT_PK2(path.rng.c1, path.rng.blk);
            st[16] = RT_PK2(path.rng.left, path.rng.bv); st[17] = RT_PK2(path.rng.a0, path.rng.a1);
            st[18] = RT_PK2(path.rng.a2, path.rng.a3); st[19] = RT_PK2(path.rng.b0, path.rng.b1);
            st[20] = RT_PK2(path.rng.b2, path.rng.b3);
            st[21] = RT_PK2(tr.prim, tr.scope); st[22] = RT_PK2(tr.cls, path.depth_left);
            st[23] = RT_PK2(px, py); st[24] = RT_PK2(chunk, s);
            st[25] = RT_PK2((have ? 1u : 0u) | (retired ? 2u : 0u) | (path.alive ? 4u : 0u), 0u);
#undef RT_PK2
#pragma unroll
            for (int part = 0; part < RT_XCH_PARTS; ++part) {
                if (part) __syncthreads(); /* the previous round's readers are done with the buffer */
#pragma unroll
                for (int i = part * RT_XCH_PER; i < (part + 1) * RT_XCH_PER && i < RT_XCH_QW; ++i) xch[(i - part * RT_XCH_PER) * RT_SORT_BLOCK + dest] = st[i];
                RT_STAMP(10);
                __syncthreads();
                RT_STAMP(11);
#pragma unroll
                for (int i = part * RT_XCH_PER; i < (part + 1) * RT_XCH_PER && i < RT_XCH_QW; ++i) st[i] = xch[(i - part * RT_XCH_PER) * RT_SORT_BLOCK + threadIdx.x];
            }
#define RT_UP2(v, a, b) do { (a) = (uint32_t)(v); (b) = (uint32_t)((v) >> 32); } while (0)
            path.ray.o.x = rt_u2d(st[0]); path.ray.o.y = rt_u2d(st[1]); path.ray.o.z = rt_u2d(st[2]);
            path.ray.d.x = rt_u2d(st[3]); path.ray.d.y = rt_u2d(st[4]); path.ray.d.z = rt_u2d(st[5]);
            path.ray.time = rt_u2d(st[6]);
            path.beta.x = rt_u2d(st[7]); path.beta.y = rt_u2d(st[8]); path.beta.z = rt_u2d(st[9]);
            sum.x = rt_u2d(st[10]); sum.y = rt_u2d(st[11]); sum.z = rt_u2d(st[12]);
            tr.t = rt_u2d(st[13]);
            RT_UP2(st[14], path.rng.k0, path.rng.k1); RT_UP2(st[15], path.rng.c1, path.rng.blk);
            RT_UP2(st[16], path.rng.left, path.rng.bv); RT_UP2(st[17], path.rng.a0, path.rng.a1);
            RT_UP2(st[18], path.rng.a2, path.rng.a3); RT_UP2(st[19], path.rng.b0, path.rng.b1);
            RT_UP2(st[20], path.rng.b2, path.rng.b3);
            RT_UP2(st[21], tr.prim, tr.scope); RT_UP2(st[22], tr.cls, path.depth_left);
            RT_UP2(st[23], px, py); RT_UP2(st[24], chunk, s);
            const uint32_t flags = (uint32_t)st[25];
            have = (flags & 1u) != 0u; retired = (flags & 2u) != 0u; path.alive = (flags & 4u) != 0u;
#undef RT_UP2
        }
        RT_STAMP(7);
        /* 4. shading (coherent within a wave after the sort) */
        if (!retired) {
            /* a path's radiance is written by its terminal only (rt_path_shade: the emitting material never scatters), so it is zero here
             * and nothing of it is carried across the loop or through the exchange */
#if !RT_SORTED_CARRY_RADIANCE
            path.radiance = rt_v3(RT_R(0.0), RT_R(0.0), RT_R(0.0));
#endif
            rt_path_shade<Cfg>(sc, path, tr);
            if (!path.alive) {
                sum = rt_v3d_add(sum, path.radiance); /* pixel_color += ray_color(..), main.rs:972-989 */
#if RT_SORTED_CARRY_RADIANCE
                path.radiance = rt_v3(RT_R(0.0), RT_R(0.0), RT_R(0.0));
#endif
                ++s;
            }
        }
        RT_STAMP(5);
    }
    if (segs) atomicAdd(&counters[1], segs);
#ifdef RT_STAMPS
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 16; ++k) atomicAdd(&g_stamp_total[k], rt_stamp_acc[threadIdx.x >> 6][k]);
#endif
}

#endif
