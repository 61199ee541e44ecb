/* rt_kernel_plain.h -- body of the plain persistent render kernel (no workgroup-level reordering), shared by context.hip and by
 * the reference-stream build of the same kernel (context_ref.hip, compiled with RT_RNG_REFSTREAM inside its own namespace). */
#ifndef RT_KERNEL_PLAIN_H
#define RT_KERNEL_PLAIN_H

#include "rt_kernel_sorted.h"
#if !defined(RT_RNG_REFSTREAM)
#include "rt_walk_pair.h"
#define RT_HAVE_PW 1
#else
#undef RT_HAVE_PW /* the reference-stream build of this header keeps the one-entry-per-step walk */
struct RtPwView { int unused; };
#endif
/* the pair walk in the kernel that also reorders the finished paths: RT_PW_SS_STACK entries per lane instead of RT_PW_STACK (what leaves
 * room for the exchange buffer next to the pair walk's stacks and queues at three workgroups per CU), the exchange in RT_PW_SS_PARTS rounds */
#define RT_PW_SS_STACK 12
#ifndef RT_PW_SS_PARTS
#define RT_PW_SS_PARTS 3
#endif
#ifndef RT_CAM_AT_USE
#define RT_CAM_AT_USE 0 /* 1: rt_render_ss_body reads the camera block by scalar loads where a path begins instead of holding it across the loop
                            (its 24 doubles sit in SGPRs spilled to VGPR lanes and scratch).  Measured at 128 spp, kernel Mpaths/s: random_scene
                            933 -> 904, final_scene 336 -> 334 (the scalar loads' s_waitcnt also waits for the LDS): off */
#endif
#ifndef RT_PW_BOX_STEPS
#define RT_PW_BOX_STEPS 6 /* inner records a lane visits per vote */
#endif
#ifndef RT_MEDIA_PREFILL
#define RT_MEDIA_PREFILL 1
#endif
#ifndef RT_PW_LEAF_STEPS
#define RT_PW_LEAF_STEPS 1 /* pending groups a lane handles per leaf vote */
#endif
#ifndef RT_PW_VOTES
#define RT_PW_VOTES 8u /* lanes with a pending group that make a wave do leaf work.
 * Tuned inside the render kernel on random_scene 1200x800x100 (kernel Mpaths/s on the reference tree / the SAH tree; the one-entry-
 * per-step walk: 716 / 993; profiles/r03_pair_walk_tuning.txt): (box steps per vote, votes, queue entries) (1,24,4) 673 / 912,
 * (2,24,4) 777 / 1049, (2,8,4) 833 / 1098, (2,40,4) 738 / 1020, (4,8,4) 868 / 1142, (6,8,4) 874 / 1160, (6,8,8) 883 / 1171,
 * (8,8,8) 879 / 1156, (12,8,8) 881 / 1143; two groups per leaf vote: -3 %.  The votes cost as much as a step's bookkeeping, so a
 * lane does several inner records per vote (as the stack walk's two steps per vote); leaf work is started as soon as 8 lanes wait. */
#endif

/* 16-bit entries (node index < 32768, wrapper-exit flag in bit 15): half the LDS, used together with
 * the LDS node cache */
struct LdsStack16 {
    uint16_t* base;
    int sp;
    __device__ __forceinline__ void push(uint32_t v) { base[sp * RT_BLOCK] = (uint16_t)((v & 0x7FFFu) | ((v & RT_POP_FLAG) ? 0x8000u : 0u)); ++sp; }
    __device__ __forceinline__ void poke(int above, uint32_t v) { base[(sp + above) * RT_BLOCK] = (uint16_t)((v & 0x7FFFu) | ((v & RT_POP_FLAG) ? 0x8000u : 0u)); }
    __device__ __forceinline__ uint32_t pop() { --sp; uint32_t x = base[sp * RT_BLOCK]; return (x & 0x7FFFu) | ((x & 0x8000u) ? RT_POP_FLAG : 0u); }
    __device__ __forceinline__ uint32_t at(int i) const { uint32_t x = base[i * RT_BLOCK]; return (x & 0x7FFFu) | ((x & 0x8000u) ? RT_POP_FLAG : 0u); }
    __device__ __forceinline__ void put(int i, uint32_t v) { base[i * RT_BLOCK] = (uint16_t)((v & 0x7FFFu) | ((v & RT_POP_FLAG) ? 0x8000u : 0u)); }
};
/* hot halves of all nodes copied into LDS once per workgroup (scenes of <= RT_LDS_NODE_CAP nodes): the
 * stack walk's dependent node fetches then cost LDS latency instead of L2/HBM latency */
#define RT_LDS_NODE_CAP 1024
struct LdsNodes {
    static constexpr bool virt = false;
    const RtNodeHot* base;
    __device__ __forceinline__ RtNodeHot hot(uint32_t n) const { return base[n]; }
};

#ifndef RT_SLICE_IDLE
/* stack walk: finished lanes of a wave that end a slice of the walk (0: every walk runs to its end).  Measured (Mpaths/s at 48 spp): media-free
 * kernels (two steps per vote) 40: 636, 48: 662, 56: 675; media kernels 40: 190, 48: 191, 56: 189 (round 2); with the box-only steps of
 * round 3 (100 spp): 24: 244, 32: 253, 36: 253, 40: 254, 48: 251, 56: 236, 60: 222 */
#define RT_SLICE_IDLE(Cfg) (Cfg::media ? 40 : 56)
#endif
#ifndef RT_SLICE_PRIM_STEPS
#define RT_SLICE_PRIM_STEPS 0 /* a primitive-only step + n more box steps behind the box steps: measured 247-249 against 252 Mpaths/s, off */
#endif
#ifndef RT_SLICE_BOX_RUN_ORDERED
#define RT_SLICE_BOX_RUN_ORDERED 0 /* the near-far kernels (V4) keep the one-step form: the register form measured 233 against 257 Mpaths/s there (tools/v4_ab.py) */
#endif
#ifndef RT_SLICE_BOX_RUN
#define RT_SLICE_BOX_RUN 1 /* the box-only steps keep the top of the stack in a register (rt_kernel_sorted.h: rt_walk_box_run): final_scene +3 % */
#endif
#ifndef RT_SLICE_HEAVY_EVERY
#define RT_SLICE_HEAVY_EVERY 2 /* only every second full step of a media kernel takes ConstantMedium entries (rt_walk_light_step in between): +2 % */
#endif
#ifndef RT_SLICE_BOX_MIN_LANES
#define RT_SLICE_BOX_MIN_LANES 0
#endif
#ifndef RT_SLICE_BOX_STEPS
/* box-only steps per vote of the sliced stack walk (rt_walk_box_step): scenes with media only -- their full step is long (a medium is two
 * sphere tests, a logarithm and a draw) and 84 % of final_scene's visits are boxes.  Measured, final_scene 800x800x100, kernel Mpaths/s
 * (profiles/r03_box_steps.txt): 0: 215, 3: 229, 4: 251, 5: 247, 6: 243 (234 with the wait flag), 8: 224, 10: 212-215, 16: 186; a run-time
 * loop that ends once fewer than 16-40 lanes are still between boxes: 227-242 */
#define RT_SLICE_BOX_STEPS(Cfg) (Cfg::media ? 4 : 0)
#endif
#ifndef RT_SLICE_TWO_STEPS
#define RT_SLICE_TWO_STEPS(Cfg) (!Cfg::media)
#endif
/* counters[0] = next work item, counters[1] = traced segments */
#ifndef RT_SWEEP_WAVES
#define RT_SWEEP_WAVES 3 /* waves per SIMD the register allocator must leave room for in the sweep variants */
#endif
#ifndef RT_STACK_WAVES
#define RT_STACK_WAVES 3 /* waves per SIMD the stack-walk kernels are built for.  Measured (random_scene / final_scene, Mpaths/s at 48 spp):
                            2 waves (256 VGPRs, no spills) 559 / 171, 3 waves (168, spills in shading only) 662 / 188, 4 waves (128) 545 / 179 */
#endif
#define RT_PLAIN_WAVES(Cfg, CACHE) (Cfg::sweep && !Cfg::media ? RT_SWEEP_WAVES : (Cfg::sweep || CACHE ? 2 : RT_STACK_WAVES))
/* PW: the walk is the pair walk of rt_walk_pair.h (sphere scenes; `pw` = its records), else the one-entry-per-step walk */
template <class Cfg, bool CACHE, bool PW = false>
__device__ __forceinline__ void rt_render_plain_body(const RtSceneView& sc, const RtFrame& f, rt_f64* __restrict__ partial,
                                                     unsigned long long* __restrict__ counters, const RtPwView* pwp = nullptr) {
    /* the sweep variants need no traversal stack (and no LDS at all) */
    typedef typename std::conditional<CACHE, uint16_t, uint32_t>::type stack_word;
    __shared__ stack_word stack_mem[(Cfg::sweep || PW) ? 1 : RT_STACK_CAP * RT_BLOCK];
#if defined(RT_HAVE_PW)
    __shared__ uint32_t pw_ref[PW ? RT_PW_STACK * RT_BLOCK : 1];
    __shared__ float pw_ent[PW ? RT_PW_STACK * RT_BLOCK : 1];
    __shared__ uint32_t pw_q[PW ? RT_PW_QCAP * RT_BLOCK : 1];
    RtPwLds<RT_BLOCK> pwm;
    pwm.ref = pw_ref + threadIdx.x; pwm.ent = pw_ent + threadIdx.x; pwm.q = pw_q + threadIdx.x;
    RtPwLane pwl; pwl.cur = RT_PW_NONE; pwl.qh = 0u; pwl.qn = 0u; pwl.sp = 0;
#endif
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
    RtV3d sum = rt_v3d(RT_R(0.0), RT_R(0.0), RT_R(0.0));
    RtPath path;
    path.alive = false;
    unsigned long long segs = 0;

    bool walking = false; /* a suspended walk (sliced stack walk below): its closest hit so far and innermost wrapper */
    double w_best_t = RT_INF;
    uint32_t w_best_prim = RT_NONE, w_best_scope = RT_NONE, w_scope = RT_NONE;
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
                    sum = rt_v3d(RT_R(0.0), RT_R(0.0), RT_R(0.0));
                    have = true;
                }
            }
            if (!have) break; /* no work left: this lane retires */
            rt_path_begin(sc, f, f.x0 + px, rt_frame_row(f, py), f.sample_offset + s, path);
            RT_STAMP(1);
        }
#if defined(RT_HAVE_PW)
        if constexpr (PW) {
            /* ---- pair walk (rt_walk_pair.h) in slices: the same suspension rule as the stack walk below; what lives across a slice
             * is the lane's next record, its stack and queue levels (stack and queue are in LDS) and the closest hit so far ---- */
            const RtPwView& pw = *pwp;
            if (!walking) {
                segs += path.depth_left != 0u ? 1ull : 0ull;
                w_best_t = RT_INF; w_best_prim = RT_NONE;
                if (path.depth_left != 0u) {
                    /* the root's own box first (bvh.rs:32); a NaN shutter fraction goes to the one-entry-per-step walk at once */
                    const double frac0 = (path.ray.time - pw.ms_time0) / (pw.ms_time1 - pw.ms_time0);
                    if (rt_isnan(frac0)) { pwl.sp = 0; pwl.qn = 0u; pwl.qh = 0u; pwl.cur = RT_PW_NONE; w_scope = 1u; walking = true; } /* w_scope 1: redo marker */
                    else { w_scope = RT_NONE; walking = rt_pw_begin(pw, pwl, pwm, path.ray.o, rt_inv3(path.ray.d), RT_R(0.001)); }
                }
            }
            RtTrace tr;
            tr.t = RT_R(0.0); tr.prim = RT_NONE; tr.scope = RT_NONE; tr.cls = RT_CLS_TERMINAL;
            if (walking) {
                const RtV3 o = path.ray.o, d = path.ray.d, inv = rt_inv3(d);
                const double frac = (path.ray.time - pw.ms_time0) / (pw.ms_time1 - pw.ms_time0);
                double best_t = w_best_t;
                uint32_t best_prim = w_best_prim;
                bool bad = w_scope != RT_NONE;
                const uint32_t lanes_here = (uint32_t)__popcll(__ballot(1));
                const uint32_t stop_at = lanes_here > (uint32_t)RT_SLICE_IDLE(Cfg) ? lanes_here - (uint32_t)RT_SLICE_IDLE(Cfg) : 0u;
                for (;;) {
                    const bool more = !bad && !rt_pw_done(pwl);
                    if (__popcll(__ballot(more)) <= stop_at) break;
                    const bool can_box = more && rt_pw_can_box(pwl), can_leaf = more && pwl.qn > 0u;
                    const uint32_t nb = (uint32_t)__popcll(__ballot(can_box)), nl = (uint32_t)__popcll(__ballot(can_leaf));
                    if (nl >= RT_PW_VOTES || nb == 0u) {
                        if (can_leaf) {
                            rt_pw_group_step(pw, pwl, pwm, o, d, inv, frac, RT_R(0.001), best_t, best_prim);
#pragma unroll
                            for (int extra = 1; extra < RT_PW_LEAF_STEPS; ++extra)
                                if (pwl.qn > 0u) rt_pw_group_step(pw, pwl, pwm, o, d, inv, frac, RT_R(0.001), best_t, best_prim);
                            if (rt_isnan(best_t)) bad = true;
                        }
                    } else if (can_box) {
                        rt_pw_box_step(pw, pwl, pwm, o, inv, RT_R(0.001), best_t);
                        /* more records on the same vote (as the stack walk's two steps per vote: the votes cost as much as a step's bookkeeping) */
#pragma unroll
                        for (int extra = 1; extra < RT_PW_BOX_STEPS; ++extra)
                            if (rt_pw_can_box(pwl)) rt_pw_box_step(pw, pwl, pwm, o, inv, RT_R(0.001), best_t);
                    }
                }
                RT_STAMP(2);
                if (bad) {
                    /* the closest hit turned NaN (a NaN root is accepted, sphere.rs:43-48): monotonicity is gone, the segment is redone
                     * by the one-entry-per-step walk on the same LDS words (no random draws in this scene's walk: nothing to restore) */
                    LdsStack cs;
                    cs.base = pw_ref + threadIdx.x; cs.sp = 0;
                    uint32_t scope_;
                    rt_traverse_stack<Cfg, true>(sc, ns, sc.root, path.ray, RT_R(0.001), RT_INF, path.rng, cs, best_t, best_prim, scope_);
                    pwl.cur = RT_PW_NONE; pwl.sp = 0; pwl.qn = 0u; w_scope = RT_NONE;
                }
                if (rt_pw_done(pwl)) {
                    walking = false;
                    tr.t = best_t; tr.prim = best_prim; tr.scope = RT_NONE;
                    if (tr.prim != RT_NONE) {
                        const uint32_t mk = RT_MAT_KINDF(ns.hot(tr.prim).mat) & 0xFFu;
                        tr.cls = mk == RT_MAT_LAMBERTIAN ? RT_CLS_LAMBERT : mk == RT_MAT_DIELECTRIC ? RT_CLS_DIELECTRIC
                               : mk == RT_MAT_METAL ? RT_CLS_METAL : mk == RT_MAT_ISOTROPIC ? RT_CLS_OTHER : RT_CLS_TERMINAL;
                    }
                } else {
                    w_best_t = best_t; w_best_prim = best_prim;
                }
            }
            if (!walking) {
                path.radiance = rt_v3(RT_R(0.0), RT_R(0.0), RT_R(0.0)); /* written by the terminal only: not carried across the walk */
                rt_path_shade<Cfg>(sc, path, tr);
                if (!path.alive) {
                    sum = rt_v3d_add(sum, path.radiance);
                    ++s;
                }
            }
        } else
#endif
        if constexpr (!Cfg::sweep && RT_SLICE_IDLE(Cfg) > 0 && RT_WALK_MODE == 0) {
            /* ---- stack walk in slices ------------------------------------------------------------------------------
             * Walk lengths within a wave differ wildly (final_scene: 40 node visits per segment on average, several hundred
             * through the 1000-sphere cluster), and a wave that runs every walk to its end takes as long as its longest
             * lane: 122 wave-steps per segment against 40 (tools/walk_sim.cpp).  So the walk is suspended once RT_SLICE_IDLE(Cfg)
             * lanes have finished theirs: those lanes shade, start their next segment (or sample, or work item) and come
             * back with a fresh walk while the long walkers resume -- their stack is in LDS already, and the walk's rays are
             * rebuilt from the path's ray and the innermost wrapper by the operations that produced them (what leaving a
             * wrapper does anyway, rt_walk_exit).  A lane's own operations and their order do not change: same bits. */
            if (!walking) {
                segs += path.depth_left != 0u ? 1ull : 0ull;
                w_best_t = RT_INF; w_best_prim = RT_NONE; w_best_scope = RT_NONE; w_scope = RT_NONE;
                if (path.depth_left != 0u) {
                    stk.push(sc.root); walking = true;
#if RT_MEDIA_PREFILL
                    /* a ConstantMedium draws inside the walk (constant_medium.rs:85) -- for two or three lanes of a wave at a time, and a
                     * draw that finds the buffer empty generates a Philox block there.  Generate it here, where every starting lane
                     * takes part; the word stream is the same (blocks are only made earlier) */
                    if (Cfg::media) rt_rng_fill(path.rng);
#endif
                }
            }
            RtTrace tr;
            tr.t = RT_R(0.0); tr.prim = RT_NONE; tr.scope = RT_NONE; tr.cls = RT_CLS_TERMINAL;
            if (walking) {
                RtWalk k;
                k.w.o = path.ray.o; k.w.d = path.ray.d;
                k.inv_w = rt_inv3(k.w.d);
                k.time = path.ray.time; k.t_min = RT_R(0.001); k.tmin_nan = false; k.base = 0;
                k.best_t = w_best_t; k.best_prim = w_best_prim; k.best_scope = w_best_scope; k.scope = w_scope;
                if (Cfg::scope_depth == 0 || w_scope == RT_NONE) { k.cur = k.w; k.inv = k.inv_w; }
                else { k.cur = rt_ray_in_scope(sc.nodes, w_scope, k.w); k.inv = rt_inv3(k.cur.d); }
                RT_STAMP(7); /* bucket 7 here: rebuilding the walk's rays */
                const uint32_t lanes_here = (uint32_t)__popcll(__ballot(1));
                const uint32_t stop_at = lanes_here > (uint32_t)RT_SLICE_IDLE(Cfg) ? lanes_here - (uint32_t)RT_SLICE_IDLE(Cfg) : 0u;
/* box-only steps on the same vote (rt_walk_box_step; RT_SLICE_BOX_RUN: the same steps with the top of the stack in a register).  A macro,
 * not a lambda: wrapped in one, the same statements compiled to a kernel 13 % slower on final_scene (register allocation). */
#define RT_SLICE_BOX_STEPS_HERE()                                                                                                   \
    if constexpr (RT_SLICE_BOX_STEPS(Cfg) > 0 && RT_SLICE_BOX_RUN && (!Cfg::ordered || RT_SLICE_BOX_RUN_ORDERED)) {                 \
        rt_walk_box_run<Cfg, RT_SLICE_BOX_STEPS(Cfg)>(ns, k, stk);                                                                  \
    } else if constexpr (RT_SLICE_BOX_STEPS(Cfg) > 0) {                                                                             \
        bool between_boxes = true; /* until the lane's next entry is something else: it then waits for the next full step */       \
        for (int extra = 0; extra < RT_SLICE_BOX_STEPS(Cfg); ++extra) {                                                             \
            const bool go = between_boxes && !rt_walk_done(k, stk);                                                                 \
            if (RT_SLICE_BOX_MIN_LANES > 0 && (uint32_t)__popcll(__ballot(go)) < (uint32_t)RT_SLICE_BOX_MIN_LANES) break;           \
            if (go) between_boxes = rt_walk_box_step<Cfg>(ns, k, stk);                                                              \
        }                                                                                                                           \
        if constexpr (RT_SLICE_PRIM_STEPS > 0) { /* then one primitive-only step and more box steps behind it */                    \
            if (!rt_walk_done(k, stk)) between_boxes = rt_walk_prim_step<Cfg>(sc, ns, k, stk) || between_boxes;                     \
            for (int extra = 0; extra < RT_SLICE_PRIM_STEPS; ++extra)                                                               \
                if (between_boxes && !rt_walk_done(k, stk)) between_boxes = rt_walk_box_step<Cfg>(ns, k, stk);                      \
        }                                                                                                                           \
    }
                for (;;) {
                    const bool more = !rt_walk_done(k, stk);
                    if (__popcll(__ballot(more)) <= stop_at) break; /* wave-uniform: enough lanes are done (or all) */
                    if (more) rt_walk_step<Cfg, true>(sc, ns, k, path.rng, stk);
                    /* a second step on the same vote where a step is cheap (measured: random_scene +1.8 %; final_scene, whose steps
                     * can be a whole medium, -4 %) */
                    if constexpr (RT_SLICE_TWO_STEPS(Cfg)) { if (!rt_walk_done(k, stk)) rt_walk_step<Cfg, true>(sc, ns, k, path.rng, stk); }
                    RT_SLICE_BOX_STEPS_HERE()
                    if constexpr (Cfg::media && RT_SLICE_HEAVY_EVERY > 1) {
                        /* then a round whose full step leaves media alone (rt_walk_light_step) */
                        const bool more2 = !rt_walk_done(k, stk);
                        if (__popcll(__ballot(more2)) <= stop_at) break;
                        if (more2) rt_walk_light_step<Cfg>(sc, ns, k, stk);
                        RT_SLICE_BOX_STEPS_HERE()
                    }
                }
#undef RT_SLICE_BOX_STEPS_HERE
                RT_STAMP(2);
                if (rt_walk_done(k, stk)) {
                    walking = false;
                    tr.t = k.best_t; tr.prim = k.best_prim; tr.scope = k.best_scope;
                    if (tr.prim != RT_NONE) {
                        const uint32_t mk = RT_MAT_KINDF(ns.hot(tr.prim).mat) & 0xFFu;
                        tr.cls = mk == RT_MAT_LAMBERTIAN ? RT_CLS_LAMBERT : mk == RT_MAT_DIELECTRIC ? RT_CLS_DIELECTRIC
                               : mk == RT_MAT_METAL ? RT_CLS_METAL : mk == RT_MAT_ISOTROPIC ? RT_CLS_OTHER : RT_CLS_TERMINAL;
                    }
                } else {
                    w_best_t = k.best_t; w_best_prim = k.best_prim; w_best_scope = k.best_scope; w_scope = k.scope;
                }
            }
            if (!walking) {
                path.radiance = rt_v3(RT_R(0.0), RT_R(0.0), RT_R(0.0)); /* written by the terminal only: not carried across the walk */
                rt_path_shade<Cfg>(sc, path, tr);
                if (!path.alive) {
                    sum = rt_v3d_add(sum, path.radiance);
                    ++s;
                }
            }
        } else {
            segs += path.depth_left != 0u ? 1ull : 0ull;
            rt_path_step<Cfg>(sc, ns, path, stk);
            if (!path.alive) {
                sum = rt_v3d_add(sum, path.radiance); /* pixel_color += ray_color(..), main.rs:972-989 */
                ++s;
            }
        }
        RT_STAMP(6);
    }
    if (segs) atomicAdd(&counters[1], segs);
#ifdef RT_STAMPS
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 16; ++k) atomicAdd(&g_stamp_total[k], rt_stamp_acc[threadIdx.x >> 6][k]);
#endif
}


/* ---- sliced stack walk + workgroup-level reordering of the paths whose walk has ended ("ss") ---------------------------------
 * PMC of the final_scene kernel: 18 600 VALU instructions per wave and 64 segments, of which the walk by itself needs 8 500
 * (the lab's W0c): more than half are shading, hit records and regeneration, run by each wave for the ~40 lanes that left the
 * slice -- every material's, every texture's and every primitive kind's code for two or three lanes each (a Perlin-textured
 * sphere: 7 octaves of noise for one lane).  The reordering kernel of the small scenes (rt_kernel_sorted.h) cures exactly that,
 * but it traces every walk to its end between two barriers.  Here the two are combined: every wave walks its slice on its own
 * (no barrier inside the walk), and at the END of a slice the workgroup sorts the paths whose walk is over by what they do next
 * (Lambertian with a solid colour / Lambertian with a texture / dielectric / metal / isotropic / terminal / retired) and hands
 * each to a lane that is free -- the lanes still walking keep their paths, their walk state and their stack columns.  The sorted
 * rank r of a finished path and the index q of a free lane both run from 0 to the number of finished paths: a finished lane
 * writes its state to slot r of the exchange buffer, a free lane reads slot q.  A path is a pure function of its state, so the
 * frame is bit-identical to the plain kernel's (the GPU suite compares with the CPU build of the core). */
/* CAP = stack entries per lane, PARTS = rounds of the exchange.  Built as (32, 3): 32 KB of stacks + 9 qwords x 256 paths = 18 KB, three
 * workgroups per CU, six barriers per slice; (16, 2) -- 16 KB + 27 KB, four barriers, for walks of at most 16 entries -- measured the
 * same (306 against 306 Mpaths/s): only the first barrier of a slice waits for anything. */
#ifndef RT_SS_IDLE
/* finished walks per wave (RT_SS_WG_SLICE: per workgroup, x 4) that end a slice.  Measured, kernel Mpaths/s: final_scene 800x800x100 32: 302,
 * 36: 304, 40: 306-309, 44: 304, 48: 302, 56: 282 (the plain kernel: 273); random_scene 1200x800x100 (pair walk) 32: 842, 40: 907, 48: 933,
 * 56: 925 (the plain kernel: 878)
 * Again with round 4's tree, work items and compiler options (profiles/r04_ab.txt): random_scene 48: 1285, 52: 1297, 56: 1297, 60: 1280, 62: 1234;
 * final_scene 36: 399.5, 40: 400.4, 44: 397.7 */
#define RT_SS_IDLE(Cfg) (Cfg::media ? 40 : 54)
#endif
#ifndef RT_SS_BOX_STEPS
/* box-only steps per round of these kernels: with the slice ended for the whole workgroup a shorter round pays (the stop is noticed sooner).
 * final_scene 800x800x100, kernel Mpaths/s: register form (V3) 4: 307-309, 3: 314-319, 2: 312-318; one-step form (V4, SAH + near-far) 4: 331,
 * 3: 337, 2: 346 */
#define RT_SS_BOX_STEPS(Cfg) (Cfg::media ? (Cfg::ordered ? 2 : 3) : 0)
#endif
/* The workgroup's counter is read while other waves add to it -- a hint, not a synchronisation: a stale value only delays the stop by a
 * round, and the loop ends by itself when the wave's own walks are over.  By the cycle stamps the waves still spend 18 % of their time at the sort's barrier
 * (final_scene; 10 % on random_scene): measured and NOT the cure -- ending the slice as soon as any wave has no walk left (302 against
 * 304 Mpaths/s), or only after every wave has passed 2 / 4 / 8 checks in it (312-318 against 316-318); profiles/r03_slice_sort.txt. */
/* Round 4: the counter is ONE word per slice that every wave adds its newly finished walks to (a relaxed workgroup-scope atomic add by
 * the wave's first walking lane, `ds_add_u32`) and reads back with a relaxed atomic load: the same hint with defined behaviour in the
 * HIP memory model (round 3 had four plain words written by one wave and read by the others with only a compiler fence between). */
#ifndef RT_SS_ATOMIC_COUNTER
#define RT_SS_ATOMIC_COUNTER 1 /* 0: round 3's form (A/B only): one plain word per wave, read by the others with a compiler fence in between */
#endif
#if RT_SS_ATOMIC_COUNTER
#define RT_SS_PUBLISH(done_now)                                                                                                   \
    do {                                                                                                                          \
        const uint32_t d_ = (done_now);                                                                                           \
        if (d_ != ss_pub) {                                                                                                       \
            if (lane == lead_) __hip_atomic_fetch_add(&ss_done[parity], d_ - ss_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            ss_pub = d_;                                                                                                          \
        }                                                                                                                         \
    } while (0)
#define RT_SS_TOTAL() __hip_atomic_load(&ss_done[parity], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#else
#define RT_SS_PUBLISH(done_now) do { if (lane == lead_) ss_done4[parity][wave] = (done_now); asm volatile("" ::: "memory"); } while (0)
#define RT_SS_TOTAL() (ss_done4[parity][0] + ss_done4[parity][1] + ss_done4[parity][2] + ss_done4[parity][3])
#endif
#ifndef RT_SS_WG_SLICE
#define RT_SS_WG_SLICE 1 /* the slice ends for the whole workgroup at once: every wave publishes how many of its walks have ended and all stop
                            when the workgroup's total reaches 4 x RT_SS_IDLE -- the waves then reach the sort's barrier within a round of each other.
                            Measured: with every wave ending its own slice (0) the reordering only ties the plain kernel (272 against 273 Mpaths/s:
                            31 % fewer VALU instructions, but 69 % of the wave cycles waiting instead of 50 %); with this, 306-309 */
#endif
#define RT_SS_KEYS 7u /* Lambertian with a solid colour, Lambertian with a texture, dielectric, metal, isotropic, terminal, retired (splitting the first by
                         the kind of primitive hit, rect or not: measured, no difference) */
/* PW: the walk is the pair walk of sphere scenes (rt_walk_pair.h; CAP = its stack entries per lane), else the one-entry-per-step walk */
/* HC > 0: the walk reads the scene's WALK TABLE (rt_walk_table.h; the context keeps it behind the node array, RT_WT_OFFSET) and keeps its
 * first HC records -- the most visited nodes -- in LDS: final_scene's 256 most visited nodes take 70 % of the visits, and a 64-byte
 * record read from LDS costs an eighth of what the four divergent 16-byte lane-loads cost the vector L1, which paces this walk. */
template <class Cfg, int CAP, int PARTS, bool PW = false, int HC = 0>
__device__ __forceinline__ void rt_render_ss_body(const RtSceneView& sc, const RtFrame& f, rt_f64* __restrict__ partial,
                                                  unsigned long long* __restrict__ counters, const RtPwView* pwp = nullptr) {
    static_assert(!Cfg::sweep && RT_WALK_MODE == 0, "stack-walk variants only");
    static_assert(!(PW && HC > 0), "the pair walk has its own records");
    constexpr int RT_SS_PER = (RT_XCH_QW + PARTS - 1) / PARTS;
    __shared__ uint32_t stack_mem[PW ? 1 : CAP * RT_BLOCK];
    __shared__ RtNodeHot hc_mem[HC > 0 ? HC : 1];
#if defined(RT_HAVE_PW)
    __shared__ uint32_t pw_ref[PW ? CAP * RT_BLOCK : 1];
    __shared__ float pw_ent[PW ? CAP * RT_BLOCK : 1];
    __shared__ uint32_t pw_q[PW ? RT_PW_QCAP * RT_BLOCK : 1];
    RtPwLds<RT_BLOCK> pwm;
    pwm.ref = pw_ref + threadIdx.x; pwm.ent = pw_ent + threadIdx.x; pwm.q = pw_q + threadIdx.x;
    RtPwLane pwl; pwl.cur = RT_PW_NONE; pwl.qh = 0u; pwl.qn = 0u; pwl.sp = 0;
#else
    static_assert(!PW, "no pair walk in this build");
#endif
    __shared__ unsigned long long xch[RT_SS_PER * RT_BLOCK];
    __shared__ uint32_t cnt[4][RT_SS_KEYS + 1u]; /* [wave][key]; last column: lanes of the wave that are not walking */
    __shared__ uint32_t ss_done[2];              /* [slice parity]: walks of the workgroup that have ended in this slice */
    uint32_t parity = 0u, ss_pub = 0u;           /* ss_pub: what this wave has added to ss_done[parity] so far (wave-uniform) */
    if (threadIdx.x < 2u) ss_done[threadIdx.x] = 0u;
#if !RT_SS_ATOMIC_COUNTER
    __shared__ uint32_t ss_done4[2][4];
    if (threadIdx.x < 8u) ss_done4[threadIdx.x >> 2][threadIdx.x & 3u] = 0u;
#endif
    __syncthreads();
    LdsStack stk;
    stk.base = stack_mem + threadIdx.x;
    stk.sp = 0;
    RtGlobalNodes ns;
    ns.p = sc.nodes;
    typename std::conditional<(HC > 0), RtWalkNodes, RtGlobalNodes>::type wn; /* where the WALK reads its records */
    if constexpr (HC > 0) {
        const RtNodeHot* table = reinterpret_cast<const RtNodeHot*>(reinterpret_cast<const unsigned char*>(sc.nodes) + RT_WT_OFFSET(sc.n_nodes));
        const uint32_t nc = sc.n_nodes < (uint32_t)HC ? sc.n_nodes : (uint32_t)HC;
        const uint4* src = reinterpret_cast<const uint4*>(table);
        uint4* dst = reinterpret_cast<uint4*>(hc_mem);
        for (uint32_t i = threadIdx.x; i < nc * 4u; i += RT_BLOCK) dst[i] = src[i];
        __syncthreads();
        wn.lds = hc_mem; wn.glob = table; wn.nc = nc;
    } else {
        wn.p = sc.nodes;
    }
    const uint32_t walk_root = HC > 0 ? 0u : sc.root; /* the root's walk id is 0 */
    const unsigned long long n_items = rt_item_count(f);
    const unsigned long long npix = (unsigned long long)f.tile_w * f.tile_h;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long item = (unsigned long long)blockIdx.x * RT_BLOCK + threadIdx.x;
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
    bool walking = false;
    double w_best_t = RT_INF;
    uint32_t w_best_prim = RT_NONE, w_best_scope = RT_NONE, w_scope = RT_NONE;
#ifdef RT_STAMPS
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 16; ++k) rt_stamp_acc[threadIdx.x >> 6][k] = 0;
        rt_stamp_last[threadIdx.x >> 6] = __builtin_amdgcn_s_memtime();
    }
#endif
    for (;;) {
        RT_STAMP(6);
        /* 1. regeneration: next sample of the lane's item, or a new item (lanes between two walks only) */
        if (!walking && !path.alive && !retired) {
            const uint32_t s_end = chunk * f.chunk + f.chunk < f.spp ? chunk * f.chunk + f.chunk : f.spp;
            if (have && s == s_end) {
                rt_f64* dst = partial + ((unsigned long long)chunk * npix + (unsigned long long)py * f.tile_w + px) * 3ull;
                dst[0] = sum.x; dst[1] = sum.y; dst[2] = sum.z;
                have = false;
            }
            while (!have) {
                if (!fresh) {
                    const unsigned long long need = __ballot(1);
                    const uint32_t cntn = (uint32_t)__popcll(need);
                    const uint32_t rank = lane_prefix(need);
                    unsigned long long base_item = 0;
                    if (rank == 0u) base_item = atomicAdd(&counters[0], (unsigned long long)cntn);
                    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_item);
                    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_item >> 32));
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
#if !RT_CAM_AT_USE
            else rt_path_begin(sc, f, f.x0 + px, rt_frame_row(f, py), f.sample_offset + s, path);
#else
            else {
                /* the camera block is read where it is used, by scalar loads from the kernel-argument segment: held across the loop its
                 * 24 doubles cost 48 SGPRs the walk needs (they were spilled to VGPR lanes and to scratch) */
                typedef const __attribute__((address_space(4))) double* kd_t;
                kd_t kc = (kd_t)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(RtSceneView, camera));
                asm volatile("" : "+s"(kc));
                RtCamera cam;
                cam.origin = rt_v3(kc[0], kc[1], kc[2]); cam.lower_left_corner = rt_v3(kc[3], kc[4], kc[5]);
                cam.horizontal = rt_v3(kc[6], kc[7], kc[8]); cam.vertical = rt_v3(kc[9], kc[10], kc[11]);
                cam.u = rt_v3(kc[12], kc[13], kc[14]); cam.v = rt_v3(kc[15], kc[16], kc[17]); cam.w = rt_v3(kc[18], kc[19], kc[20]);
                cam.lens_radius = kc[21]; cam.time0 = kc[22]; cam.time1 = kc[23];
                rt_path_begin_cam(cam, f, f.x0 + px, rt_frame_row(f, py), f.sample_offset + s, path);
            }
#endif
        }
        RT_STAMP(1);
        RtTrace tr;
        tr.t = RT_R(0.0); tr.prim = RT_NONE; tr.scope = RT_NONE; tr.cls = retired ? RT_CLS_IDLE : RT_CLS_TERMINAL;
        uint32_t key = retired ? 6u : 5u;
#if defined(RT_HAVE_PW)
        if constexpr (PW) {
            /* 2. + 3. the pair walk in slices (the plain kernel's loop, rt_render_plain_body), the slice ended for the whole workgroup */
            const RtPwView& pw = *pwp;
            if (!walking && !retired) {
                segs += path.depth_left != 0u ? 1ull : 0ull;
                w_best_t = RT_INF; w_best_prim = RT_NONE;
                if (path.depth_left != 0u) {
                    const double frac0 = (path.ray.time - pw.ms_time0) / (pw.ms_time1 - pw.ms_time0);
                    if (rt_isnan(frac0)) { pwl.sp = 0; pwl.qn = 0u; pwl.qh = 0u; pwl.cur = RT_PW_NONE; w_scope = 1u; walking = true; } /* w_scope 1: redo marker */
                    else { w_scope = RT_NONE; walking = rt_pw_begin(pw, pwl, pwm, path.ray.o, rt_inv3(path.ray.d), RT_R(0.001)); }
                }
            }
            if (walking) {
                const RtV3 o = path.ray.o, d = path.ray.d, inv = rt_inv3(d);
                const double frac = (path.ray.time - pw.ms_time0) / (pw.ms_time1 - pw.ms_time0);
                double best_t = w_best_t;
                uint32_t best_prim = w_best_prim;
                bool bad = w_scope != RT_NONE;
                const uint32_t lanes_here = (uint32_t)__popcll(__ballot(1));
                const uint32_t stop_at = lanes_here > (uint32_t)RT_SS_IDLE(Cfg) ? lanes_here - (uint32_t)RT_SS_IDLE(Cfg) : 0u;
                const uint32_t lead_ = (uint32_t)__ffsll((long long)__ballot(1)) - 1u;
                (void)stop_at; (void)lead_;
                for (;;) {
                    const bool more = !bad && !rt_pw_done(pwl);
                    bool on_;
                    {
                        const uint32_t n_ = (uint32_t)__popcll(__ballot(more));
                        if constexpr (RT_SS_WG_SLICE) {
                            RT_SS_PUBLISH(lanes_here - n_);
                            on_ = n_ != 0u && RT_SS_TOTAL() < 4u * (uint32_t)RT_SS_IDLE(Cfg);
                        } else on_ = n_ > stop_at;
                    }
                    if (!on_) break;
                    const bool can_box = more && rt_pw_can_box(pwl), can_leaf = more && pwl.qn > 0u;
                    const uint32_t nb = (uint32_t)__popcll(__ballot(can_box)), nl = (uint32_t)__popcll(__ballot(can_leaf));
                    if (nl >= RT_PW_VOTES || nb == 0u) {
                        if (can_leaf) {
                            rt_pw_group_step(pw, pwl, pwm, o, d, inv, frac, RT_R(0.001), best_t, best_prim);
#pragma unroll
                            for (int extra = 1; extra < RT_PW_LEAF_STEPS; ++extra)
                                if (pwl.qn > 0u) rt_pw_group_step(pw, pwl, pwm, o, d, inv, frac, RT_R(0.001), best_t, best_prim);
                            if (rt_isnan(best_t)) bad = true;
                        }
                    } else if (can_box) {
                        rt_pw_box_step(pw, pwl, pwm, o, inv, RT_R(0.001), best_t);
#pragma unroll
                        for (int extra = 1; extra < RT_PW_BOX_STEPS; ++extra)
                            if (rt_pw_can_box(pwl)) rt_pw_box_step(pw, pwl, pwm, o, inv, RT_R(0.001), best_t);
                    }
                }
                if (bad) {
                    /* the closest hit turned NaN: the segment is redone by the one-entry-per-step walk on the same LDS words */
                    LdsStack cs;
                    cs.base = pw_ref + threadIdx.x; cs.sp = 0;
                    uint32_t scope_;
                    rt_traverse_stack<Cfg, true>(sc, ns, sc.root, path.ray, RT_R(0.001), RT_INF, path.rng, cs, best_t, best_prim, scope_);
                    pwl.cur = RT_PW_NONE; pwl.sp = 0; pwl.qn = 0u; w_scope = RT_NONE;
                }
                if (rt_pw_done(pwl)) {
                    walking = false;
                    tr.t = best_t; tr.prim = best_prim; tr.scope = RT_NONE;
                    if (tr.prim != RT_NONE) {
                        const uint32_t kf = RT_MAT_KINDF(ns.hot(tr.prim).mat), mk = kf & 0xFFu;
                        tr.cls = mk == RT_MAT_LAMBERTIAN ? RT_CLS_LAMBERT : mk == RT_MAT_DIELECTRIC ? RT_CLS_DIELECTRIC
                               : mk == RT_MAT_METAL ? RT_CLS_METAL : mk == RT_MAT_ISOTROPIC ? RT_CLS_OTHER : RT_CLS_TERMINAL;
                        key = mk == RT_MAT_LAMBERTIAN ? ((kf & RT_MAT_SOLID) ? 0u : 1u) : mk == RT_MAT_DIELECTRIC ? 2u
                            : mk == RT_MAT_METAL ? 3u : mk == RT_MAT_ISOTROPIC ? 4u : 5u;
                    }
                } else {
                    w_best_t = best_t; w_best_prim = best_prim;
                }
            }
        } else
#endif
        {
        /* 2. a path between two walks starts its next segment's walk */
        if (!walking && !retired) {
            segs += path.depth_left != 0u ? 1ull : 0ull;
            w_best_t = RT_INF; w_best_prim = RT_NONE; w_best_scope = RT_NONE; w_scope = RT_NONE;
            if (path.depth_left != 0u) {
                stk.push(walk_root); walking = true;
#if RT_MEDIA_PREFILL
                if (Cfg::media) rt_rng_fill(path.rng);
#endif
            }
        }
        /* 3. one slice of the wave's walks (the plain kernel's loop) */
        if (walking) {
            RtWalk k;
            k.w.o = path.ray.o; k.w.d = path.ray.d;
            k.inv_w = rt_inv3(k.w.d);
            k.time = path.ray.time; k.t_min = RT_R(0.001); k.tmin_nan = false; k.base = 0;
            k.best_t = w_best_t; k.best_prim = w_best_prim; k.best_scope = w_best_scope; k.scope = w_scope;
            if (Cfg::scope_depth == 0 || w_scope == RT_NONE) { k.cur = k.w; k.inv = k.inv_w; }
            else { k.cur = rt_ray_in_scope(sc.nodes, w_scope, k.w); k.inv = rt_inv3(k.cur.d); }
            const uint32_t lanes_here = (uint32_t)__popcll(__ballot(1));
            const uint32_t stop_at = lanes_here > (uint32_t)RT_SS_IDLE(Cfg) ? lanes_here - (uint32_t)RT_SS_IDLE(Cfg) : 0u;
            const uint32_t lead_ = (uint32_t)__ffsll((long long)__ballot(1)) - 1u; /* the first walking lane publishes for the wave */
            (void)stop_at; (void)lead_;
            /* wave-uniform: does the slice go on?  (n = lanes of this wave still walking; a macro, see RT_SLICE_BOX_STEPS_HERE) */
#define RT_SS_GOES_ON(n, out)                                                                                         \
    if constexpr (RT_SS_WG_SLICE) {                                                                                   \
        RT_SS_PUBLISH(lanes_here - (n));                                                                              \
        out = (n) != 0u && RT_SS_TOTAL() < 4u * (uint32_t)RT_SS_IDLE(Cfg);                                            \
    } else out = (n) > stop_at;
#define RT_SS_BOX_STEPS_HERE()                                                                                      \
    if constexpr (RT_SS_BOX_STEPS(Cfg) > 0 && RT_SLICE_BOX_RUN && (!Cfg::ordered || RT_SLICE_BOX_RUN_ORDERED)) { \
        rt_walk_box_run<Cfg, RT_SS_BOX_STEPS(Cfg)>(wn, k, stk);                                                  \
    } else if constexpr (RT_SS_BOX_STEPS(Cfg) > 0) {                                                             \
        bool between_boxes = true;                                                                                  \
        for (int extra = 0; extra < RT_SS_BOX_STEPS(Cfg); ++extra)                                               \
            if (between_boxes && !rt_walk_done(k, stk)) between_boxes = rt_walk_box_step<Cfg>(wn, k, stk);          \
    }
            for (;;) {
                const bool more = !rt_walk_done(k, stk);
                bool on_; { const uint32_t n_ = (uint32_t)__popcll(__ballot(more)); RT_SS_GOES_ON(n_, on_) }
                if (!on_) break;
                if (more) rt_walk_step<Cfg, true>(sc, wn, k, path.rng, stk);
                if constexpr (RT_SLICE_TWO_STEPS(Cfg)) { if (!rt_walk_done(k, stk)) rt_walk_step<Cfg, true>(sc, wn, k, path.rng, stk); }
                RT_SS_BOX_STEPS_HERE()
                if constexpr (Cfg::media && RT_SLICE_HEAVY_EVERY > 1) {
                    const bool more2 = !rt_walk_done(k, stk);
                    { const uint32_t n_ = (uint32_t)__popcll(__ballot(more2)); RT_SS_GOES_ON(n_, on_) }
                    if (!on_) break;
                    if (more2) rt_walk_light_step<Cfg>(sc, wn, k, stk);
                    RT_SS_BOX_STEPS_HERE()
                }
            }
#undef RT_SS_BOX_STEPS_HERE
#undef RT_SS_GOES_ON
            if (rt_walk_done(k, stk)) {
                walking = false;
                tr.t = k.best_t; tr.prim = k.best_prim; tr.scope = k.best_scope;
                if (tr.prim != RT_NONE) {
                    const uint32_t kf = RT_MAT_KINDF(ns.hot(tr.prim).mat), mk = kf & 0xFFu;
                    tr.cls = mk == RT_MAT_LAMBERTIAN ? RT_CLS_LAMBERT : mk == RT_MAT_DIELECTRIC ? RT_CLS_DIELECTRIC
                           : mk == RT_MAT_METAL ? RT_CLS_METAL : mk == RT_MAT_ISOTROPIC ? RT_CLS_OTHER : RT_CLS_TERMINAL;
                    key = mk == RT_MAT_LAMBERTIAN ? ((kf & RT_MAT_SOLID) ? 0u : 1u) : mk == RT_MAT_DIELECTRIC ? 2u
                        : mk == RT_MAT_METAL ? 3u : mk == RT_MAT_ISOTROPIC ? 4u : 5u;
                }
            } else {
                w_best_t = k.best_t; w_best_prim = k.best_prim; w_best_scope = k.best_scope; w_scope = k.scope;
            }
        }
        }
        RT_STAMP(2);
        /* 4. rank of every finished path in (key, wave, lane) order; index of every free lane in (wave, lane) order */
        const bool fin = !walking;
        uint32_t my_rank = 0;
#pragma unroll
        for (uint32_t c = 0; c < RT_SS_KEYS; ++c) {
            const unsigned long long m = __ballot(fin && key == c);
            if (fin && key == c) my_rank = lane_prefix(m);
            if (lane == 0u) cnt[wave][c] = (uint32_t)__popcll(m);
        }
        const unsigned long long free_m = __ballot(fin);
        uint32_t q = lane_prefix(free_m);
        if (lane == 0u) cnt[wave][RT_SS_KEYS] = (uint32_t)__popcll(free_m);
        RT_STAMP(8);
        __syncthreads();
#ifdef RT_STAMPS
        { /* diagnostic builds: the first barrier's wait by why this wave left its slice -- 9: the workgroup's count was reached, 12: the
             wave's own walks ran out first, 13: the wave had no walk to begin with */
            const uint32_t left_ = (uint32_t)__popcll(__ballot(walking));
            const uint32_t had_ = (uint32_t)__popcll(__ballot(!retired));
            RT_STAMP(had_ == 0u ? 13 : (left_ == 0u ? 12 : 9));
        }
#else
        RT_STAMP(9);
#endif
        if constexpr (RT_SS_WG_SLICE) { /* the next slice's counter: nobody touches it before the barriers below */
            if (threadIdx.x == 0u) __hip_atomic_store(&ss_done[parity ^ 1u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#if !RT_SS_ATOMIC_COUNTER
            if (lane == 0u) ss_done4[parity ^ 1u][wave] = 0u;
#endif
            parity ^= 1u; ss_pub = 0u;
        }
        uint32_t dest = my_rank, idle_total = 0;
#pragma unroll
        for (uint32_t c = 0; c < RT_SS_KEYS; ++c) {
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) {
                const uint32_t v = cnt[w][c];
                if (c < key || (c == key && w < wave)) dest += v;
                if (c == RT_SS_KEYS - 1u) idle_total += v;
            }
        }
#pragma unroll
        for (uint32_t w = 0; w < 4u; ++w) { const uint32_t v = cnt[w][RT_SS_KEYS]; if (w < wave) q += v; }
        if (idle_total == (uint32_t)RT_BLOCK) break; /* every lane of the workgroup has retired (uniform; none can be walking) */
        /* 5. the finished paths change lanes */
        {
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
            for (int part = 0; part < PARTS; ++part) {
                if (part) __syncthreads(); /* the previous round's readers are done with the buffer */
                if (fin) {
#pragma unroll
                    for (int i = part * RT_SS_PER; i < (part + 1) * RT_SS_PER && i < RT_XCH_QW; ++i) xch[(i - part * RT_SS_PER) * RT_BLOCK + dest] = st[i];
                }
                __syncthreads();
                if (fin) {
#pragma unroll
                    for (int i = part * RT_SS_PER; i < (part + 1) * RT_SS_PER && i < RT_XCH_QW; ++i) st[i] = xch[(i - part * RT_SS_PER) * RT_BLOCK + q];
                }
            }
            if (fin) {
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
        }
        RT_STAMP(7); /* diagnostic builds: bucket 7 = the exchange here */
        /* 6. shading, coherent within a wave after the sort */
        if (fin && !retired) {
            path.radiance = rt_v3(RT_R(0.0), RT_R(0.0), RT_R(0.0)); /* a path's radiance is written by its terminal only (rt_path_shade): nothing to carry */
            rt_path_shade<Cfg>(sc, path, tr);
            if (!path.alive) {
                sum = rt_v3d_add(sum, path.radiance);
                ++s;
            }
        }
    }
    if (segs) atomicAdd(&counters[1], segs);
#ifdef RT_STAMPS
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 16; ++k) atomicAdd(&g_stamp_total[k], rt_stamp_acc[threadIdx.x >> 6][k]);
#endif
}

#endif
