/* rt_wavefront.h -- the big scenes (stack-walk variants) as a wavefront path tracer.
 *
 * Why: on random_scene / final_scene the persistent megakernel keeps 22 % / 8 % of its lanes busy (PMC).  A SIMT model of
 * the stack walk fed with the real per-segment node sequences (tools/walk_sim.cpp) reproduces that figure and says where it
 * goes: every step of a wave runs the box code AND the primitive code, because some lane is at a box and some lane at a
 * sphere, and the sphere test (two f64 divides and a square root) costs three box tests while ~12 % of the lanes are in it.
 * What the model says helps: (1) run only ONE kind of node per step -- the kind most lanes of the wave are waiting at --
 * and (2) hand a finished lane the next ray at once instead of waiting for the wave's longest walk: 2.4x / 2.7x fewer
 * wave-cycles per segment at an unchanged number of steps.  (1) without (2) adds steps (measured in round 1 inside the
 * megakernel: slower, its walk is latency-bound at 3 waves per SIMD); (2) needs the ray somewhere a lane can fetch it from.
 * Hence this form -- the one the north star sketches: per-ray state in HBM as SoA queues (this is the 128-byte record of
 * ray 56 B, throughput 24 B, generator 24 B, ids 8 B, hit 16 B), compacted with ballot + prefix every bounce,
 * and one bounce = two kernels:
 *   wf_trace  persistent; walk state only (no throughput, no generator unless the scene has media), so it fits 4-5 waves
 *             per SIMD; each lane holds ONE pending stack entry with its node record already fetched; per step the wave votes
 *             for the node kind with the most waiting lanes and only those lanes advance (pop, request the next record);
 *             lanes whose walk ended store (t, prim, scope) and take the next queue slots (coalesced) once 32 are idle.
 *   wf_shade  one thread per queue slot: the megakernel's own rt_path_shade, survivors appended to the other queue
 *             (wave-level compaction: ballot, mbcnt, one atomic per wave), finished samples store their radiance.
 * The bounce loop runs on the device: queue lengths live in a device array, every kernel reads its own, nothing is read
 * back except once every 8 bounces to stop early.
 * A path is the same pure function of its state as in the megakernel -- same core functions, same per-lane order of node
 * visits (the vote only decides WHEN a lane's next visit happens), same generator stream (the buffered block is dropped
 * at a hand-over and regenerated: same words) -- and a pixel's samples are summed in sample order per chunk
 * (wf_chunk_sum), so the frame is bit-identical to the megakernel's (tests/test_gpu_parity.py).
 * MEASURED (profiles/r02_wavefront_*): the trace kernel reaches the model's lane utilisation (59 % / 52 % against the
 * megakernel's 24 % / 9 %) with a VALU that is 38 % / 27 % busy -- the walk is paced by the latency and bookkeeping of a
 * step, which the model does not price -- and the whole form ends at 0.7-0.8x the megakernel (random_scene 384 vs 474,
 * final_scene 119 vs 116 Mpaths/s at 16-24 spp; 379 vs 542 and 116 vs 157 at 100 spp).  Opt-in, not the default.
 */
#ifndef RT_WAVEFRONT_H
#define RT_WAVEFRONT_H

#include "rt_kernel_sorted.h"

/* one queue: component c of slot s at f[c * cap + s] / u[c * cap + s] */
struct WfQueue {
    double* f;
    uint32_t* u;
    unsigned long long cap;
};
enum { WF_OX = 0, WF_OY, WF_OZ, WF_DX, WF_DY, WF_DZ, WF_TIME, WF_BX, WF_BY, WF_BZ, WF_HIT_T, WF_NF };
enum { WU_PID = 0, WU_DEPTH, WU_RNG_BLK, WU_RNG_LEFT, WU_RNG_A0, WU_RNG_A1, WU_RNG_A2, WU_RNG_A3, WU_PRIM, WU_SCOPE, WU_NU };
#define WF_MAX_BOUNCES 64 /* queue-length slots kept on the device; max_depth above this uses the megakernel */

/* generator state across a hand-over: the block generated ahead (B) is dropped -- it is block blk-1 and comes back
 * word for word when it is needed */
__device__ __forceinline__ void wf_rng_store(const RtRng& r, const WfQueue& q, unsigned long long s) {
    q.u[WU_RNG_BLK * q.cap + s] = r.blk - r.bv;
    q.u[WU_RNG_LEFT * q.cap + s] = r.left;
    q.u[WU_RNG_A0 * q.cap + s] = r.a0; q.u[WU_RNG_A1 * q.cap + s] = r.a1;
    q.u[WU_RNG_A2 * q.cap + s] = r.a2; q.u[WU_RNG_A3 * q.cap + s] = r.a3;
}
/* path id = s_local * npix + pixel (pixel row-major in the tile); its stream is keyed like rt_path_begin's */
__device__ __forceinline__ RtRng wf_rng_load(const RtFrame& f, uint32_t s0, uint32_t pid, const WfQueue& q, unsigned long long s) {
    const uint32_t npix = f.tile_w * f.tile_h;
    const uint32_t s_local = pid / npix, pixel = pid - s_local * npix;
    const uint32_t py = pixel / f.tile_w, px = pixel - py * f.tile_w;
    RtRng r = rt_rng_pixel_sample((uint64_t)rt_frame_row(f, py) * f.width + (f.x0 + px), f.sample_offset + s0 + s_local, f.global_seed);
    r.blk = q.u[WU_RNG_BLK * q.cap + s];
    r.left = q.u[WU_RNG_LEFT * q.cap + s];
    r.a0 = q.u[WU_RNG_A0 * q.cap + s]; r.a1 = q.u[WU_RNG_A1 * q.cap + s];
    r.a2 = q.u[WU_RNG_A2 * q.cap + s]; r.a3 = q.u[WU_RNG_A3 * q.cap + s];
    return r;
}

/* device-side bookkeeping of one pass: n[b] = length of the queue traced at bounce b, next[b] = the trace kernel's slot
 * dispenser for that bounce, segs = traced segments */
struct WfCounters {
    unsigned long long n[WF_MAX_BOUNCES + 1];
    unsigned long long next[WF_MAX_BOUNCES + 1];
    unsigned long long segs;
};
__global__ void wf_init_counters(WfCounters* c, unsigned long long n0) {
    const uint32_t i = threadIdx.x;
    if (i <= WF_MAX_BOUNCES) { c->n[i] = i == 0u ? n0 : 0ull; c->next[i] = 0ull; }
    /* segs accumulates over the passes of a render; reset by the host */
}

/* camera rays of samples [s0, s0+s_cnt) of every pixel of the tile into queue slots 0..n-1; path id = slot */
__global__ void wf_generate(RtSceneView sc, RtFrame f, WfQueue q, uint32_t s0, uint32_t s_cnt, double* __restrict__ sample_rad) {
    const unsigned long long npix = (unsigned long long)f.tile_w * f.tile_h;
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= npix * s_cnt) return;
    /* slots in the megakernel's work-item order would need the 8x8 decode; row-major pixels within a sample are coherent
     * enough for camera rays (64 consecutive pixels of a row) */
    const uint32_t s_local = (uint32_t)(gid / npix);
    const uint32_t pixel = (uint32_t)(gid % npix);
    const uint32_t px = pixel % f.tile_w, py = pixel / f.tile_w;
    RtPath p;
    rt_path_begin(sc, f, f.x0 + px, rt_frame_row(f, py), f.sample_offset + s0 + s_local, p);
    if (f.max_depth == 0u) { /* main.rs:59-61 at the camera ray: black, nothing to trace */
        const RtV3 r = p.radiance + rt_mul(p.beta, rt_v3(0.0, 0.0, 0.0));
        sample_rad[gid * 3ull] = r.x; sample_rad[gid * 3ull + 1ull] = r.y; sample_rad[gid * 3ull + 2ull] = r.z;
        return;
    }
    q.f[WF_OX * q.cap + gid] = p.ray.o.x; q.f[WF_OY * q.cap + gid] = p.ray.o.y; q.f[WF_OZ * q.cap + gid] = p.ray.o.z;
    q.f[WF_DX * q.cap + gid] = p.ray.d.x; q.f[WF_DY * q.cap + gid] = p.ray.d.y; q.f[WF_DZ * q.cap + gid] = p.ray.d.z;
    q.f[WF_TIME * q.cap + gid] = p.ray.time;
    q.f[WF_BX * q.cap + gid] = p.beta.x; q.f[WF_BY * q.cap + gid] = p.beta.y; q.f[WF_BZ * q.cap + gid] = p.beta.z;
    q.u[WU_PID * q.cap + gid] = (uint32_t)gid;
    q.u[WU_DEPTH * q.cap + gid] = p.depth_left;
    wf_rng_store(p.rng, q, gid);
}

/* What one visit of the stack walk reads, as one 64-byte-aligned record per node (the context builds the array from
 * the flat nodes): a lane's fetch is four aligned dwordx4 loads inside ONE cache line.  (From the 96-byte RtNode the same
 * visit took seven loads over two lines -- measured: the walk's loads queue up in the vector cache, 52 % of the trace
 * kernel's wave-cycles were spent at s_waitcnt with a 99 % L1 hit rate.)  d[] as in RtNode; a MovingSphere keeps its
 * radius in d[6] and the scene-wide (time0, time1) are kernel arguments: the reference's scenes give every moving sphere
 * the same shutter interval (main.rs:230-237,700-707), so (time - time0) / (time1 - time0) -- the same operands for
 * every sphere a ray meets -- is evaluated once per ray.  Scenes where the intervals differ keep the megakernel. */
struct alignas(64) WfRec {
    double d[7];
    uint32_t kind; /* RtNode.kind */
    uint32_t b;    /* RtNode.b */
};
static_assert(sizeof(WfRec) == 64, "one cache line");
struct WfRecs {
    const WfRec* p;
    double ms_time0, ms_time1;
};

/* the three rect kinds from a walk record: axis by select, then rt_rect_hot_t's own arithmetic (aarect.rs:46-56,84-94,152-162) */
__device__ __forceinline__ bool wf_rect_t(const WfRec& nd, uint32_t kind, RtV3 o, RtV3 d, double t_min, double t_max, double& t_out) {
    const double oa = kind == RT_XY ? o.z : (kind == RT_XZ ? o.y : o.x);
    const double da = kind == RT_XY ? d.z : (kind == RT_XZ ? d.y : d.x);
    const double ob = kind == RT_YZ ? o.y : o.x;
    const double db = kind == RT_YZ ? d.y : d.x;
    const double oc = kind == RT_XY ? o.y : o.z;
    const double dc = kind == RT_XY ? d.y : d.z;
    const double t = (nd.d[4] - oa) / da;
    const bool in_t = !((t < t_min) | (t > t_max));
    const double b = ob + t * db;
    const double c = oc + t * dc;
    const bool in_rect = !((b < nd.d[0]) | (b > nd.d[1]) | (c < nd.d[2]) | (c > nd.d[3]));
    if (in_t & in_rect) { t_out = t; return true; }
    return false;
}

enum { WF_K_BOX = 0, WF_K_SPHERE, WF_K_MSPHERE, WF_K_RECT, WF_K_WRAP, WF_K_EXIT, WF_K_MEDIUM, WF_NK, WF_K_NONE = 15 };

#ifndef RT_WF_REFILL
#define RT_WF_REFILL 32u /* idle lanes of a wave that trigger a refill (the SIMT model is flat between 8 and 32) */
#endif
#ifndef RT_WF_TRACE_WAVES
#define RT_WF_TRACE_WAVES(Cfg, WRAP) ((Cfg::media || WRAP) ? 4 : 5)
#endif

template <int STRIDE>
struct WfStack {
    uint32_t* base; /* this lane's entry 0; entry e at base[e * STRIDE] (bank = lane: conflict-free at any depth) */
    int sp;
    __device__ __forceinline__ void push(uint32_t v) { base[sp * STRIDE] = v; ++sp; }
    __device__ __forceinline__ void poke(int above, uint32_t v) { base[(sp + above) * STRIDE] = v; }
    __device__ __forceinline__ uint32_t pop() { --sp; return base[sp * STRIDE]; }
};
#define RT_WF_LDS_BLOCK 1024 /* workgroup of the LDS-resident form: one per CU, 4 waves per SIMD */
#define RT_WF_LDS_NODES 1024 /* scenes up to this many nodes keep their walk records in LDS (64 KB) */

/* WRAP: the scene has Translate / RotateY / FlipFace wrapper nodes (otherwise the world ray is the only ray space).
 * LDSRECS: the walk records were copied into LDS by the caller (`lrecs`), node fetches are ds_read_b128. */
template <class Cfg, bool WRAP, int BLOCK, bool LDSRECS>
__device__ __forceinline__ void wf_trace_body(const RtSceneView& sc, const RtFrame& f, const WfQueue& q, WfCounters* __restrict__ ctr,
                                              uint32_t bounce, uint32_t s0, const WfRecs& recs, uint32_t* stack_mem, const WfRec* lrecs) {
    WfStack<BLOCK> stk;
    stk.base = stack_mem + threadIdx.x;
    stk.sp = 0;
    const RtGlobalNodes ns{sc.nodes};
    const unsigned long long n = ctr->n[bounce];
    if (n == 0ull) return;
    const unsigned long long total_waves = (unsigned long long)gridDim.x * (BLOCK / 64);
    unsigned long long w_next = 0, w_end = 0; /* wave-uniform: this wave's current batch of queue slots */
    bool exhausted = false;                   /* wave-uniform: the dispenser has nothing left */
    bool have = false;
    bool unsent = false; /* the walk has ended and its result is still in k: written out at the next refill, not in the step
                            loop (a store's acknowledgement would sit in front of every following node fetch in vmcnt order) */
    unsigned long long slot = 0;
    RtWalk k;
    k.t_min = 0.001; k.tmin_nan = false; k.base = 0; /* main.rs:62: world.hit(ray, 0.001, INFINITY) */
    uint32_t e = 0;
    WfRec nd;
    double ms_frac = 0.0; /* (time - time0) / (time1 - time0) of MovingSphere::center, moving_sphere.rs:23-26 */
    RtRng rng = rt_rng_make(0u, 0u, 0u, 0u, 0u);
    unsigned long long segs = 0;

    for (;;) {
        /* ---- refill: idle lanes take the next queue slots (consecutive slots -> coalesced loads) ---- */
        {
            const unsigned long long idle = __ballot(!have);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (!exhausted && n_idle >= RT_WF_REFILL) {
                if (w_next >= w_end) {
                    /* guided self-scheduling: big batches while the queue is long (one atomic per ~1k slots keeps the
                     * single dispenser word far below its ~88 atomics/us), small ones at the end (short tail) */
                    unsigned long long base = 0, batch = 0;
                    if ((threadIdx.x & 63u) == 0u) {
                        const unsigned long long seen = __hip_atomic_load(&ctr->next[bounce], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long remaining = n > seen ? n - seen : 0ull;
                        batch = remaining / (4ull * total_waves);
                        batch = batch > 2048ull ? 2048ull : (batch < 64ull ? 64ull : (batch & ~63ull));
                        base = atomicAdd(&ctr->next[bounce], batch);
                    }
                    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
                    const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t)batch);
                    base = ((unsigned long long)bhi << 32) | blo;
                    w_next = base < n ? base : n;
                    w_end = base + nb < n ? base + nb : n;
                    if (w_next >= w_end) exhausted = true;
                }
                if (unsent) {
                    q.f[WF_HIT_T * q.cap + slot] = k.best_t;
                    q.u[WU_PRIM * q.cap + slot] = k.best_prim;
                    q.u[WU_SCOPE * q.cap + slot] = k.best_scope;
                    if (Cfg::media) wf_rng_store(rng, q, slot);
                    unsent = false;
                }
                const unsigned long long avail = w_end - w_next;
                if (avail != 0ull) {
                    const uint32_t rank = lane_prefix(idle);
                    const uint32_t take = n_idle < avail ? n_idle : (uint32_t)avail;
                    if (!have && rank < take) {
                        slot = w_next + rank;
                        RtRay ray;
                        ray.o = rt_v3(q.f[WF_OX * q.cap + slot], q.f[WF_OY * q.cap + slot], q.f[WF_OZ * q.cap + slot]);
                        ray.d = rt_v3(q.f[WF_DX * q.cap + slot], q.f[WF_DY * q.cap + slot], q.f[WF_DZ * q.cap + slot]);
                        ray.time = q.f[WF_TIME * q.cap + slot];
                        if (Cfg::media) rng = wf_rng_load(f, s0, q.u[WU_PID * q.cap + slot], q, slot); /* ConstantMedium draws while being traversed */
                        /* rt_walk_begin, with the root as the pending entry instead of a push */
                        k.w.o = ray.o; k.w.d = ray.d; k.cur = k.w;
                        k.inv_w = rt_inv3(k.w.d); k.inv = k.inv_w;
                        k.time = ray.time; k.best_t = RT_INF;
                        k.scope = RT_NONE; k.best_prim = RT_NONE; k.best_scope = RT_NONE;
                        stk.sp = 0;
                        if (Cfg::msphere) ms_frac = (ray.time - recs.ms_time0) / (recs.ms_time1 - recs.ms_time0);
                        e = sc.root;
                        nd = LDSRECS ? lrecs[e] : recs.p[e];
                        have = true;
                        segs += 1ull;
                    }
                    w_next += take;
                }
            }
            if (!RT_WAVE_ANY(have)) {
                if (exhausted) break;
                continue;
            }
        }

        /* ---- vote: the node kind most lanes are waiting at ---- */
        uint32_t cls = WF_K_NONE;
        if (have) {
            if (WRAP && (e & RT_POP_FLAG)) cls = WF_K_EXIT;
            else {
                const uint32_t kind = nd.kind & RT_KIND_MASK;
                cls = kind <= RT_BVH1 ? WF_K_BOX : kind == RT_SPHERE ? WF_K_SPHERE : kind == RT_MSPHERE ? WF_K_MSPHERE
                    : kind <= RT_YZ ? WF_K_RECT : kind <= RT_FLIP ? WF_K_WRAP : WF_K_MEDIUM;
            }
        }
        uint32_t pick = 0, best_cnt = 0;
#pragma unroll
        for (uint32_t c = 0; c < WF_NK; ++c) {
            if ((c == WF_K_MSPHERE && !Cfg::msphere) || ((c == WF_K_WRAP || c == WF_K_EXIT) && !WRAP) || (c == WF_K_MEDIUM && !Cfg::media)) continue;
            const uint32_t cnt = (uint32_t)__popcll(__ballot(cls == c));
            if (cnt > best_cnt) { best_cnt = cnt; pick = c; }
        }
        const bool act = cls == pick; /* best_cnt >= 1: some lane has work */
        bool do_pop = true;           /* after its visit a lane either pops or steps into the next node in pre-order */

        if (pick == WF_K_BOX) {
            /* BVHNode::hit bvh.rs:25-50 up to the recursion: own box, then left child now, right child later */
            if (act) {
                bool hit;
                if (RT_WAVE_ANY(rt_isnan(k.best_t))) hit = rt_aabb_hit(nd.d, k.cur.o, k.inv, k.t_min, k.best_t);
                else hit = rt_aabb_hit_fast<false>(nd.d, k.cur.o, k.inv, k.t_min, k.best_t);
                if (hit) {
                    uint32_t first = e + 1u;
                    if ((nd.kind & RT_KIND_MASK) == RT_BVH2) {
                        uint32_t second = nd.b;
                        const uint32_t ord = Cfg::ordered ? (nd.kind >> RT_BVH_ORDER_SHIFT) & RT_BVH_ORDER_MASK : 0u; /* opt-in near-far order, as rt_walk_box */
                        if (Cfg::ordered && ord != 0u) {
                            const double da = ord == 1u ? k.cur.d.x : (ord == 2u ? k.cur.d.y : k.cur.d.z);
                            const bool left_lower = (nd.kind & RT_BVH_LEFT_LOWER) != 0u;
                            if ((da < 0.0 && left_lower) || (da > 0.0 && !left_lower)) { first = nd.b; second = e + 1u; }
                        }
                        stk.push(second);
                    }
                    e = first;
                    do_pop = false;
                }
            }
        } else if (pick == WF_K_SPHERE) {
            if (act) {
                double t;
                if (rt_sphere_root(rt_v3(nd.d[0], nd.d[1], nd.d[2]), nd.d[3], k.cur.o, k.cur.d, k.t_min, k.best_t, t) && (!Cfg::ordered || rt_tie_ok(t, e, k.best_t, k.best_prim))) {
                    k.best_t = t; k.best_prim = e; k.best_scope = k.scope;
                }
            }
        } else if (Cfg::msphere && pick == WF_K_MSPHERE) {
            if constexpr (Cfg::msphere) {
                if (act) {
                    /* MovingSphere::hit moving_sphere.rs:31-70; center(time) = c0 + frac * (c1 - c0) with the ray's frac */
                    const RtV3 c0 = rt_v3(nd.d[0], nd.d[1], nd.d[2]), c1 = rt_v3(nd.d[3], nd.d[4], nd.d[5]);
                    const RtV3 center = c0 + ms_frac * (c1 - c0);
                    double t;
                    if (rt_sphere_root(center, nd.d[6], k.cur.o, k.cur.d, k.t_min, k.best_t, t) && (!Cfg::ordered || rt_tie_ok(t, e, k.best_t, k.best_prim))) {
                        k.best_t = t; k.best_prim = e; k.best_scope = k.scope;
                    }
                }
            }
        } else if (pick == WF_K_RECT) {
            if (act) {
                double t;
                if (wf_rect_t(nd, nd.kind & RT_KIND_MASK, k.cur.o, k.cur.d, k.t_min, k.best_t, t) && (!Cfg::ordered || rt_tie_ok(t, e, k.best_t, k.best_prim))) {
                    k.best_t = t; k.best_prim = e; k.best_scope = k.scope;
                }
            }
        } else if (WRAP && pick == WF_K_WRAP) {
            if (act) {
                /* Translate::hit hittable.rs:207-211 / RotateY::hit :238-251 / FlipFace::hit :287 on the way in */
                const uint32_t kind = nd.kind & RT_KIND_MASK;
                stk.push(e | RT_POP_FLAG);
                k.scope = e;
                if (kind != RT_FLIP) {
                    k.cur = rt_scope_in(nd, k.cur);
                    if (kind == RT_ROTATE_Y) k.inv = rt_inv3(k.cur.d);
                }
                e = e + 1u;
                do_pop = false;
            }
        } else if (WRAP && pick == WF_K_EXIT) {
            if (act) rt_walk_exit(sc, k, e);
        } else if (Cfg::media) {
            /* ConstantMedium::hit constant_medium.rs:58-113: the two boundary walks are ordinary (unvoted) walks above this
             * lane's stack level -- rare nodes with tiny boundaries in the reference's scenes */
            if (act) rt_walk_other<Cfg, true>(sc, ns, k, e, ns.hot(e), rng, stk);
        }

        /* ---- advance the lanes that were served ---- */
        if (act) {
            if (do_pop) {
                if (stk.sp == 0) {
                    have = false;
                    unsent = true;
                } else {
                    e = stk.pop();
                }
            }
            if (have && !(WRAP && (e & RT_POP_FLAG))) nd = LDSRECS ? lrecs[e] : recs.p[e]; /* requested now, needed at the next vote */
        }
    }
    if (unsent) {
        q.f[WF_HIT_T * q.cap + slot] = k.best_t;
        q.u[WU_PRIM * q.cap + slot] = k.best_prim;
        q.u[WU_SCOPE * q.cap + slot] = k.best_scope;
        if (Cfg::media) wf_rng_store(rng, q, slot);
    }
    if (segs) atomicAdd(&ctr->segs, segs);
}

template <class Cfg, bool WRAP, int CAP>
__global__ __launch_bounds__(RT_BLOCK, RT_WF_TRACE_WAVES(Cfg, WRAP)) void wf_trace(RtSceneView sc, RtFrame f, WfQueue q, WfCounters* __restrict__ ctr,
                                                                               uint32_t bounce, uint32_t s0, WfRecs recs) {
    __shared__ uint32_t stack_mem[CAP * RT_BLOCK];
    wf_trace_body<Cfg, WRAP, RT_BLOCK, false>(sc, f, q, ctr, bounce, s0, recs, stack_mem, nullptr);
}
/* scenes of <= RT_WF_LDS_NODES nodes: walk records in LDS (64 KB) + 16-entry stacks of 1024 lanes (64 KB), one workgroup per CU */
template <class Cfg, bool WRAP>
__global__ __launch_bounds__(RT_WF_LDS_BLOCK) void wf_trace_lds(RtSceneView sc, RtFrame f, WfQueue q, WfCounters* __restrict__ ctr,
                                                               uint32_t bounce, uint32_t s0, WfRecs recs) {
    __shared__ uint32_t stack_mem[16 * RT_WF_LDS_BLOCK];
    __shared__ WfRec lrecs[RT_WF_LDS_NODES];
    {
        const uint4* src = reinterpret_cast<const uint4*>(recs.p);
        uint4* dst = reinterpret_cast<uint4*>(lrecs);
        for (uint32_t i = threadIdx.x; i < sc.n_nodes * 4u; i += RT_WF_LDS_BLOCK) dst[i] = src[i];
        __syncthreads();
    }
    wf_trace_body<Cfg, WRAP, RT_WF_LDS_BLOCK, true>(sc, f, q, ctr, bounce, s0, recs, stack_mem, lrecs);
}

/* PLAIN TRACE KERNEL (round 3).  The trace-only harness (walk_lab.hip) showed that the product's own one-entry-per-step walk,
 * run by itself -- walk state only, idle lanes refilled from the ray list once 32 wait -- answers 1.9 G rays/s on final_scene's
 * rays and 3.2 G rays/s on random_scene's, where the vote-scheduled kernel above does 0.7 G rays/s and the megakernel's whole
 * step 1.1 / 2.4 G segments/s (profiles/r03_lab_*).  So this is that loop on the queue: per lane rt_walk_begin / rt_walk_step /
 * rt_walk_done of rt_core.h on the 64-byte hot halves of the flat nodes, results written at the refill.  Same visits, same
 * order, same operands per lane as the megakernel's walk: same bits. */
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK, 3) void wf_trace_plain(RtSceneView sc, RtFrame f, WfQueue q, WfCounters* __restrict__ ctr, uint32_t bounce, uint32_t s0,
                                                             WfRecs recs) {
    __shared__ uint32_t stack_mem[RT_STACK_CAP * RT_BLOCK];
    (void)recs;
    LdsStack stk;
    stk.base = stack_mem + threadIdx.x;
    stk.sp = 0;
    const RtGlobalNodes ns{sc.nodes};
    const unsigned long long n = ctr->n[bounce];
    if (n == 0ull) return;
    const unsigned long long total_waves = (unsigned long long)gridDim.x * (RT_BLOCK / 64);
    unsigned long long w_next = 0, w_end = 0;
    bool exhausted = false, have = false, unsent = false;
    unsigned long long slot = 0;
    RtWalk k;
    RtRng rng = rt_rng_make(0u, 0u, 0u, 0u, 0u);
    unsigned long long segs = 0;
    for (;;) {
        {
            const unsigned long long idle = __ballot(!have);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (!exhausted && (n_idle >= RT_WF_REFILL)) {
                if (w_next >= w_end) {
                    unsigned long long base = 0, batch = 0;
                    if ((threadIdx.x & 63u) == 0u) {
                        const unsigned long long seen = __hip_atomic_load(&ctr->next[bounce], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long remaining = n > seen ? n - seen : 0ull;
                        batch = remaining / (4ull * total_waves);
                        batch = batch > 2048ull ? 2048ull : (batch < 64ull ? 64ull : (batch & ~63ull));
                        base = atomicAdd(&ctr->next[bounce], batch);
                    }
                    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
                    const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t)batch);
                    base = ((unsigned long long)bhi << 32) | blo;
                    w_next = base < n ? base : n;
                    w_end = base + nb < n ? base + nb : n;
                    if (w_next >= w_end) exhausted = true;
                }
                if (unsent) {
                    q.f[WF_HIT_T * q.cap + slot] = k.best_t;
                    q.u[WU_PRIM * q.cap + slot] = k.best_prim;
                    q.u[WU_SCOPE * q.cap + slot] = k.best_scope;
                    if (Cfg::media) wf_rng_store(rng, q, slot);
                    unsent = false;
                }
                const unsigned long long avail = w_end - w_next;
                if (avail != 0ull) {
                    const uint32_t rank = lane_prefix(idle);
                    const uint32_t take = n_idle < avail ? n_idle : (uint32_t)avail;
                    if (!have && rank < take) {
                        slot = w_next + rank;
                        RtRay ray;
                        ray.o = rt_v3(q.f[WF_OX * q.cap + slot], q.f[WF_OY * q.cap + slot], q.f[WF_OZ * q.cap + slot]);
                        ray.d = rt_v3(q.f[WF_DX * q.cap + slot], q.f[WF_DY * q.cap + slot], q.f[WF_DZ * q.cap + slot]);
                        ray.time = q.f[WF_TIME * q.cap + slot];
                        if (Cfg::media) rng = wf_rng_load(f, s0, q.u[WU_PID * q.cap + slot], q, slot); /* ConstantMedium draws while being traversed */
                        stk.sp = 0;
                        rt_walk_begin(k, sc.root, ray, 0.001, RT_INF, stk); /* main.rs:62: world.hit(ray, 0.001, INFINITY) */
                        have = true;
                        segs += 1ull;
                    }
                    w_next += take;
                }
            }
            if (!RT_WAVE_ANY(have)) {
                if (exhausted) break;
                continue;
            }
        }
        if (have) {
            rt_walk_step<Cfg, true>(sc, ns, k, rng, stk);
            if (rt_walk_done(k, stk)) { have = false; unsent = true; }
        }
    }
    if (unsent) {
        q.f[WF_HIT_T * q.cap + slot] = k.best_t;
        q.u[WU_PRIM * q.cap + slot] = k.best_prim;
        q.u[WU_SCOPE * q.cap + slot] = k.best_scope;
        if (Cfg::media) wf_rng_store(rng, q, slot);
    }
    if (segs) atomicAdd(&ctr->segs, segs);
}

/* one bounce of shading: grid-stride over the queue's slots (coalesced), survivors appended to `qo` */
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK) void wf_shade(RtSceneView sc, RtFrame f, WfQueue qi, WfQueue qo, WfCounters* __restrict__ ctr, uint32_t bounce,
                                                     uint32_t s0, double* __restrict__ sample_rad) {
    const unsigned long long n = ctr->n[bounce];
    const unsigned long long stride = (unsigned long long)gridDim.x * RT_BLOCK;
    const unsigned long long first = (unsigned long long)blockIdx.x * RT_BLOCK + (threadIdx.x & ~63u); /* this wave's first slot */
    for (unsigned long long wbase = first; wbase < n; wbase += stride) {
        const unsigned long long s = wbase + (threadIdx.x & 63u);
        bool alive = false;
        RtPath p;
        uint32_t pid = 0;
        if (s < n) {
            pid = qi.u[WU_PID * qi.cap + s];
            p.ray.o = rt_v3(qi.f[WF_OX * qi.cap + s], qi.f[WF_OY * qi.cap + s], qi.f[WF_OZ * qi.cap + s]);
            p.ray.d = rt_v3(qi.f[WF_DX * qi.cap + s], qi.f[WF_DY * qi.cap + s], qi.f[WF_DZ * qi.cap + s]);
            p.ray.time = qi.f[WF_TIME * qi.cap + s];
            p.beta = rt_v3(qi.f[WF_BX * qi.cap + s], qi.f[WF_BY * qi.cap + s], qi.f[WF_BZ * qi.cap + s]);
            p.radiance = rt_v3(0.0, 0.0, 0.0); /* only a terminal adds to it (rt_path_shade), and a terminal ends the path */
            p.depth_left = qi.u[WU_DEPTH * qi.cap + s];
            p.alive = true;
            p.rng = wf_rng_load(f, s0, pid, qi, s);
            RtTrace tr;
            tr.t = qi.f[WF_HIT_T * qi.cap + s];
            tr.prim = qi.u[WU_PRIM * qi.cap + s];
            tr.scope = qi.u[WU_SCOPE * qi.cap + s];
            tr.cls = 0u;
            rt_path_shade<Cfg>(sc, p, tr);
            /* the depth budget ran out: the megakernel's next step adds beta (.) 0 without a hit test (main.rs:59-61) */
            if (p.alive && p.depth_left == 0u) rt_path_shade<Cfg>(sc, p, tr);
            alive = p.alive;
            if (!alive) {
                double* dst = sample_rad + (unsigned long long)pid * 3ull;
                dst[0] = p.radiance.x; dst[1] = p.radiance.y; dst[2] = p.radiance.z;
            }
        }
        /* wave-level compaction: ballot + prefix, one atomic per wave */
        const unsigned long long m = __ballot(alive);
        if (m) {
            const uint32_t cnt = (uint32_t)__popcll(m), rank = lane_prefix(m);
            unsigned long long base = 0;
            const uint32_t leader = (uint32_t)(__ffsll((long long)m) - 1);
            if ((threadIdx.x & 63u) == leader) base = atomicAdd(&ctr->n[bounce + 1u], (unsigned long long)cnt);
            const uint32_t lo = __shfl((uint32_t)base, (int)leader), hi = __shfl((uint32_t)(base >> 32), (int)leader);
            if (alive) {
                const unsigned long long o = (((unsigned long long)hi << 32) | lo) + rank;
                qo.f[WF_OX * qo.cap + o] = p.ray.o.x; qo.f[WF_OY * qo.cap + o] = p.ray.o.y; qo.f[WF_OZ * qo.cap + o] = p.ray.o.z;
                qo.f[WF_DX * qo.cap + o] = p.ray.d.x; qo.f[WF_DY * qo.cap + o] = p.ray.d.y; qo.f[WF_DZ * qo.cap + o] = p.ray.d.z;
                qo.f[WF_TIME * qo.cap + o] = p.ray.time;
                qo.f[WF_BX * qo.cap + o] = p.beta.x; qo.f[WF_BY * qo.cap + o] = p.beta.y; qo.f[WF_BZ * qo.cap + o] = p.beta.z;
                qo.u[WU_PID * qo.cap + o] = pid;
                qo.u[WU_DEPTH * qo.cap + o] = p.depth_left;
                wf_rng_store(p.rng, qo, o);
            }
        }
    }
}

/* The long tail: after a few bounces the queue is a small fraction of the pass (path lengths fall off geometrically) and a
 * bounce costs its fixed latency -- one wave walking alone takes ~0.3 ms whatever the queue length.  From bounce
 * RT_WF_BOUNCES on, the surviving paths are finished by this kernel instead: persistent, one path per lane from the queue
 * to its end with the megakernel's own rt_path_step (plain stack walk), the lane taking the next path when its own ends. */
#ifndef RT_WF_BOUNCES
#define RT_WF_BOUNCES 6u
#endif
#ifndef RT_WF_BOUNCES_PLAIN
#define RT_WF_BOUNCES_PLAIN 12u /* with the plain trace kernel (profiles/r03_wavefront_ab.txt: final_scene 155 / 145 / 136 / 109 Mpaths/s with 6 / 12 / 20 / 50 wavefront bounces, random_scene 387 / 346 with 6 / 12) */
#endif
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK, 3) void wf_finish(RtSceneView sc, RtFrame f, WfQueue q, WfCounters* __restrict__ ctr, uint32_t bounce, uint32_t s0,
                                                        double* __restrict__ sample_rad) {
    __shared__ uint32_t stack_mem[RT_STACK_CAP * RT_BLOCK];
    LdsStack stk;
    stk.base = stack_mem + threadIdx.x;
    stk.sp = 0;
    const RtGlobalNodes ns{sc.nodes};
    const unsigned long long n = ctr->n[bounce];
    if (n == 0ull) return;
    RtPath p;
    p.alive = false;
    uint32_t pid = 0;
    bool have = false, out_of_work = false;
    unsigned long long segs = 0;
    for (;;) {
        if (!have && !out_of_work) {
            /* wave-aggregated fetch: the lanes that need a path share one atomic */
            const unsigned long long need = __ballot(1);
            const uint32_t cnt = (uint32_t)__popcll(need), rank = lane_prefix(need);
            unsigned long long base = 0;
            if (rank == 0u) base = atomicAdd(&ctr->next[bounce], (unsigned long long)cnt);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
            const unsigned long long s = (((unsigned long long)hi << 32) | lo) + rank;
            if (s < n) {
                pid = q.u[WU_PID * q.cap + s];
                p.ray.o = rt_v3(q.f[WF_OX * q.cap + s], q.f[WF_OY * q.cap + s], q.f[WF_OZ * q.cap + s]);
                p.ray.d = rt_v3(q.f[WF_DX * q.cap + s], q.f[WF_DY * q.cap + s], q.f[WF_DZ * q.cap + s]);
                p.ray.time = q.f[WF_TIME * q.cap + s];
                p.beta = rt_v3(q.f[WF_BX * q.cap + s], q.f[WF_BY * q.cap + s], q.f[WF_BZ * q.cap + s]);
                p.radiance = rt_v3(0.0, 0.0, 0.0);
                p.depth_left = q.u[WU_DEPTH * q.cap + s];
                p.alive = true;
                p.rng = wf_rng_load(f, s0, pid, q, s);
                have = true;
            } else {
                out_of_work = true;
            }
        }
        if (!RT_WAVE_ANY(have)) break;
        if (have) {
            segs += p.depth_left != 0u ? 1ull : 0ull;
            rt_path_step<Cfg>(sc, ns, p, stk);
            if (!p.alive) {
                double* dst = sample_rad + (unsigned long long)pid * 3ull;
                dst[0] = p.radiance.x; dst[1] = p.radiance.y; dst[2] = p.radiance.z;
                have = false;
            }
        }
    }
    if (segs) atomicAdd(&ctr->segs, segs);
}

/* sum of a pixel's samples of one pass, in sample order, added to the chunk's running sum (the megakernel's
 * `sum = sum + radiance` per sample; `first` starts the chunk at 0.0) */
__global__ void wf_chunk_sum(const double* __restrict__ sample_rad, double* __restrict__ partial_chunk,
                             unsigned long long npix, uint32_t s_cnt, uint32_t first) {
    const unsigned long long pixel = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pixel >= npix) return;
    double* dst = partial_chunk + pixel * 3ull;
    RtV3 sum = first ? rt_v3(0.0, 0.0, 0.0) : rt_v3(dst[0], dst[1], dst[2]);
    for (uint32_t s = 0; s < s_cnt; ++s) {
        const double* r = sample_rad + ((unsigned long long)s * npix + pixel) * 3ull;
        sum = sum + rt_v3(r[0], r[1], r[2]);
    }
    dst[0] = sum.x; dst[1] = sum.y; dst[2] = sum.z;
}

#endif
