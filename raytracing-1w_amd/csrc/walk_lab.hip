/* walk_lab.hip -- TRACE-ONLY HARNESS ("walk lab") for closest-hit candidates.  gfx950 only.
 *
 * Why it exists (round-2 review): every experiment on the BVH walk used to cost a full render and a full test suite,
 * and the decision on wide / restructured walks had been taken by estimate.  The lab separates the question "how fast can
 * `world.hit(ray, 0.001, inf)` (src/main.rs:62 -> src/bvh.rs:25-50, src/aabb.rs:13-32 and the leaves' `hit`) be answered
 * for the rays real paths produce?" from everything else:
 *
 *   rt1w_lab_dump_rays   runs real paths of a tile with the product's own core (rt_path_begin / rt_path_step) and
 *                        writes down the ray each path traces at bounce 0, 1, 2, ... -- the ray population of a render;
 *   rt1w_lab_set_rays    uploads any selection / order of them;
 *   rt1w_lab_trace       times ONLY the closest-hit search over those rays with one of the walks below, in a persistent
 *                        kernel whose idle lanes refill from the ray list (as the render kernels do), and returns
 *                        (t, primitive) per ray so that candidates are compared with the product's walk BIT FOR BIT.
 *
 * Walks:
 *   mode 0  W0: the product's stack walk (rt_walk_begin / rt_walk_step of rt_core.h), one stack entry per step.
 *   mode 7  W0b: W0 with four box-only steps behind every full step (rt_walk_box_step): a lane between boxes advances several nodes
 *           per execution of the rare kinds' code.  In the product for scenes with media (rt_kernel_plain.h: RT_SLICE_BOX_STEPS).
 *   mode 10 W0c: W0b with the top of the stack in a register during the box-only steps (rt_kernel_sorted.h: rt_walk_box_run).
 *   mode 8  W3: W0b with pair records for the BVH nodes that only steer (rt_walk_w3.h: rt_walkp_step): a steering node's record holds
 *           both children's boxes in f32 rounded outward; a child whose box fails is never popped or fetched; gates (BVH nodes
 *           directly above a primitive, a wrapper or a medium) keep their exact f64 test.  mode 9: the same without box-only steps.
 *   mode 3  W0q: W0 with the node records fetched by quads (four lanes share each 64-byte access) and transposed with DPP.
 *   mode 6  W2: the phased walk of rt_walk2.h -- W1's two phases for EVERY scene (wrappers, media, rects, nested BVHs), inner boxes
 *           in f32 rounded outward; what the product's stack-walk kernels are to run.
 *   mode 1  W1: pair walk in two phases (see below) -- media-free, wrapper-free scenes whose primitives sit under
 *           BVHChild::One or under a two-object BVHChild::Two (what BVHNode::new builds, bvh.rs:63-79).
 *
 * The lab is a measuring instrument: it ships in the library, but no render entry point reaches it.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rt1w.h"
#include "rt_kernel_sorted.h" /* rt_core.h, LdsStack, lane_prefix */
#include "rt_walk2.h"
#include "rt_pairs_build.h"
#include "scene.h"
#include "walk_lab.h"

#include "rt1w_internal.h"
/* this file is librt1w_lab.so: it reaches the product library through its exported entries only */
namespace rt1wlab { inline void set_error(const std::string& m) { rt1w_internal_set_error(m.c_str()); } }

namespace {

bool lab_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    rt1wlab::set_error(std::string("walk lab: ") + what + ": " + hipGetErrorString(e));
    return false;
}

/* ------------------------------------------------------------------------------------------------ ray dump -- */

/* one thread per path of the tile (work items of one sample: 8x8 pixel blocks, like the render kernels): the ray traced at
 * bounce b goes to out[(b * n_paths + path) * 8] = {o.xyz, d.xyz, time, 1.0}; paths that ended before bounce b leave 0.0 in
 * slot 7.  Variant V3 (every feature, stack walk) serves every scene. */
__global__ __launch_bounds__(RT_BLOCK, 2) void lab_dump_kernel(RtSceneView sc, RtFrame f, uint32_t n_bounces, unsigned long long n_paths,
                                                               double* __restrict__ out) {
    __shared__ uint32_t stack_mem[RT_STACK_CAP * RT_BLOCK];
    LdsStack stk;
    stk.base = stack_mem + threadIdx.x;
    stk.sp = 0;
    RtGlobalNodes ns{sc.nodes};
    const unsigned long long item = (unsigned long long)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (item >= rt_item_count(f)) return;
    uint32_t px, py, chunk;
    rt_item_decode(f, item, px, py, chunk);
    if (px >= f.tile_w || py >= f.tile_h) return;
    const unsigned long long path = ((unsigned long long)py * f.tile_w + px) * f.spp + chunk;
    RtPath p;
    rt_path_begin(sc, f, f.x0 + px, rt_frame_row(f, py), f.sample_offset + chunk, p);
    for (uint32_t b = 0; b < n_bounces; ++b) {
        double* o = out + ((unsigned long long)b * n_paths + path) * 8ull;
        const bool traces = p.alive && p.depth_left != 0u;
        o[0] = p.ray.o.x; o[1] = p.ray.o.y; o[2] = p.ray.o.z;
        o[3] = p.ray.d.x; o[4] = p.ray.d.y; o[5] = p.ray.d.z;
        o[6] = p.ray.time; o[7] = traces ? 1.0 : 0.0;
        if (!traces) { for (uint32_t k = b + 1; k < n_bounces; ++k) out[((unsigned long long)k * n_paths + path) * 8ull + 7ull] = 0.0; return; }
        rt_path_step<RtCfgV3>(sc, ns, p, stk);
    }
}

/* ------------------------------------------------------------------------------------------- common frame -- */

struct LabRay { double o[3], d[3], time, pad; };
struct LabHit { double t; uint32_t prim; uint32_t flags; }; /* flags: bit 0 = answered by the fallback walk */

/* wave-aggregated fetch of the next ray index for the lanes that ask (one atomic per wave) */
__device__ __forceinline__ unsigned long long lab_fetch(bool want, unsigned long long* counter) {
    unsigned long long idx = ~0ull;
    if (want) {
        const unsigned long long need = __ballot(1);
        const uint32_t cnt = (uint32_t)__popcll(need);
        const uint32_t rank = lane_prefix(need);
        unsigned long long base = 0;
        if (rank == 0u) base = atomicAdd(counter, (unsigned long long)cnt);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        idx = (((unsigned long long)hi << 32) | lo) + rank;
    }
    return idx;
}

/* ------------------------------------------------------------------------------------------------- W0 -- */

/* the product's walk, one entry per step; idle lanes refill once `refill_idle` of them wait (or nothing else can step) */
template <class Cfg, int BOXSTEPS = 0, bool RUN = false>
__global__ __launch_bounds__(RT_BLOCK, 3) void lab_trace_w0(RtSceneView sc, const LabRay* __restrict__ rays, unsigned long long n,
                                                            LabHit* __restrict__ out, unsigned long long* counter, uint32_t refill_idle,
                                                            unsigned long long* stats) {
    __shared__ uint32_t stack_mem[RT_STACK_CAP * RT_BLOCK];
    LdsStack stk;
    stk.base = stack_mem + threadIdx.x;
    stk.sp = 0;
    RtGlobalNodes ns{sc.nodes};
    RtWalk k;
    RtRng rng = rt_rng_make(0u, 0u, 0u, 0u, RT_DOMAIN_RENDER);
    unsigned long long mine = ~0ull;
    bool walking = false, exhausted = false;
    unsigned long long steps = 0, wave_steps = 0;
    for (;;) {
        const bool idle = !walking;
        const unsigned long long idle_m = __ballot(idle && !exhausted);
        const unsigned long long walk_m = __ballot(walking);
        if ((uint32_t)__popcll(idle_m) >= refill_idle || walk_m == 0ull) {
            if (idle && !exhausted) {
                if (mine != ~0ull) { LabHit h; h.t = k.best_t; h.prim = k.best_prim; h.flags = 0u; out[mine] = h; mine = ~0ull; }
            }
            const unsigned long long idx = lab_fetch(idle && !exhausted, counter);
            if (idle && !exhausted) {
                if (idx < n) {
                    const LabRay r = rays[idx];
                    RtRay w; w.o = rt_v3(r.o[0], r.o[1], r.o[2]); w.d = rt_v3(r.d[0], r.d[1], r.d[2]); w.time = r.time;
                    rng = rt_rng_make((uint32_t)idx, (uint32_t)(idx >> 32), 0u, 0u, RT_DOMAIN_RENDER);
                    stk.sp = 0;
                    rt_walk_begin(k, sc.root, w, 0.001, RT_INF, stk);
                    mine = idx; walking = true;
                } else exhausted = true;
            }
        }
        if (__ballot(walking) == 0ull) break;
        if (walking) {
            rt_walk_step<Cfg, true>(sc, ns, k, rng, stk);
            ++steps;
            if constexpr (BOXSTEPS > 0 && RUN) { /* W0c: the same steps with the top of the stack in a register (rt_walk_box_run) */
                rt_walk_box_run<Cfg, BOXSTEPS>(ns, k, stk);
            } else if constexpr (BOXSTEPS > 0) { /* W0b: box-only steps behind the full step (rt_walk_box_step; what the render kernels of media scenes do) */
                bool between_boxes = true;
#pragma unroll
                for (int extra = 0; extra < BOXSTEPS; ++extra)
                    if (between_boxes && !rt_walk_done(k, stk)) { between_boxes = rt_walk_box_step<Cfg>(ns, k, stk); steps += between_boxes ? 1ull : 0ull; }
            }
            if (rt_walk_done(k, stk)) walking = false;
        }
        ++wave_steps;
    }
    if (mine != ~0ull) { LabHit h; h.t = k.best_t; h.prim = k.best_prim; h.flags = 0u; out[mine] = h; }
    if (stats) {
        atomicAdd(&stats[0], steps);
        if ((threadIdx.x & 63u) == 0u) atomicAdd(&stats[1], wave_steps);
    }
}

/* ------------------------------------------------------------------------------------------------- W4 -- */

/* W4: the product's steps (one full step + BOXSTEPS box-only steps per round, as W0c) with the walks REGROUPED ACROSS THE FOUR WAVES of
 * a workgroup by the kind of their next entry.  final_scene's walk is bound by kind divergence (finding 2): nearly
 * every full step of a wave runs the sphere, rect, wrapper, exit and medium code for two or three lanes each.  Here the walk state of
 * the workgroup's 256 rays lives in LDS (11 qwords + 7 dwords per slot, the per-slot stack next to it), and before every round the
 * slots are counting-sorted by (inside a wrapper?, kind of the top entry): lane i of the workgroup then takes the i-th slot of that
 * order, loads what the step needs, steps, and stores what changed (closest hit, scope, stack level, draws).  A wave then holds one or
 * two kinds.  Per ray nothing changes -- the same entries in the same order with the same operands, whichever lane executes them -- so
 * the hits are W0's bit for bit.  A ConstantMedium's draw comes from the ray's own counter-based stream: the slot carries the number of
 * draws taken so far and the lane at the medium regenerates the generator at that position (rt_rng_rewind).
 * Slices as in the render kernels, per workgroup: once `slice_idle` of the slice's walks have ended every lane goes home to its own
 * slot, writes a finished walk's hit out and starts the next ray there. */
#define LAB_W4_NODE_CAP 8192 /* node kinds as bytes in LDS */
#define LAB_W4_STACK 24
#define LAB_W4_DONE 15u
struct LabStack16 { /* 16-bit entries (node index < 32768, wrapper-exit flag in bit 15), entry e of the slot at base[e * RT_BLOCK] */
    uint16_t* base;
    int sp;
    __device__ __forceinline__ static uint16_t enc(uint32_t v) { return (uint16_t)((v & 0x7FFFu) | ((v & RT_POP_FLAG) ? 0x8000u : 0u)); }
    __device__ __forceinline__ static uint32_t dec(uint32_t x) { return (x & 0x7FFFu) | ((x & 0x8000u) ? RT_POP_FLAG : 0u); }
    __device__ __forceinline__ void push(uint32_t v) { base[sp * RT_BLOCK] = enc(v); ++sp; }
    __device__ __forceinline__ void poke(int above, uint32_t v) { base[(sp + above) * RT_BLOCK] = enc(v); }
    __device__ __forceinline__ uint32_t pop() { --sp; return dec(base[sp * RT_BLOCK]); }
    __device__ __forceinline__ uint32_t at(int i) const { return dec(base[i * RT_BLOCK]); }
    __device__ __forceinline__ void put(int i, uint32_t v) { base[i * RT_BLOCK] = enc(v); }
};
/* inclusive prefix sum over the 64 lanes of a wave (row shifts + row broadcasts, the rocPRIM pattern) */
__device__ __forceinline__ uint32_t lab_wave_scan(uint32_t v, uint32_t lane) {
    const uint32_t rl = lane & 15u;
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x111, 0xf, 0xf, false); if (rl >= 1u) v += t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x112, 0xf, 0xf, false); if (rl >= 2u) v += t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x114, 0xf, 0xf, false); if (rl >= 4u) v += t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x118, 0xf, 0xf, false); if (rl >= 8u) v += t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x142, 0xf, 0xf, false); if ((lane & 31u) >= 16u) v += t; /* row_bcast:15 */
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x143, 0xf, 0xf, false); if (lane >= 32u) v += t;         /* row_bcast:31 */
    return v;
}
template <class Cfg, int BOXSTEPS>
__global__ __launch_bounds__(RT_BLOCK, 3) void lab_trace_w4(RtSceneView sc, const uint8_t* __restrict__ cls_tab, const LabRay* __restrict__ rays, unsigned long long n,
                                                            LabHit* __restrict__ out, unsigned long long* counter, uint32_t slice_idle,
                                                            unsigned long long* stats) {
    __shared__ double s64[11 * RT_BLOCK];   /* w.o xyz, w.d xyz, 1/w.d xyz, time, closest t */
    __shared__ uint32_t s32[7 * RT_BLOCK];  /* scope, closest primitive, its scope, stack level, draws taken, ray index lo / hi */
    __shared__ uint16_t stack_mem[LAB_W4_STACK * RT_BLOCK];
    __shared__ uint8_t cls_lds[LAB_W4_NODE_CAP];
    __shared__ uint16_t perm[RT_BLOCK];
    __shared__ uint32_t cnt[64]; /* [key][wave] */
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < sc.n_nodes && i < (uint32_t)LAB_W4_NODE_CAP; i += RT_BLOCK) cls_lds[i] = cls_tab[i];
    RtGlobalNodes ns{sc.nodes};
    unsigned long long mine = ~0ull;
    bool exhausted = false;
    unsigned long long steps = 0, rounds = 0, slices = 0;
    s32[3 * RT_BLOCK + tid] = 0u;
    __syncthreads();
    /* key of a slot: what its top entry is (bit 3: inside a wrapper, where the step rebuilds the wrapper's ray first) */
    auto slot_key = [&](uint32_t slot, uint32_t sp, uint32_t scope) -> uint32_t {
        if (sp == 0u) return LAB_W4_DONE;
        const uint32_t x = stack_mem[(sp - 1u) * RT_BLOCK + slot];
        if (x & 0x8000u) return 14u;
        return (uint32_t)cls_lds[x] | (scope != RT_NONE ? 8u : 0u);
    };
    for (;;) { /* ---- a slice: home, then rounds ---- */
        {
            const bool idle = s32[3 * RT_BLOCK + tid] == 0u;
            if (idle && mine != ~0ull) {
                LabHit h; h.t = s64[10 * RT_BLOCK + tid]; h.prim = s32[1 * RT_BLOCK + tid]; h.flags = 0u; out[mine] = h; mine = ~0ull;
            }
            const unsigned long long idx = lab_fetch(idle && !exhausted, counter);
            if (idle && !exhausted) {
                if (idx < n) {
                    const LabRay r = rays[idx];
                    const RtV3 inv = rt_inv3(rt_v3(r.d[0], r.d[1], r.d[2]));
                    s64[0 * RT_BLOCK + tid] = r.o[0]; s64[1 * RT_BLOCK + tid] = r.o[1]; s64[2 * RT_BLOCK + tid] = r.o[2];
                    s64[3 * RT_BLOCK + tid] = r.d[0]; s64[4 * RT_BLOCK + tid] = r.d[1]; s64[5 * RT_BLOCK + tid] = r.d[2];
                    s64[6 * RT_BLOCK + tid] = inv.x; s64[7 * RT_BLOCK + tid] = inv.y; s64[8 * RT_BLOCK + tid] = inv.z;
                    s64[9 * RT_BLOCK + tid] = r.time; s64[10 * RT_BLOCK + tid] = RT_INF;
                    s32[0 * RT_BLOCK + tid] = RT_NONE; s32[1 * RT_BLOCK + tid] = RT_NONE; s32[2 * RT_BLOCK + tid] = RT_NONE;
                    s32[3 * RT_BLOCK + tid] = 1u; s32[4 * RT_BLOCK + tid] = 0u;
                    s32[5 * RT_BLOCK + tid] = (uint32_t)idx; s32[6 * RT_BLOCK + tid] = (uint32_t)(idx >> 32);
                    stack_mem[tid] = LabStack16::enc(sc.root);
                    mine = idx;
                } else exhausted = true;
            }
        }
        uint32_t myslot = tid;
        uint32_t key = slot_key(tid, s32[3 * RT_BLOCK + tid], s32[0 * RT_BLOCK + tid]);
        uint32_t stop_at = 0u;
        bool first = true, finished = false;
        ++slices;
        for (;;) { /* ---- a round: sort the slots by key, one full step + the box-only steps ---- */
            /* rank among the lanes of this wave with the same key, and that key's count */
            unsigned long long same = ~0ull;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned long long m = __ballot((key >> b) & 1u);
                same &= ((key >> b) & 1u) ? m : ~m;
            }
            same &= __ballot(1);
            const uint32_t rank = lane_prefix(same);
            if (lane < 16u) cnt[lane * 4u + wave] = 0u;
            if (rank == 0u) cnt[key * 4u + wave] = (uint32_t)__popcll(same);
            __syncthreads();
            const uint32_t c = cnt[lane];
            const uint32_t incl = lab_wave_scan(c, lane);
            const uint32_t excl = incl - c;
            const uint32_t dest = (uint32_t)__shfl((int)excl, (int)(key * 4u + wave)) + rank;
            const uint32_t active = (uint32_t)__builtin_amdgcn_readlane((int)excl, 60); /* slots whose walk is not over */
            if (first) {
                if (active == 0u) { finished = true; break; }
                stop_at = active > slice_idle ? active - slice_idle : 0u;
                first = false;
            } else if (active <= stop_at) break;
            perm[dest] = (uint16_t)myslot;
            __syncthreads();
            myslot = perm[tid];
            ++rounds;
            if (tid < active) {
                const uint32_t sl = myslot;
                RtWalk k;
                k.w.o = rt_v3(s64[0 * RT_BLOCK + sl], s64[1 * RT_BLOCK + sl], s64[2 * RT_BLOCK + sl]);
                k.w.d = rt_v3(s64[3 * RT_BLOCK + sl], s64[4 * RT_BLOCK + sl], s64[5 * RT_BLOCK + sl]);
                k.inv_w = rt_v3(s64[6 * RT_BLOCK + sl], s64[7 * RT_BLOCK + sl], s64[8 * RT_BLOCK + sl]);
                k.time = s64[9 * RT_BLOCK + sl]; k.best_t = s64[10 * RT_BLOCK + sl];
                k.scope = s32[0 * RT_BLOCK + sl]; k.best_prim = s32[1 * RT_BLOCK + sl]; k.best_scope = s32[2 * RT_BLOCK + sl];
                uint32_t ndraw = s32[4 * RT_BLOCK + sl];
                k.t_min = 0.001; k.tmin_nan = false; k.base = 0;
                LabStack16 stk; stk.base = stack_mem + sl; stk.sp = (int)s32[3 * RT_BLOCK + sl];
                if (Cfg::scope_depth == 0 || k.scope == RT_NONE) { k.cur = k.w; k.inv = k.inv_w; }
                else {
                    bool rotated;
                    k.cur = rt_ray_in_scope_r(sc.nodes, k.scope, k.w, rotated);
                    k.inv = k.inv_w;
                    if (rotated) k.inv = rt_inv3(k.cur.d);
                }
                /* the full step (rt_walk_step), a medium's draw taken from the ray's own stream at the position the slot has reached */
                {
                    const uint32_t e = stk.pop();
                    if (Cfg::scope_depth > 0 && (e & RT_POP_FLAG)) rt_walk_exit(sc, k, e);
                    else {
                        const RtNodeHot nd = ns.hot(e);
                        const uint32_t km = nd.kind & RT_KIND_MASK;
                        if (km <= RT_BVH1) rt_walk_box<Cfg, false>(k, e, nd, stk);
                        else if (km <= RT_YZ) rt_walk_leaf<Cfg>(sc, k, e, nd);
                        else if (Cfg::scope_depth > 0 && km <= RT_FLIP) rt_walk_wrap(k, e, nd, stk);
                        else if (Cfg::media) {
                            RtRng rng = rt_rng_make(s32[5 * RT_BLOCK + sl], s32[6 * RT_BLOCK + sl], 0u, 0u, RT_DOMAIN_RENDER);
                            RtRngMark m; m.bv = 0u; m.left = (ndraw & 1u) ? 2u : 0u; m.blk = (ndraw + 1u) >> 1;
                            rt_rng_rewind(rng, m);
                            rt_walk_other<Cfg, true>(sc, ns, k, e, nd, rng, stk);
                            ndraw = (4u * (rng.blk - rng.bv) - rng.left) >> 1;
                        }
                    }
                    ++steps;
                }
                if constexpr (BOXSTEPS > 0) rt_walk_box_run<Cfg, BOXSTEPS>(ns, k, stk);
                s64[10 * RT_BLOCK + sl] = k.best_t;
                s32[0 * RT_BLOCK + sl] = k.scope; s32[1 * RT_BLOCK + sl] = k.best_prim; s32[2 * RT_BLOCK + sl] = k.best_scope;
                s32[3 * RT_BLOCK + sl] = (uint32_t)stk.sp; s32[4 * RT_BLOCK + sl] = ndraw;
                key = stk.sp >= LAB_W4_STACK - 2 ? LAB_W4_DONE /* cannot happen on a scene the host admitted; never walk past the slot's words */
                                                 : slot_key(sl, (uint32_t)stk.sp, k.scope);
                if (key == LAB_W4_DONE) s32[3 * RT_BLOCK + sl] = 0u;
            } else key = LAB_W4_DONE;
            if (rounds > (1ull << 26)) { finished = true; break; } /* safety net of the experiment: a grid that cannot drain must not exist */
        }
        if (finished) break;
        __syncthreads(); /* the count table is written again at once by the next slice's first round */
    }
    if (stats) {
        atomicAdd(&stats[0], steps);
        if (lane == 0u) { atomicAdd(&stats[1], rounds); atomicAdd(&stats[2], slices); }
    }
}

/* ------------------------------------------------------------------------------------------------- W3 -- */

/* W3: the one-entry-per-step walk with PAIR RECORDS for the BVH nodes that only steer (rt_walk_w3.h: rt_walkp_step; `sc.nodes` is the
 * patched copy, rt_pairs_build.h) and BOXSTEPS box-only steps behind every full step */
template <class Cfg, int BOXSTEPS>
__global__ __launch_bounds__(RT_BLOCK, 3) void lab_trace_w3(RtSceneView sc, const RtPairRec* __restrict__ pairs, const LabRay* __restrict__ rays, unsigned long long n,
                                                            LabHit* __restrict__ out, unsigned long long* counter, uint32_t refill_idle,
                                                            unsigned long long* stats) {
    __shared__ uint32_t stack_mem[RT_STACK_CAP * RT_BLOCK];
    LdsStack stk;
    stk.base = stack_mem + threadIdx.x;
    stk.sp = 0;
    RtGlobalNodes ns{sc.nodes};
    RtWalk k;
    RtRng rng = rt_rng_make(0u, 0u, 0u, 0u, RT_DOMAIN_RENDER);
    unsigned long long mine = ~0ull;
    bool walking = false, exhausted = false;
    unsigned long long steps = 0, wave_steps = 0;
    for (;;) {
        const bool idle = !walking;
        const unsigned long long idle_m = __ballot(idle && !exhausted);
        const unsigned long long walk_m = __ballot(walking);
        if ((uint32_t)__popcll(idle_m) >= refill_idle || walk_m == 0ull) {
            if (idle && !exhausted) {
                if (mine != ~0ull) { LabHit h; h.t = k.best_t; h.prim = k.best_prim; h.flags = 0u; out[mine] = h; mine = ~0ull; }
            }
            const unsigned long long idx = lab_fetch(idle && !exhausted, counter);
            if (idle && !exhausted) {
                if (idx < n) {
                    const LabRay r = rays[idx];
                    RtRay w; w.o = rt_v3(r.o[0], r.o[1], r.o[2]); w.d = rt_v3(r.d[0], r.d[1], r.d[2]); w.time = r.time;
                    rng = rt_rng_make((uint32_t)idx, (uint32_t)(idx >> 32), 0u, 0u, RT_DOMAIN_RENDER);
                    stk.sp = 0;
                    rt_walk_begin(k, sc.root, w, 0.001, RT_INF, stk);
                    mine = idx; walking = true;
                } else exhausted = true;
            }
        }
        if (__ballot(walking) == 0ull) break;
        if (walking) {
            rt_walkp_step<Cfg, true>(sc, ns, pairs, k, rng, stk);
            ++steps;
            if constexpr (BOXSTEPS > 0) { 
                bool between_boxes = true;
#pragma unroll
                for (int extra = 0; extra < BOXSTEPS; ++extra)
                    if (between_boxes && !rt_walk_done(k, stk)) { between_boxes = rt_walkp_box_step<Cfg>(ns, pairs, k, stk); steps += between_boxes ? 1ull : 0ull; }
            }
            if (rt_walk_done(k, stk)) walking = false;
        }
        ++wave_steps;
    }
    if (mine != ~0ull) { LabHit h; h.t = k.best_t; h.prim = k.best_prim; h.flags = 0u; out[mine] = h; }
    if (stats) {
        atomicAdd(&stats[0], steps);
        if ((threadIdx.x & 63u) == 0u) atomicAdd(&stats[1], wave_steps);
    }
}

/* ------------------------------------------------------------------------------------------------ W0q -- */

/* QUAD-TRANSPOSED RECORD FETCH.  PMC of W0 (profiles/r03_lab_*): the L1 (TCP) is busy in 97 % of the cycles at 0.71 accesses per
 * cycle while the VALU is busy in under half of them -- the walk is bound by the NUMBER OF L1 ACCESSES, and a lane that fetches
 * its own 64-byte record with four 16-byte loads makes four of them, each to a line no neighbour shares.  The same bytes cost
 * a quarter of the accesses when the four lanes of a quad fetch them together: load k of the step has every lane of the quad
 * read ITS 16-byte quarter of quad-lane k's record (64 contiguous bytes per quad = one access), and a 4x4 transpose inside the
 * quad (two DPP butterfly stages) hands every lane the four quarters of its own record.  Which lane runs which ray, and what it
 * computes, is unchanged.  All 64 lanes must execute the fetch (DPP reads need their quad partners): it sits outside the
 * per-lane control flow, lanes without an entry fetch node 0. */
template <int CTRL>
__device__ __forceinline__ uint32_t lab_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ uint4 lab_dpp4(uint4 v) { return make_uint4(lab_dpp<CTRL>(v.x), lab_dpp<CTRL>(v.y), lab_dpp<CTRL>(v.z), lab_dpp<CTRL>(v.w)); }
__device__ __forceinline__ uint4 lab_sel4(bool c, uint4 a, uint4 b) { return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }
__device__ __forceinline__ RtNodeHot lab_quad_fetch(const RtNode* nodes, uint32_t e) {
    const uint32_t j = threadIdx.x & 3u;
    const char* base = reinterpret_cast<const char*>(nodes) + 16u * j;
    const uint32_t e0 = lab_dpp<0x00>(e), e1 = lab_dpp<0x55>(e), e2 = lab_dpp<0xAA>(e), e3 = lab_dpp<0xFF>(e);
    const uint4 v0 = *reinterpret_cast<const uint4*>(base + (size_t)e0 * sizeof(RtNode));
    const uint4 v1 = *reinterpret_cast<const uint4*>(base + (size_t)e1 * sizeof(RtNode));
    const uint4 v2 = *reinterpret_cast<const uint4*>(base + (size_t)e2 * sizeof(RtNode));
    const uint4 v3 = *reinterpret_cast<const uint4*>(base + (size_t)e3 * sizeof(RtNode));
    /* lane j holds v[k] = quarter j of record k; it wants r[m] = quarter m of record j = v[j] of lane m: a 4x4 transpose */
    const bool even = (j & 1u) == 0u, lo = (j & 2u) == 0u;
    const uint4 a0 = lab_sel4(even, v0, lab_dpp4<0xB1>(v1)), a1 = lab_sel4(even, lab_dpp4<0xB1>(v0), v1);
    const uint4 a2 = lab_sel4(even, v2, lab_dpp4<0xB1>(v3)), a3 = lab_sel4(even, lab_dpp4<0xB1>(v2), v3);
    const uint4 r0 = lab_sel4(lo, a0, lab_dpp4<0x4E>(a2)), r2 = lab_sel4(lo, lab_dpp4<0x4E>(a0), a2);
    const uint4 r1 = lab_sel4(lo, a1, lab_dpp4<0x4E>(a3)), r3 = lab_sel4(lo, lab_dpp4<0x4E>(a1), a3);
    union { uint4 q[4]; RtNodeHot h; } u;
    u.q[0] = r0; u.q[1] = r1; u.q[2] = r2; u.q[3] = r3;
    return u.h;
}

template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK, 3) void lab_trace_w0q(RtSceneView sc, const LabRay* __restrict__ rays, unsigned long long n,
                                                             LabHit* __restrict__ out, unsigned long long* counter, uint32_t refill_idle,
                                                             unsigned long long* stats) {
    __shared__ uint32_t stack_mem[RT_STACK_CAP * RT_BLOCK];
    LdsStack stk;
    stk.base = stack_mem + threadIdx.x;
    stk.sp = 0;
    RtGlobalNodes ns{sc.nodes};
    RtWalk k;
    RtRng rng = rt_rng_make(0u, 0u, 0u, 0u, RT_DOMAIN_RENDER);
    unsigned long long mine = ~0ull;
    bool walking = false, exhausted = false;
    unsigned long long steps = 0, wave_steps = 0;
    for (;;) {
        const bool idle = !walking;
        const unsigned long long idle_m = __ballot(idle && !exhausted);
        const unsigned long long walk_m = __ballot(walking);
        if ((uint32_t)__popcll(idle_m) >= refill_idle || walk_m == 0ull) {
            if (idle && !exhausted) {
                if (mine != ~0ull) { LabHit h; h.t = k.best_t; h.prim = k.best_prim; h.flags = 0u; out[mine] = h; mine = ~0ull; }
            }
            const unsigned long long idx = lab_fetch(idle && !exhausted, counter);
            if (idle && !exhausted) {
                if (idx < n) {
                    const LabRay r = rays[idx];
                    RtRay w; w.o = rt_v3(r.o[0], r.o[1], r.o[2]); w.d = rt_v3(r.d[0], r.d[1], r.d[2]); w.time = r.time;
                    rng = rt_rng_make((uint32_t)idx, (uint32_t)(idx >> 32), 0u, 0u, RT_DOMAIN_RENDER);
                    stk.sp = 0;
                    rt_walk_begin(k, sc.root, w, 0.001, RT_INF, stk);
                    mine = idx; walking = true;
                } else exhausted = true;
            }
        }
        if (__ballot(walking) == 0ull) break;
        /* pop (per lane), fetch (all lanes together), visit (per lane) */
        uint32_t e = 0u;
        bool node = false;
        if (walking) {
            e = stk.pop();
            if (Cfg::scope_depth > 0 && (e & RT_POP_FLAG)) rt_walk_exit(sc, k, e);
            else node = true;
        }
        const RtNodeHot nd = lab_quad_fetch(sc.nodes, node ? e : 0u);
        if (node) rt_walk_visit<Cfg, true>(sc, ns, k, rng, stk, e, nd);
        if (walking) { ++steps; if (rt_walk_done(k, stk)) walking = false; }
        ++wave_steps;
    }
    if (mine != ~0ull) { LabHit h; h.t = k.best_t; h.prim = k.best_prim; h.flags = 0u; out[mine] = h; }
    if (stats) {
        atomicAdd(&stats[0], steps);
        if ((threadIdx.x & 63u) == 0u) atomicAdd(&stats[1], wave_steps);
    }
}

/* --------------------------------------------------------------------------------------- gather probe -- */

/* What can the L1 deliver for the walk's access pattern?  Every lane reads `iters` 64-byte records of the node array at pseudo-random
 * indices (an LCG per lane; DEP: the next index also depends on the bytes just read, as a walk's does).  QUAD 0: four 16-byte loads
 * per lane to its own record (the walk's fetch); QUAD 1: the quad-shared form (four loads, each 64 contiguous bytes per quad, no
 * transpose: the probe only sums the bytes).  Result: records per second, to set against visits per second of the walks. */
template <int QUAD, int DEP>
__global__ __launch_bounds__(256) void lab_gather_kernel(const RtNode* __restrict__ nodes, uint32_t n_nodes, uint32_t iters, unsigned long long* sink) {
    uint32_t state = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0u;
    const uint32_t j = threadIdx.x & 3u;
    for (uint32_t it = 0; it < iters; ++it) {
        state = state * 1664525u + 1013904223u;
        const uint32_t e = (uint32_t)(((unsigned long long)(state >> 8) * n_nodes) >> 24);
        uint4 v0, v1, v2, v3;
        if (QUAD) {
            const char* base = reinterpret_cast<const char*>(nodes) + 16u * j;
            const uint32_t e0 = lab_dpp<0x00>(e), e1 = lab_dpp<0x55>(e), e2 = lab_dpp<0xAA>(e), e3 = lab_dpp<0xFF>(e);
            v0 = *reinterpret_cast<const uint4*>(base + (size_t)e0 * sizeof(RtNode));
            v1 = *reinterpret_cast<const uint4*>(base + (size_t)e1 * sizeof(RtNode));
            v2 = *reinterpret_cast<const uint4*>(base + (size_t)e2 * sizeof(RtNode));
            v3 = *reinterpret_cast<const uint4*>(base + (size_t)e3 * sizeof(RtNode));
        } else {
            const uint4* p = reinterpret_cast<const uint4*>(nodes + e);
            v0 = p[0]; v1 = p[1]; v2 = p[2]; v3 = p[3];
        }
        const uint32_t x = v0.x ^ v0.w ^ v1.y ^ v1.z ^ v2.x ^ v2.w ^ v3.y ^ v3.z;
        acc ^= x;
        if (DEP) state ^= x & 0xFFu;
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

/* ------------------------------------------------------------------------------------------------- W1 -- */

/* PAIR WALK IN TWO PHASES.
 *
 * What the reference does per ray (bvh.rs:25-50): depth first, left before right, every BVH node's box tested with the
 * closest hit so far as t_max, every object under a passed node tested the same way.  What costs on the GPU is not that
 * arithmetic but running it for a few lanes at a time: in a one-entry-per-step walk each wave step executes the box code,
 * the sphere code and the moving-sphere code one after the other, each for the lanes that happen to be at that kind
 * (25 % of the lanes active per VALU instruction on random_scene, PMC).
 *
 * Here a lane's walk is split into the two kinds of work, and the WAVE decides which kind it runs next:
 *   box phase   every stepping lane visits one inner node: ONE 128-byte record holds the boxes of both children, both are
 *               tested, the left one is entered (the right one pushed) -- no leaf arithmetic in this phase.  A child that
 *               is a LEAF GROUP (a BVH node whose children are primitives: BVHChild::One(prim) or Two(prim, prim)) is not
 *               entered but appended to the lane's queue of pending groups, in the order the reference reaches them;
 *   leaf phase  every lane with a pending group handles its oldest one: the group's own box is tested with the closest hit
 *               AS IT IS NOW, then its one or two primitives in the reference's order (left = first, bvh.rs:38-47).
 *
 * Exactness.  The box phase runs ahead of the leaf phase, so its box tests see a closest hit that may be STALE (not yet
 * lowered by the pending groups).  A stale t_max can only make a box pass that the reference would have failed, never the
 * reverse (the slab verdict is min(exit, t_max) > enter: monotone in t_max).  Every primitive's gate -- its group's own box
 * -- is re-evaluated in the leaf phase at exactly the closest hit the reference has when it reaches that group, because
 * groups are handled strictly in the reference's order and nothing else changes the closest hit.  The inner boxes above a
 * group need no second look: a child's box lies inside its parent's (surrounding_box, aabb.rs:35-52), so a group box that
 * passes at t implies every box above it passes at t, hence at the larger t_max the reference tested it with.  So the set
 * of primitives tested, their order and every operand are the reference's: same bits.  Stack entries carry the entry
 * distance of the pushed box (rounded DOWN to f32: culling at the pop is allowed to miss, never to over-cull).
 * A closest hit that turns NaN (a NaN root is accepted, sphere.rs:43-48) breaks monotonicity: such a ray is answered by
 * the product's walk from scratch (flag bit 0).
 */
struct LabPNode { /* inner node: both children's boxes (min.xyz, max.xyz) and what the children are */
    double lb[6], rb[6];
    uint32_t l, r;       /* child reference: LAB_LEAF | group index, or inner index */
    uint32_t pad[6];
};                       /* 128 bytes */
struct LabPrim {
    double c0[3], dc[3]; /* Sphere: centre, 0; MovingSphere: center0, center1 - center0 (moving_sphere.rs:23-26) */
    double radius;
    uint32_t id;         /* index of the primitive's RtNode (what the product's walk reports) */
    uint32_t moving;
};                       /* 64 bytes */
struct LabGroup {
    double box[6];       /* the group's own box: BVHNode.aabb of the One / Two node */
    uint32_t n, pad[3];
    LabPrim p[2];
};                       /* 192 bytes */
struct LabPNode32 { /* the same, boxes stored in f32 rounded OUTWARD: inner boxes only steer the walk (conservative culling), so they need
                      not be the reference's bits -- the arithmetic on them stays f64 and monotone, hence never culls what the exact test keeps */
    float lb[6], rb[6];
    uint32_t l, r;
    uint32_t pad[2];
};                       /* 64 bytes: four 16-byte loads instead of eight */
#define LAB_LEAF 0x80000000u
#define LAB_NONE 0xFFFFFFFFu
#define LAB_QCAP 4       /* pending groups per lane */
#define LAB_W1_STACK 24
#define LAB_W1_BLOCK 256

struct LabW1Scene {
    const LabPNode* inner;
    const LabPNode32* inner32;
    const LabGroup* groups;
    double root_box[6];
    uint32_t root;       /* reference of the root (inner or LAB_LEAF | group) */
    double ms_time0, ms_time1;
};

__device__ __forceinline__ void lab_slab(const double* bb, RtV3 o, RtV3 inv, double t_min, double& enter, double& exit_) {
    /* aabb.rs:14-29 with the interval kept by max/min (rt_aabb_hit_fast's arithmetic), without folding t_max in */
    double lo = t_min, hi = RT_INF;
#define LAB_AX(minv, maxv, ov, iv)                       \
    {                                                    \
        double t0 = ((minv) - (ov)) * (iv);              \
        double t1 = ((maxv) - (ov)) * (iv);              \
        if ((iv) < 0.0) { double s_ = t0; t0 = t1; t1 = s_; } \
        lo = rt_vmax(t0, lo);                            \
        hi = rt_vmin(t1, hi);                            \
    }
    LAB_AX(bb[0], bb[3], o.x, inv.x)
    LAB_AX(bb[1], bb[4], o.y, inv.y)
    LAB_AX(bb[2], bb[5], o.z, inv.z)
#undef LAB_AX
    enter = lo; exit_ = hi;
}

#define LAB_LDS_INNER 256 /* inner pair records kept in LDS by the experiment kernel (32 KB) */
#define LAB_LDS_STACK 16
template <bool LDS_INNER, bool F32 = false>
__global__ __launch_bounds__(LAB_W1_BLOCK, LDS_INNER ? (F32 ? 3 : 2) : (F32 ? 4 : 3)) void lab_trace_w1(LabW1Scene ps, const LabRay* __restrict__ rays, unsigned long long n,
                                                                 LabHit* __restrict__ out, unsigned long long* counter, uint32_t refill_idle,
                                                                 uint32_t leaf_votes, unsigned long long* stats, uint32_t n_inner) {
    __shared__ uint32_t s_ref[((LDS_INNER || F32) ? LAB_LDS_STACK : LAB_W1_STACK) * LAB_W1_BLOCK];
    __shared__ float s_ent[((LDS_INNER || F32) ? LAB_LDS_STACK : LAB_W1_STACK) * LAB_W1_BLOCK];
    __shared__ uint32_t s_q[LAB_QCAP * LAB_W1_BLOCK];
    __shared__ LabPNode s_inner[(LDS_INNER && !F32) ? LAB_LDS_INNER : 1];
    __shared__ LabPNode32 s_inner32[(LDS_INNER && F32) ? LAB_LDS_INNER : 1];
    if constexpr (LDS_INNER) {
        const uint4* src = F32 ? reinterpret_cast<const uint4*>(ps.inner32) : reinterpret_cast<const uint4*>(ps.inner);
        uint4* dst = F32 ? reinterpret_cast<uint4*>(s_inner32) : reinterpret_cast<uint4*>(s_inner);
        for (uint32_t i = threadIdx.x; i < n_inner * (F32 ? 4u : 8u); i += LAB_W1_BLOCK) dst[i] = src[i];
        __syncthreads();
    }
    uint32_t* const my_ref = s_ref + threadIdx.x;
    float* const my_ent = s_ent + threadIdx.x;
    uint32_t* const my_q = s_q + threadIdx.x;

    RtV3 o = rt_v3(0, 0, 0), d = rt_v3(0, 0, 0), inv = rt_v3(0, 0, 0);
    double frac = 0.0, best_t = RT_INF;
    uint32_t best_prim = RT_NONE;
    int sp = 0;
    uint32_t cur = LAB_NONE, qh = 0u, qn = 0u;
    unsigned long long mine = ~0ull;
    bool walking = false, exhausted = false, bad = false;
    unsigned long long n_box = 0, n_leaf = 0, w_box = 0, w_leaf = 0;
    const double t_min = 0.001;

    for (;;) {
        /* ---- refill -------------------------------------------------------------------------------------------- */
        {
            const bool idle = !walking;
            const unsigned long long idle_m = __ballot(idle && !exhausted);
            if ((uint32_t)__popcll(idle_m) >= refill_idle || __ballot(walking) == 0ull) {
                if (idle && !exhausted && mine != ~0ull) {
                    LabHit h; h.t = best_t; h.prim = best_prim; h.flags = bad ? 1u : 0u; out[mine] = h; mine = ~0ull;
                }
                const unsigned long long idx = lab_fetch(idle && !exhausted, counter);
                if (idle && !exhausted) {
                    if (idx < n) {
                        const LabRay r = rays[idx];
                        o = rt_v3(r.o[0], r.o[1], r.o[2]); d = rt_v3(r.d[0], r.d[1], r.d[2]);
                        inv = rt_inv3(d);
                        frac = (r.time - ps.ms_time0) / (ps.ms_time1 - ps.ms_time0);
                        best_t = RT_INF; best_prim = RT_NONE; bad = false;
                        sp = 0; qh = 0u; qn = 0u; cur = LAB_NONE;
                        mine = idx; walking = true;
                        /* the root's own box, exactly as the reference tests it first (bvh.rs:32) */
                        if (rt_aabb_hit_fast<false>(ps.root_box, o, inv, t_min, best_t)) {
                            if (ps.root & LAB_LEAF) { my_q[0] = ps.root & ~LAB_LEAF; qn = 1u; }
                            else cur = ps.root;
                        } else walking = false;
                        if (rt_isnan(frac)) { bad = true; walking = false; } /* no moving-sphere arithmetic on a NaN shutter fraction here */
                    } else exhausted = true;
                }
            }
        }
        if (__ballot(walking) == 0ull) break;

        /* ---- which phase ----------------------------------------------------------------------------------------- */
        const bool can_box = walking && (cur != LAB_NONE || sp > 0) && qn + 2u <= (uint32_t)LAB_QCAP;
        const bool can_leaf = walking && qn > 0u;
        const uint32_t nb = (uint32_t)__popcll(__ballot(can_box)), nl = (uint32_t)__popcll(__ballot(can_leaf));
        if (nl >= leaf_votes || nb == 0u) {
            /* ---- leaf phase: the oldest pending group of every lane that has one ---------------------------------- */
            if (can_leaf) {
                const uint32_t g = my_q[(qh & (LAB_QCAP - 1u)) * LAB_W1_BLOCK];
                qh += 1u; qn -= 1u;
                const LabGroup& G = ps.groups[g];
                ++n_leaf;
                if (rt_aabb_hit_fast<false>(G.box, o, inv, t_min, best_t)) { /* the gate, at the closest hit as it is NOW */
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (i < (int)G.n) {
                            const LabPrim& P = G.p[i];
                            RtV3 c = rt_v3(P.c0[0], P.c0[1], P.c0[2]);
                            if (P.moving) c = c + frac * rt_v3(P.dc[0], P.dc[1], P.dc[2]); /* moving_sphere.rs:23-26 */
                            double t;
                            if (rt_sphere_root(c, P.radius, o, d, t_min, best_t, t)) { best_t = t; best_prim = P.id; }
                        }
                    }
                    if (rt_isnan(best_t)) { bad = true; walking = false; }
                }
            }
            ++w_leaf;
        } else {
            /* ---- box phase: one inner node per stepping lane ------------------------------------------------------ */
            if (can_box) {
                if (cur == LAB_NONE) { /* back to the nearest pushed right child */
                    --sp;
                    const uint32_t ref = my_ref[sp * LAB_W1_BLOCK];
                    const float ent = my_ent[sp * LAB_W1_BLOCK];
                    if ((double)ent < best_t) { /* conservative cull: ent <= the box's entry distance */
                        if (ref & LAB_LEAF) { my_q[((qh + qn) & (LAB_QCAP - 1u)) * LAB_W1_BLOCK] = ref & ~LAB_LEAF; qn += 1u; }
                        else cur = ref;
                    }
                }
                if (cur != LAB_NONE) {
                    ++n_box;
                    double el, xl, er, xr;
                    uint32_t l, r;
                    if constexpr (F32) {
                        const LabPNode32 P = LDS_INNER ? s_inner32[cur] : ps.inner32[cur];
                        double lb[6], rb[6];
#pragma unroll
                        for (int a = 0; a < 6; ++a) { lb[a] = (double)P.lb[a]; rb[a] = (double)P.rb[a]; }
                        lab_slab(lb, o, inv, t_min, el, xl);
                        lab_slab(rb, o, inv, t_min, er, xr);
                        l = P.l; r = P.r;
                    } else {
                        const LabPNode& P = LDS_INNER ? s_inner[cur] : ps.inner[cur];
                        lab_slab(P.lb, o, inv, t_min, el, xl);
                        lab_slab(P.rb, o, inv, t_min, er, xr);
                        l = P.l; r = P.r;
                    }
                    const bool pl = rt_vmin(xl, best_t) > el, pr = rt_vmin(xr, best_t) > er;
                    uint32_t next = LAB_NONE;
                    bool left_entered = false;
                    if (pl) {
                        if (l & LAB_LEAF) { my_q[((qh + qn) & (LAB_QCAP - 1u)) * LAB_W1_BLOCK] = l & ~LAB_LEAF; qn += 1u; }
                        else { next = l; left_entered = true; }
                    }
                    if (pr) {
                        if (left_entered) { /* after the left subtree: bvh.rs:38-47 */
                            my_ref[sp * LAB_W1_BLOCK] = r;
                            my_ent[sp * LAB_W1_BLOCK] = __double2float_rd(er);
                            ++sp;
                        } else if (r & LAB_LEAF) { my_q[((qh + qn) & (LAB_QCAP - 1u)) * LAB_W1_BLOCK] = r & ~LAB_LEAF; qn += 1u; }
                        else next = r;
                    }
                    cur = next;
                }
            }
            ++w_box;
        }
        if (walking && cur == LAB_NONE && sp == 0 && qn == 0u) walking = false;
    }
    if (mine != ~0ull) { LabHit h; h.t = best_t; h.prim = best_prim; h.flags = bad ? 1u : 0u; out[mine] = h; }
    if (stats) {
        atomicAdd(&stats[0], n_box);
        atomicAdd(&stats[2], n_leaf);
        if ((threadIdx.x & 63u) == 0u) { atomicAdd(&stats[1], w_box); atomicAdd(&stats[3], w_leaf); }
    }
}

/* ------------------------------------------------------------------------------------------------- W2 -- */

/* the phased walk of rt_walk2.h (every scene): the wave votes for box work or leaf work; lanes refill like W0's */
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK, 3) void lab_trace_w2(RtSceneView sc, RtW2View w2, const LabRay* __restrict__ rays, unsigned long long n,
                                                            LabHit* __restrict__ out, unsigned long long* counter, uint32_t refill_idle,
                                                            uint32_t leaf_votes, unsigned long long* stats, uint32_t box_steps) {
    __shared__ uint32_t s_ref[RT_W2_STACK * RT_BLOCK];
    __shared__ float s_ent[RT_W2_STACK * RT_BLOCK];
    __shared__ uint32_t s_q[RT_W2_QCAP * RT_BLOCK];
    RtW2Stack<RT_BLOCK> stk;
    stk.ref = s_ref + threadIdx.x; stk.ent = s_ent + threadIdx.x; stk.q = s_q + threadIdx.x;
    stk.sp = 0; stk.qh = 0u; stk.qn = 0u;
    RtGlobalNodes ns{sc.nodes};
    RtWalk k;
    RtW2Lane L; L.cur = RT_W2_NONE; L.nan_seen = false;
    RtRng rng = rt_rng_make(0u, 0u, 0u, 0u, RT_DOMAIN_RENDER);
    RtRngMark mark = rt_rng_mark(rng);
    RtRay w; w.o = w.d = rt_v3(0, 0, 0); w.time = 0.0;
    unsigned long long mine = ~0ull;
    bool walking = false, exhausted = false, redone = false;
    unsigned long long n_box = 0, n_leaf = 0, w_box = 0, w_leaf = 0, n_other = 0, n_redo = 0;
    for (;;) {
        {
            const bool idle = !walking;
            const unsigned long long idle_m = __ballot(idle && !exhausted);
            if ((uint32_t)__popcll(idle_m) >= refill_idle || __ballot(walking) == 0ull) {
                if (idle && !exhausted && mine != ~0ull) {
                    LabHit h; h.t = k.best_t; h.prim = k.best_prim; h.flags = redone ? 1u : 0u; out[mine] = h; mine = ~0ull;
                }
                const unsigned long long idx = lab_fetch(idle && !exhausted, counter);
                if (idle && !exhausted) {
                    if (idx < n) {
                        const LabRay r = rays[idx];
                        w.o = rt_v3(r.o[0], r.o[1], r.o[2]); w.d = rt_v3(r.d[0], r.d[1], r.d[2]); w.time = r.time;
                        rng = rt_rng_make((uint32_t)idx, (uint32_t)(idx >> 32), 0u, 0u, RT_DOMAIN_RENDER);
                        mark = rt_rng_mark(rng);
                        rt_w2_begin(w2, L, k, w, 0.001, RT_INF, stk);
                        mine = idx; walking = true; redone = false;
                    } else exhausted = true;
                }
            }
        }
        if (__ballot(walking) == 0ull) break;
        const bool can_box = walking && rt_w2_is_boxwork(L.cur) && stk.qn + 2u <= (uint32_t)RT_W2_QCAP;
        const bool can_leaf = walking && (stk.qn > 0u || rt_w2_is_other(L.cur));
        const uint32_t nb = (uint32_t)__popcll(__ballot(can_box)), nl = (uint32_t)__popcll(__ballot(can_leaf));
        if (nl >= leaf_votes || nb == 0u) {
            if (can_leaf) {
                if (stk.qn > 0u) { rt_w2_group_step<Cfg>(sc, ns, L, k, stk); ++n_leaf; }
                else { rt_w2_other_step<Cfg>(sc, ns, w2, L, k, rng, stk); ++n_other; }
            }
            ++w_leaf;
        } else {
            if (can_box) {
                rt_w2_box_step<RtW2Stack<RT_BLOCK>, RT_BLOCK>(w2, L, k, stk); ++n_box;
                for (uint32_t extra = 1; extra < box_steps; ++extra) { /* more box work on the same vote (rt_kernel_plain.h: RT_PW_BOX_STEPS) */
                    rt_w2_refetch<RtW2Stack<RT_BLOCK>, RT_BLOCK>(L, k, stk);
                    if (!(rt_w2_is_boxwork(L.cur) && stk.qn + 2u <= (uint32_t)RT_W2_QCAP)) break;
                    rt_w2_box_step<RtW2Stack<RT_BLOCK>, RT_BLOCK>(w2, L, k, stk); ++n_box;
                }
            }
            ++w_box;
        }
        if (walking) {
            if (L.nan_seen) { /* the closest hit turned NaN: the ray is handed back (flag bit 0) and answered by the classic walk */
                L.cur = RT_W2_NONE; L.nan_seen = false; redone = true; ++n_redo;
                stk.sp = 0; stk.qn = 0u;
                walking = false;
            } else {
                rt_w2_refetch<RtW2Stack<RT_BLOCK>, RT_BLOCK>(L, k, stk);
                if (rt_w2_done(L, stk)) walking = false;
            }
        }
    }
    if (mine != ~0ull) { LabHit h; h.t = k.best_t; h.prim = k.best_prim; h.flags = redone ? 1u : 0u; out[mine] = h; }
    if (stats) {
        atomicAdd(&stats[0], n_box);
        atomicAdd(&stats[2], n_leaf);
        atomicAdd(&stats[4], n_other);
        atomicAdd(&stats[5], n_redo);
        if ((threadIdx.x & 63u) == 0u) { atomicAdd(&stats[1], w_box); atomicAdd(&stats[3], w_leaf); }
    }
}

/* rays W1 could not answer (flag bit 0): the product's walk from scratch, one thread per ray */
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK, 2) void lab_fallback(RtSceneView sc, const LabRay* __restrict__ rays, unsigned long long n, LabHit* __restrict__ out) {
    __shared__ uint32_t stack_mem[RT_STACK_CAP * RT_BLOCK];
    LdsStack stk;
    stk.base = stack_mem + threadIdx.x;
    stk.sp = 0;
    const unsigned long long i = (unsigned long long)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n || (out[i].flags & 1u) == 0u) return;
    RtGlobalNodes ns{sc.nodes};
    const LabRay r = rays[i];
    RtRay w; w.o = rt_v3(r.o[0], r.o[1], r.o[2]); w.d = rt_v3(r.d[0], r.d[1], r.d[2]); w.time = r.time;
    RtRng rng = rt_rng_make((uint32_t)i, (uint32_t)(i >> 32), 0u, 0u, RT_DOMAIN_RENDER);
    double t; uint32_t prim, scope;
    rt_traverse_stack<Cfg, true>(sc, ns, sc.root, w, 0.001, RT_INF, rng, stk, t, prim, scope);
    LabHit h; h.t = t; h.prim = prim; h.flags = 1u;
    out[i] = h;
}

} // namespace

/* ------------------------------------------------------------------------------------------------------ host -- */

struct rt1w_lab {
    rt1w_context* ctx = nullptr;
    RtSceneView view{};
    int device = 0, cus = 0;
    int variant = 3;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    LabRay* d_rays = nullptr; unsigned long long n_rays = 0;
    LabHit* d_hits = nullptr;
    unsigned long long* d_counter = nullptr; /* [0] ray counter, [1..8] stats */
    /* W1 */
    bool w1_ok = false; std::string w1_why;
    LabPNode* d_inner = nullptr; LabGroup* d_groups = nullptr; LabPNode32* d_inner32 = nullptr;
    LabW1Scene w1{};
    uint32_t n_inner = 0, n_groups = 0, stack_need = 0;
    /* W3 */
    RtNode* d_w3_nodes = nullptr; RtPairRec* d_w3_pairs = nullptr;
    RtSceneView w3_view{};
    uint32_t n_w3_pairs = 0;
    /* W2 */
    RtW2Inner* d_w2_inner = nullptr; uint32_t* d_w2_wref = nullptr;
    RtW2View w2{};
    bool w2_ok = false;
    /* W4 */
    uint8_t* d_w4_cls = nullptr; bool w4_ok = false, sphere_media = false; uint32_t n_nodes = 0;
};

namespace {

/* flat pre-order nodes -> inner pair records + leaf groups; false (with a reason) if the scene is outside W1's scope */
struct W1Builder {
    const std::vector<RtNode>& N;
    std::vector<LabPNode>& inner;
    std::vector<LabGroup>& groups;
    bool ok = true;
    std::string why;
    uint32_t kind(uint32_t i) const { return N[i].kind & RT_KIND_MASK; }
    bool is_prim(uint32_t i) const { return kind(i) == RT_SPHERE || kind(i) == RT_MSPHERE; }
    bool is_group(uint32_t i) const {
        if (kind(i) == RT_BVH1) return is_prim(N[i].a);
        if (kind(i) == RT_BVH2) return is_prim(N[i].a) && is_prim(N[i].b);
        return false;
    }
    LabPrim prim_of(uint32_t i) const {
        LabPrim p; memset(&p, 0, sizeof p);
        const RtNode& n = N[i];
        p.id = i;
        if (kind(i) == RT_MSPHERE) {
            p.moving = 1u;
            for (int a = 0; a < 3; ++a) { p.c0[a] = n.d[a]; p.dc[a] = n.d[3 + a] - n.d[a]; } /* center1 - center0: the reference's own subtraction */
            p.radius = n.e[2];
        } else {
            for (int a = 0; a < 3; ++a) { p.c0[a] = n.d[a]; p.dc[a] = 0.0; }
            p.radius = n.d[3];
        }
        return p;
    }
    uint32_t go(uint32_t i) {
        if (!ok) return LAB_NONE;
        if (is_group(i)) {
            LabGroup g; memset(&g, 0, sizeof g);
            for (int a = 0; a < 6; ++a) g.box[a] = N[i].d[a];
            g.n = kind(i) == RT_BVH2 ? 2u : 1u;
            g.p[0] = prim_of(N[i].a);
            if (g.n == 2u) g.p[1] = prim_of(N[i].b);
            groups.push_back(g);
            return LAB_LEAF | (uint32_t)(groups.size() - 1);
        }
        if (kind(i) == RT_BVH1) { ok = false; why = "BVHChild::One over a BVH node"; return LAB_NONE; } /* not built by BVHNode::new, legal through the ABI */
        if (kind(i) != RT_BVH2 || is_prim(N[i].a) || is_prim(N[i].b)) { ok = false; why = "a BVHChild::Two with one primitive and one subtree"; return LAB_NONE; }
        const uint32_t me = (uint32_t)inner.size();
        inner.push_back(LabPNode());
        memset(&inner[me], 0, sizeof(LabPNode));
        const uint32_t a = N[i].a, b = N[i].b;
        for (int q = 0; q < 6; ++q) { inner[me].lb[q] = N[a].d[q]; inner[me].rb[q] = N[b].d[q]; }
        const uint32_t l = go(a);
        const uint32_t r = go(b);
        inner[me].l = l; inner[me].r = r;
        return me;
    }
};
bool build_w1(const std::vector<RtNode>& N, uint32_t root, std::vector<LabPNode>& inner, std::vector<LabGroup>& groups, LabW1Scene& sc,
              std::string& why) {
    W1Builder b{N, inner, groups};
    bool seen_ms = false;
    double t0 = 0.0, t1 = 1.0;
    for (uint32_t i = 0; i < N.size(); ++i) {
        const uint32_t k = b.kind(i);
        if (k == RT_BVH2 || k == RT_BVH1) continue;
        if (!b.is_prim(i)) { why = "a node that is neither a BVH node nor a sphere (W1 covers sphere scenes only so far)"; return false; }
        if (N[i].kind & RT_LEAF_FLIPPED) { why = "FlipFace"; return false; }
        if (k == RT_MSPHERE) {
            if (!seen_ms) { t0 = N[i].e[0]; t1 = N[i].e[1]; seen_ms = true; }
            else if (memcmp(&t0, &N[i].e[0], 8) != 0 || memcmp(&t1, &N[i].e[1], 8) != 0) { why = "moving spheres with different shutter intervals"; return false; }
        }
    }
    sc.ms_time0 = t0; sc.ms_time1 = t1;
    const uint32_t k0 = b.kind(root);
    if (k0 != RT_BVH2 && k0 != RT_BVH1) { why = "the root is not a BVH node"; return false; }
    for (int a = 0; a < 6; ++a) sc.root_box[a] = N[root].d[a];
    sc.root = b.go(root);
    why = b.why;
    return b.ok;
}

template <class T>
bool lab_upload(T** dst, const std::vector<T>& v) {
    *dst = nullptr;
    const size_t bytes = (v.size() ? v.size() : 1) * sizeof(T);
    if (!lab_ok(hipMalloc((void**)dst, bytes), "hipMalloc")) return false;
    if (v.size() && !lab_ok(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy")) return false;
    return true;
}

} // namespace

extern "C" {

int rt1w_lab_create(rt1w_context* c, const rt1w_scene* s, rt1w_lab** out) {
    if (!c || !s || !out) { rt1wlab::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (!s->committed) { rt1wlab::set_error("scene not committed"); return RT1W_ERR_STATE; }
    rt1w_lab* l = new (std::nothrow) rt1w_lab();
    if (!l) { rt1wlab::set_error("out of memory"); return RT1W_ERR_NOMEM; }
    l->ctx = c;
    memcpy(&l->view, rt1w_internal_view(c), sizeof(RtSceneView));
    l->device = rt1w_internal_device(c);
    hipDeviceProp_t prop;
    if (!lab_ok(hipSetDevice(l->device), "hipSetDevice") || !lab_ok(hipGetDeviceProperties(&prop, l->device), "hipGetDeviceProperties") ||
        !lab_ok(hipStreamCreate(&l->stream), "hipStreamCreate") || !lab_ok(hipEventCreate(&l->ev0), "hipEventCreate") ||
        !lab_ok(hipEventCreate(&l->ev1), "hipEventCreate") || !lab_ok(hipMalloc((void**)&l->d_counter, 16 * sizeof(unsigned long long)), "hipMalloc")) {
        rt1w_lab_destroy(l); return RT1W_ERR_DEVICE;
    }
    l->cus = prop.multiProcessorCount;
    l->variant = (!s->has_media && s->scope_depth == 0u) ? 5 : (s->has_media ? 3 : 2);
    std::vector<LabPNode> inner; std::vector<LabGroup> groups;
    l->w1_ok = build_w1(s->flat_nodes, s->flat_root, inner, groups, l->w1, l->w1_why) && s->stack_need <= (uint32_t)LAB_W1_STACK;
    if (l->w1_ok) {
        std::vector<LabPNode32> inner32(inner.size());
        for (size_t i = 0; i < inner.size(); ++i) {
            memset(&inner32[i], 0, sizeof(LabPNode32));
            for (int a = 0; a < 6; ++a) {
                const bool is_min = a < 3;
                for (int side = 0; side < 2; ++side) {
                    const double x = side ? inner[i].rb[a] : inner[i].lb[a];
                    float f = (float)x;
                    if (is_min ? ((double)f > x) : ((double)f < x)) f = nextafterf(f, is_min ? -INFINITY : INFINITY); /* outward */
                    (side ? inner32[i].rb : inner32[i].lb)[a] = f;
                }
            }
            inner32[i].l = inner[i].l; inner32[i].r = inner[i].r;
        }
        if (!lab_upload(&l->d_inner, inner) || !lab_upload(&l->d_groups, groups) || !lab_upload(&l->d_inner32, inner32)) { rt1w_lab_destroy(l); return RT1W_ERR_DEVICE; }
        l->w1.inner = l->d_inner; l->w1.groups = l->d_groups; l->w1.inner32 = l->d_inner32;
        l->n_inner = (uint32_t)inner.size(); l->n_groups = (uint32_t)groups.size(); l->stack_need = s->stack_need;
    }
    {
        std::vector<RtW2Inner> w2i; std::vector<uint32_t> wref; uint32_t root_ref = 0u;
        rt_walk2_build(s->flat_nodes, s->flat_root, w2i, wref, root_ref);
        if (!lab_upload(&l->d_w2_inner, w2i) || !lab_upload(&l->d_w2_wref, wref)) { rt1w_lab_destroy(l); return RT1W_ERR_DEVICE; }
        l->w2.inner = l->d_w2_inner; l->w2.wref = l->d_w2_wref; l->w2.root = root_ref; l->w2.n_inner = (uint32_t)w2i.size();
        l->w2_ok = s->stack_need + 2u <= (uint32_t)RT_W2_STACK;
        l->stack_need = s->stack_need;
    }
    {
        std::vector<RtNode> patched(s->flat_nodes);
        std::vector<RtPairRec> pairs;
        rt_pairs_build(patched, pairs);
        patched.push_back(RtNode()); /* the spare record behind the array (scene.cpp) */
        if (pairs.empty()) pairs.push_back(RtPairRec());
        if (!lab_upload(&l->d_w3_nodes, patched) || !lab_upload(&l->d_w3_pairs, pairs)) { rt1w_lab_destroy(l); return RT1W_ERR_DEVICE; }
        l->w3_view = l->view; l->w3_view.nodes = l->d_w3_nodes;
        l->n_w3_pairs = (uint32_t)pairs.size();
    }
    {
        /* W4: the sort key of a node (0 BVH node, 1 sphere, 2 rect, 3 moving sphere, 4 wrapper, 5 medium) as one byte per node */
        std::vector<uint8_t> cls(s->flat_nodes.size());
        for (size_t i = 0; i < cls.size(); ++i) {
            const uint32_t km = s->flat_nodes[i].kind & RT_KIND_MASK;
            cls[i] = km <= RT_BVH1 ? 0 : km == RT_SPHERE ? 1 : km == RT_MSPHERE ? 3 : km <= RT_YZ ? 2 : km <= RT_FLIP ? 4 : 5;
        }
        if (!lab_upload(&l->d_w4_cls, cls)) { rt1w_lab_destroy(l); return RT1W_ERR_DEVICE; }
        l->n_nodes = (uint32_t)cls.size();
        l->sphere_media = s->has_media && s->media_bare_spheres;
        l->w4_ok = l->n_nodes <= (uint32_t)LAB_W4_NODE_CAP && s->stack_need + 2u <= (uint32_t)LAB_W4_STACK && (!s->has_media || l->sphere_media);
        l->stack_need = s->stack_need;
    }
    *out = l;
    return RT1W_OK;
}

void rt1w_lab_destroy(rt1w_lab* l) {
    if (!l) return;
    if (l->d_rays) (void)hipFree(l->d_rays);
    if (l->d_hits) (void)hipFree(l->d_hits);
    if (l->d_counter) (void)hipFree(l->d_counter);
    if (l->d_inner) (void)hipFree(l->d_inner);
    if (l->d_groups) (void)hipFree(l->d_groups);
    if (l->d_inner32) (void)hipFree(l->d_inner32);
    if (l->d_w3_nodes) (void)hipFree(l->d_w3_nodes);
    if (l->d_w3_pairs) (void)hipFree(l->d_w3_pairs);
    if (l->d_w4_cls) (void)hipFree(l->d_w4_cls);
    if (l->d_w2_inner) (void)hipFree(l->d_w2_inner);
    if (l->d_w2_wref) (void)hipFree(l->d_w2_wref);
    if (l->ev0) (void)hipEventDestroy(l->ev0);
    if (l->ev1) (void)hipEventDestroy(l->ev1);
    if (l->stream) (void)hipStreamDestroy(l->stream);
    delete l;
}

int rt1w_lab_info(const rt1w_lab* l, uint32_t out[4]) {
    if (!l || !out) { rt1wlab::set_error("null argument"); return RT1W_ERR_INVALID; }
    out[0] = l->w1_ok ? 1u : 0u; out[1] = l->n_inner; out[2] = l->n_groups; out[3] = (uint32_t)l->variant;
    if (!l->w1_ok) rt1wlab::set_error("W1 not available for this scene: " + l->w1_why);
    return RT1W_OK;
}

/* out_ms[4] = kernel ms of {own record, quad-shared} x {independent, dependent} indices; grid = blocks_per_cu x CUs of 256 threads */
int rt1w_lab_gather_probe(rt1w_lab* l, uint32_t iters, uint32_t blocks_per_cu, double out_ms[4], uint64_t* records_per_launch) {
    if (!l || !out_ms) { rt1wlab::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (!lab_ok(hipSetDevice(l->device), "hipSetDevice")) return RT1W_ERR_DEVICE;
    const int grid = l->cus * (int)(blocks_per_cu ? blocks_per_cu : 4u);
    for (int v = 0; v < 4; ++v) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(l->ev0, l->stream);
            if (v == 0) hipLaunchKernelGGL((lab_gather_kernel<0, 0>), dim3(grid), dim3(256), 0, l->stream, l->view.nodes, l->view.n_nodes, iters, l->d_counter);
            else if (v == 1) hipLaunchKernelGGL((lab_gather_kernel<1, 0>), dim3(grid), dim3(256), 0, l->stream, l->view.nodes, l->view.n_nodes, iters, l->d_counter);
            else if (v == 2) hipLaunchKernelGGL((lab_gather_kernel<0, 1>), dim3(grid), dim3(256), 0, l->stream, l->view.nodes, l->view.n_nodes, iters, l->d_counter);
            else hipLaunchKernelGGL((lab_gather_kernel<1, 1>), dim3(grid), dim3(256), 0, l->stream, l->view.nodes, l->view.n_nodes, iters, l->d_counter);
            (void)hipEventRecord(l->ev1, l->stream);
            if (!lab_ok(hipGetLastError(), "launch(gather)") || !lab_ok(hipEventSynchronize(l->ev1), "gather kernel")) return RT1W_ERR_DEVICE;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, l->ev0, l->ev1);
            if (ms < best) best = ms;
        }
        out_ms[v] = best;
    }
    if (records_per_launch) *records_per_launch = (uint64_t)grid * 256ull * iters;
    return RT1W_OK;
}

int rt1w_lab_dump_rays(rt1w_lab* l, const rt1w_render_params* p, uint32_t n_bounces, double* out_host) {
    if (!l || !p || !out_host || n_bounces == 0u) { rt1wlab::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (p->tile_w == 0 || p->tile_h == 0 || p->x0 + p->tile_w > p->width || p->y0 + p->tile_h > p->height || p->spp == 0) { rt1wlab::set_error("bad tile"); return RT1W_ERR_INVALID; }
    RtFrame f; memset(&f, 0, sizeof f);
    f.width = p->width; f.height = p->height; f.x0 = p->x0; f.y0 = p->y0; f.tile_w = p->tile_w; f.tile_h = p->tile_h;
    f.spp = p->spp; f.sample_offset = p->sample_offset; f.max_depth = p->max_depth ? p->max_depth : 50u; f.global_seed = p->global_seed;
    f.chunk = 1u; f.n_chunks = p->spp; f.strip_rows = 0u; f.strip_period = 0u; f.probe = 0u;
    const unsigned long long n_paths = (unsigned long long)f.tile_w * f.tile_h * f.spp;
    const size_t bytes = (size_t)n_bounces * n_paths * 8u * sizeof(double);
    double* d_out = nullptr;
    if (!lab_ok(hipSetDevice(l->device), "hipSetDevice") || !lab_ok(hipMalloc((void**)&d_out, bytes), "hipMalloc(rays)")) return RT1W_ERR_NOMEM;
    (void)hipMemsetAsync(d_out, 0, bytes, l->stream);
    const unsigned long long items = rt_item_count(f);
    const unsigned grid = (unsigned)((items + RT_BLOCK - 1) / RT_BLOCK);
    hipLaunchKernelGGL(lab_dump_kernel, dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, f, n_bounces, n_paths, d_out);
    bool ok = lab_ok(hipGetLastError(), "launch(dump)") && lab_ok(hipStreamSynchronize(l->stream), "dump kernel") &&
              lab_ok(hipMemcpy(out_host, d_out, bytes, hipMemcpyDeviceToHost), "hipMemcpy(rays)");
    (void)hipFree(d_out);
    return ok ? RT1W_OK : RT1W_ERR_DEVICE;
}

int rt1w_lab_set_rays(rt1w_lab* l, const double* rays, uint64_t n) {
    if (!l || !rays || n == 0) { rt1wlab::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (!lab_ok(hipSetDevice(l->device), "hipSetDevice")) return RT1W_ERR_DEVICE;
    if (l->d_rays) { (void)hipFree(l->d_rays); l->d_rays = nullptr; }
    if (l->d_hits) { (void)hipFree(l->d_hits); l->d_hits = nullptr; }
    if (!lab_ok(hipMalloc((void**)&l->d_rays, n * sizeof(LabRay)), "hipMalloc(rays)") || !lab_ok(hipMalloc((void**)&l->d_hits, n * sizeof(LabHit)), "hipMalloc(hits)") ||
        !lab_ok(hipMemcpy(l->d_rays, rays, n * sizeof(LabRay), hipMemcpyHostToDevice), "hipMemcpy(rays)")) return RT1W_ERR_DEVICE;
    l->n_rays = n;
    return RT1W_OK;
}

/* params: [0] refill_idle (lanes), [1] leaf_votes (W1, W2), [2] blocks per CU (0: the occupancy query's), [3] box steps per vote (W2) */
int rt1w_lab_trace(rt1w_lab* l, int mode, const uint32_t params[4], int repeats, double* out_t, uint32_t* out_prim, uint32_t* out_flags,
                   double* ms_best, uint64_t stats_out[8]) {
    if (!l || !l->d_rays) { rt1wlab::set_error("no rays set"); return RT1W_ERR_STATE; }
    if (mode == 1 && !l->w1_ok) { rt1wlab::set_error("W1 not available for this scene: " + l->w1_why); return RT1W_ERR_UNSUPPORTED; }
    if (mode == 2 && (!l->w1_ok || l->n_inner > LAB_LDS_INNER || l->stack_need > LAB_LDS_STACK)) { rt1wlab::set_error("W1 with LDS-resident inner records: scene too big or W1 unavailable"); return RT1W_ERR_UNSUPPORTED; }
    if (mode == 4 && (!l->w1_ok || l->stack_need > LAB_LDS_STACK)) { rt1wlab::set_error("W1 with f32 inner boxes: W1 unavailable or tree too deep for the experiment's stack"); return RT1W_ERR_UNSUPPORTED; }
    if (mode == 5 && (!l->w1_ok || l->stack_need > LAB_LDS_STACK || l->n_inner > LAB_LDS_INNER)) { rt1wlab::set_error("W1c with LDS-resident inner records: not for this scene"); return RT1W_ERR_UNSUPPORTED; }
    if (mode == 6 && !l->w2_ok) { rt1wlab::set_error("phased walk: the scene needs a deeper stack than RT_W2_STACK"); return RT1W_ERR_UNSUPPORTED; }
    if ((mode == 11 || mode == 13) && !l->w4_ok) { rt1wlab::set_error("W4: more than 8192 nodes, a deeper stack than the experiment's, or a medium whose boundary is not a bare sphere"); return RT1W_ERR_UNSUPPORTED; }
    if (mode == 12 && !l->sphere_media) { rt1wlab::set_error("W0c of the sphere-media kernels: not such a scene"); return RT1W_ERR_UNSUPPORTED; }
    if (mode < 0 || mode > 13) { rt1wlab::set_error("unknown walk"); return RT1W_ERR_INVALID; }
    if (!lab_ok(hipSetDevice(l->device), "hipSetDevice")) return RT1W_ERR_DEVICE;
    const uint32_t refill = params && params[0] ? params[0] : 16u;
    const uint32_t votes = params && params[1] ? params[1] : 24u;
    const uint32_t box_steps = params && params[3] ? params[3] : 1u;
    int per_cu = 0;
    const void* fn = nullptr;
    typedef RtCfgSphereMedia<RtCfgV3> LabCfgSM;
    if (mode == 11) fn = l->variant == 5 ? (const void*)lab_trace_w4<RtCfgV5, 4> : (l->variant == 3 ? (const void*)lab_trace_w4<LabCfgSM, 4> : (const void*)lab_trace_w4<RtCfgV2, 4>);
    else if (mode == 13) fn = l->variant == 5 ? (const void*)lab_trace_w4<RtCfgV5, 0> : (l->variant == 3 ? (const void*)lab_trace_w4<LabCfgSM, 0> : (const void*)lab_trace_w4<RtCfgV2, 0>);
    else if (mode == 12) fn = (const void*)lab_trace_w0<LabCfgSM, 4, true>;
    else if (mode == 10) fn = l->variant == 5 ? (const void*)lab_trace_w0<RtCfgV5, 4, true> : (l->variant == 3 ? (const void*)lab_trace_w0<RtCfgV3, 4, true> : (const void*)lab_trace_w0<RtCfgV2, 4, true>);
    else if (mode == 8) fn = l->variant == 5 ? (const void*)lab_trace_w3<RtCfgV5, 4> : (l->variant == 3 ? (const void*)lab_trace_w3<RtCfgV3, 4> : (const void*)lab_trace_w3<RtCfgV2, 4>);
    else if (mode == 9) fn = l->variant == 5 ? (const void*)lab_trace_w3<RtCfgV5, 0> : (l->variant == 3 ? (const void*)lab_trace_w3<RtCfgV3, 0> : (const void*)lab_trace_w3<RtCfgV2, 0>);
    else if (mode == 7) fn = l->variant == 5 ? (const void*)lab_trace_w0<RtCfgV5, 4> : (l->variant == 3 ? (const void*)lab_trace_w0<RtCfgV3, 4> : (const void*)lab_trace_w0<RtCfgV2, 4>);
    else if (mode == 0) fn = l->variant == 5 ? (const void*)lab_trace_w0<RtCfgV5> : (l->variant == 3 ? (const void*)lab_trace_w0<RtCfgV3> : (const void*)lab_trace_w0<RtCfgV2>);
    else if (mode == 3) fn = l->variant == 5 ? (const void*)lab_trace_w0q<RtCfgV5> : (l->variant == 3 ? (const void*)lab_trace_w0q<RtCfgV3> : (const void*)lab_trace_w0q<RtCfgV2>);
    else if (mode == 6) fn = l->variant == 5 ? (const void*)lab_trace_w2<RtCfgV5> : (l->variant == 3 ? (const void*)lab_trace_w2<RtCfgV3> : (const void*)lab_trace_w2<RtCfgV2>);
    else if (mode == 4) fn = (const void*)lab_trace_w1<false, true>;
    else if (mode == 5) fn = (const void*)lab_trace_w1<true, true>;
    else fn = mode == 1 ? (const void*)lab_trace_w1<false> : (const void*)lab_trace_w1<true>;
    if (!lab_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, RT_BLOCK, 0), "occupancy")) return RT1W_ERR_DEVICE;
    if (per_cu < 1) per_cu = 1;
    if (params && params[2]) per_cu = (int)params[2] < per_cu ? (int)params[2] : per_cu;
    const int grid = l->cus * per_cu;
    double best = 1e30;
    for (int rep = 0; rep < (repeats > 0 ? repeats : 1); ++rep) {
        (void)hipMemsetAsync(l->d_counter, 0, 16 * sizeof(unsigned long long), l->stream);
        (void)hipEventRecord(l->ev0, l->stream);
        if (mode == 11 || mode == 13) {
            const uint32_t idle = params && params[0] ? params[0] : 160u;
#define LAB_W4(CFG, BS) hipLaunchKernelGGL((lab_trace_w4<CFG, BS>), dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_w4_cls, l->d_rays, l->n_rays, l->d_hits, l->d_counter, idle, l->d_counter + 1)
            if (mode == 11) { if (l->variant == 5) LAB_W4(RtCfgV5, 4); else if (l->variant == 3) LAB_W4(LabCfgSM, 4); else LAB_W4(RtCfgV2, 4); }
            else { if (l->variant == 5) LAB_W4(RtCfgV5, 0); else if (l->variant == 3) LAB_W4(LabCfgSM, 0); else LAB_W4(RtCfgV2, 0); }
#undef LAB_W4
        } else if (mode == 12) {
            hipLaunchKernelGGL((lab_trace_w0<LabCfgSM, 4, true>), dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
        } else if (mode == 10) {
            if (l->variant == 5) hipLaunchKernelGGL((lab_trace_w0<RtCfgV5, 4, true>), dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
            else if (l->variant == 3) hipLaunchKernelGGL((lab_trace_w0<RtCfgV3, 4, true>), dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
            else hipLaunchKernelGGL((lab_trace_w0<RtCfgV2, 4, true>), dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
        } else if (mode == 8 || mode == 9) {
#define LAB_W3(CFG, BS) hipLaunchKernelGGL((lab_trace_w3<CFG, BS>), dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->w3_view, l->d_w3_pairs, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1)
            if (mode == 8) { if (l->variant == 5) LAB_W3(RtCfgV5, 4); else if (l->variant == 3) LAB_W3(RtCfgV3, 4); else LAB_W3(RtCfgV2, 4); }
            else { if (l->variant == 5) LAB_W3(RtCfgV5, 0); else if (l->variant == 3) LAB_W3(RtCfgV3, 0); else LAB_W3(RtCfgV2, 0); }
#undef LAB_W3
        } else if (mode == 7) {
            if (l->variant == 5) hipLaunchKernelGGL((lab_trace_w0<RtCfgV5, 4>), dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
            else if (l->variant == 3) hipLaunchKernelGGL((lab_trace_w0<RtCfgV3, 4>), dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
            else hipLaunchKernelGGL((lab_trace_w0<RtCfgV2, 4>), dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
        } else if (mode == 0) {
            if (l->variant == 5) hipLaunchKernelGGL(lab_trace_w0<RtCfgV5>, dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
            else if (l->variant == 3) hipLaunchKernelGGL(lab_trace_w0<RtCfgV3>, dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
            else hipLaunchKernelGGL(lab_trace_w0<RtCfgV2>, dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
        } else if (mode == 6) {
            if (l->variant == 5) hipLaunchKernelGGL(lab_trace_w2<RtCfgV5>, dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->w2, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, votes, l->d_counter + 1, box_steps);
            else if (l->variant == 3) hipLaunchKernelGGL(lab_trace_w2<RtCfgV3>, dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->w2, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, votes, l->d_counter + 1, box_steps);
            else hipLaunchKernelGGL(lab_trace_w2<RtCfgV2>, dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->w2, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, votes, l->d_counter + 1, box_steps);
        } else if (mode == 3) {
            if (l->variant == 5) hipLaunchKernelGGL(lab_trace_w0q<RtCfgV5>, dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
            else if (l->variant == 3) hipLaunchKernelGGL(lab_trace_w0q<RtCfgV3>, dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
            else hipLaunchKernelGGL(lab_trace_w0q<RtCfgV2>, dim3(grid), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, l->d_counter + 1);
        } else {
            if (mode == 5) hipLaunchKernelGGL((lab_trace_w1<true, true>), dim3(grid), dim3(LAB_W1_BLOCK), 0, l->stream, l->w1, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, votes, l->d_counter + 1, l->n_inner);
            else if (mode == 4) hipLaunchKernelGGL((lab_trace_w1<false, true>), dim3(grid), dim3(LAB_W1_BLOCK), 0, l->stream, l->w1, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, votes, l->d_counter + 1, l->n_inner);
            else if (mode == 1) hipLaunchKernelGGL(lab_trace_w1<false>, dim3(grid), dim3(LAB_W1_BLOCK), 0, l->stream, l->w1, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, votes, l->d_counter + 1, l->n_inner);
            else hipLaunchKernelGGL(lab_trace_w1<true>, dim3(grid), dim3(LAB_W1_BLOCK), 0, l->stream, l->w1, l->d_rays, l->n_rays, l->d_hits, l->d_counter, refill, votes, l->d_counter + 1, l->n_inner);
        }
        (void)hipEventRecord(l->ev1, l->stream);
        if (!lab_ok(hipGetLastError(), "launch(trace)") || !lab_ok(hipEventSynchronize(l->ev1), "trace kernel")) return RT1W_ERR_DEVICE;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, l->ev0, l->ev1);
        if (ms < best) best = ms;
    }
    if (mode == 6) {
        const unsigned g2 = (unsigned)((l->n_rays + RT_BLOCK - 1) / RT_BLOCK);
        if (l->variant == 5) hipLaunchKernelGGL(lab_fallback<RtCfgV5>, dim3(g2), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits);
        else if (l->variant == 3) hipLaunchKernelGGL(lab_fallback<RtCfgV3>, dim3(g2), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits);
        else hipLaunchKernelGGL(lab_fallback<RtCfgV2>, dim3(g2), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits);
        if (!lab_ok(hipStreamSynchronize(l->stream), "fallback kernel")) return RT1W_ERR_DEVICE;
    }
    if (mode == 1 || mode == 2 || mode == 4 || mode == 5) { /* rays the pair walk handed back: the product's walk answers them (not timed: a handful per million) */
        const unsigned g2 = (unsigned)((l->n_rays + RT_BLOCK - 1) / RT_BLOCK);
        if (l->variant == 5) hipLaunchKernelGGL(lab_fallback<RtCfgV5>, dim3(g2), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits);
        else hipLaunchKernelGGL(lab_fallback<RtCfgV2>, dim3(g2), dim3(RT_BLOCK), 0, l->stream, l->view, l->d_rays, l->n_rays, l->d_hits);
        if (!lab_ok(hipStreamSynchronize(l->stream), "fallback kernel")) return RT1W_ERR_DEVICE;
    }
    if (ms_best) *ms_best = best;
    if (stats_out) {
        unsigned long long h[16];
        (void)hipMemcpy(l->d_counter + 8, &per_cu, sizeof(int), hipMemcpyHostToDevice); /* stats[7] = workgroups per CU of the launch */
        if (!lab_ok(hipMemcpy(h, l->d_counter, sizeof h, hipMemcpyDeviceToHost), "hipMemcpy(stats)")) return RT1W_ERR_DEVICE;
        for (int i = 0; i < 8; ++i) stats_out[i] = h[1 + i];
    }
    if (out_t || out_prim || out_flags) {
        std::vector<LabHit> h(l->n_rays);
        if (!lab_ok(hipMemcpy(h.data(), l->d_hits, l->n_rays * sizeof(LabHit), hipMemcpyDeviceToHost), "hipMemcpy(hits)")) return RT1W_ERR_DEVICE;
        for (unsigned long long i = 0; i < l->n_rays; ++i) {
            if (out_t) out_t[i] = h[i].t;
            if (out_prim) out_prim[i] = h[i].prim;
            if (out_flags) out_flags[i] = h[i].flags;
        }
    }
    return RT1W_OK;
}

} // extern "C"
