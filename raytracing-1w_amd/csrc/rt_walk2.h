/* rt_walk2.h -- the PHASED WALK: closest hit of the reference's BVH walk (src/bvh.rs:25-50, src/aabb.rs:13-32 and every
 * leaf's `hit`) with the box work and the leaf work of a lane separated, so that a wave runs ONE kind of work at a time.
 * Device code (gfx950); the CPU test build keeps the one-entry-per-step walk of rt_core.h, and the GPU tests require the
 * two to agree bit for bit.
 *
 * Why.  PMC of the one-entry-per-step walk on final_scene's rays (profiles/r03_lab_*): VALU busy in 67 % of the cycles with
 * 21 % of the lanes active -- every wave step executes the box code, the rect code, the sphere code, ... one after the other,
 * each for the few lanes that happen to be at that kind.  Here a wave votes for box work or leaf work and all its lanes
 * that have such work do it together.
 *
 * The walk tree (built on the host from the flattened scene, rt_walk2_build):
 *   INNER   a BVHChild::Two whose children are both BVH nodes.  One 64-byte record holds BOTH children's boxes, stored in
 *           f32 rounded OUTWARD, and the children's references.  Inner boxes only STEER the walk: they may pass where the
 *           reference's test fails (never the reverse -- the arithmetic on them stays f64 and monotone), because ...
 *   PGROUP  ... a BVH node whose children are primitives (BVHChild::One(prim) / Two(prim, prim): what BVHNode::new makes of
 *           every primitive, bvh.rs:63-79) is GATED EXACTLY: its own box (the reference's f64 record) is tested with the
 *           closest hit the reference has when it reaches that node, then its primitives in the reference's order.
 *   OTHER   everything else (wrappers, media, BVH nodes with such children, wrapper exits) is handled by the one-entry-
 *           per-step code of rt_core.h, at the moment the reference handles it.
 *
 * Exactness.  A lane does its box work AHEAD of its pending leaf work: the closest hit its box tests see may be stale
 * (larger than the reference's at that point).  The slab verdict min(exit, t_max) > enter is monotone in t_max, so a stale
 * or outward-rounded box test can only pass more.  Pending PGROUPs are handled strictly in the reference's order and nothing
 * else changes the closest hit in between (OTHER entries wait until nothing is pending), so every gate sees the
 * reference's closest hit.  The boxes above a PGROUP need no second look: a child's box lies inside its parent's
 * (surrounding_box, aabb.rs:35-52), so a gate that passes at t implies every box above passes at t, hence at the larger
 * t_max the reference tested it with; a gate that fails keeps the primitives untested exactly as the reference does,
 * whichever box failed there.  Hence the same primitives are tested, in the same order, with the same operands: same bits.
 * Monotonicity needs a closest hit that is not NaN (a NaN root is accepted by sphere.rs:43-48 / aarect.rs:48-56): the lane
 * then restarts the segment with the one-entry-per-step walk (the generator is restored: a ConstantMedium draws inside the
 * walk, constant_medium.rs:85).
 */
#ifndef RT1W_WALK2_H
#define RT1W_WALK2_H

#include "rt_core.h"

#define RT_W2_INNER 0x40000000u  /* reference to an inner pair record (low bits: index into RtW2View::inner) */
#define RT_W2_PGROUP 0x20000000u /* reference to a primitive group (low bits: index of the BVH node in the flat node array) */
#define RT_W2_INDEX 0x1FFFFFFFu
#define RT_W2_NONE 0xFFFFFFFFu   /* no entry (never a valid reference: RT_POP_FLAG | INNER | PGROUP all set) */
#define RT_W2_QCAP 4             /* pending primitive groups per lane (a power of two; 8 costs the lab kernel a workgroup per CU) */
#define RT_W2_STACK 24           /* stack entries per lane the phased kernels are built with (deeper scenes keep the classic walk) */

struct RtW2Inner {
    float lb[6], rb[6]; /* boxes of the left and the right child (min.xyz, max.xyz), rounded outward */
    uint32_t l, r;      /* tagged references of the children */
    uint32_t pad[2];
}; /* 64 bytes */

struct RtW2View {
    const RtW2Inner* inner;
    const uint32_t* wref; /* tagged reference of every flat node (BVH nodes: INNER | PGROUP | plain index; others: plain index) */
    uint32_t root;        /* wref[scene root] */
    uint32_t n_inner;
};

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

/* per-lane storage in LDS: [entry][lane] like LdsStack.  `ent` = entry distance of the pushed box rounded DOWN to f32
 * (culling at the pop may miss, never over-cull); -inf for entries that must not be culled. */
template <int BLOCK>
struct RtW2Stack {
    uint32_t* ref;
    float* ent;
    uint32_t* q;
    int sp;
    uint32_t qh, qn;
    /* the LdsStack interface, for the code of rt_core.h that runs on the same stack (wrappers, media boundary walks) */
    __device__ __forceinline__ void push(uint32_t v) { ref[sp * BLOCK] = v; ent[sp * BLOCK] = -__builtin_inff(); ++sp; }
    __device__ __forceinline__ void poke(int above, uint32_t v) { ref[(sp + above) * BLOCK] = v; ent[(sp + above) * BLOCK] = -__builtin_inff(); }
    __device__ __forceinline__ uint32_t pop() { --sp; return ref[sp * BLOCK]; }
    __device__ __forceinline__ void push_ent(uint32_t v, float e) { ref[sp * BLOCK] = v; ent[sp * BLOCK] = e; ++sp; }
    __device__ __forceinline__ void enqueue(uint32_t node) { q[((qh + qn) & (RT_W2_QCAP - 1u)) * BLOCK] = node; qn += 1u; }
    __device__ __forceinline__ uint32_t dequeue() { const uint32_t v = q[(qh & (RT_W2_QCAP - 1u)) * BLOCK]; qh += 1u; qn -= 1u; return v; }
};

/* aabb.rs:14-29 with the interval kept by max/min (rt_aabb_hit_fast's arithmetic) on a box given in f32, t_max not folded in */
__device__ __forceinline__ void rt_w2_slab(const float* bb, RtV3 o, RtV3 inv, double t_min, double& enter, double& exit_) {
    double lo = t_min, hi = RT_INF;
#define RT_W2_AX(minv, maxv, ov, iv)                     \
    {                                                    \
        double t0 = ((double)(minv) - (ov)) * (iv);      \
        double t1 = ((double)(maxv) - (ov)) * (iv);      \
        if ((iv) < 0.0) { double s_ = t0; t0 = t1; t1 = s_; } \
        lo = rt_vmax(t0, lo);                            \
        hi = rt_vmin(t1, hi);                            \
    }
    RT_W2_AX(bb[0], bb[3], o.x, inv.x)
    RT_W2_AX(bb[1], bb[4], o.y, inv.y)
    RT_W2_AX(bb[2], bb[5], o.z, inv.z)
#undef RT_W2_AX
    enter = lo; exit_ = hi;
}

struct RtW2Lane {
    uint32_t cur;  /* the next entry this lane handles (tagged), RT_W2_NONE: take one from the stack */
    bool nan_seen; /* the closest hit turned NaN: the segment has to be redone by the classic walk */
};

template <class Stack>
__device__ __forceinline__ void rt_w2_begin(const RtW2View& w2, RtW2Lane& L, RtWalk& k, const RtRay& world, double t_min, double t_max, Stack& stk) {
    k.w.o = world.o; k.w.d = world.d;
    k.cur = k.w;
    k.inv_w = rt_inv3(k.w.d);
    k.inv = k.inv_w;
    k.time = world.time; k.t_min = t_min; k.best_t = t_max;
    k.scope = RT_NONE; k.best_prim = RT_NONE; k.best_scope = RT_NONE;
    k.tmin_nan = rt_isnan(t_min);
    k.base = 0;
    stk.sp = 0; stk.qh = 0u; stk.qn = 0u;
    L.cur = w2.root;
    L.nan_seen = false;
}
template <class Stack>
__device__ __forceinline__ bool rt_w2_done(const RtW2Lane& L, const Stack& stk) { return L.cur == RT_W2_NONE && stk.sp == 0 && stk.qn == 0u; }
/* what kind of work the lane's next entry is */
__device__ __forceinline__ bool rt_w2_is_boxwork(uint32_t cur) { return cur != RT_W2_NONE && (cur & RT_POP_FLAG) == 0u && (cur & (RT_W2_INNER | RT_W2_PGROUP)) != 0u; }
__device__ __forceinline__ bool rt_w2_is_other(uint32_t cur) { return cur != RT_W2_NONE && ((cur & RT_POP_FLAG) != 0u || (cur & (RT_W2_INNER | RT_W2_PGROUP)) == 0u); }

/* BOX WORK of one lane: an inner pair record, or a primitive group to append to the pending queue */
template <class Stack, int BLOCK>
__device__ __forceinline__ void rt_w2_box_step(const RtW2View& w2, RtW2Lane& L, RtWalk& k, Stack& stk) {
    const uint32_t cur = L.cur;
    if (cur & RT_W2_PGROUP) {
        stk.enqueue(cur & RT_W2_INDEX);
        L.cur = RT_W2_NONE;
        return;
    }
    const RtW2Inner P = w2.inner[cur & RT_W2_INDEX];
    double el, xl, er, xr;
    rt_w2_slab(P.lb, k.cur.o, k.inv, k.t_min, el, xl);
    rt_w2_slab(P.rb, k.cur.o, k.inv, k.t_min, er, xr);
    const bool pl = rt_vmin(xl, k.best_t) > el, pr = rt_vmin(xr, k.best_t) > er;
    uint32_t next = RT_W2_NONE;
    bool left_entered = false;
    if (pl) {
        if ((P.l & (RT_W2_PGROUP | RT_POP_FLAG)) == RT_W2_PGROUP) stk.enqueue(P.l & RT_W2_INDEX); /* a group is not entered: it joins the pending ones */
        else { next = P.l; left_entered = true; }
    }
    if (pr) {
        if (left_entered) stk.push_ent(P.r, __double2float_rd(er)); /* after the left subtree: bvh.rs:38-47 */
        else if ((P.r & (RT_W2_PGROUP | RT_POP_FLAG)) == RT_W2_PGROUP) stk.enqueue(P.r & RT_W2_INDEX);
        else next = P.r;
    }
    L.cur = next;
}

/* LEAF WORK: the oldest pending primitive group -- the reference's BVHNode::hit on that node: its box with the closest hit as
 * it is NOW, then its one or two primitives, left first */
template <class Cfg, class Stack, class NS>
__device__ __forceinline__ void rt_w2_group_step(const RtSceneView& sc, const NS& ns, RtW2Lane& L, RtWalk& k, Stack& stk) {
    const uint32_t e = stk.dequeue();
    const RtNodeHot g = ns.hot(e);
    if (rt_aabb_hit_fast<false>(g.d, k.cur.o, k.inv, k.t_min, k.best_t)) {
        const RtNodeHot a = ns.hot(e + 1u);
        rt_walk_leaf<Cfg>(sc, k, e + 1u, a);
        if ((g.kind & RT_KIND_MASK) == RT_BVH2) {
            const RtNodeHot b = ns.hot(g.b);
            rt_walk_leaf<Cfg>(sc, k, g.b, b);
        }
        if (rt_isnan(k.best_t)) L.nan_seen = true;
    }
}

/* OTHER WORK (nothing pending): the entry is handled as the one-entry-per-step walk handles it; the children of a BVH node
 * come back as tagged references so that the subtrees below continue in phases */
template <class Cfg, class Stack, class NS>
__device__ __forceinline__ void rt_w2_other_step(const RtSceneView& sc, const NS& ns, const RtW2View& w2, RtW2Lane& L, RtWalk& k, RtRng& rng, Stack& stk) {
    const uint32_t e = L.cur;
    L.cur = RT_W2_NONE;
    if (e & RT_POP_FLAG) { if (Cfg::scope_depth > 0) rt_walk_exit(sc, k, e); return; }
    const RtNodeHot nd = ns.hot(e);
    const uint32_t km = nd.kind & RT_KIND_MASK;
    if (km <= RT_BVH1) {
        bool hit;
        if (k.tmin_nan || rt_isnan(k.best_t)) hit = rt_aabb_hit(nd.d, k.cur.o, k.inv, k.t_min, k.best_t);
        else hit = rt_aabb_hit_fast<false>(nd.d, k.cur.o, k.inv, k.t_min, k.best_t);
        if (hit) {
            if (km == RT_BVH2) stk.push(w2.wref[nd.b]); /* the right child below the left one: bvh.rs:38-47 */
            L.cur = w2.wref[e + 1u];
        }
    } else if (km <= RT_YZ) {
        rt_walk_leaf<Cfg>(sc, k, e, nd);
        if (rt_isnan(k.best_t)) L.nan_seen = true;
    } else if (Cfg::scope_depth > 0 && km <= RT_FLIP) {
        stk.push(e | RT_POP_FLAG);
        k.scope = e;
        if (km != RT_FLIP) {
            k.cur = rt_scope_in(nd, k.cur);
            if (km == RT_ROTATE_Y) k.inv = rt_inv3(k.cur.d);
        }
        L.cur = w2.wref[e + 1u];
    } else if (Cfg::media) {
        rt_walk_other<Cfg, true>(sc, ns, k, e, nd, rng, stk);
        if (rt_isnan(k.best_t)) L.nan_seen = true;
    }
}

/* take the nearest pushed entry if the lane has none; an entry whose box starts behind the closest hit is dropped */
template <class Stack, int BLOCK>
__device__ __forceinline__ void rt_w2_refetch(RtW2Lane& L, const RtWalk& k, Stack& stk) {
    if (L.cur == RT_W2_NONE && stk.sp > 0) {
        --stk.sp;
        const uint32_t ref = stk.ref[stk.sp * BLOCK];
        const float ent = stk.ent[stk.sp * BLOCK];
        if ((double)ent < k.best_t || rt_isnan(k.best_t)) L.cur = ref;
    }
}

#endif /* device */

#if !defined(RT_W2_DEVICE_ONLY) /* host: the builder (plain C++, also seen by hipcc's device pass of a translation unit that calls it) */
#include <cmath>
#include <vector>
/* flat pre-order nodes -> inner pair records + the tagged reference of every node.  Works for every scene (what is not an
 * INNER or a PGROUP stays a plain index and is handled by the classic code). */
inline void rt_walk2_build(const std::vector<RtNode>& N, uint32_t root, std::vector<RtW2Inner>& inner, std::vector<uint32_t>& wref, uint32_t& root_ref) {
    auto kind = [&](uint32_t i) { return N[i].kind & RT_KIND_MASK; };
    auto is_bvh = [&](uint32_t i) { return kind(i) <= RT_BVH1; };
    auto is_prim = [&](uint32_t i) { return kind(i) >= RT_SPHERE && kind(i) <= RT_YZ; };
    inner.clear();
    wref.assign(N.size(), 0u);
    std::vector<uint32_t> inner_of(N.size(), RT_W2_NONE);
    for (uint32_t i = 0; i < N.size(); ++i) { /* pre-order: a node's inner index precedes its descendants' */
        wref[i] = i;
        if (!is_bvh(i)) continue;
        const bool two = kind(i) == RT_BVH2;
        const uint32_t a = N[i].a, b = two ? N[i].b : a;
        /* the exactness argument needs every box below an inner node to lie INSIDE that node's box.  surrounding_box gives that,
         * but an AABox does not: its bounding box (aabox.rs:98-103) is the bare box while the rects of its side BVH are padded by
         * 0.0001 (aarect.rs:74-79), so the side BVH sticks out of the BVH nodes above the AABox.  A node whose child's box is not
         * inside its own is therefore tested exactly, at the reference's moment (OTHER), like the reference does. */
        auto inside = [&](uint32_t c) {
            for (int q = 0; q < 3; ++q) if (!(N[c].d[q] >= N[i].d[q]) || !(N[c].d[q + 3] <= N[i].d[q + 3])) return false;
            return true;
        };
        if (two && is_bvh(a) && is_bvh(b) && inside(a) && inside(b)) {
            inner_of[i] = (uint32_t)inner.size();
            inner.push_back(RtW2Inner());
            wref[i] = RT_W2_INNER | inner_of[i];
        } else if (is_prim(a) && is_prim(b)) {
            wref[i] = RT_W2_PGROUP | i;
        }
        /* (a BVH node that is neither -- mixed children, one child, a child sticking out -- keeps its plain index) */
    }
    auto outward = [](double x, bool is_min) {
        float f = (float)x;
        if (is_min ? ((double)f > x) : ((double)f < x)) f = std::nextafterf(f, is_min ? -INFINITY : INFINITY);
        return f;
    };
    for (uint32_t i = 0; i < N.size(); ++i) {
        if (inner_of[i] == RT_W2_NONE) continue;
        RtW2Inner& P = inner[inner_of[i]];
        const uint32_t a = N[i].a, b = N[i].b;
        for (int q = 0; q < 6; ++q) { P.lb[q] = outward(N[a].d[q], q < 3); P.rb[q] = outward(N[b].d[q], q < 3); }
        P.l = wref[a]; P.r = wref[b];
        P.pad[0] = P.pad[1] = 0u;
    }
    root_ref = wref[root];
}
#endif
#endif
