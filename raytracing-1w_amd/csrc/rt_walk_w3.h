/* rt_walk_w3.h -- lab candidate W3: the one-entry-per-step stack walk with PAIR RECORDS for the BVH nodes that only steer.  Measured on
 * hardware in the trace-only harness and NOT in the product (final_scene 0.90x of the walk it extends, random_scene 1.00x: docs/LAB_NOTES.md):
 * built into librt1w_lab.so (walk_lab.hip) and into tools/pairs_check.cpp only. */
#ifndef RT1W_WALK_W3_H
#define RT1W_WALK_W3_H
#include "rt_core.h"

/* ---- the stack walk with PAIR RECORDS for the BVH nodes that only steer ------------------------------------------------------
 * A BVH node all of whose children are BVH nodes lying inside its own box decides nothing about the result: its box test only
 * prunes, and every primitive, wrapper and medium below it is still gated by the exact test of the BVH node directly above it
 * (aabb.rs:13-32 is monotone: a box inside another one is missed whenever the outer one is, whatever closest hit either test
 * saw).  Such a node may therefore be tested conservatively and EARLY: its record here holds BOTH children's boxes in f32 rounded
 * outward (64 bytes for two box tests), a child that passes is pushed as a tagged entry (RT_PAIR_FLAG | record) if it is such a
 * node itself -- it is never fetched from the node array, its box is not tested again -- or as its plain node index if it is a
 * gate, whose own f64 box is then tested exactly when it is popped, at the closest hit the reference has there.  A child whose
 * box fails is never popped or fetched at all: about half of the one-entry-per-step walk's node visits are such misses.
 * The records are per context (rt_pairs_build below); the node array the walk reads carries, in the otherwise unused `mat` word
 * of such a BVH node, 1 + the index of its pair record, for the places where the node is reached by its plain index (the root,
 * a wrapper's child, a gate's child): its own box is tested exactly there and the tagged entry is pushed. */
#define RT_PAIR_FLAG 0x40000000u
struct RtPairRec {
    float lb[6], rb[6]; /* boxes of the left and the right child (min.xyz, max.xyz), rounded outward */
    uint32_t l, r;      /* the entries to push: RT_PAIR_FLAG | record, or a node index; r == RT_NONE: BVHChild::One */
    uint32_t pad[2];
}; /* 64 bytes */
RT_HD void rt_pair_slab(const float* bb, RtV3 o, RtV3 inv, double t_min, double& enter, double& exit_) {
    double lo = t_min, hi = RT_INF; /* rt_aabb_hit_fast's arithmetic on the widened box, t_max not folded in */
#define RT_PAIR_AX(minv, maxv, ov, iv)                   \
    {                                                    \
        double t0 = ((double)(minv) - (ov)) * (iv);      \
        double t1 = ((double)(maxv) - (ov)) * (iv);      \
        if ((iv) < RT_R(0.0)) { double s_ = t0; t0 = t1; t1 = s_; } \
        lo = rt_vmax(t0, lo);                            \
        hi = rt_vmin(t1, hi);                            \
    }
    RT_PAIR_AX(bb[0], bb[3], o.x, inv.x)
    RT_PAIR_AX(bb[1], bb[4], o.y, inv.y)
    RT_PAIR_AX(bb[2], bb[5], o.z, inv.z)
#undef RT_PAIR_AX
    enter = lo; exit_ = hi;
}
/* a popped pair entry: both children's boxes, the survivors pushed (right below left, bvh.rs:38-47) */
template <class Stack>
RT_HD void rt_walkp_pair(const RtPairRec& P, RtWalk& k, Stack& stk) {
    double el, xl, er, xr;
    rt_pair_slab(P.lb, k.cur.o, k.inv, k.t_min, el, xl);
    rt_pair_slab(P.rb, k.cur.o, k.inv, k.t_min, er, xr);
    const bool pass = k.tmin_nan || rt_isnan(k.best_t); /* the reference's test never fails on a NaN bound (aabb.rs:27: `t_max <= t_min` is false) */
    const bool pl = pass || rt_vmin(xl, k.best_t) > el;
    const bool pr = P.r != RT_NONE && (pass || rt_vmin(xr, k.best_t) > er);
    stk.poke(0, (pl && !pr) ? P.l : P.r);
    stk.poke(1, P.l);
    stk.sp += (pl ? 1 : 0) + (pr ? 1 : 0);
}
/* a BVH node reached by its plain index: its own box, exactly; then its pair entry if it has one, else its children */
template <class Cfg, class Stack>
RT_HD void rt_walkp_box(RtWalk& k, uint32_t e, const RtNodeHot& nd, Stack& stk) {
    bool hit;
    if (RT_WAVE_ANY(k.tmin_nan || rt_isnan(k.best_t))) hit = rt_aabb_hit(nd.d, k.cur.o, k.inv, k.t_min, k.best_t);
    else hit = rt_aabb_hit_fast<false>(nd.d, k.cur.o, k.inv, k.t_min, k.best_t);
    const bool two = (nd.kind & RT_KIND_MASK) == RT_BVH2;
    const uint32_t pi = nd.mat;
    stk.poke(0, pi != 0u ? (RT_PAIR_FLAG | (pi - 1u)) : (two ? nd.b : e + 1u));
    stk.poke(1, e + 1u);
    stk.sp += hit ? ((two && pi == 0u) ? 2 : 1) : 0;
}
template <class Cfg, bool MEDIA, class Stack, class NS>
RT_HD void rt_walkp_step(const RtSceneView& sc, const NS& ns, const RtPairRec* pairs, RtWalk& k, RtRng& rng, Stack& stk) {
    const uint32_t e = stk.pop();
    if (Cfg::scope_depth > 0 && (e & RT_POP_FLAG)) { rt_walk_exit(sc, k, e); return; }
    if (e & RT_PAIR_FLAG) { RT_STAT_VISIT(15u); rt_walkp_pair(pairs[e & ~RT_PAIR_FLAG], k, stk); return; }
    const RtNodeHot nd = ns.hot(e);
    const uint32_t km = nd.kind & RT_KIND_MASK;
    RT_STAT_VISIT(km);
    if (km <= RT_BVH1) rt_walkp_box<Cfg>(k, e, nd, stk);
    else if (km <= RT_YZ) rt_walk_leaf<Cfg>(sc, k, e, nd);
    else if (Cfg::scope_depth > 0 && km <= RT_FLIP) rt_walk_wrap(k, e, nd, stk);
    else if (Cfg::media) rt_walk_other<Cfg, MEDIA>(sc, ns, k, e, nd, rng, stk);
}
/* the box-only step (rt_walk_box_step) of this walk: pair entries and BVH nodes */
template <class Cfg, class Stack, class NS>
RT_HD bool rt_walkp_box_step(const NS& ns, const RtPairRec* pairs, RtWalk& k, Stack& stk) {
    const uint32_t e = stk.pop();
    bool taken = false;
    if (!(Cfg::scope_depth > 0 && (e & RT_POP_FLAG))) {
        if (e & RT_PAIR_FLAG) { RT_STAT_VISIT(15u); rt_walkp_pair(pairs[e & ~RT_PAIR_FLAG], k, stk); taken = true; }
        else {
            const RtNodeHot nd = ns.hot(e);
            if ((nd.kind & RT_KIND_MASK) <= RT_BVH1) { RT_STAT_VISIT(nd.kind & RT_KIND_MASK); rt_walkp_box<Cfg>(k, e, nd, stk); taken = true; }
        }
    }
    if (!taken) stk.sp += 1;
    return taken;
}

#endif
