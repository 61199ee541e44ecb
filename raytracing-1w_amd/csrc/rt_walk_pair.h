/* rt_walk_pair.h -- the PAIR WALK of sphere scenes: closest hit of the reference's BVH walk (src/bvh.rs:25-50, src/aabb.rs:13-32,
 * src/sphere.rs:22-63, src/moving_sphere.rs:29-70) for scenes that are nothing but a BVH over Sphere / MovingSphere (random_scene,
 * BASELINE config C2), in two phases per wave.  Candidate "W1c" of the trace-only harness (walk_lab.hip), where it answered
 * random_scene's rays 1.2-1.4x faster than the one-entry-per-step walk with the same bits (profiles/r03_lab_*); built into the
 * render kernel rt_render_kernel_pw (rt_kernel_plain.h).
 *
 * What it changes against the one-entry-per-step walk of rt_core.h:
 *   - the L1 is the wall on this scene (0.8 lane-loads per clock per CU of ~1.0 the cache delivers for divergent 16-byte loads,
 *     profiles/r03_l1_gather_probe.txt): an INNER node (BVHChild::Two over two BVH nodes) is one 64-byte record holding BOTH
 *     children's boxes in f32 rounded OUTWARD -- four loads for two box tests instead of eight;
 *   - a wave runs ONE kind of work at a time: box work (every stepping lane visits an inner record) or leaf work (every lane with
 *     a pending GROUP -- a BVHChild::One(sphere) or Two(sphere, sphere) -- tests that group's own f64 box and its spheres).
 * Exactness: see rt_walk2.h (the same argument; this is its sphere-only, compact-record form).  Inner boxes only steer; every
 * sphere is gated by its group's own box, in f64, at the closest hit the reference has when it reaches that group, because the
 * pending groups of a lane are handled strictly in the reference's order.  A closest hit that turns NaN hands the segment to the
 * one-entry-per-step walk (the walk draws no random numbers on such a scene: nothing to restore).
 */
#ifndef RT1W_WALK_PAIR_H
#define RT1W_WALK_PAIR_H

#include "rt_core.h"
#if !defined(__HIPCC__)
#include <cmath>
#endif

#define RT_PW_LEAF 0x80000000u
#define RT_PW_NONE 0xFFFFFFFFu
#ifndef RT_PW_QCAP
#define RT_PW_QCAP 8   /* pending groups per lane (a power of two) */
#endif
#define RT_PW_STACK 16 /* pushed right children per lane (scenes that need more keep the one-entry-per-step walk) */

struct RtPwInner { /* boxes of the left and right child (min.xyz, max.xyz) in f32 rounded outward; child references: RT_PW_LEAF | group, or inner index */
    float lb[6], rb[6];
    uint32_t l, r;
    uint32_t pad[2];
}; /* 64 bytes */
struct RtPwPrim {
    double c0[3], dc[3]; /* Sphere: centre, 0; MovingSphere: center0, center1 - center0 (moving_sphere.rs:23-26) */
    double radius;
    uint32_t id;         /* index of the primitive's RtNode: what the walk reports */
    uint32_t moving;
}; /* 64 bytes (36 in the f32 build, where `double` is `float`: context_f32.hip builds its own records from its own node array) */
struct RtPwGroup {
    double box[6];       /* the group's own box: BVHNode.aabb of the One / Two node, the reference's bits */
    uint32_t n, pad[3];
    RtPwPrim p[2];
}; /* 192 bytes */
struct RtPwView {
    const RtPwInner* inner;
    const RtPwGroup* groups;
    double root_box[6];
    uint32_t root;       /* inner index, or RT_PW_LEAF | group */
    uint32_t pad;
    double ms_time0, ms_time1; /* the scene's one shutter interval (main.rs:230-237: every moving sphere has (0, 1)) */
};

/* The lane functions below are plain C++ apart from the directed rounding: the CPU test build (oracle/oracle_flat.cpp) compiles the
 * same text with bound-checked stacks and queues (RtPwLds::ref / ent / q may be any type with operator[]) and a randomised schedule. */
#if defined(__HIPCC__)
#define RT_PW_FN __device__ __forceinline__
#define RT_PW_ROUND_DOWN(x) __double2float_rd(x)
#else
#define RT_PW_FN inline
static inline float rt_pw_round_down_host(double x) { float f = (float)x; return ((double)f > x) ? std::nextafterf(f, -INFINITY) : f; }
#define RT_PW_ROUND_DOWN(x) rt_pw_round_down_host(x)
#endif
/* per-lane walk state that lives across the slices of a walk: the next inner record, the stack level, the queue window (the
 * stack and the queue themselves are in LDS) */
struct RtPwLane {
    uint32_t cur, qh, qn;
    int sp;
};
template <int BLOCK, class RefT = uint32_t*, class EntT = float*, class QT = uint32_t*>
struct RtPwLds {
    RefT ref; /* [RT_PW_STACK][BLOCK] */
    EntT ent; /* entry distance of the pushed box, rounded DOWN (culling at the pop may miss, never over-cull) */
    QT q;     /* [RT_PW_QCAP][BLOCK] */
};
RT_PW_FN void rt_pw_slab(const float* bb, RtV3 o, RtV3 inv, double t_min, double& enter, double& exit_) {
    /* aabb.rs:14-29 with the interval kept by max/min (rt_aabb_hit_fast's arithmetic), t_max not folded in */
    double lo = t_min, hi = RT_INF;
#define RT_PW_AX(minv, maxv, ov, iv)                     \
    {                                                    \
        double t0 = ((double)(minv) - (ov)) * (iv);      \
        double t1 = ((double)(maxv) - (ov)) * (iv);      \
        if ((iv) < RT_R(0.0)) { double s_ = t0; t0 = t1; t1 = s_; } \
        lo = rt_vmax(t0, lo);                            \
        hi = rt_vmin(t1, hi);                            \
    }
    RT_PW_AX(bb[0], bb[3], o.x, inv.x)
    RT_PW_AX(bb[1], bb[4], o.y, inv.y)
    RT_PW_AX(bb[2], bb[5], o.z, inv.z)
#undef RT_PW_AX
    enter = lo; exit_ = hi;
}
/* the root's own box, exactly as the reference tests it first (bvh.rs:32); false: the ray misses the scene */
template <int BLOCK, class RefT, class EntT, class QT>
RT_PW_FN bool rt_pw_begin(const RtPwView& pw, RtPwLane& L, RtPwLds<BLOCK, RefT, EntT, QT>& m, RtV3 o, RtV3 inv, double t_min) {
    L.sp = 0; L.qh = 0u; L.qn = 0u; L.cur = RT_PW_NONE;
    if (!rt_aabb_hit_fast<false>(pw.root_box, o, inv, t_min, RT_INF)) return false;
    if (pw.root & RT_PW_LEAF) { m.q[0] = pw.root & ~RT_PW_LEAF; L.qn = 1u; }
    else L.cur = pw.root;
    return true;
}
RT_PW_FN bool rt_pw_done(const RtPwLane& L) { return L.cur == RT_PW_NONE && L.sp == 0 && L.qn == 0u; }
RT_PW_FN bool rt_pw_can_box(const RtPwLane& L) { return (L.cur != RT_PW_NONE || L.sp > 0) && L.qn + 2u <= (uint32_t)RT_PW_QCAP; }

/* BOX WORK of one lane: back to the nearest pushed right child if there is no current record, then one inner record */
template <int BLOCK, class RefT, class EntT, class QT>
RT_PW_FN void rt_pw_box_step(const RtPwView& pw, RtPwLane& L, RtPwLds<BLOCK, RefT, EntT, QT>& m, RtV3 o, RtV3 inv, double t_min, double best_t) {
    if (L.cur == RT_PW_NONE) {
        --L.sp;
        const uint32_t ref = m.ref[L.sp * BLOCK];
        const float ent = m.ent[L.sp * BLOCK];
        if ((double)ent < best_t) { /* conservative: ent <= the box's entry distance */
            if (ref & RT_PW_LEAF) { m.q[((L.qh + L.qn) & (RT_PW_QCAP - 1u)) * BLOCK] = ref & ~RT_PW_LEAF; L.qn += 1u; }
            else L.cur = ref;
        }
    }
    if (L.cur != RT_PW_NONE) {
        const RtPwInner P = pw.inner[L.cur];
        double el, xl, er, xr;
        rt_pw_slab(P.lb, o, inv, t_min, el, xl);
        rt_pw_slab(P.rb, o, inv, t_min, er, xr);
        const bool pl = rt_vmin(xl, best_t) > el, pr = rt_vmin(xr, best_t) > er;
        uint32_t next = RT_PW_NONE;
        bool left_entered = false;
        if (pl) {
            if (P.l & RT_PW_LEAF) { m.q[((L.qh + L.qn) & (RT_PW_QCAP - 1u)) * BLOCK] = P.l & ~RT_PW_LEAF; L.qn += 1u; }
            else { next = P.l; left_entered = true; }
        }
        if (pr) {
            if (left_entered) { /* after the left subtree: bvh.rs:38-47 */
                m.ref[L.sp * BLOCK] = P.r;
                m.ent[L.sp * BLOCK] = RT_PW_ROUND_DOWN(er);
                ++L.sp;
            } else if (P.r & RT_PW_LEAF) { m.q[((L.qh + L.qn) & (RT_PW_QCAP - 1u)) * BLOCK] = P.r & ~RT_PW_LEAF; L.qn += 1u; }
            else next = P.r;
        }
        L.cur = next;
    }
}
/* LEAF WORK of one lane: its oldest pending group -- the reference's BVHNode::hit on that node: its box with the closest hit as
 * it is NOW, then its one or two spheres, left first.  `frac` = (time - time0) / (time1 - time0) of MovingSphere::center */
template <int BLOCK, class RefT, class EntT, class QT>
RT_PW_FN void rt_pw_group_step(const RtPwView& pw, RtPwLane& L, RtPwLds<BLOCK, RefT, EntT, QT>& m, RtV3 o, RtV3 d, RtV3 inv, double frac, double t_min,
                                                 double& best_t, uint32_t& best_prim) {
    const uint32_t g = m.q[(L.qh & (RT_PW_QCAP - 1u)) * BLOCK];
    L.qh += 1u; L.qn -= 1u;
    const RtPwGroup& G = pw.groups[g];
    if (rt_aabb_hit_fast<false>(G.box, o, inv, t_min, best_t)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < (int)G.n) {
                const RtPwPrim& P = G.p[i];
                RtV3 c = rt_v3(P.c0[0], P.c0[1], P.c0[2]);
                if (P.moving) c = c + frac * rt_v3(P.dc[0], P.dc[1], P.dc[2]); /* moving_sphere.rs:23-26 */
                double t;
                if (rt_sphere_root(c, P.radius, o, d, t_min, best_t, t)) { best_t = t; best_prim = P.id; }
            }
        }
    }
}

#if !defined(RT_PW_DEVICE_ONLY)
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
/* flat pre-order nodes -> inner pair records + groups; false (with the reason) if the scene is outside the pair walk's scope */
struct RtPwBuilder {
    const std::vector<RtNode>& N;
    std::vector<RtPwInner>& inner;
    std::vector<RtPwGroup>& groups;
    bool ok = true;
    std::string why;
    uint32_t kind(uint32_t i) const { return N[i].kind & RT_KIND_MASK; }
    bool is_prim(uint32_t i) const { return kind(i) == RT_SPHERE || kind(i) == RT_MSPHERE; }
    bool is_group(uint32_t i) const {
        if (kind(i) == RT_BVH1) return is_prim(N[i].a);
        if (kind(i) == RT_BVH2) return is_prim(N[i].a) && is_prim(N[i].b);
        return false;
    }
    RtPwPrim prim_of(uint32_t i) const {
        RtPwPrim p; std::memset(&p, 0, sizeof p);
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
    static float outward(double x, bool is_min) {
        float f = (float)x;
        if (is_min ? ((double)f > x) : ((double)f < x)) f = std::nextafterf(f, is_min ? -INFINITY : INFINITY);
        return f;
    }
    uint32_t go(uint32_t i) {
        if (!ok) return RT_PW_NONE;
        if (is_group(i)) {
            RtPwGroup g; std::memset(&g, 0, sizeof g);
            for (int a = 0; a < 6; ++a) g.box[a] = N[i].d[a];
            g.n = kind(i) == RT_BVH2 ? 2u : 1u;
            g.p[0] = prim_of(N[i].a);
            if (g.n == 2u) g.p[1] = prim_of(N[i].b);
            groups.push_back(g);
            return RT_PW_LEAF | (uint32_t)(groups.size() - 1);
        }
        if (kind(i) == RT_BVH1) { ok = false; why = "BVHChild::One over a BVH node"; return RT_PW_NONE; }
        if (kind(i) != RT_BVH2 || is_prim(N[i].a) || is_prim(N[i].b)) { ok = false; why = "a BVHChild::Two with one primitive and one subtree"; return RT_PW_NONE; }
        const uint32_t a = N[i].a, b = N[i].b;
        for (int q = 0; q < 3; ++q) /* the exactness argument needs the children's boxes inside the node's own (surrounding_box gives that) */
            if (!(N[a].d[q] >= N[i].d[q]) || !(N[a].d[q + 3] <= N[i].d[q + 3]) || !(N[b].d[q] >= N[i].d[q]) || !(N[b].d[q + 3] <= N[i].d[q + 3])) {
                ok = false; why = "a child's box is not inside its parent's"; return RT_PW_NONE;
            }
        const uint32_t me = (uint32_t)inner.size();
        inner.push_back(RtPwInner());
        std::memset(&inner[me], 0, sizeof(RtPwInner));
        for (int q = 0; q < 6; ++q) { inner[me].lb[q] = outward(N[a].d[q], q < 3); inner[me].rb[q] = outward(N[b].d[q], q < 3); }
        const uint32_t l = go(a);
        const uint32_t r = go(b);
        inner[me].l = l; inner[me].r = r;
        return me;
    }
};
inline bool rt_pw_build(const std::vector<RtNode>& N, uint32_t root, std::vector<RtPwInner>& inner, std::vector<RtPwGroup>& groups, RtPwView& view,
                        std::string& why) {
    inner.clear(); groups.clear();
    RtPwBuilder b{N, inner, groups};
    bool seen_ms = false;
    double t0 = 0.0, t1 = 1.0;
    for (uint32_t i = 0; i < N.size(); ++i) {
        const uint32_t k = b.kind(i);
        if (k == RT_BVH2 || k == RT_BVH1) continue;
        if (!b.is_prim(i)) { why = "a node that is neither a BVH node nor a sphere"; return false; }
        if (k == RT_MSPHERE) {
            if (!seen_ms) { t0 = N[i].e[0]; t1 = N[i].e[1]; seen_ms = true; }
            else if (std::memcmp(&t0, &N[i].e[0], sizeof t0) != 0 || std::memcmp(&t1, &N[i].e[1], sizeof t1) != 0) { why = "moving spheres with different shutter intervals"; return false; }
        }
    }
    view.ms_time0 = t0; view.ms_time1 = t1;
    const uint32_t k0 = b.kind(root);
    if (k0 != RT_BVH2 && k0 != RT_BVH1) { why = "the root is not a BVH node"; return false; }
    for (int a = 0; a < 6; ++a) view.root_box[a] = N[root].d[a];
    view.root = b.go(root);
    view.pad = 0u;
    why = b.why;
    return b.ok;
}
#endif
#endif
