/* rt_core.h -- the per-pixel sample loop and the iterative `ray_color`, over the
 * flattened scene (rt_flat.h).  This is the text the HIP kernel is built from.
 * It is also compiled for the host by oracle/oracle_flat.cpp, as TEST
 * infrastructure only (sanitizer builds, CPU-side checks of the flattener);
 * the shipped library has no CPU render path.
 *
 * Reference map (paths under /root/reference/src):
 *   rt_path_begin   main.rs:964-971 (seed, jitter, Camera::get_ray camera.rs:61-73)
 *   rt_path_step    main.rs:51-116 / :118-190, recursion unrolled into
 *                   L += beta*emitted ; beta *= att*spdf/pdf (or att)
 *   rt_traverse     bvh.rs:25-50 + aabb.rs:13-32 + every leaf `hit`
 *   rt_finish_hit   HitRecord::new hittable.rs:21-46 and the wrapper fix-ups
 *                   hittable.rs:206-226 (Translate), :237-279 (RotateY), :287-292 (FlipFace)
 *   rt_scatter...   material.rs, constant_medium.rs:36-51, pdf.rs, onb.rs, math.rs
 *   rt_texture      texture.rs:40-89, perlin.rs:46-106
 *
 * Traversal order is the reference's: depth first, left child before right,
 * with the running closest t as t_max (bvh.rs:38-47 is exactly that, because
 * the t_max handed to the right child is the left child's hit).  This matters
 * for results, not only speed: ConstantMedium::hit draws a random number
 * inside the traversal (constant_medium.rs:85) and ties in t go to the child
 * tested later.
 */
#ifndef RT1W_CORE_H
#define RT1W_CORE_H

#include <type_traits>
#include "rt_flat.h"

#define RT_POP_FLAG 0x80000000u
/* diagnostic build only (-DRT_STAMPS, never shipped): RT_STAMP(k) charges the wave's cycles
 * since the previous stamp to bucket k */
#ifndef RT_STAMP
#define RT_STAMP(k) ((void)0)
#endif
#ifndef RT_STAT_MAT
#define RT_STAT_MAT(mk) ((void)0)
#endif
#ifndef RT_STAT_VISIT
#define RT_STAT_VISIT(kind) ((void)0) /* hook for offline visit statistics (tools only) */
#endif

/* "does any lane of this wave want this?" -- a scalar branch on the GPU; the CPU test
 * build runs one lane at a time */
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_WAVE_ANY(p) (__ballot((p)) != 0ull)
/* make a wave-uniform loaded value "arrived" here, so that the waitcnt pass does not
 * carry a pending scalar load into the loop that follows */
#define RT_SCALAR_READY(x) asm volatile("" ::"s"(x))
#else
#define RT_WAVE_ANY(p) (p)
#define RT_SCALAR_READY(x) ((void)0)
#endif

struct RtRay { RtV3 o, d; double time; };
struct RtRayOD { RtV3 o, d; };

/* HitRecord, hittable.rs:10-18 (material as table index) */
struct RtHit {
    RtV3 p, n;
    double t, u, v;
    uint32_t mat;
    bool front;
};

struct RtPath {
    RtRay ray;
    RtV3 beta;      /* product of (attenuation*scattering_pdf/pdf) so far */
    RtV3 radiance;  /* sum of beta*emitted so far */
    RtRng rng;
    uint32_t depth_left;
    bool alive;
};

/* ------------------------------------------------------------ geometry -- */

RT_HD RtV3 rt_at(RtV3 o, RtV3 d, double t) { return o + t * d; } /* ray.rs:12-14 */

/* AABB::hit aabb.rs:13-32; inv = 1/direction is the same value at every node of
 * one ray, so it is computed once per ray-space instead of per node. */
RT_HD bool rt_aabb_hit(const double* bb, RtV3 o, RtV3 inv, double t_min, double t_max) {
#define RT_SLAB(minv, maxv, ov, iv)                      \
    {                                                    \
        double t0 = ((minv) - (ov)) * (iv);              \
        double t1 = ((maxv) - (ov)) * (iv);              \
        if ((iv) < RT_R(0.0)) { double s_ = t0; t0 = t1; t1 = s_; } \
        t_min = t0 > t_min ? t0 : t_min;                 \
        t_max = t1 < t_max ? t1 : t_max;                 \
        if (t_max <= t_min) return false;                \
    }
    RT_SLAB(bb[0], bb[3], o.x, inv.x)
    RT_SLAB(bb[1], bb[4], o.y, inv.y)
    RT_SLAB(bb[2], bb[5], o.z, inv.z)
#undef RT_SLAB
    return true;
}

/* The same test with the running interval kept by max/min instructions.  Identical result
 * whenever t_min and t_max are not NaN on entry: `t0 > t_min ? t0 : t_min` keeps t_min when t0
 * is NaN and so does maxNum; equal values or zeros of either sign give the same comparisons
 * afterwards (t_min, t_max are not outputs).  Callers take the literal form when a NaN bound
 * is present (a NaN root was accepted earlier -- reference behaviour, reproduced). */
#if defined(RT_F32)
RT_HD double rt_vmax(double a, double b) { return __builtin_fmaxf(a, b); }
RT_HD double rt_vmin(double a, double b) { return __builtin_fminf(a, b); }
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(RT_NO_ASM_MINMAX)
/* the bare instructions: __builtin_fmax/fmin make the compiler canonicalise each operand first
 * (a v_max_f64 x,x,x apiece); every operand here is an arithmetic result or a previous max/min */
RT_HD double rt_vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_HD double rt_vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
RT_HD double rt_vmax(double a, double b) { return __builtin_fmax(a, b); }
RT_HD double rt_vmin(double a, double b) { return __builtin_fmin(a, b); }
#endif
/* EARLY: leave at the first closed axis (pays when few lanes of the wave are in the test, e.g. inside a medium's
 * boundary walk: then the whole wave often leaves; chosen per scene -- kernels of scenes with media use it throughout).  !EARLY: no early return -- a full wave leaves early only if all of
 * its lanes do, which is rare, while the exits cost every box an exec-mask save/restore and a branch per axis; the
 * verdict is the conjunction of the three "interval still open" tests in either form (what the interval becomes after a
 * failed axis is never looked at).  Measured: Cornell +1.8 %, random_scene +2 % without the exits; cornel_smoke -2.8 %. */
template <bool EARLY>
RT_HD bool rt_aabb_hit_fast(const double* bb, RtV3 o, RtV3 inv, double t_min, double t_max) {
    bool open = true;
#define RT_SLAB(minv, maxv, ov, iv)                      \
    {                                                    \
        double t0 = ((minv) - (ov)) * (iv);              \
        double t1 = ((maxv) - (ov)) * (iv);              \
        if ((iv) < RT_R(0.0)) { double s_ = t0; t0 = t1; t1 = s_; } \
        t_min = rt_vmax(t0, t_min);                      \
        t_max = rt_vmin(t1, t_max);                      \
        if (EARLY) { if (t_max <= t_min) return false; } \
        else open = open & !(t_max <= t_min);            \
    }
    RT_SLAB(bb[0], bb[3], o.x, inv.x)
    RT_SLAB(bb[1], bb[4], o.y, inv.y)
    RT_SLAB(bb[2], bb[5], o.z, inv.z)
#undef RT_SLAB
    return open;
}

/* Sphere::hit sphere.rs:31-48 / MovingSphere::hit moving_sphere.rs:38-55: the root only */
RT_HD bool rt_sphere_root(RtV3 center, double radius, RtV3 o, RtV3 d, double t_min, double t_max,
                          double& root_out) {
    RtV3 oc = o - center;
    double a = rt_mag2(d);
    double half_b = rt_dot(oc, d);
    double c = rt_mag2(oc) - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < RT_R(0.0)) return false;
    double sqrtd = rt_sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    root_out = root;
    return true;
}
/* MovingSphere::center moving_sphere.rs:23-26 */
RT_HD RtV3 rt_msphere_center(const RtNode& nd, double time) {
    RtV3 c0 = rt_v3(nd.d[0], nd.d[1], nd.d[2]), c1 = rt_v3(nd.d[3], nd.d[4], nd.d[5]);
    return c0 + ((time - nd.e[0]) / (nd.e[1] - nd.e[0])) * (c1 - c0);
}
/* XYRect/XZRect/YZRect::hit aarect.rs:46-56,84-94,152-162: t and the in-bounds test.
 * (oa,da) is the axis normal to the plane, (ob,db),(oc,dc) the two in-plane axes. */
RT_HD bool rt_rect_t(const RtNode& nd, double oa, double da, double ob, double db, double oc,
                     double dc, double t_min, double t_max, double& t_out) {
    double t = (nd.d[4] - oa) / da;
    if (t < t_min || t > t_max) return false;
    double b = ob + t * db;
    double c = oc + t * dc;
    if (b < nd.d[0] || b > nd.d[1] || c < nd.d[2] || c > nd.d[3]) return false;
    t_out = t;
    return true;
}
/* the three rect kinds as one code path: axis selection by value instead of a 3-way branch
 * (same operands reach the same operations as in rt_rect_t) */
RT_HD bool rt_rect_any_t(const RtNode& nd, uint32_t kind, RtV3 o, RtV3 d, double t_min, double t_max, double& t) {
    double oa = kind == RT_XY ? o.z : (kind == RT_XZ ? o.y : o.x);
    double da = kind == RT_XY ? d.z : (kind == RT_XZ ? d.y : d.x);
    double ob = kind == RT_YZ ? o.y : o.x;
    double db = kind == RT_YZ ? d.y : d.x;
    double oc = kind == RT_XY ? o.y : o.z;
    double dc = kind == RT_XY ? d.y : d.z;
    return rt_rect_t(nd, oa, da, ob, db, oc, dc, t_min, t_max, t);
}
template <class Cfg>
RT_HD bool rt_prim_t(const RtNode& nd, uint32_t kind, RtV3 o, RtV3 d, double time, double t_min, double t_max,
                     double& t) {
    if (kind >= RT_XY) return rt_rect_any_t(nd, kind, o, d, t_min, t_max, t);
    if (Cfg::msphere && kind == RT_MSPHERE)
        return rt_sphere_root(rt_msphere_center(nd, time), nd.e[2], o, d, t_min, t_max, t);
    return rt_sphere_root(rt_v3(nd.d[0], nd.d[1], nd.d[2]), nd.d[3], o, d, t_min, t_max, t);
}

/* Translate::hit hittable.rs:207-211 / RotateY::hit hittable.rs:238-251: ray into the wrapper's space */
template <class NodeT>
RT_HD RtRayOD rt_scope_in(const NodeT& s, RtRayOD r) {
    if (s.kind == RT_TRANSLATE) {
        r.o = r.o - rt_v3(s.d[0], s.d[1], s.d[2]);
    } else if (s.kind == RT_ROTATE_Y) {
        double sn = s.d[0], cs = s.d[1];
        RtV3 o = r.o, d = r.d;
        r.o.x = cs * o.x - sn * o.z;
        r.o.z = sn * o.x + cs * o.z;
        r.d.x = cs * d.x - sn * d.z;
        r.d.z = sn * d.x + cs * d.z;
    }
    return r;
}
/* the way back out: hittable.rs:215-225, :255-278, :288-291.  `inner` is the ray in
 * the wrapper's space (`moved` / `rotated_r`): both re-run HitRecord::new with it,
 * taking the child's already-forwarded normal as "outward" (reference quirk Q5). */
RT_HD void rt_scope_out(const RtNode& s, RtRayOD inner, RtHit& h) {
    if (s.kind == RT_TRANSLATE) {
        h.p = h.p + rt_v3(s.d[0], s.d[1], s.d[2]);
        bool front = rt_dot(inner.d, h.n) < RT_R(0.0);
        h.n = front ? h.n : -h.n;
        h.front = front;
    } else if (s.kind == RT_ROTATE_Y) {
        double sn = s.d[0], cs = s.d[1];
        RtV3 p = h.p, n = h.n;
        p.x = cs * h.p.x + sn * h.p.z;
        p.z = -sn * h.p.x + cs * h.p.z;
        n.x = cs * h.n.x + sn * h.n.z;
        n.z = -sn * h.n.x + cs * h.n.z;
        bool front = rt_dot(inner.d, n) < RT_R(0.0);
        h.p = p;
        h.n = front ? n : -n;
        h.front = front;
    } else { /* RT_FLIP */
        h.front = !h.front;
    }
}

/* wrapper chain above a primitive, outermost first (depth <= RT_MAX_SCOPE_DEPTH) */
struct RtChain { uint32_t s0, s1, s2; };
RT_HD RtChain rt_chain(const RtNode* nodes, uint32_t scope) {
    RtChain c; c.s0 = c.s1 = c.s2 = RT_NONE;
    while (scope != RT_NONE) {
        c.s2 = c.s1; c.s1 = c.s0; c.s0 = scope;
        scope = nodes[scope].b;
    }
    return c;
}
RT_HD RtRayOD rt_ray_in_scope(const RtNode* nodes, uint32_t scope, RtRayOD world) {
    if (scope == RT_NONE) return world;
    RtChain c = rt_chain(nodes, scope);
    RtRayOD r = rt_scope_in(nodes[c.s0], world);
    if (c.s1 != RT_NONE) r = rt_scope_in(nodes[c.s1], r);
    if (c.s2 != RT_NONE) r = rt_scope_in(nodes[c.s2], r);
    return r;
}
/* the same, also telling whether a RotateY is among the wrappers: if none is, the direction -- and 1/direction -- is the world's */
RT_HD RtRayOD rt_ray_in_scope_r(const RtNode* nodes, uint32_t scope, RtRayOD world, bool& rotated) {
    rotated = false;
    if (scope == RT_NONE) return world;
    RtChain c = rt_chain(nodes, scope);
    rotated = (nodes[c.s0].kind & RT_KIND_MASK) == RT_ROTATE_Y;
    RtRayOD r = rt_scope_in(nodes[c.s0], world);
    if (c.s1 != RT_NONE) { rotated |= (nodes[c.s1].kind & RT_KIND_MASK) == RT_ROTATE_Y; r = rt_scope_in(nodes[c.s1], r); }
    if (c.s2 != RT_NONE) { rotated |= (nodes[c.s2].kind & RT_KIND_MASK) == RT_ROTATE_Y; r = rt_scope_in(nodes[c.s2], r); }
    return r;
}

/* sphere_uv math.rs:67-71 */
RT_HD void rt_sphere_uv(RtV3 p, double& u, double& v) {
    double theta = rt_acos(-p.y);
    double phi = rt_atan2(-p.z, p.x) + RT_R(RT_PI);
    u = phi / (RT_R(2.0) * RT_R(RT_PI));
    v = theta / RT_R(RT_PI);
}

/* The leaf's own HitRecord (sphere.rs:50-62, moving_sphere.rs:57-69, aarect.rs:58-71,
 * 96-109,164-177, constant_medium.rs:98-106) in the leaf's space.  u,v are only
 * evaluated when the material's texture reads them (image texture); they are
 * unobservable otherwise. */
template <class Cfg>
RT_HD void rt_leaf_record(const RtNode& nd, RtRayOD r, double time, double t, bool want_uv,
                          RtHit& h) {
    h.t = t; h.u = RT_R(0.0); h.v = RT_R(0.0); h.mat = nd.mat;
    h.p = rt_at(r.o, r.d, t);
    RtV3 on;
    const uint32_t kind = nd.kind & RT_KIND_MASK;
    const bool flipped = (nd.kind & RT_LEAF_FLIPPED) != 0u;
    if (kind == RT_SPHERE || kind == RT_MSPHERE) {
        if (!Cfg::msphere || kind == RT_SPHERE) on = (h.p - rt_v3(nd.d[0], nd.d[1], nd.d[2])) / nd.d[3];
        else on = (h.p - rt_msphere_center(nd, time)) / nd.e[2];
        if (Cfg::tex && want_uv) rt_sphere_uv(on, h.u, h.v);
    } else if (Cfg::media && kind == RT_MEDIUM) {
        h.n = rt_v3(RT_R(1.0), RT_R(0.0), RT_R(0.0));
        h.front = true;
        return;
    } else {
        double b, c;
        if (kind == RT_XY) { on = rt_v3(RT_R(0.0), RT_R(0.0), RT_R(1.0)); b = r.o.x + t * r.d.x; c = r.o.y + t * r.d.y; }
        else if (kind == RT_XZ) { on = rt_v3(RT_R(0.0), RT_R(1.0), RT_R(0.0)); b = r.o.x + t * r.d.x; c = r.o.z + t * r.d.z; }
        else { on = rt_v3(RT_R(1.0), RT_R(0.0), RT_R(0.0)); b = r.o.y + t * r.d.y; c = r.o.z + t * r.d.z; }
        if (Cfg::tex && want_uv) {
            h.u = (b - nd.d[0]) / (nd.d[1] - nd.d[0]);
            h.v = (c - nd.d[2]) / (nd.d[3] - nd.d[2]);
        }
    }
    /* HitRecord::new hittable.rs:30-35 */
    bool front = rt_dot(r.d, on) < RT_R(0.0);
    h.n = front ? on : -on;
    h.front = flipped ? !front : front; /* FlipFace::hit hittable.rs:288-291 flips the flag only */
}

/* Full hit record of the winning leaf: leaf record in its own space, then the
 * wrapper fix-ups innermost to outermost. */
template <class Cfg>
RT_HD void rt_finish_hit(const RtSceneView& sc, const RtRay& world, uint32_t prim, uint32_t scope,
                         double t, RtHit& h) {
    const RtNode* nodes = sc.nodes;
    const RtNode& nd = nodes[prim];
    bool want_uv = Cfg::tex && (RT_MAT_KINDF(nd.mat) & RT_MAT_NEEDS_UV) != 0u;
    RtRayOD r0; r0.o = world.o; r0.d = world.d;
    if (Cfg::scope_depth == 0 || scope == RT_NONE) {
        rt_leaf_record<Cfg>(nd, r0, world.time, t, want_uv, h);
        return;
    }
    RtChain c = rt_chain(nodes, scope);
    RtRayOD r1 = rt_scope_in(nodes[c.s0], r0);
    RtRayOD r2 = r1, r3 = r1;
    if (c.s1 != RT_NONE) { r2 = rt_scope_in(nodes[c.s1], r1); r3 = r2; }
    if (c.s2 != RT_NONE) r3 = rt_scope_in(nodes[c.s2], r2);
    rt_leaf_record<Cfg>(nd, r3, world.time, t, want_uv, h);
    if (c.s2 != RT_NONE) rt_scope_out(nodes[c.s2], r3, h);
    if (c.s1 != RT_NONE) rt_scope_out(nodes[c.s1], r2, h);
    rt_scope_out(nodes[c.s0], r1, h);
}

/* static sphere and the three rects, from the hot half of the record (same operations as
 * rt_prim_t).  `kind` is wave-uniform in the sweep, so the axis choice is a scalar branch. */
RT_HD bool rt_rect_hot_t(const RtNodeHot& nd, double oa, double da, double ob, double db, double oc, double dc,
                         double t_min, double t_max, double& t_out) {
    /* both of aarect.rs' rejections as one conjunction (no exec-mask region for the second half: the wave runs it
     * anyway unless every lane fails the first); a NaN passes each comparison exactly as it does there */
    const double t = (nd.d[4] - oa) / da;
    const bool in_t = !((t < t_min) | (t > t_max));
    const double b = ob + t * db;
    const double c = oc + t * dc;
    const bool in_rect = !((b < nd.d[0]) | (b > nd.d[1]) | (c < nd.d[2]) | (c > nd.d[3]));
    if (in_t & in_rect) { t_out = t; return true; }
    return false;
}
RT_HD bool rt_prim_hot_t(const RtNodeHot& nd, uint32_t kind, RtV3 o, RtV3 d, double t_min, double t_max, double& t_out) {
    if (kind == RT_XY) return rt_rect_hot_t(nd, o.z, d.z, o.x, d.x, o.y, d.y, t_min, t_max, t_out);
    if (kind == RT_XZ) return rt_rect_hot_t(nd, o.y, d.y, o.x, d.x, o.z, d.z, t_min, t_max, t_out);
    if (kind == RT_YZ) return rt_rect_hot_t(nd, o.x, d.x, o.y, d.y, o.z, d.z, t_min, t_max, t_out);
    return rt_sphere_root(rt_v3(nd.d[0], nd.d[1], nd.d[2]), nd.d[3], o, d, t_min, t_max, t_out);
}

/* static sphere / rects from the hot half when the kind differs per lane (stack walk): axis by select */
RT_HD bool rt_prim_hot_sel_t(const RtNodeHot& nd, uint32_t kind, RtV3 o, RtV3 d, double t_min, double t_max, double& t_out) {
    if (kind >= RT_XY) {
        double oa = kind == RT_XY ? o.z : (kind == RT_XZ ? o.y : o.x);
        double da = kind == RT_XY ? d.z : (kind == RT_XZ ? d.y : d.x);
        double ob = kind == RT_YZ ? o.y : o.x;
        double db = kind == RT_YZ ? d.y : d.x;
        double oc = kind == RT_XY ? o.y : o.z;
        double dc = kind == RT_XY ? d.y : d.z;
        return rt_rect_hot_t(nd, oa, da, ob, db, oc, dc, t_min, t_max, t_out);
    }
    return rt_sphere_root(rt_v3(nd.d[0], nd.d[1], nd.d[2]), nd.d[3], o, d, t_min, t_max, t_out);
}

/* Where the sweep gets the hot half of node n (n is wave-uniform):
 *  - RtGlobalNodes: from memory (scalar loads on the GPU);
 *  - a lane-resident source (context.hip, scenes of <= 64 nodes): lane i of every wave keeps
 *    node i's hot words in registers for the whole kernel and node n is read with
 *    v_readlane -- the traversal then touches no memory at all. */
struct RtGlobalNodes {
    static constexpr bool virt = false;
    const RtNode* p;
    RT_HD RtNodeHot hot(uint32_t n) const { return *reinterpret_cast<const RtNodeHot*>(p + n); }
};
/*  - RtWalkNodes (stack-walk kernels with the node cache): records of the WALK TABLE (rt_walk_table.h), addressed by a walk id
 *    instead of the node index -- the table lists the nodes most likely to be visited first, and its first `nc` records sit in
 *    LDS.  `virt`: stack entries are walk ids; a record names its children / itself by the fields below instead of e + 1 / e. */
struct RtWalkNodes {
    static constexpr bool virt = true;
    const RtNodeHot* lds;  /* records [0, nc) */
    const RtNodeHot* glob; /* the whole table */
    uint32_t nc;
    RT_HD RtNodeHot hot(uint32_t v) const { return v < nc ? lds[v] : glob[v]; }
};
/* where the context keeps the walk table: behind the node array and its spare record, at the next 128-byte boundary */
#define RT_WT_OFFSET(n_nodes) ((((size_t)(n_nodes) + 1u) * sizeof(RtNode) + 127u) & ~(size_t)127u)
#define RT_WT_INLINE_SPHERE 0x1000u /* walk-table record of a ConstantMedium: its boundary is a bare Sphere, centre in d[1..3], radius in d[4] */
/* what a record says about its neighbours.  Node arrays: the left / only child is the next node in pre-order and an entry IS the node
 * index.  Walk table: BVH node: skip = left / only child's walk id, b = right child's; wrapper: skip = child's walk id, mat = own node
 * index; leaf and medium: skip = own node index (medium: mat = walk id of the boundary's root) */
template <class NS> RT_HD uint32_t rt_ns_child(uint32_t e, const RtNodeHot& nd) { if constexpr (NS::virt) return nd.skip; else return e + 1u; }
template <class NS> RT_HD uint32_t rt_ns_leaf_id(uint32_t e, const RtNodeHot& nd) { if constexpr (NS::virt) return nd.skip; else return e; }
template <class NS> RT_HD uint32_t rt_ns_wrap_id(uint32_t e, const RtNodeHot& nd) { if constexpr (NS::virt) return nd.mat; else return e; }

/* ----------------------------------------------------------- traversal -- */

RT_HD RtV3 rt_inv3(RtV3 d) { return rt_v3(RT_R(1.0) / d.x, RT_R(1.0) / d.y, RT_R(1.0) / d.z); }

/* ConstantMedium::hit constant_medium.rs:58-113, given the two boundary roots t1, t2 */
RT_HD bool rt_medium_t(double neg_inv_density, RtV3 d, double t1, double t2, double t_min, double t_max, RtRng& rng,
                       double& t_out) {
    double rec1 = rt_max(t1, t_min);
    double rec2 = rt_min(t2, t_max);
    if (rec1 >= rec2) return false;
    rec1 = rt_max(rec1, RT_R(0.0));
    double ray_length = rt_mag(d);
    double distance_inside_boundary = (rec2 - rec1) * ray_length;
    rt_rng_reserve(rng, rt_rng_need_u64(rng));
    double hit_distance = neg_inv_density * rt_log(rt_take_f64(rng));
    if (hit_distance > distance_inside_boundary) return false;
    t_out = rec1 + hit_distance / ray_length;
    return true;
}
RT_HD bool rt_medium_t(const RtNode& nd, RtV3 d, double t1, double t2, double t_min, double t_max, RtRng& rng, double& t_out) {
    return rt_medium_t(nd.d[0], d, t1, t2, t_min, t_max, rng, t_out);
}

/* Closest hit of a subtree for a ray within [t_min, t_max], with an explicit per-lane stack (LDS
 * on the GPU), written as a resumable walk: rt_walk_begin pushes the root, rt_walk_step handles ONE
 * stack entry whatever its kind, rt_walk_done says when the stack is back to its base.  (A kernel
 * that ran one step per iteration and let finished lanes shade in batches while the others kept
 * walking was measured: it only ties the plain loop at equal occupancy and needs more registers.)
 * The walk is bound by the latency of the dependent node fetches, so the fewest iterations win (a
 * "while-while" split of box and leaf work was measured 0.6x).
 * Visiting order is the reference's (bvh.rs:38-47): left subtree, then right subtree with the
 * closest t so far as t_max. */
struct RtWalk {
    RtRayOD w;          /* the ray in the subtree's outer space */
    RtRayOD cur;        /* the ray in the current wrapper's space */
    RtV3 inv_w, inv;    /* 1/direction of w and of cur */
    double time, t_min, best_t;
    uint32_t scope, best_prim, best_scope;
    int base;           /* stack level at entry */
    bool tmin_nan;
};

/* `inv_known`: 1/direction of `world` if the caller has it already (a ConstantMedium's boundary is walked with the very ray the
 * outer walk holds, constant_medium.rs:62-69: the same three quotients, not computed again twice per medium) */
template <class Stack>
RT_HD void rt_walk_begin(RtWalk& k, uint32_t root, const RtRay& world, double t_min, double t_max, Stack& stk, const RtV3* inv_known = nullptr) {
    k.w.o = world.o; k.w.d = world.d;
    k.cur = k.w;
    k.inv_w = inv_known ? *inv_known : rt_inv3(k.w.d);
    k.inv = k.inv_w;
    k.time = world.time; k.t_min = t_min; k.best_t = t_max;
    k.scope = RT_NONE; k.best_prim = RT_NONE; k.best_scope = RT_NONE;
    k.tmin_nan = rt_isnan(t_min);
    k.base = stk.sp;
    stk.push(root);
}
template <class Stack>
RT_HD bool rt_walk_done(const RtWalk& k, const Stack& stk) { return stk.sp <= k.base; }

template <class Cfg, bool MEDIA, class Stack, class NS>
RT_HD bool rt_traverse_stack(const RtSceneView& sc, const NS& ns, uint32_t root, const RtRay& world, double t_min,
                             double t_max, RtRng& rng, Stack& stk, double& out_t, uint32_t& out_prim,
                             uint32_t& out_scope, const RtV3* inv_known = nullptr);

/* The walk's work, one piece per kind of stack entry (MEDIA=false is the flavour used for a ConstantMedium's
 * boundary, where only t is consumed, constant_medium.rs:62-69). */
enum { RT_WK_NONE = 0, RT_WK_BOX = 1, RT_WK_LEAF = 2, RT_WK_WRAP = 3, RT_WK_EXIT = 4, RT_WK_OTHER = 5 };
RT_HD uint32_t rt_walk_class(uint32_t kind) {
    return kind <= RT_BVH1 ? RT_WK_BOX : (kind <= RT_YZ ? RT_WK_LEAF : (kind <= RT_FLIP ? RT_WK_WRAP : RT_WK_OTHER));
}
/* leaving a wrapper: back to the parent's ray (recomputed from the outer ray by the same operations that
 * produced it, hence the same bits) */
RT_HD void rt_walk_exit(const RtSceneView& sc, RtWalk& k, uint32_t e) {
    const RtNode* nodes = sc.nodes;
    k.scope = nodes[e & ~RT_POP_FLAG].b;
    if (k.scope == RT_NONE) { k.cur = k.w; k.inv = k.inv_w; }
    else {
        bool rotated;
        k.cur = rt_ray_in_scope_r(nodes, k.scope, k.w, rotated);
        k.inv = k.inv_w; /* Translate / FlipFace leave the direction alone: the same three quotients (hittable.rs:207-211) */
        if (rotated) k.inv = rt_inv3(k.cur.d);
    }
}
#ifndef RT_BRANCHLESS_PUSH
#define RT_BRANCHLESS_PUSH 1
#endif
template <class Cfg, bool EARLY, class NS = RtGlobalNodes, class Stack>
RT_HD void rt_walk_box(RtWalk& k, uint32_t e, const RtNodeHot& nd, Stack& stk) {
    const uint32_t left = rt_ns_child<NS>(e, nd); /* the left / only child */
    bool hit;
    if (RT_WAVE_ANY(k.tmin_nan || rt_isnan(k.best_t))) hit = rt_aabb_hit(nd.d, k.cur.o, k.inv, k.t_min, k.best_t);
    else hit = rt_aabb_hit_fast<EARLY>(nd.d, k.cur.o, k.inv, k.t_min, k.best_t);
#if RT_BRANCHLESS_PUSH /* the CPU test build runs the same form: its stack bound-checks every poke (oracle_flat.cpp) */
    if constexpr (!Cfg::ordered) {
        /* both slots are written whatever the test said and the stack pointer moves by 0, 1 or 2: no nested exec-mask regions.
         * A miss, or a BVHChild::One, leaves dead words above the top: the flattener counts TWO slots for every BVH node
         * (scene.cpp, Flattener::emit), so the footprint of the pokes is inside stack_need */
        const bool two = (nd.kind & RT_KIND_MASK) == RT_BVH2;
        stk.poke(0, two ? nd.b : left); /* BVH2: the right child below the left one (bvh.rs:38-47); BVH1: its only child */
        stk.poke(1, left);
        stk.sp += hit ? (two ? 2 : 1) : 0;
        return;
    }
#endif
    if (hit) {
        if ((nd.kind & RT_KIND_MASK) == RT_BVH2) {
            uint32_t first = left, second = nd.b; /* left child (the next node in pre-order), then the right one: bvh.rs:38-47 */
            const uint32_t ord = Cfg::ordered ? (nd.kind >> RT_BVH_ORDER_SHIFT) & RT_BVH_ORDER_MASK : 0u;
            if (Cfg::ordered && ord != 0u) { /* opt-in near-far order (variant V4 only): the child on the ray's near side first */
                const double da = ord == 1u ? k.cur.d.x : (ord == 2u ? k.cur.d.y : k.cur.d.z);
                const bool left_lower = (nd.kind & RT_BVH_LEFT_LOWER) != 0u;
                if ((da < RT_R(0.0) && left_lower) || (da > RT_R(0.0) && !left_lower)) { first = nd.b; second = left; }
            }
            stk.push(second);
            stk.push(first);
        } else {
            stk.push(left); /* only child */
        }
    }
}
/* Ties in t go to the primitive the reference tests LATER (bvh.rs:40, sphere.rs:43, aarect.rs:48: `t_max < root` keeps an
 * equal root), i.e. to the larger pre-order index.  In the reference's own visiting order a later test always has the larger
 * index, so this never rejects anything there; under the opt-in near-far order it is what keeps equal-t hits the reference's. */
RT_HD bool rt_tie_ok(double t, uint32_t e, double best_t, uint32_t best_prim) {
    return !(t == best_t && best_prim != RT_NONE && e < best_prim);
}
template <class Cfg, class NS = RtGlobalNodes>
RT_HD void rt_walk_leaf(const RtSceneView& sc, RtWalk& k, uint32_t e_, const RtNodeHot& nd) {
    const uint32_t e = rt_ns_leaf_id<NS>(e_, nd); /* the node index: what the walk reports */
    const uint32_t kind = nd.kind & RT_KIND_MASK;
    double t;
    bool hit;
    if (Cfg::msphere && kind == RT_MSPHERE) hit = rt_prim_t<Cfg>(sc.nodes[e], kind, k.cur.o, k.cur.d, k.time, k.t_min, k.best_t, t);
    else hit = rt_prim_hot_sel_t(nd, kind, k.cur.o, k.cur.d, k.t_min, k.best_t, t);
    if (hit && (!Cfg::ordered || rt_tie_ok(t, e, k.best_t, k.best_prim))) { k.best_t = t; k.best_prim = e; k.best_scope = k.scope; }
}
template <class NS = RtGlobalNodes, class Stack>
RT_HD void rt_walk_wrap(RtWalk& k, uint32_t e_, const RtNodeHot& nd, Stack& stk) {
    const uint32_t kind = nd.kind & RT_KIND_MASK;
    const uint32_t e = rt_ns_wrap_id<NS>(e_, nd); /* scopes and exit entries are node indices in every form of the walk */
    stk.push(e | RT_POP_FLAG);
    k.scope = e;
    if (kind != RT_FLIP) {
        k.cur = rt_scope_in(nd, k.cur);
        if (kind == RT_ROTATE_Y) k.inv = rt_inv3(k.cur.d);
    }
    stk.push(rt_ns_child<NS>(e_, nd));
}
template <class Cfg, bool MEDIA, class Stack, class NS>
RT_HD void rt_walk_other(const RtSceneView& sc, const NS& ns, RtWalk& k, uint32_t e_, const RtNodeHot& nd, RtRng& rng, Stack& stk) {
    if (MEDIA && Cfg::media && (nd.kind & RT_KIND_MASK) == RT_MEDIUM) {
        /* ConstantMedium::hit constant_medium.rs:58-113: two complete boundary walks, then the free-flight draw */
        const uint32_t e = rt_ns_leaf_id<NS>(e_, nd);
        RtRay br; br.o = k.cur.o; br.d = k.cur.d; br.time = k.time;
        double t1, t2, t; uint32_t p_, s_;
        bool both;
        bool sphere; RtV3 c; double radius; uint32_t broot;
        if constexpr (NS::virt) {
            sphere = (nd.kind & RT_WT_INLINE_SPHERE) != 0u; c = rt_v3(nd.d[1], nd.d[2], nd.d[3]); radius = nd.d[4]; broot = nd.mat;
        } else {
            const RtNodeHot bn = ns.hot(e + 1u);
            sphere = (bn.kind & RT_KIND_MASK) == RT_SPHERE; c = rt_v3(bn.d[0], bn.d[1], bn.d[2]); radius = bn.d[3]; broot = e + 1u;
        }
        if (sphere) {
            /* the boundary is a bare Sphere (every medium of the reference's scenes): its walk is one stack entry, one leaf
             * test -- run the two tests (sphere.rs:31-48 with (-inf, inf), then (t1 + 0.0001, inf)) without the walk around them */
            both = rt_sphere_root(c, radius, br.o, br.d, -RT_INF, RT_INF, t1) && rt_sphere_root(c, radius, br.o, br.d, t1 + RT_R(0.0001), RT_INF, t2);
        } else if constexpr (Cfg::sphere_media) {
            both = false; (void)p_; (void)s_; (void)broot; /* not reached: the host gives these kernels only scenes without such a medium (RtCfgSphereMedia) */
        } else {
            both = rt_traverse_stack<Cfg, false>(sc, ns, broot, br, -RT_INF, RT_INF, rng, stk, t1, p_, s_, &k.inv) &&
                   rt_traverse_stack<Cfg, false>(sc, ns, broot, br, t1 + RT_R(0.0001), RT_INF, rng, stk, t2, p_, s_, &k.inv);
        }
        if (both && rt_medium_t(nd.d[0], k.cur.d, t1, t2, k.t_min, k.best_t, rng, t)) {
            k.best_t = t; k.best_prim = e; k.best_scope = k.scope;
        }
    }
}

#ifndef RT_MEDIUM_DEFER
#define RT_MEDIUM_DEFER 0 /* 0: media run where they are met; 1, 2: parked and run wave-wide (rt_traverse_stack): MEASURED slower,
                            final_scene 118 -> 106 / 114 Mpaths/s at 16 spp, bit-identical (profiles/r02_medium_defer.txt) */
#endif
/* what a popped node entry `e` does once its record `nd` is there */
template <class Cfg, bool MEDIA, class Stack, class NS>
RT_HD void rt_walk_visit(const RtSceneView& sc, const NS& ns, RtWalk& k, RtRng& rng, Stack& stk, uint32_t e, const RtNodeHot& nd) {
    const uint32_t km = nd.kind & RT_KIND_MASK;
    RT_STAT_VISIT(km);
    /* the kinds are numbered so that each class is a range: the most frequent one costs one compare */
    if (km <= RT_BVH1) rt_walk_box<Cfg, false, NS>(k, e, nd, stk); /* the slab test without early exits (they paid while media boundaries were walked) */
    else if (km <= RT_YZ) rt_walk_leaf<Cfg, NS>(sc, k, e, nd);
    else if (Cfg::scope_depth > 0 && km <= RT_FLIP) rt_walk_wrap<NS>(k, e, nd, stk); /* scope_depth 0: the scene has no wrapper node */
    else if (Cfg::media) rt_walk_other<Cfg, MEDIA>(sc, ns, k, e, nd, rng, stk);
}
/* one stack entry */
template <class Cfg, bool MEDIA, class Stack, class NS>
RT_HD void rt_walk_step(const RtSceneView& sc, const NS& ns, RtWalk& k, RtRng& rng, Stack& stk) {
    uint32_t e = stk.pop();
    if (Cfg::scope_depth > 0 && (e & RT_POP_FLAG)) { rt_walk_exit(sc, k, e); return; }
    const RtNodeHot nd = ns.hot(e); /* the hot 64 bytes, fetched in one go */
    rt_walk_visit<Cfg, MEDIA>(sc, ns, k, rng, stk, e, nd);
}

/* a step that only a BVH node takes: the lane pops its next entry and, if that is a BVH node, tests its box and pushes the children;
 * any other entry is put back untouched and waits for the next full step.  Same visits, same order per lane -- it only lets the
 * lanes that are between boxes (84 % of final_scene's visits) advance several nodes for each execution of the rare kinds' code
 * (media, wrappers, primitives), which a full step runs for two or three lanes of the wave. */
template <class Cfg, class Stack, class NS>
RT_HD bool rt_walk_box_step(const NS& ns, RtWalk& k, Stack& stk) {
    const uint32_t e = stk.pop();
    bool taken = false;
    if (!(Cfg::scope_depth > 0 && (e & RT_POP_FLAG))) {
        const RtNodeHot nd = ns.hot(e);
        if ((nd.kind & RT_KIND_MASK) <= RT_BVH1) { RT_STAT_VISIT(nd.kind & RT_KIND_MASK); rt_walk_box<Cfg, false, NS>(k, e, nd, stk); taken = true; }
    }
    if (!taken) stk.sp += 1; /* the entry is still where it was */
    return taken;
}

/* a full step that leaves a ConstantMedium where it is (the lane waits for the next step that takes media): everything else as
 * rt_walk_step.  A medium is the longest piece of the walk by far (two sphere roots, a logarithm, a draw) and runs for two or three
 * lanes of a wave: a kernel that alternates between the two kinds of step executes that code half as often. */
template <class Cfg, class Stack, class NS>
RT_HD void rt_walk_light_step(const RtSceneView& sc, const NS& ns, RtWalk& k, Stack& stk) {
    const uint32_t e = stk.pop();
    if (Cfg::scope_depth > 0 && (e & RT_POP_FLAG)) { rt_walk_exit(sc, k, e); return; }
    const RtNodeHot nd = ns.hot(e);
    const uint32_t km = nd.kind & RT_KIND_MASK;
    if (km <= RT_BVH1) { RT_STAT_VISIT(km); rt_walk_box<Cfg, false, NS>(k, e, nd, stk); }
    else if (km <= RT_YZ) { RT_STAT_VISIT(km); rt_walk_leaf<Cfg, NS>(sc, k, e, nd); }
    else if (Cfg::scope_depth > 0 && km <= RT_FLIP) { RT_STAT_VISIT(km); rt_walk_wrap<NS>(k, e, nd, stk); }
    else stk.sp += 1;
}

/* the same for a primitive (sphere, moving sphere, rect): a step that only a lane whose next entry is a primitive takes */
template <class Cfg, class Stack, class NS>
RT_HD bool rt_walk_prim_step(const RtSceneView& sc, const NS& ns, RtWalk& k, Stack& stk) {
    const uint32_t e = stk.pop();
    bool taken = false;
    if (!(Cfg::scope_depth > 0 && (e & RT_POP_FLAG))) {
        const RtNodeHot nd = ns.hot(e);
        const uint32_t km = nd.kind & RT_KIND_MASK;
        if (km > RT_BVH1 && km <= RT_YZ) { RT_STAT_VISIT(km); rt_walk_leaf<Cfg, NS>(sc, k, e, nd); taken = true; }
    }
    if (!taken) stk.sp += 1;
    return taken;
}

template <class Cfg, bool MEDIA, class Stack, class NS>
RT_HD bool rt_traverse_stack(const RtSceneView& sc, const NS& ns, uint32_t root, const RtRay& world, double t_min,
                             double t_max, RtRng& rng, Stack& stk, double& out_t, uint32_t& out_prim,
                             uint32_t& out_scope, const RtV3* inv_known) {
    RtWalk k;
    rt_walk_begin(k, root, world, t_min, t_max, stk, inv_known);
    if constexpr (MEDIA && Cfg::media && RT_MEDIUM_DEFER != 0) {
        /* A ConstantMedium costs two complete boundary walks, a logarithm and a draw (rt_walk_other), and the lanes of a wave
         * reach their media at different steps: run in place, that code executes once per lane with the other 63 waiting
         * (final_scene: every ray crosses the global fog, constant_medium.rs:58-113 -- 8.7 % of the lanes active, PMC).  So a
         * lane that pops a medium PARKS on it (it simply does not pop again) while the others walk on, and the wave runs the
         * medium code for all parked lanes together once no lane can step (RT_MEDIUM_DEFER 1) or once at least as many lanes
         * are parked as still walk (2).  A lane's own sequence of operations is untouched -- it only waits -- so every bit
         * of its result is; on the CPU build (one lane) "no lane can step" is true at once and this is the plain loop. */
        uint32_t parked = RT_NONE;
        for (;;) {
            const bool can_step = parked == RT_NONE && !rt_walk_done(k, stk);
#if defined(__HIP_DEVICE_COMPILE__)
            const unsigned long long stepping = __ballot(can_step), waiting = __ballot(parked != RT_NONE);
            if (stepping == 0ull && waiting == 0ull) break;
            const bool round = stepping == 0ull || (RT_MEDIUM_DEFER == 2 && __popcll(waiting) >= __popcll(stepping));
#else
            if (!can_step && parked == RT_NONE) break;
            const bool round = !can_step;
#endif
            if (round) {
                if (parked != RT_NONE) {
                    rt_walk_other<Cfg, MEDIA>(sc, ns, k, parked, ns.hot(parked), rng, stk);
                    parked = RT_NONE;
                }
                continue;
            }
            if (can_step) {
                const uint32_t e = stk.pop();
                if (e & RT_POP_FLAG) { rt_walk_exit(sc, k, e); continue; }
                const RtNodeHot nd = ns.hot(e);
                const uint32_t cls = rt_walk_class(nd.kind & RT_KIND_MASK);
                RT_STAT_VISIT(nd.kind & RT_KIND_MASK);
                if (cls == RT_WK_BOX) rt_walk_box<Cfg, false, NS>(k, e, nd, stk);
                else if (cls == RT_WK_LEAF) rt_walk_leaf<Cfg, NS>(sc, k, e, nd);
                else if (cls == RT_WK_WRAP) rt_walk_wrap<NS>(k, e, nd, stk);
                else parked = e;
            }
        }
    } else {
        while (!rt_walk_done(k, stk)) rt_walk_step<Cfg, MEDIA>(sc, ns, k, rng, stk);
    }
    out_t = k.best_t; out_prim = k.best_prim; out_scope = k.best_scope;
    return k.best_prim != RT_NONE;
}

/* The same walk with fewer steps: the lane keeps its CURRENT node in a register (stepping into the left child costs no stack
 * traffic), and a box node is visited together with its left child when that is a box too -- the reference tests the left
 * child immediately after the parent with the same closest hit (bvh.rs:36-39), so doing both in one step changes neither the
 * order nor any operand.  The left child's record is the next one in memory and is requested with the parent's (the node
 * array carries one spare record at its end).  The walk is paced by the latency and the bookkeeping of a step, not by the
 * slab arithmetic (PMC), so fewer, fatter steps pay: 48 -> 33 steps per segment on random_scene. */
/* RT_WALK_MODE: 0 = the classic loop (every visit is a pop), 1 = current node in a register (stepping into the first child
 * costs no stack traffic), 2 = 1 + the fused left child.  MEASURED (random_scene / final_scene, Mpaths/s): mode 0 475 / 117,
 * mode 2 395 / 105 -- the second record costs 16 registers in a kernel that already spills, and its loads compete with the
 * walk's own; mode 1: see DESIGN section 8.  Exactness of every mode is covered by the CPU tier (the tests build oracle B
 * with the library's default). */
#ifndef RT_WALK_MODE
#define RT_WALK_MODE 0
#endif
#define RT_FUSED_WALK (RT_WALK_MODE == 2)
template <class Cfg, class Stack>
RT_HD void rt_walk_push_children(RtWalk& k, uint32_t e, const RtNodeHot& nd, Stack& stk, uint32_t& first) {
    /* children of a BVH node whose box was hit: `first` is visited next, the other one (if any) is pushed */
    first = e + 1u;
    if ((nd.kind & RT_KIND_MASK) == RT_BVH2) {
        uint32_t second = nd.b;
        const uint32_t ord = Cfg::ordered ? (nd.kind >> RT_BVH_ORDER_SHIFT) & RT_BVH_ORDER_MASK : 0u;
        if (Cfg::ordered && ord != 0u) {
            const double da = ord == 1u ? k.cur.d.x : (ord == 2u ? k.cur.d.y : k.cur.d.z);
            const bool left_lower = (nd.kind & RT_BVH_LEFT_LOWER) != 0u;
            if ((da < RT_R(0.0) && left_lower) || (da > RT_R(0.0) && !left_lower)) { first = nd.b; second = e + 1u; }
        }
        stk.push(second);
    }
}
template <class Cfg, bool EARLY>
RT_HD bool rt_walk_slab(const RtWalk& k, const RtNodeHot& nd) {
    if (RT_WAVE_ANY(k.tmin_nan || rt_isnan(k.best_t))) return rt_aabb_hit(nd.d, k.cur.o, k.inv, k.t_min, k.best_t);
    return rt_aabb_hit_fast<EARLY>(nd.d, k.cur.o, k.inv, k.t_min, k.best_t);
}
template <class Cfg, bool MEDIA, class Stack, class NS>
RT_HD bool rt_traverse_stack_fused(const RtSceneView& sc, const NS& ns, uint32_t root, const RtRay& world, double t_min,
                                   double t_max, RtRng& rng, Stack& stk, double& out_t, uint32_t& out_prim, uint32_t& out_scope) {
    RtWalk k;
    k.w.o = world.o; k.w.d = world.d;
    k.cur = k.w;
    k.inv_w = rt_inv3(k.w.d);
    k.inv = k.inv_w;
    k.time = world.time; k.t_min = t_min; k.best_t = t_max;
    k.scope = RT_NONE; k.best_prim = RT_NONE; k.best_scope = RT_NONE;
    k.tmin_nan = rt_isnan(t_min);
    k.base = stk.sp;
    uint32_t e = root;
    for (;;) {
        bool do_pop = true;
        if (e & RT_POP_FLAG) {
            rt_walk_exit(sc, k, e);
        } else {
            const RtNodeHot nd = ns.hot(e);
            const RtNodeHot nl = RT_FUSED_WALK ? ns.hot(e + 1u) : nd; /* the next record in pre-order = the left child of a BVH node */
            const uint32_t cls = rt_walk_class(nd.kind & RT_KIND_MASK);
            RT_STAT_VISIT(nd.kind & RT_KIND_MASK);
            if (cls == RT_WK_BOX) {
                if (rt_walk_slab<Cfg, Cfg::media>(k, nd)) {
                    uint32_t first;
                    rt_walk_push_children<Cfg>(k, e, nd, stk, first);
                    const bool left_first = first == e + 1u;
                    e = first;
                    do_pop = false;
                    /* the left child in the same step when it is a box too (same closest hit: nothing happened in between) */
                    if (RT_FUSED_WALK && left_first && rt_walk_class(nl.kind & RT_KIND_MASK) == RT_WK_BOX) {
                        RT_STAT_VISIT(nl.kind & RT_KIND_MASK);
                        if (rt_walk_slab<Cfg, Cfg::media>(k, nl)) {
                            uint32_t f2;
                            rt_walk_push_children<Cfg>(k, e, nl, stk, f2);
                            e = f2;
                        } else {
                            do_pop = true;
                        }
                    }
                }
            } else if (cls == RT_WK_LEAF) {
                rt_walk_leaf<Cfg>(sc, k, e, nd);
            } else if (cls == RT_WK_WRAP) {
                const uint32_t kind = nd.kind & RT_KIND_MASK;
                stk.push(e | RT_POP_FLAG);
                k.scope = e;
                if (kind != RT_FLIP) {
                    k.cur = rt_scope_in(nd, k.cur);
                    if (kind == RT_ROTATE_Y) k.inv = rt_inv3(k.cur.d);
                }
                e = e + 1u;
                do_pop = false;
            } else {
                rt_walk_other<Cfg, MEDIA>(sc, ns, k, e, nd, rng, stk);
            }
        }
        if (do_pop) {
            if (stk.sp <= k.base) break;
            e = stk.pop();
        }
    }
    out_t = k.best_t; out_prim = k.best_prim; out_scope = k.best_scope;
    return k.best_prim != RT_NONE;
}

/* The same traversal without a stack, for small scenes.  Nodes are stored in
 * depth-first pre-order and every node knows where its subtree ends (`skip`), so
 * the reference's walk (left subtree, then right subtree) is "visit node n, then
 * either n+1 or skip[n]": each lane keeps only the index `cur` of the next node it
 * has to visit.  The loop index n is the same in all lanes of a wave, so the node
 * record comes in through SCALAR loads, the node kind is a scalar branch (no
 * divergence between node kinds), and lanes that do not visit node n just sit out
 * that step.  Every lane still visits exactly the nodes the stack walk would, in
 * the same order, with the same arithmetic. */
template <class Cfg, bool MEDIA, class NS>
RT_HD bool rt_traverse_sweep(const RtSceneView& sc, const NS& ns, uint32_t root, const RtRay& world, double t_min,
                             double t_max, RtRng& rng, double& out_t, uint32_t& out_prim, uint32_t& out_scope) {
    const RtNode* nodes = sc.nodes;
    RtRayOD w; w.o = world.o; w.d = world.d;
    RtRayOD cur_ray = w;
    const RtV3 inv_w = rt_inv3(w.d);
    RtV3 inv = inv_w;
    uint32_t scope = RT_NONE, scope_end = RT_NONE;
    double best_t = t_max;
    uint32_t best_prim = RT_NONE, best_scope = RT_NONE;
    const bool tmin_nan = rt_isnan(t_min);
    const uint32_t end = ns.hot(root).skip;
    uint32_t cur = root;
    for (uint32_t n = root; n < end; ++n) {
        const RtNodeHot nd = ns.hot(n);
        /* lanes whose wrapper's subtree ended before n go back to the parent's ray */
        if (RT_WAVE_ANY(scope_end <= n)) {
            if (scope_end <= n) {
                do {
                    scope = nodes[scope].b;
                    scope_end = (scope == RT_NONE) ? RT_NONE : nodes[scope].skip;
                } while (scope_end <= n);
                /* the parent's ray, recomputed from the outer ray by the operations that made it */
                if (scope == RT_NONE) { cur_ray = w; inv = inv_w; }
                else { cur_ray = rt_ray_in_scope(nodes, scope, w); inv = rt_inv3(cur_ray.d); }
            }
        }
        bool active = (cur == n);
        if (!RT_WAVE_ANY(active)) continue;
        const uint32_t kind = nd.kind & RT_KIND_MASK;
        if (active) {
            RT_STAT_VISIT(kind);
            if (kind <= RT_BVH1) {
                bool hit;
                if (RT_WAVE_ANY(tmin_nan || rt_isnan(best_t))) hit = rt_aabb_hit(nd.d, cur_ray.o, inv, t_min, best_t);
                else hit = rt_aabb_hit_fast<Cfg::media>(nd.d, cur_ray.o, inv, t_min, best_t);
                cur = hit ? n + 1u : nd.skip;
            } else if (kind <= RT_YZ) {
                double t;
                bool hit;
                if (Cfg::msphere && kind == RT_MSPHERE) hit = rt_prim_t<Cfg>(nodes[n], kind, cur_ray.o, cur_ray.d, world.time, t_min, best_t, t);
                else hit = rt_prim_hot_t(nd, kind, cur_ray.o, cur_ray.d, t_min, best_t, t);
                if (hit) { best_t = t; best_prim = n; best_scope = scope; }
                cur = n + 1u;
            } else if (kind <= RT_FLIP) {
                scope = n; scope_end = nd.skip;
                if (kind == RT_TRANSLATE) {
                    cur_ray.o = cur_ray.o - rt_v3(nd.d[0], nd.d[1], nd.d[2]);
                } else if (kind == RT_ROTATE_Y) {
                    double sn = nd.d[0], cs = nd.d[1];
                    RtV3 o = cur_ray.o, d = cur_ray.d;
                    cur_ray.o.x = cs * o.x - sn * o.z;
                    cur_ray.o.z = sn * o.x + cs * o.z;
                    cur_ray.d.x = cs * d.x - sn * d.z;
                    cur_ray.d.z = sn * d.x + cs * d.z;
                    inv = rt_inv3(cur_ray.d);
                }
                cur = n + 1u;
            } else {
                if (MEDIA && Cfg::media && kind == RT_MEDIUM) {
                    const RtNode& full = nodes[n];
                    RtRay br; br.o = cur_ray.o; br.d = cur_ray.d; br.time = world.time;
                    double t1, t2, t; uint32_t p_, s_;
                    if (rt_traverse_sweep<Cfg, false>(sc, ns, n + 1u, br, -RT_INF, RT_INF, rng, t1, p_, s_) &&
                        rt_traverse_sweep<Cfg, false>(sc, ns, n + 1u, br, t1 + RT_R(0.0001), RT_INF, rng, t2, p_, s_) &&
                        rt_medium_t(full, cur_ray.d, t1, t2, t_min, best_t, rng, t)) {
                        best_t = t; best_prim = n; best_scope = scope;
                    }
                }
                cur = nd.skip;
            }
        }
    }
    out_t = best_t; out_prim = best_prim; out_scope = best_scope;
    return best_prim != RT_NONE;
}

/* The same sweep for a scene whose node kinds and subtree ends are known at compile time (Topo::kind[], Topo::skip[]):
 * the loop over n is unrolled into straight-line code that follows the tree -- a subtree's code is guarded by "some
 * lane of the wave is at its root", node kinds need no dispatch, the node record of step I is fetched at a constant
 * offset, and a wrapper's subtree simply runs with the inner ray while the outer one stays live.  Every lane still
 * visits exactly the nodes of the reference's walk, in its order, with the same arithmetic. */
template <class Topo, class Cfg, bool MEDIA, uint32_t I, uint32_t END, class NS>
RT_HD void rt_sweep_static(const RtSceneView& sc, const NS& ns, const RtRayOD& ray, const RtV3& inv, double time, double t_min,
                           bool tmin_nan, RtRng& rng, uint32_t& cur, double& best_t, uint32_t& best_prim) {
    if constexpr (I < END) {
        constexpr uint32_t kind = Topo::kind[I] & RT_KIND_MASK;
        constexpr uint32_t skip = Topo::skip[I];
        if constexpr (kind <= RT_BVH1) {
            if (RT_WAVE_ANY(cur == I)) {
                const RtNodeHot nd = ns.hot(I);
                if (cur == I) {
                    bool hit;
                    if (RT_WAVE_ANY(tmin_nan || rt_isnan(best_t))) hit = rt_aabb_hit(nd.d, ray.o, inv, t_min, best_t);
                    else hit = rt_aabb_hit_fast<Cfg::media>(nd.d, ray.o, inv, t_min, best_t);
                    cur = hit ? I + 1u : skip;
                }
                rt_sweep_static<Topo, Cfg, MEDIA, I + 1u, skip>(sc, ns, ray, inv, time, t_min, tmin_nan, rng, cur, best_t, best_prim);
            }
            rt_sweep_static<Topo, Cfg, MEDIA, skip, END>(sc, ns, ray, inv, time, t_min, tmin_nan, rng, cur, best_t, best_prim);
        } else if constexpr (kind <= RT_YZ) {
            if (RT_WAVE_ANY(cur == I)) {
                const RtNodeHot nd = ns.hot(I);
                if (cur == I) {
                    double t;
                    bool hit;
                    if constexpr (Cfg::msphere && kind == RT_MSPHERE) hit = rt_prim_t<Cfg>(sc.nodes[I], kind, ray.o, ray.d, time, t_min, best_t, t);
                    else hit = rt_prim_hot_t(nd, kind, ray.o, ray.d, t_min, best_t, t);
                    if (hit) { best_t = t; best_prim = I; }
                    cur = I + 1u;
                }
            }
            rt_sweep_static<Topo, Cfg, MEDIA, I + 1u, END>(sc, ns, ray, inv, time, t_min, tmin_nan, rng, cur, best_t, best_prim);
        } else if constexpr (kind <= RT_FLIP) {
            if (RT_WAVE_ANY(cur == I)) {
                const RtNodeHot nd = ns.hot(I);
                cur = (cur == I) ? I + 1u : cur;
                if constexpr (kind == RT_FLIP) {
                    rt_sweep_static<Topo, Cfg, MEDIA, I + 1u, skip>(sc, ns, ray, inv, time, t_min, tmin_nan, rng, cur, best_t, best_prim);
                } else {
                    /* Translate::hit hittable.rs:207-211 / RotateY::hit :238-251 */
                    const RtRayOD inner = rt_scope_in(nd, ray);
                    if constexpr (kind == RT_ROTATE_Y) {
                        const RtV3 inv_inner = rt_inv3(inner.d);
                        rt_sweep_static<Topo, Cfg, MEDIA, I + 1u, skip>(sc, ns, inner, inv_inner, time, t_min, tmin_nan, rng, cur, best_t, best_prim);
                    } else {
                        rt_sweep_static<Topo, Cfg, MEDIA, I + 1u, skip>(sc, ns, inner, inv, time, t_min, tmin_nan, rng, cur, best_t, best_prim);
                    }
                }
            }
            rt_sweep_static<Topo, Cfg, MEDIA, skip, END>(sc, ns, ray, inv, time, t_min, tmin_nan, rng, cur, best_t, best_prim);
        } else {
            if (RT_WAVE_ANY(cur == I)) {
                if (cur == I) {
                    if constexpr (MEDIA && Cfg::media && kind == RT_MEDIUM) {
                        /* ConstantMedium::hit constant_medium.rs:58-113: the boundary is a sweep of its own subtree */
                        const RtNode& full = sc.nodes[I];
                        double t1 = RT_INF, t2 = RT_INF, t;
                        uint32_t c1 = I + 1u, p1 = RT_NONE, c2 = I + 1u, p2 = RT_NONE;
                        rt_sweep_static<Topo, Cfg, false, I + 1u, skip>(sc, ns, ray, inv, time, -RT_INF, false, rng, c1, t1, p1);
                        if (p1 != RT_NONE) {
                            const double lo = t1 + RT_R(0.0001);
                            rt_sweep_static<Topo, Cfg, false, I + 1u, skip>(sc, ns, ray, inv, time, lo, rt_isnan(lo), rng, c2, t2, p2);
                            if (p2 != RT_NONE && rt_medium_t(full, ray.d, t1, t2, t_min, best_t, rng, t)) { best_t = t; best_prim = I; }
                        }
                    }
                    cur = skip;
                }
            }
            rt_sweep_static<Topo, Cfg, MEDIA, skip, END>(sc, ns, ray, inv, time, t_min, tmin_nan, rng, cur, best_t, best_prim);
        }
    }
}

template <class Cfg, class Stack, class NS>
RT_HD bool rt_closest_hit(const RtSceneView& sc, const NS& ns, const RtRay& ray, double t_min, double t_max, RtRng& rng,
                          Stack& stk, double& t, uint32_t& prim, uint32_t& scope) {
    if constexpr (Cfg::sweep && !std::is_void<typename Cfg::Topo>::value) {
        typedef typename Cfg::Topo Topo;
        RtRayOD w; w.o = ray.o; w.d = ray.d;
        const RtV3 inv = rt_inv3(w.d);
        uint32_t cur = Topo::root;
        t = t_max; prim = RT_NONE;
        rt_sweep_static<Topo, Cfg, true, Topo::root, Topo::skip[Topo::root]>(sc, ns, w, inv, ray.time, t_min, rt_isnan(t_min), rng, cur, t, prim);
        scope = prim == RT_NONE ? RT_NONE : ns.hot(prim).b; /* leaves keep their innermost wrapper in `b` */
        return prim != RT_NONE;
    } else if constexpr (Cfg::sweep) return rt_traverse_sweep<Cfg, true>(sc, ns, sc.root, ray, t_min, t_max, rng, t, prim, scope);
    else if constexpr (RT_WALK_MODE != 0) return rt_traverse_stack_fused<Cfg, true>(sc, ns, sc.root, ray, t_min, t_max, rng, stk, t, prim, scope);
    else return rt_traverse_stack<Cfg, true>(sc, ns, sc.root, ray, t_min, t_max, rng, stk, t, prim, scope);
}

/* ------------------------------------------------------------ textures -- */

/* Perlin::noise perlin.rs:46-72 + perlin_interp :88-106 */
RT_HD double rt_perlin_noise(const RtPerlin& pl, RtV3 p) {
    double fx = rt_floor(p.x), fy = rt_floor(p.y), fz = rt_floor(p.z);
    double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    /* `as isize` saturates; & 255 afterwards */
    int64_t i = (fx >= RT_R(9.2e18)) ? INT64_MAX : (fx <= -RT_R(9.2e18)) ? INT64_MIN : (fx != fx ? 0 : (int64_t)fx);
    int64_t j = (fy >= RT_R(9.2e18)) ? INT64_MAX : (fy <= -RT_R(9.2e18)) ? INT64_MIN : (fy != fy ? 0 : (int64_t)fy);
    int64_t k = (fz >= RT_R(9.2e18)) ? INT64_MAX : (fz <= -RT_R(9.2e18)) ? INT64_MIN : (fz != fz ? 0 : (int64_t)fz);
    double uu = u * u * (RT_R(3.0) - RT_R(2.0) * u);
    double vv = v * v * (RT_R(3.0) - RT_R(2.0) * v);
    double ww = w * w * (RT_R(3.0) - RT_R(2.0) * w);
    double accum = RT_R(0.0);
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                /* wrapping add like the release build of the reference */
                uint32_t ii = (uint32_t)(((uint64_t)i + (uint64_t)di) & 255u);
                uint32_t jj = (uint32_t)(((uint64_t)j + (uint64_t)dj) & 255u);
                uint32_t kk = (uint32_t)(((uint64_t)k + (uint64_t)dk) & 255u);
                uint32_t idx = pl.perm_x[ii] ^ pl.perm_y[jj] ^ pl.perm_z[kk];
                RtV3 c = rt_v3(pl.ranvec[idx * 3 + 0], pl.ranvec[idx * 3 + 1], pl.ranvec[idx * 3 + 2]);
                double fi = (double)di, fj = (double)dj, fk = (double)dk;
                RtV3 weight_v = rt_v3(u - fi, v - fj, w - fk);
                accum += (fi * uu + (RT_R(1.0) - fi) * (RT_R(1.0) - uu)) * (fj * vv + (RT_R(1.0) - fj) * (RT_R(1.0) - vv)) *
                         (fk * ww + (RT_R(1.0) - fk) * (RT_R(1.0) - ww)) * rt_dot(c, weight_v);
            }
    return accum;
}
/* Perlin::turb perlin.rs:74-86 */
RT_HD double rt_perlin_turb(const RtPerlin& pl, RtV3 p, int depth) {
    double accum = RT_R(0.0), weight = RT_R(1.0);
    RtV3 temp_p = p;
    for (int i = 0; i < depth; ++i) {
        accum += weight * rt_perlin_noise(pl, temp_p);
        weight *= RT_R(0.5);
        temp_p = temp_p * RT_R(2.0);
    }
    return rt_abs(accum);
}
/* `x as u32` of Rust: saturating, NaN -> 0 */
RT_HD uint32_t rt_as_u32(double x) {
    if (!(x > RT_R(0.0))) return 0u;
    if (x >= RT_R(4294967295.0)) return 4294967295u;
    return (uint32_t)x;
}
/* Texture::value texture.rs:40-89 */
template <class Cfg>
RT_HD RtV3 rt_texture(const RtSceneView& sc, uint32_t tex, double u, double v, RtV3 p) {
    if (!Cfg::tex) { /* every texture of the scene is a SolidColor */
        const RtTexture& t = sc.textures[tex];
        return rt_v3(t.d[0], t.d[1], t.d[2]);
    }
    for (;;) {
        const RtTexture& t = sc.textures[tex];
        if (t.kind == RT_TEX_CHECKER) {
            double sines = rt_sin(RT_R(10.0) * p.x) * rt_sin(RT_R(10.0) * p.y) * rt_sin(RT_R(10.0) * p.z);
            tex = (sines < RT_R(0.0)) ? t.a : t.b;
            continue;
        }
        if (t.kind == RT_TEX_SOLID) return rt_v3(t.d[0], t.d[1], t.d[2]);
        if (t.kind == RT_TEX_NOISE) {
            double s = RT_R(1.0) * RT_R(0.5) * (RT_R(1.0) + rt_sin(t.d[0] * p.z + RT_R(10.0) * rt_perlin_turb(sc.perlin[t.a], p, 7)));
            return rt_v3(s, s, s);
        }
        /* RT_TEX_IMAGE texture.rs:67-88 */
        double uc = u < RT_R(0.0) ? RT_R(0.0) : (u > RT_R(1.0) ? RT_R(1.0) : u);
        double vc = RT_R(1.0) - (v < RT_R(0.0) ? RT_R(0.0) : (v > RT_R(1.0) ? RT_R(1.0) : v));
        uint32_t i = rt_as_u32(uc * (double)t.a);
        uint32_t j = rt_as_u32(vc * (double)t.b);
        i = i < t.a - 1u ? i : t.a - 1u;
        j = j < t.b - 1u ? j : t.b - 1u;
        const uint8_t* px = sc.images + t.c + ((size_t)j * t.a + i) * 3u;
        const double COLOR_SCALE = RT_R(1.0) / RT_R(255.0);
        return rt_v3((double)px[0] * COLOR_SCALE, (double)px[1] * COLOR_SCALE, (double)px[2] * COLOR_SCALE);
    }
}

/* texture.value(u, v, p) of a material: a SolidColor's value is kept in the material record itself */
template <class Cfg>
RT_HD RtV3 rt_mat_colour(const RtSceneView& sc, const RtMaterial& m, double u, double v, RtV3 p) {
    if (!Cfg::tex || (m.kind & RT_MAT_SOLID) != 0u) return rt_v3(m.d[0], m.d[1], m.d[2]);
    return rt_texture<Cfg>(sc, m.tex, u, v, p);
}

/* ------------------------------------------------------------ samplers -- */

/* math.rs:6-18 */
RT_HD RtV3 rt_random_in_unit_sphere(RtRng& rng) {
    for (;;) {
        rt_rng_reserve(rng, rt_rng_need_2u64(rng));
        double x = rt_take_range(rng, -RT_R(1.0), RT_R(1.0));
        double y = rt_take_range(rng, -RT_R(1.0), RT_R(1.0));
        rt_rng_reserve(rng, rt_rng_need_u64(rng));
        double z = rt_take_range(rng, -RT_R(1.0), RT_R(1.0));
        RtV3 v = rt_v3(x, y, z);
        if (rt_mag2(v) < RT_R(1.0)) return v;
    }
}
/* math.rs:30-37 */
RT_HD RtV3 rt_random_in_unit_disk(RtRng& rng) {
    for (;;) {
        rt_rng_reserve(rng, rt_rng_need_2u64(rng));
        double x = rt_take_range(rng, -RT_R(1.0), RT_R(1.0));
        double y = rt_take_range(rng, -RT_R(1.0), RT_R(1.0));
        RtV3 p = rt_v3(x, y, RT_R(0.0));
        if (rt_mag2(p) < RT_R(1.0)) return p;
    }
}
/* math.rs:39-49 (caller has reserved two 64-bit draws) */
RT_HD RtV3 rt_random_cosine_direction(RtRng& rng) {
    double r1 = rt_take_f64(rng);
    double r2 = rt_take_f64(rng);
    double z = rt_sqrt(RT_R(1.0) - r2);
    double phi = RT_R(2.0) * RT_R(RT_PI) * r1;
    double s, c;
    rt_sincos(phi, s, c);
    double sr2 = rt_sqrt(r2);
    return rt_v3(c * sr2, s * sr2, z);
}
/* math.rs:51-65 (caller has reserved two 64-bit draws) */
RT_HD RtV3 rt_random_to_sphere(double radius, double distance_squared, RtRng& rng) {
    double r1 = rt_take_f64(rng);
    double r2 = rt_take_f64(rng);
    double z = RT_R(1.0) + r2 * (rt_sqrt(RT_R(1.0) - radius * radius / distance_squared) - RT_R(1.0));
    double phi = RT_R(2.0) * RT_R(RT_PI) * r1;
    double s, c;
    rt_sincos(phi, s, c);
    double q = rt_sqrt(RT_R(1.0) - z * z);
    return rt_v3(c * q, s * q, z);
}

/* Onb onb.rs:13-28 */
struct RtOnb { RtV3 u, v, w; };
RT_HD RtOnb rt_onb_from_w(RtV3 n) {
    RtOnb o;
    o.w = rt_normalize(n);
    RtV3 a = (rt_abs(o.w.x) > RT_R(0.9)) ? rt_v3(RT_R(0.0), RT_R(1.0), RT_R(0.0)) : rt_v3(RT_R(1.0), RT_R(0.0), RT_R(0.0));
    o.v = rt_normalize(rt_cross(o.w, a));
    o.u = rt_cross(o.w, o.v);
    return o;
}
RT_HD RtV3 rt_onb_local(const RtOnb& o, RtV3 a) { return o.u * a.x + o.v * a.y + o.w * a.z; }

/* ---------------------------------------------------------------- lights -- */

/* Hittable::pdf_value of one light: XZRect aarect.rs:119-138, Sphere sphere.rs:72-90,
 * anything else the trait default 0.0 (hittable.rs:66-68) */
RT_HD double rt_light_pdf_value(const RtNode& l, RtV3 o, RtV3 v) {
    if (l.kind == RT_XZ) {
        double t;
        if (!rt_rect_t(l, o.y, v.y, o.x, v.x, o.z, v.z, RT_R(0.001), RT_INF, t)) return RT_R(0.0);
        RtV3 on = rt_v3(RT_R(0.0), RT_R(1.0), RT_R(0.0));
        RtV3 normal = (rt_dot(v, on) < RT_R(0.0)) ? on : -on;
        double area = (l.d[1] - l.d[0]) * (l.d[3] - l.d[2]);
        double distance_squared = t * t * rt_mag2(v);
        double cosine = rt_abs(rt_dot(v, normal) / rt_mag(v));
        return distance_squared / (cosine * area);
    }
    if (l.kind == RT_SPHERE) {
        double t;
        RtV3 center = rt_v3(l.d[0], l.d[1], l.d[2]);
        if (!rt_sphere_root(center, l.d[3], o, v, RT_R(0.001), RT_INF, t)) return RT_R(0.0);
        double cos_theta_max = rt_sqrt(RT_R(1.0) - l.d[3] * l.d[3] / rt_mag2(center - o));
        double solid_angle = RT_R(2.0) * RT_R(RT_PI) * (RT_R(1.0) - cos_theta_max);
        return RT_R(1.0) / solid_angle;
    }
    return RT_R(0.0);
}
/* Hittable::random of one light: aarect.rs:140-147, sphere.rs:92-99, default (1,0,0)
 * (caller has reserved two 64-bit draws) */
RT_HD RtV3 rt_light_random(const RtNode& l, RtV3 o, RtRng& rng) {
    if (l.kind == RT_XZ) {
        double x = rt_take_range(rng, l.d[0], l.d[1]);
        double z = rt_take_range(rng, l.d[2], l.d[3]);
        return rt_v3(x, l.d[4], z) - o;
    }
    if (l.kind == RT_SPHERE) {
        RtV3 direction = rt_v3(l.d[0], l.d[1], l.d[2]) - o;
        double distance_squared = rt_mag2(direction);
        RtOnb uvw = rt_onb_from_w(direction);
        return rt_onb_local(uvw, rt_random_to_sphere(l.d[3], distance_squared, rng));
    }
    return rt_v3(RT_R(1.0), RT_R(0.0), RT_R(0.0));
}
/* impl Hittable for [T]: pdf_value hittable.rs:144-150 */
RT_HD double rt_lights_pdf_value(const RtSceneView& sc, RtV3 o, RtV3 v) {
    double weight = RT_R(1.0) / (double)sc.n_lights;
    double sum = RT_R(0.0);
    for (uint32_t i = 0; i < sc.n_lights; ++i) sum = sum + weight * rt_light_pdf_value(sc.lights[i], o, v);
    return sum;
}

/* ------------------------------------------------------------ materials -- */

RT_HD RtV3 rt_reflect(RtV3 v, RtV3 n) { return v - RT_R(2.0) * rt_dot(v, n) * n; } /* material.rs:94-96 */
RT_HD RtV3 rt_refract(RtV3 uv, RtV3 n, double etai_over_etat) {               /* material.rs:114-119 */
    double cos_theta = rt_min(rt_dot(-uv, n), RT_R(1.0));
    RtV3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    RtV3 r_out_parallel = -rt_sqrt(rt_abs(RT_R(1.0) - rt_mag2(r_out_perp))) * n;
    return r_out_perp + r_out_parallel;
}
RT_HD double rt_reflectance(double cosine, double ref_idx) {                  /* material.rs:121-125 */
    double r0 = (RT_R(1.0) - ref_idx) / (RT_R(1.0) + ref_idx);
    r0 = r0 * r0;
    return r0 + (RT_R(1.0) - r0) * rt_pow5(RT_R(1.0) - cosine);
}

/* --------------------------------------------------------------- camera -- */

/* main.rs:964-971 + Camera::get_ray camera.rs:61-73 */
RT_HD void rt_path_begin_cam(const RtCamera& c, const RtFrame& f, uint32_t i, uint32_t j,
                         uint32_t sample, RtPath& p);
RT_HD void rt_path_begin(const RtSceneView& sc, const RtFrame& f, uint32_t i, uint32_t j,
                         uint32_t sample, RtPath& p) { rt_path_begin_cam(sc.camera, f, i, j, sample, p); }
RT_HD void rt_path_begin_cam(const RtCamera& c, const RtFrame& f, uint32_t i, uint32_t j,
                         uint32_t sample, RtPath& p) {
#if defined(RT_RNG_REFSTREAM)
    /* main.rs:964-967: the pixel's stream is seeded once, before its sample loop, and every sample draws on from where
     * the previous one stopped (a lane runs all samples of its pixel in order: chunk = spp, sample_offset = 0) */
    if (sample == 0u) p.rng = rt_rng_make((uint32_t)((uint64_t)j * f.width + i), (uint32_t)(((uint64_t)j * f.width + i) >> 32), 0u, 0u, 0u);
#else
    p.rng = rt_rng_pixel_sample((uint64_t)j * f.width + i, sample, f.global_seed);
#endif
    rt_rng_reserve(p.rng, 4u);
    double u = ((double)i + rt_take_f64(p.rng)) / (double)(f.width - 1u);
    double v = ((double)j + rt_take_f64(p.rng)) / (double)(f.height - 1u);
    RtV3 rd = c.lens_radius * rt_random_in_unit_disk(p.rng);
    RtV3 offset = c.u * rd.x + c.v * rd.y;
    p.ray.o = c.origin + offset;
    p.ray.d = c.lower_left_corner + u * c.horizontal + v * c.vertical - c.origin - offset;
    rt_rng_reserve(p.rng, rt_rng_need_u64(p.rng));
    p.ray.time = rt_take_range(p.rng, c.time0, c.time1);
    p.beta = rt_v3(RT_R(1.0), RT_R(1.0), RT_R(1.0));
    p.radiance = rt_v3(RT_R(0.0), RT_R(0.0), RT_R(0.0));
    p.depth_left = f.max_depth;
    p.alive = true;
}

/* ------------------------------------------------------------ integrator -- */

/* One level of ray_color (main.rs:51-116; with no lights :118-190).  The
 * recursion  L = emitted + W (.) L'/pdf  is carried as  radiance += beta (.) emitted,
 * beta = beta (.) W / pdf.  Every terminal adds beta (.) value -- including the zero
 * of depth exhaustion (main.rs:59-61) -- so a non-finite beta poisons the sample
 * exactly as it does through the reference's multiplications. */
/* The closest hit found by the first half of a step, and what the second half will do with
 * it: the class is the sort key of the workgroup-level reordering (context.hip). */
struct RtTrace {
    double t;
    uint32_t prim, scope;
    uint32_t cls; /* RT_CLS_* */
};
/* order = position in the sorted workgroup: Lambertian paths fill the first waves, the paths that end
 * (and will regenerate) the last one; other orders were measured 0.5-4 % slower */
enum { RT_CLS_LAMBERT = 0, RT_CLS_DIELECTRIC = 1, RT_CLS_METAL = 2, RT_CLS_OTHER = 3, RT_CLS_TERMINAL = 4, RT_CLS_IDLE = 5, RT_N_CLS = 6 };

/* first half of one level of ray_color: depth check (main.rs:59-61) and world.hit (main.rs:62) */
template <class Cfg, class Stack, class NS>
RT_HD RtTrace rt_path_trace(const RtSceneView& sc, const NS& ns, RtPath& p, Stack& stk) {
    RtTrace tr;
    tr.t = RT_R(0.0); tr.prim = RT_NONE; tr.scope = RT_NONE; tr.cls = RT_CLS_TERMINAL;
    if (p.depth_left == 0u) return tr;
    bool found = rt_closest_hit<Cfg>(sc, ns, p.ray, RT_R(0.001), RT_INF, p.rng, stk, tr.t, tr.prim, tr.scope);
    if (!found) { tr.prim = RT_NONE; return tr; }
    uint32_t mk = RT_MAT_KINDF(ns.hot(tr.prim).mat) & 0xFFu; /* the node carries its material's kind word */
    tr.cls = mk == RT_MAT_LAMBERTIAN ? RT_CLS_LAMBERT
           : mk == RT_MAT_DIELECTRIC ? RT_CLS_DIELECTRIC
           : mk == RT_MAT_METAL ? RT_CLS_METAL
           : mk == RT_MAT_ISOTROPIC ? RT_CLS_OTHER : RT_CLS_TERMINAL;
    return tr;
}

/* second half: emitted / scatter / the next ray (main.rs:63-115).  The recursion
 * L = emitted + W (.) L'/pdf  is carried as  radiance += beta (.) emitted,
 * beta = beta (.) W / pdf.  Every terminal adds beta (.) value -- including the zero
 * of depth exhaustion (main.rs:59-61) -- so a non-finite beta poisons the sample
 * exactly as it does through the reference's multiplications. */
template <class Cfg>
RT_HD void rt_path_shade(const RtSceneView& sc, RtPath& p, const RtTrace& tr) {
    if (p.depth_left == 0u) {
        p.radiance = p.radiance + rt_mul(p.beta, rt_v3(RT_R(0.0), RT_R(0.0), RT_R(0.0)));
        p.alive = false;
        return;
    }
    if (tr.prim == RT_NONE) {
        p.radiance = p.radiance + rt_mul(p.beta, sc.background);
        p.alive = false;
        return;
    }
    const double t = tr.t;
    const uint32_t prim = tr.prim, scope = tr.scope;
    RtHit h;
    rt_finish_hit<Cfg>(sc, p.ray, prim, scope, t, h);
    RT_STAMP(3);
    const RtMaterial& m = sc.materials[RT_MAT_INDEX(h.mat)];
    uint32_t mk = RT_MAT_KINDF(h.mat) & 0xFFu;
    RT_STAT_MAT(mk);

    /* emitted: DiffuseLight material.rs:168-181 (front face only); every other
     * material returns the default (0,0,0) (material.rs:40-49), whose addition in
     * main.rs:92 changes nothing, so it is not carried.  In this reference the
     * emitting material never scatters, so a path's radiance is beta (.) its
     * terminal value (light, background, or the zero of depth exhaustion). */
    if (mk == RT_MAT_DIFFUSE_LIGHT) {
        RtV3 emitted = h.front ? rt_mat_colour<Cfg>(sc, m, h.u, h.v, h.p) : rt_v3(RT_R(0.0), RT_R(0.0), RT_R(0.0));
        p.radiance = p.radiance + rt_mul(p.beta, emitted);
        p.alive = false; /* DiffuseLight::scatter -> None, main.rs:110-112 */
        return;
    }

    if (mk == RT_MAT_LAMBERTIAN) {
        /* Lambertian::scatter material.rs:71-80 -> ScatterKind::Pdf(CosinePdf) */
        RtV3 attenuation = rt_mat_colour<Cfg>(sc, m, h.u, h.v, h.p);
        RtOnb uvw = rt_onb_from_w(h.n);
        RtV3 dir;
        double pdf;
        /* MixturePdf::generate pdf.rs:62-68 (p0 = HittablePdf{lights}, p1 = CosinePdf), or CosinePdf alone.
         * The cosine lobe (math.rs:39-49), the sphere light (sphere.rs:92-99 -> math.rs:51-65) and the
         * rect light (aarect.rs:140-147) each consume two 64-bit draws, so the draws are taken once for
         * the three cases and only the conversion differs; cosine and sphere share one sincos. */
        uint32_t lk = RT_NONE; /* RT_NONE: cosine lobe */
        uint32_t li = 0u;
        if (sc.n_lights > 0u) {
            rt_rng_fill(p.rng);
            if (rt_take_bool(p.rng)) {
                li = rt_gen_below(p.rng, sc.n_lights); /* choose, hittable.rs:153 */
                lk = sc.lights[li].kind;
            }
        }
        if (lk == RT_NONE || lk == RT_XZ || lk == RT_SPHERE) {
            rt_rng_reserve(p.rng, rt_rng_need_2u64(p.rng));
            uint64_t q1 = rt_take_u64(p.rng);
            uint64_t q2 = rt_take_u64(p.rng);
            if (lk == RT_XZ) {
                const RtNode& l = sc.lights[li];
                double x = rt_range_from_bits(q1, l.d[0], l.d[1]);
                if (!(x < l.d[1])) { /* rounding onto `high`: the retry takes the next draw (rand 0.8 sample_single) */
                    x = rt_range_from_bits(q2, l.d[0], l.d[1]);
                    while (!(x < l.d[1])) x = rt_range_from_bits(rt_next_u64(p.rng), l.d[0], l.d[1]);
                    q2 = rt_next_u64(p.rng);
                }
                double z = rt_range_from_bits(q2, l.d[2], l.d[3]);
                while (!(z < l.d[3])) z = rt_range_from_bits(rt_next_u64(p.rng), l.d[2], l.d[3]);
                dir = rt_v3(x, l.d[4], z) - h.p;
            } else {
                double r1 = rt_f64_from_bits(q1);
                double r2 = rt_f64_from_bits(q2);
                double phi = RT_R(2.0) * RT_R(RT_PI) * r1;
                double sn, cs;
                rt_sincos(phi, sn, cs);
                if (lk == RT_SPHERE) {
                    const RtNode& l = sc.lights[li];
                    RtV3 direction = rt_v3(l.d[0], l.d[1], l.d[2]) - h.p;
                    double distance_squared = rt_mag2(direction);
                    RtOnb lw = rt_onb_from_w(direction);
                    double z = RT_R(1.0) + r2 * (rt_sqrt(RT_R(1.0) - l.d[3] * l.d[3] / distance_squared) - RT_R(1.0));
                    double q = rt_sqrt(RT_R(1.0) - z * z);
                    dir = rt_onb_local(lw, rt_v3(cs * q, sn * q, z));
                } else {
                    double z = rt_sqrt(RT_R(1.0) - r2);
                    double sr2 = rt_sqrt(r2);
                    dir = rt_onb_local(uvw, rt_v3(cs * sr2, sn * sr2, z));
                }
            }
        } else {
            dir = rt_v3(RT_R(1.0), RT_R(0.0), RT_R(0.0)); /* Hittable::random default, hittable.rs:69-71 */
        }
        double p1 = rt_max(rt_dot(rt_normalize(dir), uvw.w) / RT_R(RT_PI), RT_R(0.0)); /* CosinePdf::value pdf.rs:37-40 */
        if (sc.n_lights > 0u) {
            double p0 = rt_lights_pdf_value(sc, h.p, dir);
            pdf = RT_R(0.5) * p0 + RT_R(0.5) * p1; /* MixturePdf::value pdf.rs:58-60 */
        } else {
            pdf = p1;
        }
        /* Lambertian::scattering_pdf material.rs:82-91 */
        double spdf = rt_max(rt_dot(h.n, rt_normalize(dir)) / RT_R(RT_PI), RT_R(0.0));
        p.beta = rt_mul(p.beta, attenuation * spdf) / pdf;
        p.ray.o = h.p; p.ray.d = dir;
        p.ray.time = h.t; /* main.rs:86: time = hit_record.t (reference quirk Q1) */
        RT_STAMP(4);
    } else if (mk == RT_MAT_METAL) {
        /* Metal::scatter material.rs:99-111 */
        RtV3 reflected = rt_reflect(rt_normalize(p.ray.d), h.n);
        RtV3 dir = reflected + m.d[3] * rt_random_in_unit_sphere(p.rng);
        p.beta = rt_mul(p.beta, rt_v3(m.d[0], m.d[1], m.d[2]));
        p.ray.o = h.p; p.ray.d = dir;
    } else if (mk == RT_MAT_DIELECTRIC) {
        /* Dielectric::scatter material.rs:133-160 */
        double refraction_ratio = h.front ? RT_R(1.0) / m.d[0] : m.d[0];
        RtV3 unit_direction = rt_normalize(p.ray.d);
        double cos_theta = rt_min(rt_dot(-unit_direction, h.n), RT_R(1.0));
        double sin_theta = rt_sqrt(RT_R(1.0) - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > RT_R(1.0);
        RtV3 dir;
        if (cannot_refract || rt_reflectance(cos_theta, refraction_ratio) > rt_gen_f64(p.rng)) /* checked draw: rarely-run site */
            dir = rt_reflect(unit_direction, h.n);
        else
            dir = rt_refract(unit_direction, h.n, refraction_ratio);
        p.beta = rt_mul(p.beta, rt_v3(RT_R(1.0), RT_R(1.0), RT_R(1.0)));
        p.ray.o = h.p; p.ray.d = dir;
    } else if (Cfg::media && mk == RT_MAT_ISOTROPIC) {
        /* Isotropic::scatter constant_medium.rs:37-50 */
        RtV3 attenuation = rt_mat_colour<Cfg>(sc, m, h.u, h.v, h.p);
        RtV3 dir = rt_random_in_unit_sphere(p.rng);
        p.beta = rt_mul(p.beta, attenuation);
        p.ray.o = h.p; p.ray.d = dir;
    } else {
        /* impl Material for (): scatter -> None, emitted (0,0,0): main.rs:110-112 */
        p.radiance = p.radiance + rt_mul(p.beta, rt_v3(RT_R(0.0), RT_R(0.0), RT_R(0.0)));
        p.alive = false;
        return;
    }
    p.depth_left -= 1u;
    RT_STAMP(5);
}

/* One level of ray_color (main.rs:51-116; with no lights :118-190) */
template <class Cfg, class Stack, class NS>
RT_HD void rt_path_step(const RtSceneView& sc, const NS& ns, RtPath& p, Stack& stk) {
    RtTrace tr = rt_path_trace<Cfg>(sc, ns, p, stk);
    rt_path_shade<Cfg>(sc, p, tr);
}

/* ----------------------------------------------------------------- color -- */

/* Color::into_sampled color.rs:14-21 */
RT_HD RtV3 rt_into_sampled(RtV3 sum, uint32_t samples_per_pixel) {
    double scale = RT_R(1.0) / (double)samples_per_pixel;
    double r = rt_isnan(sum.x) ? RT_R(0.0) : sum.x;
    double g = rt_isnan(sum.y) ? RT_R(0.0) : sum.y;
    double b = rt_isnan(sum.z) ? RT_R(0.0) : sum.z;
    return rt_v3(r, g, b) * scale;
}
/* Display for SampledColor color.rs:56-65: (256 * sqrt(c).clamp(0, 0.999)) as usize */
RT_HD uint32_t rt_quantize(double c) {
    double s = rt_sqrt(c);
    if (s < RT_R(0.0)) s = RT_R(0.0);
    if (s > RT_R(0.999)) s = RT_R(0.999);
    double q = RT_R(256.0) * s;
    return (q != q) ? 0u : (uint32_t)q;
}

/* work item -> (pixel, chunk): items are numbered so that 64 consecutive items
 * are an 8x8 pixel block of one chunk (rays of one wave start coherent). */
RT_HD void rt_item_decode(const RtFrame& f, uint64_t item, uint32_t& px, uint32_t& py, uint32_t& chunk) {
#if !defined(RT_NO_PROBE)
    if (f.probe) item = ((item >> 6) << 6) | ((item >> 6) & 63u); /* RtFrame::probe: 64 consecutive ids (one wave's) -> one item, one pixel per 8x8 block */
#endif
    uint32_t bw = (f.tile_w + 7u) >> 3, bh = (f.tile_h + 7u) >> 3;
    uint64_t per_chunk = (uint64_t)bw * bh * 64u;
    chunk = (uint32_t)(item / per_chunk);
    uint64_t r = item % per_chunk;
    uint32_t blk = (uint32_t)(r >> 6), in = (uint32_t)(r & 63u);
    px = (blk % bw) * 8u + (in & 7u);
    py = (blk / bw) * 8u + (in >> 3);
}
RT_HD uint64_t rt_item_count(const RtFrame& f) {
    uint32_t bw = (f.tile_w + 7u) >> 3, bh = (f.tile_h + 7u) >> 3;
    return (uint64_t)bw * bh * 64u * f.n_chunks;
}

#endif
