/* scene.cpp -- scene half of the C ABI (include/rt1w.h): constructors, BVH build,
 * camera, flatten.  Reference citations are paths under /root/reference/src. */
#include "scene.h"
#include "rt1w_internal.h"

#include <algorithm>
#include <cstring>
#include <new>

namespace rt1w {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

/* surrounding_box aabb.rs:35-52 (f64::min / f64::max) */
static AABB surrounding_box(const AABB& a, const AABB& b) {
    AABB r;
    r.minimum = rt_v3(rt_min(a.minimum.x, b.minimum.x), rt_min(a.minimum.y, b.minimum.y),
                      rt_min(a.minimum.z, b.minimum.z));
    r.maximum = rt_v3(rt_max(a.maximum.x, b.maximum.x), rt_max(a.maximum.y, b.maximum.y),
                      rt_max(a.maximum.z, b.maximum.z));
    return r;
}

static RtV3 msphere_center(const HostHittable& h, double time) { /* moving_sphere.rs:23-26 */
    RtV3 c0 = rt_v3(h.d[0], h.d[1], h.d[2]), c1 = rt_v3(h.d[3], h.d[4], h.d[5]);
    return c0 + ((time - h.d[6]) / (h.d[7] - h.d[6])) * (c1 - c0);
}

/* Hittable::bounding_box of every implementor */
static bool bounding_box(const rt1w_scene& s, int id, double time0, double time1, AABB& out) {
    const HostHittable& h = s.hittables[id];
    switch (h.kind) {
        case RT_SPHERE: { /* sphere.rs:65-70 */
            RtV3 c = rt_v3(h.d[0], h.d[1], h.d[2]), r = rt_v3(h.d[3], h.d[3], h.d[3]);
            out.minimum = c - r; out.maximum = c + r; return true;
        }
        case RT_MSPHERE: { /* moving_sphere.rs:72-84 */
            RtV3 r = rt_v3(h.d[8], h.d[8], h.d[8]);
            AABB b0{msphere_center(h, time0) - r, msphere_center(h, time0) + r};
            AABB b1{msphere_center(h, time1) - r, msphere_center(h, time1) + r};
            out = surrounding_box(b0, b1); return true;
        }
        case RT_XY: /* aarect.rs:74-79 */
            out.minimum = rt_v3(h.d[0], h.d[2], h.d[4] - 0.0001);
            out.maximum = rt_v3(h.d[1], h.d[3], h.d[4] + 0.0001); return true;
        case RT_XZ: /* aarect.rs:112-117 */
            out.minimum = rt_v3(h.d[0], h.d[4] - 0.0001, h.d[2]);
            out.maximum = rt_v3(h.d[1], h.d[4] + 0.0001, h.d[3]); return true;
        case RT_YZ: /* aarect.rs:180-185 */
            out.minimum = rt_v3(h.d[4] - 0.0001, h.d[0], h.d[2]);
            out.maximum = rt_v3(h.d[4] + 0.0001, h.d[1], h.d[3]); return true;
        case H_AABOX: out = h.box; return true; /* aabox.rs:98-103 */
        case H_BVH: out = h.box; return true;   /* bvh.rs:21-23 */
        case RT_TRANSLATE: { /* hittable.rs:228-233 */
            AABB c;
            if (!bounding_box(s, h.child, time0, time1, c)) return false;
            RtV3 off = rt_v3(h.d[0], h.d[1], h.d[2]);
            out.minimum = c.minimum + off; out.maximum = c.maximum + off; return true;
        }
        case RT_ROTATE_Y: /* hittable.rs:281-283 */
            if (!h.has_box) return false;
            out = h.box; return true;
        case RT_FLIP: return bounding_box(s, h.child, time0, time1, out);   /* hittable.rs:294-296 */
        case RT_MEDIUM: return bounding_box(s, h.child, time0, time1, out); /* constant_medium.rs:54-56 */
        default: return false;
    }
}

/* float-ord 0.3.1 FloatOrd: total order on the bit pattern */
static uint64_t float_ord_key(double x) {
    uint64_t u = rt_d2u(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

/* BVHNode::new bvh.rs:54-103.  Returns hittable id of the new node or <0. */
static int bvh_new(rt1w_scene& s, std::vector<int> objects, double time0, double time1) {
    size_t len = objects.size();
    if (len == 0) { set_error("objects mut not be empty (bvh.rs:61)"); return RT1W_ERR_INVALID; }
    HostHittable node;
    node.kind = H_BVH;
    node.has_box = true;
    node.d[0] = time0; node.d[1] = time1; /* kept for the opt-in rebuild (rt1w_scene_set_bvh_build) */
    if (len == 1) {
        int obj = objects.back();
        if (!bounding_box(s, obj, time0, time1, node.box)) {
            set_error("Bounding Box is required (bvh.rs:67)"); return RT1W_ERR_INVALID;
        }
        node.left = obj;
    } else if (len == 2) {
        int left = objects[1];  /* objects.pop(): the LAST element becomes left, bvh.rs:72 */
        int right = objects[0];
        AABB lb, rb;
        if (!bounding_box(s, left, time0, time1, lb) || !bounding_box(s, right, time0, time1, rb)) {
            set_error("Bounding Box is required (bvh.rs:74-75)"); return RT1W_ERR_INVALID;
        }
        node.box = surrounding_box(lb, rb);
        node.left = left; node.right = right;
    } else {
        uint32_t axis = rt_gen_below(s.rng, 3u); /* rng.gen_range(0..=2) bvh.rs:84 */
        std::vector<std::pair<uint64_t, int>> keyed;
        keyed.reserve(len);
        for (int o : objects) {
            AABB b;
            if (!bounding_box(s, o, time0, time1, b)) {
                set_error("Bounding Box is required (bvh.rs:86)"); return RT1W_ERR_INVALID;
            }
            keyed.push_back({float_ord_key(rt_get(b.minimum, (int)axis)), o});
        }
        /* sort_by_key is a stable sort */
        std::stable_sort(keyed.begin(), keyed.end(),
                         [](const std::pair<uint64_t, int>& a, const std::pair<uint64_t, int>& b) {
                             return a.first < b.first;
                         });
        std::vector<int> left, right;
        for (size_t i = 0; i < len; ++i) (i < len / 2 ? left : right).push_back(keyed[i].second);
        int l = bvh_new(s, left, time0, time1);
        if (l < 0) return l;
        int r = bvh_new(s, right, time0, time1);
        if (r < 0) return r;
        node.box = surrounding_box(s.hittables[l].box, s.hittables[r].box);
        node.left = l; node.right = r;
        s.hittables[l].used = s.hittables[r].used = true;
    }
    s.hittables.push_back(node);
    return (int)s.hittables.size() - 1;
}

/* ---- flatten ---------------------------------------------------------- */

struct Flattener {
    rt1w_scene& s;
    std::vector<RtNode>& out;
    uint32_t max_scope = 0;
    bool media = false;
    bool ok = true;
    int err = RT1W_OK;

    static RtNode blank(uint32_t kind) {
        RtNode n;
        std::memset(&n, 0, sizeof n);
        n.kind = kind; n.mat = 0; n.a = RT_NONE; n.b = RT_NONE; n.skip = RT_NONE; n.pad = 0;
        return n;
    }
    void fail(int code, const char* msg) { if (ok) { ok = false; err = code; set_error(msg); } }

    /* ---- opt-in SAH rebuild (the reference's build is the random-axis median split of bvh.rs:84-100) ----
     * Same leaf set, same box arithmetic (surrounding_box of the children), another tree: every split minimises
     * area(L)*|L| + area(R)*|R| over the three axes and all positions of the centroid order. */
    struct SahItem { int id; AABB box; RtV3 c; int32_t orig; };
    std::vector<int32_t>* topo = nullptr; /* where the rebuilt trees are written down (rt1w_scene_get_bvh_topology) */
    bool collect_leaves(int id, double time0, double time1, std::vector<SahItem>& items) {
        const HostHittable& h = s.hittables[id];
        if (h.kind == H_BVH) {
            if (!collect_leaves(h.left, time0, time1, items)) return false;
            return h.right < 0 || collect_leaves(h.right, time0, time1, items);
        }
        SahItem it;
        it.id = id;
        if (!bounding_box(s, id, time0, time1, it.box)) return false;
        it.c = 0.5 * (it.box.minimum + it.box.maximum);
        it.orig = (int32_t)items.size(); /* position in the reference tree's left-to-right leaf order */
        items.push_back(it);
        return true;
    }
    static double half_area(const AABB& b) {
        const RtV3 e = b.maximum - b.minimum;
        return e.x * e.y + e.y * e.z + e.z * e.x;
    }
    /* reorders items[lo, hi) and returns mid: left = [lo, mid), right = [mid, hi) */
    static size_t sah_split(std::vector<SahItem>& items, size_t lo, size_t hi) {
        const size_t n = hi - lo;
        double best = RT_INF;
        size_t best_i = n / 2;
        std::vector<SahItem> best_order;
        std::vector<SahItem> v(items.begin() + (long)lo, items.begin() + (long)hi);
        std::vector<double> right_area(n + 1, 0.0);
        for (int axis = 0; axis < 3; ++axis) {
            std::stable_sort(v.begin(), v.end(), [axis](const SahItem& a, const SahItem& b) { return rt_get(a.c, axis) < rt_get(b.c, axis); });
            AABB acc = v[n - 1].box;
            for (size_t i = n - 1; i >= 1; --i) {
                acc = (i == n - 1) ? v[i].box : surrounding_box(v[i].box, acc);
                right_area[i] = half_area(acc);
            }
            acc = v[0].box;
            for (size_t i = 1; i < n; ++i) {
                if (i > 1) acc = surrounding_box(acc, v[i - 1].box);
                const double cost = half_area(acc) * (double)i + right_area[i] * (double)(n - i);
                if (cost < best) { best = cost; best_i = i; best_order = v; }
            }
        }
        if (!best_order.empty()) std::copy(best_order.begin(), best_order.end(), items.begin() + (long)lo);
        return lo + best_i;
    }
    /* `lone`: this range is one object whose sibling is a subtree.  BVHNode::new never hangs a bare object next to a subtree: a
     * one-object half becomes BVHChild::One with the object's own box (bvh.rs:63-70), so that every object sits under a One or under
     * a two-object Two.  The rebuilt trees keep that shape (stream code -2): the object is tested behind its own box, as in the
     * reference's trees, and the phased pair walk (rt_walk_pair.h) finds every primitive in a primitive group. */
    uint32_t emit_sah(std::vector<SahItem>& items, size_t lo, size_t hi, uint32_t parent_scope, uint32_t scope_depth, bool in_boundary,
                      uint32_t* need, bool lone = false) {
        if (hi - lo == 1) {
            if (lone) {
                if (topo) topo->push_back(-2);
                const uint32_t idx = (uint32_t)out.size();
                RtNode n = blank(RT_BVH1);
                const AABB& box = items[lo].box;
                n.d[0] = box.minimum.x; n.d[1] = box.minimum.y; n.d[2] = box.minimum.z;
                n.d[3] = box.maximum.x; n.d[4] = box.maximum.y; n.d[5] = box.maximum.z;
                out.push_back(n);
                if (topo) topo->push_back(items[lo].orig);
                uint32_t na = 0;
                const uint32_t a = emit(items[lo].id, parent_scope, scope_depth, in_boundary, &na);
                out[idx].a = a;
                *need = std::max(2u, na);
                return idx;
            }
            if (topo) topo->push_back(items[lo].orig);
            return emit(items[lo].id, parent_scope, scope_depth, in_boundary, need);
        }
        if (topo) topo->push_back(-1);
        const size_t mid = sah_split(items, lo, hi);
        AABB lb = items[lo].box, rb = items[mid].box;
        for (size_t i = lo + 1; i < mid; ++i) lb = surrounding_box(lb, items[i].box);
        for (size_t i = mid + 1; i < hi; ++i) rb = surrounding_box(rb, items[i].box);
        const AABB box = surrounding_box(lb, rb);
        const uint32_t idx = (uint32_t)out.size();
        RtNode n = blank(RT_BVH2);
        n.d[0] = box.minimum.x; n.d[1] = box.minimum.y; n.d[2] = box.minimum.z;
        n.d[3] = box.maximum.x; n.d[4] = box.maximum.y; n.d[5] = box.maximum.z;
        out.push_back(n);
        uint32_t na = 0, nb = 0;
        const bool mixed = (mid - lo == 1) != (hi - mid == 1); /* one object next to a subtree */
        const uint32_t a = emit_sah(items, lo, mid, parent_scope, scope_depth, in_boundary, &na, mixed && mid - lo == 1);
        const uint32_t b = emit_sah(items, mid, hi, parent_scope, scope_depth, in_boundary, &nb, mixed && hi - mid == 1);
        out[idx].a = a; out[idx].b = b;
        *need = std::max(2u, std::max(1u + na, nb));
        return idx;
    }

    /* ---- opt-in RT1W_BVH_BEST_AXIS: `BVHNode::new` as written (bvh.rs:54-103) with ONE thing chosen instead of drawn ----
     * The reference draws the split axis of every node from an entropy-seeded generator (bvh.rs:84, main.rs:803), sorts the objects
     * stably by their box minimum on it (float-ord, bvh.rs:85-86) and splits at len/2 (bvh.rs:87); one and two objects get the shapes of
     * bvh.rs:63-79.  Every axis sequence is a tree some run of the reference builds.  Here the axis of a node is the one whose
     * median split has the lowest area(L)*|L| + area(R)*|R| (ties: the lower axis); everything else -- the sort key, the stable
     * order, the split position, left = last of two, the boxes (surrounding_box of the children) -- is the reference's rule on the
     * objects in the order the host handed them to this `BVHNode::new` call.  Nested `BVHNode::new` calls stay separate BVHs. */
    struct BaItem { int id; AABB box; int32_t orig; /* position in the call's own object list: the leaf's number in the topology stream */ };
    static AABB box_of(const std::vector<BaItem>& v, size_t lo, size_t hi) {
        AABB b = v[lo].box;
        for (size_t i = lo + 1; i < hi; ++i) b = surrounding_box(b, v[i].box);
        return b;
    }
    uint32_t emit_best_axis(std::vector<BaItem> objs, uint32_t parent_scope, uint32_t scope_depth, bool in_boundary, uint32_t* need, AABB* box_out) {
        const size_t len = objs.size();
        const uint32_t idx = (uint32_t)out.size();
        if (len == 1) { /* BVHChild::One, aabb = the object's own box (bvh.rs:63-70) */
            if (topo) { topo->push_back(-2); topo->push_back(objs[0].orig); }
            RtNode n = blank(RT_BVH1);
            const AABB& box = objs[0].box;
            n.d[0] = box.minimum.x; n.d[1] = box.minimum.y; n.d[2] = box.minimum.z;
            n.d[3] = box.maximum.x; n.d[4] = box.maximum.y; n.d[5] = box.maximum.z;
            out.push_back(n);
            uint32_t na = 0;
            const uint32_t a = emit(objs[0].id, parent_scope, scope_depth, in_boundary, &na);
            out[idx].a = a;
            *need = std::max(2u, na);
            *box_out = box;
            return idx;
        }
        if (topo) topo->push_back(-1);
        out.push_back(blank(RT_BVH2));
        uint32_t na = 0, nb = 0, a, b;
        AABB lb, rb;
        if (len == 2) { /* BVHChild::Two(objects.pop(), objects.pop()): the LAST object is the left child (bvh.rs:71-79) */
            lb = objs[1].box; rb = objs[0].box;
            if (topo) topo->push_back(objs[1].orig);
            a = emit(objs[1].id, parent_scope, scope_depth, in_boundary, &na);
            if (topo) topo->push_back(objs[0].orig);
            b = emit(objs[0].id, parent_scope, scope_depth, in_boundary, &nb);
        } else {
            std::vector<BaItem> best_order;
            double best = RT_INF;
            for (int axis = 0; axis < 3; ++axis) {
                std::vector<BaItem> v = objs;
                std::stable_sort(v.begin(), v.end(), [axis](const BaItem& x, const BaItem& y) {
                    return float_ord_key(rt_get(x.box.minimum, axis)) < float_ord_key(rt_get(y.box.minimum, axis));
                });
                const double cost = Flattener::half_area(box_of(v, 0, len / 2)) * (double)(len / 2) + Flattener::half_area(box_of(v, len / 2, len)) * (double)(len - len / 2);
                if (best_order.empty() || cost < best) { best = cost; best_order = v; }
            }
            std::vector<BaItem> l(best_order.begin(), best_order.begin() + (long)(len / 2)), r(best_order.begin() + (long)(len / 2), best_order.end());
            a = emit_best_axis(l, parent_scope, scope_depth, in_boundary, &na, &lb);
            b = emit_best_axis(r, parent_scope, scope_depth, in_boundary, &nb, &rb);
        }
        const AABB box = surrounding_box(lb, rb);
        RtNode& n = out[idx];
        n.d[0] = box.minimum.x; n.d[1] = box.minimum.y; n.d[2] = box.minimum.z;
        n.d[3] = box.maximum.x; n.d[4] = box.maximum.y; n.d[5] = box.maximum.z;
        n.a = a; n.b = b;
        *need = std::max(2u, std::max(1u + na, nb));
        *box_out = box;
        return idx;
    }

    /* returns node index; *need = stack entries in use beyond the popped entry of
     * this node while its subtree is processed */
    uint32_t emit(int id, uint32_t parent_scope, uint32_t scope_depth, bool in_boundary, uint32_t* need) {
        const HostHittable h = s.hittables[id];
        *need = 0;
        if (!ok) return RT_NONE;
        switch (h.kind) {
            case RT_SPHERE: case RT_MSPHERE: case RT_XY: case RT_XZ: case RT_YZ: {
                RtNode n = blank(h.kind);
                std::memcpy(n.d, h.d, sizeof n.d);
                n.e[0] = h.d[6]; n.e[1] = h.d[7]; n.e[2] = h.d[8];
                n.mat = (uint32_t)h.mat;
                n.b = parent_scope; /* innermost wrapper above the leaf */
                out.push_back(n);
                return (uint32_t)out.size() - 1;
            }
            case H_AABOX: /* AABox::hit delegates to the side BVH, aabox.rs:88-96 */
                return emit(h.child, parent_scope, scope_depth, in_boundary, need);
            case H_BVH: {
                if (s.bvh_build == RT1W_BVH_SAH) {
                    /* opt-in: the tree over this BVH's leaf set is rebuilt by surface-area heuristic (see emit_sah) */
                    std::vector<SahItem> items;
                    if (collect_leaves(id, h.d[0], h.d[1], items) && items.size() >= 2)
                        return emit_sah(items, 0, items.size(), parent_scope, scope_depth, in_boundary, need);
                }
                if (s.bvh_build == RT1W_BVH_BEST_AXIS) {
                    const auto call = s.bvh_calls.find(id);
                    if (call != s.bvh_calls.end() && call->second.size() >= 2) {
                        std::vector<BaItem> objs;
                        bool good = true;
                        for (size_t k = 0; k < call->second.size() && good; ++k) {
                            BaItem it;
                            it.id = call->second[k];
                            it.orig = (int32_t)k;
                            good = bounding_box(s, it.id, h.d[0], h.d[1], it.box);
                            objs.push_back(it);
                        }
                        AABB box;
                        if (good) return emit_best_axis(objs, parent_scope, scope_depth, in_boundary, need, &box);
                    }
                }
                uint32_t idx = (uint32_t)out.size();
                RtNode n = blank(h.right >= 0 ? RT_BVH2 : RT_BVH1);
                n.d[0] = h.box.minimum.x; n.d[1] = h.box.minimum.y; n.d[2] = h.box.minimum.z;
                n.d[3] = h.box.maximum.x; n.d[4] = h.box.maximum.y; n.d[5] = h.box.maximum.z;
                out.push_back(n);
                uint32_t na = 0, nb = 0;
                uint32_t a = emit(h.left, parent_scope, scope_depth, in_boundary, &na);
                uint32_t b = RT_NONE;
                if (h.right >= 0) b = emit(h.right, parent_scope, scope_depth, in_boundary, &nb);
                out[idx].a = a; out[idx].b = b;
                /* a BVHChild::One counts two slots like a Two: the walk writes both slots above the top whatever the node is
                 * (rt_walk_box's branchless push) */
                *need = (h.right >= 0) ? std::max(2u, std::max(1u + na, nb)) : std::max(2u, na);
                return idx;
            }
            case RT_TRANSLATE: case RT_ROTATE_Y: case RT_FLIP: {
                if (h.kind == RT_FLIP) { /* FlipFace(leaf): fold into the leaf record */
                    uint32_t ck = s.hittables[h.child].kind;
                    if (ck == RT_SPHERE || ck == RT_MSPHERE || ck == RT_XY || ck == RT_XZ || ck == RT_YZ) {
                        uint32_t idx = emit(h.child, parent_scope, scope_depth, in_boundary, need);
                        if (ok) out[idx].kind |= RT_LEAF_FLIPPED;
                        return idx;
                    }
                }
                if (scope_depth + 1 > RT_MAX_SCOPE_DEPTH) {
                    fail(RT1W_ERR_UNSUPPORTED, "more than RT_MAX_SCOPE_DEPTH nested Translate/RotateY/FlipFace wrappers");
                    return RT_NONE;
                }
                max_scope = std::max(max_scope, scope_depth + 1);
                uint32_t idx = (uint32_t)out.size();
                RtNode n = blank(h.kind);
                std::memcpy(n.d, h.d, sizeof n.d);
                n.b = parent_scope;
                out.push_back(n);
                uint32_t nc = 0;
                uint32_t c = emit(h.child, idx, scope_depth + 1, in_boundary, &nc);
                out[idx].a = c;
                *need = std::max(2u, 1u + nc);
                return idx;
            }
            case RT_MEDIUM: {
                if (in_boundary) {
                    fail(RT1W_ERR_UNSUPPORTED, "ConstantMedium inside a ConstantMedium boundary");
                    return RT_NONE;
                }
                media = true;
                uint32_t idx = (uint32_t)out.size();
                RtNode n = blank(RT_MEDIUM);
                n.d[0] = h.d[0];
                n.mat = (uint32_t)h.mat;
                n.b = parent_scope;
                out.push_back(n);
                uint32_t nc = 0;
                /* the boundary is traversed with the medium's own ray as its outer ray */
                uint32_t c = emit(h.child, RT_NONE, 0, true, &nc);
                out[idx].a = c;
                *need = 1u + nc;
                return idx;
            }
            default:
                fail(RT1W_ERR_INVALID, "unknown hittable kind");
                return RT_NONE;
        }
    }
};

static bool texture_needs_uv(const rt1w_scene& s, uint32_t tex) {
    const RtTexture& t = s.textures[tex];
    if (t.kind == RT_TEX_IMAGE) return true;
    if (t.kind == RT_TEX_CHECKER) return texture_needs_uv(s, t.a) || texture_needs_uv(s, t.b);
    return false;
}

} // namespace rt1w

using namespace rt1w;

/* ---- helpers for the ABI ---------------------------------------------- */

#define CHECK_SCENE(s)                                                       \
    if (!(s)) { set_error("null scene"); return RT1W_ERR_INVALID; }           \
    if ((s)->committed) { set_error("scene is committed (immutable)"); return RT1W_ERR_STATE; }

static bool valid_tex(const rt1w_scene* s, int t) { return t >= 0 && (size_t)t < s->textures.size(); }
static bool valid_mat(const rt1w_scene* s, int m) { return m >= 0 && (size_t)m < s->materials.size(); }
static int take_child(rt1w_scene* s, int id) {
    if (id < 0 || (size_t)id >= s->hittables.size()) { set_error("bad hittable id"); return RT1W_ERR_INVALID; }
    if (s->hittables[id].used) { set_error("hittable id already owned by a parent"); return RT1W_ERR_INVALID; }
    s->hittables[id].used = true;
    return RT1W_OK;
}
static int add_hittable(rt1w_scene* s, const HostHittable& h) {
    s->hittables.push_back(h);
    return (int)s->hittables.size() - 1;
}
static int add_material(rt1w_scene* s, uint32_t kind, uint32_t tex, double d0, double d1, double d2, double d3) {
    RtMaterial m;
    std::memset(&m, 0, sizeof m);
    m.kind = kind; m.tex = tex; m.d[0] = d0; m.d[1] = d1; m.d[2] = d2; m.d[3] = d3;
    s->materials.push_back(m);
    return (int)s->materials.size() - 1;
}

extern "C" {

const char* rt1w_last_error(void) { return g_error.c_str(); }
void rt1w_internal_set_error(const char* msg) { set_error(msg ? msg : ""); } /* rt1w_internal.h */
const char* rt1w_version(void) { return "rt1w-mi355x 0.1 (gfx950, f64, philox4x32-10)"; }

int rt1w_scene_create(uint64_t build_seed, rt1w_scene** out) {
    if (!out) { set_error("null out"); return RT1W_ERR_INVALID; }
    rt1w_scene* s = new (std::nothrow) rt1w_scene();
    if (!s) { set_error("out of memory"); return RT1W_ERR_NOMEM; }
    s->rng = rt_rng_build(build_seed);
    *out = s;
    return RT1W_OK;
}
void rt1w_scene_destroy(rt1w_scene* s) { delete s; }

int rt1w_scene_rng_f64(rt1w_scene* s, double* out) {
    CHECK_SCENE(s);
    if (!out) { set_error("null out"); return RT1W_ERR_INVALID; }
    *out = rt_gen_f64(s->rng);
    return RT1W_OK;
}
int rt1w_scene_rng_range(rt1w_scene* s, double low, double high, double* out) {
    CHECK_SCENE(s);
    if (!out || !(low < high)) { set_error("bad range"); return RT1W_ERR_INVALID; }
    *out = rt_gen_range(s->rng, low, high);
    return RT1W_OK;
}

/* ---- textures ---- */
int rt1w_texture_solid(rt1w_scene* s, const double rgb[3]) {
    CHECK_SCENE(s);
    if (!rgb) { set_error("null rgb"); return RT1W_ERR_INVALID; }
    RtTexture t; std::memset(&t, 0, sizeof t);
    t.kind = RT_TEX_SOLID; t.d[0] = rgb[0]; t.d[1] = rgb[1]; t.d[2] = rgb[2];
    s->textures.push_back(t);
    return (int)s->textures.size() - 1;
}
int rt1w_texture_checker(rt1w_scene* s, int odd, int even) {
    CHECK_SCENE(s);
    if (!valid_tex(s, odd) || !valid_tex(s, even)) { set_error("bad texture id"); return RT1W_ERR_INVALID; }
    RtTexture t; std::memset(&t, 0, sizeof t);
    t.kind = RT_TEX_CHECKER; t.a = (uint32_t)odd; t.b = (uint32_t)even;
    s->textures.push_back(t);
    return (int)s->textures.size() - 1;
}
int rt1w_texture_noise_tables(rt1w_scene* s, double scale, const double ranvec[768],
                              const uint32_t perm_x[256], const uint32_t perm_y[256],
                              const uint32_t perm_z[256]) {
    CHECK_SCENE(s);
    if (!ranvec || !perm_x || !perm_y || !perm_z) { set_error("null table"); return RT1W_ERR_INVALID; }
    for (int i = 0; i < 256; ++i)
        if (perm_x[i] > 255u || perm_y[i] > 255u || perm_z[i] > 255u) {
            set_error("perm entry out of range"); return RT1W_ERR_INVALID;
        }
    RtPerlin p;
    std::memcpy(p.ranvec, ranvec, sizeof p.ranvec);
    std::memcpy(p.perm_x, perm_x, sizeof p.perm_x);
    std::memcpy(p.perm_y, perm_y, sizeof p.perm_y);
    std::memcpy(p.perm_z, perm_z, sizeof p.perm_z);
    s->perlin.push_back(p);
    RtTexture t; std::memset(&t, 0, sizeof t);
    t.kind = RT_TEX_NOISE; t.d[0] = scale; t.a = (uint32_t)s->perlin.size() - 1;
    s->textures.push_back(t);
    return (int)s->textures.size() - 1;
}
int rt1w_texture_noise(rt1w_scene* s, double scale) {
    CHECK_SCENE(s);
    /* Perlin::new perlin.rs:25-43 */
    RtPerlin p;
    for (int i = 0; i < 256; ++i) {
        double x = rt_gen_range(s->rng, -1.0, 1.0);
        double y = rt_gen_range(s->rng, -1.0, 1.0);
        double z = rt_gen_range(s->rng, -1.0, 1.0);
        RtV3 v = rt_normalize(rt_v3(x, y, z));
        p.ranvec[i * 3 + 0] = v.x; p.ranvec[i * 3 + 1] = v.y; p.ranvec[i * 3 + 2] = v.z;
    }
    uint32_t* perms[3] = {p.perm_x, p.perm_y, p.perm_z};
    for (int a = 0; a < 3; ++a) { /* generate_perm perlin.rs:16-23; shuffle = Fisher-Yates from the end */
        uint32_t* q = perms[a];
        for (uint32_t i = 0; i < 256; ++i) q[i] = i;
        for (uint32_t i = 255; i >= 1; --i) {
            uint32_t j = rt_gen_below(s->rng, i + 1u);
            std::swap(q[i], q[j]);
        }
    }
    return rt1w_texture_noise_tables(s, scale, p.ranvec, p.perm_x, p.perm_y, p.perm_z);
}
int rt1w_texture_image(rt1w_scene* s, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    CHECK_SCENE(s);
    if (!rgb8 || width == 0 || height == 0) { set_error("bad image"); return RT1W_ERR_INVALID; }
    size_t bytes = (size_t)width * height * 3;
    if (s->images.size() + bytes > 0xFFFFFFF0ull) { set_error("image pool too large"); return RT1W_ERR_UNSUPPORTED; }
    RtTexture t; std::memset(&t, 0, sizeof t);
    t.kind = RT_TEX_IMAGE; t.a = width; t.b = height; t.c = (uint32_t)s->images.size();
    s->images.insert(s->images.end(), rgb8, rgb8 + bytes);
    while (s->images.size() % 16) s->images.push_back(0);
    s->textures.push_back(t);
    return (int)s->textures.size() - 1;
}

/* ---- materials ---- */
int rt1w_material_lambertian(rt1w_scene* s, int tex) {
    CHECK_SCENE(s);
    if (!valid_tex(s, tex)) { set_error("bad texture id"); return RT1W_ERR_INVALID; }
    return add_material(s, RT_MAT_LAMBERTIAN, (uint32_t)tex, 0, 0, 0, 0);
}
int rt1w_material_metal(rt1w_scene* s, const double albedo[3], double fuzz) {
    CHECK_SCENE(s);
    if (!albedo) { set_error("null albedo"); return RT1W_ERR_INVALID; }
    return add_material(s, RT_MAT_METAL, 0, albedo[0], albedo[1], albedo[2], fuzz);
}
int rt1w_material_dielectric(rt1w_scene* s, double ir) {
    CHECK_SCENE(s);
    return add_material(s, RT_MAT_DIELECTRIC, 0, ir, 0, 0, 0);
}
int rt1w_material_diffuse_light(rt1w_scene* s, int tex) {
    CHECK_SCENE(s);
    if (!valid_tex(s, tex)) { set_error("bad texture id"); return RT1W_ERR_INVALID; }
    return add_material(s, RT_MAT_DIFFUSE_LIGHT, (uint32_t)tex, 0, 0, 0, 0);
}
int rt1w_material_null(rt1w_scene* s) {
    CHECK_SCENE(s);
    return add_material(s, RT_MAT_NULL, 0, 0, 0, 0, 0);
}

/* ---- hittables ---- */
int rt1w_hittable_sphere(rt1w_scene* s, const double c[3], double radius, int material) {
    CHECK_SCENE(s);
    if (!c || !valid_mat(s, material)) { set_error("bad sphere argument"); return RT1W_ERR_INVALID; }
    HostHittable h; h.kind = RT_SPHERE; h.mat = material;
    h.d[0] = c[0]; h.d[1] = c[1]; h.d[2] = c[2]; h.d[3] = radius;
    return add_hittable(s, h);
}
int rt1w_hittable_moving_sphere(rt1w_scene* s, const double c0[3], const double c1[3], double time0,
                                double time1, double radius, int material) {
    CHECK_SCENE(s);
    if (!c0 || !c1 || !valid_mat(s, material)) { set_error("bad moving_sphere argument"); return RT1W_ERR_INVALID; }
    HostHittable h; h.kind = RT_MSPHERE; h.mat = material;
    h.d[0] = c0[0]; h.d[1] = c0[1]; h.d[2] = c0[2]; h.d[3] = c1[0]; h.d[4] = c1[1]; h.d[5] = c1[2];
    h.d[6] = time0; h.d[7] = time1; h.d[8] = radius;
    return add_hittable(s, h);
}
static int add_rect(rt1w_scene* s, uint32_t kind, double a0, double a1, double b0, double b1, double k, int material) {
    CHECK_SCENE(s);
    if (!valid_mat(s, material)) { set_error("bad material id"); return RT1W_ERR_INVALID; }
    HostHittable h; h.kind = kind; h.mat = material;
    h.d[0] = a0; h.d[1] = a1; h.d[2] = b0; h.d[3] = b1; h.d[4] = k;
    return add_hittable(s, h);
}
int rt1w_hittable_xy_rect(rt1w_scene* s, double x0, double x1, double y0, double y1, double k, int m) {
    return add_rect(s, RT_XY, x0, x1, y0, y1, k, m);
}
int rt1w_hittable_xz_rect(rt1w_scene* s, double x0, double x1, double z0, double z1, double k, int m) {
    return add_rect(s, RT_XZ, x0, x1, z0, z1, k, m);
}
int rt1w_hittable_yz_rect(rt1w_scene* s, double y0, double y1, double z0, double z1, double k, int m) {
    return add_rect(s, RT_YZ, y0, y1, z0, z1, k, m);
}
int rt1w_hittable_aabox(rt1w_scene* s, const double p0[3], const double p1[3], int material) {
    CHECK_SCENE(s);
    if (!p0 || !p1 || !valid_mat(s, material)) { set_error("bad aabox argument"); return RT1W_ERR_INVALID; }
    /* AABox::new aabox.rs:22-84: six sides in this order, then BVHNode::new(sides, 0.0, 1.0, rng) */
    std::vector<int> sides;
    sides.push_back(rt1w_hittable_xy_rect(s, p0[0], p1[0], p0[1], p1[1], p1[2], material));
    sides.push_back(rt1w_hittable_xy_rect(s, p0[0], p1[0], p0[1], p1[1], p0[2], material));
    sides.push_back(rt1w_hittable_xz_rect(s, p0[0], p1[0], p0[2], p1[2], p1[1], material));
    sides.push_back(rt1w_hittable_xz_rect(s, p0[0], p1[0], p0[2], p1[2], p0[1], material));
    sides.push_back(rt1w_hittable_yz_rect(s, p0[1], p1[1], p0[2], p1[2], p1[0], material));
    sides.push_back(rt1w_hittable_yz_rect(s, p0[1], p1[1], p0[2], p1[2], p0[0], material));
    for (int id : sides) s->hittables[id].used = true;
    int bvh = bvh_new(*s, sides, 0.0, 1.0);
    if (bvh < 0) return bvh;
    s->bvh_calls[bvh] = sides;
    s->hittables[bvh].used = true;
    HostHittable h; h.kind = H_AABOX; h.child = bvh; h.has_box = true;
    h.box.minimum = rt_v3(p0[0], p0[1], p0[2]); h.box.maximum = rt_v3(p1[0], p1[1], p1[2]);
    return add_hittable(s, h);
}
int rt1w_hittable_translate(rt1w_scene* s, int child, const double offset[3]) {
    CHECK_SCENE(s);
    if (!offset) { set_error("null offset"); return RT1W_ERR_INVALID; }
    int rc = take_child(s, child);
    if (rc < 0) return rc;
    HostHittable h; h.kind = RT_TRANSLATE; h.child = child;
    h.d[0] = offset[0]; h.d[1] = offset[1]; h.d[2] = offset[2];
    return add_hittable(s, h);
}
int rt1w_hittable_rotate_y(rt1w_scene* s, int child, double time0, double time1, double angle_deg) {
    CHECK_SCENE(s);
    int rc = take_child(s, child);
    if (rc < 0) return rc;
    /* RotateY::new hittable.rs:158-202; Deg -> Rad is deg * (PI/180) in cgmath */
    double radians = angle_deg * (RT_PI / 180.0);
    double sin_theta, cos_theta;
    rt_sincos(radians, sin_theta, cos_theta);
    HostHittable h; h.kind = RT_ROTATE_Y; h.child = child;
    h.d[0] = sin_theta; h.d[1] = cos_theta;
    AABB bbox;
    if (bounding_box(*s, child, time0, time1, bbox)) {
        RtV3 mn = rt_v3(RT_INF, RT_INF, RT_INF), mx = rt_v3(-RT_INF, -RT_INF, -RT_INF);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    double fi = (double)i, fj = (double)j, fk = (double)k;
                    double x = fi * bbox.maximum.x + (1.0 - fi) * bbox.minimum.x;
                    double y = fj * bbox.maximum.y + (1.0 - fj) * bbox.minimum.y;
                    double z = fk * bbox.maximum.z + (1.0 - fk) * bbox.minimum.z;
                    double newx = cos_theta * x + sin_theta * z;
                    double newz = -sin_theta * x + cos_theta * z;
                    mn = rt_v3(rt_min(mn.x, newx), rt_min(mn.y, y), rt_min(mn.z, newz));
                    mx = rt_v3(rt_max(mx.x, newx), rt_max(mx.y, y), rt_max(mx.z, newz));
                }
        h.box.minimum = mn; h.box.maximum = mx; h.has_box = true;
    }
    return add_hittable(s, h);
}
int rt1w_hittable_flip_face(rt1w_scene* s, int child) {
    CHECK_SCENE(s);
    int rc = take_child(s, child);
    if (rc < 0) return rc;
    HostHittable h; h.kind = RT_FLIP; h.child = child;
    return add_hittable(s, h);
}
int rt1w_hittable_constant_medium(rt1w_scene* s, int boundary, double density, int texture) {
    CHECK_SCENE(s);
    if (!valid_tex(s, texture)) { set_error("bad texture id"); return RT1W_ERR_INVALID; }
    int rc = take_child(s, boundary);
    if (rc < 0) return rc;
    /* ConstantMedium::new constant_medium.rs:22-28 */
    HostHittable h; h.kind = RT_MEDIUM; h.child = boundary;
    h.mat = add_material(s, RT_MAT_ISOTROPIC, (uint32_t)texture, 0, 0, 0, 0);
    h.d[0] = -1.0 / density;
    return add_hittable(s, h);
}
int rt1w_hittable_bvh(rt1w_scene* s, const int* children, uint32_t n, double time0, double time1) {
    CHECK_SCENE(s);
    if (n > 0 && !children) { set_error("null children"); return RT1W_ERR_INVALID; }
    std::vector<int> objs(children, children + n);
    for (int id : objs) {
        int rc = take_child(s, id);
        if (rc < 0) return rc;
    }
    const int root = bvh_new(*s, objs, time0, time1);
    if (root >= 0) s->bvh_calls[root] = objs;
    return root;
}

/* ---- scene-level ---- */
int rt1w_scene_set_world(rt1w_scene* s, int hittable) {
    CHECK_SCENE(s);
    int rc = take_child(s, hittable);
    if (rc < 0) return rc;
    s->world = hittable;
    return RT1W_OK;
}
int rt1w_scene_set_lights(rt1w_scene* s, const int* hittables, uint32_t n) {
    CHECK_SCENE(s);
    if (n > 0 && !hittables) { set_error("null lights"); return RT1W_ERR_INVALID; }
    std::vector<int> l(hittables, hittables + n);
    for (int id : l) {
        int rc = take_child(s, id);
        if (rc < 0) return rc;
    }
    s->lights = l;
    return RT1W_OK;
}
int rt1w_scene_set_background(rt1w_scene* s, const double rgb[3]) {
    CHECK_SCENE(s);
    if (!rgb) { set_error("null rgb"); return RT1W_ERR_INVALID; }
    s->background = rt_v3(rgb[0], rgb[1], rgb[2]);
    return RT1W_OK;
}
int rt1w_scene_set_camera(rt1w_scene* s, const double look_from[3], const double look_at[3],
                          const double vup[3], double vfov_deg, double aspect_ratio, double aperture,
                          double focus_dist, double time0, double time1) {
    CHECK_SCENE(s);
    if (!look_from || !look_at || !vup) { set_error("null camera vector"); return RT1W_ERR_INVALID; }
    if (!(time0 < time1)) { set_error("camera needs time0 < time1 (gen_range panics otherwise, camera.rs:71)"); return RT1W_ERR_INVALID; }
    /* Camera::new camera.rs:22-59 */
    RtV3 lf = rt_v3(look_from[0], look_from[1], look_from[2]);
    RtV3 la = rt_v3(look_at[0], look_at[1], look_at[2]);
    RtV3 up = rt_v3(vup[0], vup[1], vup[2]);
    double theta = vfov_deg * (RT_PI / 180.0);
    double h = rt_tan(theta / 2.0);
    double viewport_height = 2.0 * h;
    double viewport_width = aspect_ratio * viewport_height;
    RtCamera c;
    c.w = rt_normalize(lf - la);
    c.u = rt_normalize(rt_cross(up, c.w));
    c.v = rt_cross(c.w, c.u);
    c.origin = lf;
    c.horizontal = focus_dist * viewport_width * c.u;
    c.vertical = focus_dist * viewport_height * c.v;
    c.lower_left_corner = c.origin - c.horizontal / 2.0 - c.vertical / 2.0 - focus_dist * c.w;
    c.lens_radius = aperture / 2.0;
    c.time0 = time0; c.time1 = time1;
    s->camera = c; s->has_camera = true;
    return RT1W_OK;
}

static int flatten_scene(rt1w_scene* s);
int rt1w_scene_commit(rt1w_scene* s) {
    CHECK_SCENE(s);
    if (s->world < 0) { set_error("world not set"); return RT1W_ERR_STATE; }
    if (!s->has_camera) { set_error("camera not set"); return RT1W_ERR_STATE; }
    /* material flags */
    for (RtMaterial& m : s->materials) {
        uint32_t k = m.kind & 0xFFu;
        bool has_tex = (k == RT_MAT_LAMBERTIAN || k == RT_MAT_DIFFUSE_LIGHT || k == RT_MAT_ISOTROPIC);
        m.kind = k | ((has_tex && texture_needs_uv(*s, m.tex)) ? RT_MAT_NEEDS_UV : 0u);
        if (has_tex && s->textures[m.tex].kind == RT_TEX_SOLID) { /* SolidColor::value texture.rs:26-28: the colour, whatever u,v,p */
            const RtTexture& t = s->textures[m.tex];
            m.d[0] = t.d[0]; m.d[1] = t.d[1]; m.d[2] = t.d[2];
            m.kind |= RT_MAT_SOLID;
        }
    }
    if (s->materials.size() > 0xFFFFu) { set_error("more than 65535 materials"); return RT1W_ERR_UNSUPPORTED; }
    int rc = flatten_scene(s);
    if (rc < 0) return rc;
    s->committed = true;
    return RT1W_OK;
}

/* scene graph -> pre-order node records (and the light list); run by commit and again by rt1w_scene_set_bvh_build */
static int flatten_scene(rt1w_scene* s) {
    s->flat_nodes.clear();
    s->bvh_topology.clear();
    Flattener f{*s, s->flat_nodes};
    f.topo = &s->bvh_topology;
    uint32_t need = 0;
    uint32_t root = f.emit(s->world, RT_NONE, 0, false, &need);
    if (!f.ok) { s->flat_nodes.clear(); return f.err; }
    /* pre-order subtree ends (skip) for the sweep traversal */
    {
        std::vector<RtNode>& N = s->flat_nodes;
        struct R { static uint32_t fill(std::vector<RtNode>& N, uint32_t i) {
            uint32_t end = i + 1;
            uint32_t k = N[i].kind & RT_KIND_MASK;
            if (k == RT_BVH2) { end = fill(N, N[i].a); end = fill(N, N[i].b); }
            else if (k == RT_BVH1 || k == RT_TRANSLATE || k == RT_ROTATE_Y || k == RT_FLIP || k == RT_MEDIUM) end = fill(N, N[i].a);
            N[i].skip = end;
            return end;
        } };
        R::fill(N, root);
        for (uint32_t i = 0; i < N.size(); ++i) {
            uint32_t k = N[i].kind & RT_KIND_MASK;
            if (k == RT_BVH2 && (N[i].a != i + 1 || N[i].b != N[N[i].a].skip)) { set_error("internal: nodes not in pre-order"); return RT1W_ERR_INVALID; }
            if ((k == RT_BVH1 || (k >= RT_TRANSLATE && k <= RT_MEDIUM)) && N[i].a != i + 1) { set_error("internal: nodes not in pre-order"); return RT1W_ERR_INVALID; }
        }
    }
    for (RtNode& n : s->flat_nodes) {
        uint32_t k = n.kind & RT_KIND_MASK;
        if ((k >= RT_SPHERE && k <= RT_YZ) || k == RT_MEDIUM) n.mat = n.mat | (s->materials[n.mat].kind << 16);
    }
    s->flat_root = root;
    s->stack_need = 1u + need;
    s->scope_depth = f.max_scope;
    s->has_media = f.media;
    s->has_tex = false;
    for (const RtTexture& t : s->textures) if (t.kind != RT_TEX_SOLID) s->has_tex = true;
    s->has_msphere = false;
    for (const RtNode& n : s->flat_nodes) if ((n.kind & RT_KIND_MASK) == RT_MSPHERE) s->has_msphere = true;
    s->media_bare_spheres = s->has_media;
    for (size_t i = 0; i < s->flat_nodes.size(); ++i) /* a medium's boundary is the next node in pre-order (rt_flat.h) */
        if ((s->flat_nodes[i].kind & RT_KIND_MASK) == RT_MEDIUM && (i + 1 >= s->flat_nodes.size() || (s->flat_nodes[i + 1].kind & RT_KIND_MASK) != RT_SPHERE))
            s->media_bare_spheres = false;
    if (s->stack_need > RT_STACK_CAP) {
        set_error("scene needs a deeper traversal stack than RT_STACK_CAP");
        s->flat_nodes.clear();
        return RT1W_ERR_UNSUPPORTED;
    }
    /* lights: only XZRect and Sphere override pdf_value/random (aarect.rs:119-147,
     * sphere.rs:72-99); everything else keeps the trait defaults (hittable.rs:66-71) */
    s->flat_lights.clear();
    for (int id : s->lights) {
        const HostHittable& h = s->hittables[id];
        RtNode n = Flattener::blank(RT_DEFAULT);
        if (h.kind == RT_XZ || h.kind == RT_SPHERE) {
            n.kind = h.kind;
            std::memcpy(n.d, h.d, sizeof n.d);
            n.mat = (uint32_t)h.mat;
        }
        s->flat_lights.push_back(n);
    }
    return RT1W_OK;
}

/* ---- opt-in walk order ------------------------------------------------- */
namespace {
/* rough centre of a flat subtree in its parent's space: only used to pick which child is "near" */
RtV3 subtree_centre(const std::vector<RtNode>& N, uint32_t i) {
    const RtNode& n = N[i];
    switch (n.kind & RT_KIND_MASK) {
        case RT_BVH2: case RT_BVH1: return rt_v3(0.5 * (n.d[0] + n.d[3]), 0.5 * (n.d[1] + n.d[4]), 0.5 * (n.d[2] + n.d[5]));
        case RT_SPHERE: return rt_v3(n.d[0], n.d[1], n.d[2]);
        case RT_MSPHERE: return rt_v3(0.5 * (n.d[0] + n.d[3]), 0.5 * (n.d[1] + n.d[4]), 0.5 * (n.d[2] + n.d[5]));
        case RT_XY: return rt_v3(0.5 * (n.d[0] + n.d[1]), 0.5 * (n.d[2] + n.d[3]), n.d[4]);
        case RT_XZ: return rt_v3(0.5 * (n.d[0] + n.d[1]), n.d[4], 0.5 * (n.d[2] + n.d[3]));
        case RT_YZ: return rt_v3(n.d[4], 0.5 * (n.d[0] + n.d[1]), 0.5 * (n.d[2] + n.d[3]));
        case RT_TRANSLATE: { RtV3 c = subtree_centre(N, n.a); return rt_v3(c.x + n.d[0], c.y + n.d[1], c.z + n.d[2]); }
        case RT_ROTATE_Y: { RtV3 c = subtree_centre(N, n.a); return rt_v3(n.d[1] * c.x + n.d[0] * c.z, c.y, -n.d[0] * c.x + n.d[1] * c.z); }
        default: return subtree_centre(N, n.a); /* FlipFace, ConstantMedium: the child's */
    }
}
bool subtree_has(const std::vector<RtNode>& N, uint32_t i, uint32_t kind) {
    for (uint32_t j = i; j < N[i].skip; ++j) if ((N[j].kind & RT_KIND_MASK) == kind) return true;
    return false;
}
} // namespace

int rt1w_scene_set_walk_order(rt1w_scene* s, uint32_t mode) {
    if (!s) { set_error("null scene"); return RT1W_ERR_INVALID; }
    if (!s->committed) { set_error("scene not committed"); return RT1W_ERR_STATE; }
    if (mode > RT1W_WALK_NEAR_FAR_ALL) { set_error("unknown walk order"); return RT1W_ERR_INVALID; }
    std::vector<RtNode>& N = s->flat_nodes;
    uint32_t annotated = 0;
    const uint32_t clear = ~((uint32_t)(RT_BVH_ORDER_MASK << RT_BVH_ORDER_SHIFT) | (uint32_t)RT_BVH_LEFT_LOWER);
    for (uint32_t i = 0; i < N.size(); ++i) {
        if ((N[i].kind & RT_KIND_MASK) != RT_BVH2) continue;
        N[i].kind &= clear;
        /* a ConstantMedium draws a random number while it is visited (constant_medium.rs:85), with the closest hit so far as
         * its t_max: nodes with a medium below them keep the reference's order, so every medium is reached with the same
         * closest hit and the random stream is consumed identically */
        if (mode == RT1W_WALK_REFERENCE || subtree_has(N, i, RT_MEDIUM)) continue;
        /* a scattered ray carries time = hit t (main.rs:86,145), so MovingSphere::center extrapolates the sphere far outside
         * the bounding box the BVH holds for it (moving_sphere.rs:23-26,72-84): whether such a sphere is tested at all depends on
         * which boxes were pruned before -- on the order.  Measured on random_scene: 2 of 9600 pixels differ.  So the
         * result-preserving mode leaves subtrees with moving spheres alone too; RT1W_WALK_NEAR_FAR_ALL does not */
        if (mode == RT1W_WALK_NEAR_FAR && subtree_has(N, i, RT_MSPHERE)) continue;
        const RtV3 l = subtree_centre(N, N[i].a), r = subtree_centre(N, N[i].b);
        const double dx = rt_abs(l.x - r.x), dy = rt_abs(l.y - r.y), dz = rt_abs(l.z - r.z);
        uint32_t axis = dx >= dy ? (dx >= dz ? 0u : 2u) : (dy >= dz ? 1u : 2u);
        const double la = rt_get(l, (int)axis), ra = rt_get(r, (int)axis);
        if (!(la < ra) && !(la > ra)) continue; /* coincident (or NaN): nothing to order by */
        N[i].kind |= (axis + 1u) << RT_BVH_ORDER_SHIFT;
        if (la < ra) N[i].kind |= RT_BVH_LEFT_LOWER;
        ++annotated;
    }
    /* the order-aware kernel variant costs ~15 % by itself (random_scene 587 -> 493 Mpaths/s with 1 node in 5 annotated): it is
     * only chosen when at least a quarter of the BVH nodes carry an order -- leaving the others in the reference's order is
     * always allowed */
    uint32_t bvh2 = 0;
    for (const RtNode& n : N) bvh2 += (n.kind & RT_KIND_MASK) == RT_BVH2 ? 1u : 0u;
    s->walk_annotated = annotated * 4u >= bvh2 ? annotated : 0u;
    s->walk_order = mode;
    return RT1W_OK;
}

int rt1w_scene_set_bvh_build(rt1w_scene* s, uint32_t mode) {
    if (!s) { set_error("null scene"); return RT1W_ERR_INVALID; }
    if (!s->committed) { set_error("scene not committed"); return RT1W_ERR_STATE; }
    if (mode > RT1W_BVH_BEST_AXIS) { set_error("unknown BVH build"); return RT1W_ERR_INVALID; }
    const uint32_t before = s->bvh_build;
    s->bvh_build = mode;
    int rc = flatten_scene(s);
    if (rc < 0) { /* e.g. a tree deeper than the traversal stack: back to what it was */
        const std::string why = rt1w_last_error();
        s->bvh_build = before;
        if (flatten_scene(s) < 0) return RT1W_ERR_STATE;
        set_error(why);
        return rc;
    }
    return rt1w_scene_set_walk_order(s, s->walk_order); /* the order annotations live in the node records */
}

int64_t rt1w_scene_get_bvh_topology(const rt1w_scene* s, int32_t* out, uint64_t capacity) {
    if (!s) { set_error("null scene"); return RT1W_ERR_INVALID; }
    if (!s->committed) { set_error("scene not committed"); return RT1W_ERR_STATE; }
    const uint64_t n = s->bvh_topology.size();
    if (out) {
        if (capacity < n) { set_error("buffer too small"); return RT1W_ERR_INVALID; }
        if (n) std::memcpy(out, s->bvh_topology.data(), n * sizeof(int32_t));
    }
    return (int64_t)n;
}

int rt1w_scene_get_info(const rt1w_scene* s, rt1w_scene_info* out) {
    if (!s || !out) { set_error("null argument"); return RT1W_ERR_INVALID; }
    if (!s->committed) { set_error("scene not committed"); return RT1W_ERR_STATE; }
    out->n_nodes = (uint32_t)s->flat_nodes.size();
    out->n_lights = (uint32_t)s->flat_lights.size();
    out->n_materials = (uint32_t)s->materials.size();
    out->n_textures = (uint32_t)s->textures.size();
    out->n_perlin = (uint32_t)s->perlin.size();
    out->stack_need = s->stack_need;
    out->scope_depth = s->scope_depth;
    out->has_media = s->has_media ? 1u : 0u;
    out->has_textures = s->has_tex ? 1u : 0u;
    out->has_moving = s->has_msphere ? 1u : 0u;
    out->variant = (uint32_t)rt_pick_variant((uint32_t)s->flat_nodes.size(), s->has_media, s->has_tex, s->has_msphere, s->scope_depth, s->walk_annotated != 0u);
    out->bytes = s->flat_nodes.size() * sizeof(RtNode) + s->flat_lights.size() * sizeof(RtNode) +
                 s->materials.size() * sizeof(RtMaterial) + s->textures.size() * sizeof(RtTexture) +
                 s->perlin.size() * sizeof(RtPerlin) + s->images.size();
    return RT1W_OK;
}

uint32_t rt1w_default_chunk(uint32_t tile_w, uint32_t tile_h, uint32_t spp) {
    /* Work item = (pixel, chunk of samples).  The persistent kernel's tail is about half an item long, so:
     *  - aim for >= ~16M items (>= 80 per resident lane; measured on C3: 4M items 1764, 16M items 1790 Mpaths/s),
     *  - never more than 512 samples in an item (big frames at 10k spp: a whole pixel would be seconds of tail),
     *  - never split below 8 samples, and keep the chunk partial sums (24 B per item) under 8 GiB. */
    const uint64_t target_items = 16u << 20;
    uint64_t pixels = (uint64_t)tile_w * tile_h;
    if (pixels == 0 || spp == 0) return 1;
    uint64_t n_chunks = (target_items + pixels - 1) / pixels;
    const uint64_t by_len = ((uint64_t)spp + 511u) / 512u;
    if (n_chunks < by_len) n_chunks = by_len;
    const uint64_t by_mem = ((8ull << 30) / 24u) / pixels;
    if (n_chunks > by_mem) n_chunks = by_mem;
    if (n_chunks < 1) n_chunks = 1;
    if (n_chunks > spp) n_chunks = spp;
    uint32_t chunk = (uint32_t)((spp + n_chunks - 1) / n_chunks);
    if (chunk < 8u) chunk = spp < 8u ? spp : 8u;
    return chunk;
}

uint32_t rt1w_scene_default_chunk(const rt1w_scene* s, uint32_t tile_w, uint32_t tile_h, uint32_t spp) {
    if (!s || !s->committed) return 0u;
    const int variant = rt_pick_variant((uint32_t)s->flat_nodes.size(), s->has_media, s->has_tex, s->has_msphere, s->scope_depth, s->walk_annotated != 0u);
    return variant >= 2 ? 1u : rt1w_default_chunk(tile_w, tile_h, spp);
}

int64_t rt1w_scene_copy_flat(const rt1w_scene* s, int what, void* buf, uint64_t cap) {
    if (!s) { set_error("null scene"); return RT1W_ERR_INVALID; }
    if (!s->committed) { set_error("scene not committed"); return RT1W_ERR_STATE; }
    const void* src = nullptr;
    uint64_t bytes = 0;
    struct CamBg { RtCamera cam; RtV3 bg; uint32_t root, pad; } cb;
    switch (what) {
        case 0: src = s->flat_nodes.data(); bytes = s->flat_nodes.size() * sizeof(RtNode); break;
        case 1: src = s->flat_lights.data(); bytes = s->flat_lights.size() * sizeof(RtNode); break;
        case 2: src = s->materials.data(); bytes = s->materials.size() * sizeof(RtMaterial); break;
        case 3: src = s->textures.data(); bytes = s->textures.size() * sizeof(RtTexture); break;
        case 4: src = s->perlin.data(); bytes = s->perlin.size() * sizeof(RtPerlin); break;
        case 5: src = s->images.data(); bytes = s->images.size(); break;
        case 6:
            std::memset(&cb, 0, sizeof cb);
            cb.cam = s->camera; cb.bg = s->background; cb.root = s->flat_root;
            src = &cb; bytes = sizeof cb; break;
        default: set_error("bad selector"); return RT1W_ERR_INVALID;
    }
    if (!buf) return (int64_t)bytes;
    if (cap < bytes) { set_error("buffer too small"); return RT1W_ERR_INVALID; }
    if (bytes) std::memcpy(buf, src, bytes);
    return (int64_t)bytes;
}

} /* extern "C" */
