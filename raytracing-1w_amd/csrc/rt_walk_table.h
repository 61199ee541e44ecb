/* rt_walk_table.h -- the WALK TABLE of a flattened scene: one 64-byte record per node for the stack-walk kernels that keep the most
 * visited nodes in LDS (rt_kernel_plain.h: rt_render_ss_body with HC > 0; rt_core.h: RtWalkNodes).  Host code, header-only: the
 * context builds the table once (context.hip) and the CPU test build of the core walks the same table (oracle/oracle_flat.cpp).
 *
 * Why.  A stack walk fetches a node per lane and step, by the lane's own index: four 16-byte lane-loads per visit, and the vector L1
 * answers about one divergent lane-load per clock and CU (profiles/r03_l1_gather_probe.txt) -- final_scene's walk runs at 0.69 of that
 * ceiling.  But the visits are concentrated: the 256 most visited of its 7760 nodes take 71 % of them (tools/visit_top.cpp), because every
 * ray meets the top of every BVH (bvh.rs:25-50 descends from the root).  Those records fit the LDS a workgroup has left (16 KB), where a
 * 64-byte read costs an eighth of the L1's time.
 *
 * What.  The table holds the nodes in WALK-ID order: first the most visited nodes -- by the visits the context counted in a small render of
 * the scene's own camera, or (no count: the CPU test build) by a best-first expansion from the root on the half-area of the bounding boxes --
 * then every other node in pre-order.  A record is the node's hot half with the fields a stack walk does
 * not use rewritten to name its neighbours by walk id (rt_core.h, rt_ns_child / rt_ns_leaf_id / rt_ns_wrap_id):
 *   BVH node   skip = walk id of the left / only child, b = walk id of the right child
 *   wrapper    skip = walk id of the child, mat = its own node index (scopes stay node indices), b = enclosing scope (unchanged)
 *   leaf       skip = its own node index (what the walk reports), b, mat unchanged
 *   medium     skip = its own node index, d[0] = neg_inv_density (unchanged); boundary a bare Sphere: kind |= RT_WT_INLINE_SPHERE,
 *              d[1..3] = centre, d[4] = radius (the two boundary tests need no second fetch); otherwise mat = walk id of the boundary's root
 * The walk visits the same nodes in the same order with the same operands whichever array it reads them from: same bits. */
#ifndef RT1W_WALK_TABLE_H
#define RT1W_WALK_TABLE_H

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <queue>
#include <string>
#include <vector>
#include "rt_core.h"

#ifndef RT_WT_CACHE_MAX
#define RT_WT_CACHE_MAX 256u /* records the kernels keep in LDS (16 KB) */
#endif

struct RtWalkTable {
    std::vector<RtNodeHot> rec;   /* [n_nodes], walk-id order */
    std::vector<uint32_t> id_of;  /* node index -> walk id */
    uint32_t n_first = 0;         /* records placed by priority (<= RT_WT_CACHE_MAX); the rest follow in pre-order */
};

/* false (with the reason) for a scene the table cannot describe; `root` = the scene's root node */
/* `visits`: measured visits per node (the context's count), or nullptr for the estimate by box area */
inline bool rt_walk_table_build(const std::vector<RtNode>& N, uint32_t root, const uint32_t* visits, RtWalkTable& T, std::string& why) {
    const uint32_t n = (uint32_t)N.size();
    T.rec.clear(); T.id_of.assign(n, RT_NONE); T.n_first = 0;
    if (n == 0 || root >= n) { why = "no nodes"; return false; }
    auto kind = [&](uint32_t i) { return N[i].kind & RT_KIND_MASK; };
    auto area = [&](uint32_t i) { const double ex = N[i].d[3] - N[i].d[0], ey = N[i].d[4] - N[i].d[1], ez = N[i].d[5] - N[i].d[2]; return ex * ey + ey * ez + ez * ex; };
    /* children a walk pushes from node i (media: the boundary subtree is walked from the medium's own step) */
    auto children = [&](uint32_t i, uint32_t out[2]) -> int {
        const uint32_t k = kind(i);
        if (k == RT_BVH2) { out[0] = i + 1u; out[1] = N[i].b; return 2; }
        if (k == RT_BVH1 || k == RT_TRANSLATE || k == RT_ROTATE_Y || k == RT_FLIP) { out[0] = i + 1u; return 1; }
        if (k == RT_MEDIUM && kind(i + 1u) != RT_SPHERE) { out[0] = i + 1u; return 1; }
        return 0;
    };
    uint32_t next_id = 0;
    if (visits) {
        /* 1a. by measured visits (a bare-Sphere boundary is never fetched by the table's walk: its medium carries it), the root first */
        std::vector<uint32_t> order;
        for (uint32_t i = 0; i < n; ++i)
            if (i != root && visits[i] > 0u && !(i > 0u && kind(i - 1u) == RT_MEDIUM && kind(i) == RT_SPHERE)) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return visits[a] > visits[b]; });
        T.id_of[root] = next_id++;
        for (size_t q = 0; q < order.size() && next_id < RT_WT_CACHE_MAX; ++q) T.id_of[order[q]] = next_id++;
    } else {
    /* 1b. best-first from the root: (priority, order of discovery) */
    struct Item { double p; uint32_t seq, node; };
    struct Less { bool operator()(const Item& a, const Item& b) const { return a.p < b.p || (a.p == b.p && a.seq > b.seq); } };
    std::priority_queue<Item, std::vector<Item>, Less> heap;
    uint32_t seq = 0;
    heap.push({RT_INF, seq++, root});
    while (!heap.empty() && next_id < RT_WT_CACHE_MAX) {
        const Item it = heap.top(); heap.pop();
        if (it.node >= n || T.id_of[it.node] != RT_NONE) { why = "a node is reached twice"; return false; }
        T.id_of[it.node] = next_id++;
        uint32_t ch[2];
        const int nc = children(it.node, ch);
        for (int c = 0; c < nc; ++c) {
            if (ch[c] >= n) { why = "a child index is out of range"; return false; }
            double p = it.p;
            if (kind(ch[c]) <= RT_BVH1) { const double a = area(ch[c]); p = (a == a && a < p) ? a : p; } /* a BVH node is met when its own box is; anything else whenever its parent is */
            heap.push({p, seq++, ch[c]});
        }
    }
    }
    T.n_first = next_id;
    /* 2. everything else in pre-order */
    for (uint32_t i = 0; i < n; ++i)
        if (T.id_of[i] == RT_NONE) T.id_of[i] = next_id++;
    /* 3. the records */
    T.rec.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        RtNodeHot r;
        std::memcpy(&r, &N[i], sizeof r);
        const uint32_t k = kind(i);
        if (k <= RT_BVH1) {
            if (i + 1u >= n || (k == RT_BVH2 && N[i].b >= n)) { why = "a child index is out of range"; return false; }
            r.skip = T.id_of[i + 1u];
            r.b = k == RT_BVH2 ? T.id_of[N[i].b] : RT_NONE;
        } else if (k <= RT_YZ) {
            r.skip = i;
        } else if (k <= RT_FLIP) {
            if (i + 1u >= n) { why = "a wrapper without a child"; return false; }
            r.skip = T.id_of[i + 1u];
            r.mat = i;
        } else if (k == RT_MEDIUM) {
            if (i + 1u >= n) { why = "a medium without a boundary"; return false; }
            r.skip = i;
            if (kind(i + 1u) == RT_SPHERE) {
                r.kind |= RT_WT_INLINE_SPHERE;
                r.d[1] = N[i + 1u].d[0]; r.d[2] = N[i + 1u].d[1]; r.d[3] = N[i + 1u].d[2]; r.d[4] = N[i + 1u].d[3];
            } else {
                r.mat = T.id_of[i + 1u];
            }
        } else {
            why = "a node kind the walk table does not know"; return false;
        }
        T.rec[T.id_of[i]] = r;
    }
    return true;
}

#endif
