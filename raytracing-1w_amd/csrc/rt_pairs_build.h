/* rt_pairs_build.h -- host side of the stack walk with pair records (rt_walk_w3.h: RtPairRec, rt_walkp_step): which BVH nodes only
 * steer, their records, and the node array patched for that walk.  Per context; the scene's own flat arrays are not touched. */
#ifndef RT1W_PAIRS_BUILD_H
#define RT1W_PAIRS_BUILD_H
#include <cmath>
#include <cstring>
#include <vector>
#include "rt_core.h"
#include "rt_walk_w3.h"

struct RtPairsInfo {
    uint32_t n_bvh = 0, n_steer = 0, n_not_inside = 0;
};
/* `nodes` is a COPY of the flat node array: the `mat` word of every steering BVH node becomes 1 + its record's index (0 stays
 * "none": the flattener writes 0 there, scene.cpp).  Steering = BVHChild::Two / One (bvh.rs:9-12) whose children are all BVH
 * nodes with boxes inside the node's own box (BVHNode::new builds the box with surrounding_box, bvh.rs:96-101, so that holds
 * unless a child reports another box than the one it tests -- AABox does: aabox.rs:98-103 against aarect.rs:74-79). */
inline RtPairsInfo rt_pairs_build(std::vector<RtNode>& nodes, std::vector<RtPairRec>& pairs) {
    RtPairsInfo info;
    pairs.clear();
    const uint32_t n = (uint32_t)nodes.size();
    auto kind = [&](uint32_t i) { return nodes[i].kind & RT_KIND_MASK; };
    auto is_bvh = [&](uint32_t i) { return i < n && kind(i) <= RT_BVH1; };
    auto inside = [&](uint32_t c, uint32_t p) {
        for (int q = 0; q < 3; ++q)
            if (!(nodes[c].d[q] >= nodes[p].d[q]) || !(nodes[c].d[q + 3] <= nodes[p].d[q + 3])) return false;
        return true;
    };
    auto outward = [](double x, bool is_min) {
        float f = (float)x;
        if (is_min ? ((double)f > x) : ((double)f < x)) f = std::nextafterf(f, is_min ? -INFINITY : INFINITY);
        return f;
    };
    std::vector<uint32_t> rec_of(n, RT_NONE);
    for (uint32_t i = 0; i < n; ++i) {
        if (!is_bvh(i)) continue;
        ++info.n_bvh;
        const bool two = kind(i) == RT_BVH2;
        const uint32_t a = i + 1u, b = two ? nodes[i].b : RT_NONE;
        if (!is_bvh(a) || (two && !is_bvh(b))) continue;
        if (!inside(a, i) || (two && !inside(b, i))) { ++info.n_not_inside; continue; }
        rec_of[i] = (uint32_t)pairs.size();
        pairs.push_back(RtPairRec());
        ++info.n_steer;
    }
    for (uint32_t i = 0; i < n; ++i) {
        if (rec_of[i] == RT_NONE) continue;
        RtPairRec& P = pairs[rec_of[i]];
        std::memset(&P, 0, sizeof P);
        const bool two = kind(i) == RT_BVH2;
        const uint32_t a = i + 1u, b = two ? nodes[i].b : RT_NONE;
        for (int q = 0; q < 6; ++q) {
            P.lb[q] = outward(nodes[a].d[q], q < 3);
            P.rb[q] = two ? outward(nodes[b].d[q], q < 3) : (q < 3 ? INFINITY : -INFINITY);
        }
        P.l = rec_of[a] != RT_NONE ? (RT_PAIR_FLAG | rec_of[a]) : a;
        P.r = two ? (rec_of[b] != RT_NONE ? (RT_PAIR_FLAG | rec_of[b]) : b) : RT_NONE;
        nodes[i].mat = rec_of[i] + 1u;
    }
    return info;
}
#endif
