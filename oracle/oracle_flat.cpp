/* oracle_flat.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * The device kernel's core (raytracing-1w_amd/csrc/rt_core.h) compiled for the
 * host with g++, driven by the same work decomposition as the HIP kernel
 * (context.hip: chunk partial sums, then resolve).  Purpose:
 *   - lets the CPU-only test tier check the flattener + iterative integrator
 *     against the literal recursive oracle (oracle.cpp) without a GPU;
 *   - is what the GPU result must equal BIT FOR BIT (same text, same
 *     arithmetic, -ffp-contract=off on both sides);
 *   - can be built with -fsanitize=address,undefined (GPU sanitizers are not
 *     available on the pool).
 * librt1w.so does not contain or call this; it has no CPU render path.
 */
#include <atomic>
#include <thread>
#include <vector>
#include <cstring>

#include "rt_core.h"

/* tests/test_static_sweep.py compiles this file a second time with a generated header that defines scene topologies
 * as compile-time arrays (what the library's run-time compiler does for the GPU, jit.cpp) and adds them as variants
 * 100, 101, ...: the unrolled sweep (rt_sweep_static) against the generic one, on the CPU. */
#ifdef ORC_STATIC_TOPO_H
#include ORC_STATIC_TOPO_H
#endif

namespace {
struct HostStack {
    uint32_t e[RT_STACK_CAP + 8];
    int sp = 0;
    int max_sp = 0;
    bool overflow = false;
    void push(uint32_t v) {
        if (sp >= RT_STACK_CAP) { overflow = true; return; }
        e[sp++] = v;
        if (sp > max_sp) max_sp = sp;
    }
    /* write without moving the top (rt_walk_box's branchless push): the touched slot counts towards the high-water mark and
     * must lie inside the stack like a push */
    void poke(int above, uint32_t v) {
        if (sp + above >= RT_STACK_CAP) { overflow = true; return; }
        e[sp + above] = v;
        if (sp + above + 1 > max_sp) max_sp = sp + above + 1;
    }
    uint32_t pop() { return e[--sp]; }
};
}

template <class Cfg>
static void run_path(const RtSceneView& sc, const RtFrame& f, uint32_t px, uint32_t py, uint32_t s, HostStack& stk,
                     RtV3& sum, uint64_t& segs, RtPath& path) { /* `path` lives as long as the pixel: the reference-stream build
                                                                   (-DRT_RNG_REFSTREAM) draws all samples of a pixel from one stream */
    rt_path_begin(sc, f, f.x0 + px, rt_frame_row(f, py), f.sample_offset + s, path);
    while (path.alive) {
        segs += path.depth_left != 0u ? 1u : 0u;
        RtGlobalNodes ns{sc.nodes};
        rt_path_step<Cfg>(sc, ns, path, stk);
    }
    sum = sum + path.radiance;
}

extern "C" {

struct orcflat_cam_bg { RtCamera cam; RtV3 bg; uint32_t root, pad; };

/* arrays are the bytes of rt1w_scene_copy_flat selectors 0..6 */
int orcflat_render(const void* nodes, uint32_t n_nodes, const void* lights, uint32_t n_lights, const void* materials,
                   uint32_t n_materials, const void* textures, uint32_t n_textures, const void* perlin, const void* images,
                   const void* cam_bg, const RtFrame* frame, int variant, int out_sum, int threads, double* out,
                   uint64_t* segments_out, uint32_t* max_stack_out) {
    RtSceneView sc;
    std::memset(&sc, 0, sizeof sc);
    const orcflat_cam_bg* cb = (const orcflat_cam_bg*)cam_bg;
    sc.nodes = (const RtNode*)nodes; sc.lights = (const RtNode*)lights;
    sc.materials = (const RtMaterial*)materials; sc.textures = (const RtTexture*)textures;
    sc.perlin = (const RtPerlin*)perlin; sc.images = (const uint8_t*)images;
    sc.root = cb->root; sc.n_nodes = n_nodes; sc.n_lights = n_lights; sc.n_materials = n_materials; sc.n_textures = n_textures;
    sc.camera = cb->cam; sc.background = cb->bg;
    RtFrame f = *frame;
    if (f.chunk == 0 || f.chunk > f.spp) f.chunk = f.spp;
    f.n_chunks = (f.spp + f.chunk - 1u) / f.chunk;
    const uint64_t npix = (uint64_t)f.tile_w * f.tile_h;
    std::atomic<uint64_t> next(0), seg_total(0);
    std::atomic<uint32_t> max_stack(0);
    std::atomic<int> bad(0);
    auto worker = [&]() {
        HostStack stk;
        uint64_t segs = 0;
        for (;;) {
            uint64_t p = next.fetch_add(1);
            if (p >= npix) break;
            uint32_t px = (uint32_t)(p % f.tile_w), py = (uint32_t)(p / f.tile_w);
            RtV3 total = rt_v3(0.0, 0.0, 0.0);
            RtPath path;
            for (uint32_t c = 0; c < f.n_chunks; ++c) {
                uint32_t s = c * f.chunk;
                uint32_t s_end = s + f.chunk < f.spp ? s + f.chunk : f.spp;
                RtV3 sum = rt_v3(0.0, 0.0, 0.0);
                for (; s < s_end; ++s) {
                    switch (variant) { /* the same feature-specialised variants the GPU library builds */
                        case 0: run_path<RtCfgV0>(sc, f, px, py, s, stk, sum, segs, path); break;
                        case 1: run_path<RtCfgV1>(sc, f, px, py, s, stk, sum, segs, path); break;
                        case 2: run_path<RtCfgV2>(sc, f, px, py, s, stk, sum, segs, path); break;
                        case 4: run_path<RtCfgV4>(sc, f, px, py, s, stk, sum, segs, path); break;
                        case 5: run_path<RtCfgV5>(sc, f, px, py, s, stk, sum, segs, path); break;
#ifdef ORC_STATIC_CASES
                        ORC_STATIC_CASES
#endif
                        default: run_path<RtCfgV3>(sc, f, px, py, s, stk, sum, segs, path); break;
                    }
                }
                total = total + sum;
            }
            if (!out_sum) total = rt_into_sampled(total, f.spp);
            out[p * 3 + 0] = total.x; out[p * 3 + 1] = total.y; out[p * 3 + 2] = total.z;
        }
        seg_total += segs;
        uint32_t m = (uint32_t)stk.max_sp, cur = max_stack.load();
        while (m > cur && !max_stack.compare_exchange_weak(cur, m)) {}
        if (stk.overflow) bad = 1;
    };
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    if (segments_out) *segments_out = seg_total.load();
    if (max_stack_out) *max_stack_out = max_stack.load();
    return bad.load() ? -2 : 0;
}

/* 1: built with -DRT_RNG_REFSTREAM (the reference's ChaCha12 stream per pixel; include/rt1w_num.h) */
int orcflat_is_refstream(void) {
#ifdef RT_RNG_REFSTREAM
    return 1;
#else
    return 0;
#endif
}
/* number of compile-time topologies built in as variants 100.. (0 in the ordinary build) */
int orcflat_n_static(void) {
#ifdef ORC_N_STATIC
    return ORC_N_STATIC;
#else
    return 0;
#endif
}
uint64_t orcflat_item_count(const RtFrame* f) { return rt_item_count(*f); }
void orcflat_item_decode(const RtFrame* f, uint64_t item, uint32_t out[3]) { rt_item_decode(*f, item, out[0], out[1], out[2]); }
uint32_t orcflat_sizeof(int what) {
    switch (what) { case 0: return sizeof(RtNode); case 1: return sizeof(RtMaterial); case 2: return sizeof(RtTexture);
                    case 3: return sizeof(RtPerlin); case 4: return sizeof(RtCamera); case 5: return sizeof(RtFrame); default: return 0; }
}

/* rt_rng_mark / rt_rng_rewind (include/rt1w_num.h): after `before` mixed draws the stream is marked, `n` words are drawn into
 * out_a, the stream is rewound to the mark and the same `n` words drawn into out_b: they must be equal.  Bit i of `pattern`
 * picks a 32-bit (0) or 64-bit (1) draw for draw i (mod 32).  Built for the Philox streams and, in liborc_flat_ref.so, for
 * the reference's ChaCha12 stream. */
void orcflat_rng_rewind_check(uint64_t seed, uint32_t pattern, uint32_t before, uint32_t n, uint64_t* out_a, uint64_t* out_b) {
    RtRng r = rt_rng_pixel_sample(seed, 3u, 7u);
    for (uint32_t i = 0; i < before; ++i) { if ((pattern >> (i & 31u)) & 1u) (void)rt_next_u64(r); else (void)rt_next_u32(r); }
    const RtRngMark m = rt_rng_mark(r);
    for (uint32_t i = 0; i < n; ++i) out_a[i] = ((pattern >> ((i + before) & 31u)) & 1u) ? rt_next_u64(r) : (uint64_t)rt_next_u32(r);
    rt_rng_rewind(r, m);
    for (uint32_t i = 0; i < n; ++i) out_b[i] = ((pattern >> ((i + before) & 31u)) & 1u) ? rt_next_u64(r) : (uint64_t)rt_next_u32(r);
}

} /* extern "C" */
