// tools/pairs_check.cpp -- the stack walk with pair records (rt_walk_w3.h: rt_walkp_step) against the one-entry-per-step walk on the CPU:
// every segment of every path walked both ways (t, primitive, scope and the generator state compared), steps per segment by kind.
// Measurement / debugging tool, not product.  Build: g++ -O2 -std=c++17 -ffp-contract=off -Iinclude -Iraytracing-1w_amd/csrc
// tools/pairs_check.cpp -o /tmp/pairs_check -Lraytracing-1w_amd -lrt1w -Wl,-rpath,$PWD/raytracing-1w_amd ; /tmp/pairs_check <arm> <W> <H> <spp> [sah]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
static unsigned long long g_hist_a[16], g_hist_b[16];
static unsigned long long* g_hist = g_hist_a;
#define RT_STAT_VISIT(kind) do { ++g_hist[(kind) & 15]; } while (0)
#include "rt_core.h"
#include "rt_pairs_build.h"
#include "rt1w.h"
struct HostStack { uint32_t e[96]; int sp = 0; void push(uint32_t v) { e[sp++] = v; } void poke(int above, uint32_t v) { if (sp + above >= 96) { printf("stack overflow\n"); exit(2); } e[sp + above] = v; } uint32_t pop() { return e[--sp]; } };
struct cam_bg { RtCamera cam; RtV3 bg; uint32_t root, pad; };
int main(int argc, char** argv) {
    int arm = atoi(argv[1]), W = atoi(argv[2]), H = atoi(argv[3]), spp = atoi(argv[4]);
    const bool sah = argc > 5;
    static const char* names[16] = {"bvh2", "bvh1", "sphere", "msphere", "xy", "xz", "yz", "translate", "rotate_y", "flip", "medium", "?", "?", "?", "?", "PAIR"};
    std::vector<uint8_t> earth(1024 * 512 * 3, 128);
    rt1w_scene* s = nullptr; uint32_t def[3];
    if (rt1w_scene_build_reference(arm, 1, (double)W / H, earth.data(), 1024, 512, &s, def)) { printf("fail\n"); return 1; }
    if (sah) rt1w_scene_set_bvh_build(s, RT1W_BVH_SAH);
    std::vector<std::vector<uint8_t>> a(7);
    for (int i = 0; i < 7; i++) { int64_t n = rt1w_scene_copy_flat(s, i, nullptr, 0); a[i].resize(n > 0 ? n + 96 : 16); rt1w_scene_copy_flat(s, i, a[i].data(), a[i].size()); }
    rt1w_scene_info inf; rt1w_scene_get_info(s, &inf);
    RtSceneView sc; memset(&sc, 0, sizeof sc);
    const cam_bg* cb = (const cam_bg*)a[6].data();
    sc.nodes = (const RtNode*)a[0].data(); sc.lights = (const RtNode*)a[1].data(); sc.materials = (const RtMaterial*)a[2].data(); sc.textures = (const RtTexture*)a[3].data();
    sc.perlin = (const RtPerlin*)a[4].data(); sc.images = a[5].data(); sc.root = cb->root; sc.n_nodes = inf.n_nodes; sc.n_lights = inf.n_lights; sc.n_materials = inf.n_materials; sc.n_textures = inf.n_textures;
    sc.camera = cb->cam; sc.background = cb->bg;
    std::vector<RtNode> patched(sc.nodes, sc.nodes + inf.n_nodes + 1);
    patched.resize(inf.n_nodes);
    std::vector<RtPairRec> pairs;
    const RtPairsInfo pi = rt_pairs_build(patched, pairs);
    patched.push_back(RtNode());
    RtSceneView sc2 = sc; sc2.nodes = patched.data();
    printf("arm %d%s nodes %u: BVH nodes %u, steering (pair records) %u, not inside their parent %u\n", arm, sah ? " SAH" : "", inf.n_nodes, pi.n_bvh, pi.n_steer, pi.n_not_inside);
    RtFrame f; memset(&f, 0, sizeof f); f.width = W; f.height = H; f.tile_w = W; f.tile_h = H; f.spp = spp; f.max_depth = 50; f.chunk = spp; f.n_chunks = 1;
    HostStack stk, stk2; RtGlobalNodes ns{sc.nodes}, ns2{sc2.nodes};
    unsigned long long segs = 0, bad = 0, max_sp = 0;
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) for (int k = 0; k < spp; k++) {
        RtPath p; rt_path_begin(sc, f, x, y, k, p);
        while (p.alive) {
            segs += p.depth_left != 0u;
            if (p.depth_left != 0u) {
                RtRng r2 = p.rng;
                RtWalk w;
                rt_walk_begin(w, sc2.root, p.ray, RT_R(0.001), RT_INF, stk2);
                g_hist = g_hist_b;
                while (!rt_walk_done(w, stk2)) { rt_walkp_step<RtCfgV3, true>(sc2, ns2, pairs.data(), w, r2, stk2); if ((unsigned long long)stk2.sp > max_sp) max_sp = stk2.sp; }
                g_hist = g_hist_a;
                RtRng r1 = p.rng;
                double t; uint32_t prim, scope;
                const bool found = rt_closest_hit<RtCfgV3>(sc, ns, p.ray, RT_R(0.001), RT_INF, r1, stk, t, prim, scope);
                const bool found2 = w.best_prim != RT_NONE;
                if (found != found2 || (found && (memcmp(&t, &w.best_t, 8) != 0 || prim != w.best_prim || scope != w.best_scope)) || memcmp(&r1, &r2, sizeof r1) != 0) {
                    if (bad < 5) printf("MISMATCH pixel %d %d sample %d: classic %d %.17g %u %u | pairs %d %.17g %u %u\n", x, y, k, (int)found, t, prim, scope, (int)found2, w.best_t, w.best_prim, w.best_scope);
                    ++bad;
                }
            }
            RtTrace tr = rt_path_trace<RtCfgV3>(sc, ns, p, stk);
            rt_path_shade<RtCfgV3>(sc, p, tr);
        }
    }
    for (int h = 0; h < 2; ++h) {
        const unsigned long long* g = h ? g_hist_b : g_hist_a;
        unsigned long long tot = 0; for (int i = 0; i < 16; i++) tot += g[i];
        /* the classic histogram counts every segment twice (the comparison walk and rt_path_trace) */
        const double den = h ? (double)segs : 2.0 * (double)segs;
        printf("%-22s %.2f steps/segment:", h ? "walk with pair records" : "one entry per step", (double)tot / den);
        for (int i = 0; i < 16; i++) if (g[i]) printf(" %s %.2f", names[i], (double)g[i] / den);
        printf("\n");
    }
    printf("segments %llu, mismatches %llu, deepest stack %llu\n", segs, bad, max_sp);
    rt1w_scene_destroy(s);
    return bad ? 1 : 0;
}
