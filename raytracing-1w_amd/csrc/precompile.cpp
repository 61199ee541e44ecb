/* rt1w_precompile -- build step: compiles the topology-specialised kernels of the reference's own scene
 * arms (src/main.rs:815-936, build_seed as given) into <package>/kernels, so that the scenes everybody
 * renders never meet the run-time compiler.  No GPU needed (hiprtc cross-compiles for gfx950).
 *   rt1w_precompile <out_dir> [build_seed ...]
 */
#include <dirent.h>

#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>

#include "jit.h"

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: rt1w_precompile <out_dir> [build_seed ...]\n"); return 2; }
    const std::string dir = argv[1];
    int failures = 0;
    std::set<std::string> keep;
    for (int a = 2; a < (argc > 2 ? argc : 3); ++a) {
        unsigned long long seed = argc > 2 ? std::strtoull(argv[a], nullptr, 10) : 1ull;
        for (int arm = 0; arm < 8; ++arm) {
            const double aspect = (arm == 5 || arm == 6 || arm == 7) ? 1.0 : 16.0 / 9.0; /* main.rs:798,868,896,917 */
            rt1w_scene* s = nullptr;
            uint32_t defaults[3];
            static unsigned char dummy_earth[2 * 2 * 3] = {0};
            if (rt1w_scene_build_reference(arm, seed, aspect, dummy_earth, 2, 2, &s, defaults) < 0) {
                std::fprintf(stderr, "arm %d: %s\n", arm, rt1w_last_error()); ++failures; continue;
            }
            if (rt1w::jit_eligible(*s)) {
                rt1w::JitInfo info;
                int rc = rt1w::jit_precompile_to(*s, dir, info);
                if (rc < 0) { std::fprintf(stderr, "arm %d seed %llu: %s\n", arm, seed, info.message.c_str()); ++failures; }
                else keep.insert(info.path.substr(info.path.rfind('/') + 1));
                if (rc >= 0) std::printf("arm %d seed %llu: %zu nodes -> %s (%s, %.1f s)\n", arm, seed, s->flat_nodes.size(), info.path.c_str(),
                                 info.from_cache ? "present" : "compiled", info.compile_ms / 1e3);
                if (arm == 5 || arm == 6) { /* the Cornell arms also in single precision (RT1W_PRECISION_F32) */
                    rt1w::JitInfo i32;
                    int r32 = rt1w::jit_precompile_to(*s, dir, i32, true);
                    if (r32 < 0) { std::fprintf(stderr, "arm %d seed %llu (f32): %s\n", arm, seed, i32.message.c_str()); ++failures; }
                    else { keep.insert(i32.path.substr(i32.path.rfind('/') + 1));
                           std::printf("arm %d seed %llu f32: -> %s (%s, %.1f s)\n", arm, seed, i32.path.c_str(), i32.from_cache ? "present" : "compiled", i32.compile_ms / 1e3); }
                }
            }
            rt1w_scene_destroy(s);
        }
    }
    /* kernels of earlier builds (other header text = other key) are dead weight: drop them */
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) {
            std::string n = e->d_name;
            if (n.rfind("sweep_", 0) == 0 && n.size() > 6 && n.substr(n.size() - 6) == ".hsaco" && !keep.count(n)) std::remove((dir + "/" + n).c_str());
        }
        closedir(d);
    }
    return failures ? 1 : 0;
}
