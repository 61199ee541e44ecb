/* rt1w -- command-line front end over the C ABI: what the reference's `main` does
 * (src/main.rs:797-1010: pick a scene arm, render, print the P3 image on stdout, progress on
 * stderr), with the pixel loop running on the GPU.  The reference has no flags (the arm is
 * the literal `match 5`, main.rs:815); here the literals are options with the reference's
 * values as defaults.
 *   rt1w [--scene N] [--width W] [--height H] [--spp S] [--depth D] [--seed G] [--build-seed B]
 *        [--device I] [--earth file.rgb8 W H] [--out file.ppm] [--specialise | --generic]
 *        [--reference-stream] [--f32] [--near-far] [--sah]
 * --reference-stream draws from the reference's own StdRng per pixel (RT1W_RNG_REFERENCE): `rt1w --reference-stream` prints
 * what `cargo run` of the reference prints, byte for byte (Cornell arm, 600x600, 100 spp).  --f32: RT1W_PRECISION_F32.
 * --near-far: rt1w_scene_set_walk_order(RT1W_WALK_NEAR_FAR).  --sah: rt1w_scene_set_bvh_build(RT1W_BVH_SAH).
 * --specialise compiles the kernel for this scene's topology now if the kernel cache has none (rt1w_context_specialise;
 * by default only a cached kernel is used, and renders of >= 2^35 paths compile on their own); --generic forbids it.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt1w.h"

static int fail(const char* what) {
    std::fprintf(stderr, "rt1w: %s: %s\n", what, rt1w_last_error());
    return 1;
}

int main(int argc, char** argv) {
    int arm = 5, device = 0;
    bool specialise = false, generic = false, reference_stream = false, f32 = false, near_far = false, sah = false;
    long width = -1, height = -1, spp = -1, depth = 50; /* MAX_DEPTH main.rs:801 */
    unsigned long long build_seed = 1, seed = 0;
    std::string out_path, earth_path;
    unsigned earth_w = 0, earth_h = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "rt1w: %s needs a value\n", name); std::exit(2); }
            return argv[++i];
        };
        if (a == "--scene") arm = std::atoi(next("--scene"));
        else if (a == "--width") width = std::atol(next("--width"));
        else if (a == "--height") height = std::atol(next("--height"));
        else if (a == "--spp") spp = std::atol(next("--spp"));
        else if (a == "--depth") depth = std::atol(next("--depth"));
        else if (a == "--seed") seed = std::strtoull(next("--seed"), nullptr, 10);
        else if (a == "--build-seed") build_seed = std::strtoull(next("--build-seed"), nullptr, 10);
        else if (a == "--device") device = std::atoi(next("--device"));
        else if (a == "--out") out_path = next("--out");
        else if (a == "--specialise") specialise = true;
        else if (a == "--generic") generic = true;
        else if (a == "--reference-stream") reference_stream = true;
        else if (a == "--f32") f32 = true;
        else if (a == "--near-far") near_far = true;
        else if (a == "--sah") sah = true;
        else if (a == "--earth") { earth_path = next("--earth"); earth_w = (unsigned)std::atoi(next("--earth W")); earth_h = (unsigned)std::atoi(next("--earth H")); }
        else { std::fprintf(stderr, "usage: rt1w [--scene N] [--width W] [--height H] [--spp S] [--depth D] [--seed G] [--build-seed B] [--device I] [--earth file.rgb8 W H] [--out file.ppm] [--specialise | --generic] [--reference-stream] [--f32] [--near-far] [--sah]\n"); return 2; }
    }
    std::vector<unsigned char> earth;
    if (!earth_path.empty()) {
        FILE* f = std::fopen(earth_path.c_str(), "rb");
        if (!f) { std::perror("rt1w: --earth"); return 1; }
        earth.resize((size_t)earth_w * earth_h * 3);
        size_t got = std::fread(earth.data(), 1, earth.size(), f);
        std::fclose(f);
        if (got != earth.size()) { std::fprintf(stderr, "rt1w: --earth: short file\n"); return 1; }
    }
    /* aspect_ratio: 16/9 by default, 1.0 for arms 5, 6 and the final scene (main.rs:798,868,896,917);
     * an explicit --width/--height pair overrides it */
    bool square = (arm == 5 || arm == 6 || arm < 0 || arm > 6);
    double aspect = square ? 1.0 : 16.0 / 9.0;
    if (width > 0 && height > 0) aspect = (double)width / (double)height;
    rt1w_scene* scene = nullptr;
    uint32_t defaults[3];
    if (rt1w_scene_build_reference(arm, build_seed, aspect, earth.empty() ? nullptr : earth.data(), earth_w, earth_h, &scene, defaults) < 0)
        return fail("scene");
    if (width <= 0) width = defaults[0];
    if (height <= 0) height = (long)((double)width / aspect); /* main.rs:939 */
    if (spp <= 0) spp = defaults[2];
    if (sah && rt1w_scene_set_bvh_build(scene, RT1W_BVH_SAH) < 0) return fail("BVH build");
    if (near_far && rt1w_scene_set_walk_order(scene, RT1W_WALK_NEAR_FAR) < 0) return fail("walk order");
    rt1w_context* ctx = nullptr;
    if (rt1w_context_create(device, scene, &ctx) < 0) return fail("context");
    if (specialise && !generic) {
        rt1w_specialise_info si;
        if (rt1w_context_specialise(ctx, 0, &si) < 0) std::fprintf(stderr, "rt1w: not specialised: %s\n", rt1w_last_error());
        else std::fprintf(stderr, "rt1w: kernel %s (%s, %.1f s in the compiler)\n", si.key, si.from_cache ? "from the kernel cache" : "compiled", si.compile_ms / 1e3);
    }
    rt1w_render_params p;
    std::memset(&p, 0, sizeof p);
    p.width = (uint32_t)width; p.height = (uint32_t)height; p.tile_w = p.width; p.tile_h = p.height;
    p.spp = (uint32_t)spp; p.max_depth = (uint32_t)depth; p.global_seed = (uint32_t)seed;
    if (generic) p.flags |= RT1W_GENERIC;
    if (reference_stream) p.flags |= RT1W_RNG_REFERENCE;
    if (f32) p.precision = RT1W_PRECISION_F32;
    /* the reference collects the rows top-down, counting them down on stderr (main.rs:957-960,995-998), then prints them
     * (main.rs:1003-1007); here the rows are quantised on the device and written as their strips land */
    std::vector<unsigned char> img((size_t)width * height * 3);
    FILE* o = out_path.empty() ? stdout : std::fopen(out_path.c_str(), "w");
    if (!o) { std::perror("rt1w: --out"); return 1; }
    std::fprintf(o, "P3\n%u %u\n255\n", p.width, p.height);
    struct Sink { FILE* o; const unsigned char* img; uint32_t w, written; std::string line; } sink{o, img.data(), p.width, 0, {}};
    auto on_rows = [](void* user, uint32_t rows_done, uint32_t rows_total) -> int {
        Sink& k = *static_cast<Sink*>(user);
        for (; k.written < rows_done; ++k.written) {
            const unsigned char* row = k.img + (size_t)k.written * k.w * 3;
            k.line.clear();
            char buf[16];
            for (uint32_t i = 0; i < k.w; ++i) {
                int n = std::snprintf(buf, sizeof buf, "%u %u %u\n", row[3 * i], row[3 * i + 1], row[3 * i + 2]); /* color.rs:59-64 */
                k.line.append(buf, (size_t)n);
            }
            std::fwrite(k.line.data(), 1, k.line.size(), k.o);
        }
        std::fprintf(stderr, "\rScanlines remaining: %u ", rows_total - rows_done);
        return 0;
    };
    rt1w_stats st;
    std::fprintf(stderr, "rt1w: scene arm %d, %ldx%ld, %ld spp, depth %ld\n", arm, width, height, spp, depth);
    if (rt1w_render_rows(ctx, &p, 0, RT1W_ROWS_U8, img.data(), on_rows, &sink, &st) < 0) return fail("render");
    std::fprintf(stderr, "\nDone\nrt1w: %.1f ms kernels, %.1f Mpaths/s, %.2f segments/path, kernel variant V%u%s\n", st.kernel_ms,
                 (double)st.paths / st.kernel_ms / 1e3, (double)st.segments / (double)st.paths, st.variant, (st.sorted & 4u) ? " (scene-specialised)" : "");
    if (o != stdout) std::fclose(o);
    rt1w_context_destroy(ctx);
    rt1w_scene_destroy(scene);
    return 0;
}
