/* rt1w.h -- C ABI of the MI355X path tracer (librt1w.so).
 *
 * The reference (hatoo/raytracing-1w, /root/reference) has NO FFI or plugin
 * interface: the hot path is a closure inlined in `main`
 * (src/main.rs:957-1001) over `camera`, `world: BVHNode`,
 * `lights: Option<Vec<Box<dyn Hittable>>>`, `background`, image size, spp and
 * MAX_DEPTH.  This header is the seam a Rust host would bind with
 * `extern "C"`: the scene is described through constructors that mirror the
 * reference's own (same names, same argument order and meaning), flattened
 * once, uploaded once, and `rt1w_render` replaces the closure.
 * INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *   - every function returns an `int`: >= 0 success (an id where the function
 *     creates something), < 0 one of RT1W_ERR_*; `rt1w_last_error()` returns a
 *     thread-local message.  Nothing aborts or throws across the ABI; the
 *     reference's panics (src/bvh.rs:61,67, src/hittable.rs:153) become
 *     RT1W_ERR_INVALID.
 *   - caller owns every buffer it passes; the library copies what it keeps.
 *   - a scene is immutable after rt1w_scene_commit and may back many contexts;
 *     one context = one GPU + one HIP stream; calls on distinct contexts are
 *     thread-safe, calls on one context are serialised by the caller.
 *   - there is no CPU render path: without a usable GPU rt1w_context_create
 *     fails with RT1W_ERR_DEVICE.
 *   - all geometry/colour arithmetic is f64 (`type Float = f64`, src/main.rs:1).
 */
#ifndef RT1W_H
#define RT1W_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* librt1w.so is built with -fvisibility=hidden: what this header declares is everything it exports (plus three rt1w_internal_* hooks
 * of its own diagnostics library, csrc/rt1w_internal.h) */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define RT1W_OK 0
#define RT1W_ERR_INVALID (-1)     /* bad argument / bad id / empty BVH (src/bvh.rs:61) */
#define RT1W_ERR_UNSUPPORTED (-2) /* graph shape the device format cannot express */
#define RT1W_ERR_DEVICE (-3)      /* no GPU, HIP error */
#define RT1W_ERR_NOMEM (-4)
#define RT1W_ERR_STATE (-5)       /* e.g. mutate after commit, render before commit */
#define RT1W_ERR_CANCELLED (-6)   /* the progress callback of rt1w_render_rows asked to stop */

typedef struct rt1w_scene rt1w_scene;
typedef struct rt1w_context rt1w_context;

const char* rt1w_last_error(void);
const char* rt1w_version(void);

/* ---- scene construction: one-shot host work (src/main.rs:192-795, 807-951) ---- */

/* `build_seed` seeds the scene-build random stream that replaces the
 * reference's entropy-seeded `MyRng::from_entropy()` (src/main.rs:803).  Every
 * constructor below that takes `rng` in the reference (AABox::new, BVHNode::new,
 * NoiseTexture::new) draws from it at call time, in call order, exactly like
 * the Rust constructors do. */
int rt1w_scene_create(uint64_t build_seed, rt1w_scene** out);
void rt1w_scene_destroy(rt1w_scene* s);

/* draws for the host's own scene code (`rng.gen()`, `rng.gen_range(a..b)` in
 * random_scene / final_scene, src/main.rs:212-245,653,777-779) from the same
 * build stream; value returned through *out */
int rt1w_scene_rng_f64(rt1w_scene* s, double* out);
int rt1w_scene_rng_range(rt1w_scene* s, double low, double high, double* out);

/* Texture implementors (src/texture.rs).  Return a texture id. */
int rt1w_texture_solid(rt1w_scene* s, const double rgb[3]);                 /* SolidColor   texture.rs:13-16 */
int rt1w_texture_checker(rt1w_scene* s, int odd, int even);                 /* CheckerTexture texture.rs:18-22 */
int rt1w_texture_noise(rt1w_scene* s, double scale);                        /* NoiseTexture256::new(scale, rng) texture.rs:32-38, perlin.rs:25-43 */
int rt1w_texture_noise_tables(rt1w_scene* s, double scale, const double ranvec[768],
                              const uint32_t perm_x[256], const uint32_t perm_y[256],
                              const uint32_t perm_z[256]);                  /* same, caller-supplied tables */
int rt1w_texture_image(rt1w_scene* s, const uint8_t* rgb8, uint32_t width, uint32_t height); /* DynamicImage texture.rs:67-89 (decoded RGB8, row 0 = top) */

/* Material implementors (src/material.rs).  Return a material id. */
int rt1w_material_lambertian(rt1w_scene* s, int albedo_texture);            /* material.rs:52-55 */
int rt1w_material_metal(rt1w_scene* s, const double albedo[3], double fuzz);/* material.rs:57-61 */
int rt1w_material_dielectric(rt1w_scene* s, double ir);                     /* material.rs:127-130 */
int rt1w_material_diffuse_light(rt1w_scene* s, int emit_texture);           /* material.rs:63-66 */
int rt1w_material_null(rt1w_scene* s);                                      /* impl Material for () material.rs:68 */

/* Hittable implementors.  Return a hittable id.  A hittable id may be used as
 * a child exactly once (the reference owns children by Box). */
int rt1w_hittable_sphere(rt1w_scene* s, const double center[3], double radius, int material);  /* sphere.rs:16-20 */
int rt1w_hittable_moving_sphere(rt1w_scene* s, const double center0[3], const double center1[3],
                                double time0, double time1, double radius, int material);      /* moving_sphere.rs:13-20 */
int rt1w_hittable_xy_rect(rt1w_scene* s, double x0, double x1, double y0, double y1, double k, int material); /* aarect.rs:15-22 */
int rt1w_hittable_xz_rect(rt1w_scene* s, double x0, double x1, double z0, double z1, double k, int material); /* aarect.rs:25-32 */
int rt1w_hittable_yz_rect(rt1w_scene* s, double y0, double y1, double z0, double z1, double k, int material); /* aarect.rs:35-42 */
int rt1w_hittable_aabox(rt1w_scene* s, const double p0[3], const double p1[3], int material);  /* AABox::new aabox.rs:22-84 (draws rng) */
int rt1w_hittable_translate(rt1w_scene* s, int child, const double offset[3]);                 /* hittable.rs:49-52 */
int rt1w_hittable_rotate_y(rt1w_scene* s, int child, double time0, double time1, double angle_deg); /* RotateY::new hittable.rs:158-202 */
int rt1w_hittable_flip_face(rt1w_scene* s, int child);                                         /* hittable.rs:61 */
int rt1w_hittable_constant_medium(rt1w_scene* s, int boundary, double density, int texture);   /* ConstantMedium::new constant_medium.rs:22-28 */
int rt1w_hittable_bvh(rt1w_scene* s, const int* children, uint32_t n, double time0, double time1); /* BVHNode::new bvh.rs:54-103 (draws rng) */

/* what `main` hands to the pixel loop (src/main.rs:807-951) */
int rt1w_scene_set_world(rt1w_scene* s, int hittable);
int rt1w_scene_set_lights(rt1w_scene* s, const int* hittables, uint32_t n); /* n = 0: `lights = None` -> ray_color_without_light_objects */
int rt1w_scene_set_background(rt1w_scene* s, const double rgb[3]);
int rt1w_scene_set_camera(rt1w_scene* s, const double look_from[3], const double look_at[3],
                          const double vup[3], double vfov_deg, double aspect_ratio,
                          double aperture, double focus_dist, double time0, double time1); /* Camera::new camera.rs:22-59 */
/* BVH is already built by the constructors; commit flattens the graph into the
 * device record arrays and freezes the scene. */
int rt1w_scene_commit(rt1w_scene* s);

/* the scene table of `main` (src/main.rs:815-937): arm 0 random_scene, 1 two_spheres,
 * 2 two_perlin_spheres, 3 earth, 4 simple_light, 5 cornel_box, 6 cornel_smoke,
 * any other final_scene; built through the constructors above.  `aspect_ratio`
 * is what `main` passes to Camera::new (1.0 for arms 5,6,7; 16/9 otherwise, or
 * the caller's override).  `earth_rgb8` (1024x512 decoded assets/earthmap.jpg or
 * any RGB8 image) is needed by arms 3 and 7, else may be NULL.
 * `defaults[3]` receives the arm's image_width, image_height, samples_per_pixel. */
int rt1w_scene_build_reference(int arm, uint64_t build_seed, double aspect_ratio,
                               const uint8_t* earth_rgb8, uint32_t earth_w, uint32_t earth_h,
                               rt1w_scene** out, uint32_t defaults[3]);

/* OPT-IN traversal order of the BVHs.  Default RT1W_WALK_REFERENCE: every BVH node's children are visited
 * left then right exactly as `BVHNode::hit` does (src/bvh.rs:38-47) -- the walk then tests the very primitives the reference
 * tests, in its order, which is what makes the default results provably the reference's.  RT1W_WALK_NEAR_FAR visits the
 * child on the ray's near side first (fewer node visits: the closest hit is found earlier and prunes the rest).  The
 * closest hit does not depend on the order; what does is (a) which of two primitives hit at exactly the same t wins -- kept
 * the reference's by preferring the larger pre-order index -- (b) the random numbers a ConstantMedium draws while being
 * visited -- kept by leaving every node with a medium below it in the reference's order -- and (c) primitive tests the
 * reference never runs because a box test with a fresher t_max pruned them: equal "almost surely" (a primitive hit outside
 * its own bounding box by rounding would be needed) -- except for MovingSphere: the reference gives a scattered ray
 * time = hit t (src/main.rs:86,145), which moves such a sphere far outside the box the BVH holds for it, so whether it is
 * tested depends on the order.  RT1W_WALK_NEAR_FAR therefore also leaves subtrees with moving spheres in the reference's
 * order: with it the FRAMES were identical on every scene tested (all eight arms, random graphs), but that is a measurement, not
 * a proof -- on axis-aligned geometry a box's entry distance can equal a neighbouring face's hit distance, so which of two
 * edge-on hits survives a prune can depend on the order -- and path lengths (rt1w_stats.segments) may differ for a few rays.
 * RT1W_WALK_NEAR_FAR_ALL reorders the moving-sphere subtrees too and does change frames (measured on random_scene: 2 of
 * 9600 pixels differ at 4 spp).  Call on a committed scene, before creating contexts. */
#define RT1W_WALK_REFERENCE 0u
#define RT1W_WALK_NEAR_FAR 1u
#define RT1W_WALK_NEAR_FAR_ALL 2u
int rt1w_scene_set_walk_order(rt1w_scene* s, uint32_t mode);

/* OPT-IN build of the BVHs.  Default RT1W_BVH_REFERENCE: `BVHNode::new` as written -- a random axis per
 * node, objects sorted by their box minimum on it, split at the median (src/bvh.rs:84-100).  RT1W_BVH_SAH rebuilds the tree
 * over every BVH's leaf set by the surface-area heuristic (each split minimises area(L)*|L| + area(R)*|R| over the three axes
 * and every position of the centroid order; boxes are still `surrounding_box` of the children, src/aabb.rs:42-55): fewer
 * node visits per ray.  The closest hit of a ray does not depend on the tree, but everything that depends on the ORDER of the
 * walk does -- ties in t, the random numbers a ConstantMedium draws while visited, MovingSphere tests of scattered rays
 * (see rt1w_scene_set_walk_order) -- so frames are the reference's only statistically, never bit for bit in general
 * (measured: DESIGN.md section 5).  Combine with RT1W_WALK_NEAR_FAR* freely.  Call on a committed scene, before creating
 * contexts; RT1W_ERR_UNSUPPORTED (scene unchanged) if the rebuilt trees need a deeper traversal stack than the kernels have. */
#define RT1W_BVH_REFERENCE 0u
#define RT1W_BVH_SAH 1u
/* RT1W_BVH_BEST_AXIS: `BVHNode::new` as written -- the reference's sort key, stable order, median split and one- / two-object shapes
 * (src/bvh.rs:60-100) -- with the axis of src/bvh.rs:84 CHOSEN (the one whose median split has the lowest area(L)*|L| + area(R)*|R|)
 * instead of drawn from the entropy-seeded generator of src/main.rs:803.  Every axis sequence is a tree some run of the reference
 * builds, so unlike RT1W_BVH_SAH this is a tree the reference itself can produce; nested `BVHNode::new` calls stay separate BVHs.
 * In its topology stream (rt1w_scene_get_bvh_topology) a leaf's number is the object's position in the list its `BVHNode::new` call
 * received (src/bvh.rs:55), and a BVHNode that is an object of another BVH is a leaf whose own tree follows its number. */
#define RT1W_BVH_BEST_AXIS 2u
int rt1w_scene_set_bvh_build(rt1w_scene* s, uint32_t mode);
/* The trees RT1W_BVH_SAH built in place of `BVHNode::new` (src/bvh.rs:54-103), written down so that they can be checked from
 * outside (tests feed them to the literal oracle, which then runs `BVHNode::hit` src/bvh.rs:25-50 over the same trees).
 * One stream of int32 for the whole scene, the rebuilt BVHs in the order the flattener meets them (depth first): each tree in
 * pre-order, -1 = an inner node (`BVHChild::Two`, its two subtrees follow), -2 = a `BVHChild::One` (its one object follows: an
 * object whose sibling is a subtree keeps a node of its own with its own box, as in the reference's trees, src/bvh.rs:63-70),
 * k >= 0 = the k-th leaf of that BVH counted left to
 * right through the tree `BVHNode::new` had built (nested BVHNodes are merged into their parent's leaf set); what a leaf holds
 * inside (an AABox's side BVH, a wrapped BVH, a medium's boundary) follows right after the leaf's number.  Returns the number
 * of entries (0 under RT1W_BVH_REFERENCE); `out` may be NULL to ask for the size. */
int64_t rt1w_scene_get_bvh_topology(const rt1w_scene* s, int32_t* out, uint64_t capacity);

/* introspection of the committed flat scene (tests, DESIGN.md numbers) */
typedef struct rt1w_scene_info {
    uint32_t n_nodes, n_lights, n_materials, n_textures, n_perlin;
    uint32_t stack_need;   /* traversal stack entries the scene needs */
    uint32_t scope_depth;  /* deepest wrapper nesting */
    uint32_t has_media;
    uint32_t has_textures; /* any non-solid texture */
    uint32_t has_moving;   /* any MovingSphere */
    uint32_t variant;      /* kernel variant rt1w_render picks (V0..V5) */
    uint64_t bytes;        /* bytes uploaded per context */
} rt1w_scene_info;
int rt1w_scene_get_info(const rt1w_scene* s, rt1w_scene_info* out);
/* copy of the flat arrays (the exact bytes a context uploads); `what`: 0 nodes,
 * 1 lights, 2 materials, 3 textures, 4 perlin, 5 images, 6 camera+background.
 * Returns bytes written, or needed size if buf==NULL. */
int64_t rt1w_scene_copy_flat(const rt1w_scene* s, int what, void* buf, uint64_t cap);

/* ---- execution (replaces src/main.rs:957-1001) ---- */

int rt1w_device_count(void);
/* uploads the committed scene's flat arrays once (they are immutable afterwards); loads the scene-specialised kernel if the kernel
 * cache has it; counts node visits in a 128 x 128 x 1 render of the scene's own camera (5-15 ms) to rank the records the stack-walk
 * kernels keep in LDS (csrc/rt_walk_table.h) */
int rt1w_context_create(int device_id, const rt1w_scene* s, rt1w_context** out);
void rt1w_context_destroy(rt1w_context* c);

#define RT1W_OUT_SUM 1u   /* write raw per-pixel sums (for sample-range sharding) instead of into_sampled means */
#define RT1W_LDS_NODES 4u /* experiment: stack variants read node records from an LDS copy (scenes <= 1024 nodes); measured slower than the default */
#define RT1W_GENERIC 8u   /* do not use a scene-specialised kernel even if the context has one (rt1w_context_specialise) */
#define RT1W_WAVEFRONT 16u /* opt-in, big scenes (stack-walk variants) and the one-shot entries only: path state queued in HBM as SoA records, a trace kernel + a shade kernel per bounce, a finish kernel for the tail; bit-identical to the default, measured 0.4-0.9x its speed (docs/LAB_NOTES.md).  Since round 4 the form lives in the diagnostics library librt1w_lab.so (csrc/wavefront.hip), which registers itself with librt1w.so when it is loaded: without it the flag answers RT1W_ERR_UNSUPPORTED.  Scenes that run a sweep kernel ignore the flag (stats.sorted bit 3 says what ran); rt1w_render_rows refuses it */
#define RT1W_OUT_FRAME 32u /* rt1w_render only: `out_rgb` is the WHOLE image [height][width][3] (row 0 = j = 0) and the call writes just its tile's pixels at their image positions -- several contexts / processes fill one (shared, pinned) host frame: the host gather of the image-tiled multi-GPU job */
#define RT1W_RNG_REFERENCE 64u /* PARITY MODE: draw from the reference's own generator instead of the Philox streams -- `StdRng::seed_from_u64(j * image_width + i)` (src/main.rs:964; ChaCha12, rand 0.8.4), one stream per pixel drawn on through all its samples in order (sample_offset must be 0, global_seed is ignored).  The frame is then the Rust program's own, pixel for pixel: the GPU reproduces rest_of_your_life.png.  Slower than the default (a lane owns a pixel for all its samples) */
#define RT1W_CLASSIC_WALK 128u /* tests/ablation: sphere scenes (random_scene) walk their BVH with the pair walk of csrc/rt_walk_pair.h by default (box work and leaf work in separate phases, inner boxes in f32 rounded outward, every sphere gated by its group's own f64 box at the reference's moment: the same frames bit for bit, stats.sorted bit 7 says it ran); this flag keeps the one-entry-per-step walk.  Likewise scenes whose every ConstantMedium is bounded by a bare Sphere (final_scene) run stack-walk kernels built without the general boundary walks (rt_flat.h: RtCfgSphereMedia; stats.sorted bit 8); this flag keeps the general kernels */
#define RT1W_PROBE_COHERENT 0x40000000u /* MEASUREMENT ONLY -- the frame written is NOT the image: every wave's 64 lanes trace the SAME path (one pixel of every 8x8 block, each 64 times), so the kernel runs without divergence and its instruction count per traced segment (rocprofv3 SQ_INSTS_VALU x 64 / stats.segments) is what ONE path needs in this kernel's code: the `necessary` side of bench.py's `roofline.valu` (tools/bench_pmc.sh).  Replaces nothing of the reference: a property of this implementation's measurement */
#define RT1W_NO_NODE_CACHE 0x10000u /* tests/ablation: big scenes' stack-walk kernels keep the most visited node records in LDS (csrc/rt_walk_table.h: ranked by a visit count at context creation; stats.sorted bit 10 says the cache ran; same frames bit for bit); this flag keeps the kernels that read every record from memory */
#define RT1W_UNSORTED 2u  /* tests/ablation: use the plain persistent kernel (no workgroup-level path reordering: neither the reordering kernel of the small scenes nor, for sphere-media scenes such as final_scene, the reordering of the finished paths at the end of every slice of the stack walk, stats.sorted bit 9) */
#define RT1W_FORCE_VARIANT(v) ((((uint32_t)(v)) + 1u) << 8) /* tests: force kernel variant v (must be valid for the scene) */

typedef struct rt1w_render_params {
    uint32_t width, height;         /* image_width, image_height (src/main.rs:799,939) */
    uint32_t x0, y0, tile_w, tile_h;/* tile to render; y is the reference's row index j (j = height-1 is the TOP row of the PPM) */
    uint32_t spp;                   /* samples_per_pixel rendered by this call */
    uint32_t sample_offset;         /* first absolute sample index; 0 unless sharding samples */
    uint32_t max_depth;             /* MAX_DEPTH = 50 (src/main.rs:801) */
    uint32_t global_seed;
    uint32_t chunk;                 /* samples per work item, 0 = library default (rt1w_scene_default_chunk) */
    uint32_t flags;                 /* RT1W_OUT_* */
    /* Row-interleaved tile, for tiling ONE image over several GPUs with balanced load (the reference hands rows to rayon
     * workers, src/main.rs:957-963; Cornell rows differ in cost by region): the tile's rows are strips of `strip_rows`
     * image rows taken every `strip_period` rows, i.e. tile row r is image row y0 + (r / strip_rows) * strip_period +
     * r % strip_rows.  Rank k of n passes y0 = k * strip_rows, strip_period = n * strip_rows, tile_h = the rows it owns.
     * Both 0: an ordinary contiguous tile.  One launch renders all of the rank's strips. */
    uint32_t strip_rows, strip_period;
    uint32_t precision;             /* RT1W_PRECISION_*: 0 = f64, the reference's `type Float = f64` (src/main.rs:1) */
    uint32_t partial_mib;           /* memory-constrained hosts / tests: upper bound, in MiB, of the buffer of chunk partial sums (24 B per work item); 0 = 8192.
                                       A render that needs more runs as several passes over sample ranges -- the same bits (rt1w_scene_default_chunk) */
} rt1w_render_params;
#define RT1W_PRECISION_F64 0u
/* the reference's switch set the other way, `type Float = f32`: rays, hit records, boxes, camera and colours in f32 (the
 * random draws are still made in 64 bits and rounded, the elementary functions are evaluated in 64 bits and rounded once, a
 * pixel's samples are summed in f64).  Statistically equal to the f64 frame, not bitwise.  Runs the generic kernels, or the
 * f32 build of the scene-specialised kernel where rt1w_context_specialise made one (renders themselves never compile). */
#define RT1W_PRECISION_F32 1u

typedef struct rt1w_stats {
    uint64_t paths;        /* pixels * spp */
    uint64_t segments;     /* traced rays (camera + bounces) */
    double kernel_ms;      /* HIP-event time of the render kernel(s) on the context's stream */
    double total_ms;       /* host wall time of the call incl. device->host copy */
    uint32_t chunk, n_chunks;
    uint32_t grid, block;
    uint32_t variant;      /* feature variant of the kernel (V0..V5: rt_flat.h).  With bit 2 of `sorted` set the kernel that ran is the
                              scene-specialised SWEEP kernel whatever walk this number names (scenes of 65-256 nodes report a stack
                              variant here because that is what the generic code would have used) */
    uint32_t sorted;       /* bit 0: the reordering kernel ran; bit 1: node records in LDS; bit 2: scene-specialised kernel; bit 3:
                              wavefront form; bit 4: reference-stream kernel (RT1W_RNG_REFERENCE); bit 5: f32 kernel; bit 7: pair walk
                              (sphere scenes); bit 8: sphere-media build of the stack walk; bit 9: finished paths reordered across the
                              workgroup at the end of every slice of the stack walk; bit 10: most visited node records in LDS (walk table) */
} rt1w_stats;

/* default work-item size for a (tile, spp): deterministic, documented in DESIGN.md.  This is the scene-independent rule (what the
 * small scenes' reordering kernels run with); see rt1w_scene_default_chunk for what a render of a given scene uses */
uint32_t rt1w_default_chunk(uint32_t tile_w, uint32_t tile_h, uint32_t spp);
/* The work-item size rt1w_render* use for THIS scene when rt1w_render_params.chunk is 0 (the samples of a pixel are summed per work
 * item, then the items in order: src/main.rs:966-992 sums them all in order, which is the case chunk = 1).  Scenes that run a
 * stack-walk kernel (more than 64 nodes) take ONE sample per item: a lane whose path ended fetches its next item with the other free
 * lanes of its wave, so they restart on neighbouring pixels (coherent camera rays; measured +8-15 % on final_scene / random_scene),
 * and the pixel sum is the reference's own sequential sum; scenes on the sweep kernels keep rt1w_default_chunk.  Renders whose
 * partial sums would not fit 8 GiB run as several passes over sample ranges with the same bits.  A host that tiles one image over
 * several GPUs passes this value (for the WHOLE frame) as `chunk` to every tile.  Committed scenes only (0 otherwise). */
uint32_t rt1w_scene_default_chunk(const rt1w_scene* s, uint32_t tile_w, uint32_t tile_h, uint32_t spp);

/* Renders the tile into caller memory: out_rgb[(y - y0) * tile_w + (x - x0)][3],
 * row 0 = j = y0.  Values are the reference's `pixel_color.into_sampled(spp)`
 * (src/main.rs:992, color.rs:14-21), i.e. linear f64 means before gamma, or raw
 * sums with RT1W_OUT_SUM. */
int rt1w_render(rt1w_context* c, const rt1w_render_params* p, double* out_rgb, rt1w_stats* stats);
/* same, but `d_out_rgb` is a device pointer on the context's GPU (e.g. a torch
 * tensor); no host copy.  Synchronises the context's stream before returning. */
int rt1w_render_device(rt1w_context* c, const rt1w_render_params* p, void* d_out_rgb, rt1w_stats* stats);

/* same render, but the tile comes back already quantised ON THE DEVICE exactly as the reference prints it
 * (Display for SampledColor src/color.rs:56-65: (256 * sqrt(c).clamp(0, 0.999)) as usize) and in the reference's row
 * order (top row = j = y0 + tile_h - 1 first, src/main.rs:957-960,1003-1007): out_rgb8[tile_h][tile_w][3].  No host
 * post-pass, 1/8 of the device->host bytes.   */
int rt1w_render_u8(rt1w_context* c, const rt1w_render_params* p, uint8_t* out_rgb8, rt1w_stats* stats);

/* Strip-wise render with progress, for big frames (C5 is 199 MB of f64 means).  The reference renders the rows top-down
 * and reports progress as rows finish (src/main.rs:957-960 row order, :995-998 the stderr progress line).  This entry
 * traces the tile in strips of `strip_rows` image rows from the top row (j = y0 + tile_h - 1) downwards; the finished strip
 * is copied to the host on a second stream while the next one is traced, and `progress(user, rows_done, rows_total)` is
 * called on the calling thread each time a strip has landed in `out` (rows_done counts from the top).  A non-zero return
 * from the callback stops the render: strips already reported are valid, the call returns RT1W_ERR_CANCELLED.
 *   format RT1W_ROWS_F64: out = double[tile_h][tile_w][3], the layout of rt1w_render (row 0 = j = y0);
 *   format RT1W_ROWS_U8 : out = uint8_t[tile_h][tile_w][3], the layout of rt1w_render_u8 (top row first), so a PPM writer can
 *                         stream rows out as they are reported.
 * Every strip uses the sample-chunk size of the whole tile, so the result is bit-identical to rt1w_render / rt1w_render_u8.
 * strip_rows = 0 picks about 16 strips, fewer when a strip would hold less than ~4M (pixel, sample-chunk) work items.  `progress` may be NULL.  stats (optional) are totals over the strips. */
#define RT1W_ROWS_F64 0
#define RT1W_ROWS_U8 1
typedef int (*rt1w_progress_fn)(void* user, uint32_t rows_done, uint32_t rows_total);
int rt1w_render_rows(rt1w_context* c, const rt1w_render_params* p, uint32_t strip_rows, int format, void* out,
                     rt1w_progress_fn progress, void* user, rt1w_stats* stats);

/* Page-locked host memory for output frames (hipHostMalloc / hipHostRegister): device->host copies into it run at full
 * PCIe rate and asynchronously.  rt1w_host_register pins memory the caller already owns, e.g. a POSIX shared-memory
 * mapping that several single-GPU processes fill with RT1W_OUT_FRAME. */
int rt1w_host_alloc(uint64_t bytes, void** out);
int rt1w_host_free(void* p);
int rt1w_host_register(void* p, uint64_t bytes);
int rt1w_host_unregister(void* p);

/* ---- scene-specialised kernels ----
 * Small scenes are traversed by a stackless pre-order sweep.  When the node kinds and subtree ends are compile-time
 * constants that sweep unrolls into straight-line code along the scene's own tree (same arithmetic, bit-identical
 * frames; Cornell box, 29 nodes: 1.3x the generic sweep; scenes of 65-256 nodes: 1.5-2.3x the stack walk).  The constants
 * are only known once a scene is committed, so such a kernel is generated per scene topology and compiled with hiprtc
 * (1-8 s), then kept in a kernel cache: <directory of librt1w.so>/kernels (filled by the build for the reference's own
 * scene arms) and $RT1W_KERNEL_CACHE or ~/.cache/rt1w.  rt1w_context_create looks the scene up in the cache and uses a
 * hit silently; rt1w_context_specialise compiles on a miss.  A render of >= 2^35 paths (2^32 for scenes of more than 64 nodes) compiles on
 * its own -- the compile then costs less than it saves -- unless RT1W_NO_JIT is set in the environment.  Scenes of more than 256 nodes keep the
 * generic kernels: RT1W_ERR_UNSUPPORTED.  RT1W_GENERIC in rt1w_render_params.flags selects the generic kernel for one
 * render. */
#define RT1W_SPECIALISE_CACHED_ONLY 1u /* do not run the compiler: RT1W_ERR_STATE on a cache miss */
typedef struct rt1w_specialise_info {
    char key[24];        /* cache key: hash of the generated source, the library's embedded headers and the compiler options */
    uint32_t active;     /* 1: renders on this context now use the specialised kernel */
    uint32_t from_cache; /* 1: the code object came from a cache, 0: it was compiled by this call */
    double compile_ms;   /* time spent in hiprtc by this call */
    uint32_t grid, vgprs;/* persistent grid of the kernel; 0 if unknown */
} rt1w_specialise_info;
int rt1w_context_specialise(rt1w_context* c, uint32_t flags, rt1w_specialise_info* info /* may be NULL */);
/* cache key of the specialised kernel of a committed scene (16 hex digits + NUL): the code object is
 * `sweep_<key>.hsaco` in the kernel cache.  No GPU needed.  RT1W_ERR_UNSUPPORTED for scenes of more than 256 nodes. */
int rt1w_scene_kernel_key(const rt1w_scene* s, char out[24]);

/* ---- output side (src/color.rs) ---- */

/* Color::into_sampled (color.rs:14-21) over n pixels: NaN scrub of the SUM, then * 1/spp */
int rt1w_resolve(const double* sums, uint64_t n_pixels, uint32_t spp, double* means);
/* Display for SampledColor (color.rs:56-65): gamma-2, clamp, *256, truncate; n values -> n bytes */
int rt1w_quantize(const double* means, uint64_t n_values, uint8_t* out);
/* the whole P3 text of src/main.rs:953,1003-1007 for an image whose row 0 is j = 0
 * (rows are emitted top-down, j = height-1 first).  Returns bytes written
 * (excluding NUL) or needed size if buf==NULL. */
int64_t rt1w_format_ppm(const double* means, uint32_t width, uint32_t height, char* buf, uint64_t cap);

/* sizeof of the ABI structs as this library was compiled (bindings check their own layout against it):
 * 0 rt1w_render_params, 1 rt1w_stats, 2 rt1w_scene_info, 3 rt1w_specialise_info; 0 for anything else */
uint32_t rt1w_abi_sizeof(int what);

/* ---- diagnostics ---- */
/* evaluates the numerical contract (include/rt1w_num.h) ON THE DEVICE for n inputs:
 * fn 0 a/b, 1 sqrt|a|, 2 sin a, 3 cos a, 4 acos(a/(|a|+1)), 5 atan2(a,b), 6 log|b|,
 * 7 first gen_f64 of stream (pixel=i, sample=(uint32)a_bits), 8 gen_range(-1,1) of same.
 * Used by the GPU tests to prove host/device bit equality. */
int rt1w_debug_eval(rt1w_context* c, int fn, const double* a, const double* b, double* out, uint64_t n);
/* AABB::hit (src/aabb.rs:13-32) ON THE DEVICE for n cases, in[i] = {min[3], max[3], origin[3], direction[3], t_min, t_max}:
 * `out_literal` from the compare/select form, `out_fast` from the max/min-instruction form the kernels use when no
 * bound is NaN.  Tests compare both with the host's literal form. */
int rt1w_debug_aabb(rt1w_context* c, const double* in, int* out_literal, int* out_fast, uint64_t n);
/* The shading-side leaf functions ON THE DEVICE for n inputs in[i] = {u, v, p.x, p.y, p.z}, out[i] = 3 doubles:
 * mode 0 `Texture::value(u, v, p)` of texture `tex` of the context's scene (src/texture.rs:40-89) -> rgb;
 * mode 1 `Perlin::noise(p)` and `Perlin::turb(p, 7)` of Perlin table `tex` (src/perlin.rs:46-86) -> out[0], out[1];
 * mode 2 `sphere_uv(p)` (src/math.rs:67-71) -> out[0] = u, out[1] = v.
 * Known-answer tests of what no artefact of the reference reaches (lattice points of the noise, checker parity, poles and
 * seam of sphere_uv). */
int rt1w_debug_texture(rt1w_context* c, int mode, uint32_t tex, const double* in, double* out, uint64_t n);
/* per-phase wave-cycle totals of the render kernel since the last reset.  Only the diagnostic build
 * (make stamps -> librt1w_stamps.so, -DRT_STAMPS) fills them (returns 1); the shipped library executes no
 * stamp and returns 0 with zeros.  Buckets: see context.hip. */
int rt1w_debug_stamps(rt1w_context* c, uint64_t out[16], int reset);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
