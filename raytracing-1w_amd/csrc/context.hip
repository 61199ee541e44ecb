/* context.hip -- execution half of the C ABI: device context, the persistent-thread
 * render kernel and the resolve kernel.  gfx950 only.
 *
 * Kernel shape (replaces the rayon loop of src/main.rs:957-1001):
 *   - one lane owns one path at a time; a lane whose path ended regenerates in
 *     place (next sample of its work item, or a new work item), so lanes of a
 *     wave stay busy although path lengths differ (1..50 segments);
 *   - a work item = (pixel, chunk of `chunk` consecutive samples).  Items are
 *     handed out by one global counter; the first grid-size items are assigned
 *     statically, later ones by a wave-aggregated atomic (one atomic per wave per
 *     refill round: ballot + mbcnt prefix);
 *   - the pixel sum of a chunk lives in registers and is stored once
 *     (24 B per item, `partial[chunk][pixel]`); no floating-point atomics, so the
 *     summation order -- and therefore every bit of the result -- is fixed:
 *     sum over chunks in order of (sum over the chunk's samples in order);
 *   - the BVH traversal stack is per-lane in LDS, laid out [entry][lane] so that a
 *     wave's pushes/pops are bank-conflict free whatever depth each lane is at;
 *   - Philox4x32-10 state is 11 VGPRs (rt1w_num.h), nothing RNG-related in memory.
 */
#include <hip/hip_runtime.h>

#include <chrono>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rt1w.h"
#include "rt_kernel_plain.h"
#include "scene.h"
#include "jit.h"
#include "rt1w_internal.h"

/* context_ref.hip: the plain kernels built with the reference's own random stream (RT1W_RNG_REFERENCE) */
extern "C" int rt1w_internal_ref_blocks_per_cu(int stack_walk);
extern "C" int rt1w_internal_ref_launch(int stack_walk, const void* view, const void* frame, double* partial, unsigned long long* counters,
                                        int grid, hipStream_t stream);
extern "C" unsigned rt1w_internal_ref_sizeof(int what);
/* context_f32.hip: the kernels in single precision (RT1W_PRECISION_F32) and the f32 copies of the scene arrays */
extern "C" int rt1w_internal_f32_create(const void* nodes, uint32_t n_nodes, const void* lights, uint32_t n_lights, const void* materials,
                                        uint32_t n_materials, const void* textures, uint32_t n_textures, const void* perlin, uint32_t n_perlin,
                                        const void* view64, void** out);
extern "C" void rt1w_internal_f32_destroy(void* h);
extern "C" int rt1w_internal_f32_blocks_per_cu(int variant, int sorted); /* sorted 2: the pair-walk kernel of sphere scenes */
extern "C" int rt1w_internal_f32_pw(void* h, unsigned stack_cap); /* 1: the scene has f32 pair-walk records and fits `stack_cap` entries */
extern "C" unsigned rt1w_internal_f32_view(void* h, void* out, unsigned cap); /* bytes of the f32 RtSceneView (kernel argument) */
extern "C" int rt1w_internal_f32_launch(void* h, int variant, int sorted, const void* frame, double* partial, unsigned long long* counters, int grid,
                                        hipStream_t stream);

#include "rt_kernels.h"
#include "rt_walk_table.h"

namespace {

/* AABB slab test on the device, both forms, for tests: in[i] = {bb[6], o[3], d[3], t_min, t_max} */
__global__ void rt_debug_aabb_kernel(const double* in, int* out_literal, int* out_fast, unsigned long long n) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* p = in + i * 14;
    RtV3 o = rt_v3(p[6], p[7], p[8]);
    RtV3 inv = rt_inv3(rt_v3(p[9], p[10], p[11]));
    out_literal[i] = rt_aabb_hit(p, o, inv, p[12], p[13]) ? 1 : 0;
    const bool fe = rt_aabb_hit_fast<true>(p, o, inv, p[12], p[13]), fn = rt_aabb_hit_fast<false>(p, o, inv, p[12], p[13]);
    out_fast[i] = (fe == fn) ? (fe ? 1 : 0) : 2; /* 2: the two max/min forms disagree -- never equal to the literal's 0/1 */
}

__global__ void rt_debug_eval_kernel(int fn, const double* a, const double* b, double* out, unsigned long long n) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b[i], r = 0.0;
    switch (fn) {
        case 0: r = x / y; break;
        case 1: r = rt_sqrt(rt_abs(x)); break;
        case 2: r = rt_sin(x); break;
        case 3: r = rt_cos(x); break;
        case 4: r = rt_acos(x / (rt_abs(x) + 1.0)); break;
        case 5: r = rt_atan2(x, y); break;
        case 6: r = rt_log(rt_abs(y)); break;
        case 7: { RtRng g = rt_rng_pixel_sample(i, (uint32_t)rt_d2u(x), 0u); r = rt_gen_f64(g); } break;
        case 8: { RtRng g = rt_rng_pixel_sample(i, (uint32_t)rt_d2u(x), 0u); (void)rt_gen_f64(g); r = rt_gen_range(g, -1.0, 1.0); } break;
        default: break;
    }
    out[i] = r;
}

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    rt1w::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}

} // namespace

/* Everything one in-flight render needs.  Lane 0 serves the one-shot entries; rt1w_render_rows keeps two strips in
 * flight, one per lane, so that the next strip's workgroups fill the CUs as the previous strip's persistent kernel tails off
 * and its device->host copy runs under the other lane's tracing. */
struct RtLane {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double* d_partial = nullptr; size_t partial_bytes = 0;
    unsigned long long* d_counters = nullptr;
    unsigned long long* h_counters = nullptr; /* pinned */
    void* d_strip = nullptr; void* h_strip = nullptr; size_t strip_bytes = 0; /* rt1w_render_rows: device strip + pinned host strip */
};

struct rt1w_context {
    int device = 0;
    RtLane lane[2];
    hipEvent_t ev_first = nullptr;
    void* d_nodes = nullptr; void* d_lights = nullptr; void* d_materials = nullptr;
    void* d_textures = nullptr; void* d_perlin = nullptr; void* d_images = nullptr;
    RtSceneView view{};
    double* d_out = nullptr; size_t out_bytes = 0;
    int grid[RT_N_VARIANTS] = {};
    int grid_sphere_media[RT_N_VARIANTS] = {};
    int grid_ss[2][RT_N_VARIANTS] = {};
    int grid_ss_hc[2][RT_N_VARIANTS] = {}; /* the same kernels with the node cache (rt_render_kernel_ss_hc); 0: this context has no walk table */
    bool walk_table = false; uint32_t walk_table_first = 0;
    bool sphere_media = false; /* every medium of the scene is bounded by a bare Sphere: g_kernels_sphere_media serve */
    int grid_sorted[RT_N_VARIANTS] = {};
    int grid_cached[RT_N_VARIANTS] = {};
    int variant = 0;
    bool has_media = false, has_tex = false, has_msphere = false;
    uint32_t n_nodes = 0, scope_depth = 0;
    void* wf_state = nullptr; /* the wavefront form's own state (librt1w_lab.so: wavefront.hip), freed through its destroy hook */
    uint32_t stack_need = 0;
    int ref_grid[4] = {0, 0, 0, 0}; /* reference-stream kernels: sweep, stack walk, reordering V0, reordering every-feature */
    void* f32_scene = nullptr;   /* context_f32.hip: f32 copies of the scene arrays, built at the first f32 render */
    bool f32_tried = false;
    /* pair walk (rt_walk_pair.h): records of an eligible scene (sphere-only, variant 5), the kernel's grid */
    void* d_pw_inner = nullptr; void* d_pw_groups = nullptr;
    RtPwView pw{};
    bool pw_ok = false; std::string pw_why;
    int pw_grid = 0, pw_ss_grid = 0;
    /* host copies of the flat arrays the two opt-in modes convert on first use (a scene may be destroyed before its contexts) */
    std::vector<RtNode> h_nodes, h_lights; std::vector<RtMaterial> h_materials; std::vector<RtTexture> h_textures; std::vector<RtPerlin> h_perlin;
    int f32_grid[RT_N_VARIANTS][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; /* [variant][plain, reordering, pair walk] */
    /* scene-specialised kernel (jit.cpp): generated source (empty: scene not eligible), loaded module */
    std::string jit_src, jit_key;
    hipModule_t jit_mod = nullptr;
    hipFunction_t jit_fn = nullptr;
    int jit_grid = 0;
    int jit_block = RT_SORT_BLOCK; /* the specialised kernel's workgroup size = its sort domain (its __launch_bounds__; experiments build it for 512) */
    uint32_t jit_vgprs = 0;
    bool jit_failed = false; /* a compile was tried and failed: do not try again on this context */
    /* the same for RT1W_PRECISION_F32: loaded / compiled at the first f32 render of a specialised context */
    std::string jit32_src;
    hipModule_t jit32_mod = nullptr;
    hipFunction_t jit32_fn = nullptr;
    int jit32_grid = 0;
    bool jit32_tried = false, jit32_failed = false;
    std::string jit32_error;
};

typedef void (*render_kernel_t)(RtSceneView, RtFrame, double*, unsigned long long*);
static render_kernel_t const g_kernels[RT_N_VARIANTS] = {rt_render_kernel<RtCfgV0>, rt_render_kernel<RtCfgV1>,
                                                         rt_render_kernel<RtCfgV2>, rt_render_kernel<RtCfgV3>, rt_render_kernel<RtCfgV4>,
                                                         rt_render_kernel<RtCfgV5>};
/* the stack variants with media, for scenes whose media are all bounded by a bare Sphere (rt_flat.h: RtCfgSphereMedia) */
static render_kernel_t const g_kernels_sphere_media[RT_N_VARIANTS] = {nullptr, nullptr, nullptr, rt_render_kernel<RtCfgSphereMedia<RtCfgV3>>,
                                                                      rt_render_kernel<RtCfgSphereMedia<RtCfgV4>>, nullptr};
/* the stack-walk kernels with the finished paths reordered across the workgroup at the end of every slice (rt_render_ss_body): the default
 * for the scenes they cover; [1] = the sphere-media builds */
static render_kernel_t const g_kernels_ss[2][RT_N_VARIANTS] = {
    {nullptr, nullptr, rt_render_kernel_ss<RtCfgV2, RT_SS_CAP, 3>, rt_render_kernel_ss<RtCfgV3, RT_SS_CAP, 3>, nullptr, rt_render_kernel_ss<RtCfgV5, RT_SS_CAP, 3>},
    {nullptr, nullptr, nullptr, rt_render_kernel_ss<RtCfgSphereMedia<RtCfgV3>, RT_SS_CAP, 3>, rt_render_kernel_ss<RtCfgSphereMedia<RtCfgV4>, RT_SS_CAP, 3>, nullptr}};
/* ... and with the scene's most visited nodes in LDS (rt_walk_table.h): what a context with a walk table runs */
static render_kernel_t const g_kernels_ss_hc[2][RT_N_VARIANTS] = {
    {nullptr, nullptr, rt_render_kernel_ss_hc<RtCfgV2>, rt_render_kernel_ss_hc<RtCfgV3>, nullptr, rt_render_kernel_ss_hc<RtCfgV5>},
    {nullptr, nullptr, nullptr, rt_render_kernel_ss_hc<RtCfgSphereMedia<RtCfgV3>>, rt_render_kernel_ss_hc<RtCfgSphereMedia<RtCfgV4>>, nullptr}};
/* stack variants with the LDS node cache (scenes of <= RT_LDS_NODE_CAP nodes; opt-in: RT1W_LDS_NODES).  Measured on
 * random_scene: 464 Mpaths/s (80 KB LDS -> 2 waves/SIMD) against 486 for the plain variant at 3 waves/SIMD. */
static render_kernel_t const g_kernels_cached[RT_N_VARIANTS] = {nullptr, nullptr, rt_render_kernel<RtCfgV2, true>, rt_render_kernel<RtCfgV3, true>, nullptr, rt_render_kernel<RtCfgV5, true>};
/* the reordering kernel exists for the variants where it pays (measured): the sweep variants */
static render_kernel_t const g_kernels_sorted[RT_N_VARIANTS] = {rt_render_kernel_sorted<RtCfgV0>, rt_render_kernel_sorted<RtCfgV1>,
                                                                nullptr, nullptr, nullptr, nullptr}; /* stack variants: measured 0.55x (LDS for stack + exchange halves occupancy) */

namespace {

bool upload(void** dst, const void* src, size_t bytes) {
    *dst = nullptr;
    size_t alloc = bytes ? bytes : 16;
    if (!hip_ok(hipMalloc(dst, alloc), "hipMalloc(scene)")) return false;
    if (bytes && !hip_ok(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice), "hipMemcpy(scene)")) return false;
    return true;
}

/* the node array with one spare (zeroed) record behind it (the fused walk requests record e + 1 together with record e), and behind
 * that, at RT_WT_OFFSET, room for the scene's walk table (rt_walk_table.h; written by build_walk_table once the visits are counted) */
bool upload_nodes(void** dst, const std::vector<RtNode>& nodes) {
    const size_t n = nodes.size();
    std::vector<unsigned char> buf(RT_WT_OFFSET(n) + n * sizeof(RtNodeHot), 0);
    if (n) memcpy(buf.data(), nodes.data(), n * sizeof(RtNode));
    return upload(dst, buf.data(), buf.size());
}

/* The walk table of a stack-walk scene (rt_walk_table.h), ranked by MEASURED visits: a 128 x 128 x 1 render of the scene's own camera
 * with a kernel that counts the node fetches (rt_visit_count_kernel: a few ms, once per context).  Which nodes the kernels keep in LDS
 * changes where a record is read from and nothing else -- a scene whose table cannot be built, or whose walk needs more stack than the
 * cached kernels have, simply keeps the kernels without the cache. */
static_assert(RT_SS_HC_RECORDS == (int)RT_WT_CACHE_MAX, "the kernels cache exactly the records the table ranks");
bool build_walk_table(rt1w_context* c, const std::vector<RtNode>& nodes, uint32_t root, uint32_t stack_need, std::string& why) {
    const uint32_t n = (uint32_t)nodes.size();
    if (n == 0u) { why = "no nodes"; return false; }
    if (stack_need > (uint32_t)RT_SS_HC_CAP) { why = "the walk needs more than RT_SS_HC_CAP stack entries"; return false; }
    uint32_t* d_visits = nullptr;
    std::vector<uint32_t> visits(n, 0u);
    if (!hip_ok(hipMalloc((void**)&d_visits, (size_t)n * 4u), "hipMalloc(visits)")) return false;
    bool ok = hip_ok(hipMemset(d_visits, 0, (size_t)n * 4u), "hipMemset(visits)");
    if (ok) {
        RtFrame f;
        memset(&f, 0, sizeof f);
        f.width = f.tile_w = 128u; f.height = f.tile_h = 128u; f.spp = 1u; f.max_depth = 50u; f.chunk = 1u; f.n_chunks = 1u;
        hipLaunchKernelGGL(rt_visit_count_kernel, dim3(64), dim3(256), 0, c->lane[0].stream, c->view, f, d_visits);
        ok = hip_ok(hipGetLastError(), "rt_visit_count_kernel") &&
             hip_ok(hipMemcpyAsync(visits.data(), d_visits, (size_t)n * 4u, hipMemcpyDeviceToHost, c->lane[0].stream), "hipMemcpy(visits)") &&
             hip_ok(hipStreamSynchronize(c->lane[0].stream), "hipStreamSynchronize(visits)");
    }
    (void)hipFree(d_visits);
    if (!ok) { why = "the visit count failed"; return false; }
    RtWalkTable T;
    if (!rt_walk_table_build(nodes, root, visits.data(), T, why)) return false;
    if (!hip_ok(hipMemcpy((unsigned char*)c->d_nodes + RT_WT_OFFSET(n), T.rec.data(), (size_t)n * sizeof(RtNodeHot), hipMemcpyHostToDevice), "hipMemcpy(walk table)")) {
        why = "upload failed"; return false;
    }
    c->walk_table_first = T.n_first;
    return true;
}

/* RT1W_PRECISION_F32: the f32 copies of the scene arrays, at the first f32 render */
int ensure_f32_scene(rt1w_context* c) {
    if (c->f32_scene) return RT1W_OK;
    if (c->f32_tried) { rt1w::set_error("could not build the single-precision scene arrays"); return RT1W_ERR_DEVICE; }
    c->f32_tried = true;
    if (rt1w_internal_f32_create(c->h_nodes.data(), (uint32_t)c->h_nodes.size(), c->h_lights.data(), (uint32_t)c->h_lights.size(),
                                 c->h_materials.data(), (uint32_t)c->h_materials.size(), c->h_textures.data(), (uint32_t)c->h_textures.size(),
                                 c->h_perlin.data(), (uint32_t)c->h_perlin.size(), &c->view, &c->f32_scene) != 0) {
        c->f32_scene = nullptr;
        rt1w::set_error("could not build the single-precision scene arrays"); return RT1W_ERR_DEVICE;
    }
    return RT1W_OK;
}
int validate(const rt1w_context* c, const rt1w_render_params* p) {
    if (!c || !p) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (p->width < 2 || p->height < 2) { rt1w::set_error("width and height must be >= 2 (u,v divide by W-1, H-1; main.rs:968-969)"); return RT1W_ERR_INVALID; }
    if (p->tile_w == 0 || p->tile_h == 0 || (uint64_t)p->x0 + p->tile_w > p->width || (uint64_t)p->y0 + p->tile_h > p->height) {
        rt1w::set_error("tile outside the image"); return RT1W_ERR_INVALID;
    }
    if (p->spp == 0) { rt1w::set_error("spp must be > 0"); return RT1W_ERR_INVALID; }
    if ((p->strip_rows == 0) != (p->strip_period == 0) || p->strip_period < p->strip_rows) {
        rt1w::set_error("strip_rows / strip_period: both 0, or 0 < strip_rows <= strip_period"); return RT1W_ERR_INVALID;
    }
    if (p->precision != RT1W_PRECISION_F64 && p->precision != RT1W_PRECISION_F32) { rt1w::set_error("unknown precision"); return RT1W_ERR_UNSUPPORTED; }
    if (p->strip_rows) {
        const uint64_t last = (uint64_t)p->tile_h - 1u;
        const uint64_t j = (uint64_t)p->y0 + (last / p->strip_rows) * p->strip_period + last % p->strip_rows;
        if (j >= p->height) { rt1w::set_error("interleaved tile: last strip outside the image"); return RT1W_ERR_INVALID; }
    }
    if ((uint64_t)p->sample_offset + p->spp > 0xFFFFFFFFull) { rt1w::set_error("sample index overflow"); return RT1W_ERR_INVALID; }
    return RT1W_OK;
}

/* the shading-side leaf functions ON THE DEVICE, for known-answer tests of what no artefact of the reference pins:
 * mode 0 Texture::value(u, v, p) of texture `tex` (texture.rs:40-89); mode 1 Perlin::noise(p) and Perlin::turb(p, 7) of
 * Perlin table `tex` (perlin.rs:46-86); mode 2 sphere_uv(p) (math.rs:67-71).  in[i] = {u, v, p.x, p.y, p.z}. */
__global__ void rt_debug_texture_kernel(RtSceneView sc, int mode, uint32_t tex, const double* __restrict__ in, double* __restrict__ out, unsigned long long n) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double u = in[i * 5], v = in[i * 5 + 1];
    const RtV3 p = rt_v3(in[i * 5 + 2], in[i * 5 + 3], in[i * 5 + 4]);
    RtV3 r = rt_v3(0.0, 0.0, 0.0);
    if (mode == 0) r = rt_texture<RtCfgV1>(sc, tex, u, v, p);
    else if (mode == 1) { r.x = rt_perlin_noise(sc.perlin[tex], p); r.y = rt_perlin_turb(sc.perlin[tex], p, 7); }
    else { double uu, vv; rt_sphere_uv(p, uu, vv); r.x = uu; r.y = vv; }
    out[i * 3] = r.x; out[i * 3 + 1] = r.y; out[i * 3 + 2] = r.z;
}

__global__ void rt_init_counters_kernel(unsigned long long* ctr, unsigned long long next_item, uint32_t keep_segments) { ctr[0] = next_item; if (!keep_segments) ctr[1] = 0ull; }

bool lane_init(RtLane& l) {
    if (l.stream) return true;
    /* ordinary (blocking) streams: ordered after work the caller queued on the null stream, e.g. on the tensor handed to
     * rt1w_render_device; the two lanes still run concurrently with each other */
    return hip_ok(hipStreamCreate(&l.stream), "hipStreamCreate") &&
           hip_ok(hipEventCreate(&l.ev0), "hipEventCreate") && hip_ok(hipEventCreate(&l.ev1), "hipEventCreate") &&
           hip_ok(hipMalloc((void**)&l.d_counters, 2 * sizeof(unsigned long long)), "hipMalloc(counters)") &&
           hip_ok(hipHostMalloc((void**)&l.h_counters, 2 * sizeof(unsigned long long), hipHostMallocDefault), "hipHostMalloc(counters)");
}
void lane_destroy(RtLane& l) {
    if (l.d_partial) (void)hipFree(l.d_partial);
    if (l.d_counters) (void)hipFree(l.d_counters);
    if (l.h_counters) (void)hipHostFree(l.h_counters);
    if (l.d_strip) (void)hipFree(l.d_strip);
    if (l.h_strip) (void)hipHostFree(l.h_strip);
    if (l.ev0) (void)hipEventDestroy(l.ev0);
    if (l.ev1) (void)hipEventDestroy(l.ev1);
    if (l.stream) (void)hipStreamDestroy(l.stream);
    l = RtLane();
}

#ifndef RT_PARTIAL_BUDGET
#define RT_PARTIAL_BUDGET (8ull << 30)
#endif
struct RtLaunch { RtFrame f; unsigned long long npix; unsigned long long partial_budget = RT_PARTIAL_BUDGET; int variant, grid, block; bool sorted, cached, jit, ref, f32, pw = false, sphere_media = false, ss = false, hc = false; };
int specialise_f32(rt1w_context* c, bool allow_compile);

/* what the launch will need, without launching: frame, variant, launch shape */
int render_plan(rt1w_context* c, const rt1w_render_params* p, RtLaunch& L) {
    RtFrame& f = L.f;
    f.width = p->width; f.height = p->height;
    f.x0 = p->x0; f.y0 = p->y0; f.tile_w = p->tile_w; f.tile_h = p->tile_h;
    f.spp = p->spp; f.sample_offset = p->sample_offset; f.max_depth = p->max_depth;
    f.global_seed = p->global_seed;
    f.strip_rows = p->strip_rows; f.strip_period = p->strip_period;
    f.probe = (p->flags & RT1W_PROBE_COHERENT) ? 1u : 0u;
    /* stack-walk scenes: one sample per work item (include/rt1w.h: rt1w_scene_default_chunk), except for the wavefront form, whose passes are
     * per chunk; the reference stream sets its own below */
    f.chunk = p->chunk ? p->chunk : ((c->variant >= 2 && !(p->flags & RT1W_WAVEFRONT)) ? 1u : rt1w_default_chunk(p->tile_w, p->tile_h, p->spp));
    if (f.chunk > f.spp) f.chunk = f.spp;
    f.n_chunks = (f.spp + f.chunk - 1u) / f.chunk;
    L.npix = (unsigned long long)f.tile_w * f.tile_h;
    L.partial_budget = p->partial_mib ? ((unsigned long long)p->partial_mib << 20) : RT_PARTIAL_BUDGET;
    L.ref = false; L.f32 = false;
    if (p->precision == RT1W_PRECISION_F32) {
        if (p->flags & (RT1W_RNG_REFERENCE | RT1W_WAVEFRONT | RT1W_LDS_NODES)) { rt1w::set_error("RT1W_PRECISION_F32 has the default kernels only"); return RT1W_ERR_INVALID; }
        { const int rcf = ensure_f32_scene(c); if (rcf < 0) return rcf; }
        int v = c->variant == 4 ? 3 : c->variant; /* the order-aware variant exists in f64 only */
        if ((p->flags >> 8) & 0xFFu) {
            v = (int)((p->flags >> 8) & 0xFFu) - 1;
            if (v == 4 || !rt_variant_valid(v, c->n_nodes, c->has_media, c->has_tex, c->has_msphere, c->scope_depth)) {
                rt1w::set_error("forced kernel variant does not cover this scene's features"); return RT1W_ERR_INVALID;
            }
        }
        /* a context that runs a scene-specialised kernel in f64 uses the f32 build of that kernel too -- from the kernel
         * caches only: renders never compile (rt1w_context_specialise does, for both precisions) */
        if (!(p->flags & (RT1W_GENERIC | RT1W_UNSORTED)) && !((p->flags >> 8) & 0xFFu)) (void)specialise_f32(c, false);
        if (c->jit32_fn && !(p->flags & (RT1W_GENERIC | RT1W_UNSORTED)) && !((p->flags >> 8) & 0xFFu)) {
            L.f32 = true; L.jit = true; L.sorted = true; L.cached = false; L.variant = v; L.grid = c->jit32_grid; L.block = RT_SORT_BLOCK;
            return RT1W_OK;
        }
        const bool sorted = !(p->flags & RT1W_UNSORTED); /* v >= 2: the slice-end reordering of the stack walks */
        L.ss = sorted && v >= 2;
        /* sphere scenes: the pair walk (rt_walk_pair.h) in this precision too -- in the kernel that reorders the finished paths */
        L.pw = L.ss && v == 5 && !(p->flags & RT1W_CLASSIC_WALK) && rt1w_internal_f32_pw(c->f32_scene, (unsigned)RT_PW_SS_STACK) == 1;
        int& g = c->f32_grid[v][L.pw ? 2 : (sorted ? 1 : 0)];
        if (!g) {
            hipDeviceProp_t prop;
            if (!hip_ok(hipGetDeviceProperties(&prop, c->device), "hipGetDeviceProperties")) return RT1W_ERR_DEVICE;
            g = prop.multiProcessorCount * rt1w_internal_f32_blocks_per_cu(v, L.pw ? 2 : (sorted ? 1 : 0));
        }
        L.f32 = true; L.jit = false; L.sorted = sorted; L.cached = false; L.variant = v; L.grid = g; L.block = sorted ? RT_SORT_BLOCK : RT_BLOCK;
        return RT1W_OK;
    }
    if (p->flags & RT1W_RNG_REFERENCE) {
        /* the reference's own stream: one lane owns a pixel for all its samples (main.rs:964-989) */
        if (p->sample_offset != 0u) { rt1w::set_error("RT1W_RNG_REFERENCE: one stream per pixel, sample_offset must be 0"); return RT1W_ERR_INVALID; }
        f.chunk = f.spp; f.n_chunks = 1u; f.global_seed = 0u;
        const bool stack_walk = c->n_nodes > RT_SWEEP_MAX_NODES;
        /* small scenes: through the workgroup-level path reordering (the stream's state travels with the path) unless RT1W_UNSORTED */
        const int mode = stack_walk ? 1 : ((p->flags & RT1W_UNSORTED) ? 0 : (c->variant == 0 ? 2 : 3));
        if (!c->ref_grid[mode]) {
            hipDeviceProp_t prop;
            if (!hip_ok(hipGetDeviceProperties(&prop, c->device), "hipGetDeviceProperties")) return RT1W_ERR_DEVICE;
            c->ref_grid[mode] = prop.multiProcessorCount * rt1w_internal_ref_blocks_per_cu(mode);
        }
        L.ref = true; L.jit = false; L.sorted = mode >= 2; L.cached = false;
        L.variant = stack_walk ? 3 : (mode == 2 ? 0 : 1); L.grid = c->ref_grid[mode]; L.block = mode >= 2 ? RT_SORT_BLOCK : RT_BLOCK;
        return RT1W_OK;
    }
    int variant = c->variant;
    if ((p->flags >> 8) & 0xFFu) {
        variant = (int)((p->flags >> 8) & 0xFFu) - 1;
        if (!rt_variant_valid(variant, c->n_nodes, c->has_media, c->has_tex, c->has_msphere, c->scope_depth)) {
            rt1w::set_error("forced kernel variant does not cover this scene's features"); return RT1W_ERR_INVALID;
        }
    }
    L.variant = variant;
    L.jit = c->jit_fn != nullptr && !(p->flags & (RT1W_GENERIC | RT1W_UNSORTED | RT1W_LDS_NODES)) && !((p->flags >> 8) & 0xFFu);
    if (L.jit) { L.sorted = true; L.cached = false; L.grid = c->jit_grid; L.block = c->jit_block; return RT1W_OK; }
    /* sphere scenes: the pair walk (same frames, bit for bit), unless the caller asks for the one-entry-per-step walk */
    if (c->pw_ok && variant == 5 && !(p->flags & (RT1W_CLASSIC_WALK | RT1W_LDS_NODES | RT1W_WAVEFRONT))) {
        L.pw = true; L.sorted = false; L.cached = false; L.grid = c->pw_grid; L.block = RT_BLOCK;
        L.ss = c->pw_ss_grid > 0 && !(p->flags & RT1W_UNSORTED);
        if (L.ss) L.grid = c->pw_ss_grid;
        return RT1W_OK;
    }
    L.sorted = g_kernels_sorted[variant] != nullptr && !(p->flags & RT1W_UNSORTED);
    L.cached = !L.sorted && g_kernels_cached[variant] != nullptr && c->n_nodes <= RT_LDS_NODE_CAP && (p->flags & RT1W_LDS_NODES);
    L.sphere_media = !L.sorted && !L.cached && c->sphere_media && g_kernels_sphere_media[variant] != nullptr && !(p->flags & RT1W_CLASSIC_WALK);
    L.ss = !L.sorted && !L.cached && g_kernels_ss[L.sphere_media ? 1 : 0][variant] != nullptr && !(p->flags & RT1W_UNSORTED);
    L.hc = L.ss && c->walk_table && c->grid_ss_hc[L.sphere_media ? 1 : 0][variant] > 0 && !(p->flags & RT1W_NO_NODE_CACHE);
    L.grid = L.sorted ? c->grid_sorted[variant] : (L.cached ? c->grid_cached[variant] : (L.hc ? c->grid_ss_hc[L.sphere_media ? 1 : 0][variant] : (L.ss ? c->grid_ss[L.sphere_media ? 1 : 0][variant] : (L.sphere_media ? c->grid_sphere_media[variant] : c->grid[variant]))));
    L.block = L.sorted ? RT_SORT_BLOCK : RT_BLOCK;
    return RT1W_OK;
}

/* chunk partial sums: [chunk][pixel][3] f64.  A render whose partial sums would exceed this budget runs as several PASSES over
 * consecutive chunk ranges (= sample ranges): every pass is one launch of the persistent kernel into the same buffer, and the resolve
 * kernel adds the pass's chunks to the running pixel sums in order -- the same additions in the same order as one pass, so the same
 * bits whatever the budget (one sample per work item on a big frame at 10 000 spp would otherwise need 150 GB) */
uint32_t chunks_per_pass(const RtLaunch& L, unsigned long long bytes = RT_PARTIAL_BUDGET) {
    const unsigned long long per_chunk = L.npix * 3ull * sizeof(double);
    unsigned long long n = per_chunk ? bytes / per_chunk : 1ull;
    if (n < 1ull) n = 1ull;
    return n >= L.f.n_chunks ? L.f.n_chunks : (uint32_t)n;
}

int lane_reserve_partial(RtLane& l, const RtLaunch& L) {
    size_t need = (size_t)L.npix * chunks_per_pass(L, L.partial_budget) * 3 * sizeof(double);
    if (need > l.partial_bytes) {
        if (l.d_partial) (void)hipFree(l.d_partial);
        l.d_partial = nullptr; l.partial_bytes = 0;
        if (!hip_ok(hipMalloc((void**)&l.d_partial, need), "hipMalloc(partial sums)")) return RT1W_ERR_NOMEM;
        l.partial_bytes = need;
    }
    return RT1W_OK;
}

/* enqueue on the lane's stream: counters, trace kernel, resolve into d_out, counters back to pinned memory.  No host wait. */
int render_launch(rt1w_context* c, RtLane& l, const rt1w_render_params* p, const RtLaunch& L, double* d_out) {
    (void)hipEventRecord(l.ev0, l.stream);
    const uint32_t cpp = chunks_per_pass(L, l.partial_bytes < L.partial_budget ? l.partial_bytes : L.partial_budget); /* what the lane's buffer holds (lane_reserve_partial), within the caller's bound */
    const uint32_t n_pass = (L.f.n_chunks + cpp - 1u) / cpp;
    for (uint32_t pass = 0; pass < n_pass; ++pass) {
    /* this pass's chunks as a frame of their own: samples [c0 * chunk, ...) of the call, absolute sample indices through sample_offset */
    RtFrame PF = L.f;
    const uint32_t c0 = pass * cpp;
    PF.n_chunks = L.f.n_chunks - c0 < cpp ? L.f.n_chunks - c0 : cpp;
    PF.sample_offset = L.f.sample_offset + c0 * L.f.chunk;
    PF.spp = (L.f.spp - c0 * L.f.chunk < PF.n_chunks * L.f.chunk) ? L.f.spp - c0 * L.f.chunk : PF.n_chunks * L.f.chunk;
    hipLaunchKernelGGL(rt_init_counters_kernel, dim3(1), dim3(1), 0, l.stream, l.d_counters, (unsigned long long)L.grid * L.block, pass ? 1u : 0u);
    if (L.f32 && L.jit) {
        unsigned char view32[512];
        if (!c->f32_scene || rt1w_internal_f32_view(c->f32_scene, view32, sizeof view32) == 0u) { rt1w::set_error("single-precision scene missing"); return RT1W_ERR_DEVICE; }
        RtFrame frame = PF;
        double* partial = l.d_partial;
        unsigned long long* counters = l.d_counters;
        void* args[] = {view32, &frame, &partial, &counters};
        if (!hip_ok(hipModuleLaunchKernel(c->jit32_fn, (unsigned)L.grid, 1, 1, (unsigned)L.block, 1, 1, 0, l.stream, args, nullptr), "specialised f32 kernel launch")) return RT1W_ERR_DEVICE;
    } else if (L.f32) {
        if (!c->f32_scene || rt1w_internal_f32_launch(c->f32_scene, L.variant, L.pw ? 2 : (L.sorted ? 1 : 0), &PF, l.d_partial, l.d_counters, L.grid, l.stream) != 0) {
            rt1w::set_error("single-precision kernel launch failed"); return RT1W_ERR_DEVICE;
        }
    } else if (L.ref) {
        if (rt1w_internal_ref_sizeof(0) != sizeof(RtSceneView) || rt1w_internal_ref_sizeof(1) != sizeof(RtFrame) ||
            rt1w_internal_ref_launch(L.variant == 3 ? 1 : (L.sorted ? (L.variant == 0 ? 2 : 3) : 0), &c->view, &PF, l.d_partial, l.d_counters, L.grid, l.stream) != 0) {
            rt1w::set_error("reference-stream kernel launch failed"); return RT1W_ERR_DEVICE;
        }
    } else if (L.jit) {
        RtSceneView view = c->view;
        RtFrame frame = PF;
        double* partial = l.d_partial;
        unsigned long long* counters = l.d_counters;
        void* args[] = {&view, &frame, &partial, &counters};
        if (!hip_ok(hipModuleLaunchKernel(c->jit_fn, (unsigned)L.grid, 1, 1, (unsigned)L.block, 1, 1, 0, l.stream, args, nullptr), "specialised kernel launch")) return RT1W_ERR_DEVICE;
    } else if (L.pw) {
        if (L.ss) hipLaunchKernelGGL(rt_render_kernel_pw_ss<RtCfgV5>, dim3(L.grid), dim3(L.block), 0, l.stream, c->view, c->pw, PF, l.d_partial, l.d_counters);
        else hipLaunchKernelGGL(rt_render_kernel_pw<RtCfgV5>, dim3(L.grid), dim3(L.block), 0, l.stream, c->view, c->pw, PF, l.d_partial, l.d_counters);
    } else {
        hipLaunchKernelGGL(L.sorted ? g_kernels_sorted[L.variant] : (L.cached ? g_kernels_cached[L.variant] : (L.hc ? g_kernels_ss_hc[L.sphere_media ? 1 : 0][L.variant] : L.ss ? g_kernels_ss[L.sphere_media ? 1 : 0][L.variant] : (L.sphere_media ? g_kernels_sphere_media[L.variant] : g_kernels[L.variant]))),
                           dim3(L.grid), dim3(L.block), 0, l.stream, c->view, PF, l.d_partial, l.d_counters);
    }
    {
        unsigned int rb = 256;
        unsigned int rg = (unsigned int)((L.npix + rb - 1) / rb);
        hipLaunchKernelGGL(rt_resolve_kernel, dim3(rg), dim3(rb), 0, l.stream, l.d_partial, d_out, L.npix, PF.n_chunks,
                           L.f.spp, (p->flags & RT1W_OUT_SUM) ? 1u : 0u, (pass ? 1u : 0u) | (pass + 1u < n_pass ? 2u : 0u));
    }
    } /* passes */
    (void)hipEventRecord(l.ev1, l.stream);
    if (!hip_ok(hipGetLastError(), "kernel launch")) return RT1W_ERR_DEVICE;
    if (!hip_ok(hipMemcpyAsync(l.h_counters, l.d_counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, l.stream), "counter copy")) return RT1W_ERR_DEVICE;
    return RT1W_OK;
}

/* wait for everything enqueued on the lane and fill the stats of its last launch */
int render_finish(RtLane& l, const RtLaunch& L, rt1w_stats* stats) {
    if (!hip_ok(hipStreamSynchronize(l.stream), "render kernel")) return RT1W_ERR_DEVICE;
    if (stats) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, l.ev0, l.ev1);
        stats->paths = L.npix * L.f.spp;
        stats->segments = l.h_counters[1];
        stats->kernel_ms = ms;
        stats->chunk = L.f.chunk; stats->n_chunks = L.f.n_chunks;
        stats->grid = (uint32_t)L.grid; stats->block = (uint32_t)L.block;
        stats->variant = (uint32_t)L.variant; stats->sorted = ((L.sorted && !L.ss) ? 1u : 0u) | (L.cached ? 2u : 0u) | (L.jit ? 4u : 0u) | (L.ref ? 16u : 0u) | (L.f32 ? 32u : 0u) | (L.pw ? 128u : 0u) | (L.sphere_media ? 256u : 0u) | (L.ss ? 512u : 0u) | (L.hc ? 1024u : 0u);
    }
    return RT1W_OK;
}

/* load the specialised kernel of this context's scene: from the caches, or (allow_compile) from the compiler */
int specialise(rt1w_context* c, bool allow_compile, rt1w::JitInfo& info) {
    if (c->jit_fn) { info = rt1w::JitInfo(); info.key = c->jit_key; info.from_cache = true; return RT1W_OK; }
    if (c->jit_src.empty()) { rt1w::set_error("scene has more than RT_JIT_MAX_NODES nodes: no specialised kernel"); return RT1W_ERR_UNSUPPORTED; }
    std::vector<char> code;
    int rc = rt1w::jit_get_code(c->jit_src, allow_compile, code, info);
    c->jit_key = info.key;
    if (rc < 0) { rt1w::set_error("specialised kernel: " + info.message); return rc; }
    if (!hip_ok(hipModuleLoadData(&c->jit_mod, code.data()), "hipModuleLoadData(specialised kernel)")) {
        c->jit_mod = nullptr;
        if (!info.from_cache) return RT1W_ERR_DEVICE;
        /* a cached object the driver refuses (truncated, foreign toolchain): forget it and, if allowed, compile afresh */
        rt1w::jit_invalidate(info);
        if (!allow_compile) return RT1W_ERR_DEVICE;
        rc = rt1w::jit_get_code(c->jit_src, true, code, info, true);
        if (rc < 0) { rt1w::set_error("specialised kernel: " + info.message); return rc; }
        if (!hip_ok(hipModuleLoadData(&c->jit_mod, code.data()), "hipModuleLoadData(specialised kernel)")) { c->jit_mod = nullptr; return RT1W_ERR_DEVICE; }
    }
    hipFunction_t fn = nullptr;
    if (!hip_ok(hipModuleGetFunction(&fn, c->jit_mod, "rt_jit_sorted"), "hipModuleGetFunction")) {
        (void)hipModuleUnload(c->jit_mod); c->jit_mod = nullptr; return RT1W_ERR_DEVICE;
    }
    int per_cu = 0, max_threads = 0;
    c->jit_block = RT_SORT_BLOCK;
    if (hipFuncGetAttribute(&max_threads, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, fn) == hipSuccess && (max_threads == 512 || max_threads == 128)) c->jit_block = max_threads;
    if (!hip_ok(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, c->jit_block, 0), "occupancy query")) per_cu = 1;
    if (per_cu < 1) per_cu = 1;
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, c->device), "hipGetDeviceProperties")) { (void)hipModuleUnload(c->jit_mod); c->jit_mod = nullptr; return RT1W_ERR_DEVICE; }
    int vg = 0;
    if (hipFuncGetAttribute(&vg, HIP_FUNC_ATTRIBUTE_NUM_REGS, fn) == hipSuccess) c->jit_vgprs = (uint32_t)vg;
    c->jit_grid = prop.multiProcessorCount * per_cu;
    c->jit_fn = fn;
    return RT1W_OK;
}

/* the f32 build of the scene-specialised kernel (RT1W_PRECISION_F32): loaded from the kernel caches, compiled only when
 * `allow_compile` (rt1w_context_specialise).  A failure is remembered with its reason (rt1w_last_error at specialise time). */
int specialise_f32(rt1w_context* c, bool allow_compile) {
    if (c->jit32_fn) return RT1W_OK;
    if (!c->jit_fn || c->jit32_src.empty()) return RT1W_ERR_STATE;
    if (c->jit32_tried && !allow_compile) return RT1W_ERR_STATE;
    if (c->jit32_failed) { rt1w::set_error("f32 specialised kernel: " + c->jit32_error); return RT1W_ERR_DEVICE; }
    c->jit32_tried = true;
    std::vector<char> code;
    rt1w::JitInfo info;
    int rc = rt1w::jit_get_code(c->jit32_src, allow_compile && !getenv("RT1W_NO_JIT"), code, info);
    if (rc < 0) {
        if (allow_compile) { c->jit32_failed = true; c->jit32_error = info.message; rt1w::set_error("f32 specialised kernel: " + info.message); }
        return rc;
    }
    hipFunction_t fn = nullptr;
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (!hip_ok(hipModuleLoadData(&c->jit32_mod, code.data()), "hipModuleLoadData(f32 specialised kernel)") ||
        !hip_ok(hipModuleGetFunction(&fn, c->jit32_mod, "rt_jit_sorted"), "hipModuleGetFunction(f32 specialised kernel)") ||
        !hip_ok(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, RT_SORT_BLOCK, 0), "occupancy query") ||
        !hip_ok(hipGetDeviceProperties(&prop, c->device), "hipGetDeviceProperties")) {
        if (c->jit32_mod) { (void)hipModuleUnload(c->jit32_mod); c->jit32_mod = nullptr; }
        if (info.from_cache) rt1w::jit_invalidate(info);
        c->jit32_failed = true; c->jit32_error = rt1w_last_error();
        return RT1W_ERR_DEVICE;
    }
    c->jit32_fn = fn;
    c->jit32_grid = prop.multiProcessorCount * (per_cu < 1 ? 1 : per_cu);
    return RT1W_OK;
}

/* ---- wavefront form: lives in librt1w_lab.so (wavefront.hip), which registers itself here when it is loaded ---- */
static rt1w_wf_render_fn g_wf_render = nullptr;
static rt1w_wf_destroy_fn g_wf_destroy = nullptr;

int render_wavefront(rt1w_context* c, const rt1w_render_params* p, const RtLaunch& L, double* d_out, rt1w_stats* stats) {
    if (!g_wf_render) {
        rt1w::set_error("RT1W_WAVEFRONT: the wavefront form (a measured, slower opt-in) is part of librt1w_lab.so -- load that library, it registers itself");
        return RT1W_ERR_UNSUPPORTED;
    }
    RtLane& l = c->lane[0];
    rt1w_wf_call k;
    memset(&k, 0, sizeof k);
    k.device = c->device; k.view = &c->view; k.frame = &L.f; k.npix = L.npix; k.variant = L.variant;
    k.n_nodes = c->n_nodes; k.scope_depth = c->scope_depth; k.stack_need = c->stack_need;
    k.h_nodes = c->h_nodes.data(); k.n_h_nodes = (uint32_t)c->h_nodes.size();
    k.stream = (void*)l.stream; k.d_partial = l.d_partial; k.state = &c->wf_state; k.stats = stats;
    (void)hipEventRecord(l.ev0, l.stream);
    const int rc = g_wf_render(&k);                /* enqueues generate / trace / shade / finish / chunk sums on the lane's stream */
    if (rc < 0) return rc;
    {
        unsigned int rb = 256;
        unsigned int rg = (unsigned int)((L.npix + rb - 1) / rb);
        hipLaunchKernelGGL(rt_resolve_kernel, dim3(rg), dim3(rb), 0, l.stream, l.d_partial, d_out, L.npix, L.f.n_chunks, L.f.spp,
                           (p->flags & RT1W_OUT_SUM) ? 1u : 0u, 0u);
    }
    (void)hipEventRecord(l.ev1, l.stream);
    if (!hip_ok(hipGetLastError(), "kernel launch") || !hip_ok(hipStreamSynchronize(l.stream), "wavefront render")) return RT1W_ERR_DEVICE;
    if (stats) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, l.ev0, l.ev1);
        stats->paths = L.npix * L.f.spp; stats->segments = k.h_segments ? *k.h_segments : 0ull; stats->kernel_ms = ms;
        stats->chunk = L.f.chunk; stats->n_chunks = L.f.n_chunks;
    }
    return RT1W_OK;
}

int render_common(rt1w_context* c, const rt1w_render_params* p, double* d_out, rt1w_stats* stats) {
    /* a render this long repays the 1-8 s of the compiler: 2^35 paths where the gain is ~1.3x (scenes the generic sweep
     * handles), 2^32 where it is 1.5-2.3x (scenes the generic code hands to the stack walk).  RT1W_NO_JIT: never compile
     * behind the caller's back */
    if (!c->jit_fn && !c->jit_failed && !c->jit_src.empty() && !(p->flags & (RT1W_GENERIC | RT1W_UNSORTED)) &&
        (unsigned long long)p->tile_w * p->tile_h * p->spp >= (c->n_nodes <= RT_SWEEP_MAX_NODES ? (1ull << 35) : (1ull << 32)) &&
        !getenv("RT1W_NO_JIT")) {
        rt1w::JitInfo info;
        if (specialise(c, true, info) < 0) c->jit_failed = true;
    }
    RtLaunch L;
    int rc = render_plan(c, p, L);
    if (rc < 0) return rc;
    RtLane& l = c->lane[0];
    if ((rc = lane_reserve_partial(l, L)) < 0) return rc;
    if ((p->flags & RT1W_WAVEFRONT) && !L.jit && !L.sorted && !L.ref && !L.f32) {
        if (chunks_per_pass(L, L.partial_budget) < L.f.n_chunks) { rt1w::set_error("RT1W_WAVEFRONT renders in one pass: its chunk partial sums must fit the budget (rt1w_render_params.partial_mib)"); return RT1W_ERR_UNSUPPORTED; }
        return render_wavefront(c, p, L, d_out, stats);
    }
    if ((rc = render_launch(c, l, p, L, d_out)) < 0) return rc;
    return render_finish(l, L, stats);
}

} // namespace

extern "C" {


int rt1w_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt1w_context_create(int device_id, const rt1w_scene* s, rt1w_context** out) {
    if (!s || !out) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (!s->committed) { rt1w::set_error("scene not committed"); return RT1W_ERR_STATE; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        rt1w::set_error("no HIP device: librt1w has no CPU render path"); return RT1W_ERR_DEVICE;
    }
    if (device_id < 0 || device_id >= n) { rt1w::set_error("bad device id"); return RT1W_ERR_INVALID; }
    if (!hip_ok(hipSetDevice(device_id), "hipSetDevice")) return RT1W_ERR_DEVICE;
    rt1w_context* c = new (std::nothrow) rt1w_context();
    if (!c) { rt1w::set_error("out of memory"); return RT1W_ERR_NOMEM; }
    c->device = device_id;
    bool ok = lane_init(c->lane[0]) &&
              upload_nodes(&c->d_nodes, s->flat_nodes) &&
              upload(&c->d_lights, s->flat_lights.data(), s->flat_lights.size() * sizeof(RtNode)) &&
              upload(&c->d_materials, s->materials.data(), s->materials.size() * sizeof(RtMaterial)) &&
              upload(&c->d_textures, s->textures.data(), s->textures.size() * sizeof(RtTexture)) &&
              upload(&c->d_perlin, s->perlin.data(), s->perlin.size() * sizeof(RtPerlin)) &&
              upload(&c->d_images, s->images.data(), s->images.size());
    if (!ok) { rt1w_context_destroy(c); return RT1W_ERR_DEVICE; }
    RtSceneView& v = c->view;
    v.nodes = (const RtNode*)c->d_nodes; v.lights = (const RtNode*)c->d_lights;
    v.materials = (const RtMaterial*)c->d_materials; v.textures = (const RtTexture*)c->d_textures;
    v.perlin = (const RtPerlin*)c->d_perlin; v.images = (const uint8_t*)c->d_images;
    v.root = s->flat_root; v.n_nodes = (uint32_t)s->flat_nodes.size(); v.n_lights = (uint32_t)s->flat_lights.size();
    v.n_materials = (uint32_t)s->materials.size(); v.n_textures = (uint32_t)s->textures.size(); v.pad = 0;
    v.camera = s->camera; v.background = s->background;
    /* persistent grid: as many blocks as are resident at once */
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, device_id), "hipGetDeviceProperties")) { rt1w_context_destroy(c); return RT1W_ERR_DEVICE; }
    for (int v = 0; v < RT_N_VARIANTS; ++v) {
        int per_cu = 0;
        if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, g_kernels[v], RT_BLOCK, 0), "occupancy query")) {
            rt1w_context_destroy(c); return RT1W_ERR_DEVICE;
        }
        if (per_cu < 1) per_cu = 1;
        c->grid[v] = prop.multiProcessorCount * per_cu;
        if (g_kernels_sphere_media[v]) {
            per_cu = 0;
            if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, g_kernels_sphere_media[v], RT_BLOCK, 0), "occupancy query")) {
                rt1w_context_destroy(c); return RT1W_ERR_DEVICE;
            }
            if (per_cu < 1) per_cu = 1;
            c->grid_sphere_media[v] = prop.multiProcessorCount * per_cu;
        }
        for (int sm = 0; sm < 2; ++sm) {
            if (!g_kernels_ss[sm][v]) continue;
            per_cu = 0;
            if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, g_kernels_ss[sm][v], RT_BLOCK, 0), "occupancy query")) {
                rt1w_context_destroy(c); return RT1W_ERR_DEVICE;
            }
            if (per_cu < 1) per_cu = 1;
            c->grid_ss[sm][v] = prop.multiProcessorCount * per_cu;
        }
        if (g_kernels_cached[v]) {
            per_cu = 0;
            if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, g_kernels_cached[v], RT_BLOCK, 0), "occupancy query")) {
                rt1w_context_destroy(c); return RT1W_ERR_DEVICE;
            }
            if (per_cu < 1) per_cu = 1;
            c->grid_cached[v] = prop.multiProcessorCount * per_cu;
        }
        if (g_kernels_sorted[v]) {
            per_cu = 0;
            if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, g_kernels_sorted[v], RT_SORT_BLOCK, 0), "occupancy query")) {
                rt1w_context_destroy(c); return RT1W_ERR_DEVICE;
            }
            if (per_cu < 1) per_cu = 1;
            c->grid_sorted[v] = prop.multiProcessorCount * per_cu;
        }
    }
    c->has_media = s->has_media; c->has_tex = s->has_tex; c->has_msphere = s->has_msphere;
    c->sphere_media = s->has_media && s->media_bare_spheres;
    c->n_nodes = (uint32_t)s->flat_nodes.size();
    c->scope_depth = s->scope_depth;
    c->stack_need = s->stack_need;
    c->variant = rt_pick_variant(c->n_nodes, c->has_media, c->has_tex, c->has_msphere, c->scope_depth, s->walk_annotated != 0u);
    if (c->variant == 5 && s->stack_need + 1u <= (uint32_t)RT_PW_STACK) {
        /* a sphere scene: records of the pair walk (rt_walk_pair.h); a scene outside its scope keeps the one-entry-per-step walk */
        std::vector<RtPwInner> pin; std::vector<RtPwGroup> pgr;
        if (rt_pw_build(s->flat_nodes, s->flat_root, pin, pgr, c->pw, c->pw_why)) {
            int per_cu = 0;
            if (!upload(&c->d_pw_inner, pin.data(), pin.size() * sizeof(RtPwInner)) || !upload(&c->d_pw_groups, pgr.data(), pgr.size() * sizeof(RtPwGroup)) ||
                !hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rt_render_kernel_pw<RtCfgV5>, RT_BLOCK, 0), "occupancy query")) {
                rt1w_context_destroy(c); return RT1W_ERR_DEVICE;
            }
            c->pw.inner = (const RtPwInner*)c->d_pw_inner; c->pw.groups = (const RtPwGroup*)c->d_pw_groups;
            c->pw_grid = prop.multiProcessorCount * (per_cu < 1 ? 1 : per_cu);
            c->pw_ok = true;
            if (s->stack_need + 1u <= (uint32_t)RT_PW_SS_STACK) { /* shallow enough for the kernel that also reorders the finished paths */
                per_cu = 0;
                if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rt_render_kernel_pw_ss<RtCfgV5>, RT_BLOCK, 0), "occupancy query")) {
                    rt1w_context_destroy(c); return RT1W_ERR_DEVICE;
                }
                c->pw_ss_grid = prop.multiProcessorCount * (per_cu < 1 ? 1 : per_cu);
            }
        }
    } else c->pw_why = "not a wrapper-free, media-free scene of more than 64 nodes, or its tree is deeper than the pair walk's stack";
    {
        /* the walk table and the kernels that keep its head in LDS: what a stack-walk scene's renders run (sphere scenes run the pair
         * walk by default; the table serves their one-entry-per-step renders, RT1W_CLASSIC_WALK).  Built for every scene: a small
         * scene's renders with a forced stack-walk variant (tests) go through it as well */
        std::string why;
        c->walk_table = build_walk_table(c, s->flat_nodes, s->flat_root, s->stack_need, why);
        if (c->walk_table) {
            for (int sm = 0; sm < 2; ++sm)
                for (int v = 0; v < RT_N_VARIANTS; ++v) {
                    if (!g_kernels_ss_hc[sm][v]) continue;
                    int per_cu = 0;
                    if (!hip_ok(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, g_kernels_ss_hc[sm][v], RT_BLOCK, 0), "occupancy query")) {
                        rt1w_context_destroy(c); return RT1W_ERR_DEVICE;
                    }
                    c->grid_ss_hc[sm][v] = prop.multiProcessorCount * (per_cu < 1 ? 1 : per_cu);
                }
        }
    }
    /* the opt-in modes' own data (f32 scene arrays, the wavefront form's walk records) are built at their first use:
     * ensure_f32_scene; the wavefront form's in librt1w_lab.so */
    c->h_nodes = s->flat_nodes; c->h_lights = s->flat_lights; c->h_materials = s->materials; c->h_textures = s->textures; c->h_perlin = s->perlin;
    if (rt1w::jit_eligible(*s)) {
        c->jit_src = rt1w::jit_source(*s);
        c->jit32_src = rt1w::jit_source(*s, true);
        rt1w::JitInfo info;
        (void)specialise(c, false, info); /* a cache hit is used from the first render on; a miss costs nothing */
    }
    *out = c;
    return RT1W_OK;
}

void rt1w_context_destroy(rt1w_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    void* bufs[] = {c->d_nodes, c->d_lights, c->d_materials, c->d_textures, c->d_perlin, c->d_images, c->d_out};
    if (c->wf_state && g_wf_destroy) g_wf_destroy(c->wf_state);
    for (void* b : bufs) if (b) (void)hipFree(b);
    rt1w_internal_f32_destroy(c->f32_scene);
    if (c->d_pw_inner) (void)hipFree(c->d_pw_inner);
    if (c->d_pw_groups) (void)hipFree(c->d_pw_groups);
    lane_destroy(c->lane[0]);
    lane_destroy(c->lane[1]);
    if (c->jit_mod) (void)hipModuleUnload(c->jit_mod);
    if (c->jit32_mod) (void)hipModuleUnload(c->jit32_mod);
    if (c->ev_first) (void)hipEventDestroy(c->ev_first);
    delete c;
}

int rt1w_context_specialise(rt1w_context* c, uint32_t flags, rt1w_specialise_info* out) {
    if (!c) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return RT1W_ERR_DEVICE;
    rt1w::JitInfo info;
    const bool had = c->jit_fn != nullptr;
    int rc = specialise(c, !(flags & RT1W_SPECIALISE_CACHED_ONLY), info);
    if (rc == RT1W_OK && c->jit_fn) (void)specialise_f32(c, !(flags & RT1W_SPECIALISE_CACHED_ONLY)); /* optional: a failure leaves f32 renders on the generic kernels */
    if (out) {
        memset(out, 0, sizeof *out);
        snprintf(out->key, sizeof out->key, "%s", c->jit_key.c_str());
        out->active = c->jit_fn ? 1u : 0u;
        out->from_cache = (had || info.from_cache) ? 1u : 0u;
        out->compile_ms = info.compile_ms;
        out->grid = (uint32_t)c->jit_grid; out->vgprs = c->jit_vgprs;
    }
    return rc;
}

int rt1w_render_device(rt1w_context* c, const rt1w_render_params* p, void* d_out_rgb, rt1w_stats* stats) {
    int rc = validate(c, p);
    if (rc < 0) return rc;
    if (!d_out_rgb) { rt1w::set_error("null output"); return RT1W_ERR_INVALID; }
    if (p->flags & RT1W_OUT_FRAME) { rt1w::set_error("RT1W_OUT_FRAME is a host-output mode (rt1w_render)"); return RT1W_ERR_INVALID; }
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return RT1W_ERR_DEVICE;
    auto t0 = std::chrono::steady_clock::now();
    rc = render_common(c, p, (double*)d_out_rgb, stats);
    if (rc == RT1W_OK && stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int rt1w_render_u8(rt1w_context* c, const rt1w_render_params* p, uint8_t* out_rgb8, rt1w_stats* stats) {
    int rc = validate(c, p);
    if (rc < 0) return rc;
    if (!out_rgb8) { rt1w::set_error("null output"); return RT1W_ERR_INVALID; }
    if (p->flags & RT1W_OUT_SUM) { rt1w::set_error("RT1W_OUT_SUM has no 8-bit form"); return RT1W_ERR_INVALID; }
    if ((p->flags & RT1W_OUT_FRAME) || p->strip_rows) { rt1w::set_error("rt1w_render_u8 takes contiguous tiles only"); return RT1W_ERR_INVALID; }
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return RT1W_ERR_DEVICE;
    auto t0 = std::chrono::steady_clock::now();
    size_t npix = (size_t)p->tile_w * p->tile_h;
    size_t bytes = npix * 3 * sizeof(double) + npix * 3; /* framebuffer + quantised image behind it */
    if (bytes > c->out_bytes) {
        if (c->d_out) (void)hipFree(c->d_out);
        c->d_out = nullptr; c->out_bytes = 0;
        if (!hip_ok(hipMalloc((void**)&c->d_out, bytes), "hipMalloc(framebuffer)")) return RT1W_ERR_NOMEM;
        c->out_bytes = bytes;
    }
    rc = render_common(c, p, c->d_out, stats);
    if (rc < 0) return rc;
    uint8_t* d_u8 = reinterpret_cast<uint8_t*>(c->d_out + npix * 3);
    hipLaunchKernelGGL(rt_quantize_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, c->lane[0].stream, c->d_out, d_u8, p->tile_w, p->tile_h);
    if (!hip_ok(hipMemcpyAsync(out_rgb8, d_u8, npix * 3, hipMemcpyDeviceToHost, c->lane[0].stream), "quantised image copy") ||
        !hip_ok(hipStreamSynchronize(c->lane[0].stream), "quantise kernel")) return RT1W_ERR_DEVICE;
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT1W_OK;
}

int rt1w_render_rows(rt1w_context* c, const rt1w_render_params* p, uint32_t strip_rows, int format, void* out,
                     rt1w_progress_fn progress, void* user, rt1w_stats* stats) {
    int rc = validate(c, p);
    if (rc < 0) return rc;
    if (!out) { rt1w::set_error("null output"); return RT1W_ERR_INVALID; }
    if (format != RT1W_ROWS_F64 && format != RT1W_ROWS_U8) { rt1w::set_error("unknown output format"); return RT1W_ERR_INVALID; }
    if (format == RT1W_ROWS_U8 && (p->flags & RT1W_OUT_SUM)) { rt1w::set_error("RT1W_OUT_SUM has no 8-bit form"); return RT1W_ERR_INVALID; }
    if ((p->flags & RT1W_OUT_FRAME) || p->strip_rows) { rt1w::set_error("rt1w_render_rows takes contiguous tiles only"); return RT1W_ERR_INVALID; }
    if (p->flags & RT1W_WAVEFRONT) { rt1w::set_error("rt1w_render_rows runs the persistent kernels only (RT1W_WAVEFRONT is a one-shot form)"); return RT1W_ERR_UNSUPPORTED; }
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return RT1W_ERR_DEVICE;
    auto t0 = std::chrono::steady_clock::now();
    const uint32_t H = p->tile_h, W = p->tile_w;
    const uint32_t tile_chunk = p->chunk ? p->chunk : ((c->variant >= 2) ? 1u : rt1w_default_chunk(W, H, p->spp)); /* the whole tile's chunking (rt1w_scene_default_chunk) */
    if (strip_rows == 0) {
        /* about 16 strips, but never so thin that a strip has fewer than ~4M work items (pixel x sample chunk): the
         * persistent kernel needs that many to keep its tail short, and the chunking is the whole tile's by contract */
        const uint64_t n_chunks = (p->spp + (uint64_t)(tile_chunk < p->spp ? tile_chunk : p->spp) - 1u) / (tile_chunk < p->spp ? tile_chunk : p->spp);
        const uint64_t min_rows = ((4ull << 20) + (uint64_t)W * n_chunks - 1u) / ((uint64_t)W * n_chunks);
        uint64_t rows = (H + 15u) / 16u;
        if (rows < min_rows) rows = min_rows;
        rows = (rows + 7u) & ~7ull;
        strip_rows = rows > H ? H : (uint32_t)rows;
    }
    if (strip_rows > H) strip_rows = H;
    /* one strip: f64 means, and behind them the quantised bytes when asked for */
    const size_t strip_px = (size_t)strip_rows * W;
    const size_t f64_bytes = strip_px * 3 * sizeof(double);
    const size_t dev_bytes = f64_bytes + (format == RT1W_ROWS_U8 ? strip_px * 3 : 0);
    rt1w_render_params sp = *p;
    sp.chunk = tile_chunk; /* same sums, same bits */
    sp.tile_h = strip_rows;
    RtLaunch plan;
    if ((rc = render_plan(c, &sp, plan)) < 0) return rc;
    /* all allocation up front: hipMalloc/hipFree in the loop would serialise the two lanes */
    if (!c->ev_first && !hip_ok(hipEventCreate(&c->ev_first), "hipEventCreate")) return RT1W_ERR_DEVICE;
    for (int k = 0; k < 2; ++k) {
        RtLane& l = c->lane[k];
        if (!lane_init(l)) return RT1W_ERR_DEVICE;
        if ((rc = lane_reserve_partial(l, plan)) < 0) return rc;
        if (dev_bytes > l.strip_bytes) {
            if (l.d_strip) (void)hipFree(l.d_strip);
            if (l.h_strip) (void)hipHostFree(l.h_strip);
            l.d_strip = l.h_strip = nullptr; l.strip_bytes = 0;
            if (!hip_ok(hipMalloc(&l.d_strip, dev_bytes), "hipMalloc(strip)") ||
                !hip_ok(hipHostMalloc(&l.h_strip, dev_bytes, hipHostMallocDefault), "hipHostMalloc(strip)")) return RT1W_ERR_NOMEM;
            l.strip_bytes = dev_bytes;
        }
    }
    const uint32_t n_strips = (H + strip_rows - 1u) / strip_rows;
    struct Flight { RtLaunch L; uint32_t top, rows; } fl[2];
    rt1w_stats total; memset(&total, 0, sizeof total);
    uint32_t rows_done = 0;
    /* strip i goes to lane i & 1: trace, resolve, (quantise,) copy to the lane's pinned strip -- all on the lane's stream */
    auto launch = [&](uint32_t i) -> int {
        RtLane& l = c->lane[i & 1u];
        Flight& F = fl[i & 1u];
        F.top = i * strip_rows;
        F.rows = (H - F.top < strip_rows) ? H - F.top : strip_rows;
        sp.tile_h = F.rows;
        sp.y0 = p->y0 + (H - F.top - F.rows);
        int r = render_plan(c, &sp, F.L);
        if (r < 0) return r;
        if (i == 0) (void)hipEventRecord(c->ev_first, l.stream);
        if ((r = render_launch(c, l, &sp, F.L, (double*)l.d_strip)) < 0) return r;
        const uint8_t* src = (const uint8_t*)l.d_strip;
        const size_t npix = (size_t)F.rows * W;
        if (format == RT1W_ROWS_U8) {
            uint8_t* d_u8 = (uint8_t*)l.d_strip + f64_bytes;
            hipLaunchKernelGGL(rt_quantize_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, l.stream,
                               (const double*)l.d_strip, d_u8, W, F.rows);
            src = d_u8;
        }
        const size_t nbytes = format == RT1W_ROWS_U8 ? npix * 3 : npix * 3 * sizeof(double);
        if (!hip_ok(hipMemcpyAsync(l.h_strip, src, nbytes, hipMemcpyDeviceToHost, l.stream), "strip copy")) return RT1W_ERR_DEVICE;
        return RT1W_OK;
    };
    /* waits for strip i, lands it in the caller's buffer and reports it; > 0 = the callback asked to stop */
    auto land = [&](uint32_t i) -> int {
        RtLane& l = c->lane[i & 1u];
        Flight& F = fl[i & 1u];
        rt1w_stats st;
        int r = render_finish(l, F.L, &st);
        if (r < 0) return r;
        total.paths += st.paths; total.segments += st.segments;
        total.chunk = st.chunk; total.n_chunks = st.n_chunks; total.grid = st.grid; total.block = st.block;
        total.variant = st.variant; total.sorted = st.sorted;
        if (i + 1u == n_strips) { float ms = 0.f; (void)hipEventElapsedTime(&ms, c->ev_first, l.ev1); total.kernel_ms = ms; }
        if (format == RT1W_ROWS_U8) memcpy((uint8_t*)out + (size_t)F.top * W * 3, l.h_strip, (size_t)F.rows * W * 3);
        else memcpy((double*)out + (size_t)(H - F.top - F.rows) * W * 3, l.h_strip, (size_t)F.rows * W * 3 * sizeof(double));
        rows_done += F.rows;
        return (progress && progress(user, rows_done, H) != 0) ? 1 : 0;
    };
    if ((rc = launch(0)) < 0) return rc;
    for (uint32_t i = 0; i < n_strips; ++i) {
        if (i + 1u < n_strips && (rc = launch(i + 1u)) < 0) { (void)hipStreamSynchronize(c->lane[i & 1u].stream); return rc; }
        rc = land(i);
        if (rc != 0) {
            /* error or cancel: the strip already in flight on the other lane is left to finish, unreported */
            if (i + 1u < n_strips) (void)hipStreamSynchronize(c->lane[(i + 1u) & 1u].stream);
            if (rc < 0) return rc;
            if (i + 1u == n_strips) break; /* asked to stop after the last strip: nothing left to stop */
            rt1w::set_error("cancelled by the progress callback");
            return RT1W_ERR_CANCELLED;
        }
    }
    if (stats) {
        *stats = total; /* kernel_ms: first strip's start to last strip's end on the device (the strips overlap) */
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT1W_OK;
}

int rt1w_render(rt1w_context* c, const rt1w_render_params* p, double* out_rgb, rt1w_stats* stats) {
    int rc = validate(c, p);
    if (rc < 0) return rc;
    if (!out_rgb) { rt1w::set_error("null output"); return RT1W_ERR_INVALID; }
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return RT1W_ERR_DEVICE;
    auto t0 = std::chrono::steady_clock::now();
    size_t bytes = (size_t)p->tile_w * p->tile_h * 3 * sizeof(double);
    if (bytes > c->out_bytes) {
        if (c->d_out) (void)hipFree(c->d_out);
        c->d_out = nullptr; c->out_bytes = 0;
        if (!hip_ok(hipMalloc((void**)&c->d_out, bytes), "hipMalloc(framebuffer)")) return RT1W_ERR_NOMEM;
        c->out_bytes = bytes;
    }
    rc = render_common(c, p, c->d_out, stats);
    if (rc < 0) return rc;
    if (p->flags & RT1W_OUT_FRAME) {
        /* the tile's rows to their places in the caller's whole-image buffer: one copy per strip (a strip of full-width rows
         * is contiguous in both), queued on the stream, one wait */
        hipStream_t st = c->lane[0].stream;
        const size_t px = 3 * sizeof(double);
        const uint32_t srows = p->strip_rows ? p->strip_rows : p->tile_h;
        for (uint32_t r = 0; r < p->tile_h; r += srows) {
            const uint32_t rows = p->tile_h - r < srows ? p->tile_h - r : srows;
            const uint64_t j = p->strip_rows ? (uint64_t)p->y0 + (uint64_t)(r / srows) * p->strip_period : (uint64_t)p->y0 + r;
            double* dst = out_rgb + (j * p->width + p->x0) * 3u;
            const double* src = c->d_out + (size_t)r * p->tile_w * 3u;
            hipError_t e = (p->tile_w == p->width)
                ? hipMemcpyAsync(dst, src, (size_t)rows * p->tile_w * px, hipMemcpyDeviceToHost, st)
                : hipMemcpy2DAsync(dst, (size_t)p->width * px, src, (size_t)p->tile_w * px, (size_t)p->tile_w * px, rows, hipMemcpyDeviceToHost, st);
            if (!hip_ok(e, "framebuffer copy")) return RT1W_ERR_DEVICE;
        }
        if (!hip_ok(hipStreamSynchronize(st), "framebuffer copy")) return RT1W_ERR_DEVICE;
    } else if (!hip_ok(hipMemcpy(out_rgb, c->d_out, bytes, hipMemcpyDeviceToHost), "framebuffer copy")) return RT1W_ERR_DEVICE;
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT1W_OK;
}

uint32_t rt1w_abi_sizeof(int what) {
    switch (what) {
        case 0: return (uint32_t)sizeof(rt1w_render_params);
        case 1: return (uint32_t)sizeof(rt1w_stats);
        case 2: return (uint32_t)sizeof(rt1w_scene_info);
        case 3: return (uint32_t)sizeof(rt1w_specialise_info);
        default: return 0u;
    }
}

int rt1w_host_alloc(uint64_t bytes, void** out) {
    if (!out || bytes == 0) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    *out = nullptr;
    if (!hip_ok(hipHostMalloc(out, (size_t)bytes, hipHostMallocPortable), "hipHostMalloc")) return RT1W_ERR_NOMEM;
    return RT1W_OK;
}
int rt1w_host_free(void* p) { if (p && !hip_ok(hipHostFree(p), "hipHostFree")) return RT1W_ERR_DEVICE; return RT1W_OK; }
int rt1w_host_register(void* p, uint64_t bytes) {
    if (!p || bytes == 0) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (!hip_ok(hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable), "hipHostRegister")) return RT1W_ERR_DEVICE;
    return RT1W_OK;
}
int rt1w_host_unregister(void* p) { if (p && !hip_ok(hipHostUnregister(p), "hipHostUnregister")) return RT1W_ERR_DEVICE; return RT1W_OK; }

/* for walk_lab.hip (diagnostics): the scene view the kernels get, and the device */
void rt1w_internal_register_wavefront(rt1w_wf_render_fn render, rt1w_wf_destroy_fn destroy) { g_wf_render = render; g_wf_destroy = destroy; }
const void* rt1w_internal_view(const rt1w_context* c) { return &c->view; }
int rt1w_internal_device(const rt1w_context* c) { return c->device; }

int rt1w_debug_aabb(rt1w_context* c, const double* in, int* out_literal, int* out_fast, uint64_t n) {
    if (!c || !in || !out_literal || !out_fast) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (n == 0) return RT1W_OK;
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return RT1W_ERR_DEVICE;
    double* din = nullptr; int *d0 = nullptr, *d1 = nullptr;
    int rc = RT1W_OK;
    if (!hip_ok(hipMalloc((void**)&din, n * 14 * sizeof(double)), "hipMalloc") || !hip_ok(hipMalloc((void**)&d0, n * sizeof(int)), "hipMalloc") ||
        !hip_ok(hipMalloc((void**)&d1, n * sizeof(int)), "hipMalloc")) rc = RT1W_ERR_NOMEM;
    if (rc == RT1W_OK) {
        (void)hipMemcpy(din, in, n * 14 * sizeof(double), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(rt_debug_aabb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->lane[0].stream, din, d0, d1, (unsigned long long)n);
        if (!hip_ok(hipStreamSynchronize(c->lane[0].stream), "debug kernel")) rc = RT1W_ERR_DEVICE;
        else { (void)hipMemcpy(out_literal, d0, n * sizeof(int), hipMemcpyDeviceToHost); (void)hipMemcpy(out_fast, d1, n * sizeof(int), hipMemcpyDeviceToHost); }
    }
    if (din) (void)hipFree(din);
    if (d0) (void)hipFree(d0);
    if (d1) (void)hipFree(d1);
    return rc;
}

int rt1w_debug_texture(rt1w_context* c, int mode, uint32_t tex, const double* in, double* out, uint64_t n) {
    if (!c || !in || !out) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (n == 0) return RT1W_OK;
    RtSceneView view;
    memcpy(&view, &c->view, sizeof view);
    if (mode < 0 || mode > 2) { rt1w::set_error("unknown mode"); return RT1W_ERR_INVALID; }
    if (mode == 0 && tex >= view.n_textures) { rt1w::set_error("bad texture id"); return RT1W_ERR_INVALID; }
    if (mode == 1 && view.perlin == nullptr) { rt1w::set_error("the scene has no Perlin table"); return RT1W_ERR_INVALID; }
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return RT1W_ERR_DEVICE;
    double *din = nullptr, *dout = nullptr;
    int rc = RT1W_OK;
    if (!hip_ok(hipMalloc((void**)&din, n * 5 * sizeof(double)), "hipMalloc") || !hip_ok(hipMalloc((void**)&dout, n * 3 * sizeof(double)), "hipMalloc")) rc = RT1W_ERR_NOMEM;
    if (rc == RT1W_OK) {
        (void)hipMemcpy(din, in, n * 5 * sizeof(double), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(rt_debug_texture_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, view, mode, tex, din, dout, (unsigned long long)n);
        if (!hip_ok(hipGetLastError(), "launch") || !hip_ok(hipDeviceSynchronize(), "texture probe kernel")) rc = RT1W_ERR_DEVICE;
        else (void)hipMemcpy(out, dout, n * 3 * sizeof(double), hipMemcpyDeviceToHost);
    }
    if (din) (void)hipFree(din);
    if (dout) (void)hipFree(dout);
    return rc;
}


int rt1w_debug_stamps(rt1w_context* c, uint64_t out[16], int reset) {
    if (!c || !out) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    for (int i = 0; i < 16; ++i) out[i] = 0;
    int valid = 0;
    (void)hipSetDevice(c->device);
#ifdef RT_STAMPS
    {
        unsigned long long h[16];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp_total), sizeof h);
        for (int i = 0; i < 16; ++i) out[i] += h[i];
        if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_total), z, sizeof z); }
        valid = 1;
    }
#endif
    /* a scene-specialised kernel compiled with RT1W_JIT_STAMPS=1 in the environment carries its own counters */
    if (c->jit_mod) {
        hipDeviceptr_t dptr = nullptr;
        size_t bytes = 0;
        if (hipModuleGetGlobal(&dptr, &bytes, c->jit_mod, "g_stamp_total") == hipSuccess && bytes == 16 * sizeof(unsigned long long)) {
            unsigned long long h[16];
            (void)hipMemcpy(h, dptr, sizeof h, hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i) out[i] += h[i];
            if (reset) (void)hipMemset(dptr, 0, sizeof h);
            valid = 1;
        }
    }
    (void)hipGetLastError(); /* a kernel without the counters is not an error: do not leave "named symbol not found" for the next launch check */
    return valid;
}

int rt1w_debug_eval(rt1w_context* c, int fn, const double* a, const double* b, double* out, uint64_t n) {
    if (!c || !a || !b || !out) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (n == 0) return RT1W_OK;
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return RT1W_ERR_DEVICE;
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    size_t bytes = (size_t)n * sizeof(double);
    int rc = RT1W_OK;
    if (!hip_ok(hipMalloc((void**)&da, bytes), "hipMalloc") || !hip_ok(hipMalloc((void**)&db, bytes), "hipMalloc") ||
        !hip_ok(hipMalloc((void**)&dout, bytes), "hipMalloc")) rc = RT1W_ERR_NOMEM;
    if (rc == RT1W_OK) {
        (void)hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
        (void)hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(rt_debug_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->lane[0].stream, fn, da, db, dout, (unsigned long long)n);
        if (!hip_ok(hipStreamSynchronize(c->lane[0].stream), "debug kernel")) rc = RT1W_ERR_DEVICE;
        else (void)hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    }
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (dout) (void)hipFree(dout);
    return rc;
}

} /* extern "C" */
