/* context_f32.hip -- the render kernels in single precision (RT1W_PRECISION_F32).
 *
 * The reference chooses its arithmetic with one alias, `type Float = f64;` (src/main.rs:1); setting it to f32 turns every
 * vector, ray, hit record, bounding box, camera and colour of the program into f32.  This translation unit does the same to
 * the device core: it compiles rt_core.h / rt_kernel_*.h a second time, inside its own namespace, with `double` redefined
 * to `float` (RT_F32).  What stays 64-bit is spelled rt_f64 in the headers: the generator's word -> number conversions,
 * the internals of sin/cos/atan2/acos/log (evaluated in 64 bits, rounded once), the bit tricks, and the per-pixel sums
 * ("f32 traversal and shading, f64 accumulation").  The scene arrays are converted to the f32 record layouts once per
 * context; BVH boxes are rounded OUTWARD and widened by 1e-5 of their magnitude, because the reference's 0.0001 pad of a
 * rect's box (src/aarect.rs:74-79) is 1.6 f32 ulps at k = 555 and nothing at the final scene's coordinates.
 * Parity of this mode is statistical by nature (tests/test_gpu_parity.py::test_f32_mode_*): block means within Monte-Carlo
 * noise of the f64 frame, no NaN pixels beyond the f64 frame's, no light leaks at the k = 555 walls.
 */
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <string.h>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "rt_kernel_plain.h" /* the f64 record layouts, for the converters (global namespace) */

#undef RT1W_NUM_H
#undef RT1W_FLAT_H
#undef RT1W_CORE_H
#undef RT_KERNEL_SORTED_H
#undef RT_KERNEL_PLAIN_H
#undef RT1W_WALK_PAIR_H
#define RT_F32 1
#define double float

namespace rtf32 {
#include "rt1w_num.h"
#include "rt_flat.h"
#include "rt_core.h"
#include "rt_kernel_sorted.h"
#include "rt_kernel_plain.h"

/* waves per SIMD the f32 kernels are built for.  The Cornell variant V0 fits 4 (127 VGPRs, no spill).  The feature-rich variants do
 * not: held to 128 registers they spill 140-200 of them (and the reordering kernels missed the bound anyway: 3 and 2 waves), so
 * they are built for 3 like their f64 forms (static figures: hipcc -Rpass-analysis=kernel-resource-usage, tools/kernel_resources.py). */
#define RT_F32_WAVES(Cfg) ((Cfg::sweep && !Cfg::media && !Cfg::tex && !Cfg::msphere) ? 4 : 3)
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK, RT_F32_WAVES(Cfg)) void rt_render_kernel_f32(RtSceneView sc, RtFrame f, rt_f64* __restrict__ partial, unsigned long long* __restrict__ counters) {
    rt_render_plain_body<Cfg, false>(sc, f, partial, counters);
}
template <class Cfg>
__global__ __launch_bounds__(RT_SORT_BLOCK, RT_SORT_WAVES(Cfg)) void rt_render_kernel_sorted_f32(RtSceneView sc, RtFrame f, rt_f64* __restrict__ partial, unsigned long long* __restrict__ counters) {
    rt_render_sorted_body<Cfg>(sc, f, partial, counters);
}
/* the stack-walk variants with the finished paths reordered at the end of every slice (rt_kernel_plain.h: rt_render_ss_body) */
template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK, RT_F32_WAVES(Cfg)) void rt_render_kernel_ss_f32(RtSceneView sc, RtFrame f, rt_f64* __restrict__ partial, unsigned long long* __restrict__ counters) {
    rt_render_ss_body<Cfg, RT_STACK_CAP, 3>(sc, f, partial, counters);
}
/* sphere scenes: the pair walk (rt_walk_pair.h: inner boxes and group boxes are both this build's f32 boxes, widened like every BVH box of
 * this mode) in slices + the reordering of the finished paths */
__global__ __launch_bounds__(RT_BLOCK, 3) void rt_render_kernel_pw_ss_f32(RtSceneView sc, RtPwView pw, RtFrame f, rt_f64* __restrict__ partial, unsigned long long* __restrict__ counters) {
    rt_render_ss_body<RtCfgV5, RT_PW_SS_STACK, RT_PW_SS_PARTS, true>(sc, f, partial, counters, &pw);
}
typedef void (*kernel_t)(RtSceneView, RtFrame, rt_f64*, unsigned long long*);
static kernel_t const g_plain[RT_N_VARIANTS] = {rt_render_kernel_f32<RtCfgV0>, rt_render_kernel_f32<RtCfgV1>, rt_render_kernel_f32<RtCfgV2>, rt_render_kernel_f32<RtCfgV3>,
                                                nullptr, rt_render_kernel_f32<RtCfgV5>};
static kernel_t const g_sorted[RT_N_VARIANTS] = {rt_render_kernel_sorted_f32<RtCfgV0>, rt_render_kernel_sorted_f32<RtCfgV1>, rt_render_kernel_ss_f32<RtCfgV2>,
                                                 rt_render_kernel_ss_f32<RtCfgV3>, nullptr, rt_render_kernel_ss_f32<RtCfgV5>};
} // namespace rtf32

#undef double

namespace {

float down(double x) { float f = (float)x; return ((double)f > x) ? nextafterf(f, -INFINITY) : f; }
float up(double x) { float f = (float)x; return ((double)f < x) ? nextafterf(f, INFINITY) : f; }

rtf32::RtNode conv_node(const ::RtNode& n) {
    rtf32::RtNode o;
    memset(&o, 0, sizeof o);
    o.kind = n.kind; o.skip = n.skip; o.b = n.b; o.mat = n.mat; o.a = n.a; o.pad = n.pad;
    const uint32_t k = n.kind & RT_KIND_MASK;
    if (k == RT_BVH2 || k == RT_BVH1) {
        for (int i = 0; i < 3; ++i) {
            const double mag = fmax(1.0, fmax(fabs(n.d[i]), fabs(n.d[i + 3])));
            o.d[i] = down(n.d[i] - 1e-5 * mag);
            o.d[i + 3] = up(n.d[i + 3] + 1e-5 * mag);
        }
    } else {
        for (int i = 0; i < 6; ++i) o.d[i] = (float)n.d[i];
    }
    for (int i = 0; i < 3; ++i) o.e[i] = (float)n.e[i];
    return o;
}

struct F32Scene {
    void* nodes = nullptr; void* lights = nullptr; void* materials = nullptr; void* textures = nullptr; void* perlin = nullptr;
    void* pw_inner = nullptr; void* pw_groups = nullptr;
    rtf32::RtSceneView view;
    rtf32::RtPwView pw;
    bool pw_ok = false;
    uint32_t pw_stack = 0; /* pushed right children a walk can hold at once: the depth of the inner-record tree */
};

template <class T>
bool upload_vec(void** dst, const std::vector<T>& v) {
    *dst = nullptr;
    const size_t bytes = v.size() * sizeof(T);
    if (hipMalloc(dst, bytes ? bytes : 16) != hipSuccess) return false;
    return !bytes || hipMemcpy(*dst, v.data(), bytes, hipMemcpyHostToDevice) == hipSuccess;
}

rtf32::RtV3 v3f(const ::RtV3& v) { rtf32::RtV3 o; o.x = (float)v.x; o.y = (float)v.y; o.z = (float)v.z; return o; }

} // namespace

extern "C" void rt1w_internal_f32_destroy(void* h) {
    F32Scene* s = static_cast<F32Scene*>(h);
    if (!s) return;
    void* bufs[] = {s->nodes, s->lights, s->materials, s->textures, s->perlin, s->pw_inner, s->pw_groups};
    for (void* b : bufs) if (b) (void)hipFree(b);
    delete s;
}

/* builds the f32 copies of the scene arrays on the current device; `images` is the context's own device copy (bytes) */
extern "C" int rt1w_internal_f32_create(const void* nodes_, uint32_t n_nodes, const void* lights_, uint32_t n_lights, const void* materials_,
                                        uint32_t n_materials, const void* textures_, uint32_t n_textures, const void* perlin_, uint32_t n_perlin,
                                        const void* view64_, void** out) {
    const ::RtNode* nodes = static_cast<const ::RtNode*>(nodes_);
    const ::RtNode* lights = static_cast<const ::RtNode*>(lights_);
    const ::RtMaterial* materials = static_cast<const ::RtMaterial*>(materials_);
    const ::RtTexture* textures = static_cast<const ::RtTexture*>(textures_);
    const ::RtPerlin* perlin = static_cast<const ::RtPerlin*>(perlin_);
    const ::RtSceneView& v64 = *static_cast<const ::RtSceneView*>(view64_);
    F32Scene* s = new (std::nothrow) F32Scene();
    if (!s) return -1;
    std::vector<rtf32::RtNode> fn(n_nodes + 1u), fl(n_lights); /* + one spare record: the fused walk reads record e + 1 with record e */
    memset(&fn[n_nodes], 0, sizeof fn[n_nodes]);
    for (uint32_t i = 0; i < n_nodes; ++i) fn[i] = conv_node(nodes[i]);
    for (uint32_t i = 0; i < n_lights; ++i) fl[i] = conv_node(lights[i]);
    std::vector<rtf32::RtMaterial> fm(n_materials);
    for (uint32_t i = 0; i < n_materials; ++i) {
        memset(&fm[i], 0, sizeof fm[i]);
        for (int k = 0; k < 4; ++k) fm[i].d[k] = (float)materials[i].d[k];
        fm[i].kind = materials[i].kind; fm[i].tex = materials[i].tex;
    }
    std::vector<rtf32::RtTexture> ft(n_textures);
    for (uint32_t i = 0; i < n_textures; ++i) {
        memset(&ft[i], 0, sizeof ft[i]);
        for (int k = 0; k < 3; ++k) ft[i].d[k] = (float)textures[i].d[k];
        ft[i].kind = textures[i].kind; ft[i].a = textures[i].a; ft[i].b = textures[i].b; ft[i].c = textures[i].c;
    }
    std::vector<rtf32::RtPerlin> fp(n_perlin);
    for (uint32_t i = 0; i < n_perlin; ++i) {
        for (int k = 0; k < 256 * 3; ++k) fp[i].ranvec[k] = (float)perlin[i].ranvec[k];
        memcpy(fp[i].perm_x, perlin[i].perm_x, sizeof fp[i].perm_x);
        memcpy(fp[i].perm_y, perlin[i].perm_y, sizeof fp[i].perm_y);
        memcpy(fp[i].perm_z, perlin[i].perm_z, sizeof fp[i].perm_z);
    }
    if (!upload_vec(&s->nodes, fn) || !upload_vec(&s->lights, fl) || !upload_vec(&s->materials, fm) || !upload_vec(&s->textures, ft) ||
        !upload_vec(&s->perlin, fp)) { rt1w_internal_f32_destroy(s); return -1; }
    rtf32::RtSceneView& v = s->view;
    memset(&v, 0, sizeof v);
    v.nodes = (const rtf32::RtNode*)s->nodes; v.lights = (const rtf32::RtNode*)s->lights;
    v.materials = (const rtf32::RtMaterial*)s->materials; v.textures = (const rtf32::RtTexture*)s->textures;
    v.perlin = (const rtf32::RtPerlin*)s->perlin; v.images = v64.images;
    v.root = v64.root; v.n_nodes = v64.n_nodes; v.n_lights = v64.n_lights; v.n_materials = v64.n_materials; v.n_textures = v64.n_textures;
    const ::RtCamera& c = v64.camera;
    v.camera.origin = v3f(c.origin); v.camera.lower_left_corner = v3f(c.lower_left_corner); v.camera.horizontal = v3f(c.horizontal);
    v.camera.vertical = v3f(c.vertical); v.camera.u = v3f(c.u); v.camera.v = v3f(c.v); v.camera.w = v3f(c.w);
    v.camera.lens_radius = (float)c.lens_radius; v.camera.time0 = (float)c.time0; v.camera.time1 = (float)c.time1;
    v.background = v3f(v64.background);
    /* pair-walk records of a sphere scene, from THIS build's node array (boxes already widened by conv_node) */
    {
        std::vector<rtf32::RtNode> only(fn.begin(), fn.begin() + n_nodes);
        std::vector<rtf32::RtPwInner> pin;
        std::vector<rtf32::RtPwGroup> pgr;
        std::string why;
        memset(&s->pw, 0, sizeof s->pw);
        if (n_nodes > 0 && rtf32::rt_pw_build(only, v64.root, pin, pgr, s->pw, why) && upload_vec(&s->pw_inner, pin) && upload_vec(&s->pw_groups, pgr)) {
            s->pw.inner = (const rtf32::RtPwInner*)s->pw_inner; s->pw.groups = (const rtf32::RtPwGroup*)s->pw_groups;
            /* deepest chain of inner records: a walk pushes at most one right child per level */
            struct D { static uint32_t of(const std::vector<rtf32::RtPwInner>& v, uint32_t i) {
                if (i & RT_PW_LEAF) return 0u;
                const uint32_t a = of(v, v[i].l), b = of(v, v[i].r);
                return 1u + (a > b ? a : b);
            } };
            s->pw_stack = (s->pw.root & RT_PW_LEAF) ? 0u : D::of(pin, s->pw.root);
            s->pw_ok = true;
        }
    }
    *out = s;
    return 0;
}

extern "C" int rt1w_internal_f32_pw(void* h, unsigned stack_cap) {
    F32Scene* s = static_cast<F32Scene*>(h);
    return (s && s->pw_ok && s->pw_stack <= stack_cap) ? 1 : 0;
}

extern "C" unsigned rt1w_internal_f32_view(void* h, void* out, unsigned cap) {
    F32Scene* s = static_cast<F32Scene*>(h);
    if (!s || cap < sizeof s->view) return 0u;
    memcpy(out, &s->view, sizeof s->view);
    return (unsigned)sizeof s->view;
}

extern "C" int rt1w_internal_f32_blocks_per_cu(int variant, int sorted) {
    int per_cu = 0;
    if (variant < 0 || variant >= RT_N_VARIANTS) return 0;
    if (sorted == 2) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rtf32::rt_render_kernel_pw_ss_f32, RT_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        return per_cu;
    }
    rtf32::kernel_t k = sorted ? rtf32::g_sorted[variant] : rtf32::g_plain[variant];
    if (!k) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, sorted ? RT_SORT_BLOCK : RT_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    return per_cu;
}

/* `frame` = the bytes of an RtFrame (no floating-point fields: same layout in both builds) */
extern "C" int rt1w_internal_f32_launch(void* h, int variant, int sorted, const void* frame, double* partial, unsigned long long* counters, int grid,
                                        hipStream_t stream) {
    F32Scene* s = static_cast<F32Scene*>(h);
    if (variant < 0 || variant >= RT_N_VARIANTS) return -1;
    rtf32::kernel_t k = sorted ? rtf32::g_sorted[variant] : rtf32::g_plain[variant];
    if (!s || !k) return -1;
    rtf32::RtFrame f;
    static_assert(sizeof(rtf32::RtFrame) == sizeof(::RtFrame), "RtFrame has no floating-point fields");
    memcpy(&f, frame, sizeof f);
    if (sorted == 2) { /* the pair-walk kernel of sphere scenes (variant 5 only; the caller asked rt1w_internal_f32_pw first) */
        if (variant != 5 || !s->pw_ok) return -1;
        hipLaunchKernelGGL(rtf32::rt_render_kernel_pw_ss_f32, dim3(grid), dim3(RT_BLOCK), 0, stream, s->view, s->pw, f, partial, counters);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(sorted ? RT_SORT_BLOCK : RT_BLOCK), 0, stream, s->view, f, partial, counters);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
