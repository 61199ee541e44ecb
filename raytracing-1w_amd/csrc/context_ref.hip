/* context_ref.hip -- the render kernels once more, with the REFERENCE's OWN random stream (RT1W_RNG_REFERENCE).
 *
 * The north star mandates a counter-based Philox stream per (pixel, sample): that is what every other kernel of this
 * library draws from, and it makes the framebuffer different from the Rust program's sample for sample (same
 * distribution, other numbers).  This translation unit compiles the SAME core (rt_core.h, rt_kernel_plain.h) with
 * RT_RNG_REFSTREAM defined, inside its own namespace so that nothing of it can meet the default build at link time:
 * include/rt1w_num.h then generates the words of `StdRng::seed_from_u64(j * image_width + i)` (src/main.rs:964) -- ChaCha12
 * keyed by the PCG32 expansion of the pixel seed, as rand 0.8.4 / rand_chacha 0.3.1 / rand_core 0.6.3 define it -- with
 * BlockRng's unaligned next_u64, and a pixel's samples draw on from one stream in order (chunk = spp).  Everything else
 * -- traversal, shading, draw order, draw shapes, summation order -- is the text the default kernels are built from.
 * With it the GPU reproduces the reference's own render, rest_of_your_life.png (600x600, 100 spp), pixel for pixel
 * (tests/test_gpu_parity.py::test_gpu_reference_stream_reproduces_the_reference_png): the elementary functions here are
 * the numerical contract's, not libm's, but an ulp only matters when it flips a branch (~1e-13 per decision).
 * A parity mode: a path slot owns a pixel for all its samples (chunk = spp) and the ChaCha block is recomputed per four words.
 * Round 4: small scenes run it through the workgroup-level path reordering too (rt_render_sorted_body: the generator state is part of
 * the exchanged path state, so a pixel's stream moves with its path from lane to lane) -- the same frame, bit for bit, as the plain
 * kernel (RT1W_UNSORTED), at about three times its rate where there are enough pixels to fill the GPU.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <string.h>
#include <type_traits>

#define RT_RNG_REFSTREAM 1
#define RT_XCH_PARTS 2 /* the reordering kernels' exchange in two rounds: 27 KB of LDS per workgroup, room for a fourth workgroup per CU */

namespace rtref {
#include "rt1w_num.h"
#include "rt_flat.h"
#include "rt_core.h"
#include "rt_kernel_plain.h"

typedef RtCfg<true, true, true, true> CfgSweep;   /* scenes of <= RT_SWEEP_MAX_NODES nodes, every feature */
typedef RtCfg<true, true, true, false> CfgStack;  /* every scene */

template <class Cfg>
__global__ __launch_bounds__(RT_BLOCK, 2) void rt_render_kernel_ref(RtSceneView sc, RtFrame f, double* __restrict__ partial,
                                                                   unsigned long long* __restrict__ counters) {
    rt_render_plain_body<Cfg, false>(sc, f, partial, counters);
}
/* the reordering kernel (rt_kernel_sorted.h) on the reference's stream: RtCfgV0 for scenes of solid colours without media or moving
 * spheres (Cornell: what the reference's own PNG shows), the every-feature sweep otherwise */
/* four waves per SIMD for the solid-colour build (128 VGPRs, 68 B of scratch: Cornell 600x600 603 -> 620 Mpaths/s, 4K 1428 -> 1560); the
 * every-feature build would spill 300 B per lane there and keeps three */
template <class Cfg>
__global__ __launch_bounds__(RT_SORT_BLOCK, (Cfg::tex || Cfg::media || Cfg::msphere) ? RT_SORT_WAVES(Cfg) : 4) void rt_render_kernel_ref_sorted(RtSceneView sc, RtFrame f, double* __restrict__ partial,
                                                                                           unsigned long long* __restrict__ counters) {
    rt_render_sorted_body<Cfg>(sc, f, partial, counters);
}
} // namespace rtref

/* called by context.hip; `view` / `frame` are the bytes of its RtSceneView / RtFrame (same layout: same headers) */
/* mode: 0 sweep (plain kernel), 1 stack walk (plain kernel), 2 reordering kernel V0, 3 reordering kernel with every feature */
extern "C" int rt1w_internal_ref_blocks_per_cu(int stack_walk) {
    int per_cu = 0;
    if (stack_walk >= 2) {
        hipError_t e2 = stack_walk == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rtref::rt_render_kernel_ref_sorted<rtref::RtCfgV0>, RT_SORT_BLOCK, 0)
                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rtref::rt_render_kernel_ref_sorted<rtref::CfgSweep>, RT_SORT_BLOCK, 0);
        return e2 == hipSuccess && per_cu > 0 ? per_cu : 1;
    }
    hipError_t e = stack_walk ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rtref::rt_render_kernel_ref<rtref::CfgStack>, RT_BLOCK, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rtref::rt_render_kernel_ref<rtref::CfgSweep>, RT_BLOCK, 0);
    return e == hipSuccess && per_cu > 0 ? per_cu : 1;
}
extern "C" int rt1w_internal_ref_launch(int stack_walk, const void* view, const void* frame, double* partial, unsigned long long* counters,
                                        int grid, hipStream_t stream) {
    rtref::RtSceneView v;
    rtref::RtFrame f;
    memcpy(&v, view, sizeof v);
    memcpy(&f, frame, sizeof f);
    if (stack_walk == 2) hipLaunchKernelGGL(rtref::rt_render_kernel_ref_sorted<rtref::RtCfgV0>, dim3(grid), dim3(RT_SORT_BLOCK), 0, stream, v, f, partial, counters);
    else if (stack_walk == 3) hipLaunchKernelGGL(rtref::rt_render_kernel_ref_sorted<rtref::CfgSweep>, dim3(grid), dim3(RT_SORT_BLOCK), 0, stream, v, f, partial, counters);
    else if (stack_walk) hipLaunchKernelGGL(rtref::rt_render_kernel_ref<rtref::CfgStack>, dim3(grid), dim3(RT_BLOCK), 0, stream, v, f, partial, counters);
    else hipLaunchKernelGGL(rtref::rt_render_kernel_ref<rtref::CfgSweep>, dim3(grid), dim3(RT_BLOCK), 0, stream, v, f, partial, counters);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" unsigned rt1w_internal_ref_sizeof(int what) { return what == 0 ? (unsigned)sizeof(rtref::RtSceneView) : (unsigned)sizeof(rtref::RtFrame); }
