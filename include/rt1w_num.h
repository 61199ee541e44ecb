/* rt1w_num.h -- the numerical contract shared by the HIP device code, the host
 * scene code and the CPU oracle.
 *
 * Why this header exists: the north-star parity claim is "same f64 framebuffer,
 * bit-identical PPM".  f64 + - * / sqrt are IEEE correctly rounded on x86-64 and
 * on gfx950, so the only sources of CPU/GPU divergence are (a) fused
 * multiply-add contraction, (b) vendor libm differences in the last ulp and
 * (c) the random stream.  (a) is removed by building every translation unit
 * with -ffp-contract=off; (b) and (c) are removed by defining the elementary
 * functions and the generator HERE, in plain + - * / sqrt arithmetic with a
 * fixed evaluation order, and compiling the same text for both sides.
 *
 * What of the reference this restates (all paths under /root/reference):
 *   - per-pixel RNG seeding, src/main.rs:964 (`MyRng::seed_from_u64(j*W+i)`):
 *     the reference generator is StdRng = ChaCha12 from the un-vendored crates
 *     rand 0.8.4 / rand_chacha 0.3.1 / rand_core 0.6.3 (Cargo.lock).  The north
 *     star replaces it with a counter-based Philox4x32-10 keyed by the same
 *     pixel seed ("Philox-in-register RNG"); stream-level parity with ChaCha is
 *     therefore out of scope, distribution-level parity is in scope.
 *   - the *sampling shapes* of rand 0.8.4 on top of the word stream
 *     (Standard f64, UniformFloat::sample_single, Standard bool,
 *     UniformInt::sample_single_inclusive used by `choose`/`shuffle`/
 *     `gen_range(0..=2)`), call sites src/main.rs:968-969, src/camera.rs:71,
 *     src/math.rs:9-11,32,40-41,56-57, src/pdf.rs:63, src/hittable.rs:153,
 *     src/aarect.rs:142-144, src/material.rs:146, src/constant_medium.rs:85,
 *     src/bvh.rs:84, src/perlin.rs:21,30-32.  rand is not vendored, so these
 *     shapes are restated from its published algorithm (parity unpinned).
 *   - cgmath 0.18 vector semantics used throughout the reference
 *     (dot = (x*x'+y*y')+z*z', normalize = v*(1/|v|), element-wise v/s).
 *
 * Philox4x32-10 is pinned by the Random123 known-answer vectors
 * (tests/test_num_contract.py).  The elementary functions are pinned against
 * glibc/mpmath within a stated ulp bound (same test file).
 */
#ifndef RT1W_NUM_H
#define RT1W_NUM_H

#include <stdint.h>

#if defined(__HIPCC__)
#if !defined(__HIPCC_RTC__) /* the run-time compiler has the runtime header built in */
#include <hip/hip_runtime.h>
#endif
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

/* The f32 build (RT1W_PRECISION_F32; csrc/context_f32.hip) compiles this header and the core with `double` redefined to
 * `float` -- the reference's own switch, `type Float = f64` -> f32 (src/main.rs:1).  What must stay 64-bit there (bit
 * tricks, the generator's conversions, the elementary functions' internals, the pixel sums) is spelled rt_f64, which that
 * translation unit declares before it redefines `double`. */
#if !defined(RT_F32)
typedef double rt_f64;
#endif

/* ------------------------------------------------------------------ bits -- */

RT_HD uint64_t rt_d2u(rt_f64 x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
RT_HD rt_f64 rt_u2d(uint64_t u) { rt_f64 x; __builtin_memcpy(&x, &u, 8); return x; }

#define RT_INF (__builtin_huge_val())
/* a constant in the working precision of the build: `double` itself in the f64 builds (a no-op), float where the f32 build
 * redefines the keyword (csrc/context_f32.hip) -- there an untyped literal would drag the expression around it into f64 */
#define RT_R(x) ((double)(x))
#define RT_PI 3.14159265358979323846264338327950288

RT_HD bool rt_isnan(double x) { return x != x; }
RT_HD bool rt_isnan64(rt_f64 x) { return x != x; }
#if defined(RT_F32)
RT_HD double rt_abs(double x) { return __builtin_fabsf(x); }
#else
RT_HD double rt_abs(double x) { return rt_u2d(rt_d2u(x) & 0x7FFFFFFFFFFFFFFFull); }
#endif
RT_HD rt_f64 rt_abs64(rt_f64 x) { return rt_u2d(rt_d2u(x) & 0x7FFFFFFFFFFFFFFFull); }
/* Rust f64::min / f64::max: if one operand is NaN the other is returned. */
RT_HD double rt_min(double a, double b) { return (a < b || rt_isnan(b)) ? a : b; }
RT_HD double rt_max(double a, double b) { return (a > b || rt_isnan(b)) ? a : b; }
#if defined(RT_F32)
RT_HD double rt_sqrt(double x) { return __builtin_sqrtf(x); }
#else
RT_HD double rt_sqrt(double x) { return __builtin_sqrt(x); }
#endif
RT_HD rt_f64 rt_sqrt64(rt_f64 x) { return __builtin_sqrt(x); }
/* floor without libm: |x| >= 2^52 is already integral. */
RT_HD rt_f64 rt_floor64(rt_f64 x) {
    if (!(rt_abs64(x) < 4503599627370496.0)) return x;
    rt_f64 t = (rt_f64)(int64_t)x; /* truncates toward zero */
    return (t > x) ? t - 1.0 : t;
}
#if defined(RT_F32)
RT_HD double rt_floor(double x) { return (double)rt_floor64((rt_f64)x); } /* exact: the floor of a float is a float */
#else
RT_HD double rt_floor(double x) { return rt_floor64(x); }
#endif

/* --------------------------------------------------------- Philox4x32-10 -- */

struct RtPhiloxOut { uint32_t w0, w1, w2, w3; };

RT_HD RtPhiloxOut rt_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    RtPhiloxOut o; o.w0 = c0; o.w1 = c1; o.w2 = c2; o.w3 = c3; return o;
}

/* Word stream: block b of a stream is Philox(ctr=(b, c1, c2, c3), key=(k0,k1));
 * words are consumed in order w0..w3.  A 64-bit draw takes an even-aligned word
 * pair (low word first), so it never straddles a block. */
#define RT_DOMAIN_RENDER 0x52454E44u /* "REND" */
#define RT_DOMAIN_BUILD 0x424C4453u  /* "BLDS" */

/* Buffer: A = words of the current block still unconsumed (a0 is the next word, `left` of
 * them), B = the following block, generated ahead (`bv` set).  Keeping B lets the device
 * code generate blocks at a few phase starts where most lanes of a wave take part
 * (rt_rng_reserve / rt_rng_fill) instead of inside every draw; the word stream and its
 * consumption order are exactly those of a one-block buffer.  No indexed access anywhere,
 * so the whole state stays in registers. */
struct RtRng {
    uint32_t k0, k1, c1, c2, c3;
    uint32_t blk;  /* next block index to generate */
    uint32_t left; /* unconsumed words of A (0..4) */
    uint32_t a0, a1, a2, a3;
    uint32_t bv;   /* B holds a generated block */
    uint32_t b0, b1, b2, b3;
};

#if defined(RT_RNG_CHECK)
#include <cstdio>
#include <cstdlib>
#define RT_RNG_ASSERT(c) do { if (!(c)) { std::fprintf(stderr, "rt1w_num.h: RNG buffer underflow (missing reserve)\n"); std::abort(); } } while (0)
#else
#define RT_RNG_ASSERT(c) ((void)0)
#endif

#if defined(RT_RNG_REFSTREAM)
/* REFERENCE-STREAM BUILD (kernels of csrc/context_ref.hip and oracle/liborc_flat_ref.so only; selected per render with
 * RT1W_RNG_REFERENCE): the word stream is the reference's own `StdRng::seed_from_u64(j * image_width + i)` (src/main.rs:964,
 * `type MyRng = StdRng`, main.rs:2) = ChaCha12 keyed by the PCG32 expansion of the pixel seed (rand 0.8.4, rand_chacha 0.3.1,
 * rand_core 0.6.3 -- un-vendored crates, published algorithms restated; pinned by oracle/refstream.h's known answers and by
 * the reference's own PNG), ONE stream per pixel drawn on through all of its samples.  Group `blk` of four words is
 * words 4*(blk & 3).. of ChaCha block blk >> 2 (the 64-word buffer of rand_chacha is just four consecutive blocks); the
 * block is recomputed for each group -- four times the arithmetic, no indexed buffer, and the state stays the RtRng the
 * kernels already carry (k0,k1 = the pixel seed). */
RT_HD uint32_t rt_rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
RT_HD void rt_rng_gen_b(RtRng& r) {
    /* rand_core SeedableRng::seed_from_u64: eight PCG32 (XSH RR) outputs, state advanced first */
    uint64_t st = ((uint64_t)r.k1 << 32) | r.k0;
    uint32_t key[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        st = st * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27);
        const uint32_t rot = (uint32_t)(st >> 59);
        key[i] = (xs >> rot) | (xs << ((32u - rot) & 31u));
    }
    const uint32_t ctr = r.blk >> 2; /* 64-bit block counter in the reference; 2^34 words per pixel are out of reach here */
    uint32_t x0 = 0x61707865u, x1 = 0x3320646eu, x2 = 0x79622d32u, x3 = 0x6b206574u;
    uint32_t x4 = key[0], x5 = key[1], x6 = key[2], x7 = key[3], x8 = key[4], x9 = key[5], x10 = key[6], x11 = key[7];
    uint32_t x12 = ctr, x13 = 0u, x14 = 0u, x15 = 0u;
#define RT_QR(a, b, c, d) a += b; d ^= a; d = rt_rotl32(d, 16); c += d; b ^= c; b = rt_rotl32(b, 12); a += b; d ^= a; d = rt_rotl32(d, 8); c += d; b ^= c; b = rt_rotl32(b, 7);
#pragma unroll
    for (int i = 0; i < 6; ++i) { /* ChaCha12: six double rounds */
        RT_QR(x0, x4, x8, x12) RT_QR(x1, x5, x9, x13) RT_QR(x2, x6, x10, x14) RT_QR(x3, x7, x11, x15)
        RT_QR(x0, x5, x10, x15) RT_QR(x1, x6, x11, x12) RT_QR(x2, x7, x8, x13) RT_QR(x3, x4, x9, x14)
    }
#undef RT_QR
    x0 += 0x61707865u; x1 += 0x3320646eu; x2 += 0x79622d32u; x3 += 0x6b206574u;
    x4 += key[0]; x5 += key[1]; x6 += key[2]; x7 += key[3]; x8 += key[4]; x9 += key[5]; x10 += key[6]; x11 += key[7];
    x12 += ctr;
    const uint32_t g = r.blk & 3u;
    r.b0 = g == 0u ? x0 : (g == 1u ? x4 : (g == 2u ? x8 : x12));
    r.b1 = g == 0u ? x1 : (g == 1u ? x5 : (g == 2u ? x9 : x13));
    r.b2 = g == 0u ? x2 : (g == 1u ? x6 : (g == 2u ? x10 : x14));
    r.b3 = g == 0u ? x3 : (g == 1u ? x7 : (g == 2u ? x11 : x15));
    r.blk += 1u; r.bv = 1u;
}
#else
RT_HD void rt_rng_gen_b(RtRng& r) {
    RtPhiloxOut o = rt_philox4x32_10(r.blk, r.c1, r.c2, r.c3, r.k0, r.k1);
    r.b0 = o.w0; r.b1 = o.w1; r.b2 = o.w2; r.b3 = o.w3;
    r.blk += 1u; r.bv = 1u;
}
#endif
/* make sure the next n words can be taken without generating (n <= 4 + left) */
RT_HD void rt_rng_reserve(RtRng& r, uint32_t n) {
    if (r.left + (r.bv << 2) < n) rt_rng_gen_b(r);
}
/* words a 64-bit draw (resp. two of them) may consume, counting the alignment skip */
#if defined(RT_RNG_REFSTREAM)
/* rand_core BlockRng::next_u64: two consecutive words at the current index, no alignment */
RT_HD uint32_t rt_rng_need_u64(const RtRng&) { return 2u; }
RT_HD uint32_t rt_rng_need_2u64(const RtRng&) { return 4u; }
#else
RT_HD uint32_t rt_rng_need_u64(const RtRng& r) { return 2u + (r.left & 1u); }
RT_HD uint32_t rt_rng_need_2u64(const RtRng& r) { return 4u + (r.left & 1u); }
#endif
/* top up B unconditionally */
RT_HD void rt_rng_fill(RtRng& r) { if (!r.bv) rt_rng_gen_b(r); }

RT_HD RtRng rt_rng_make(uint32_t k0, uint32_t k1, uint32_t c1, uint32_t c2, uint32_t c3) {
    RtRng r;
    r.k0 = k0; r.k1 = k1; r.c1 = c1; r.c2 = c2; r.c3 = c3;
    r.blk = 0u; r.left = 0u; r.a0 = r.a1 = r.a2 = r.a3 = 0u;
    r.bv = 0u; r.b0 = r.b1 = r.b2 = r.b3 = 0u;
    return r;
}
/* Stream of sample `sample` of the pixel whose reference seed is `pixel_seed`
 * (= j*W+i, src/main.rs:964). */
RT_HD RtRng rt_rng_pixel_sample(uint64_t pixel_seed, uint32_t sample, uint32_t global_seed) {
    return rt_rng_make((uint32_t)pixel_seed, (uint32_t)(pixel_seed >> 32), sample, global_seed, RT_DOMAIN_RENDER);
}
/* Sequential stream used by the one-shot scene build (replaces the reference's
 * entropy-seeded `MyRng::from_entropy()`, src/main.rs:803). */
RT_HD RtRng rt_rng_build(uint64_t build_seed) {
    return rt_rng_make((uint32_t)build_seed, (uint32_t)(build_seed >> 32), 0u, 0u, RT_DOMAIN_BUILD);
}

/* Position of a stream, small enough to keep around (3 numbers), and the way back to it: the blocks are a pure function of
 * (key, counter), so the buffered words are regenerated.  Used where a stretch of code may have to be redone from the same
 * random state (the phased walk's restart, csrc/rt_walk2.h). */
struct RtRngMark { uint32_t blk, left, bv; };
RT_HD RtRngMark rt_rng_mark(const RtRng& r) { RtRngMark m; m.blk = r.blk; m.left = r.left; m.bv = r.bv; return m; }
RT_HD void rt_rng_rewind(RtRng& r, RtRngMark m) {
    /* A = the last `left` words of block (blk - 1 - bv), B = block blk - 1 when bv */
    if (m.left > 0u) {
        r.blk = m.blk - 1u - m.bv;
        rt_rng_gen_b(r);
        const uint32_t w0 = r.b0, w1 = r.b1, w2 = r.b2, w3 = r.b3;
        r.a0 = m.left == 4u ? w0 : (m.left == 3u ? w1 : (m.left == 2u ? w2 : w3));
        r.a1 = m.left == 4u ? w1 : (m.left == 3u ? w2 : w3);
        r.a2 = m.left == 4u ? w2 : w3;
        r.a3 = w3;
    }
    if (m.bv) { r.blk = m.blk - 1u; rt_rng_gen_b(r); }
    r.blk = m.blk; r.left = m.left; r.bv = m.bv;
}

/* ---- unchecked takes: the caller has reserved ---- */
RT_HD void rt_rng_pull(RtRng& r) { /* A is empty: B becomes A */
    RT_RNG_ASSERT(r.bv);
    r.a0 = r.b0; r.a1 = r.b1; r.a2 = r.b2; r.a3 = r.b3;
    r.left = 4u; r.bv = 0u;
}
RT_HD uint32_t rt_take_u32(RtRng& r) {
    if (r.left == 0u) rt_rng_pull(r);
    uint32_t w = r.a0;
    r.a0 = r.a1; r.a1 = r.a2; r.a2 = r.a3;
    r.left -= 1u;
    return w;
}
#if defined(RT_RNG_REFSTREAM)
RT_HD uint64_t rt_take_u64(RtRng& r) { /* low word first, wherever the index stands (may cross into the next group) */
    const uint64_t lo = rt_take_u32(r);
    const uint64_t hi = rt_take_u32(r);
    return (hi << 32) | lo;
}
#else
RT_HD uint64_t rt_take_u64(RtRng& r) {
    if (r.left & 1u) { r.a0 = r.a1; r.a1 = r.a2; r.a2 = r.a3; r.left -= 1u; } /* even-align */
    if (r.left == 0u) rt_rng_pull(r);
    uint64_t v = ((uint64_t)r.a1 << 32) | r.a0;
    r.a0 = r.a2; r.a1 = r.a3;
    r.left -= 2u;
    return v;
}
#endif
/* ---- checked draws (reserve + take): what the host code and the literal oracle use ---- */
RT_HD uint32_t rt_next_u32(RtRng& r) { rt_rng_reserve(r, 1u); return rt_take_u32(r); }
RT_HD uint64_t rt_next_u64(RtRng& r) { rt_rng_reserve(r, rt_rng_need_u64(r)); return rt_take_u64(r); }

/* rand 0.8 `rng.gen::<f64>()`: 53 random bits scaled into [0,1). */
RT_HD rt_f64 rt_gen_f64(RtRng& r) {
    return (rt_f64)(rt_next_u64(r) >> 11) * (1.0 / 9007199254740992.0);
}
/* rand 0.8 `rng.gen_range(low..high)` for f64 (UniformFloat::sample_single):
 * 52 mantissa bits -> [1,2) -> minus 1 -> *scale + low, retry if >= high. */
RT_HD rt_f64 rt_gen_range(RtRng& r, rt_f64 low, rt_f64 high) {
    rt_f64 scale = high - low;
    for (;;) {
        rt_f64 v12 = rt_u2d((rt_next_u64(r) >> 12) | 0x3FF0000000000000ull);
        rt_f64 res = (v12 - 1.0) * scale + low;
        if (res < high) return res;
    }
}
/* rand 0.8 `rng.gen::<bool>()`: sign bit of one u32. */
RT_HD bool rt_gen_bool(RtRng& r) { return (int32_t)rt_next_u32(r) < 0; }
/* rand 0.8 UniformInt<u32>::sample_single_inclusive(0, n-1) -- used by
 * `slice.choose`, `shuffle` (gen_index) and `gen_range(0..=2)`:
 * widening multiply with the conservative power-of-two zone. */
RT_HD uint32_t rt_gen_below(RtRng& r, uint32_t n) {
    uint32_t lz = (uint32_t)__builtin_clz(n);
    uint32_t zone = (n << lz) - 1u;
    for (;;) {
        uint64_t m = (uint64_t)rt_next_u32(r) * n;
        if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
    }
}

/* conversions of one 64-bit draw (rand 0.8 Standard f64 / UniformFloat) */
RT_HD rt_f64 rt_f64_from_bits(uint64_t q) { return (rt_f64)(q >> 11) * (1.0 / 9007199254740992.0); }
RT_HD rt_f64 rt_range_from_bits(uint64_t q, rt_f64 low, rt_f64 high) {
    rt_f64 scale = high - low;
    rt_f64 v12 = rt_u2d((q >> 12) | 0x3FF0000000000000ull);
    return (v12 - 1.0) * scale + low;
}
/* ---- the same shapes over reserved words (device core; see rt_rng_reserve) ---- */
RT_HD rt_f64 rt_take_f64(RtRng& r) { return (rt_f64)(rt_take_u64(r) >> 11) * (1.0 / 9007199254740992.0); }
RT_HD rt_f64 rt_take_range(RtRng& r, rt_f64 low, rt_f64 high) {
    rt_f64 scale = high - low;
    rt_f64 v12 = rt_u2d((rt_take_u64(r) >> 12) | 0x3FF0000000000000ull);
    rt_f64 res = (v12 - 1.0) * scale + low;
    while (!(res < high)) { /* rounding pushed the value onto `high`: draw again (checked) */
        v12 = rt_u2d((rt_next_u64(r) >> 12) | 0x3FF0000000000000ull);
        res = (v12 - 1.0) * scale + low;
    }
    return res;
}
RT_HD bool rt_take_bool(RtRng& r) { return (int32_t)rt_take_u32(r) < 0; }

/* -------------------------------------------------- elementary functions -- */
/* All of these are evaluated with + - * / only, in the order written, so the
 * host and the device produce the same bits.  Accuracy targets (checked in
 * tests/test_num_contract.py): sin/cos <= 1 ulp for |x| <= 1e5, atan2/acos/log
 * <= 2 ulp. */

/* kernel sin/cos on [-pi/4, pi/4] (fdlibm minimax polynomials, degree 13/14) */
RT_HD rt_f64 rt_ksin(rt_f64 x, rt_f64 y) {
    const rt_f64 S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    rt_f64 z = x * x;
    rt_f64 v = z * x;
    rt_f64 r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
RT_HD rt_f64 rt_kcos(rt_f64 x, rt_f64 y) {
    const rt_f64 C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    rt_f64 z = x * x;
    rt_f64 r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    rt_f64 hz = 0.5 * z;
    rt_f64 w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}
/* Cody-Waite reduction by pi/2 in three 33-bit pieces; exact products for
 * |n| < 2^20, i.e. |x| < ~1.6e6 (accuracy degrades gradually beyond; callers
 * return NaN for |x| >= 2^30, far outside any scene's coordinates, so the
 * rt_f64->integer conversion below is always in range on both targets).
 * Returns quadrant (0..3), remainder in (hi, lo). */
struct RtRem { rt_f64 hi, lo; int n; };
RT_HD RtRem rt_rem_pio2(rt_f64 x) {
    const rt_f64 INVPIO2 = 6.36619772367581382433e-01;
    const rt_f64 P1 = 1.57079632673412561417e+00;  /* 0x3FF921FB54400000 */
    const rt_f64 P1T = 6.07710050650619224932e-11;
    const rt_f64 P2 = 6.07710050630396597660e-11;  /* 0x3DD0B4611A600000 */
    const rt_f64 P2T = 2.02226624879595063154e-21;
    const rt_f64 P3 = 2.02226624871116645580e-21;  /* 0x3BA3198A2E000000 */
    const rt_f64 P3T = 8.47842766036889956997e-32;
    rt_f64 t = x * INVPIO2;
    rt_f64 fn = rt_floor64(t + 0.5);
    int n = (int)((int64_t)fn & 3);
    /* three-stage subtraction, keeping a tail */
    rt_f64 r = x - fn * P1;
    rt_f64 w;
    rt_f64 y0;
    /* second iteration (always; cost is small, accuracy uniform) */
    rt_f64 t2 = r;
    w = fn * P2;
    r = t2 - w;
    w = fn * P2T - ((t2 - r) - w);
    y0 = r - w;
    /* third iteration */
    rt_f64 t3 = r;
    rt_f64 w3 = fn * P3;
    r = t3 - w3;
    w = fn * P3T - ((t3 - r) - w3);
    y0 = r - w;
    RtRem o;
    o.hi = y0;
    o.lo = (r - y0) - w;
    o.n = n;
    return o;
}
#define RT_TRIG_MAX 1073741824.0 /* 2^30 */
RT_HD rt_f64 rt_sin(rt_f64 x) {
    if (rt_abs64(x) < 0.78539816339744830962) return rt_ksin(x, 0.0);
    if (!(rt_abs64(x) < RT_TRIG_MAX)) return rt_u2d(0x7FF8000000000000ull);
    RtRem q = rt_rem_pio2(x);
    if (q.n == 0) return rt_ksin(q.hi, q.lo);
    if (q.n == 1) return rt_kcos(q.hi, q.lo);
    if (q.n == 2) return -rt_ksin(q.hi, q.lo);
    return -rt_kcos(q.hi, q.lo);
}
RT_HD rt_f64 rt_cos(rt_f64 x) {
    if (rt_abs64(x) < 0.78539816339744830962) return rt_kcos(x, 0.0);
    if (!(rt_abs64(x) < RT_TRIG_MAX)) return rt_u2d(0x7FF8000000000000ull);
    RtRem q = rt_rem_pio2(x);
    if (q.n == 0) return rt_kcos(q.hi, q.lo);
    if (q.n == 1) return -rt_ksin(q.hi, q.lo);
    if (q.n == 2) return -rt_kcos(q.hi, q.lo);
    return rt_ksin(q.hi, q.lo);
}
/* sin and cos of one argument with one shared reduction (same values as above) */
RT_HD void rt_sincos(rt_f64 x, rt_f64& s, rt_f64& c) {
    rt_f64 hi = x, lo = 0.0;
    int n = 0;
    if (!(rt_abs64(x) < RT_TRIG_MAX)) { s = c = rt_u2d(0x7FF8000000000000ull); return; }
    if (!(rt_abs64(x) < 0.78539816339744830962)) { RtRem q = rt_rem_pio2(x); hi = q.hi; lo = q.lo; n = q.n; }
    rt_f64 ks = rt_ksin(hi, lo), kc = rt_kcos(hi, lo);
    s = (n == 0) ? ks : (n == 1) ? kc : (n == 2) ? -ks : -kc;
    c = (n == 0) ? kc : (n == 1) ? -ks : (n == 2) ? -kc : ks;
}
RT_HD rt_f64 rt_tan(rt_f64 x) { rt_f64 s, c; rt_sincos(x, s, c); return s / c; }

/* atan for x >= 0 (fdlibm breakpoints 7/16, 11/16, 19/16, 39/16) */
RT_HD rt_f64 rt_atan_pos(rt_f64 x) {
    const rt_f64 AT0 = 3.33333333333329318027e-01, AT1 = -1.99999999998764832476e-01,
                 AT2 = 1.42857142725034663711e-01, AT3 = -1.11111104054623557880e-01,
                 AT4 = 9.09088713343650656196e-02, AT5 = -7.69187620504482999495e-02,
                 AT6 = 6.66107313738753120669e-02, AT7 = -5.83357013379057348645e-02,
                 AT8 = 4.97687799461593236017e-02, AT9 = -3.65315727442169155270e-02,
                 AT10 = 1.62858201153657823623e-02;
    rt_f64 hi, lo;
    if (x < 0.4375) { hi = 0.0; lo = 0.0; }
    else if (x < 0.6875) { hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17;
                           x = (2.0 * x - 1.0) / (2.0 + x); }
    else if (x < 1.1875) { hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17;
                           x = (x - 1.0) / (x + 1.0); }
    else if (x < 2.4375) { hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17;
                           x = (x - 1.5) / (1.0 + 1.5 * x); }
    else { hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17;
           x = -1.0 / x; }
    rt_f64 z = x * x;
    rt_f64 w = z * z;
    rt_f64 s1 = z * (AT0 + w * (AT2 + w * (AT4 + w * (AT6 + w * (AT8 + w * AT10)))));
    rt_f64 s2 = w * (AT1 + w * (AT3 + w * (AT5 + w * (AT7 + w * AT9))));
    if (hi == 0.0) return x - x * (s1 + s2);
    return hi - ((x * (s1 + s2) - lo) - x);
}
RT_HD rt_f64 rt_atan2(rt_f64 y, rt_f64 x) {
    const rt_f64 PI_LO = 1.2246467991473531772e-16;
    if (rt_isnan64(x) || rt_isnan64(y)) return x + y;
    bool yneg = (rt_d2u(y) >> 63) != 0, xneg = (rt_d2u(x) >> 63) != 0;
    rt_f64 ay = rt_abs64(y), ax = rt_abs64(x);
    if (ay == 0.0) return xneg ? (yneg ? -RT_PI : RT_PI) : y;
    if (ax == 0.0) return yneg ? -1.57079632679489655800e+00 : 1.57079632679489655800e+00;
    rt_f64 z;
    if (ax == RT_INF && ay == RT_INF) z = 7.85398163397448278999e-01;
    else if (ax == RT_INF) z = 0.0;
    else if (ay == RT_INF) z = 1.57079632679489655800e+00;
    else z = rt_atan_pos(ay / ax);
    if (!xneg) return yneg ? -z : z;
    rt_f64 r = RT_PI - (z - PI_LO);
    return yneg ? -r : r;
}
/* acos(x) = 2*atan2(sqrt(1-x), sqrt(1+x)) -- benign cancellation only */
RT_HD rt_f64 rt_acos(rt_f64 x) {
    if (!(x >= -1.0 && x <= 1.0)) return rt_u2d(0x7FF8000000000000ull);
    return 2.0 * rt_atan2(rt_sqrt64(1.0 - x), rt_sqrt64(1.0 + x));
}
/* natural log (fdlibm): x = 2^k * (1+f), sqrt(1/2) < 1+f < sqrt(2) */
RT_HD rt_f64 rt_log(rt_f64 x) {
    const rt_f64 LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const rt_f64 LG1 = 6.666666666666735130e-01, LG2 = 3.999999999940941908e-01,
                 LG3 = 2.857142874366239149e-01, LG4 = 2.222219843214978396e-01,
                 LG5 = 1.818357216161805012e-01, LG6 = 1.531383769920937332e-01,
                 LG7 = 1.479819860511658591e-01;
    if (rt_isnan64(x)) return x;
    if (x < 0.0) return rt_u2d(0x7FF8000000000000ull);
    if (x == 0.0) return -RT_INF;
    if (x == RT_INF) return x;
    int k = 0;
    uint64_t u = rt_d2u(x);
    if ((u >> 52) == 0) { x = x * 18014398509481984.0; k -= 54; u = rt_d2u(x); } /* subnormal */
    k += (int)((u >> 52) & 0x7FF) - 1023;
    u = (u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    rt_f64 m = rt_u2d(u);
    if (m > 1.41421356237309504880) { m = m * 0.5; k += 1; }
    rt_f64 f = m - 1.0;
    rt_f64 s = f / (2.0 + f);
    rt_f64 z = s * s;
    rt_f64 w = z * z;
    rt_f64 t1 = w * (LG2 + w * (LG4 + w * LG6));
    rt_f64 t2 = z * (LG1 + w * (LG3 + w * (LG5 + w * LG7)));
    rt_f64 R = t2 + t1;
    rt_f64 hfsq = 0.5 * f * f;
    rt_f64 dk = (rt_f64)k;
    return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}
/* x.powf(5.0) of src/material.rs:124, as a fixed multiplication chain */
RT_HD double rt_pow5(double x) { double x2 = x * x; return (x2 * x2) * x; }

#if defined(RT_F32)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_F32_ELEMENTARY_F64)
/* device f32 build: the elementary functions of a float argument in single precision (the compiler's own: ~1-2 ulp of f32).
 * This mode's parity with the reference is statistical (tests/test_gpu_parity.py::test_f32_mode_*), not bitwise, and the
 * 64-bit evaluations were half of the feature-rich f32 kernels' instructions. */
RT_HD double rt_sin(double x) { return ::sinf(x); }
RT_HD double rt_cos(double x) { return ::cosf(x); }
RT_HD void rt_sincos(double x, double& s, double& c) { s = ::sinf(x); c = ::cosf(x); }
RT_HD double rt_atan2(double y, double x) { return ::atan2f(y, x); }
RT_HD double rt_acos(double x) { return ::acosf(x); }
RT_HD double rt_log(double x) { return ::logf(x); }
#else
/* f32 build on the host (and with RT_F32_ELEMENTARY_F64): the functions above run in 64 bits and the result is rounded once */
RT_HD void rt_sincos(double x, double& s, double& c) { rt_f64 s64, c64; rt_sincos((rt_f64)x, s64, c64); s = (double)s64; c = (double)c64; }
#endif
#endif

/* ------------------------------------------------------------------ vec3 -- */
/* cgmath 0.18 semantics (un-vendored; restated): element-wise ops,
 * dot = (x*x' + y*y') + z*z', normalize = v * (1/|v|). */

struct RtV3 { double x, y, z; };

RT_HD RtV3 rt_v3(double x, double y, double z) { RtV3 v; v.x = x; v.y = y; v.z = z; return v; }
RT_HD RtV3 operator+(RtV3 a, RtV3 b) { return rt_v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD RtV3 operator-(RtV3 a, RtV3 b) { return rt_v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD RtV3 operator-(RtV3 a) { return rt_v3(-a.x, -a.y, -a.z); }
RT_HD RtV3 operator*(RtV3 a, double s) { return rt_v3(a.x * s, a.y * s, a.z * s); }
RT_HD RtV3 operator*(double s, RtV3 a) { return rt_v3(s * a.x, s * a.y, s * a.z); }
RT_HD RtV3 operator/(RtV3 a, double s) { return rt_v3(a.x / s, a.y / s, a.z / s); }
RT_HD RtV3 rt_mul(RtV3 a, RtV3 b) { return rt_v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD double rt_dot(RtV3 a, RtV3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD double rt_mag2(RtV3 a) { return rt_dot(a, a); }
RT_HD double rt_mag(RtV3 a) { return rt_sqrt(rt_dot(a, a)); }
RT_HD RtV3 rt_normalize(RtV3 a) { return a * (1.0 / rt_mag(a)); }
RT_HD RtV3 rt_cross(RtV3 a, RtV3 b) {
    return rt_v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_HD double rt_get(RtV3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
/* sums of a pixel's samples: 64-bit in every build (src/main.rs:972-989 adds Float's; the f32 build keeps the sum wider) */
struct RtV3d { rt_f64 x, y, z; };
RT_HD RtV3d rt_v3d(rt_f64 x, rt_f64 y, rt_f64 z) { RtV3d v; v.x = x; v.y = y; v.z = z; return v; }
RT_HD RtV3d rt_v3d_add(RtV3d a, RtV3 b) { return rt_v3d(a.x + (rt_f64)b.x, a.y + (rt_f64)b.y, a.z + (rt_f64)b.z); }

#endif /* RT1W_NUM_H */
