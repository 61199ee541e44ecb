/* refstream.h -- CPU ORACLE, TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * "Reference-stream" mode of the literal oracle: the generator and the draw shapes of the
 * reference's own `type MyRng = StdRng` (src/main.rs:2), so that oracle/oracle.cpp compiled
 * with -DORC_REFSTREAM reproduces the reference's *actual* per-pixel random stream
 * (`MyRng::seed_from_u64((j * image_width + i) as u64)`, src/main.rs:964; one stream per
 * pixel, drawn sequentially over the samples, src/main.rs:967-989) and can be compared
 * pixel for pixel with the artefact of the reference's own run, rest_of_your_life.png.
 *
 * The algorithm lives in third-party crates that are NOT vendored under /root/reference
 * (Cargo.lock: rand 0.8.4, rand_chacha 0.3.1, rand_core 0.6.3).  Their published algorithms
 * are restated here:
 *   - rand 0.8.4  rngs::StdRng            = rand_chacha::ChaCha12Rng
 *   - rand_core 0.6.3 SeedableRng::seed_from_u64 (default impl): the u64 is expanded into the
 *     32-byte seed by eight PCG32 (XSH RR) outputs, state advanced first, little-endian
 *   - rand_chacha 0.3.1 ChaCha12Core: DJB layout -- constants, 8 key words, 64-bit block
 *     counter (words 12,13), 64-bit stream id (words 14,15, zero); 12 rounds; the core
 *     produces four consecutive blocks per refill (a 64-word buffer, blocks in counter order)
 *   - rand_core 0.6.3 block::BlockRng::{next_u32,next_u64}: a u64 is two consecutive words
 *     (low first) at the CURRENT index -- no alignment -- and may straddle a refill
 *   - rand 0.8.4 Standard f64 (53 bits * 2^-53), UniformFloat<f64>::sample_single
 *     ([1,2) from 52 bits, minus 1, * scale + low, retry if >= high), Standard bool (sign bit
 *     of next_u32), UniformInt<u32>::sample_single_inclusive (widening multiply, zone =
 *     (range << lz) - 1) as used by SliceRandom::choose via gen_index
 * Pins: RFC 7539 2.3.2-style ChaCha20 block vector (same quarter round, rounds = 20), the
 * ChaCha12 all-zero-key keystream of draft-strombergson-chacha-test-vectors (TC1), rand 0.8's
 * own `test_stdrng_construction` value (tests/test_refstream.py) -- and, end to end, the
 * reference's PNG itself.
 *
 * Elementary functions in this mode are the platform libm's (Rust's f64::sin etc. lower to
 * libm calls), not the deterministic ones of include/rt1w_num.h: an ulp there or in cgmath's
 * association only matters when it flips a branch, which happens with probability ~1e-13 per
 * decision -- invisible at the pixel level.
 */
#ifndef ORC_REFSTREAM_H
#define ORC_REFSTREAM_H

#include <cmath>
#include <cstdint>
#include <cstring>

struct RefRng {
    uint32_t key[8];
    uint64_t counter; /* block counter of the next refill's first block */
    uint32_t results[64];
    uint32_t index;   /* 64 = buffer exhausted */
};

static inline uint32_t ref_rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

/* one ChaCha block with `rounds` rounds (DJB variant: 64-bit counter in words 12/13, 64-bit nonce in 14/15) */
static inline void ref_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                      key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t x[16];
    std::memcpy(x, s, sizeof x);
#define REF_QR(a, b, c, d)                                   \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = ref_rotl(x[d], 16);   \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = ref_rotl(x[b], 12);   \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = ref_rotl(x[d], 8);    \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = ref_rotl(x[b], 7);
    for (int r = 0; r < rounds; r += 2) {
        REF_QR(0, 4, 8, 12) REF_QR(1, 5, 9, 13) REF_QR(2, 6, 10, 14) REF_QR(3, 7, 11, 15)
        REF_QR(0, 5, 10, 15) REF_QR(1, 6, 11, 12) REF_QR(2, 7, 8, 13) REF_QR(3, 4, 9, 14)
    }
#undef REF_QR
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

/* rand_chacha 0.3.1 ChaCha12Core::generate: four blocks, counters n..n+3, in order */
static inline void ref_refill(RefRng& r) {
    for (int b = 0; b < 4; ++b) ref_chacha_block(r.key, r.counter + (uint64_t)b, 0u, 12, r.results + 16 * b);
    r.counter += 4u;
}
static inline void ref_generate_and_set(RefRng& r, uint32_t index) { ref_refill(r); r.index = index; }

static inline RefRng ref_from_seed(const uint8_t seed[32]) {
    RefRng r;
    for (int i = 0; i < 8; ++i)
        r.key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    r.counter = 0u;
    std::memset(r.results, 0, sizeof r.results);
    r.index = 64u;
    return r;
}
/* rand_core 0.6.3 SeedableRng::seed_from_u64 */
static inline RefRng ref_seed_from_u64(uint64_t state) {
    uint8_t seed[32];
    for (int i = 0; i < 8; ++i) {
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        uint32_t x = (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
        seed[4 * i] = (uint8_t)x; seed[4 * i + 1] = (uint8_t)(x >> 8); seed[4 * i + 2] = (uint8_t)(x >> 16); seed[4 * i + 3] = (uint8_t)(x >> 24);
    }
    return ref_from_seed(seed);
}

/* rand_core 0.6.3 BlockRng::next_u32 / next_u64 */
static inline uint32_t rt_next_u32(RefRng& r) {
    if (r.index >= 64u) ref_generate_and_set(r, 0u);
    return r.results[r.index++];
}
static inline uint64_t rt_next_u64(RefRng& r) {
    uint32_t index = r.index;
    if (index < 63u) {
        r.index += 2u;
        return ((uint64_t)r.results[index + 1] << 32) | r.results[index];
    } else if (index >= 64u) {
        ref_generate_and_set(r, 2u);
        return ((uint64_t)r.results[1] << 32) | r.results[0];
    } else {
        uint64_t x = r.results[63];
        ref_generate_and_set(r, 1u);
        uint64_t y = r.results[0];
        return (y << 32) | x;
    }
}
/* rand 0.8.4 draw shapes (same text as include/rt1w_num.h's, on this generator) */
static inline double rt_gen_f64(RefRng& r) { return (double)(rt_next_u64(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline double rt_gen_range(RefRng& r, double low, double high) {
    double scale = high - low;
    for (;;) {
        uint64_t bits = (rt_next_u64(r) >> 12) | 0x3FF0000000000000ull;
        double v12; std::memcpy(&v12, &bits, 8);
        double res = (v12 - 1.0) * scale + low;
        if (res < high) return res;
    }
}
static inline bool rt_gen_bool(RefRng& r) { return (int32_t)rt_next_u32(r) < 0; }
static inline uint32_t rt_gen_below(RefRng& r, uint32_t n) {
    uint32_t lz = (uint32_t)__builtin_clz(n);
    uint32_t zone = (n << lz) - 1u;
    for (;;) {
        uint64_t m = (uint64_t)rt_next_u32(r) * n;
        if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
    }
}

#endif /* ORC_REFSTREAM_H */
