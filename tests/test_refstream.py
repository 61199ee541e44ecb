"""Pins the literal oracle to the reference's OWN output, pixel for pixel.

`rest_of_your_life.png` (README.md:19) is what master's `main` prints for its hard-wired Cornell arm (main.rs:815,868-870:
600x600, 100 spp, depth 50).  Every pixel's random stream is deterministic -- `MyRng::seed_from_u64((j*image_width+i))`,
main.rs:964, `type MyRng = StdRng`, main.rs:2 -- and nothing entropy-seeded reaches a Cornell pixel (bvh.rs:84 only reorders
the walk).  oracle/oracle.cpp built with -DORC_REFSTREAM (liborc_ref.so) swaps the numerical contract's Philox streams and
deterministic elementary functions for a restatement of that generator (oracle/refstream.h: ChaCha12, rand_core's
seed_from_u64 and BlockRng, rand 0.8.4's draw shapes) and libm; every other line -- scene, BVH, primitives, wrappers,
materials, pdfs, both integrators, the sample loop, into_sampled, the quantiser -- is the SAME source text as the
Philox-mode oracle that the GPU path is compared with.  Fixtures: tests/golden/cornell_png_pixels.json (+ crop),
generated from the PNG by tests/golden/make_golden.py."""
import ctypes as C
import hashlib
import json
import os
import zlib

import numpy as np

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
PIX = json.load(open(os.path.join(HERE, "golden", "cornell_png_pixels.json")))
CROP = np.load(os.path.join(HERE, "golden", "cornell_png_crop.npy"))


def _block(key_words, counter, rounds):
    key = (C.c_uint32 * 8)(*key_words)
    out = (C.c_uint32 * 16)()
    orc.REF.orc_ref_chacha_block(key, counter, rounds, out)
    return np.array(out, dtype="<u4").tobytes().hex()


def test_chacha_block_known_answers():
    """All-zero key/nonce/counter keystream blocks: ChaCha20 (RFC 7539 / djb), ChaCha12 and ChaCha8
    (draft-strombergson-chacha-test-vectors TC1).  Same quarter round, only the round count differs."""
    z = [0] * 8
    assert _block(z, 0, 20).startswith("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7")
    assert _block(z, 0, 12).startswith("9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f")
    assert _block(z, 0, 8).startswith("3e00ef2f895f40d67f5bb8e81f09a5a1")
    assert _block(z, 1, 20).startswith("9f07e7be5551387a98ba977c732d080dcb0f29a048e3656912c6533e32ee7aed")


def test_stdrng_construction_value_of_rand_0_8():
    """rand 0.8 src/rngs/std.rs `test_stdrng_construction`: the two u64 the crate itself asserts for StdRng (= ChaCha12):
    pins rounds, word order inside the 4-block buffer, the little-endian key, next_u64 and fill_bytes."""
    seed = (C.c_uint8 * 32)(1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0, *([0] * 16))
    out = (C.c_uint64 * 2)()
    orc.REF.orc_ref_stdrng_construction(seed, out)
    assert [out[0], out[1]] == [10719222850664546238, 14064965282130556830]


def test_block_rng_u64_is_unaligned_and_straddles_refills():
    """rand_core BlockRng::next_u64: two consecutive words at the current index, low word first, across a refill too."""
    n = 200
    w = (C.c_uint64 * n)()
    orc.REF.orc_ref_words(12345, 0, 0, n, w)
    words = [int(x) for x in w]
    for skip in (0, 1, 63, 64, 127):
        q = (C.c_uint64 * 8)()
        orc.REF.orc_ref_words(12345, 1, skip, 8, q)
        for i in range(8):
            assert int(q[i]) == words[skip + 2 * i] | (words[skip + 2 * i + 1] << 32)


def _top_down_u8(sc, means):
    return np.ascontiguousarray(sc.quantize(means)[::-1])      # P3 rows are j = H-1 first (main.rs:959 `.rev()`)


def test_reference_stream_oracle_reproduces_the_reference_png_crop():
    """128x128 crop (glass sphere, box edge, floor, wall), all checkpoints in one pass: 100 spp equals the PNG on every
    pixel; 50 and 200 spp do not, outside the black box face (the PNG's spp is the shipped default, main.rs:870)."""
    assert orc.REF.orc_is_refstream() == 1 and orc.A.orc_is_refstream() == 0
    sc = orc.OracleScene(5, build_seed=1, refstream=True)
    x0, y0t, w, h = PIX["crop_top_down"]
    tile = (x0, 600 - y0t - h, w, h)                           # rows of the tile are bottom-up j
    frames, _ = sc.render_checkpoints(600, 600, [50, 100, 200], tile=tile)
    eq = [float((_top_down_u8(sc, f) == CROP).all(axis=2).mean()) for f in frames]
    assert eq[1] == 1.0, f"fraction of equal pixels at 100 spp: {eq[1]}"
    assert eq[0] < 0.8 and eq[2] < 0.8, eq


def test_reference_stream_oracle_reproduces_the_reference_png_full_frame():
    """The whole 600x600 frame at 100 spp: sha256 of the RGB bytes equals the PNG's, i.e. 360 000 of 360 000 pixels are
    equal.  Also independent of the BVH build seed (bvh.rs:84 is entropy-seeded in the reference): a second seed gives
    the same bytes."""
    for seed in (1, 20211003):
        sc = orc.OracleScene(5, build_seed=seed, refstream=True)
        img, st = sc.render(600, 600, 100)
        q = _top_down_u8(sc, img)
        crc = [zlib.crc32(q[r].tobytes()) for r in range(600)]
        bad = [r for r in range(600) if crc[r] != PIX["row_crc32_top_down"][r]]
        assert not bad, f"{len(bad)} rows differ from the reference PNG, first: {bad[:5]}"
        assert hashlib.sha256(q.tobytes()).hexdigest() == PIX["sha256_rgb_top_down"]


def test_philox_mode_matches_the_png_at_the_monte_carlo_noise_level():
    """The generator swap (ChaCha12 per pixel -> Philox per sample) is the ONLY difference between the pinned build and
    the build the GPU is compared with, and it must be invisible statistically: at the PNG's own 100 spp and full
    resolution, global 8-bit channel means within 0.15/255 and 100x100 block means within 1.5/255 of the PNG's
    (measured: see DESIGN.md section 3)."""
    sc = orc.OracleScene(5, build_seed=1)
    img, _ = sc.render(600, 600, 100)
    q = _top_down_u8(sc, img).astype(np.float64)
    blocks = json.load(open(os.path.join(HERE, "golden", "cornell_png_blocks.json")))
    d = q.mean(axis=(0, 1)) - np.array(blocks["channel_means"])
    assert np.abs(d).max() < 0.15, d
    b = q.reshape(6, 100, 6, 100, 3).mean(axis=(1, 3)) - np.array(blocks["block_means_top_down"])
    assert np.abs(b).max() < 1.5, np.abs(b).max()


def test_kernel_core_with_the_reference_stream_reproduces_the_png_crop(rt):
    """The DEVICE core (rt_core.h + the flattened scene) compiled for the CPU with RT_RNG_REFSTREAM -- the same text
    csrc/context_ref.hip builds for the GPU (RT1W_RNG_REFERENCE) -- on the committed crop of the reference's PNG: every
    pixel equal.  The deterministic elementary functions of include/rt1w_num.h stand in for libm here, so this also shows
    that an ulp in sin/cos/acos/atan2 does not reach a pixel."""
    lib = orc.flat_ref_lib()
    sc = rt.Scene.reference(5, build_seed=1)
    x0, y0t, w, h = PIX["crop_top_down"]
    tile = (x0, 600 - y0t - h, w, h)
    img, _ = orc.flat_render(sc, 600, 600, 100, tile=tile, chunk=100, lib=lib)
    q = rt.quantize(img)[::-1]
    assert np.array_equal(q, CROP), float((q == CROP).all(axis=2).mean())
