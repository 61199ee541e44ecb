"""The flattened scene + iterative integrator (CPU build of the DEVICE core, oracle_flat)
against the literal recursive object-graph oracle, on every scene arm of the reference's
`main` table -- proves the flattener and the recursion->iteration rewrite, without a GPU.

Tolerance: the two forms follow the same path (same draws, same geometry arithmetic:
segment counts must be EQUAL) and differ only in the association of the colour
products, so per channel |a-b| <= 1e-12 * |a|."""
import json
import os

import numpy as np
import pytest

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
RTOL = 1e-12

CASES = {0: (60, 40, 8), 1: (48, 27, 8), 2: (48, 27, 8), 3: (48, 27, 8), 4: (48, 27, 16), 5: (64, 64, 16), 6: (48, 48, 8),
         7: (48, 48, 8)}


def close(a, b):
    both_nan = np.isnan(a) & np.isnan(b)
    ok = both_nan | (np.abs(a - b) <= RTOL * np.abs(a)) | (a == b)
    return bool(ok.all())


@pytest.mark.parametrize("arm", sorted(CASES))
def test_flat_iterative_equals_literal_recursive(rt, arm):
    W, H, spp = CASES[arm]
    sc = rt.Scene.reference(arm, build_seed=1)
    oa = orc.OracleScene(arm, build_seed=1)
    a, sa = oa.render(W, H, spp)
    b, sb = orc.flat_render(sc, W, H, spp)
    assert sa["segments"] == sb["segments"], "paths diverged: traversal order or draw schedule differs"
    assert close(a, b)
    info = sc.info()
    assert sb["max_stack"] <= info["stack_need"] <= 32, "flattener's stack bound must cover what traversal uses"


def test_other_build_seed_and_global_seed(rt):
    for seed, gseed in ((7, 0), (1, 5)):
        sc = rt.Scene.reference(0, build_seed=seed)
        oa = orc.OracleScene(0, build_seed=seed)
        a, sa = oa.render(40, 30, 4, global_seed=gseed)
        b, sb = orc.flat_render(sc, 40, 30, 4, global_seed=gseed)
        assert sa["segments"] == sb["segments"] and close(a, b)


def test_depth_limit_returns_black_not_background(rt):
    # main.rs:59-61: at depth 0 the recursion returns (0,0,0); with depth 1 only directly
    # visible emitters/background contribute
    sc = rt.Scene.reference(5, build_seed=1)
    oa = orc.OracleScene(5, build_seed=1)
    a, sa = oa.render(32, 32, 4, max_depth=1)
    b, sb = orc.flat_render(sc, 32, 32, 4, max_depth=1)
    assert sa["segments"] == sb["segments"] == 32 * 32 * 4 and close(a, b)
    assert set(np.unique(a)).issubset({0.0, 15.0, 7.5, 3.75, 11.25})   # only the light (15) averaged over 4 samples


def test_tile_and_sample_offset_match_full_render(rt):
    sc = rt.Scene.reference(5, build_seed=1)
    full, _ = orc.flat_render(sc, 40, 40, 8, chunk=4)
    tile, _ = orc.flat_render(sc, 40, 40, 8, chunk=4, tile=(8, 16, 24, 10))
    assert np.array_equal(tile, full[16:26, 8:32])
    # samples 0..3 + samples 4..7 as raw sums == chunked full sum
    s0, _ = orc.flat_render(sc, 40, 40, 4, chunk=4, out_sum=True)
    s1, _ = orc.flat_render(sc, 40, 40, 4, chunk=4, out_sum=True, sample_offset=4)
    assert np.array_equal(rt.resolve(s0 + s1, 8), full)
    # and the literal oracle agrees on the tile + offset semantics
    oa = orc.OracleScene(5, build_seed=1)
    a, _ = oa.render(40, 40, 4, tile=(8, 16, 24, 10), sample_offset=4, out_sum=True)
    assert close(a, s1[16:26, 8:32])


def test_c1_cornell_200x200x64_against_golden(rt):
    """BASELINE config C1 in full: flat core vs the committed golden of the literal oracle."""
    meta = json.load(open(os.path.join(HERE, "golden", "oracle_frames.json")))["cases"]["c1_cornell_200x200x64"]
    gold = np.load(os.path.join(HERE, "golden", "oracle_frames.npz"))
    sc = rt.Scene.reference(5, build_seed=1)
    b, sb = orc.flat_render(sc, 200, 200, 64)
    assert sb["segments"] == meta["segments"]
    x0, y0, w, h = meta["crop"]
    assert close(gold["c1_cornell_200x200x64__crop"], b[y0:y0 + h, x0:x0 + w])
    bm = b.reshape(10, 20, 10, 20, 3).mean(axis=(1, 3))
    assert np.allclose(bm, gold["c1_cornell_200x200x64__block20"], rtol=1e-11, atol=0)


VALID = {0: (1, 2, 3, 5), 1: (1, 3), 2: (1, 3), 3: (1, 3), 4: (1, 3), 5: (0, 1, 2, 3), 6: (1, 3), 7: (1, 3)}   # V5: no media, no wrapper node


@pytest.mark.parametrize("arm", sorted(CASES))
def test_every_valid_kernel_variant_gives_identical_bits(rt, arm):
    """V0/V1 = stackless pre-order sweep, V2/V3/V5 = stack walk; feature-pruned or full.  All must visit the
    same nodes in the same order with the same arithmetic: bit-identical framebuffers, equal segment counts."""
    W, H, spp = CASES[arm]
    sc = rt.Scene.reference(arm, build_seed=1)
    info = sc.info()
    assert info["variant"] in VALID[arm]
    ref, sref = orc.flat_render(sc, W, H, spp, variant=3)
    for v in VALID[arm]:
        img, st = orc.flat_render(sc, W, H, spp, variant=v)
        assert st["segments"] == sref["segments"], (arm, v)
        assert np.array_equal(img, ref, equal_nan=True), (arm, v)


def test_variant_selection_rule(rt):
    assert rt.Scene.reference(5).info()["variant"] == 0      # Cornell: 30 nodes, solid colours, no media
    assert rt.Scene.reference(6).info()["variant"] == 1      # cornel_smoke: small, media
    assert rt.Scene.reference(0).info()["variant"] == 5      # random_scene: ~1000 nodes, checker, moving spheres, no wrapper node (V2 would do too)
    assert rt.Scene.reference(7).info()["variant"] == 3      # final_scene
    i = rt.Scene.reference(0).info()
    assert i["has_textures"] == 1 and i["has_moving"] == 1 and i["has_media"] == 0
