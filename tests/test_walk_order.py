"""Opt-in traversal order (rt1w_scene_set_walk_order): near child first.

Default = the reference's left-then-right (bvh.rs:38-47).  RT1W_WALK_NEAR_FAR reorders only subtrees with neither a
ConstantMedium (it draws random numbers while visited, constant_medium.rs:85) nor a MovingSphere (a scattered ray carries
time = hit t, main.rs:86,145, which moves the sphere outside its BVH box, moving_sphere.rs:23-26,72-84) below them, and
ties in t still go to the larger pre-order index = the primitive the reference tests later: frames must be BIT-IDENTICAL
to the reference-order walk on every reference arm and on the random graphs.  RT1W_WALK_NEAR_FAR_ALL also reorders subtrees
with moving spheres and is documented as not result-preserving: the test records that it does differ on random_scene."""
import numpy as np
import pytest

import orc
from dual import random_scene_pair

ARMS = {0: (96, 64, 4), 1: (48, 28, 4), 2: (48, 28, 4), 3: (48, 28, 4), 4: (48, 28, 8), 5: (48, 48, 8), 6: (48, 48, 8), 7: (56, 56, 6)}


def _order_bits(scene):
    kinds = scene.flat(0).view(np.uint32).reshape(-1, 24)[:, 0]
    return int((((kinds >> 9) & 3) != 0).sum()), int(((kinds & 0xFF) == 0).sum())


@pytest.mark.parametrize("arm", sorted(ARMS))
def test_near_far_order_is_bit_identical_on_the_reference_arms(rt, arm):
    W, H, spp = ARMS[arm]
    aspect = 1.5 if arm == 0 else None
    ref_scene = rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect)
    a, sa = orc.flat_render(ref_scene, W, H, spp, variant=3)            # the stack walk (variant 3 covers every feature)
    nf_scene = rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect).set_walk_order(1)
    b, sb = orc.flat_render(nf_scene, W, H, spp, variant=4)     # V4 = V3 + the order bits honoured
    assert sa["segments"] == sb["segments"] and np.array_equal(a, b, equal_nan=True), int((a != b).any(axis=2).sum())
    ordered, bvh2 = _order_bits(nf_scene)
    assert _order_bits(ref_scene)[0] == 0
    if arm == 7:
        assert ordered > 0.9 * bvh2              # the 400 boxes and the 1000-sphere cluster are static and media-free
    if arm == 0:
        assert ordered < 0.2 * bvh2              # moving spheres nearly everywhere
    nf_scene.set_walk_order(0)
    assert _order_bits(nf_scene)[0] == 0


def test_near_far_order_on_random_graphs(rt):
    """24 random graphs (mirror boxes, nested wrappers, media, every primitive): the FRAMES are bit-identical on all of them.
    It is an opt-in because identity is not provable: on axis-aligned geometry the entry t of one face's box and the hit t
    of the neighbouring face coincide, so which of two edge-on hits survives a prune can depend on the order -- graph
    3014 has 3 of 2240 paths whose LENGTH differs (50 against 8-21 segments; all of them end black).  Recorded, not hidden:
    at most 1 graph of the 24 may show a different segment count."""
    seg_diff = 0
    for seed in range(24):
        prod, _ = random_scene_pair(3000 + seed)
        W, H, spp = 28, 20, 4
        a, sa = orc.flat_render(prod, W, H, spp, variant=3)
        prod.set_walk_order(1)
        b, sb = orc.flat_render(prod, W, H, spp, variant=4)
        assert np.array_equal(a, b, equal_nan=True), seed
        seg_diff += sa["segments"] != sb["segments"]
    assert seg_diff <= 1, seg_diff


def test_near_far_everywhere_is_not_result_preserving_with_moving_spheres(rt):
    """Why mode 1 excludes moving spheres: on random_scene mode 2 visits fewer nodes but changes pixels."""
    W, H, spp = 120, 80, 4
    ref = rt.Scene.reference(0, build_seed=1, aspect_ratio=1.5)
    a, sa = orc.flat_render(ref, W, H, spp, variant=3)
    alln = rt.Scene.reference(0, build_seed=1, aspect_ratio=1.5).set_walk_order(2)
    b, sb = orc.flat_render(alln, W, H, spp, variant=4)
    differing = int((a != b).any(axis=2).sum())
    assert 0 < differing < 0.01 * W * H, differing


@pytest.mark.gpu
def test_near_far_order_on_the_gpu(rt, gpu_ctx_factory):
    """final_scene through the HIP stack walk and the wavefront form: bit-identical to the reference order."""
    W, H, spp = 96, 96, 8
    ref = gpu_ctx_factory(rt.Scene.reference(7, build_seed=1))
    a, sa = ref.render(W, H, spp)
    nf = gpu_ctx_factory(rt.Scene.reference(7, build_seed=1).set_walk_order(1))
    b, sb = nf.render(W, H, spp)
    c, sc_ = nf.render(W, H, spp, wavefront=True)
    assert sa["segments"] == sb["segments"] == sc_["segments"]
    assert np.array_equal(a, b, equal_nan=True) and np.array_equal(a, c, equal_nan=True)
