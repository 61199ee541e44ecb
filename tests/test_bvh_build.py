"""The BVH builds (rt1w_scene_set_bvh_build).

RT1W_BVH_REFERENCE ("reference" below) = `BVHNode::new` as written: an axis DRAWN per node (from the build seed; the reference draws it
from an entropy-seeded generator), sort by box minimum, median split (bvh.rs:84-100).  RT1W_BVH_BEST_AXIS, the default since round 4,
is the same rule with the axis chosen (second half of this file).  RT1W_BVH_SAH rebuilds the tree over every BVH's leaf set.  What must hold: the same leaves (every primitive / wrapper / medium exactly once), boxes that
are the `surrounding_box` of their children (aabb.rs:42-55), frames BIT-IDENTICAL to the reference build wherever the walk
order cannot matter (static, media-free arms at test size) and statistically equal where it can (moving spheres -- quirk Q1,
main.rs:86,145 -- and media, constant_medium.rs:85); switching back restores the reference build byte for byte.  The kernels
are the same kernels: the GPU frame of a rebuilt scene equals the CPU build of the core on it, bit for bit."""
import numpy as np
import pytest

import orc
from dual import random_scene_pair

ARMS = {0: (96, 64, 8), 1: (48, 28, 4), 2: (48, 28, 4), 3: (48, 28, 4), 4: (48, 28, 8), 5: (48, 48, 8), 6: (48, 48, 8), 7: (56, 56, 6)}
KIND_BVH2, KIND_BVH1 = 0, 1


def _drawn(rt, arm, aspect):
    """the scene arm with the axes of bvh.rs:84 drawn from build seed 1 (RT1W_BVH_REFERENCE): round 1-3's default"""
    return rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect).set_bvh_build("reference")


def _nodes(scene):
    raw = scene.flat(0)
    return raw.view(np.uint32).reshape(-1, 24), raw.view(np.float64).reshape(-1, 12)


def _leaf_multiset(scene):
    """(kind, d[0..5], e[0..2], mat) of every non-BVH record, sorted: what the tree is built OVER."""
    u, f = _nodes(scene)
    kinds = u[:, 0] & 0xFF
    rows = [bytes(u[i, 0:1] & 0x1FF) + bytes(f[i, 1:7]) + bytes(f[i, 8:11]) + bytes(u[i, 15:16]) for i in range(len(u)) if kinds[i] > KIND_BVH1]
    return sorted(rows)


@pytest.mark.parametrize("arm", sorted(ARMS))
def test_sah_build_keeps_the_leaves_and_nests_the_boxes(rt, arm):
    aspect = 1.5 if arm == 0 else None
    ref = _drawn(rt, arm, aspect)
    sah = rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect).set_bvh_build(True)
    assert _leaf_multiset(ref) == _leaf_multiset(sah)
    u, f = _nodes(sah)
    kinds = u[:, 0] & 0xFF
    # the reference's shape is kept: an object sits under a BVHChild::One or under a two-object Two, never next to a subtree
    # (bvh.rs:63-79) -- what the pair walk (rt_walk_pair.h) relies on
    for i in np.nonzero(kinds == KIND_BVH2)[0]:
        a, b = int(u[i, 22]), int(u[i, 14])
        assert (kinds[a] <= KIND_BVH1) == (kinds[b] <= KIND_BVH1), (arm, i)
    for i in np.nonzero(kinds == KIND_BVH2)[0]:
        a, b = int(u[i, 22]), int(u[i, 14])                          # children (RtNode.a, .b), pre-order: a = i + 1
        assert a == i + 1 and b == int(u[a, 1])                      # .skip of the left child
        for c in (a, b):
            if kinds[c] <= KIND_BVH1:                                # a child box lies inside its parent's (an AABox's side BVH: up to the
                #                                                      rects' 0.0001 pad, aabox.rs:98-103 against aarect.rs:74-79)
                assert (f[c, 1:4] >= f[i, 1:4] - 2e-4).all() and (f[c, 4:7] <= f[i, 4:7] + 2e-4).all()
    # back to the reference build: the very same bytes
    raw_ref = ref.flat(0).tobytes()
    assert sah.set_bvh_build(False).flat(0).tobytes() == raw_ref


@pytest.mark.parametrize("arm", sorted(ARMS))
def test_sah_build_frames(rt, arm):
    W, H, spp = ARMS[arm]
    aspect = 1.5 if arm == 0 else None
    ref = _drawn(rt, arm, aspect)
    sah = rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect).set_bvh_build(True)
    a, sa = orc.flat_render(ref, W, H, spp)
    b, sb = orc.flat_render(sah, W, H, spp)
    differing = int((a != b).any(axis=2).sum())
    if arm in (1, 2, 3, 4, 5, 6):
        # static scenes: the closest hit does not depend on the tree (cornel_smoke's two media are reached in the same order)
        assert differing == 0 and sa["segments"] == sb["segments"], (differing, sa["segments"], sb["segments"])
    else:
        # moving spheres (Q1) / media in a big BVH: other paths for a few rays, the same picture
        assert differing < 0.05 * W * H, differing
        assert abs(np.nanmean(a) - np.nanmean(b)) < 0.01 * np.nanmean(a)
        assert abs(sa["segments"] - sb["segments"]) < 0.01 * sa["segments"]


RTOL = 1e-12


def _close(a, b):
    both_nan = np.isnan(a) & np.isnan(b)
    return bool((both_nan | (np.abs(a - b) <= RTOL * np.abs(a)) | (a == b)).all())


@pytest.mark.parametrize("arm", sorted(ARMS))
def test_sah_trees_against_the_literal_oracle_on_the_same_topology(rt, arm):
    """INDEPENDENT check of the rebuilt trees (round-2 review: the SAH frames of arms 0 and 7 were only compared with the CPU
    build of the product's own core).  The product writes its trees down (rt1w_scene_get_bvh_topology); the literal oracle rebuilds
    ITS object graph over the same leaves with those trees -- boxes by its own surrounding_box, walk by its own recursive
    BVHNode::hit (bvh.rs:25-50) -- and the iterative core on the product's flattened SAH scene must reproduce it like it does
    on the reference's trees: equal segment counts, <= 1e-12 relative."""
    W, H, spp = ARMS[arm]
    aspect = 1.5 if arm == 0 else None
    sah = rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect).set_bvh_build(True)
    topo = sah.bvh_topology()
    assert (topo.size > 0 and (topo == -1).sum() >= 1) or arm == 3       # earth: one sphere under BVHChild::One, kept as built
    # the stream numbers a BVH's leaves through the tree the DRAWN axes built: that is the tree the oracle must start from
    oracle = orc.OracleScene(arm, build_seed=1, aspect_ratio=aspect, best_axis=False).apply_topology(topo)
    a, sa = oracle.render(W, H, spp)
    b, sb = orc.flat_render(sah, W, H, spp)
    assert sa["segments"] == sb["segments"], (arm, sa["segments"], sb["segments"])
    assert _close(a, b), arm
    # the reference build has no stream; a stream that does not fit is refused
    assert _drawn(rt, arm, aspect).bvh_topology().size == 0
    if topo.size:
        bad = orc.OracleScene(arm, build_seed=1, aspect_ratio=aspect, best_axis=False)
        assert bad.lib.orc_scene_apply_topology(bad._h, topo[:-1].ctypes.data_as(orc._P), topo.size - 1) != 0


def test_sah_trees_of_random_graphs_against_the_literal_oracle(rt):
    """The same on 24 random graphs: nested BVHs merged into their parent's leaf set, BVHs under wrappers and inside AABoxes,
    one-object BVHs (kept as built), media boundaries."""
    for seed in range(24):
        prod, oracle = random_scene_pair(3000 + seed, best_axis=False)
        prod.set_bvh_build(True)
        oracle.apply_topology(prod.bvh_topology())
        W, H, spp = 28, 20, 4
        a, sa = oracle.render(W, H, spp)
        b, sb = orc.flat_render(prod, W, H, spp, variant=3)
        assert sa["segments"] == sb["segments"] and _close(a, b), seed


def test_sah_build_on_random_graphs(rt):
    """24 random graphs (mirror boxes, nested wrappers, media, every primitive)."""
    identical, worst = 0, 0.0
    for seed in range(24):
        prod, _ = random_scene_pair(3000 + seed, best_axis=False)
        W, H, spp = 28, 20, 4
        a, sa = orc.flat_render(prod, W, H, spp, variant=3)
        leaves = _leaf_multiset(prod)
        prod.set_bvh_build(True)
        assert _leaf_multiset(prod) == leaves, seed
        b, sb = orc.flat_render(prod, W, H, spp, variant=3)
        same = np.array_equal(a, b, equal_nan=True)
        identical += same
        if not same:
            worst = max(worst, int((a != b).any(axis=2).sum()) / (W * H))
            assert abs(np.nanmean(a) - np.nanmean(b)) < 0.05 * max(np.nanmean(a), 1e-3), seed
    print("identical", identical, "of 24; worst differing fraction", worst)
    assert identical >= 18 and worst < 0.25, (identical, worst)


def test_sah_build_with_near_far_order(rt):
    """The order annotations are recomputed on the rebuilt tree (they live in the node records)."""
    W, H, spp = ARMS[7]
    sah = rt.Scene.reference(7, build_seed=1).set_bvh_build(True)
    a, sa = orc.flat_render(sah, W, H, spp, variant=3)
    sah.set_walk_order(1)
    kinds = _nodes(sah)[0][:, 0]
    assert int((((kinds >> 9) & 3) != 0).sum()) > 0
    b, sb = orc.flat_render(sah, W, H, spp, variant=4)
    assert np.array_equal(a, b, equal_nan=True) and sa["segments"] == sb["segments"]
    nf_first = rt.Scene.reference(7, build_seed=1).set_walk_order(1).set_bvh_build(True)      # the other call order
    assert nf_first.flat(0).tobytes() == sah.flat(0).tobytes()


def test_sah_build_errors(rt):
    lib = rt._lib
    sc = rt.Scene.reference(5, build_seed=1)
    assert lib.rt1w_scene_set_bvh_build(sc._h, 3) < 0 and b"unknown" in lib.rt1w_last_error()
    assert lib.rt1w_scene_set_bvh_build(None, 1) < 0


# ---- RT1W_BVH_BEST_AXIS: BVHNode::new as written, the axis of bvh.rs:84 chosen instead of drawn ----------------------------------

@pytest.mark.parametrize("arm", sorted(ARMS))
def test_best_axis_trees_are_the_reference_rule_and_equal_the_oracles_own(rt, arm):
    """The best-axis build keeps the reference's rule (bvh.rs:60-100: stable sort by box minimum, split at len/2, one- and two-object
    shapes, nested BVHNode::new calls separate) and only picks the axis.  Checked three ways: (1) same leaves, children boxes inside
    their parent's, the reference's shape; (2) the literal oracle rebuilt over the product's topology stream (merge_nested=False) and
    (3) the literal oracle built NATIVELY with its own statement of the rule (oracle.cpp: BVHNode::best_axis) render the same bits, and
    the CPU build of the core on the product's flattened scene reproduces them: equal segment counts, <= 1e-12 relative."""
    W, H, spp = ARMS[arm]
    aspect = 1.5 if arm == 0 else None
    ref = _drawn(rt, arm, aspect)
    ba = rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect)
    assert ba.flat(0).tobytes() == rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect).set_bvh_build("best_axis").flat(0).tobytes()   # the default
    assert _leaf_multiset(ref) == _leaf_multiset(ba)
    assert len(_nodes(ref)[0]) == len(_nodes(ba)[0])        # a median split per node: as many nodes as the reference's own tree
    u, f = _nodes(ba)
    kinds = u[:, 0] & 0xFF
    for i in np.nonzero(kinds == KIND_BVH2)[0]:
        a, b = int(u[i, 22]), int(u[i, 14])
        assert a == i + 1 and b == int(u[a, 1])
        for c in (a, b):
            if kinds[c] <= KIND_BVH1:
                assert (f[c, 1:4] >= f[i, 1:4] - 2e-4).all() and (f[c, 4:7] <= f[i, 4:7] + 2e-4).all()
    topo = ba.bvh_topology()
    by_topology = orc.OracleScene(arm, build_seed=1, aspect_ratio=aspect, best_axis=False).apply_topology(topo, merge_nested=False)
    native = orc.OracleScene(arm, build_seed=1, aspect_ratio=aspect, best_axis=True)
    a, sa = by_topology.render(W, H, spp)
    n, sn = native.render(W, H, spp)
    b, sb = orc.flat_render(ba, W, H, spp)
    assert np.array_equal(a, n, equal_nan=True) and sa["segments"] == sn["segments"], arm
    assert sa["segments"] == sb["segments"] and _close(a, b), arm
    # static media-free arms: the frame does not depend on the tree; back to the reference build: the very same bytes
    if arm in (1, 2, 3, 4, 5, 6):
        r, sr = orc.flat_render(ref, W, H, spp)
        assert np.array_equal(r, b, equal_nan=True)
    assert ba.set_bvh_build(False).flat(0).tobytes() == ref.flat(0).tobytes()
    # what the build seed draws for the axes does not reach the tree (it still reaches what the scene builders draw afterwards)
    if arm in (1, 2, 4, 5, 6):
        assert rt.Scene.reference(arm, build_seed=7, aspect_ratio=aspect).flat(0).tobytes() == rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect).flat(0).tobytes()


def test_best_axis_trees_of_random_graphs_against_the_literal_oracle(rt):
    """The same on 24 random graphs: BVHs inside BVHs (kept separate), under wrappers, inside AABoxes, one- and two-object BVHs."""
    for seed in range(24):
        prod, oracle = random_scene_pair(3000 + seed, best_axis=False)     # both sides draw, then the product's trees go over as a stream
        leaves = _leaf_multiset(prod)
        prod.set_bvh_build("best_axis")
        assert _leaf_multiset(prod) == leaves, seed
        oracle.apply_topology(prod.bvh_topology(), merge_nested=False)
        W, H, spp = 28, 20, 4
        a, sa = oracle.render(W, H, spp)
        b, sb = orc.flat_render(prod, W, H, spp, variant=3)
        assert sa["segments"] == sb["segments"] and _close(a, b), seed


def test_best_axis_picks_the_cheapest_median_split(rt):
    """Known answer: four unit spheres in a row along z (and a fifth far away on z) have one good median split -- by z.  The root's
    children must separate low z from high z whatever the build seed draws."""
    lib = rt._lib
    for seed in (1, 2, 3, 7):
        sc = rt.Scene(build_seed=seed)
        m = sc.lambertian(sc.solid_color((0.5, 0.5, 0.5)))
        ids = [sc.sphere((0.1 * ((k * 7) % 5), 0.05 * ((k * 3) % 5), z), 0.5, m) for k, z in enumerate((0.0, 30.0, 10.0, 40.0, 20.0))]
        sc.set_world(sc.bvh_node(ids))
        sc.set_camera((0, 0, -10), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
        sc.commit()
        sc.set_bvh_build("best_axis")
        u, f = _nodes(sc)
        kinds = u[:, 0] & 0xFF
        root = 0
        assert kinds[root] == KIND_BVH2
        a, b = int(u[root, 22]), int(u[root, 14])
        # left = the len/2 = 2 lowest z (0, 10), right = the rest (20, 30, 40)
        assert f[a, 6] < 15.0 and f[b, 3] > 15.0, (seed, f[a, 1:7], f[b, 1:7])


@pytest.mark.gpu
def test_sah_build_on_the_gpu(rt, gpu_ctx_factory):
    """The HIP kernels on rebuilt scenes: stack walk (random_scene, final_scene; also near-far on top and the wavefront form) and
    the scene-specialised sweep (Cornell: another topology, another generated kernel) -- bit-identical to the CPU build of the core."""
    for arm, W, H, spp, aspect in ((0, 96, 64, 8, 1.5), (7, 64, 64, 8, None), (5, 64, 64, 16, None)):
        sc = rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect).set_bvh_build(True)
        # the independent check first: the literal oracle over the product's trees (reference walk order on both sides)
        lit, sl = orc.OracleScene(arm, build_seed=1, aspect_ratio=aspect, best_axis=False).apply_topology(sc.bvh_topology()).render(W, H, spp)
        ctx0 = gpu_ctx_factory(sc)
        g0, s0 = ctx0.render(W, H, spp, generic=True)
        assert s0["segments"] == sl["segments"] and _close(lit, g0), arm
        ctx0.close()
        if arm == 7:
            sc.set_walk_order(1)
        want, sw = orc.flat_render(sc, W, H, spp)
        ctx = gpu_ctx_factory(sc)
        if arm == 5:
            assert ctx.specialise()["active"]
        got, sg = ctx.render(W, H, spp)
        assert np.array_equal(got, want, equal_nan=True) and sg["segments"] == sw["segments"], arm
        assert _close(lit, got), arm   # near-far on top (arm 7) must still give the literal oracle's frame
        if arm != 5:
            wf, swf = ctx.render(W, H, spp, wavefront=True)
            assert np.array_equal(wf, want, equal_nan=True) and swf["segments"] == sw["segments"], arm
            f32, s32 = ctx.render(W, H, spp, f32=True)
            assert abs(np.nanmean(f32) - np.nanmean(want)) < 0.02 * np.nanmean(want)


@pytest.mark.gpu
def test_best_axis_build_on_the_gpu(rt, gpu_ctx_factory):
    """The HIP kernels on the best-axis trees (pair walk on random_scene, sliced stack walk + slice-end reordering on final_scene, the
    scene-specialised sweep on Cornell): bit-identical to the CPU build of the core on the same flattened scene, and within 1e-12 of
    the literal oracle that built the same trees by its OWN statement of the rule (equal segment counts)."""
    for arm, W, H, spp, aspect in ((0, 96, 64, 8, 1.5), (7, 64, 64, 8, None), (5, 64, 64, 16, None)):
        sc = rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect).set_bvh_build("best_axis")
        lit, sl = orc.OracleScene(arm, build_seed=1, aspect_ratio=aspect, best_axis=True).render(W, H, spp)
        want, sw = orc.flat_render(sc, W, H, spp)
        ctx = gpu_ctx_factory(sc)
        if arm == 5:
            assert ctx.specialise()["active"]
        got, sg = ctx.render(W, H, spp)
        assert np.array_equal(got, want, equal_nan=True) and sg["segments"] == sw["segments"] == sl["segments"], arm
        assert _close(lit, got), arm
        if arm == 0:
            assert sg["sorted"] & 128, "random_scene on the best-axis tree did not run the pair-walk kernel"
        ctx.close()
