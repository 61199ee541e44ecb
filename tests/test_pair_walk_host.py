"""The pair walk of sphere scenes (raytracing-1w_amd/csrc/rt_walk_pair.h, the default walk of random_scene / BASELINE C2) on the CPU:
its lane functions are compiled by g++ with bound-checked stacks and queues (oracle/oracle_flat.cpp: orcflat_pair_walk) and run ray by
ray under RANDOM schedules against the one-entry-per-step walk.  What must hold: the same closest hit bit for bit whatever the order of
box work and leaf work (the exactness argument of rt_walk_pair.h), no index outside the per-lane stack (RT_PW_SS_STACK = 12 entries in the
kernel that reorders) or the queue ring (RT_PW_QCAP = 8), and the two NaN hand-overs to the classic walk -- a NaN shutter fraction and a
closest hit that turns NaN -- reached and answered like the classic walk.  Also run under ASan + UBSan (test_sanitizers.py)."""
import ctypes as C

import numpy as np
import pytest

import orc

_P = C.c_void_p


def pair_walk(lib, scene, rays, stack_cap=12, seed=1):
    nodes = scene.flat(0)
    root = int(np.frombuffer(scene.flat(6).tobytes()[-8:-4], dtype=np.uint32)[0])
    n = rays.shape[0]
    r = np.ascontiguousarray(rays, dtype=np.float64)
    t, rt_ = np.empty(n), np.empty(n)
    prim, rprim, flags = np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.uint32)
    lib.orcflat_pair_walk.restype = C.c_int
    lib.orcflat_pair_walk.argtypes = [_P, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.c_uint32, C.c_uint64, _P, _P, _P, _P, _P]
    rc = lib.orcflat_pair_walk(nodes.ctypes.data_as(_P), scene.info()["n_nodes"], root, r.ctypes.data_as(_P), n, stack_cap, seed,
                               t.ctypes.data_as(_P), prim.ctypes.data_as(_P), rt_.ctypes.data_as(_P), rprim.ctypes.data_as(_P), flags.ctypes.data_as(_P))
    assert rc == 0, "scene outside the pair walk's scope"
    return t, prim, rt_, rprim, flags


def rays_for(scene_seed, n, rng):
    """camera-like rays, rays from inside the scene (bounces), and the nasty ones"""
    o = np.empty((n, 3)); d = np.empty((n, 3)); tm = rng.uniform(0.0, 1.0, n)
    o[:] = (13.0, 2.0, 3.0)
    tgt = np.stack([rng.uniform(-11, 11, n), rng.uniform(0, 1.2, n), rng.uniform(-11, 11, n)], axis=1)
    d[:] = tgt - o
    k = n // 3                                              # bounce rays: origins near the ground plane, any direction, time = a hit t (quirk Q1)
    o[:k] = np.stack([rng.uniform(-11, 11, k), rng.uniform(0.0, 0.6, k), rng.uniform(-11, 11, k)], axis=1)
    d[:k] = rng.normal(size=(k, 3))
    tm[:k] = rng.uniform(0.0, 30.0, k)
    rays = np.zeros((n, 8))
    rays[:, 0:3], rays[:, 3:6], rays[:, 6] = o, d, tm
    return rays


def nasty(rays, rng):
    r = rays.copy()
    n = r.shape[0]
    for i in range(n):
        k = i % 8
        if k == 0: r[i, 3 + rng.integers(0, 3)] = np.nan          # NaN direction component
        elif k == 1: r[i, 6] = np.nan                              # NaN time: the shutter fraction is NaN -> classic walk at once
        elif k == 2: r[i, rng.integers(0, 3)] = np.nan             # NaN origin component
        elif k == 3: r[i, 3 + rng.integers(0, 3)] = 0.0            # a zero direction component (1/0 = inf, aabb.rs:15)
        elif k == 4: r[i, 3 + rng.integers(0, 3)] = -0.0
        elif k == 5: r[i, 3:6] = 0.0                               # no direction at all: every slab is NaN or inf
        elif k == 6: r[i, 6] = np.inf                              # time = inf: MovingSphere::center is inf or NaN
        else: r[i, 3 + rng.integers(0, 3)] = np.inf
    return r


@pytest.mark.parametrize("build", ["best_axis", "reference", "sah"])
def test_pair_walk_equals_the_classic_walk_under_random_schedules(rt, build):
    rng = np.random.default_rng(7)
    worst_redo = 0
    for seed in (1, 2, 3):
        sc = rt.Scene.reference(0, build_seed=seed, aspect_ratio=1.5).set_bvh_build(build)
        rays = rays_for(seed, 6000, rng)
        for sched in (1, 2, 3):
            t, prim, rt_, rprim, flags = pair_walk(orc.B, sc, rays, stack_cap=12, seed=sched)
            assert not (flags & 2).any(), "an index left the per-lane stack or queue"
            assert np.array_equal(prim, rprim), (build, seed, sched, int((prim != rprim).sum()))
            hit = prim != 0xFFFFFFFF
            assert np.array_equal(t[hit].view(np.uint64), rt_[hit].view(np.uint64)), (build, seed, sched)
            assert hit.mean() > 0.5
        worst_redo = max(worst_redo, int((flags & 1).sum()))
    assert worst_redo == 0          # finite rays never need the hand-over


def test_pair_walk_hands_nan_rays_to_the_classic_walk(rt):
    """Non-finite rays: whatever the classic walk answers (mostly misses; NaN roots are accepted, sphere.rs:43-48), the pair walk path
    answers the same, through its two hand-overs, and never leaves its arrays."""
    rng = np.random.default_rng(11)
    sc = rt.Scene.reference(0, build_seed=1, aspect_ratio=1.5)
    rays = nasty(rays_for(1, 4000, rng), rng)
    redo_total = 0
    for sched in (1, 2):
        t, prim, rt_, rprim, flags = pair_walk(orc.B, sc, rays, stack_cap=12, seed=sched)
        assert not (flags & 2).any()
        assert np.array_equal(prim, rprim), int((prim != rprim).sum())
        hit = prim != 0xFFFFFFFF
        assert np.array_equal(t[hit].view(np.uint64), rt_[hit].view(np.uint64))   # NaN t included: same bits
        redo_total += int((flags & 1).sum())
    # both hand-overs were reached: the NaN-time rays (1 in 8) take the first; a NaN closest hit takes the second
    nan_time = np.isnan(rays[:, 6])
    assert ((flags & 1) != 0)[nan_time].all() and redo_total > int(nan_time.sum())
