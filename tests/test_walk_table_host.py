"""The walk table (raytracing-1w_amd/csrc/rt_walk_table.h: the records the big scenes' stack-walk kernels read, the most visited ones from
LDS) on the CPU: the table is built by the same host code the context runs, and the walk's text (rt_core.h with RtWalkNodes) is compiled by
g++ (oracle/oracle_flat.cpp: orcflat_walk_table_check) and run ray by ray against the walk over the node array.  What must hold, whatever
the ranking of the nodes and wherever the cache ends: the same closest hit, primitive and scope bit for bit, the random stream left in the
same state (a ConstantMedium draws inside the walk), no deeper stack.  Rankings tried: the box-area estimate, the visit order of a real
walk, random counts (= an arbitrary permutation of the walk ids), all-zero counts (nothing but the root is ranked).  Scenes: every arm with
a stack-walk variant, including wrappers two deep, FlipFace, both media of final_scene and a medium bounded by a BOX (cornel_smoke: the
general boundary walk through the table).  Also run under ASan + UBSan (test_sanitizers.py)."""
import ctypes as C

import numpy as np
import pytest

import orc

_P = C.c_void_p
NONE = 0xFFFFFFFF


def walk_table_check(lib, scene, rays, visits=None, nc=256, variant=3):
    nodes = scene.flat(0)
    root = int(np.frombuffer(scene.flat(6).tobytes()[-8:-4], dtype=np.uint32)[0])
    n_nodes = scene.info()["n_nodes"]
    n = rays.shape[0]
    r = np.ascontiguousarray(rays, dtype=np.float64)
    t = np.empty(n)
    prim, scope, flags = np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.uint32)
    id_of = np.empty(n_nodes, np.uint32)
    v = None if visits is None else np.ascontiguousarray(visits, dtype=np.uint32)
    lib.orcflat_walk_table_check.restype = C.c_int
    lib.orcflat_walk_table_check.argtypes = [_P, C.c_uint32, C.c_uint32, _P, C.c_uint32, C.c_int, _P, C.c_uint64, _P, _P, _P, _P, _P]
    rc = lib.orcflat_walk_table_check(nodes.ctypes.data_as(_P), n_nodes, root, None if v is None else v.ctypes.data_as(_P), nc, variant,
                                      r.ctypes.data_as(_P), n, t.ctypes.data_as(_P), prim.ctypes.data_as(_P), scope.ctypes.data_as(_P),
                                      flags.ctypes.data_as(_P), id_of.ctypes.data_as(_P))
    assert rc == 0, "the walk table could not be built"
    return t, prim, scope, flags, id_of


def scene_rays(scene, n, rng):
    """camera rays of the scene's own camera plus rays from inside the scene's bounds in every direction (bounces), times in and
    outside the shutter"""
    cam = np.frombuffer(scene.flat(6).tobytes(), dtype=np.float64)
    origin, llc, hor, ver = cam[0:3], cam[3:6], cam[6:9], cam[9:12]
    nodes = np.frombuffer(scene.flat(0).tobytes(), dtype=np.uint8).reshape(-1, 96)
    root = int(np.frombuffer(scene.flat(6).tobytes()[-8:-4], dtype=np.uint32)[0])
    box = nodes[root, 8:56].copy().view(np.float64)
    lo, hi = np.maximum(box[0:3], -1500.0), np.minimum(box[3:6], 1500.0)
    rays = np.zeros((n, 8))
    k = n // 2
    u, v = rng.uniform(0, 1, k), rng.uniform(0, 1, k)
    rays[:k, 0:3] = origin
    rays[:k, 3:6] = llc + u[:, None] * hor + v[:, None] * ver - origin
    rays[k:, 0:3] = rng.uniform(lo, hi, (n - k, 3))
    rays[k:, 3:6] = rng.normal(size=(n - k, 3))
    rays[:, 6] = rng.uniform(0.0, 1.0, n)
    rays[::7, 6] = rng.uniform(0.0, 30.0, rays[::7].shape[0])  # a hit t carried as the time (quirk Q1)
    return rays


def nasty(rays, rng):
    r = rays.copy()
    for i in range(r.shape[0]):
        k = i % 6
        if k == 0: r[i, 3 + rng.integers(0, 3)] = np.nan
        elif k == 1: r[i, rng.integers(0, 3)] = np.nan
        elif k == 2: r[i, 3 + rng.integers(0, 3)] = 0.0
        elif k == 3: r[i, 3:6] = 0.0
        elif k == 4: r[i, 6] = np.nan
        else: r[i, 3 + rng.integers(0, 3)] = np.inf
    return r


ARMS = [(7, 103), (7, 3), (6, 3), (0, 3), (5, 3), (3, 3)]  # final_scene (sphere-media build and the general one), cornel_smoke (box media), random_scene, cornel_box, earth


@pytest.mark.parametrize("arm,variant", ARMS)
def test_the_table_walk_equals_the_node_array_walk_whatever_the_ranking(rt, arm, variant):
    rng = np.random.default_rng(100 + arm)
    sc = rt.Scene.reference(arm, build_seed=1)
    n_nodes = sc.info()["n_nodes"]
    rays = scene_rays(sc, 4000 if n_nodes > 1000 else 2500, rng)
    rankings = [None, np.zeros(n_nodes, np.uint32), rng.integers(0, 1000, n_nodes).astype(np.uint32),
                (np.arange(n_nodes, dtype=np.uint32)[::-1] + 1)]
    base = None
    for rk_i, visits in enumerate(rankings):
        for nc in (256, 1, 37, 100000):
            t, prim, scope, flags, id_of = walk_table_check(orc.B, sc, rays, visits=visits, nc=nc, variant=variant)
            assert not (flags & 1).any(), (arm, rk_i, nc, int((flags & 1).sum()))
            assert not (flags & 2).any(), (arm, rk_i, nc)
            assert sorted(id_of.tolist()) == list(range(n_nodes)), "walk ids are a permutation of the nodes"
            if base is None:
                base = (t.copy(), prim.copy(), scope.copy())
                assert (prim != NONE).mean() > 0.3
            else:
                hit = prim != NONE
                assert np.array_equal(prim, base[1]) and np.array_equal(scope, base[2]) and np.array_equal(t[hit].view(np.uint64), base[0][hit].view(np.uint64))
    # non-finite rays: the same answers as the node array's walk too
    t, prim, scope, flags, _ = walk_table_check(orc.B, sc, nasty(rays[:1200], rng), visits=rankings[2], nc=64, variant=variant)
    assert not (flags & 3).any()


def test_ranking_by_visits_puts_the_visited_nodes_first(rt):
    sc = rt.Scene.reference(7, build_seed=1)
    n_nodes = sc.info()["n_nodes"]
    rng = np.random.default_rng(5)
    visits = np.zeros(n_nodes, np.uint32)
    hot = rng.choice(n_nodes, 300, replace=False)
    visits[hot] = rng.integers(1, 10_000, 300)
    rays = scene_rays(sc, 200, rng)
    _, _, _, _, id_of = walk_table_check(orc.B, sc, rays, visits=visits, nc=256, variant=103)
    root = int(np.frombuffer(sc.flat(6).tobytes()[-8:-4], dtype=np.uint32)[0])
    assert id_of[root] == 0
    nodes = np.frombuffer(sc.flat(0).tobytes(), dtype=np.uint8).reshape(-1, 96)
    kind = nodes[:, 0:4].copy().view(np.uint32)[:, 0] & 0xFF
    boundary = np.zeros(n_nodes, bool)
    boundary[1:] = (kind[:-1] == 10) & (kind[1:] == 2)          # a bare Sphere right behind its ConstantMedium: never fetched through the table
    ranked = np.flatnonzero((visits > 0) & ~boundary & (np.arange(n_nodes) != root))
    order = ranked[np.argsort(-visits[ranked].astype(np.int64), kind="stable")]
    first = order[:255]
    assert np.array_equal(np.sort(id_of[first]), np.arange(1, 1 + first.size)), "the 255 most visited nodes follow the root"
    assert np.array_equal(id_of[first], np.arange(1, 1 + first.size)), "in the order of their counts"
    rest = np.setdiff1d(np.arange(n_nodes), np.concatenate([[root], first]))
    assert np.all(np.diff(id_of[rest]) > 0), "everything else keeps its pre-order"
