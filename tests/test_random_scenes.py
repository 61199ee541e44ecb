"""Randomly generated scene graphs, built call for call through the product's C ABI and in the literal oracle:
the flattener + iterative core (every valid kernel variant) must reproduce the recursive object-graph oracle."""
import numpy as np
import pytest

import orc
from dual import random_scene_pair

RTOL = 1e-12


def close(a, b):
    both_nan = np.isnan(a) & np.isnan(b)
    return bool((both_nan | (np.abs(a - b) <= RTOL * np.abs(a)) | (a == b)).all())


@pytest.mark.parametrize("seed", range(64))
def test_random_scene_graph_flat_core_vs_literal_oracle(rt, seed):
    prod, oracle = random_scene_pair(1000 + seed)
    info = prod.info()
    W, H, spp = 28, 20, 4
    a, sa = oracle.render(W, H, spp)
    ref = None
    for v in range(4):
        valid = (v in (1, 3)) or (v == 0 and info["variant"] == 0) or (v == 2 and not info["has_media"])
        if not valid:
            continue
        b, sb = orc.flat_render(prod, W, H, spp, variant=v)
        assert sb["segments"] == sa["segments"], (seed, v, info)
        assert close(a, b), (seed, v, info)
        if ref is None:
            ref = b
        assert np.array_equal(ref, b, equal_nan=True), (seed, v)
    assert sb["max_stack"] <= info["stack_need"]


@pytest.mark.gpu
def test_random_scene_graphs_on_the_gpu(rt, gpu_ctx_factory):
    """The same random graphs through the HIP kernels (default kernel, plain kernel, every valid forced variant):
    bit-identical to the CPU build of the core, equal segment counts, and within 1e-12 of the literal oracle."""
    for seed in range(16):
        prod, oracle = random_scene_pair(2000 + seed)
        info = prod.info()
        ctx = gpu_ctx_factory(prod)
        W, H, spp = 40, 28, 6
        g, sg = ctx.render(W, H, spp)
        b, sb = orc.flat_render(prod, W, H, spp, chunk=sg["chunk"])
        assert sg["segments"] == sb["segments"] and np.array_equal(g, b, equal_nan=True), (seed, info)
        a, sa = oracle.render(W, H, spp)
        assert sa["segments"] == sg["segments"] and close(a, g), (seed, info)
        u, su = ctx.render(W, H, spp, unsorted=True)
        assert np.array_equal(u, g, equal_nan=True), (seed, "unsorted")
        for v in (1, 3):
            f, sf = ctx.render(W, H, spp, variant=v)
            assert np.array_equal(f, g, equal_nan=True), (seed, v)
        ctx.close()
