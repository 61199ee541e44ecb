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
    for v in (0, 1, 2, 3, 5):
        valid = (v in (1, 3)) or (v == 0 and info["variant"] == 0) or (v == 2 and not info["has_media"]) or \
                (v == 5 and not info["has_media"] and info["scope_depth"] == 0)
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
    n_wavefront = 0
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
        # the forced stack walk reads its nodes through the walk table, the most visited records from LDS (bit 10), general media
        # boundaries included; without the cache: the same bits
        assert bool(sf["sorted"] & 1024) == (info["stack_need"] <= 16), (seed, sf["sorted"], info)
        f2, sf2 = ctx.render(W, H, spp, variant=3, no_node_cache=True)
        assert not (sf2["sorted"] & 1024) and sf2["segments"] == sg["segments"] and np.array_equal(f2, g, equal_nan=True), (seed, "no node cache")
        # the wavefront form on the same graphs (forced stack-walk variant: wrappers, media, nested BVHs through the
        # vote-scheduled trace kernel with LDS-resident walk records, the shade kernel and the finish kernel)
        try:
            w, sw = ctx.render(W, H, spp, variant=3, wavefront=True)
            assert (sw["sorted"] & 8) and sw["segments"] == sg["segments"] and np.array_equal(w, g, equal_nan=True), (seed, "wavefront")
            n_wavefront += 1
        except rt.Rt1wError as e:                  # graphs whose moving spheres have different shutter intervals keep the megakernel
            assert e.code == rt.ERR_UNSUPPORTED and "shutter" in str(e), e
        # the reference-stream kernels against the CPU build of the same core with the same switch
        r, sr = ctx.render(W, H, spp, reference_stream=True)
        c, sc_ = orc.flat_render(prod, W, H, spp, chunk=spp, lib=orc.flat_ref_lib(), variant=1 if info["n_nodes"] <= 64 else 3)
        assert sr["segments"] == sc_["segments"] and np.array_equal(r, c, equal_nan=True), (seed, "reference stream")
        # f32: renders, finite where the f64 frame is, and close in the mean
        h, sh = ctx.render(W, H, spp, f32=True)
        assert (sh["sorted"] & 32) and abs(np.nanmean(h) - np.nanmean(g)) <= 0.05 * abs(np.nanmean(g)) + 1e-3, (seed, "f32")
        ctx.close()
    assert n_wavefront >= 8
