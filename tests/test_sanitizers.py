"""AddressSanitizer + UBSan over the kernel core (rt_core.h) in its CPU build -- GPU sanitizers are not available
on the pool, so memory safety of the traversal/shading code (stack bounds, node/material/texture indexing, image
texel addressing, RNG buffer discipline via RT_RNG_CHECK) is checked here, on every scene arm and kernel variant."""
import os
import subprocess
import sys
import textwrap

import pytest

import orc

SCRIPT = textwrap.dedent("""
    import ctypes, os, sys
    sys.path.insert(0, os.path.join({root!r}, 'tests'))
    import orc, numpy as np
    orc.B = ctypes.CDLL(os.path.join({root!r}, 'oracle', 'liborc_flat_asan.so'))
    import importlib; importlib.reload  # keep linters quiet
    B = orc.B
    B.orcflat_render.restype = ctypes.c_int
    rt = orc.rt()
    for arm, (W, H, spp), variants in ((5, (24, 24, 4), (0, 1, 3)), (0, (24, 16, 2), (1, 2, 3, 5)), (6, (20, 20, 4), (1, 3)),
                                       (7, (16, 16, 4), (1, 3)), (3, (16, 9, 2), (1, 3)), (2, (16, 9, 2), (1, 3))):
        sc = rt.Scene.reference(arm, build_seed=1)
        ref = None
        for v in variants:
            img, st = orc.flat_render(sc, W, H, spp, variant=v, threads=1)
            ref = img if ref is None else ref
            assert np.array_equal(img, ref, equal_nan=True)
    # the pair walk's lane functions (rt_walk_pair.h) with their bound-checked host arrays, finite and non-finite rays
    import test_pair_walk_host as pw
    rng = np.random.default_rng(3)
    sc0 = rt.Scene.reference(0, build_seed=1, aspect_ratio=1.5)
    for rays in (pw.rays_for(1, 1500, rng), pw.nasty(pw.rays_for(1, 1500, rng), rng)):
        t, prim, rt_, rprim, flags = pw.pair_walk(B, sc0, rays, stack_cap=12, seed=5)
        assert not (flags & 2).any() and np.array_equal(prim, rprim)
    # the walk over the walk table (rt_walk_table.h) with an arbitrary ranking, finite and non-finite rays
    import test_walk_table_host as wt
    for arm, variant in ((7, 103), (6, 3)):
        scw = rt.Scene.reference(arm, build_seed=1)
        rays = wt.scene_rays(scw, 1500, rng)
        visits = rng.integers(0, 50, scw.info()['n_nodes']).astype(np.uint32)
        for rr in (rays, wt.nasty(rays, rng)):
            t, prim, scope, flags, _ = wt.walk_table_check(B, scw, rr, visits=visits, nc=64, variant=variant)
            assert not (flags & 3).any()
    print('SANITIZED-OK')
""")


def test_kernel_core_under_asan_ubsan(rt):
    so = os.path.join(orc.ORACLE_DIR, "liborc_flat_asan.so")
    subprocess.check_call(["make", "-C", orc.ORACLE_DIR, "-s", "liborc_flat_asan.so"])
    assert os.path.exists(so)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the ctypes prototypes are set on orc.B at import; re-point them after swapping the library
    script = SCRIPT.format(root=orc.ROOT).replace("B.orcflat_render.restype = ctypes.c_int",
        "B.orcflat_render.restype = ctypes.c_int\n"
        "B.orcflat_render.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(orc.Frame), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]")
    p = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "SANITIZED-OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
