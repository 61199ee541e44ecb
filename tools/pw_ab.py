"""Pair walk (rt_walk_pair.h, default for sphere scenes) against the one-entry-per-step walk on random_scene 1200x800: kernel Mpaths/s, frames compared.
  python3 tools/pw_ab.py [spp]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for sah in (False, True):
    sc = rt.Scene.reference(0, aspect_ratio=1.5)
    if sah:
        sc.set_bvh_build(True)
    ctx = rt.Context(sc, 0)
    ctx.render(1200, 800, 2)
    res = {}
    for name, kw in (("pair walk", {}), ("one entry per step", {"classic_walk": True})):
        best = 0
        for _ in range(3):
            img, st = ctx.render(1200, 800, spp, **kw)
            best = max(best, st["paths"] / st["kernel_ms"] / 1e3)
        res[name] = (best, img, st)
    same = np.array_equal(res["pair walk"][1], res["one entry per step"][1], equal_nan=True) and res["pair walk"][2]["segments"] == res["one entry per step"][2]["segments"]
    print(f"random_scene 1200x800x{spp} {'SAH tree' if sah else 'reference tree'} ({sc.info()['n_nodes']} nodes): pair walk {res['pair walk'][0]:7.1f} Mpaths/s (flags {res['pair walk'][2]['sorted']}) | "
          f"one entry per step {res['one entry per step'][0]:7.1f} | ratio {res['pair walk'][0] / res['one entry per step'][0]:.3f} | frames identical: {same}", flush=True)
    ctx.close()
