"""Opt-in BVH build by surface-area heuristic (rt1w_scene_set_bvh_build) on the big scenes: kernel-time Mpaths/s of the reference
build against SAH, SAH + near-far (result-preserving / everywhere), each also in f32; differing pixels against the reference build."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")

for name, arm, aspect, W, H, spp in (("C2 random_scene", 0, 1.5, 1200, 800, 64), ("C4 final_scene", 7, None, 800, 800, 64)):
    base = None
    for label, sah, order in (("reference build", False, 0), ("near-far", False, 1), ("SAH", True, 0), ("SAH + near-far", True, 1), ("SAH + near-far everywhere", True, 2)):
        sc = rt.Scene.reference(arm, aspect_ratio=aspect)
        if sah:
            sc.set_bvh_build(True)
        if order:
            sc.set_walk_order(order)
        ctx = rt.Context(sc, 0)
        ctx.render(W, H, 2)
        rates = {}
        for f32 in (False, True):
            best = 0
            for _ in range(2):
                g, s = ctx.render(W, H, spp, f32=f32)
                best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
            rates[f32] = (best, s["segments"] / s["paths"])
            if not f32:
                img = g.copy()
        if base is None:
            base = img
        diff = int((img != base).any(axis=2).sum())
        print(f"{name:18s} {label:28s} nodes {sc.info()['n_nodes']:5d} stack {sc.info()['stack_need']:2d} | f64 {rates[False][0]:7.1f} | f32 {rates[True][0]:7.1f} Mpaths/s | "
              f"seg/path {rates[False][1]:.3f} | pixels differing from the reference build {diff} of {W * H} | mean {np.nanmean(img):.5f}", flush=True)
        ctx.close()
