"""Opt-in walk order against the reference order on the big scenes (megakernel): frames, then throughput."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import numpy as np
rt = importlib.import_module("raytracing-1w_amd")
for arm, aspect, (W, H, spp), modes in ((0, 1.5, (1200, 800, 24), (0, 1, 2)), (7, None, (800, 800, 16), (0, 1))):
    ref = None
    for mode in modes:
        sc = rt.Scene.reference(arm, aspect_ratio=aspect).set_walk_order(mode)
        ctx = rt.Context(sc, 0)
        best = 0
        for _ in range(2):
            g, s = ctx.render(W, H, spp)
            best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
        if ref is None:
            ref = g
        print("arm", arm, "walk order", mode, round(best, 1), "Mpaths/s", "differing pixels vs reference order:", int((g != ref).any(axis=2).sum()), "of", W * H, flush=True)
        ctx.close()
