"""Wavefront form (RT1W_WAVEFRONT) against the persistent megakernel on the big scenes: bits first, then throughput."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import orc
rt = orc.rt()
small = "--small" in sys.argv
for arm, aspect, (W, H, spp) in ((0, 1.5, (96, 64, 6) if small else (1200, 800, 24)), (7, None, (64, 64, 6) if small else (800, 800, 16))):
    sc = rt.Scene.reference(arm, aspect_ratio=aspect)
    ctx = rt.Context(sc, 0)
    a, sa = ctx.render(W, H, spp)
    b, sb = ctx.render(W, H, spp, wavefront=True)
    print("arm", arm, "bit-exact", np.array_equal(a, b, equal_nan=True), "segments", sa["segments"], sb["segments"], "sorted flags", sa["sorted"], sb["sorted"], flush=True)
    if not small:
        for wf in (False, True):
            best = 0
            for _ in range(2):
                g, s = ctx.render(W, H, spp, wavefront=wf)
                best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
            print("   ", "wavefront" if wf else "megakernel", round(best, 1), "Mpaths/s", "grid", s["grid"], flush=True)
