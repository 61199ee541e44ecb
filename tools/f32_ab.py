"""f32 mode (RT1W_PRECISION_F32) against the f64 kernels on Cornell and the big scenes: throughput (kernel time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import numpy as np
rt = importlib.import_module("raytracing-1w_amd")
for arm, aspect, (W, H, spp) in ((5, None, (600, 600, 200)), (0, 1.5, (1200, 800, 24)), (7, None, (800, 800, 16))):
    ctx = rt.Context(rt.Scene.reference(arm, aspect_ratio=aspect), 0)
    for name, kw in (("f64 default", {}), ("f64 generic", dict(generic=True)), ("f32", dict(f32=True))):
        best = 0
        for _ in range(3):
            g, s = ctx.render(W, H, spp, **kw)
            best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
        print("arm", arm, name, round(best, 1), "Mpaths/s", "variant", s["variant"], "flags", s["sorted"], "mean", float(np.nanmean(g)), flush=True)
    ctx.close()
