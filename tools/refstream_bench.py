"""RT1W_RNG_REFERENCE (the reference's own ChaCha12 stream per pixel): the reordering kernel (default since round 4) against the plain kernel
(RT1W_UNSORTED: one lane per pixel for all its samples), Cornell at 600x600 and at 3840x2160; kernel Mpaths/s, frames compared.
  python3 tools/refstream_bench.py [spp]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for W, H, aspect in ((600, 600, None), (3840, 2160, 16.0 / 9.0)):
    ctx = rt.Context(rt.Scene.reference(5, aspect_ratio=aspect), 0)
    frames = {}
    for label, kw in (("reordering", {}), ("plain", dict(unsorted=True))):
        ctx.render(W, H, 2, reference_stream=True, **kw)
        best = 0.0
        for _ in range(2):
            img, st = ctx.render(W, H, spp, reference_stream=True, **kw)
            best = max(best, W * H * spp / st["kernel_ms"] / 1e3)
        frames[label] = img
        print(f"Cornell {W}x{H}x{spp} reference stream, {label:10s} (flags {st['sorted']:3d}): {best:8.1f} Mpaths/s", flush=True)
    print("   frames equal:", bool(np.array_equal(frames["reordering"], frames["plain"])), flush=True)
    ctx.close()
