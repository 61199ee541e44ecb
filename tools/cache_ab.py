"""A/B: LDS node cache for the stack variants (random_scene) -- bit-exactness first, then throughput."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, orc
rt = orc.rt()
sc = rt.Scene.reference(0, aspect_ratio=1.5)
ctx = rt.Context(sc, 0)
a, sa = ctx.render(96, 64, 8)
b, sb = ctx.render(96, 64, 8, lds_nodes=True)
ok = np.array_equal(a, b) and sa["segments"] == sb["segments"]
print("nodes", sc.info()["n_nodes"], "cache flag", sb["sorted"], "bit-exact", ok, flush=True)
if not ok: sys.exit(3)
for no in (True, False):
    best = 0
    for _ in range(3):
        g, s = ctx.render(1200, 800, 50, lds_nodes=not no)
        best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
    print("global nodes" if no else "LDS nodes   ", "Mpaths/s", round(best, 1), "grid", s["grid"], flush=True)
