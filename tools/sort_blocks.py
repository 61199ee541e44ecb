"""Compare sort-domain sizes; first a small bit-exactness check against the unsorted kernel (guards against a bad build)."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import orc
rt = orc.rt()
sc = rt.Scene.reference(5)
ctx = rt.Context(sc, 0)
a, sa = ctx.render(64, 64, 4, unsorted=True)
b, sb = ctx.render(64, 64, 4)
ok = np.array_equal(a, b) and sa["segments"] == sb["segments"]
print(rt.LIB_PATH.split('/')[-1], "block", sb["block"], "bit-exact", ok, flush=True)
if not ok:
    sys.exit(3)
best = 0
for _ in range(3):
    g, s = ctx.render(600, 600, 200)
    best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
print("  Mpaths/s", round(best, 1), "grid", s["grid"], flush=True)
