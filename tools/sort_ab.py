"""A/B: workgroup-sorted kernel vs unsorted (bit-exact expected), Cornell."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import orc
rt = orc.rt()
sc = rt.Scene.reference(5)
ctx = rt.Context(sc, 0)
a, sa = ctx.render(96, 96, 16, unsorted=True)
b, sb = ctx.render(96, 96, 16)
print("sorted flag", sb["sorted"], "bit-exact", np.array_equal(a, b), "segments", sa["segments"], sb["segments"], flush=True)
for tile, spp in (((0, 0, 37, 23), 3), ((5, 7, 100, 9), 40)):
    a, sa = ctx.render(200, 200, spp, tile=tile, unsorted=True)
    b, sb = ctx.render(200, 200, spp, tile=tile)
    print("tile", tile, "bit-exact", np.array_equal(a, b), sa["segments"] == sb["segments"], flush=True)
for uns in (True, False):
    best = 0
    for _ in range(3):
        g, s = ctx.render(600, 600, 200, unsorted=uns)
        best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
    print("unsorted" if uns else "sorted  ", "Mpaths/s", round(best, 1), "grid", s["grid"], flush=True)

for arm, (W, H, spp) in {6: (600, 600, 50), 0: (600, 400, 50), 7: (400, 400, 32)}.items():
    sc2 = rt.Scene.reference(arm, aspect_ratio=1.5 if arm == 0 else None)
    c2 = rt.Context(sc2, 0)
    a, sa = c2.render(W, H, 4, unsorted=True)
    b, sb = c2.render(W, H, 4)
    res = {}
    for uns in (True, False):
        best = 0
        for _ in range(2):
            g, s = c2.render(W, H, spp, unsorted=uns)
            best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
        res[uns] = round(best, 1)
    print("arm", arm, "variant", sb["variant"], "bit-exact", np.array_equal(a, b, equal_nan=True), "unsorted", res[True], "sorted", res[False], flush=True)
