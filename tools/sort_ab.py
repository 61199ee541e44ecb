"""A/B: workgroup-sorted kernel vs unsorted (bit-exact expected), Cornell."""
import sys
import numpy as np
sys.path.insert(0, 'tests')
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
