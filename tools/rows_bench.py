"""Whole-call wall time of the host-output entries on a 4K Cornell frame (C5 geometry, reduced spp):
one-shot f64 (rt1w_render), one-shot u8 (rt1w_render_u8), strips f64 / u8 (rt1w_render_rows)."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import orc
rt = orc.rt()
W, H = 3840, 2160
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sc = rt.Scene.reference(5, aspect_ratio=16.0 / 9.0)
ctx = rt.Context(sc, 0)
ctx.render(W, H, 1)
ctx.render_rows(W, H, 1)


def timed(fn):
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        out, st = fn()
        best = min(best, time.perf_counter() - t)
    return out, st, best


a, sa, ta = timed(lambda: ctx.render(W, H, spp))
b, sb, tb = timed(lambda: ctx.render_u8(W, H, spp))
c, sc_, tc = timed(lambda: ctx.render_rows(W, H, spp))
d, sd, td = timed(lambda: ctx.render_rows(W, H, spp, u8=True))
assert np.array_equal(a, c, equal_nan=True) and np.array_equal(b, d)
paths = W * H * spp / 1e6
for name, st, t in (("render f64      ", sa, ta), ("render_u8       ", sb, tb), ("render_rows f64 ", sc_, tc), ("render_rows u8  ", sd, td)):
    print(name, "wall %.1f ms  kernels %.1f ms  %.1f Mpaths/s whole call" % (t * 1e3, st["kernel_ms"], paths / t), flush=True)
