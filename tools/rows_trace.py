"""One strip-wise 4K render, for `rocprofv3 --kernel-trace`: do the strips' kernels overlap?"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import orc
rt = orc.rt()
sc = rt.Scene.reference(5, aspect_ratio=16.0 / 9.0)
ctx = rt.Context(sc, 0)
ctx.render_rows(3840, 2160, 1)
out, st = ctx.render_rows(3840, 2160, int(sys.argv[1]) if len(sys.argv) > 1 else 128, u8=True)
print(st, flush=True)
