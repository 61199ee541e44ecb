"""Plain sliced stack-walk kernel against the same kernel with the finished paths reordered across the workgroup at the end of every
slice (the default for sphere-media scenes; the flag RT1W_UNSORTED selects the plain kernel; rt_kernel_plain.h: rt_render_ss_body): kernel Mpaths/s, frames compared bit for bit.
  python3 tools/ss_ab.py [arm] [W] [H] [spp] [--sah]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
arm, W, H, spp = (int(x) for x in (args + ["7", "800", "800", "100"][len(args):])[:4])
sc = rt.Scene.reference(arm, aspect_ratio=W / H)
if "--sah" in sys.argv:
    sc.set_bvh_build(True)
    sc.set_walk_order(1)
ctx = rt.Context(sc, 0)
ctx.render(W, H, 4)
ref = None
for rep in range(3):
    for mode in ("plain", "slice-sorted"):
        img, st = ctx.render(W, H, spp, unsorted=(mode == "plain"), classic_walk="--classic" in sys.argv)
        if ref is None:
            ref = img
        print(f"{mode:13s} V{st['variant']} flags {st['sorted']:4d}: {W * H * spp / st['kernel_ms'] / 1e3:8.1f} Mpaths/s kernel {st['kernel_ms']:8.2f} ms  segments {st['segments']}  "
              f"frame == plain: {bool(np.array_equal(ref, img, equal_nan=True))}", flush=True)
ctx.close()
