"""Wavefront form A/B on the big scenes: megakernel vs wavefront with the vote-scheduled trace kernel (round 2) vs wavefront with the plain
trace kernel (round 3), and the number of wavefront bounces before the finish kernel.  Frames must be bit-identical.
  python3 tools/wf_ab.py [spp]"""
import importlib, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for arm, W, H in ((7, 800, 800), (0, 1200, 800)):
    for sah in (False, True):
        sc = rt.Scene.reference(arm, aspect_ratio=W / H)
        if sah:
            sc.set_bvh_build(True).set_walk_order(1)
        ctx = rt.Context(sc, 0)
        ctx.render(W, H, 2)
        ref, st = None, None
        for _ in range(2):
            ref, st = ctx.render(W, H, spp)
        line = f"arm {arm} {'SAH+near-far' if sah else 'reference tree'} {W}x{H}x{spp}: megakernel V{st['variant']} {st['paths'] / st['kernel_ms'] / 1e3:7.1f} Mpaths/s"
        for env, label in (({"RT1W_WF_TRACE": "vote"}, "wf vote"), ({"RT1W_WF_BOUNCES": "6"}, "wf plain b6"), ({"RT1W_WF_BOUNCES": "12"}, "b12"),
                           ({"RT1W_WF_BOUNCES": "20"}, "b20"), ({"RT1W_WF_BOUNCES": "50"}, "b50")):
            for k in ("RT1W_WF_TRACE", "RT1W_WF_BOUNCES"):
                os.environ.pop(k, None)
            os.environ.update(env)
            best = 0
            for _ in range(2):
                img, sw = ctx.render(W, H, spp, wavefront=True)
                best = max(best, sw["paths"] / sw["kernel_ms"] / 1e3)
            same = np.array_equal(img, ref, equal_nan=True) and sw["segments"] == st["segments"]
            line += f" | {label} {best:7.1f}{'' if same else ' DIFFERENT FRAME'}"
        print(line, flush=True)
        ctx.close()
