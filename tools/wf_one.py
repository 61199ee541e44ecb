"""One wavefront (or megakernel) render of a big scene, for profiling: python tools/wf_one.py <arm> <W> <H> <spp> [mega]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
rt = importlib.import_module("raytracing-1w_amd")
arm, W, H, spp = (int(x) for x in sys.argv[1:5])
mega = len(sys.argv) > 5 and sys.argv[5] == "mega"
sc = rt.Scene.reference(arm, aspect_ratio=W / H)
ctx = rt.Context(sc, 0)
for _ in range(2):
    g, s = ctx.render(W, H, spp, wavefront=not mega)
    print("arm", arm, "mega" if mega else "wavefront", round(s["paths"] / s["kernel_ms"] / 1e3, 1), "Mpaths/s", s["kernel_ms"], "ms", "segments/path", s["segments"] / s["paths"], flush=True)
