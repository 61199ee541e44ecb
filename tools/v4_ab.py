"""final_scene 800x800: the default kernel (V3), the near-far order (V4) on the reference's tree and on the SAH tree, under several builds
of the library: python3 tools/v4_ab.py lib1.so lib2.so ...  (kernel Mpaths/s at 64 spp, best of 3)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, importlib
sys.path.insert(0, %r)
rt = importlib.import_module("raytracing-1w_amd")
out = []
for name, sah, nf in (("V3", False, False), ("near-far", False, True), ("SAH", True, False), ("SAH + near-far", True, True)):
    sc = rt.Scene.reference(7)
    if sah: sc.set_bvh_build(True)
    if nf: sc.set_walk_order(True)
    ctx = rt.Context(sc, 0)
    ctx.render(96, 64, 2)
    best = 0
    for _ in range(3):
        g, s = ctx.render(800, 800, 64)
        best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
    out.append("%%s (V%%d, flags %%d) %%.1f" %% (name, s["variant"], s["sorted"], best))
    ctx.close()
print(os.path.basename(os.environ.get("RT1W_LIB", "default")), " | ".join(out), flush=True)
''' % ROOT
for lib in sys.argv[1:]:
    env = dict(os.environ, RT1W_LIB=os.path.join(ROOT, "raytracing-1w_amd", lib))
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False, timeout=600)
