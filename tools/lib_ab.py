"""Big-scene throughput under several builds of the library: python tools/lib_ab.py lib1.so lib2.so ... (default kernels; kernel Mpaths/s,
best of 3; AB_SPP = samples per pixel, default 48; AB_TREES=1 adds the SAH + near-far trees).  A small frame's hash first: every build must
render the same bits."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, importlib, hashlib
sys.path.insert(0, %r)
rt = importlib.import_module("raytracing-1w_amd")
SPP = int(os.environ.get("AB_SPP", "48"))
cfgs = [(0, 1.5, (1200, 800, SPP), "best_axis", 0), (7, None, (800, 800, SPP), "best_axis", 0)]
if os.environ.get("AB_TREES"):
    cfgs += [(0, 1.5, (1200, 800, SPP), "sah", 1), (7, None, (800, 800, SPP), "sah", 1)]
out = []
for arm, aspect, (W, H, spp), build, nf in cfgs:
    sc = rt.Scene.reference(arm, aspect_ratio=aspect).set_bvh_build(build)
    if nf:
        sc.set_walk_order(1)
    ctx = rt.Context(sc, 0)
    g, s = ctx.render(96, 64, 4)
    h = hashlib.sha256(g.tobytes()).hexdigest()[:8]
    best = 0
    for _ in range(3):
        g, s = ctx.render(W, H, spp)
        best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
    out.append("arm %%d %%s%%s %%s %%.1f" %% (arm, build, "+nf" if nf else "", h, best))
    ctx.close()
print(os.path.basename(os.environ.get("RT1W_LIB", "default")), " | ".join(out), flush=True)
''' % ROOT
for lib in sys.argv[1:]:
    env = dict(os.environ, RT1W_LIB=os.path.join(ROOT, "raytracing-1w_amd", lib))
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False, timeout=600)
