"""Big-scene throughput under several builds of the library: python tools/lib_ab.py lib1.so lib2.so ... (megakernel, default path)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, importlib, hashlib
sys.path.insert(0, %r)
rt = importlib.import_module("raytracing-1w_amd")
out = []
SPP = int(os.environ.get("AB_SPP", "48"))
for arm, aspect, (W, H, spp) in ((0, 1.5, (1200, 800, SPP)), (7, None, (800, 800, SPP))):
    ctx = rt.Context(rt.Scene.reference(arm, aspect_ratio=aspect), 0)
    g, s = ctx.render(96, 64, 4)
    h = hashlib.sha256(g.tobytes()).hexdigest()[:10]
    best = 0
    for _ in range(3):
        g, s = ctx.render(W, H, spp)
        best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
    out.append("arm %%d hash %%s %%.1f Mpaths/s" %% (arm, h, best))
print(os.environ.get("RT1W_LIB", "default"), " | ".join(out), flush=True)
''' % ROOT
for lib in sys.argv[1:]:
    env = dict(os.environ, RT1W_LIB=os.path.join(ROOT, "raytracing-1w_amd", lib))
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False, timeout=300)
