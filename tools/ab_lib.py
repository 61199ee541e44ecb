"""A/B two builds of the library on Cornell: hash of a small frame (bit-exactness guard) + kernel-time throughput.
usage: python tools/ab_lib.py libA.so libB.so [...]   (paths relative to raytracing-1w_amd/)"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, hashlib
sys.path.insert(0, os.path.join(%r, "tests"))
import orc
rt = orc.rt()
sc = rt.Scene.reference(5)
ctx = rt.Context(sc, 0)
g, s = ctx.render(96, 96, 16)
h = hashlib.sha256(g.tobytes()).hexdigest()[:16]
best = []
for _ in range(4):
    g2, s2 = ctx.render(600, 600, 200)
    best.append(round(s2["paths"] / s2["kernel_ms"] / 1e3, 1))
print(os.environ.get("RT1W_LIB"), "hash", h, "segments", s["segments"], "Mpaths/s", best, flush=True)
''' % ROOT
for lib in sys.argv[1:]:
    env = dict(os.environ, RT1W_LIB=os.path.join(ROOT, "raytracing-1w_amd", lib))
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=True, timeout=300)
