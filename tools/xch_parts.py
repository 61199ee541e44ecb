"""(XCH_ARMS=1,2,3,4,5,6 selects the scene arms; default 5,6.)  C3 residue experiment: the exchange in RT_XCH_PARTS rounds (less LDS per workgroup) x waves per SIMD, on the specialised
Cornell and cornel_smoke kernels; every setting in a child process, frames hashed against the default build's."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, hashlib, importlib
sys.path.insert(0, %r)
rt = importlib.import_module("raytracing-1w_amd")
out = []
for arm in [int(a) for a in os.environ.get("XCH_ARMS", "5,6").split(",")]:
    ctx = rt.Context(rt.Scene.reference(arm), 0)
    info = ctx.specialise()
    g, s = ctx.render(96, 96, 16)
    h = hashlib.sha256(g.tobytes()).hexdigest()[:10]
    ctx.render(600, 600, 100)
    rates = []
    for _ in range(3):
        st = ctx.render(600, 600, 300)[1]
        rates.append(round(st["paths"] / st["kernel_ms"] / 1e3, 1))
    out.append("arm %%d hash %%s vgprs %%d grid %%d %%s" %% (arm, h, info["vgprs"], info["grid"], rates))
print("%%-58s %%s" %% (os.environ.get("RT1W_JIT_EXTRA_OPTS", "(none)"), " | ".join(out)), flush=True)
''' % ROOT
for opts in [None] + sys.argv[1:]:
    env = dict(os.environ, RT1W_KERNEL_CACHE="/tmp/rt1w_xch_cache")
    env.pop("RT1W_JIT_EXTRA_OPTS", None)
    if opts:
        env["RT1W_JIT_EXTRA_OPTS"] = opts
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, timeout=400)
    if r.returncode:
        print("FAILED:", opts, flush=True)
