"""Compiler-flag experiments on the specialised Cornell kernel: each RT1W_JIT_EXTRA_OPTS setting in a child process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, hashlib
sys.path.insert(0, os.path.join(%r, "tests"))
import orc
rt = orc.rt()
ctx = rt.Context(rt.Scene.reference(5), 0)
info = ctx.specialise()
g, s = ctx.render(96, 96, 16)
h = hashlib.sha256(g.tobytes()).hexdigest()[:12]
best = max(ctx.render(600, 600, 200)[1]["paths"] / ctx.render(600, 600, 200)[1]["kernel_ms"] / 1e3 for _ in range(2))
rates = []
for _ in range(3):
    st = ctx.render(600, 600, 200)[1]
    rates.append(round(st["paths"] / st["kernel_ms"] / 1e3, 1))
print("%%-60s hash %%s vgprs %%d grid %%d compile %%.1fs Mpaths/s %%s" %% (os.environ.get("RT1W_JIT_EXTRA_OPTS", "(none)"), h, info["vgprs"], info["grid"], info["compile_ms"] / 1e3, rates), flush=True)
''' % ROOT
for opts in [None] + sys.argv[1:]:
    env = dict(os.environ, RT1W_KERNEL_CACHE="/tmp/rt1w_opts_cache")
    env.pop("RT1W_JIT_EXTRA_OPTS", None)
    if opts:
        env["RT1W_JIT_EXTRA_OPTS"] = opts
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, timeout=300)
    if r.returncode:
        print("FAILED:", opts, flush=True)
