"""One Cornell (or arm argv[1]) render with the scene-specialised kernel, for rocprofv3 passes: python tools/c3_one.py [arm] [spp]
(RT1W_JIT_EXTRA_OPTS / RT1W_KERNEL_CACHE pass through: kernel variants are compiled on first use)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
arm = int(sys.argv[1]) if len(sys.argv) > 1 else 5
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = rt.Context(rt.Scene.reference(arm), 0)
info = ctx.specialise()
for _ in range(2):
    g, s = ctx.render(600, 600, spp)
print(os.environ.get("RT1W_JIT_EXTRA_OPTS", "(none)"), "arm", arm, "vgprs", info["vgprs"], round(s["paths"] / s["kernel_ms"] / 1e3, 1), "Mpaths/s", flush=True)
