"""Per-phase cycle shares of the render kernel (diagnostic build librt1w_stamps.so)."""
import os, sys
os.environ["RT1W_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracing-1w_amd", "librt1w_stamps.so")
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import orc
rt = orc.rt()
names = ["-", "regen", "traverse", "hit record", "shade lambert", "shade other+tail", "loop top", "sort+exchange+barriers", "sweep setup", "sweep pop+hdr", "sweep idle step", "sweep bvh step", "sweep prim step", "sweep scope step", "-", "-"]
for arm, (W, H, spp) in ((5, (600, 600, 100)),) + tuple((int(a), (400, 400, 32)) for a in sys.argv[1:]):
    sc = rt.Scene.reference(arm)
    ctx = rt.Context(sc, 0)
    ctx.render(W, H, 2, generic=True)
    ctx.debug_stamps(True)
    g, s = ctx.render(W, H, spp, generic=True)   # the stamps are in the library's own (generic) kernels
    rc, st = ctx.debug_stamps(True)
    tot = sum(st)
    print(f"arm {arm} variant {s['variant']} kernel_ms {s['kernel_ms']:.1f} segments {s['segments']} stamps_valid {rc}")
    for n, v in zip(names, st):
        if v: print(f"  {n:14s} {100.0*v/tot:5.1f}%   cycles/segment/wave {v/(s['segments']/64):8.1f}")
