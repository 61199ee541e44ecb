"""Kernel-time throughput of the BASELINE configurations' geometry at reduced spp (Mpaths/s does not depend on spp):
C2 random_scene 1200x800, C3 Cornell 600x600, C4 final_scene 800x800, C5 Cornell 3840x2160, plus cornel_smoke 600x600."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import orc
rt = orc.rt()
for name, arm, aspect, W, H, spp in (("C2 random_scene", 0, 1.5, 1200, 800, 100), ("C3 cornel_box", 5, None, 600, 600, 400),
                                     ("C4 final_scene", 7, None, 800, 800, 100), ("C5 cornel_box 4K", 5, 16.0 / 9.0, 3840, 2160, 64),
                                     ("cornel_smoke", 6, None, 600, 600, 400)):
    sc = rt.Scene.reference(arm, aspect_ratio=aspect)
    ctx = rt.Context(sc, 0)
    try:
        ctx.specialise()
    except rt.Rt1wError:
        pass
    ctx.render(W, H, 2)
    best, st = 0, None
    for _ in range(2):
        g, s = ctx.render(W, H, spp)
        if s["paths"] / s["kernel_ms"] / 1e3 > best:
            best, st = s["paths"] / s["kernel_ms"] / 1e3, s
    print(f"{name:18s} {W}x{H}  {best:8.1f} Mpaths/s  segments/path {st['segments'] / st['paths']:.2f}  variant V{st['variant']} "
          f"{'specialised' if st['sorted'] & 4 else ('sorted' if st['sorted'] & 1 else 'plain')}  nodes {sc.info()['n_nodes']}", flush=True)
