"""Kernel-time throughput of the BASELINE configurations' geometry at reduced spp (Mpaths/s does not depend on spp):
C2 random_scene 1200x800, C3 Cornell 600x600, C4 final_scene 800x800, C5 Cornell 3840x2160, plus cornel_smoke 600x600 --
the default (exact f64) kernels, then the opt-in forms: wavefront, near-far walk order, SAH rebuild, f32, the reference's own stream."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")


def best_of(ctx, W, H, spp, n=2, **kw):
    best, st = 0, None
    for _ in range(n):
        g, s = ctx.render(W, H, spp, **kw)
        r = s["paths"] / s["kernel_ms"] / 1e3
        if r > best:
            best, st = r, s
    return best, st


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
    best, st = best_of(ctx, W, H, spp)
    kind = 'specialised' if st['sorted'] & 4 else ('sorted' if st['sorted'] & 1 else 'plain')
    line = (f"{name:18s} {W}x{H} nodes {sc.info()['n_nodes']:5d} seg/path {st['segments'] / st['paths']:.2f} | default V{st['variant']} {kind} "
            f"{best:7.1f}")
    if st["variant"] >= 2:
        dr = rt.Context(rt.Scene.reference(arm, aspect_ratio=aspect).set_bvh_build("reference"), 0)
        dr.render(W, H, 2)
        o, _ = best_of(dr, W, H, spp)
        line += f" | drawn axes (seed 1) {o:7.1f}"
        dr.close()
        w, _ = best_of(ctx, W, H, spp, wavefront=True)
        line += f" | wavefront {w:7.1f}"
        nf = rt.Context(rt.Scene.reference(arm, aspect_ratio=aspect).set_walk_order(1), 0)
        nf.render(W, H, 2)
        o, _ = best_of(nf, W, H, spp)
        line += f" | near-far {o:7.1f}"
        o32, _ = best_of(nf, W, H, spp, f32=True)
        line += f" | near-far+f32 {o32:7.1f}"
        nf.close()
        sah = rt.Context(rt.Scene.reference(arm, aspect_ratio=aspect).set_bvh_build(True).set_walk_order(1), 0)
        sah.render(W, H, 2)
        o, _ = best_of(sah, W, H, spp)
        line += f" | SAH+near-far {o:7.1f}"
        o32, _ = best_of(sah, W, H, spp, f32=True)
        line += f" | SAH+near-far+f32 {o32:7.1f}"
        sah.close()
    f32, s32 = best_of(ctx, W, H, spp, f32=True)
    line += f" | f32 {f32:7.1f}"
    ref, _ = best_of(ctx, W, H, min(spp, 100), reference_stream=True)
    line += f" | reference-stream {ref:7.1f}"
    print(line + "  Mpaths/s", flush=True)
    ctx.close()
