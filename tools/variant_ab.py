"""Forced kernel variants on one scene: python tools/variant_ab.py <arm> <W> <H> <spp> <variant> [<variant> ...] (kernel-time Mpaths/s, frame hash)"""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
arm, W, H, spp = (int(a) for a in sys.argv[1:5])
for sah in (False, True):
    sc = rt.Scene.reference(arm, aspect_ratio=W / H).set_bvh_build(sah)
    ctx = rt.Context(sc, 0)
    ctx.render(W, H, 2)
    for v in [int(a) for a in sys.argv[5:]]:
        for f32 in (False, True):
            r = []
            for _ in range(3):
                g, s = ctx.render(W, H, spp, variant=v, f32=f32)
                r.append(round(s["paths"] / s["kernel_ms"] / 1e3, 1))
            print("arm", arm, "SAH" if sah else "ref", "variant", v, "f32" if f32 else "f64", r, hashlib.sha256(g.tobytes()).hexdigest()[:10], flush=True)
