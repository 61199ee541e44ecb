"""f32 mode on the big scenes, default kernels (random_scene: the pair walk in f32 since round 4) against f64 and against the f32
one-entry-per-step walk: kernel Mpaths/s, best of 3.   python3 tools/f32_pw_ab.py [spp]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for arm, aspect, W, H in ((0, 1.5, 1200, 800), (7, None, 800, 800), (5, None, 600, 600)):
    for build in ("best_axis", "reference", "sah"):
        if arm == 5 and build != "best_axis":
            continue
        sc = rt.Scene.reference(arm, aspect_ratio=aspect).set_bvh_build(build)
        ctx = rt.Context(sc, 0)
        try:
            ctx.specialise()
        except Exception:
            pass
        row = []
        for label, kw in (("f64", {}), ("f32", dict(f32=True)), ("f32 classic walk", dict(f32=True, classic_walk=True))):
            ctx.render(W, H, 4, **kw)
            best, st = 0.0, None
            for _ in range(3):
                _, st = ctx.render(W, H, spp, **kw)
                best = max(best, W * H * spp / st["kernel_ms"] / 1e3)
            row.append(f"{label} {best:7.1f} (flags {st['sorted']})")
        print(f"arm {arm} {build:10s}: " + " | ".join(row), flush=True)
        ctx.close()
