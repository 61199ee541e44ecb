import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
rt = importlib.import_module("raytracing-1w_amd")
for sah in (False, True):
    sc = rt.Scene.reference(0, aspect_ratio=1.5).set_bvh_build(sah)
    ctx = rt.Context(sc, 0)
    ctx.render(1200, 800, 2)
    for kw in ({}, dict(lds_nodes=True)):
        r = [round(s["paths"] / s["kernel_ms"] / 1e3, 1) for s in (ctx.render(1200, 800, 64, **kw)[1] for _ in range(3))]
        print("sah", sah, kw, r, flush=True)
