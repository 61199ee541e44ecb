import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
rt = importlib.import_module("raytracing-1w_amd")
for arm in (5, 6):
    for sah in (False, True):
        sc = rt.Scene.reference(arm).set_bvh_build(sah)
        ctx = rt.Context(sc, 0)
        info = ctx.specialise()
        ctx.render(600, 600, 50)
        r = [round(s["paths"] / s["kernel_ms"] / 1e3, 1) for s in (ctx.render(600, 600, 300)[1] for _ in range(3))]
        print("arm", arm, "sah", sah, "nodes", sc.info()["n_nodes"], "vgprs", info["vgprs"], "compile", info["compile_ms"], r, flush=True)
