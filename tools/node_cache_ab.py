"""The node cache of the big scenes' stack-walk kernels (csrc/rt_walk_table.h: the most visited node records in LDS) against the same
kernels reading every record from memory (RT1W_NO_NODE_CACHE): the frames must be the same bits; kernel Mpaths/s, best of 3.
usage: python tools/node_cache_ab.py [spp]   (run on a GPU box from the repo root)"""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfgs = [(7, None, (800, 800), "best_axis", 0, {}), (7, None, (800, 800), "reference", 0, {}), (7, None, (800, 800), "sah", 1, {}),
        (0, 1.5, (1200, 800), "best_axis", 0, {"classic_walk": True}), (6, None, (600, 600), "best_axis", 0, {"variant": 3, "generic": True})]
for arm, aspect, (W, H), build, nf, kw in cfgs:
    sc = rt.Scene.reference(arm, aspect_ratio=aspect, build_seed=1).set_bvh_build(build)
    if nf:
        sc.set_walk_order(1)
    ctx = rt.Context(sc, 0)
    row = []
    for off in (False, True):
        g, s = ctx.render(160, 120, 6, no_node_cache=off, **kw)
        h = hashlib.sha256(g.tobytes()).hexdigest()[:10]
        best = 0.0
        for _ in range(3):
            g2, s2 = ctx.render(W, H, SPP, no_node_cache=off, **kw)
            best = max(best, s2["paths"] / s2["kernel_ms"] / 1e3)
        row.append((h, s["segments"], hex(s["sorted"]), round(best, 1), hashlib.sha256(g2.tobytes()).hexdigest()[:10]))
    same = row[0][0] == row[1][0] and row[0][1] == row[1][1] and row[0][4] == row[1][4]
    print("arm %d %-9s%s %s: cache %s | no cache %s | %s | x%.3f" % (arm, build, "+nf" if nf else "", kw, row[0], row[1], "SAME BITS" if same else "DIFFERENT", row[0][3] / row[1][3]), flush=True)
    ctx.close()
