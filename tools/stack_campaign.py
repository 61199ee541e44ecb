"""One-off confidence run for the stack-walk kernels (sliced walk, variants V2/V3/V5, f64): GPU frame == CPU build of the core
(bits, segment counts) on many random scene graphs at a frame size that fills several workgroups, plus the SAH rebuild of each graph
and the near-far order on top (GPU == CPU build on those too)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import orc
from dual import random_scene_pair
rt = orc.rt()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W, H, spp = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (96, 64, 6)
done = bad = 0
t0 = time.time()
stats = {"media": 0, "moving": 0, "wrappers": 0, "V5": 0, "node_cache_renders": 0, "renders": 0}
for seed in range(7000, 7000 + n):
    prod, _ = random_scene_pair(seed)
    for mode in ("reference", "sah", "sah+near-far"):
        if mode == "sah":
            prod.set_bvh_build(True)
        if mode == "sah+near-far":
            prod.set_walk_order(1)
        info = prod.info()
        variants = [3] + ([2] if not info["has_media"] else []) + ([5] if (not info["has_media"] and info["scope_depth"] == 0) else [])
        if mode == "sah+near-far":
            variants = [4]
        c = rt.Context(prod, 0)
        for v in variants:
            want, sw = orc.flat_render(prod, W, H, spp, variant=v)
            got, sg = c.render(W, H, spp, variant=v)
            ok = sg["variant"] == v and sg["segments"] == sw["segments"] and np.array_equal(got, want, equal_nan=True)
            bad += 0 if ok else 1
            stats["renders"] += 1; stats["node_cache_renders"] += 1 if (sg["sorted"] & 1024) else 0
            if not ok:
                print("MISMATCH seed", seed, mode, "variant", v, info, flush=True)
        c.close()
    stats["media"] += info["has_media"]; stats["moving"] += info["has_moving"]; stats["wrappers"] += info["scope_depth"] > 0
    stats["V5"] += (not info["has_media"] and info["scope_depth"] == 0)
    done += 1
    if done % 10 == 0:
        print(done, "graphs,", bad, "mismatches,", round(time.time() - t0), "s", stats, flush=True)
print("done", done, "mismatches", bad, stats)
sys.exit(1 if bad else 0)
