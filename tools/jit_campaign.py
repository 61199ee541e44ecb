"""One-off confidence run: scene-specialised kernel == generic kernel (bits, segment counts) on many random scene graphs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.environ.setdefault("RT1W_KERNEL_CACHE", "/tmp/rt1w_campaign_cache")
import numpy as np
import orc
from dual import random_scene_pair
rt = orc.rt()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
done = bad = 0
t0 = time.time()
stats = {"media": 0, "tex": 0, "moving": 0, "depth3": 0, "gt64": 0}
for seed in range(5000, 5000 + 10 * n):
    prod, _ = random_scene_pair(seed)
    info = prod.info()
    if info["n_nodes"] > 256:
        continue
    c = rt.Context(prod, 0)
    c.specialise()
    a, sa = c.render(40, 28, 6)
    b, sb = c.render(40, 28, 6, generic=True)
    ok = bool(sa["sorted"] & 4) and sa["segments"] == sb["segments"] and np.array_equal(a, b, equal_nan=True)
    bad += 0 if ok else 1
    if not ok:
        print("MISMATCH seed", seed, info, flush=True)
    stats["media"] += info["has_media"]; stats["tex"] += info["has_textures"]; stats["moving"] += info["has_moving"]
    stats["depth3"] += info["scope_depth"] >= 3; stats["gt64"] += info["n_nodes"] > 64
    c.close()
    done += 1
    if done % 10 == 0:
        print(done, "graphs,", bad, "mismatches,", round(time.time() - t0), "s", stats, flush=True)
    if done == n:
        break
print("done", done, "mismatches", bad, stats)
sys.exit(1 if bad else 0)
