"""Quick GPU throughput check (not the contract bench): Cornell + the other scenes, kernel time only."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import orc
rt = orc.rt()
which = sys.argv[1:] or ['5']
cfg = {5: (600, 600, 200), 0: (1200, 800, 50), 7: (400, 400, 50), 6: (600, 600, 50)}
for a in which:
    arm = int(a)
    W, H, spp = cfg[arm]
    sc = rt.Scene.reference(arm, aspect_ratio=1.5 if arm == 0 else None)
    ctx = rt.Context(sc, 0)
    ctx.render(W, H, 4)
    best = 0
    for _ in range(3):
        g, s = ctx.render(W, H, spp)
        best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
    print("arm", arm, "variant", s["variant"], "Mpaths/s", round(best, 1), "grid", s["grid"], flush=True)
