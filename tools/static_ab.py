"""A/B: generic kernel vs scene-specialised kernel (rt1w_context_specialise), Cornell; hash first."""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import orc
rt = orc.rt()
sc = rt.Scene.reference(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
ctx = rt.Context(sc, 0)
print("specialise:", ctx.specialise(), flush=True)
for generic in (True, False):
    g, s = ctx.render(96, 96, 16, generic=generic)
    h = hashlib.sha256(g.tobytes()).hexdigest()[:16]
    best = []
    for _ in range(4):
        g2, s2 = ctx.render(600, 600, 200, generic=generic)
        best.append(round(s2["paths"] / s2["kernel_ms"] / 1e3, 1))
    print("generic" if generic else "specialised", "hash", h, "segments", s["segments"], "sorted", s2["sorted"], "Mpaths/s", best, flush=True)
