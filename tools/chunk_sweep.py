"""C3 throughput against the sample-chunk size (work-item granularity of the persistent kernel)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import orc
rt = orc.rt()
sc = rt.Scene.reference(5)
ctx = rt.Context(sc, 0)
ctx.render(600, 600, 8)
for chunk in (0, 168, 84, 42, 21, 10, 8):
    best = 0
    for _ in range(3):
        g, s = ctx.render(600, 600, 1000, chunk=chunk)
        best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
    print("chunk", chunk, "->", s["chunk"], "n_chunks", s["n_chunks"], "Mpaths/s", round(best, 1), flush=True)
