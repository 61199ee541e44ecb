"""Work-item size (samples per item, rt1w_render_params.chunk) on the big scenes: the last items
started keep a few lanes busy while the rest of the GPU has retired, and an item of a stack-walk scene is long (final_scene: ~110 us per
segment and lane).  kernel ms and Mpaths/s per chunk;  python3 tools/chunk_tail.py [arm] [W] [H] [spp] [chunks...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
a = sys.argv[1:]
arm, W, H, spp = (int(x) for x in (a[:4] + ["7", "800", "800", "400"][len(a[:4]):]))
chunks = [int(x) for x in a[4:]] or [0, 4, 8, 16, 32, 64]
ctx = rt.Context(rt.Scene.reference(arm, aspect_ratio=W / H), 0)
ctx.render(W, H, 8)
for ch in chunks:
    best, st = 1e30, None
    for _ in range(2):
        _, s = ctx.render(W, H, spp, chunk=ch)
        if s["kernel_ms"] < best:
            best, st = s["kernel_ms"], s
    print(f"arm {arm} {W}x{H}x{spp} chunk {ch:4d} -> {st['chunk']:4d} samples per item, {st['n_chunks']:4d} chunks: kernel {best:9.2f} ms  {W * H * spp / best / 1e3:8.1f} Mpaths/s", flush=True)
ctx.close()
