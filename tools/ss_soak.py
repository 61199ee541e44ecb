"""Soak of the kernels that reorder the finished paths at the end of every slice: the same render N times must give the same bytes
every time (a missing barrier or a racy exchange would show as a frame that differs now and then), and those bytes must be the plain
kernel's.   python3 tools/ss_soak.py [repeats]"""
import hashlib
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
for arm, aspect, W, H, spp, sah in ((7, 1.0, 800, 800, 12, False), (7, 1.0, 800, 800, 12, True), (0, 1.5, 1200, 800, 12, False), (7, 1.0, 203, 157, 9, False)):
    sc = rt.Scene.reference(arm, aspect_ratio=aspect)
    if sah:
        sc.set_bvh_build(True)
        sc.set_walk_order(1)
    ctx = rt.Context(sc, 0)
    ref, sr = ctx.render(W, H, spp, unsorted=True)
    want = hashlib.sha256(ref.tobytes()).hexdigest()
    seen = {}
    for i in range(N):
        img, st = ctx.render(W, H, spp)
        assert st["sorted"] & 512
        h = hashlib.sha256(img.tobytes()).hexdigest()
        seen[h] = seen.get(h, 0) + 1
        bad += (h != want) or (st["segments"] != sr["segments"])
    print(f"arm {arm}{' SAH + near-far' if sah else ''} {W}x{H}x{spp}: {N} renders, {len(seen)} distinct frame(s), equal to the plain kernel's: {list(seen) == [want]}", flush=True)
    ctx.close()
# the specialised kernels (four waves per SIMD, exchange in two rounds for Cornell / cornel_smoke; three waves for the noise scenes):
# N renders, one frame, equal to the generic plain kernel's
for arm, W, H, spp in ((5, 600, 600, 20), (6, 600, 600, 20), (2, 400, 225, 20), (5, 203, 157, 9)):
    ctx = rt.Context(rt.Scene.reference(arm), 0)
    ref, sr = ctx.render(W, H, spp, unsorted=True, generic=True, chunk=rt.default_chunk(W, H, spp))
    want = hashlib.sha256(ref.tobytes()).hexdigest()
    seen = {}
    for i in range(N):
        img, st = ctx.render(W, H, spp)
        assert st["sorted"] & 4
        h = hashlib.sha256(img.tobytes()).hexdigest()
        seen[h] = seen.get(h, 0) + 1
        bad += (h != want) or (st["segments"] != sr["segments"])
    print(f"arm {arm} specialised {W}x{H}x{spp}: {N} renders, {len(seen)} distinct frame(s), equal to the generic plain kernel's: {list(seen) == [want]}", flush=True)
    ctx.close()
print("mismatches", bad)
sys.exit(1 if bad else 0)
