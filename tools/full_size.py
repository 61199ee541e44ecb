"""Soak: BASELINE configurations C2, C4 and C5 at their FULL sizes through rt1w_render_rows (u8, strips, progress).
usage: full_size.py [C2] [C4] [C5] [--no-node-cache]   (the flag: the stack-walk kernels without the LDS node cache, for an A/B on one box)"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import orc
rt = orc.rt()
NO_CACHE = "--no-node-cache" in sys.argv
which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["C2", "C4", "C5"]
cfg = {"C2": (0, 1.5, 1200, 800, 500), "C4": (7, None, 800, 800, 10000), "C5": (5, 16.0 / 9.0, 3840, 2160, 10000)}
for name in which:
    arm, aspect, W, H, spp = cfg[name]
    sc = rt.Scene.reference(arm, aspect_ratio=aspect)
    ctx = rt.Context(sc, 0)
    marks = []
    t0 = time.time()
    img, st = ctx.render_rows(W, H, spp, u8=True, progress=lambda d, t: marks.append((d, round(time.time() - t0, 1))) or 0, no_node_cache=NO_CACHE)
    wall = time.time() - t0
    print(f"{name}: {W}x{H}x{spp} = {st['paths'] / 1e9:.2f} Gpaths  wall {wall:.1f} s  device {st['kernel_ms'] / 1e3:.1f} s  "
          f"{st['paths'] / wall / 1e6:.1f} Mpaths/s whole call  segments/path {st['segments'] / st['paths']:.3f}  "
          f"kernel {'specialised' if st['sorted'] & 4 else 'generic'} V{st['variant']}{' +node cache' if st['sorted'] & 1024 else ''}  strips {len(marks)}  "
          f"sha256(u8) {hashlib.sha256(img.tobytes()).hexdigest()[:16]}  mean {img.mean(axis=(0, 1)).round(2).tolist()}", flush=True)
