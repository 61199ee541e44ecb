"""W4 of the walk lab (walk_lab.hip: lab_trace_w4): the walks of a workgroup regrouped across its four waves by the kind of their next
entry, against W0c (the render kernels' steps, wave by wave).  Hits compared bit for bit.

  python3 tools/w4_lab.py <arm> <W> <H> <spp> [bounces] [--sah]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import walk_lab as wl

rt = wl.rt


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    arm, W, H, spp = (int(x) for x in args[:4])
    bounces = int(args[4]) if len(args) > 4 else 6
    sc = rt.Scene.reference(arm, aspect_ratio=W / H)
    if "--sah" in sys.argv:
        sc.set_bvh_build(True)
    ctx = rt.Context(sc, 0)
    lab = wl.Lab(ctx)
    rays = lab.dump_rays(W, H, spp, bounces)
    pm = np.transpose(rays, (1, 0, 2)).reshape(-1, 8)
    v = pm[pm[:, 7] != 0.0]
    if "--small" in sys.argv:
        v = v[:200000]
    lab.set_rays(v)
    n = len(v)
    print(f"arm {arm} {W}x{H}x{spp}, {n} rays (all bounces, path order), nodes {sc.info()['n_nodes']}, V{lab.variant}", flush=True)
    if "--pmc" in sys.argv:  # one configuration of each walk, for the counter passes of tools/w4_pmc.sh
        for mode, rf in ((12 if lab.variant == 3 else 10, 40), (11, 128)):
            r = lab.trace(mode, refill=rf, repeats=2, want_hits=False)
            print(f"mode {mode}: {n / r['ms'] / 1e3:8.1f} Mrays/s", flush=True)
        return
    if "--occ" in sys.argv:  # how much the walk by itself gains from more resident waves (workgroups per CU)
        for mode, label in ((0, "W0"), (12 if lab.variant == 3 else 10, "W0c")):
            for bpc in (1, 2, 3, 4, 5):
                r = lab.trace(mode, refill=40, blocks_per_cu=bpc, want_hits=False)
                print(f"{label:4s} {r['stats'][7]} workgroups/CU: {n / r['ms'] / 1e3:8.1f} Mrays/s", flush=True)
                if r["stats"][7] < bpc:
                    break
        return
    base = lab.trace(0, refill=32)
    print(f"W0   {n / base['ms'] / 1e3:8.1f} Mrays/s", flush=True)
    modes = [(10, "W0c")]
    if lab.variant == 3:
        modes.append((12, "W0c, sphere-media kernel"))
    for mode, label in modes:
        for rf in (32, 40, 48):
            try:
                r = lab.trace(mode, refill=rf)
            except rt.Rt1wError as e:
                print(f"{label}: {e}")
                break
            print(f"{label:26s} refill {rf:3d}: {n / r['ms'] / 1e3:8.1f} Mrays/s ({base['ms'] / r['ms']:.2f}x W0) wave-steps/64 rays {r['stats'][1] * 64 / n:6.1f} hits equal W0: {wl.same_hits(base, r)}", flush=True)
    for mode, label in ((11, "W4 (sorted, 4 box-only steps)"), (13, "W4 (sorted, no box-only steps)")):
        for idle in ((96, 128, 160, 192, 224) if "--small" not in sys.argv else (160,)):
            try:
                r = lab.trace(mode, refill=idle)
            except rt.Rt1wError as e:
                print(f"{label}: {e}")
                break
            st = r["stats"]
            print(f"{label:30s} slice ends after {idle:3d} of 256: {n / r['ms'] / 1e3:8.1f} Mrays/s ({base['ms'] / r['ms']:.2f}x W0) full steps/ray {st[0] / n:5.1f} "
                  f"rounds per 256 rays {st[1] * 256 / n:6.1f} slices per 256 rays {st[2] * 256 / n:5.2f} wg/CU {st[7]} hits equal W0: {wl.same_hits(base, r)}", flush=True)


if __name__ == "__main__":
    main()
