"""Trace-only harness driver (raytracing-1w_amd/csrc/walk_lab.hip): dump the rays real paths trace at bounces 0..B-1, then time ONLY
the closest-hit search over them with each walk candidate and compare (t, primitive) with the product's walk bit for bit.

  python3 tools/walk_lab.py <arm> <W> <H> <spp> [bounces] [--sah] [--sweep]

Prints, per ray set (every bounce on its own, then all bounces in path order = the mix a render kernel holds) and per walk:
Mrays/s (best of 3 kernel times), lane-steps per ray, wave-steps per 64 rays, and whether every hit equals W0's."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
L = rt.load_lab()  # the diagnostics library (links librt1w.so, already loaded)
_P = C.c_void_p
L.rt1w_lab_create.argtypes = [_P, _P, C.POINTER(_P)]
L.rt1w_lab_destroy.argtypes = [_P]
L.rt1w_lab_info.argtypes = [_P, C.POINTER(C.c_uint32 * 4)]
L.rt1w_lab_dump_rays.argtypes = [_P, C.POINTER(rt.RenderParams), C.c_uint32, _P]
L.rt1w_lab_set_rays.argtypes = [_P, _P, C.c_uint64]
L.rt1w_lab_gather_probe.argtypes = [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_double * 4), C.POINTER(C.c_uint64)]
L.rt1w_lab_trace.argtypes = [_P, C.c_int, C.POINTER(C.c_uint32 * 4), C.c_int, _P, _P, _P, C.POINTER(C.c_double), C.POINTER(C.c_uint64 * 8)]


class Lab:
    def __init__(self, ctx):
        h = _P()
        rt._ck(L.rt1w_lab_create(ctx._h, ctx.scene._h, C.byref(h)))
        self._h = h
        info = (C.c_uint32 * 4)()
        rt._ck(L.rt1w_lab_info(self._h, C.byref(info)))
        self.w1_ok, self.n_inner, self.n_groups, self.variant = bool(info[0]), info[1], info[2], info[3]
        self.w1_why = None if self.w1_ok else rt.last_error()
        self.n = 0

    def close(self):
        if self._h:
            L.rt1w_lab_destroy(self._h)
            self._h = None

    def dump_rays(self, W, H, spp, bounces, tile=None, max_depth=50):
        x0, y0, tw, th = tile or (0, 0, W, H)
        p = rt.RenderParams(W, H, x0, y0, tw, th, spp, 0, max_depth, 0, 0, 0, 0, 0, 0, 0)
        out = np.zeros((bounces, tw * th * spp, 8), dtype=np.float64)
        rt._ck(L.rt1w_lab_dump_rays(self._h, C.byref(p), bounces, out.ctypes.data_as(_P)))
        return out

    def set_rays(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 8)
        rt._ck(L.rt1w_lab_set_rays(self._h, r.ctypes.data_as(_P), r.shape[0]))
        self.n = r.shape[0]

    def trace(self, mode, refill=16, votes=24, blocks_per_cu=0, repeats=3, want_hits=True, box_steps=1):
        prm = (C.c_uint32 * 4)(refill, votes, blocks_per_cu, box_steps)
        t = np.empty(self.n, dtype=np.float64)
        prim = np.empty(self.n, dtype=np.uint32)
        flags = np.empty(self.n, dtype=np.uint32)
        ms = C.c_double()
        st = (C.c_uint64 * 8)()
        rt._ck(L.rt1w_lab_trace(self._h, mode, C.byref(prm), repeats, t.ctypes.data_as(_P) if want_hits else None,
                                prim.ctypes.data_as(_P) if want_hits else None, flags.ctypes.data_as(_P) if want_hits else None,
                                C.byref(ms), C.byref(st)))
        return {"ms": ms.value, "t": t, "prim": prim, "flags": flags, "stats": [int(x) for x in st]}


def same_hits(a, b):
    return bool(np.array_equal(a["prim"], b["prim"]) and np.array_equal(a["t"].view(np.uint64), b["t"].view(np.uint64)))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    arm, W, H, spp = (int(x) for x in args[:4])
    bounces = int(args[4]) if len(args) > 4 else 6
    aspect = W / H
    sc = rt.Scene.reference(arm, aspect_ratio=aspect)
    if "--sah" in sys.argv:
        sc.set_bvh_build(True)
    ctx = rt.Context(sc, 0)
    lab = Lab(ctx)
    print(f"arm {arm} {W}x{H}x{spp}, nodes {sc.info()['n_nodes']}, product walk variant V{lab.variant}; W1: "
          + (f"{lab.n_inner} inner pair records + {lab.n_groups} leaf groups" if lab.w1_ok else f"unavailable ({lab.w1_why})"), flush=True)
    if "--gather" in sys.argv:
        for bpc in (2, 4, 8):
            ms = (C.c_double * 4)()
            nrec = C.c_uint64()
            rt._ck(L.rt1w_lab_gather_probe(lab._h, 2000, bpc, C.byref(ms), C.byref(nrec)))
            print(f"gather probe, {bpc} workgroups/CU, {nrec.value / 1e6:.0f} M records of 64 B from {sc.info()['n_nodes']} nodes: "
                  + "  ".join(f"{lbl} {nrec.value / m / 1e6:7.1f} Grec/s" for lbl, m in zip(("own/indep", "quad/indep", "own/dep", "quad/dep"), ms)), flush=True)
    rays = lab.dump_rays(W, H, spp, bounces)
    sets = []
    for b in range(bounces):
        v = rays[b][rays[b][:, 7] != 0.0]
        if len(v) > 1000000 and "--all-only" not in sys.argv:     # a persistent grid holds ~200 k lanes: smaller sets measure the tail, not the walk
            sets.append((f"bounce {b}", v))
    # the mix a render holds: every path's rays one after the other (path-major), i.e. neighbouring lanes hold different bounces
    pm = np.transpose(rays, (1, 0, 2)).reshape(-1, 8)
    sets.append(("all bounces, path order", pm[pm[:, 7] != 0.0]))
    sweep = "--sweep" in sys.argv
    for name, v in sets:
        lab.set_rays(v)
        base = min((lab.trace(0, refill=rf) for rf in ((16, 32, 48, 60) if sweep else (32,))), key=lambda r: r["ms"])
        n = len(v)
        print(f"{name:26s} {n:9d} rays | W0 {n / base['ms'] / 1e3:8.1f} Mrays/s  steps/ray {base['stats'][0] / n:6.1f}  wave-steps/64 rays "
              f"{base['stats'][1] * 64 / n:7.1f}", end="", flush=True)
        wb = min((lab.trace(7, refill=rf) for rf in (32, 48)), key=lambda r: r["ms"])
        print(f" | W0b (4 box-only steps) {n / wb['ms'] / 1e3:8.1f} Mrays/s ({base['ms'] / wb['ms']:.2f}x) wave-steps/64 rays {wb['stats'][1] * 64 / n:6.1f} hits equal W0: {same_hits(base, wb)}", end="", flush=True)
        for mode, label in ((10, "W0c (W0b, top of the stack in a register)"), (8, "W3 (pair records + 4 box-only steps)"), (9, "W3 without box-only steps")):
            w3 = min((lab.trace(mode, refill=rf) for rf in (32, 48)), key=lambda r: r["ms"])
            print(f" | {label} {n / w3['ms'] / 1e3:8.1f} Mrays/s ({base['ms'] / w3['ms']:.2f}x) steps/ray {w3['stats'][0] / n:5.1f} wave-steps/64 rays {w3['stats'][1] * 64 / n:6.1f} hits equal W0: {same_hits(base, w3)}", end="", flush=True)
        q = min((lab.trace(3, refill=rf) for rf in ((16, 32, 48, 60) if sweep else (32, 48))), key=lambda r: r["ms"])
        print(f" | W0q {n / q['ms'] / 1e3:8.1f} Mrays/s ({base['ms'] / q['ms']:.2f}x) hits equal W0: {same_hits(base, q)}", end="", flush=True)
        if lab.w1_ok:
            best = None
            for votes in ((8, 24, 40, 56) if sweep else (24,)):
                for refill in ((16, 32, 48, 60) if sweep else (32,)):
                    r = lab.trace(1, refill=refill, votes=votes)
                    ok = same_hits(base, r)
                    if sweep:
                        print(f"\n      W1 votes {votes:2d} refill {refill:2d}: {n / r['ms'] / 1e3:8.1f} Mrays/s same={ok}", end="")
                    if best is None or r["ms"] < best[0]["ms"]:
                        best = (r, votes, refill, ok)
            r, votes, refill, ok = best
            st = r["stats"]
            print(f"{chr(10) + '      best' if sweep else ' |'} W1 {n / r['ms'] / 1e3:8.1f} Mrays/s ({base['ms'] / r['ms']:.2f}x)  inner/ray {st[0] / n:5.1f} groups/ray {st[2] / n:5.1f}  "
                  f"wave-steps/64 rays box {st[1] * 64 / n:6.1f} leaf {st[3] * 64 / n:6.1f}  handed back {int((r['flags'] & 1).sum())}  hits equal W0: {ok}", end="")
        if lab.w1_ok and "--probe" in sys.argv:
            # what bounds the walk?  occupancy sweep of W0 / W1, and W1 with the inner records in LDS instead of behind the L1
            for mode, label in ((0, "W0"), (1, "W1"), (2, "W1 + inner records in LDS"), (4, "W1c"), (5, "W1c + inner records in LDS")):
                for bpc in (1, 2, 3, 4, 5):
                    try:
                        r = lab.trace(mode, refill=48, votes=24, blocks_per_cu=bpc, want_hits=False)
                    except rt.Rt1wError as e:
                        print(f"      {label}: {e}")
                        break
                    print(f"      {label:28s} {r['stats'][7]} workgroups/CU: {n / r['ms'] / 1e3:8.1f} Mrays/s")
                    if r['stats'][7] < bpc:
                        break
        for votes, bs in ((24, 1), (8, 1), (8, 4), (8, 6), (4, 6), (16, 6)):
            c = min((lab.trace(6, refill=rf, votes=votes, box_steps=bs) for rf in (32, 48)), key=lambda r: r["ms"])
            st = c["stats"]
            print(f"\n      W2 phased walk votes {votes:2d} box steps {bs}: {n / c['ms'] / 1e3:8.1f} Mrays/s ({base['ms'] / c['ms']:.2f}x)  per ray: inner {st[0] / n:5.1f} groups {st[2] / n:5.1f} other {st[4] / n:5.2f}  "
                  f"wave-steps/64 rays box {st[1] * 64 / n:6.1f} leaf {st[3] * 64 / n:6.1f}  handed back {int((c['flags'] & 1).sum())}  wg/CU {st[7]}  hits equal W0: {same_hits(base, c)}", end="")
        if lab.w1_ok:
            for votes in (8, 24, 40):
                c = min((lab.trace(4, refill=rf, votes=votes) for rf in (32, 48)), key=lambda r: r["ms"])
                print(f"\n      W1c (f32 inner boxes, 4 loads per pair) votes {votes}: {n / c['ms'] / 1e3:8.1f} Mrays/s ({base['ms'] / c['ms']:.2f}x)  wave-steps/64 rays box "
                      f"{c['stats'][1] * 64 / n:6.1f} leaf {c['stats'][3] * 64 / n:6.1f}  inner/ray {c['stats'][0] / n:5.1f}  workgroups/CU {c['stats'][7]}  hits equal W0: {same_hits(base, c)}", end="")
        print(flush=True)
    lab.close()
    ctx.close()


if __name__ == "__main__":
    main()
