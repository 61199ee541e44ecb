"""Per-phase cycle shares of the reordering render kernel.
  python tools/stamps.py [arms...]        generic kernels of the diagnostic build librt1w_stamps.so (make -C raytracing-1w_amd/csrc stamps)
  python tools/stamps.py --jit [arms...]  the scene-specialised kernel, compiled here with the counters in (RT1W_JIT_STAMPS=1)
  python tools/stamps.py --walk [arms...] stamps library built with -DRT_STAMPS_WALK as well: the stack walk's time by node kind"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
jit = "--jit" in args
walk = "--walk" in args   # library built with tools/experiments/walk_stamps.patch applied and -DRT_STAMPS -DRT_STAMPS_WALK: buckets 8-13 are the stack walk's kinds
args = [a for a in args if a not in ("--jit", "--walk")]
if jit:
    os.environ["RT1W_JIT_STAMPS"] = "1"
    os.environ.setdefault("RT1W_KERNEL_CACHE", "/tmp/rt1w_stamps_cache")
else:
    os.environ["RT1W_LIB"] = os.path.join(ROOT, "raytracing-1w_amd", "librt1w_stamps.so")
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orc
rt = orc.rt()
names = ["-", "regen", "traverse", "hit record", "shade lambert", "shade other+tail", "loop top", "exchange: read", "sort: ballots+counts", "sort: barrier 1 wait", "exchange: rank+write", "exchange: barrier 2 wait", "barrier 1 wait: own walks ran out (ss kernels)", "barrier 1 wait: no walk at all (ss kernels)", "-", "-"]
if walk:
    names[8:14] = ["walk: pop + fetch + class", "walk: box", "walk: leaf", "walk: wrapper entry", "walk: wrapper exit", "walk: medium (outside its boundary walks)"]
    names[2] = "walk: loop control, slice votes"
big = tuple(int(x) for x in os.environ.get("RT1W_STAMPS_SIZE", "400,400,32").split(","))  # e.g. RT1W_STAMPS_SIZE=800,800,48: enough work items to keep the tail small
for arm, (W, H, spp) in ((5, (600, 600, 100)),) + tuple((int(a), big) for a in args):
    sc = rt.Scene.reference(arm)
    ctx = rt.Context(sc, 0)
    if jit:
        print("specialise:", ctx.specialise())
    ctx.render(W, H, 2, generic=not jit)
    ctx.debug_stamps(True)
    g, s = ctx.render(W, H, spp, generic=not jit)
    rc, st = ctx.debug_stamps(True)
    tot = sum(st)
    print(f"arm {arm} variant {s['variant']} sorted {s['sorted']} kernel_ms {s['kernel_ms']:.1f} segments {s['segments']} stamps_valid {rc}")
    for n, v in zip(names, st):
        if v: print(f"  {n:24s} {100.0*v/tot:5.1f}%   cycles/segment/wave {v/(s['segments']/64):8.1f}")
