"""walk lab: which rays does a candidate walk answer differently from the product's walk?  python3 tools/lab_debug.py <arm> <W> <H> <spp> <bounces> <mode>"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import walk_lab as wl
rt = wl.rt
arm, W, H, spp, B, mode = (int(x) for x in sys.argv[1:7])
sc = rt.Scene.reference(arm, aspect_ratio=W / H)
ctx = rt.Context(sc, 0)
lab = wl.Lab(ctx)
rays = lab.dump_rays(W, H, spp, B)
pm = np.transpose(rays, (1, 0, 2)).reshape(-1, 8)
keep = pm[:, 7] != 0.0
v = pm[keep]
bounce = np.tile(np.arange(B), W * H * spp)[keep]
lab.set_rays(v)
a = lab.trace(0)
b = lab.trace(mode)
bad = np.nonzero((a["prim"] != b["prim"]) | (a["t"].view(np.uint64) != b["t"].view(np.uint64)))[0]
u = sc.flat(0).view(np.uint32).reshape(-1, 24)
kinds = u[:, 0] & 0xFF
print(len(v), "rays,", len(bad), "differ; by bounce:", np.bincount(bounce[bad], minlength=B)[:12])
def kd(p): return int(kinds[p]) if p != 0xFFFFFFFF else -1
for i in bad[:12]:
    print(f"ray {i} bounce {bounce[i]} o {v[i, :3]} d {v[i, 3:6]} time {v[i, 6]:.4f} | W0 t {a['t'][i]!r} prim {a['prim'][i]} kind {kd(a['prim'][i])} | cand t {b['t'][i]!r} prim {b['prim'][i]} kind {kd(b['prim'][i])} flags {b['flags'][i]}")
import collections
print("W0 kinds of differing rays:", collections.Counter(kd(p) for p in a["prim"][bad]).most_common(), " candidate kinds:", collections.Counter(kd(p) for p in b["prim"][bad]).most_common())
