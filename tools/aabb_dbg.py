import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, orc
rt = orc.rt()
ctx = rt.Context(rt.Scene.reference(5), 0)
rng = np.random.default_rng(7)
n = 200_000
c = np.empty((n, 14))
lo = rng.uniform(-10, 10, (n, 3)); ext = rng.uniform(0, 8, (n, 3))
c[:, 0:3] = lo; c[:, 3:6] = lo + ext
c[:, 6:9] = rng.uniform(-15, 15, (n, 3))
c[:, 9:12] = rng.normal(size=(n, 3))
c[:, 12] = rng.choice([0.001, -np.inf, 0.0, 1.0], n)
c[:, 13] = rng.choice([np.inf, 5.0, 50.0, 0.5], n)
special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 1e-300, 1e300])
k = n // 2
idx = rng.integers(0, 3, k)
c[np.arange(k), 9 + idx] = rng.choice(special, k)
k2 = n // 4
ax = rng.integers(0, 3, k2)
c[np.arange(k2), 6 + ax] = c[np.arange(k2), 0 + ax]
c[np.arange(0, n, 97), 6 + rng.integers(0, 3)] = np.nan
lit, fast = ctx.debug_aabb(c)
host = np.array([orc.A.orc_aabb_hit(r[0:3].ctypes.data_as(C.c_void_p), r[3:6].ctypes.data_as(C.c_void_p), r[6:9].ctypes.data_as(C.c_void_p), r[9:12].ctypes.data_as(C.c_void_p), r[12], r[13]) for r in np.ascontiguousarray(c[:20000])], dtype=np.int32)
bad = np.nonzero(lit[:20000] != host)[0]
print("lit vs host mismatches", len(bad))
for i in bad[:5]: print(c[i], lit[i], host[i])
bad2 = np.nonzero(lit != fast)[0]
print("lit vs fast mismatches", len(bad2))
for i in bad2[:8]: print(c[i], lit[i], fast[i])
