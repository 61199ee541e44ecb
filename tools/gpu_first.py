"""First GPU contact: GPU render vs CPU build of the same core (bit-exact expected), then a timing."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import orc
rt = orc.rt()
print(rt.version(), 'devices', rt.device_count(), flush=True)
for arm, (W, H, spp) in {5: (64, 64, 16), 0: (60, 40, 8), 6: (48, 48, 8), 7: (48, 48, 8), 2: (48, 27, 8), 3: (48, 27, 8), 4: (48, 27, 8), 1: (48, 27, 8)}.items():
    sc = rt.Scene.reference(arm, build_seed=1)
    ctx = rt.Context(sc, 0)
    g, sg = ctx.render(W, H, spp)
    b, sb = orc.flat_render(sc, W, H, spp, chunk=sg['chunk'])
    same = np.array_equal(g, b) or np.array_equal(np.nan_to_num(g, nan=-1), np.nan_to_num(b, nan=-1))
    print(f"arm {arm}: bit-exact={same} maxabs={np.nanmax(np.abs(g-b)):.3e} segsG={sg['segments']} segsB={sb['segments']} chunk={sg['chunk']}x{sg['n_chunks']} kernel_ms={sg['kernel_ms']:.2f} grid={sg['grid']}", flush=True)
    ctx.close()
# timing: Cornell 600x600
sc = rt.Scene.reference(5, build_seed=1)
ctx = rt.Context(sc, 0)
for spp in (10, 100):
    t = time.time(); g, s = ctx.render(600, 600, spp); dt = time.time() - t
    print(f"cornell 600x600x{spp}: kernel_ms={s['kernel_ms']:.1f} total_ms={s['total_ms']:.1f} wall={dt*1e3:.1f} Mpaths/s={s['paths']/s['kernel_ms']/1e3:.1f} nbar={s['segments']/s['paths']:.3f} chunk={s['chunk']}x{s['n_chunks']} mean={g.mean():.5f}", flush=True)
