import importlib, os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tools"))
import walk_lab as wl
rt = wl.rt
for arm, W, H, spp, B in ((7, 800, 800, 2, 50), (0, 1200, 800, 2, 50)):
    sc = rt.Scene.reference(arm, aspect_ratio=W / H)
    ctx = rt.Context(sc, 0)
    lab = wl.Lab(ctx)
    rays = lab.dump_rays(W, H, spp, B)
    pm = np.transpose(rays, (1, 0, 2)).reshape(-1, 8)
    v = pm[pm[:, 7] != 0.0]
    lab.set_rays(v)
    for rf in (32, 48):
        for bpc in (0, 3, 4):
            r = lab.trace(0, refill=rf, blocks_per_cu=bpc, want_hits=False)
            print(f"arm {arm}: lab W0 refill {rf} wg/CU {r['stats'][7]}: {len(v)} rays ({len(v) / (W * H * spp):.2f} per path) {len(v) / r['ms'] / 1e3:8.1f} Mrays/s steps/ray {r['stats'][0] / len(v):.1f}", flush=True)
    for _ in range(2):
        img, st = ctx.render(W, H, 16)
    print(f"arm {arm}: product megakernel V{st['variant']}: {st['paths'] / st['kernel_ms'] / 1e3:.1f} Mpaths/s = {st['segments'] / st['kernel_ms'] / 1e3:.1f} Msegments/s ({st['segments'] / st['paths']:.2f} seg/path)", flush=True)
    lab.close(); ctx.close()
