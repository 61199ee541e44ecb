"""The three BVH builds side by side on the big scenes (kernel Mpaths/s; default kernels): the reference's BVHNode::new with build seed 1,
RT1W_BVH_BEST_AXIS (the reference's rule with the axis chosen), RT1W_BVH_SAH (+ near-far with --nf).
  python3 tools/tree_ab.py [spp] [--nf]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
spp = int(args[0]) if args else 100
for arm, aspect, W, H in ((0, 1.5, 1200, 800), (7, None, 800, 800)):
    for build in ("reference", "best_axis", "sah"):
        sc = rt.Scene.reference(arm, aspect_ratio=aspect).set_bvh_build(build)
        if "--nf" in sys.argv and build != "reference":
            sc.set_walk_order(1)
        ctx = rt.Context(sc, 0)
        ctx.render(W, H, 4)
        best, st = 0.0, None
        for _ in range(3):
            img, st = ctx.render(W, H, spp)
            best = max(best, W * H * spp / st["kernel_ms"] / 1e3)
        print(f"arm {arm} {build:10s} nodes {sc.info()['n_nodes']:5d} stack {sc.info()['stack_need']:2d} V{st['variant']} flags {st['sorted']:4d}: "
              f"{best:8.1f} Mpaths/s  segments/path {st['segments'] / (W * H * spp):.3f}", flush=True)
        ctx.close()
