"""RT1W_PROBE_COHERENT (measurement mode: every wave traces ONE path 64 times): kernel time and segments per path of the probe next to the
normal render, per workload.  The counters behind bench.py's roofline.valu are taken by tools/bench_pmc.sh; this is the quick look.
  python3 tools/probe_check.py [spp]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for arm, aspect, W, H in ((5, None, 600, 600), (0, 1.5, 1200, 800), (7, None, 800, 800)):
    ctx = rt.Context(rt.Scene.reference(arm, aspect_ratio=aspect), 0)
    try:
        ctx.specialise()
    except Exception:
        pass
    for probe in (False, True):
        ctx.render(W, H, 4, probe_coherent=probe)
        _, st = ctx.render(W, H, spp, probe_coherent=probe)
        print(f"arm {arm} probe {int(probe)}: kernel {st['kernel_ms']:8.2f} ms  segments/lane-path {st['segments'] / (W * H * spp):.3f}  flags {st['sorted']}", flush=True)
    ctx.close()
