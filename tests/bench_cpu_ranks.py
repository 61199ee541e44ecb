"""TEST INFRASTRUCTURE: bench.py's rank launcher and N-rank control flow (gloo rendezvous, interleaved tiles, shared host
frame, max-over-ranks timing, --check-frame) driven WITHOUT a GPU, with the CPU build of the kernel core (oracle/oracle_flat.cpp)
standing in for Context.render.  bench.py itself knows no such backend: it is injected here, and only tests run this script
(tests/test_bench_launcher.py).  Usage mirrors bench.py: `python tests/bench_cpu_ranks.py --gpus 2 --width 48 --height 48 ...`."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402
import orc    # noqa: E402


class CpuCtx:
    def __init__(self, rt, sharding, scene):
        self.rt, self.sharding, self.scene = rt, sharding, scene

    def specialise(self):
        raise self.rt.Rt1wError(-2, "CPU test backend: no specialised kernel")

    def render(self, W, H, spp, tile=None, strips=None, frame=None, max_depth=50, generic=False, chunk=0):
        tile = tile or (0, 0, W, H)
        packed, st = orc.flat_render(self.scene, W, H, spp, max_depth=max_depth, tile=tile, strips=strips, chunk=chunk, threads=2)
        out = packed
        if frame is not None:
            x0, y0, tw, th = tile
            sr, period = strips if strips else (0, 0)
            for r in range(th):
                row = y0 + r if sr == 0 else y0 + (r // sr) * period + r % sr
                frame[row, x0:x0 + tw] = packed[r]
            out = frame
        st = dict(st, kernel_ms=1.0, chunk=chunk, n_chunks=(spp + max(chunk, 1) - 1) // max(chunk, 1), grid=0, block=0,
                  variant=self.scene.info()["variant"], sorted=0)
        return out, st

    def close(self):
        pass


class CpuBackend:
    name = "cpu-core (test)"

    def __init__(self):
        self.rt = orc.rt()
        import importlib
        self.sharding = importlib.import_module("raytracing-1w_amd.sharding")

    def check_device(self, local_rank):
        # test hook: one rank dies with a message, as a rank without its GPU would
        if os.environ.get("RT1W_TEST_FAIL_RANK") == os.environ.get("RANK", "0"):
            print(f"stdout of the failing rank {os.environ.get('RANK')}", flush=True)
            raise RuntimeError(f"injected failure of rank {os.environ.get('RANK')}: no device")

    def synchronize(self):
        pass

    def scene(self, arm, aspect, bvh, walk_order):
        return self.rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect).set_bvh_build(bvh)

    def context(self, scene, dev):
        return CpuCtx(self.rt, self.sharding, scene)

    def host_frame(self, H, W):
        return np.empty((H, W, 3), dtype=np.float64)


if __name__ == "__main__":
    bench.main(backend_factory=CpuBackend, script=os.path.abspath(__file__))
