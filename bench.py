#!/usr/bin/env python3
"""bench.py -- Mpaths/s of the per-pixel sample loop on MI355X (BASELINE.json metric).

A "step" = one render of the workload: BASELINE config C3, the reference's Cornell box
(src/main.rs:395-512, lights :873-887, camera :888-892) at 600x600, 1000 spp, depth 50,
f64, scene procedural (no external data), scene + camera resident in HBM before timing,
framebuffer left in HBM (device-resident rate; the host-copy-inclusive rate is reported
as `value_incl_d2h`, never as `value`).

N GPUs (one process per GPU, launched by torch.distributed.run): weak scaling by sample range --
rank r renders samples [r*1000, (r+1)*1000) of every pixel; no data-path collective (the
ranks only share a barrier and a max-reduce of the elapsed time).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, SPP, DEPTH = 600, 600, 1000, 50
RECORD_BYTES = 128          # f64 SoA ray-state record
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(cores):
    """Literal C++ restatement of the reference (oracle/oracle.cpp, kind 'port' -- the Rust crate
    cannot be built here), same scene/camera/size, bounded spp (Mpaths/s does not depend on spp)."""
    import orc
    sc = orc.OracleScene(5, build_seed=1)
    spp = 4
    t0 = time.time()
    _, st = sc.render(W, H, spp, max_depth=DEPTH, threads=cores)
    dt = time.time() - t0
    # scale the sample to ~12 s of CPU work
    spp2 = max(4, min(512, int(spp * 12.0 / max(dt, 1e-3))))
    t0 = time.time()
    _, st = sc.render(W, H, spp2, max_depth=DEPTH, threads=cores)
    dt = time.time() - t0
    return {"value": round(st["paths"] / dt / 1e6, 4), "unit": "Mpaths/s", "cores": cores, "kind": "port",
            "sample": f"cornell_box {W}x{H}, {spp2} spp, depth {DEPTH} ({st['paths']} paths, {dt:.1f} s, "
                      f"std::thread pool of {cores})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generic", action="store_true", help="ablation: the generic kernel instead of the scene-specialised one")
    ap.add_argument("--spp", type=int, default=SPP, help="debug only; the contract workload is 1000")
    ap.add_argument("--all-ranks-on-device", type=int, default=None,
                    help="rehearsal only (1-GPU box): every rank uses this device instead of LOCAL_RANK")
    a = ap.parse_args()

    import torch
    import orc
    rt = orc.rt()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only: barrier + max(time)

    if a.all_ranks_on_device is not None:
        local_rank = a.all_ranks_on_device
    assert torch.cuda.is_available() and rt.device_count() > local_rank, "bench.py needs the GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    scene = rt.Scene.reference(5, build_seed=1)
    ctx = rt.Context(scene, local_rank)
    # kernel specialised for this scene's topology: from the kernel cache the build fills (raytracing-1w_amd/kernels), or compiled
    # here with hiprtc (3-5 s, outside the timed region like the rest of the set-up); --generic keeps the generic kernel
    spec = None
    if not a.generic:
        try:
            spec = ctx.specialise()
        except rt.Rt1wError as e:
            spec = {"active": False, "error": str(e)}
    spp = a.spp
    out = torch.empty((H, W, 3), dtype=torch.float64, device=f"cuda:{local_rank}")
    kw = dict(max_depth=DEPTH, sample_offset=rank * spp, out_sum=(world > 1), generic=a.generic)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        ctx.render_device(out.data_ptr(), W, H, spp, **kw)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    segs = 0
    for _ in range(a.steps):
        st = ctx.render_device(out.data_ptr(), W, H, spp, **kw)
        kernel_ms.append(st["kernel_ms"])
        segs = st["segments"]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    paths_per_step = W * H * spp * world
    value = paths_per_step * a.steps / elapsed / 1e6

    if rank == 0:
        # roofline of the dominant kernel (rt_render_kernel): algorithmic bytes per launch =
        # 2 * 128 B per traced segment (ray-state record read + written once per segment) + 24 B per pixel
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        algo_bytes = 2 * RECORD_BYTES * segs + 24 * W * H
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        pmc = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tp):
            try:
                prof = json.load(open(tp))
                traffic = prof.get("hbm_bytes_per_launch")   # rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this command (profiles/)
                pmc = prof.get("pmc")
            except Exception:
                traffic = None
        line = {
            "metric": "Mpaths/s", "value": round(value, 2), "unit": "Mpaths/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (procedural Cornell box of the reference, build_seed 1; no external data)",
            "config": {"workload": f"C3 cornel_box (scene arm 5) {W}x{H}, {spp} spp per GPU, depth {DEPTH}, "
                                   f"mixture-PDF light sampling; N GPUs = sample ranges [r*{spp},(r+1)*{spp})",
                       "width": W, "height": H, "spp_per_gpu": spp, "max_depth": DEPTH,
                       "paths_per_step": paths_per_step, "segments_per_path": round(segs / (W * H * spp), 4),
                       "chunk": st["chunk"], "n_chunks": st["n_chunks"], "grid": st["grid"], "block": st["block"],
                       "kernel_variant": st["variant"], "workgroup_path_sort": bool(st.get("sorted", 0) & 1),
                       "scene_specialised_kernel": bool(st.get("sorted", 0) & 4), "specialise": spec,
                       "parallelism": f"sample-range x{world}, host gather, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": "rt_jit_sorted" if (st.get("sorted", 0) & 4) else ("rt_render_kernel_sorted<V%d>" % st["variant"] if (st.get("sorted", 0) & 1) else "rt_render_kernel<V%d>" % st["variant"]),
                         "kernel_ms": round(avg_ms, 3),
                         "algorithmic_bytes_per_launch": algo_bytes, "pmc": pmc,
                         "note": "compute-bound f64 kernel: ray state stays in VGPRs, so real HBM traffic is far "
                                 "below the algorithmic record traffic of SURVEY 8(d)"},
        }
        if world == 1:
            # host-copy-inclusive rate (rt1w_render into host memory), reported separately
            t1 = time.perf_counter()
            _, st2 = ctx.render(W, H, spp, max_depth=DEPTH)
            line["value_incl_d2h"] = round(W * H * spp / (time.perf_counter() - t1) / 1e6, 2)
            if not a.no_cpu_baseline:
                cores = min(16, os.cpu_count() or 1)
                line["cpu_baseline"] = cpu_baseline(cores)
        print(json.dumps(line), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
