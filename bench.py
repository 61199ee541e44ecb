#!/usr/bin/env python3
"""bench.py -- Mpaths/s of the per-pixel sample loop on MI355X (BASELINE.json metric).

A "step" = one render of the workload INCLUDING the device->host gather (t_render covers kernel(s) +
device->host gather; scene build/upload and PPM text excluded).  The scene and camera are resident in HBM before timing.

Workloads (--workload; BASELINE.json configs, reference lines under /root/reference/src):
  c3  (default; the configuration the metric is quoted on)  Cornell box main.rs:395-512, lights :873-887, camera :888-892,
      600x600, 1000 spp, depth 50, `ray_color` with mixture-PDF light sampling (main.rs:51-116)
  c2  random_scene main.rs:192-295, arm :816-827, 1200x800, 500 spp, `ray_color_without_light_objects` (main.rs:118-190)
  c4  final_scene main.rs:635-795, arm :916-936, 800x800, 10 000 spp (the reference's own size; --spp to shorten)
  c5  Cornell box 3840x2160 (16:9), --spp (default 1000; BASELINE names 10 000 = 22 s per step per GPU)

  N = 1  one GPU renders the whole frame into a pinned host frame (rt1w_render, RT1W_OUT_FRAME).  The c3 line also carries
         `other_configs`: C2 at full size and C4 at 400 spp measured the same way (D2H-inclusive Mpaths/s, kernel ms,
         segments/path, roofline block), each on three trees: the default build (c2, c4: BVHNode::new with the split axis chosen,
         RT1W_BVH_BEST_AXIS), the axes drawn from build seed 1 (c2_drawn_axes, c4_drawn_axes: rounds 1-3's default) and the opt-in
         SAH trees with near-far order (c2_sah, c4_sah), so every single-GPU BASELINE config has one driver-timed number per tree.
  N > 1  the SAME job shape, image-tiled over the GPUs of one node as the north star says: 16-row strips dealt round-robin
         (sharding.interleaved_tile; one launch per GPU renders all of its strips), every rank's device->host copy writes
         its strips straight into ONE shared pinned host frame (sharding.SharedFrame) -- the host gather, inside the
         timed region.  No data-path collective; ranks share a barrier and a max-reduce of the elapsed time (gloo).
         c3: weak scaling (task rule: independent units sharded across ranks): the frame grows with N at constant 1000 spp
         and constant camera -- side = 16N*round(600*sqrt(N)/16N): 600, 864, 1216, 1664, a whole number of 16-row strips per
         GPU -- so every GPU keeps C3's 3.6e8 paths per step within 4 %.  --strong keeps 600x600 for all N.
         c2 / c4 / c5 are one fixed frame each (BASELINE: "tiled across 8 MI355X"): strong scaling.
         The default N > 1 line (c3, weak) also carries `other_configs` = BASELINE's two multi-GPU configs as strong-scaling jobs of
         the same N ranks: `c5_strong` (Cornell 3840x2160, 1000 spp per step instead of 10 000) and `c4_strong` (final_scene 800x800,
         400 spp instead of 10 000), each with the D2H-inclusive rate, per-rank kernel ms (min / max: the strips' load balance),
         `n_ranks_seen`, and `gathered_frame_equals_single_gpu_frame` under --check-frame.

Launching.  `python bench.py --gpus N` with no launcher around it starts its own N rank processes (the replacement of rayon's
all-core into_par_iter, main.rs:957-963): fresh children created BEFORE this process touches the GPU or imports torch,
each given RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; the parent relays rank 0's JSON line and exits
non-zero if any child does.  Under `python -m torch.distributed.run` (WORLD_SIZE set by the launcher) it is one rank.

`value` = whole-job paths / s over the timed steps, copy included.  The device-resident rate (framebuffer left in HBM)
is reported as `value_device_resident`, never as `value`.
"""
import argparse
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEPTH = 50
RECORD_BYTES = 128          # f64 SoA ray-state record
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec

WORKLOADS = {
    "c2": dict(name="C2 random_scene", arm=0, W=1200, H=800, aspect=1.5, spp=500,
               what="spheres-only BVH (MovingSphere + Sphere), lambertian/metal/dielectric, checker ground; BSDF sampling only"),
    "c3": dict(name="C3 cornel_box", arm=5, W=600, H=600, aspect=1.0, spp=1000, what="mixture-PDF light sampling"),
    "c4": dict(name="C4 final_scene", arm=7, W=800, H=800, aspect=1.0, spp=10000,
               what="BVH of 400 boxes + 1000-sphere cluster under Translate(RotateY), Perlin + image textures, two constant_medium volumes"),
    "c5": dict(name="C5 cornel_box 16:9", arm=5, W=3840, H=2160, aspect=16.0 / 9.0, spp=1000, what="mixture-PDF light sampling, 4K"),
}
# carried by the N = 1 c3 line: (entry, workload, spp, bvh, walk order).  The *_sah entries run the opt-in SAH trees (checked against the
# literal oracle on the same topology, tests/test_bvh_build.py) with the near child first where that preserves the result
OTHER_CONFIGS = (("c2", "c2", 500, "best_axis", "reference"), ("c4", "c4", 400, "best_axis", "reference"),
                 ("c2_drawn_axes", "c2", 500, "reference", "reference"), ("c4_drawn_axes", "c4", 400, "reference", "reference"),
                 ("c2_sah", "c2", 500, "sah", "near-far"), ("c4_sah", "c4", 400, "sah", "near-far"))
# carried by the default N > 1 line: BASELINE's two multi-GPU configs, tiled over the same ranks (entry, workload, spp per step, steps)
OTHER_CONFIGS_MULTI = (("c5_strong", "c5", 1000, 2), ("c4_strong", "c4", 400, 3))
DEFAULT_BVH = "best_axis"


# --------------------------------------------------------------------------------------------------------- launcher --

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


LAUNCH_TIMEOUT_S = 3300.0   # a rank stuck in the rendezvous (or anywhere else) ends the job with exit code 124 instead of hanging it


def launch_ranks(n, argv, script=None, env_extra=None, timeout=LAUNCH_TIMEOUT_S):
    """Start `n` rank processes of `script` (default: this file) with `argv`, one per GPU, and wait for them.
    Must be called from a process that has not initialised the GPU (nothing here imports torch or the HIP library);
    the children are fresh interpreters (fork + exec of python before any GPU call -- never a re-exec of a process that
    holds the GPU).  Returns (exit code, rank 0's stdout).  A failing rank ends the others; after `timeout` seconds all are killed (124).
    Rank 0's stdout goes to a temporary FILE (a pipe would block the rank once ~64 KB are unread); the other ranks' stdout is
    sent to this process's stderr, so that whatever a failing rank prints reaches the caller."""
    import tempfile
    script = script or os.path.abspath(__file__)
    port = free_port()
    procs = []
    out_file = tempfile.TemporaryFile()

    def die_with_parent():   # in the child, between fork and exec: a launcher that is killed takes its ranks with it
        try:
            import ctypes
            import signal
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM)   # PR_SET_PDEATHSIG
        except Exception:
            pass

    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RT1W_BENCH_PARENT=str(os.getpid()))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=out_file if r == 0 else sys.stderr,
                                      preexec_fn=die_with_parent))
    t_end = None if timeout is None else time.time() + timeout
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            p = procs[r]
            if p.poll() is not None:
                live.discard(r)
                if p.returncode != 0 and rc == 0:
                    rc = p.returncode if p.returncode > 0 else 1
                    print(f"bench.py: rank {r} exited with {p.returncode}; stopping the other ranks", file=sys.stderr)
                    for q in procs:
                        if q.poll() is None:
                            q.terminate()   # the exact children started above, by pid
        if live and t_end is not None and time.time() > t_end:
            rc = rc or 124
            print(f"bench.py: ranks {sorted(live)} still running after {timeout:.0f} s; killing them", file=sys.stderr)
            for q in procs:
                if q.poll() is None:
                    q.kill()
            break
        time.sleep(0.05)
    for q in procs:
        q.wait()
    # a rank that was stopped (or died) could not remove the shared host frames it had created: their names carry this launch's port and pid
    try:
        for f in os.listdir("/dev/shm"):
            if f.startswith(f"rt1w_bench_{port}_{os.getpid()}"):
                os.unlink(os.path.join("/dev/shm", f))
    except OSError:
        pass
    out_file.seek(0)
    out0 = out_file.read().decode(errors="replace")
    out_file.close()
    return rc, out0


# ---------------------------------------------------------------------------------------------------------- backend --

class GpuBackend:
    """The product: librt1w.so through the ctypes binding.  No CPU fallback: without a GPU this raises."""
    name = "hip"

    def __init__(self):
        import torch
        self.torch = torch
        self.rt = importlib.import_module("raytracing-1w_amd")          # oracle/ is only touched by cpu_baseline()
        self.sharding = importlib.import_module("raytracing-1w_amd.sharding")

    def check_device(self, local_rank):
        assert self.torch.cuda.is_available() and self.rt.device_count() > local_rank, \
            f"bench.py needs GPU {local_rank} (devices visible: {self.rt.device_count()}); there is no CPU fallback"
        self.torch.cuda.set_device(local_rank)

    def synchronize(self):
        self.torch.cuda.synchronize()

    def scene(self, arm, aspect, bvh, walk_order):
        sc = self.rt.Scene.reference(arm, build_seed=1, aspect_ratio=aspect)
        sc.set_bvh_build(bvh)            # "best_axis" (the library's default), "reference" (axes drawn from build seed 1), "sah"
        if walk_order == "near-far":
            sc.set_walk_order(1)
        return sc

    def context(self, scene, dev):
        return self.rt.Context(scene, dev)

    def host_frame(self, H, W):
        return self.rt.pinned_empty((H, W, 3))

    def device_resident_ms(self, ctx, dev, W, H, spp, tile, strips, steps, kw):
        out = self.torch.empty((tile[3], W, 3), dtype=self.torch.float64, device=f"cuda:{dev}")
        self.torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            ctx.render_device(out.data_ptr(), W, H, spp, tile=tile, strips=strips, **kw)
        self.torch.cuda.synchronize()
        return (time.perf_counter() - t1) / steps * 1e3


# ------------------------------------------------------------------------------------------------------ CPU baseline --

def host_cores():
    """host cores this process may use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands each job a share)"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return cores


def cpu_baseline(wl, width, height):
    """Literal C++ restatement of the reference (oracle/oracle.cpp, kind 'port': the Rust crate cannot be built here --
    no rustc/cargo), same scene/camera/size, all host cores (rayon's default, src/main.rs:957-963) AND one thread,
    on a bounded spp (a sample of the same workload).  The only place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    sc = orc.OracleScene(wl["arm"], build_seed=1, aspect_ratio=width / height)
    cores = host_cores()

    def run(threads, budget_s):
        spp = 1
        t0 = time.time()
        sc.render(width, height, spp, max_depth=DEPTH, threads=threads)
        dt = time.time() - t0
        spp2 = max(1, min(512, int(spp * budget_s / max(dt, 1e-3))))
        t0 = time.time()
        _, st = sc.render(width, height, spp2, max_depth=DEPTH, threads=threads)
        dt = time.time() - t0
        return st["paths"] / dt / 1e6, spp2, st["paths"], dt

    v_all, spp_all, paths_all, dt_all = run(cores, 12.0)
    v_one, spp_one, paths_one, dt_one = run(1, 8.0)
    return {"value": round(v_all, 4), "unit": "Mpaths/s", "cores": cores, "kind": "port",
            # value / (cores x single-thread value): ~1 on a box whose cores are the job's own, far below on a throttled or shared one
            "parallel_efficiency": round(v_all / (cores * v_one), 3) if v_one > 0 else None,
            "sample": f"{wl['name']} {width}x{height}, {spp_all} spp, depth {DEPTH} ({paths_all} paths in {dt_all:.1f} s on a "
                      f"std::thread pool of {cores} = the cores this job may use (affinity / cgroup quota; os.cpu_count() = {os.cpu_count()}); oracle built -O3)",
            "single_thread": {"value": round(v_one, 4), "cores": 1,
                              "sample": f"{spp_one} spp ({paths_one} paths in {dt_one:.1f} s)"}}


def sample_passes(npix, n_chunks, partial_mib=0):
    """launches of the render kernel per render: a render whose chunk partial sums (24 B per pixel and chunk) exceed the budget (8 GiB unless
    rt1w_render_params.partial_mib says otherwise) runs as several passes over sample ranges (csrc/context.hip: chunks_per_pass)"""
    budget = (partial_mib << 20) if partial_mib else (8 << 30)
    cpp = max(1, min(n_chunks, budget // max(1, npix * 24)))
    return (n_chunks + cpp - 1) // cpp


def git_head():
    """commit of the tree: `git rev-parse` where there is a repository, else the VERSION file build() wrote (the GPU box
    receives the tree without .git)"""
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, "raytracing-1w_amd", "VERSION")).read().split()[0]
    except Exception:
        return None


KERNEL_SOURCES = ("include/rt1w_num.h", "raytracing-1w_amd/csrc/rt_flat.h", "raytracing-1w_amd/csrc/rt_core.h",
                  "raytracing-1w_amd/csrc/rt_kernel_sorted.h", "raytracing-1w_amd/csrc/rt_kernel_plain.h",
                  "raytracing-1w_amd/csrc/rt_walk_pair.h", "raytracing-1w_amd/csrc/rt_kernels.h", "raytracing-1w_amd/csrc/rt_walk_table.h",
                  "raytracing-1w_amd/csrc/Makefile") # the Makefile: the compiler options the kernels are built with


def kernel_sources_id():
    """identity of the render kernels' source text (the files the f64 kernels are compiled from): what a stored PMC measurement
    is matched against -- a commit would not do, committing the measurement itself moves it"""
    import hashlib
    h = hashlib.sha256()
    try:
        for f in KERNEL_SOURCES:
            h.update(open(os.path.join(ROOT, f), "rb").read())
    except OSError:
        return None
    return h.hexdigest()[:12]


def stored_pmc(workload, kernel, spec_key):
    """Stored PMC figures (separate rocprofv3 --pmc passes, profiles/pmc_summary.json), attached only when they were taken on
    the kernel that just ran: same kernel name, same kernel source text (kernel_sources_id), same specialisation key.
    valu_lane_issue_frac = VALU-busy fraction of the SIMD cycles x active lanes per VALU
    instruction / 64: the share of the f64 VALU lane-issue capacity doing work -- the kernel's true limiter."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        ent = json.load(open(p)).get(workload)
    except Exception:
        return None
    if not ent:
        return None
    same = (ent.get("kernel") == kernel and ent.get("specialise_key") in (None, spec_key) and
            ent.get("kernel_sources") is not None and ent.get("kernel_sources") == kernel_sources_id())
    ent = dict(ent, matches_this_run=bool(same), file="profiles/pmc_summary.json")
    ent.pop("kernel_names_seen", None)
    return ent


def valu_block(pmc):
    """The VALU-side roofline from stored counters (tools/bench_pmc.sh).  `necessary` = VALU instructions ONE path executes per traced
    segment in this kernel's code, counted by the hardware in the coherent probe (RT1W_PROBE_COHERENT: the 64 lanes of a wave trace the same
    path, so a wave-instruction is one lane's worth of necessary work): SQ_INSTS_VALU x 64 / segments counted.  `issued` = lane-slots the normal
    run issues per segment (SQ_INSTS_VALU x 64 / segments), `executed` = those with an active lane (x lanes per instruction).
    efficiency = necessary / issued: 1.0 would be 64 coherent lanes in every instruction and nothing but the path's own work."""
    pr = pmc.get("coherent_probe")
    if not pr or not pr.get("segments_counted") or not pmc.get("segments_counted"):
        return None
    necessary = pr["SQ_INSTS_VALU_per_launch"] * 64.0 / pr["segments_counted"]
    issued = pmc["SQ_INSTS_VALU_per_launch"] * 64.0 / pmc["segments_counted"]
    executed = issued * pmc["lane_utilisation"]
    return {"necessary_lane_ops_per_segment": round(necessary, 1), "executed_lane_ops_per_segment": round(executed, 1),
            "issued_lane_slots_per_segment": round(issued, 1), "efficiency": round(necessary / issued, 4),
            "necessary_over_executed": round(necessary / executed, 4),
            "peak_lane_ops_per_s": 256 * 4 * 16 * 2.4e9,   # 256 CUs x 4 SIMDs x 16 lanes per clock x 2.4 GHz
            "how": "necessary: rocprofv3 SQ_INSTS_VALU of the RT1W_PROBE_COHERENT run x 64 / its segment count; issued / executed: the same "
                   "counters of the normal run (profiles/pmc_summary.json)"}


def kernel_name(st):
    flags = st.get("sorted", 0)
    if flags & 4:
        return "rt_jit_sorted"
    if flags & 128:
        return ("rt_render_kernel_pw_ss<V%d>" if (flags & 512) else "rt_render_kernel_pw<V%d>") % st["variant"]
    if flags & 512:
        base = "rt_render_kernel_ss_hc" if (flags & 1024) else "rt_render_kernel_ss" # _hc: most visited node records in LDS (rt_walk_table.h)
        return (base + ("<V%d, sphere media>" if (flags & 256) else "<V%d>")) % st["variant"]
    if flags & 256:
        return "rt_render_kernel<V%d, sphere media>" % st["variant"]
    return ("rt_render_kernel_sorted<V%d>" if (flags & 1) else "rt_render_kernel<V%d>") % st["variant"]


def roofline_block(workload, st, kernel_ms, pixels, spec_key, spp=None):
    """roofline of the dominant kernel: algorithmic bytes per launch = 2 * 128 B per traced segment (ray-state record read +
    written once per segment) + 24 B per pixel; duration from HIP events on the kernel's own stream"""
    segs = st["segments"]
    algo_bytes = 2 * RECORD_BYTES * segs + 24 * pixels
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    kernel = kernel_name(st)
    pmc = stored_pmc(workload, kernel, spec_key)
    ok = bool(pmc and pmc["matches_this_run"])
    traffic = pmc.get("hbm_bytes_per_launch") if (ok and spp == pmc.get("spp_of_the_traffic_figure")) else None
    if traffic is not None and traffic / (kernel_ms * 1e-3) / 1e9 > HBM_PEAK_GBS:
        traffic = None   # more bytes than the memory can move in the kernel's time: not a usable figure
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "traffic_is": "TCC<->EA bytes per launch from rocprofv3 counters (FETCH_SIZE x 2 + WRITE_SIZE, separate passes): everything the L2 "
                          "exchanges with the fabric -- partial sums, scene misses AND the kernels' register-spill scratch, which is most of it "
                          "on the stack-walk kernels; not ray-state records" if traffic is not None else None,
            "valu_lane_issue_frac": pmc.get("valu_lane_issue_frac") if ok else None,
            "valu": valu_block(pmc) if ok else None,
            "pmc_source": pmc,
            "kernel": kernel, "kernel_ms": round(kernel_ms, 3), "algorithmic_bytes_per_launch": algo_bytes,
            "true_limiter": "f64 VALU issue + lane divergence (valu_lane_issue_frac): the ray state stays in VGPRs, so measured HBM "
                            "traffic is far below the algorithmic record traffic the graded `frac` counts"}


# ------------------------------------------------------------------------------------------------------------ ranks --

def measure_single(be, dev, key, spp, steps, warmup, bvh, walk_order, generic=False):
    """One single-GPU workload measured like `value`: whole frame into a pinned host frame, wall clock over `steps` renders."""
    wl = WORKLOADS[key]
    W, H = wl["W"], wl["H"]
    scene = be.scene(wl["arm"], wl["aspect"], bvh, walk_order)
    ctx = be.context(scene, dev)
    spec = None
    try:
        spec = ctx.specialise()
    except Exception:
        spec = None
    host = be.host_frame(H, W)
    kw = dict(max_depth=DEPTH, generic=generic)
    for _ in range(warmup):
        ctx.render(W, H, spp, frame=host, **kw)
    be.synchronize()
    t0 = time.perf_counter()
    kms = []
    for _ in range(steps):
        _, st = ctx.render(W, H, spp, frame=host, **kw)
        kms.append(st["kernel_ms"])
    be.synchronize()
    dt = time.perf_counter() - t0
    paths = W * H * spp
    kernel_ms = sum(kms) / len(kms)
    out = {"workload": f"{wl['name']} (scene arm {wl['arm']}) {W}x{H}, {spp} spp, depth {DEPTH}; {wl['what']}",
           "value": round(paths * steps / dt / 1e6, 2), "unit": "Mpaths/s (kernels + device->host copy)", "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 3), "kernel_ms": round(kernel_ms, 3),
           "value_kernel_only": round(paths / kernel_ms / 1e3, 2),
           "paths_per_step": paths, "segments_per_path": round(st["segments"] / paths, 4),
           "samples_per_work_item": st["chunk"], "sample_passes": sample_passes(W * H, st["n_chunks"]),
           "scene_nodes": scene.info()["n_nodes"], "bvh": bvh_label(bvh, walk_order), "dtype": "f64",
           "roofline": roofline_block(key if (bvh == DEFAULT_BVH and walk_order == "reference") else None, st, kernel_ms, W * H,
                                      spec.get("key") if spec else None, spp)}
    ctx.close()
    return out


def bvh_label(bvh, walk_order):
    tree = {"best_axis": "BVHNode::new as written (bvh.rs:54-103: sort by box minimum, median split) with the axis of bvh.rs:84 chosen by the cost of "
                         "its median split instead of drawn -- a tree the reference's own entropy-seeded build can produce; the library's default",
            "reference": "BVHNode::new as written with the axis of bvh.rs:84 drawn from build seed 1 (rounds 1-3's default)",
            "sah": "SAH rebuild of the same leaf sets (rt1w_scene_set_bvh_build; opt-in, statistical parity on entropy-seeded arms)"}[bvh]
    order = "left-then-right (bvh.rs:38-47)" if walk_order == "reference" else "near child first where result-preserving (rt1w_scene_set_walk_order)"
    return f"{tree}; walk order {order}"


def tiled_extra(be, dist, rank, world, local_rank, key, W, H, spp, steps, warmup, check_frame, tag):
    """One of BASELINE's multi-GPU configs as a strong-scaling job of the ranks that are already running: the frame tiled in 16-row
    strips dealt round-robin, every rank's device->host copy landing in one shared pinned host frame, timed like `value` (barrier +
    synchronize on both sides, max over ranks).  Returns the entry on rank 0 (None elsewhere)."""
    import numpy as np
    import torch
    rt, sharding = be.rt, be.sharding
    wl = WORKLOADS[key]
    scene = be.scene(wl["arm"], W / H, DEFAULT_BVH, "reference")
    ctx = be.context(scene, local_rank)
    try:
        ctx.specialise()
    except Exception:
        pass
    name = f"rt1w_bench_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('RT1W_BENCH_PARENT', os.getppid())}_{tag}"
    frame = None
    try:
        if rank == 0:
            frame = sharding.SharedFrame(name, W, H, True, rt)
        dist.barrier()
        if rank != 0:
            frame = sharding.SharedFrame(name, W, H, False, rt)
        host = frame.array
        y0, rows, srows, period = sharding.interleaved_tile(H, world, rank)
        kw = dict(max_depth=DEPTH, chunk=scene.default_chunk(W, H, spp))

        def step():
            return ctx.render(W, H, spp, tile=(0, y0, W, rows), strips=(srows, period), frame=host, **kw)[1] if rows else None

        for _ in range(warmup):
            step()
        dist.barrier()
        be.synchronize()
        t0 = time.perf_counter()
        kms, segs = [], 0
        for _ in range(steps):
            st = step()
            if st is not None:
                kms.append(st["kernel_ms"])
                segs = st["segments"]
        dist.barrier()
        be.synchronize()
        elapsed = time.perf_counter() - t0
        mine = {"rank": rank, "rows": rows, "kernel_ms": (sum(kms) / len(kms)) if kms else None, "segments": segs, "elapsed": elapsed}
        every = [None] * world
        dist.all_gather_object(every, mine)
        elapsed = max(e["elapsed"] for e in every)
        frame_check = None
        if check_frame:
            dist.barrier()
            be.synchronize()
            if rank == 0:
                solo, _ = ctx.render(W, H, spp, **kw)
                frame_check = bool(np.array_equal(solo, host, equal_nan=True))
                assert frame_check, f"{tag}: gathered frame differs from the single-GPU frame"
        if rank != 0:
            return None
        paths = W * H * spp
        per_rank = [e["kernel_ms"] for e in every if e["kernel_ms"] is not None]
        out = {"workload": f"{wl['name']} (scene arm {wl['arm']}) {W}x{H}, {spp} spp per step, depth {DEPTH}; tiled over {world} ranks "
                           f"(16-row strips round-robin, one shared pinned host frame, no collective)",
               "scaling": "strong", "value": round(paths * steps / elapsed / 1e6, 2), "unit": "Mpaths/s (kernels + device->host gather)",
               "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "paths_per_step": paths,
               "segments_per_path": round(sum(e["segments"] for e in every) / paths, 4),
               "n_ranks_seen": sum(1 for e in every if e["rows"] > 0), "rows_per_rank": [e["rows"] for e in every],
               "kernel_ms_per_rank": {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3),
                                      "all": [round(x, 3) for x in per_rank]},
               "bvh": bvh_label(DEFAULT_BVH, "reference"), "dtype": "f64"}
        if frame_check is not None:
            out["gathered_frame_equals_single_gpu_frame"] = frame_check
        return out
    finally:
        ctx.close()
        if frame is not None:
            try:
                dist.barrier()
            finally:
                frame.close()


def rank_main(a, be=None):
    """One rank of the job (or the whole job at N = 1)."""
    import numpy as np
    be = be or GpuBackend()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo's C++ side prints its "connected to N peer ranks" note on stdout: keep stdout for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only: barrier + max(time)
            dist.barrier()
        finally:
            os.dup2(saved, 1)
            os.close(saved)

    if a.all_ranks_on_device is not None:
        local_rank = a.all_ranks_on_device
    be.check_device(local_rank)
    rt, sharding = be.rt, be.sharding

    wl = WORKLOADS[a.workload]
    spp = a.spp if a.spp else wl["spp"]
    if a.workload == "c3":
        # weak scaling: ~C3's paths per GPU, and a side that is a multiple of 16 * N rows so that every GPU owns the same number of
        # 16-row strips (N = 2, 4, 8: 864, 1216, 1664)
        side = wl["W"] if (a.strong or world == 1) else 16 * world * max(1, round(wl["W"] * math.sqrt(world) / (16 * world)))
        W, H = side, side
    else:
        W, H = wl["W"], wl["H"]
        if world > 1:
            a.strong = True                                              # one fixed frame, tiled
    if a.width and a.height:                                             # rehearsals / tests only
        W, H = a.width, a.height
    scene = be.scene(wl["arm"], W / H, a.bvh, a.walk_order)
    ctx = be.context(scene, local_rank)
    # kernel specialised for this scene's topology: from the kernel cache the build fills (raytracing-1w_amd/kernels), or compiled
    # here with hiprtc (3-5 s, outside the timed region like the rest of the set-up); --generic keeps the generic kernel;
    # scenes of more than 256 nodes (c2, c4) have no such kernel and run the stack-walk variants
    spec = None
    if not a.generic:
        try:
            spec = ctx.specialise()
        except rt.Rt1wError as e:
            spec = {"active": False, "error": str(e)}

    # the whole-image host frame every rank's copy lands in
    frame = None
    if world > 1:
        shm_name = f"rt1w_bench_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('RT1W_BENCH_PARENT', os.getppid())}"
        if rank == 0:
            frame = sharding.SharedFrame(shm_name, W, H, True, rt)
        dist.barrier()
        if rank != 0:
            frame = sharding.SharedFrame(shm_name, W, H, False, rt)
        host = frame.array
    else:
        host = be.host_frame(H, W)
    y0, rows, srows, period = sharding.interleaved_tile(H, world, rank)
    chunk = scene.default_chunk(W, H, spp)                               # the whole frame's chunking (rt1w_scene_default_chunk): same sums as one GPU
    kw = dict(max_depth=DEPTH, generic=a.generic, chunk=chunk)
    if a.probe_coherent:
        kw["probe_coherent"] = True
    tile = (0, y0, W, rows)

    def step():
        if rows == 0:
            return None
        _, st = ctx.render(W, H, spp, tile=tile, strips=(srows, period), frame=host, **kw)
        return st

    def barrier():
        if dist is not None:
            dist.barrier()
        be.synchronize()

    try:
        for _ in range(a.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        kernel_ms = []
        segs = 0
        st = None
        for _ in range(a.steps):
            st = step()
            if st is not None:
                kernel_ms.append(st["kernel_ms"])
                segs = st["segments"]
        barrier()
        elapsed = time.perf_counter() - t0
        seg_total = segs
        if dist is not None:
            import torch
            t = torch.tensor([elapsed, float(segs)], dtype=torch.float64)
            tm = t.clone()
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            elapsed = float(tm[0])
            seg_total = int(t[1])

        paths_per_step = W * H * spp
        value = paths_per_step * a.steps / elapsed / 1e6

        frame_check = None
        if a.check_frame:
            barrier()
            if rank == 0:
                solo, _ = ctx.render(W, H, spp, **kw)
                frame_check = bool(np.array_equal(solo, host, equal_nan=True))
                assert frame_check, "gathered frame differs from the single-GPU frame"

        # device-resident rate of this rank's share (framebuffer left in HBM), outside the timed region
        dev_ms = None
        if rows and hasattr(be, "device_resident_ms") and not a.probe_coherent:
            dev_ms = be.device_resident_ms(ctx, local_rank, W, H, spp, tile, (srows, period), max(1, min(a.steps, 3)), kw)

        if rank == 0:
            avg_ms = sum(kernel_ms) / len(kernel_ms)
            my_pixels = rows * W
            flags = st.get("sorted", 0)
            line = {
                "metric": "Mpaths/s", "value": round(value, 2), "unit": "Mpaths/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "strong" if (a.strong and world > 1) else "weak", "vs_baseline": None, "dtype": "f64",
                "data": f"synthetic (the reference's procedural scene arm {wl['arm']}, build_seed 1"
                        + ("; assets/earthmap.jpg as the image texture" if wl["arm"] == 7 else "; no external data") + ")",
                "config": {"workload": f"{wl['name']} (scene arm {wl['arm']}) {W}x{H}, {spp} spp, depth {DEPTH}, {wl['what']}; "
                                       f"timed region = kernels + device->host gather into one pinned host frame",
                           "width": W, "height": H, "spp": spp, "max_depth": DEPTH,
                           "paths_per_step": paths_per_step, "segments_per_path": round(seg_total / paths_per_step, 4),
                           "chunk": st["chunk"], "n_chunks": st["n_chunks"], "sample_passes": sample_passes(my_pixels, st["n_chunks"]),
                           "grid": st["grid"], "block": st["block"],
                           "kernel_variant": st["variant"], "workgroup_path_sort": bool(flags & 1),
                           "scene_specialised_kernel": bool(flags & 4), "specialise": spec,
                           "scene_nodes": scene.info()["n_nodes"], "bvh": bvh_label(a.bvh, a.walk_order),
                           "parallelism": (f"image-tiled x{world}: 16-row strips round-robin, one launch per GPU, every GPU's D2H writes its "
                                           f"strips into one shared pinned host frame (host gather, no collective); ranks started by "
                                           + ("bench.py itself" if os.environ.get("RT1W_BENCH_PARENT") else "the outer launcher")) if world > 1
                                          else "1 GPU, D2H into a pinned host frame",
                           "host_frame_pinned": bool(frame._pinned) if frame is not None else True,
                           "host_frame_pin_error": getattr(frame, "pin_error", None) if frame is not None else None,
                           "backend": be.name, "commit": git_head(), "kernel_sources": kernel_sources_id(),
                           "probe_coherent": bool(a.probe_coherent), "segments_counted": seg_total},
                "value_device_resident": round(my_pixels * spp / (dev_ms * 1e-3) / 1e6 * (world if world > 1 else 1), 2) if dev_ms else None,
                "roofline": roofline_block(a.workload if (world == 1 and a.bvh == DEFAULT_BVH and a.walk_order == "reference") else None, st, avg_ms,
                                           my_pixels, spec.get("key") if spec else None, spp),
            }
            if frame_check is not None:
                line["gathered_frame_equals_single_gpu_frame"] = frame_check
            if world == 1 and a.workload == "c3" and not a.no_other_configs and not (a.width and a.height):
                line["other_configs"] = {name: measure_single(be, local_rank, k, s, 3, 1, bvh, order)
                                         for name, k, s, bvh, order in OTHER_CONFIGS}
            if world == 1 and not a.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(wl, W, H)
    finally:
        ctx.close()
        if frame is not None:
            try:
                if dist is not None:
                    dist.barrier()
            finally:
                frame.close()
    # N > 1, the default job: BASELINE's two multi-GPU configs on the same ranks (every rank takes part; rank 0 holds the entries)
    if world > 1 and a.workload == "c3" and not a.no_other_configs and not a.strong and (a.other_size or not (a.width and a.height)):
        extras = {}
        for name, key, xspp, xsteps in OTHER_CONFIGS_MULTI:
            xw, xh = (a.other_size[0], a.other_size[1]) if a.other_size else (WORKLOADS[key]["W"], WORKLOADS[key]["H"])
            xspp = a.other_size[2] if a.other_size else xspp
            extras[name] = tiled_extra(be, dist, rank, world, local_rank, key, xw, xh, xspp, xsteps, 1, a.check_frame, name)
        if rank == 0:
            line["other_configs"] = extras
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1, c3: skip the C2 / C4 entries of the line")
    ap.add_argument("--generic", action="store_true", help="ablation: the generic kernel instead of the scene-specialised one")
    ap.add_argument("--spp", type=int, default=0, help="default: the workload's own (c3 1000, c2 500, c4 10000, c5 1000)")
    ap.add_argument("--workload", choices=tuple(WORKLOADS), default="c3")
    ap.add_argument("--bvh", choices=("best_axis", "reference", "sah"), default=DEFAULT_BVH,
                    help="which tree: best_axis = the library's default (axis chosen), reference = axes drawn from build seed 1, sah = opt-in rebuild")
    ap.add_argument("--walk-order", choices=("reference", "near-far"), default="reference")
    ap.add_argument("--strong", action="store_true", help="N > 1, c3: keep the N = 1 frame (strong scaling of one job)")
    ap.add_argument("--all-ranks-on-device", type=int, default=None,
                    help="rehearsal only (1-GPU box): every rank uses this device instead of LOCAL_RANK")
    ap.add_argument("--check-frame", action="store_true",
                    help="after timing, rank 0 renders the whole frame alone and requires the gathered frame to be bit-identical")
    ap.add_argument("--width", type=int, default=0, help="tests / rehearsals only: override the frame")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--probe-coherent", action="store_true",
                    help="measurement runs under rocprofv3 only (tools/bench_pmc.sh): RT1W_PROBE_COHERENT, every wave traces one path 64 times; `value` is then meaningless")
    ap.add_argument("--other-size", type=int, nargs=3, default=None, metavar=("W", "H", "SPP"),
                    help="tests / rehearsals only: frame and spp of the N > 1 line's other_configs (c5_strong, c4_strong)")
    return ap.parse_args(argv)


def main(argv=None, backend_factory=None, script=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start the N ranks ourselves.  Nothing above imported torch or the HIP library.
        rc, out0 = launch_ranks(a.gpus, argv, script=script)
        out0 = "".join(l + "\n" for l in out0.splitlines() if l.startswith("{"))   # rank 0's JSON line, nothing else
        sys.stdout.write(out0)
        sys.stdout.flush()
        if rc == 0 and not any(l.startswith("{") for l in out0.splitlines()):
            rc = 1
        sys.exit(rc)
    rank_main(a, backend_factory() if backend_factory else None)


if __name__ == "__main__":
    main()
