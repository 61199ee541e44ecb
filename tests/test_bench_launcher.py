"""bench.py as the driver runs it: `python bench.py --gpus N` with NO launcher around it must start its own N ranks
(it died at the WORLD_SIZE assert).  The N-rank control flow -- rendezvous on 127.0.0.1, interleaved tiles, one
shared host frame, max-over-ranks clock, --check-frame -- is exercised here on the CPU with the CPU build of the kernel core
injected as the renderer (tests/bench_cpu_ranks.py); the GPU tier runs the real thing (test_gpu_parity.py)."""
import json
import os
import subprocess
import sys

import orc

ROOT = orc.ROOT


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_bench_starts_its_own_two_ranks_and_gathers_one_frame(rt):
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_cpu_ranks.py"), "--gpus", "2", "--width", "48", "--height", "48",
           "--spp", "4", "--steps", "1", "--warmup", "0", "--check-frame", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["backend"] == "cpu-core (test)"
    assert line["gathered_frame_equals_single_gpu_frame"] is True
    assert "bench.py itself" in line["config"]["parallelism"]
    assert line["config"]["width"] == 48 and line["config"]["paths_per_step"] == 48 * 48 * 4
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "roofline"):
        assert k in line
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rt1w_bench_")]   # the shared frame was removed


def test_bench_three_ranks_with_a_ragged_last_strip(rt):
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_cpu_ranks.py"), "--gpus", "3", "--width", "40", "--height", "56",
           "--spp", "2", "--steps", "1", "--warmup", "1", "--check-frame", "--no-cpu-baseline", "--workload", "c2"]
    p = subprocess.run(cmd, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 3 and line["scaling"] == "strong" and line["gathered_frame_equals_single_gpu_frame"] is True


def test_default_multi_rank_line_carries_baselines_multi_gpu_configs(rt):
    """`python bench.py --gpus N` with no --workload is the one command the driver runs on an 8-GPU node: besides the c3 weak line it
    must measure BASELINE's two multi-GPU configs -- C5 (4K Cornell) and C4 (final_scene) tiled over the same ranks, strong scaling --
    as `other_configs` (here at test size, CPU build of the core injected): D2H-inclusive rate, per-rank kernel ms, ranks seen,
    and the gathered frame equal to the single-rank frame."""
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_cpu_ranks.py"), "--gpus", "2", "--width", "48", "--height", "48",
           "--spp", "2", "--steps", "1", "--warmup", "0", "--check-frame", "--no-cpu-baseline", "--other-size", "40", "40", "2"]
    p = subprocess.run(cmd, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and set(line["other_configs"]) == {"c5_strong", "c4_strong"}
    for name, arm in (("c5_strong", "scene arm 5"), ("c4_strong", "scene arm 7")):
        e = line["other_configs"][name]
        assert arm in e["workload"] and e["scaling"] == "strong" and e["n_ranks_seen"] == 2 and sum(e["rows_per_rank"]) == 40
        assert e["gathered_frame_equals_single_gpu_frame"] is True
        assert e["value"] > 0 and e["paths_per_step"] == 40 * 40 * 2 and e["segments_per_path"] > 1.0
        assert e["kernel_ms_per_rank"]["min"] <= e["kernel_ms_per_rank"]["max"] and len(e["kernel_ms_per_rank"]["all"]) == 2
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rt1w_bench_")]
    # --strong and the explicit workloads are one job each: no extra entries
    p = subprocess.run(cmd[:-4] + ["--workload", "c2"], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and "other_configs" not in json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])


def test_a_failing_rank_is_heard_and_ends_the_job(rt):
    """a rank other than 0 that dies (no device, an exception) must end the whole job with a non-zero exit code, and what it printed --
    stdout included -- must reach the parent's stderr; no JSON line, no shared frame left behind, no hang"""
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_cpu_ranks.py"), "--gpus", "3", "--width", "32", "--height", "48",
           "--spp", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-other-configs"]
    env = dict(_clean_env(), RT1W_TEST_FAIL_RANK="2")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = p.stderr.decode()
    assert p.returncode != 0
    assert "injected failure of rank 2" in err and "stdout of the failing rank 2" in err and "rank 2 exited" in err
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rt1w_bench_")]


def test_launcher_times_out_instead_of_hanging(rt):
    sys.path.insert(0, ROOT)
    import bench
    script = os.path.join(ROOT, "tests", "_sleep_rank.py")
    open(script, "w").write("import sys, time\nprint('x' * 200000, flush=True)\ntime.sleep(60)\n")   # > a pipe's buffer on stdout, then stuck
    try:
        rc, out0 = bench.launch_ranks(2, [], script=script, timeout=3.0)
    finally:
        os.remove(script)
    assert rc == 124 and len(out0) >= 200000


def test_product_bench_fails_loudly_without_a_gpu(rt):
    """the real bench.py has no CPU path: on a box without a GPU every rank dies on the device check and the parent's exit
    code says so (no JSON line)"""
    if rt.device_count() > 0:
        import pytest
        pytest.skip("this check is for the CPU-only container")
    for gpus in ("1", "2"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--steps", "1", "--warmup", "0"],
                           env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode != 0
        assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_launcher_parent_never_touches_the_gpu_stack():
    """the parent process of `--gpus N` must not import torch or load librt1w.so before it forks its ranks"""
    code = ("import sys; sys.argv=['bench.py','--gpus','2']; import bench; "
            "bench.launch_ranks = lambda n, argv, script=None, **kw: (print('MODS', 'torch' in sys.modules, "
            "any('raytracing' in m for m in sys.modules)) or (0, '{}\\n')); bench.main()")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    assert "MODS False False" in p.stdout.decode()


def test_stored_pmc_is_attached_only_to_the_kernel_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic / valu_lane_issue_frac come from a stored rocprofv3 --pmc measurement (profiles/pmc_summary.json): it must
    name the same kernel, the same specialisation key and the same kernel source text as the run it is attached to"""
    sys.path.insert(0, ROOT)
    import bench
    ident = bench.kernel_sources_id()
    assert ident and len(ident) == 12
    (tmp_path / "profiles").mkdir()
    ent = {"kernel": "rt_jit_sorted", "specialise_key": "abc", "kernel_sources": ident, "spp_of_the_traffic_figure": 1000,
           "valu_lane_issue_frac": 0.4, "hbm_bytes_per_launch": 7}
    (tmp_path / "profiles" / "pmc_summary.json").write_text(json.dumps({"c3": ent, "c4": dict(ent, kernel_sources="0" * 12)}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_sources_id", lambda: ident)
    assert bench.stored_pmc("c3", "rt_jit_sorted", "abc")["matches_this_run"] is True
    assert bench.stored_pmc("c3", "rt_jit_sorted", "other key")["matches_this_run"] is False
    assert bench.stored_pmc("c3", "rt_render_kernel<V0>", "abc")["matches_this_run"] is False
    assert bench.stored_pmc("c4", "rt_jit_sorted", "abc")["matches_this_run"] is False    # measured on other kernel sources
    assert bench.stored_pmc("c2", "rt_jit_sorted", "abc") is None
    st = {"segments": 1000, "sorted": 4, "variant": 0}
    r = bench.roofline_block("c3", st, 1.0, 100, "abc", spp=1000)
    assert r["valu_lane_issue_frac"] == 0.4 and r["traffic"] == 7
    r = bench.roofline_block("c3", st, 1.0, 100, "abc", spp=10)                            # another spp: no traffic figure
    assert r["valu_lane_issue_frac"] == 0.4 and r["traffic"] is None
    r = bench.roofline_block("c4", st, 1.0, 100, "abc", spp=1000)
    assert r["valu_lane_issue_frac"] is None and r["traffic"] is None


def test_kernel_names_follow_the_stats_flags():
    """bench.py names the kernel a render ran from rt1w_stats (variant + the bits of `sorted`, include/rt1w.h): the stored PMC is
    matched by that name, so the kernels with the slice-end reordering (bit 9) must not be taken for the plain ones"""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.kernel_name({"sorted": 4 | 1, "variant": 0}) == "rt_jit_sorted"
    assert bench.kernel_name({"sorted": 1, "variant": 1}) == "rt_render_kernel_sorted<V1>"
    assert bench.kernel_name({"sorted": 0, "variant": 2}) == "rt_render_kernel<V2>"
    assert bench.kernel_name({"sorted": 128, "variant": 5}) == "rt_render_kernel_pw<V5>"
    assert bench.kernel_name({"sorted": 128 | 512, "variant": 5}) == "rt_render_kernel_pw_ss<V5>"
    assert bench.kernel_name({"sorted": 256, "variant": 3}) == "rt_render_kernel<V3, sphere media>"
    assert bench.kernel_name({"sorted": 256 | 512, "variant": 4}) == "rt_render_kernel_ss<V4, sphere media>"
    assert bench.kernel_name({"sorted": 512, "variant": 2}) == "rt_render_kernel_ss<V2>"
