#!/bin/bash
# Round-2 profile collection (run on the GPU box from the repo root): kernel-trace stats of bench.py, PMC passes of the C3 kernel,
# kernel traces + PMC of the big scenes (megakernel and wavefront), kernel traces of the f32 and reference-stream modes.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
echo "== bench kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_stats.json 2> $O/bench_stats.err
echo "== bench PMC"
tools/prof_pmc.sh r02 --steps 2 --warmup 0 > $O/bench_pmc.log 2>&1
cp gpurun_out/pmc_r02/summary.txt $O/bench_pmc_summary.txt
echo "== big scenes: megakernel kernel trace + PMC"
for arm in 0 7; do
  if [ $arm = 0 ]; then A="0 1200 800 24"; else A="7 800 800 16"; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mega_$arm -- python3 tools/wf_one.py $A mega > $O/mega_$arm.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wf_$arm -- python3 tools/wf_one.py $A > $O/wf_$arm.log 2>&1
  tools/pmc_cmd.sh r02mega$arm "rt_render_kernel" tools/wf_one.py $A mega > $O/pmc_mega_$arm.log 2>&1
  tools/pmc_cmd.sh r02wf$arm "wf_trace_lds|wf_trace|wf_shade|wf_finish" tools/wf_one.py $A > $O/pmc_wf_$arm.log 2>&1
  cp gpurun_out/pmcc_r02mega$arm/summary.txt $O/pmc_mega_${arm}_summary.txt
  cp gpurun_out/pmcc_r02wf$arm/summary.txt $O/pmc_wf_${arm}_summary.txt
done
echo "== f32 / reference stream kernel traces (Cornell)"
cat > /tmp/modes.py <<'PY'
import importlib, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
rt = importlib.import_module("raytracing-1w_amd")
ctx = rt.Context(rt.Scene.reference(5), 0)
for kw in ({}, dict(f32=True), dict(f32=True, generic=True), dict(reference_stream=True)):
    for _ in range(2):
        g, s = ctx.render(600, 600, 200 if not kw.get("reference_stream") else 100, **kw)
    print(kw, round(s["paths"] / s["kernel_ms"] / 1e3, 1), "Mpaths/s flags", s["sorted"], flush=True)
PY
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/modes -- python3 /tmp/modes.py > $O/modes.log 2>&1
find $O -name "*kernel_stats.csv" | head -20
echo done
