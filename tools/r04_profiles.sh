#!/bin/bash
# Round-4 profile collection (run on the GPU box from the repo root): GPU test log, bench line + kernel-trace stats of the same command
# (C3 and, through other_configs, C2 / C4 on three trees), stored PMC incl. the coherent probe (tools/bench_pmc.sh), self-launched 2- and
# 4-rank runs of the default multi-rank job (c3 weak + c5_strong + c4_strong), full sizes, all configurations and modes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_prof
mkdir -p $O
cd $R
echo "== gpu tests"; (timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log); tail -2 $O/gpu_tests.log
echo "== stored PMC (normal passes + coherent probe)"; tools/bench_pmc.sh > $O/bench_pmc.log 2>&1; cp gpurun_out/bench_pmc/pmc_summary.json $O/ 2>/dev/null; cp gpurun_out/bench_pmc/pmc_summary.json profiles/pmc_summary.json 2>/dev/null
echo "== bench (with the fresh PMC attached)"; timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
echo "== bench under rocprofv3 --kernel-trace --stats"
(cd /tmp && export TMPDIR=/tmp && cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_stats.err)
echo "== 2 and 4 ranks on one device, started by bench.py itself (the default multi-rank job)"
timeout -k 10 400 python3 bench.py --gpus 2 --all-ranks-on-device 0 --check-frame --steps 3 > $O/bench_2rank.json 2> $O/bench_2rank.err
timeout -k 10 400 python3 bench.py --gpus 4 --all-ranks-on-device 0 --check-frame --steps 2 > $O/bench_4rank.json 2> $O/bench_4rank.err
echo "== full sizes"; timeout -k 10 300 python3 tools/full_size.py C2 C4 C5 > $O/full_size.txt 2>&1
echo "== configs"; timeout -k 10 600 python3 tools/configs_bench.py > $O/configs.txt 2>&1
echo "== trees, f32 pair walk, reference stream, probe"; (timeout -k 10 200 python3 tools/tree_ab.py 100; timeout -k 10 300 python3 tools/f32_pw_ab.py 100; timeout -k 10 200 python3 tools/refstream_bench.py 100; timeout -k 10 100 python3 tools/probe_check.py 64) > $O/modes.txt 2>&1
find $O -name "*kernel_stats.csv" | head; echo done
