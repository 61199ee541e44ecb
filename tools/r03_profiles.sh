#!/bin/bash
# Round-3 profile collection (run on the GPU box from the repo root): GPU test log, bench line + kernel-trace stats of the same command
# (C3 and, through other_configs, C2 / C4 at the spp the line reports), stored PMC (tools/bench_pmc.sh), self-launched 2-rank runs,
# all configurations and modes (tools/configs_bench.py).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_prof
mkdir -p $O
cd $R
echo "== gpu tests"; (timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log); tail -2 $O/gpu_tests.log
echo "== bench"; timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err
echo "== bench under rocprofv3 --kernel-trace --stats"
(cd /tmp && export TMPDIR=/tmp && cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_stats.err)
echo "== 2 ranks on one device, started by bench.py itself"
timeout -k 10 300 python3 bench.py --gpus 2 --all-ranks-on-device 0 --check-frame --steps 3 > $O/bench_2rank_c3.json 2> $O/bench_2rank_c3.err
timeout -k 10 300 python3 bench.py --gpus 2 --all-ranks-on-device 0 --check-frame --steps 2 --workload c4 --spp 100 > $O/bench_2rank_c4.json 2> $O/bench_2rank_c4.err
timeout -k 10 300 python3 bench.py --workload c2 --steps 3 > $O/bench_c2.json 2> $O/bench_c2.err
echo "== stored PMC"; tools/bench_pmc.sh > $O/bench_pmc.log 2>&1; cp gpurun_out/bench_pmc/pmc_summary.json $O/ 2>/dev/null
echo "== full sizes"; timeout -k 10 300 python3 tools/full_size.py C2 C4 C5 > $O/full_size.txt 2>&1
echo "== configs"; timeout -k 10 500 python3 tools/configs_bench.py > $O/configs.txt 2>&1
echo "== reordering kernels against the plain ones, soak"; (timeout -k 10 200 python3 tools/ss_ab.py 7 800 800 100; timeout -k 10 200 python3 tools/ss_ab.py 0 1200 800 100; timeout -k 10 300 python3 tools/ss_soak.py 8) > $O/slice_sort.txt 2>&1
find $O -name "*kernel_stats.csv" | head; echo done
