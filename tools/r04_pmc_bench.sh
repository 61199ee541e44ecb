R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_prof; mkdir -p $O; cd $R
tools/bench_pmc.sh > $O/bench_pmc.log 2>&1; cp gpurun_out/bench_pmc/pmc_summary.json $O/; cp gpurun_out/bench_pmc/pmc_summary.json profiles/pmc_summary.json
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
(cd /tmp && export TMPDIR=/tmp && cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats2 -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_stats.err)
python3 -c "
import json
l=json.loads([x for x in open('$O/bench.json') if x.startswith('{')][0])
print(l['value'], l['roofline']['valu'], l['roofline']['traffic'], {k:(v['value'], v['roofline'].get('valu',{}) and v['roofline']['valu'].get('efficiency'), v['roofline'].get('traffic')) for k,v in l['other_configs'].items()})
"
