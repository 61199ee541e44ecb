#!/bin/bash
# A/B of the specialised Cornell kernel under JIT defines, with the instruction-cache and wait counters of each build:
#   tools/icache_ab.sh "" "-DRT_SHARED_DIV=0" ...      (run on the GPU box from the repo root; each argument = RT1W_JIT_EXTRA_OPTS of one build)
# -> gpurun_out/icache_ab/summary.txt: Mpaths/s and kernel ms (no profiler), then per-launch counter means.
OUT=$GRAFT_REPO_ROOT/gpurun_out/icache_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -f $OUT/counters.txt ] || rocprofv3 -L > $OUT/counters.txt 2>&1
n=0
for opts in "$@"; do
  n=$((n+1))
  export RT1W_JIT_EXTRA_OPTS="$opts"
  (cd $GRAFT_REPO_ROOT && timeout -k 10 200 python3 bench.py --workload c3 --no-cpu-baseline --no-other-configs > $OUT/v$n.json 2> $OUT/v$n.err)
  i=0
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
              "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
              "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
    i=$((i+1))
    (cd $GRAFT_REPO_ROOT && timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/v$n/p$i -- python3 bench.py --workload c3 --steps 2 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/v$n.p$i.log 2>&1) || echo "build $n pass $i failed ($ctrs)"
  done
done
export OUT N=$n
python3 - "$@" <<'PY' > $OUT/summary.txt
import csv, glob, collections, json, os, sys
OUT = os.environ["OUT"]
for n, opts in enumerate(sys.argv[1:], 1):
    line = {}
    try:
        for l in open(OUT + "/v%d.json" % n):
            if l.startswith("{"):
                line = json.loads(l)
    except OSError:
        pass
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(OUT + "/v%d/p*/**/*counter_collection.csv" % n, recursive=True)):
        for r in csv.DictReader(open(f)):
            if "rt_jit_sorted" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    print("build %d: RT1W_JIT_EXTRA_OPTS='%s'  %s Mpaths/s, kernel %s ms" % (n, opts, line.get("value"), line.get("roofline", {}).get("kernel_ms")))
    for k in sorted(m):
        print("    %-32s %.4g" % (k, m[k]))
    if "SQC_ICACHE_REQ" in m and m["SQC_ICACHE_REQ"]:
        print("    icache miss rate %.4f" % (m.get("SQC_ICACHE_MISSES", 0.0) / m["SQC_ICACHE_REQ"]))
    if "SQ_WAVE_CYCLES" in m:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if k in m:
                print("    %s / SQ_WAVE_CYCLES %.4f" % (k, m[k] / m["SQ_WAVE_CYCLES"]))
    if "SQ_WAIT_IFETCH" in m and "SQ_WAVE_CYCLES" not in m:
        pass
PY
cat $OUT/summary.txt
