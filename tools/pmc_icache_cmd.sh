#!/bin/bash
# instruction-cache / instruction-fetch counters for any python command: tools/pmc_icache_cmd.sh <tag> <script + args ...>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmci_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_BUSY_CYCLES"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && timeout -k 10 240 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1) || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "rt_render_kernel" in r["Kernel_Name"] or "rt_jit_sorted" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print(f"{k:32s} sum={v:.6g}")
PY
