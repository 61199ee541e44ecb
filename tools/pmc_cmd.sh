#!/bin/bash
# PMC passes (separate rocprofv3 --pmc runs, csv) for any python command; counters summed per kernel-name pattern.
# usage: tools/pmc_cmd.sh <tag> <kernel name regex> <python script + args ...>
TAG=$1; PAT=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT" \
            "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && timeout -k 10 240 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1) || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
pat = re.compile(r"$PAT")
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = pat.search(r["Kernel_Name"])
        if m:
            agg[m.group(0)][r["Counter_Name"]] += float(r["Counter_Value"])
with open("$OUT/summary.txt", "w") as o:
    for kn, d in sorted(agg.items()):
        for k, v in sorted(d.items()):
            line = f"{kn:24s} {k:32s} sum={v:.6g}"
            print(line); o.write(line + "\n")
PY
