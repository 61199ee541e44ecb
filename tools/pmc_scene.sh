#!/bin/bash
# PMC passes for one scene arm via tools/quick_bench.py: tools/pmc_scene.sh <tag> <arm>
TAG=$1; ARM=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/quick_bench.py $ARM > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "rt_render_kernel" in r["Kernel_Name"] or "rt_jit_sorted" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for k, v in sorted(agg.items()):
        v = v[1:] if len(v) > 1 else v   # drop the tiny warm-up launch
        line = f"{k:28s} launches={len(v):3d} mean_per_launch={sum(v)/len(v):.6g}"
        print(line); o.write(line + "\n")
PY
