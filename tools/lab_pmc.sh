#!/bin/bash
# PMC passes (separate rocprofv3 --pmc runs) over the walk lab's trace kernels: what bounds the walk -- VALU issue, the L1's tag
# look-ups (TCP / TA) or latency?   usage: tools/lab_pmc.sh <tag> <walk_lab.py args ...>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/labpmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
            "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 tools/walk_lab.py "$@" > $OUT/p$i.log 2>&1) || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"lab_trace_(w0q|w0|w2|w3<[^>]*>|w1<[^>]*>)", r["Kernel_Name"])
        k = ("lab_trace_" + m.group(1).replace(" ", "")) if m else None
        if k == "lab_trace_w0" and re.search(r"lab_trace_w0<.*, 4>", r["Kernel_Name"]): k = "lab_trace_w0b"
        if k and k.startswith("lab_trace_w3"): k = "lab_trace_w3" + ("" if re.search(r"lab_trace_w3<.*, 4>", r["Kernel_Name"]) else "_no_box_steps")
        if k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
with open("$OUT/summary.txt", "w") as o:
    for kn, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            line = f"{kn:18s} {c:44s} sum={v:.6g} dispatches={n[kn][c]}"
            print(line); o.write(line + "\n")
PY
