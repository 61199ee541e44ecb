#!/bin/bash
# counters of W4 against W0c (tools/w4_lab.py --pmc): VALU busy, lanes per VALU instruction, waiting, LDS conflicts
# usage (GPU box, repo root): tools/w4_pmc.sh <w4_lab.py args ...>
OUT=$GRAFT_REPO_ROOT/gpurun_out/w4pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
            "GRBM_GUI_ACTIVE SQ_INSTS_WAVE32_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 tools/w4_lab.py "$@" --pmc > $OUT/p$i.log 2>&1) || echo "pass $i failed"
done
export OUT
python3 - <<'PY'
import csv, glob, collections, os
OUT = os.environ["OUT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(OUT + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = "W4" if "lab_trace_w4" in r["Kernel_Name"] else ("W0c" if "lab_trace_w0" in r["Kernel_Name"] else None)
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(OUT + "/summary.txt", "w") as o:
    for k, d in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        for c, v in sorted(m.items()):
            o.write(f"{k:4s} {c:28s} {v:.6g}\n")
        if "GRBM_GUI_ACTIVE" in m and "SQ_ACTIVE_INST_VALU" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            busy = m["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0)
            lanes = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])
            o.write(f"{k:4s} VALU busy {busy:.3f}  lanes per VALU instruction {lanes:.3f}  waiting {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}  "
                    f"VALU insts {m['SQ_INSTS_VALU']:.4g} SALU {m['SQ_INSTS_SALU']:.4g} LDS {m['SQ_INSTS_LDS']:.4g}  LDS bank conflict cycles / LDS active {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_ACTIVE_INST_LDS', 1), 1):.2f}\n")
print(open(OUT + "/summary.txt").read())
PY
