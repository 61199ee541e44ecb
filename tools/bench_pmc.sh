#!/bin/bash
# PMC of bench.py's dominant kernel for the workloads c3, c2, c4 (separate rocprofv3 --pmc passes; FETCH_SIZE and WRITE_SIZE cannot
# share one) -> gpurun_out/bench_pmc/pmc_summary.json, to be copied to profiles/pmc_summary.json: the stored measurement bench.py
# attaches to its roofline block (traffic, valu_lane_issue_frac) when the kernel that runs is the kernel that was measured.
# usage: tools/bench_pmc.sh            (run on the GPU box from the repo root)
OUT=$GRAFT_REPO_ROOT/gpurun_out/bench_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in c3 c2 c4; do
  case $wl in c3) ARGS="--workload c3";; c2) ARGS="--workload c2";; c4) ARGS="--workload c4 --spp 400";; esac
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
              "SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
              "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    (cd $GRAFT_REPO_ROOT && timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/$wl/p$i -- python3 bench.py $ARGS --steps 2 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/$wl.p$i.log 2>&1) || echo "$wl pass $i failed"
  done
  # the coherent probe (RT1W_PROBE_COHERENT: every wave traces one path 64 times): the necessary VALU instructions per segment
  (cd $GRAFT_REPO_ROOT && timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/$wl/probe -- python3 bench.py $ARGS --probe-coherent --steps 2 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/$wl.probe.log 2>&1) || echo "$wl probe pass failed"
done
export OUT
python3 - <<'PY'
import csv, glob, collections, json, os, re
OUT = os.environ["OUT"]
out = {}
for wl, spp in (("c3", 1000), ("c2", 500), ("c4", 400)):
    agg = collections.defaultdict(list)
    names = collections.Counter()
    probe = collections.defaultdict(list)
    for f in sorted(glob.glob(OUT + "/%s/probe/**/*counter_collection.csv" % wl, recursive=True)):
        for r in csv.DictReader(open(f)):
            if "rt_jit_sorted" in r["Kernel_Name"] or "rt_render_kernel" in r["Kernel_Name"]:
                probe[r["Counter_Name"]].append(float(r["Counter_Value"]))
    pm = {k: sum(v) / len(v) for k, v in probe.items()}
    pline = {}
    try:
        for l in open(OUT + "/%s.probe.log" % wl):
            if l.startswith("{"):
                pline = json.loads(l)
    except OSError:
        pass
    for f in sorted(glob.glob(OUT + "/%s/p[0-9]*/**/*counter_collection.csv" % wl, recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "rt_jit_sorted" in k or "rt_render_kernel" in k:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"])); names[k] += 1
    if not agg:
        continue
    line = {}
    for l in open(OUT + "/%s.p1.log" % wl):
        if l.startswith("{"):
            line = json.loads(l)
    # counters PER RENDER: the mean over the kernel's launches x the launches one render takes (sample passes, bench.py: config.sample_passes)
    passes = line.get("config", {}).get("sample_passes", 1) or 1
    m = {k: sum(v) / len(v) * (passes if not k.startswith("GRBM") else passes) for k, v in agg.items()}
    ppasses = pline.get("config", {}).get("sample_passes", 1) or 1
    pm = {k: v * ppasses for k, v in pm.items()}
    kern = line.get("roofline", {}).get("kernel")
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
    busy = m["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0)  # 256 CUs x 4 SIMDs
    lanes = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])
    fetch_kb, write_kb = m.get("FETCH_SIZE", 0.0), m.get("WRITE_SIZE", 0.0)
    out[wl] = {"kernel": kern, "specialise_key": (line.get("config", {}).get("specialise") or {}).get("key") if kern == "rt_jit_sorted" else None,
               "spp_of_the_traffic_figure": spp, "sample_passes": passes, "commit": line.get("config", {}).get("commit"),
               "kernel_sources": line.get("config", {}).get("kernel_sources"),
               "segments_counted": line.get("config", {}).get("segments_counted"),
               "coherent_probe": ({"SQ_INSTS_VALU_per_launch": pm["SQ_INSTS_VALU"], "SQ_INSTS_SALU_per_launch": pm.get("SQ_INSTS_SALU"),
                                   "lane_utilisation": round(pm["SQ_THREAD_CYCLES_VALU"] / (64.0 * pm["SQ_ACTIVE_INST_VALU"]), 4) if pm.get("SQ_ACTIVE_INST_VALU") else None,
                                   "segments_counted": pline.get("config", {}).get("segments_counted"), "kernel": pline.get("roofline", {}).get("kernel")}
                                  if pm.get("SQ_INSTS_VALU") else None),
               "valu_busy_frac": round(busy, 4), "lane_utilisation": round(lanes, 4), "valu_lane_issue_frac": round(busy * lanes, 4),
               "wave_cycles_waiting_frac": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4),
               "SQ_INSTS_VALU_per_launch": m["SQ_INSTS_VALU"], "SQ_INSTS_SALU_per_launch": m["SQ_INSTS_SALU"],
               "l1_accesses_per_clk_per_cu": round(m.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / (cyc * 256.0), 3),
               "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb,
               "hbm_bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024.0),
               "correction": "gfx950: FETCH_SIZE tallies 64 B per 128-B request on wide coalesced reads, so it is doubled; WRITE_SIZE as is (MI355X_MICROARCH.md)",
               "source": "tools/bench_pmc.sh: rocprofv3 --pmc passes over bench.py --workload %s%s --steps 2 --warmup 0, means per launch of the dominant kernel x the launches of one render (sample passes): per-render figures" % (wl, " --spp 400" if wl == "c4" else ""),
               "kernel_names_seen": list(names)[:2]}
json.dump(out, open(OUT + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
