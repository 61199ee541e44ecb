#!/bin/bash
# a sweep of backend options over the specialised Cornell kernel (tools/jit_opts.py; frames hashed: every build must render the same bits)
# usage: tools/jit_opts_sweep.sh > gpurun_out/jit_opts_sweep.txt   (GPU box, repo root)
python3 tools/jit_opts.py \
  "-mllvm -enable-misched=false" \
  "-mllvm -enable-post-misched=false" \
  "-mllvm -enable-misched=false -mllvm -enable-post-misched=false" \
  "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1" \
  "-mllvm -amdgpu-schedule-metric-bias=0" \
  "-mllvm -amdgpu-schedule-metric-bias=50" \
  "-mllvm -amdgpu-skip-threshold=4" \
  "-mllvm -amdgpu-skip-threshold=24" \
  "-mllvm -amdgpu-skip-threshold=64" \
  "-mllvm -amdgpu-skip-threshold=1000" \
  "-mllvm -two-entry-phi-node-folding-threshold=0" \
  "-mllvm -two-entry-phi-node-folding-threshold=16" \
  "-mllvm -phi-node-folding-threshold=0" \
  "-mllvm -phi-node-folding-threshold=8" \
  "-mllvm -amdgpu-set-wave-priority=1" \
  "-mllvm -disable-block-placement" \
  "-mllvm -tail-dup-size=0" \
  "-mllvm -tail-dup-size=8" \
  "-mllvm -tail-dup-placement=false" \
  "-mllvm -disable-machine-licm" \
  "-mllvm -disable-machine-sink" \
  "-mllvm -amdgpu-opt-exec-mask-pre-ra=0" \
  "-mllvm -amdgpu-early-ifcvt=1" \
  "-mllvm -structurizecfg-skip-uniform-regions=1" \
  "-mllvm -amdgpu-disable-unclustered-high-rp-reschedule=1" \
  "-mllvm -amdgpu-use-amdgpu-trackers=1" \
  "-mllvm -amdgpu-sched-strategy=max-ilp" \
  "-mllvm -amdgpu-sched-strategy=max-memory-clause" \
  "-mllvm -amdgpu-sched-strategy=iterative-minreg" \
  "-mllvm -amdgpu-sched-strategy=iterative-ilp" \
  "-mllvm -misched-cluster=false" \
  "-mllvm -amdgpu-max-memory-clause=1" \
  "-mllvm -jump-threading-threshold=0" \
  "-mllvm -simplifycfg-sink-common=false" \
  "-mllvm -simplifycfg-hoist-common=false" \
  "-mllvm -enable-gvn-hoist=1" \
  "-mllvm -enable-gvn-sink=1" \
  "-O2" \
  "-Os"
