#!/bin/bash
# fourth round: the four-wave specialised kernel (128 VGPRs, VALU busy 98 %) under the options again, and source-form switches
python3 tools/jit_opts.py \
  "-DRT_NO_PROBE=1" \
  "-DRT_SORTED_CARRY_RADIANCE=1" \
  "-mllvm -enable-misched=false" \
  "-mllvm -amdgpu-sched-strategy=max-ilp" \
  "-mllvm -sink-insts-to-avoid-spills=1" \
  "-mllvm -disable-machine-licm" \
  "-mllvm -tail-dup-placement=false" \
  "-mllvm -phi-node-folding-threshold=0" \
  "-mllvm -two-entry-phi-node-folding-threshold=0" \
  "-mllvm -enable-pre=0" \
  "-mllvm -enable-gvn-memdep=0" \
  "-mllvm -amdgpu-opt-exec-mask-pre-ra=0" \
  "-mllvm -jump-threading-threshold=0" \
  "-mllvm -simplifycfg-sink-common=false" \
  "-fno-unroll-loops" \
  "-mllvm -unroll-threshold=1000" \
  "-mllvm -amdgpu-schedule-metric-bias=0" \
  "-mllvm -amdgpu-schedule-relaxed-occupancy=true" \
  "-mllvm -split-spill-mode=size" \
  "-mllvm -greedy-reverse-local-assignment=1" \
  "-DRT_XCH_PARTS=3" \
  "-mllvm -amdgpu-use-divergent-register-indexing=1" \
  "-mllvm -enable-tail-merge=0" \
  "-mllvm -enable-shrink-wrap=0"
