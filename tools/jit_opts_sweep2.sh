#!/bin/bash
# second round: combinations of the options that paid in tools/jit_opts_sweep.sh
S="-mllvm -structurizecfg-skip-uniform-regions=1"
P="-mllvm -phi-node-folding-threshold=0"
H="-mllvm -simplifycfg-hoist-common=false"
M="-mllvm -enable-misched=false"
T="-mllvm -tail-dup-placement=false"
python3 tools/jit_opts.py "$S" "$S $P" "$S $H" "$S $M" "$S $T" "$S $P $H" "$S $P $H $M" "$S $P $H $M $T" "$S -mllvm -structurizecfg-relaxed-uniform-regions=1" "$S -mllvm -phi-node-folding-threshold=1" "$S -mllvm -amdgpu-sched-strategy=max-ilp" "$S $P -mllvm -two-entry-phi-node-folding-threshold=0" "$S"
