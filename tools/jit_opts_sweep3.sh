#!/bin/bash
# third round (on top of the adopted -structurizecfg-skip-uniform-regions): more backend / mid-end options
python3 tools/jit_opts.py \
  "-mllvm -amdgpu-opt-vgpr-liverange=0" \
  "-mllvm -amdgpu-enable-pre-ra-optimizations=0" \
  "-mllvm -amdgpu-remove-redundant-endcf=0" \
  "-mllvm -amdgpu-scalar-ir-passes=0" \
  "-mllvm -amdgpu-early-inline-all=1" \
  "-mllvm -amdgpu-enable-rewrite-partial-reg-uses=0" \
  "-mllvm -amdgpu-use-divergent-register-indexing=1" \
  "-mllvm -amdgpu-codegenprepare-expand-div64=1" \
  "-mllvm -amdgpu-late-structurize=1" \
  "-mllvm -amdgpu-si-fold-operands=0" \
  "-fno-unroll-loops" \
  "-mllvm -unroll-threshold=1000" \
  "-mllvm -unroll-threshold=50" \
  "-fno-slp-vectorize" \
  "-fno-vectorize" \
  "-mllvm -sink-freq-percent-threshold=0" \
  "-mllvm -machine-sink-split=0" \
  "-mllvm -disable-early-ifcvt" \
  "-mllvm -enable-tail-merge=0" \
  "-mllvm -branch-fold-placement=0" \
  "-mllvm -enable-shrink-wrap=0" \
  "-mllvm -amdgpu-waitcnt-forcezero=0" \
  "-mllvm -amdgpu-atomic-optimizer-strategy=None" \
  "-mllvm -enable-loop-simplifycfg-term-folding=0" \
  "-mllvm -licm-control-flow-hoisting=1" \
  "-mllvm -enable-gvn-memdep=0" \
  "-mllvm -enable-pre=0" \
  "-mllvm -enable-load-pre=0" \
  "-mllvm -aggressive-instcombine-max-scan-instrs=0" \
  "-mllvm -speculative-execution-max-speculation-cost=0" \
  "-mllvm -spec-exec-max-speculation-cost=0" \
  "-mllvm -spec-exec-max-speculation-cost=100" \
  "-mllvm -simplifycfg-merge-cond-stores=0" \
  "-mllvm -amdgpu-dpp-combine=0" \
  "-mllvm -amdgpu-sdwa-peephole=0"
