#!/usr/bin/env python3
"""Where are the scratch (spill) instructions of a kernel?  Reads `hipcc -S --cuda-device-only` output, finds the loops
(backward branches) and prints, per loop, its line range, instruction count and the scratch loads / stores directly inside it.
usage: tools/asm_loops.py file.s [kernel-symbol]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2] if len(sys.argv) > 2 else None
start, end = 0, len(lines)
if sym:
    for i, l in enumerate(lines):
        if l.startswith(sym + ":"):
            start = i
        if start and l.strip().startswith(".end_amdhsa_kernel") is False and l.startswith("\t.size\t" + sym):
            end = i
            break
labels = {}
for i in range(start, end):
    m = re.match(r'^(\.LBB\d+_\d+):', lines[i])
    if m:
        labels[m.group(1)] = i
loops = []
for i in range(start, end):
    m = re.match(r'^\s+s_(c?branch\w*)\s+(\.LBB\d+_\d+)', lines[i])
    if m and m.group(2) in labels and labels[m.group(2)] <= i:
        loops.append((labels[m.group(2)], i))
loops = sorted(set(loops))
def is_inst(l):
    return l.startswith("\t") and not l.strip().startswith((".", ";")) and l.strip()
def innermost(i):
    best = None
    for (a, b) in loops:
        if a <= i <= b and (best is None or (b - a) < (best[1] - best[0])):
            best = (a, b)
    return best
from collections import Counter
ld, st, n_inst = Counter(), Counter(), Counter()
for i in range(start, end):
    l = lines[i]
    if not is_inst(l):
        continue
    lp = innermost(i)
    n_inst[lp] += 1
    if "scratch_load" in l:
        ld[lp] += 1
    if "scratch_store" in l:
        st[lp] += 1
print("loop(lines)            insts  scratch_ld scratch_st   nesting-parent")
for lp in [None] + loops:
    par = None
    if lp:
        for (a, b) in loops:
            if a <= lp[0] and lp[1] <= b and (a, b) != lp and (par is None or (b - a) < (par[1] - par[0])):
                par = (a, b)
    print(f"{str(lp):22s} {n_inst[lp]:6d} {ld[lp]:8d} {st[lp]:8d}     {par}")
