#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel (VGPRs, AGPRs, scratch, occupancy, LDS).
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> res.txt ; tools/kernel_resources.py res.txt [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names, rows = [], []
for b in re.split(r'(?=remark: [^\n]*Function Name)', txt):
    m = re.search(r'Function Name: (\S+)', b)
    if not m:
        continue
    def g(k):
        mm = re.search(k + r': (\d+)', b)
        return mm.group(1) if mm else '?'
    names.append(m.group(1))
    rows.append((g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]'), g('SGPRs')))
dem = subprocess.run(['c++filt'], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for d, r in zip(dem, rows):
    d = re.sub(r'\(anonymous namespace\)::', '', d)
    d = re.sub(r'\(RtSceneView.*', '', d)
    if flt and flt not in d:
        continue
    print(f"{d[:120]:120s} VGPR {r[0]:>3} AGPR {r[1]:>3} SGPR {r[5]:>3} scratch {r[2]:>4} occ {r[3]} LDS {r[4]}")
