#!/usr/bin/env python3
"""isa_phases.py <kernel.s> -- split a kernel disassembly at its s_memtime stamps (the WMX_*_PROF developer builds put one
at every phase boundary) and print, per phase, the instruction count by pipe and the measured-class histogram of
tools_dev/issue_model.py.  Static view of where a block's instructions go; no GPU needed."""
import collections
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from issue_model import classify  # noqa: E402

phases, cur = [], []
for line in open(sys.argv[1]):
    f = line.split()
    if not f or not f[0][0].isalpha() or f[0].endswith(":"):
        continue
    if f[0].startswith("s_memtime"):
        phases.append(cur)
        cur = []
        continue
    cur.append(f[0])
phases.append(cur)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for i, p in enumerate(phases):
    pipes = collections.Counter()
    cls = collections.Counter()
    for m in p:
        c = classify(m)
        if c:
            pipes["valu"] += 1
            cls[c] += 1
        else:
            pipes["salu" if m.startswith("s_") else ("lds" if m.startswith("ds_") else ("vmem" if m.startswith(("global", "buffer", "flat", "scratch")) else "other"))] += 1
    if len(p) < 6:
        continue
    print("segment %2d: %4d instr  valu %4d salu %4d lds %3d vmem %3d | %s" % (i, len(p), pipes["valu"], pipes["salu"], pipes["lds"], pipes["vmem"],
          " ".join("%s:%d" % (k.replace("v_", ""), v) for k, v in cls.most_common(top))))
