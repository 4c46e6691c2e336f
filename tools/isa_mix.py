#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc -S output: python tools/isa_mix.py file.s kernel_substring [--loops]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2]
lines = s.split("\n")
start = None
for i, l in enumerate(lines):
    if l.endswith(":") and pat in l and not l.startswith(".") and not l.startswith("\t"):
        start = i
        break
    m = re.match(r"^(\S+):\s+; @", l)
    if m and pat in m.group(1):
        start = i
        break
if start is None:
    sys.exit("kernel not found")
ins = []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") or re.match(r"^\.?L?BB", t):
        continue
    ins.append(t.split()[0])
c = Counter(ins)
print("total", len(ins), "valu", sum(v for k, v in c.items() if k.startswith("v_")), "salu", sum(v for k, v in c.items() if k.startswith("s_")),
      "lds", sum(v for k, v in c.items() if k.startswith("ds_")), "vmem", sum(v for k, v in c.items() if k.startswith("global_") or k.startswith("buffer_") or k.startswith("scratch_")))
print(c.most_common(50))
