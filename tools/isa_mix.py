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
# issue classes measured by tools/ubench/valu_rates (profiles/r02_valu_rates.txt): these VOP1/VOP2 encodings issue a wave64
# instruction in ~2.2 cycles when the SIMD holds >= 2 waves; every other vector instruction measured takes ~4.1
FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_mov_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_xnor_b32")


def is_fast(op):
    base = op.replace("_e32", "")
    return base in FAST and not op.endswith("_dpp") and not op.endswith("_sdwa") and not op.endswith("_e64")


valu = {k: v for k, v in c.items() if k.startswith("v_")}
nf = sum(v for k, v in valu.items() if is_fast(k))
nv = sum(valu.values())
if nv:
    print("valu issue classes (static): %d of %d in the ~2.2-cycle class (%.1f %%); mean %.2f cycles per instruction"
          % (nf, nv, 100.0 * nf / nv, (nf * 2.2 + (nv - nf) * 4.1) / nv))
print("total", len(ins), "valu", sum(v for k, v in c.items() if k.startswith("v_")), "salu", sum(v for k, v in c.items() if k.startswith("s_")),
      "lds", sum(v for k, v in c.items() if k.startswith("ds_")), "vmem", sum(v for k, v in c.items() if k.startswith("global_") or k.startswith("buffer_") or k.startswith("scratch_")))
print(c.most_common(50))
