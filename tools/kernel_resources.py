#!/usr/bin/env python3
"""Per-kernel register / spill / scratch usage of the product's code objects, from `hipcc -S` (no GPU needed):
python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bgreat_amd", "csrc")
print("# per-kernel resource usage of the shipped code objects (hipcc -O3 -S --offload-arch=gfx950)")
print("# kernel | vgpr_count | sgpr_count | vgpr_spill_count | sgpr_spill_count | scratch bytes per lane")
for f in ("greedy_kernels.hip", "exhaustive_kernels.hip", "anchors_kernel.hip", "batch_kernels.hip", "text_kernels.hip"):
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-S", "--cuda-device-only",
                               os.path.join(SRC, f), "-o", t.name])
        s = open(t.name).read()
    print("## " + f)
    for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?"
                         r"\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", s, re.S):
        try:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip() or m.group(1)
        except OSError:
            name = m.group(1)
        name = re.sub(r"\(BgrDeviceGraph.*$|\(unsigned.*$|\(HIP_vector.*$", "", name).replace("void bgr::(anonymous namespace)::", "").replace("bgr::", "")
        print("%-52s vgpr %3s sgpr %3s vgpr_spill %2s sgpr_spill %2s scratch %3s" % (name, m.group(5), m.group(3), m.group(6), m.group(4), m.group(2)))
