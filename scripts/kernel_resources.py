#!/usr/bin/env python3
"""Register / scratch / occupancy table of the HIP kernels from hipcc's -Rpass-analysis=kernel-resource-usage
(no GPU needed). usage: kernel_resources.py [substring filter ...]  [-D...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
filters = [a for a in sys.argv[1:] if not a.startswith("-D")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", *defs, "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "ryujin_amd", "csrc"), os.path.join(ROOT, "ryujin_amd", "csrc", "ryujin_hip.hip"),
       "-L/opt/rocm/lib", "-lrccl", "-o", "/tmp/kernel_resources.so"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
names = re.findall(r"remark: Function Name: (\S+)", txt)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for name, d, block in zip(names, dem, re.split(r"remark: Function Name: ", txt)[1:]):
    short = d.split("(")[0].replace("void ryujin_hip::", "").replace("ryujin_hip::", "")
    if filters and not any(f in short for f in filters):
        continue
    g = lambda k: (re.search(k + r": (\d+)", block) or [None, ""])[1]  # noqa: E731
    print("%-78s vgpr %4s agpr %4s scratch %4s occ %s" % (short[:78], g("VGPRs"), g("AGPRs"),
                                                          g(r"ScratchSize \[bytes/lane\]"),
                                                          g(r"Occupancy \[waves/SIMD\]")))
