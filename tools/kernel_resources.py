"""Registers, spills, scratch and occupancy of every kernel of rt_hip.hip as hipcc reports them (-Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracing_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-I../../include", "-c", "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/rt_dev.o", "rt_hip.hip"] + sys.argv[1:]
text = subprocess.run(cmd, cwd=here, capture_output=True, text=True).stderr
for block in re.split(r"remark: Function Name: ", text)[1:]:
    name = subprocess.run(["c++filt", block.split(" ")[0]], capture_output=True, text=True).stdout.strip().split("(")[0]
    get = lambda key: re.search(key + r": (\d+)", block).group(1)
    print("%-64s VGPRs %4s  spilled %3s  scratch %4s B/lane  waves/SIMD %s" % (name, get("VGPRs"), get("VGPRs Spill"), get(r"ScratchSize \[bytes/lane\]"), get(r"Occupancy \[waves/SIMD\]")))
