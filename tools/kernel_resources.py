"""Kernel name / VGPRs / scratch / occupancy from hipcc's -Rpass-analysis=kernel-resource-usage log (tools/build_ab.sh writes one per object).
usage: python tools/kernel_resources.py ab_libs/obj/<tag>_<file>.o.log [other.log: prints only the kernels that differ]"""
import re
import subprocess
import sys


def parse(path):
    out, name = {}, None
    for line in open(path):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            out[name] = {}
        for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
            if m and name:
                out[name][key.split(" ")[0]] = int(m.group(1))
    return out


a = parse(sys.argv[1])
b = parse(sys.argv[2]) if len(sys.argv) > 2 else None
for k, v in a.items():
    if b is None:
        print(f"{k:90s} vgpr {v.get('VGPRs'):4d} scratch {v.get('ScratchSize'):4d} occ {v.get('Occupancy')}")
    elif b.get(k) != v:
        w = b.get(k, {})
        print(f"{k:90s} vgpr {v.get('VGPRs')} -> {w.get('VGPRs')}  scratch {v.get('ScratchSize')} -> {w.get('ScratchSize')}  occ {v.get('Occupancy')} -> {w.get('Occupancy')}")
