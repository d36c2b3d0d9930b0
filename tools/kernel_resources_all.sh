#!/bin/bash
# VGPRs / scratch / occupancy of every kernel of the product library: recompiles every .hip source with -Rpass-analysis=kernel-resource-usage
# (objects go to a temp dir; the product build is untouched).  usage: bash tools/kernel_resources_all.sh > profiles/rNN_kernel_resources.txt
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/openmeters_amd/csrc; TMP=$(mktemp -d)
echo "# VGPRs / scratch bytes per lane / occupancy (waves per SIMD) of every kernel of libomx_hip.so: hipcc -Rpass-analysis=kernel-resource-usage on the sources of commit $(git -C $ROOT rev-parse --short HEAD) (gfx950, the Makefile's flags), tools/kernel_resources_all.sh"
for src in $C/*.hip; do
  case $(basename $src) in stft4096_pair_kernels.hip) continue;; esac  # tuning library only (Makefile: ifeq TUNING)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -I$C -x hip -c $src -o $TMP/x.o -Rpass-analysis=kernel-resource-usage > $TMP/$(basename $src).log 2>&1
done
python3 - $TMP <<'PY'
import glob, re, subprocess, sys
rows = {}
for path in sorted(glob.glob(sys.argv[1] + "/*.log")):
    name = None
    for line in open(path):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            rows[name] = {}
        for key in ("VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]"):
            m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
            if m and name:
                rows[name][key.split(" ")[0]] = int(m.group(1))
for name in sorted(rows):
    r = rows[name]
    print(f"{name:<90} vgpr {r.get('VGPRs', 0):4d} scratch {r.get('ScratchSize', 0):4d} occ {r.get('Occupancy', 0)}")
PY
rm -rf $TMP
