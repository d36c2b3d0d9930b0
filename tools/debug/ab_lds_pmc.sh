# LDS counters of the headline kernel for every ab_libs/libomx_*.so on one box — usage: gpurun -- bash tools/debug/ab_lds_pmc.sh
cd /tmp && export TMPDIR=/tmp
for lib in $GRAFT_REPO_ROOT/ab_libs/libomx_*.so; do
  tag=$(basename $lib .so)
  export OMX_HIP_LIB=$lib
  OUT=$GRAFT_REPO_ROOT/gpurun_out/ablds_$tag
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-spectrum --no-cpu-baseline --no-secondary > $OUT/log.txt 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "tri_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print("$tag", {k: f"{v:.4g}" for k, v in sorted(m.items())})
if m.get("SQ_LDS_IDX_ACTIVE"):
    print("   conflict / idx_active = %.3f   wait_lds / wave_cycles = %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], m["SQ_WAIT_INST_LDS"] / m["SQ_WAVE_CYCLES"]))
PY
done
