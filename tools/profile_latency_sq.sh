bash tools/latency_c.sh >/dev/null 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/wf_sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -o p -- /tmp/omx_latency_c > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_INSTS_SMEM --output-format csv -d $OUT/pmc2 -o p -- /tmp/omx_latency_c > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH --output-format csv -d $OUT/pmc3 -o p -- /tmp/omx_latency_c > $OUT/pmc3.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("pmc1","pmc2","pmc3"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]
            if "waveform" in k or "loudness_roles" in k or "stereometer_roles" in k:
                acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k in acc:
        print(d,k,{c: round(v/n[(k,c)],1) for c,v in acc[k].items()})
PY
