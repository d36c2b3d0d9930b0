#!/bin/bash
# A/B the K2 kernel forms through the bench (run on the GPU box): OMX_K2_FORM = 0 three-workgroups-per-CU kernel (the product's), 2 pair kernel.
# The product library ignores tuning variables: this needs the tuning build (make -C openmeters_amd/csrc TUNING=1).
export OMX_HIP_LIB=${OMX_HIP_LIB:-$PWD/openmeters_amd/csrc/libomx_hip_tuning.so}
[ -f "$OMX_HIP_LIB" ] || { echo "k2_forms.sh: $OMX_HIP_LIB not built (make -C openmeters_amd/csrc TUNING=1)" >&2; exit 1; }
for rep in 1 2; do
for f in ${FORMS:-0 2}; do
  OMX_K2_FORM=$f python bench.py --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('form $f', round(d['value']/1e6,2), 'Mframes/s ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4))"
done
done
